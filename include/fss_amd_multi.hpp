// fss_amd_multi.hpp -- one evaluation job across the GPUs of a node, for the C++ consumer (included by fss_amd.hpp).
//
// The reference's launchers evaluate on one device (include/fss/point_eval_gpu.cuh:448-460, eval_all_gpu.cuh:478-488;
// src/bench_gpu.cu:604-627 is the call pattern); its batch -- points x keys -- and its domain split without any exchange
// (SURVEY.md 8e), so a node's GPUs each take a slice:
//   * fss::gpu::BatchSlice / DomainBlock / GrottoCarries: the split, the library's one definition
//     (include/fss_amd.h: fss_amd_batch_slice / _domain_block / _grotto_carries; fss_amd/shard.py calls the same);
//   * fss::gpu::DeviceSet: one engine context and one stream per member device and, for every call, one host thread
//     per member that makes the device current (hipSetDevice) before anything else, uploads its slice of the inputs,
//     launches, and brackets the kernels with events.  Outputs stay in the producing device's memory; a call returns
//     one Shard per member (device, slice, device pointer, kernel milliseconds) and the maximum over the members --
//     the time the job took.  No collective: Grotto's full-domain evaluation, the one result that crosses block
//     boundaries, exchanges `world` bytes through the host (GrottoCarries).
// A device may be listed more than once (then its members share it: eight members on one GPU produce the bytes of an
// eight-GPU run -- tests/cxx/multi_gpu_test.cpp).  Keys, seeds and points come from HOST memory in the reference's own
// formats (key-major Cw arrays as Gen writes them); each member's buffers are the DeviceSet's, grow on demand and stay
// valid until the next call on the set.  One call at a time per DeviceSet.
// The dealer side shards the same way: GenKeysSharded generates member r's slice of a key batch ON member r, straight
// into the level-major layout (fss_amd_*_gen_level_major: no key-major intermediate, no relayout, no key upload -- 32 B
// of seeds, alpha and beta per key go up instead of in_bits x 32 B of correction words), and keeps it resident;
// EvalResidentKeysSharded then evaluates points on those keys for either party (what src/bench_gpu.cu:72-83 + :616-627
// do on one device: Gen, relayout, EvalPoint).
// A point batch can stay resident the same way: UploadPointsSharded cuts it once, EvalResidentPointsSharded evaluates
// key after key on the members' slices.
#pragma once
#ifndef FSS_AMD_HPP_INCLUDED_
#error "include fss_amd.hpp (it includes this header)"
#endif

namespace fss {
namespace gpu {

struct Slice {
  int64_t lo, hi;
};
inline Slice BatchSlice(int64_t n, int rank, int world) {
  Slice s{0, 0};
  Check(fss_amd_batch_slice(n, rank, world, &s.lo, &s.hi), "fss_amd_batch_slice");
  return s;
}
struct Block {
  int64_t first, nleaves;
};
inline Block DomainBlock(int in_bits, int rank, int world) {
  Block b{0, 0};
  Check(fss_amd_domain_block(in_bits, rank, world, &b.first, &b.nleaves), "fss_amd_domain_block");
  return b;
}
inline std::vector<uint8_t> GrottoCarries(const std::vector<uint8_t>& shard_last) {
  std::vector<uint8_t> c(shard_last.size());
  Check(fss_amd_grotto_carries(shard_last.data(), static_cast<int>(shard_last.size()), c.data()), "fss_amd_grotto_carries");
  return c;
}

class DeviceSet {
 public:
  struct Shard {
    int device;       // where `ys` lives
    int64_t first;    // first point / key / leaf of this member's slice
    int64_t count;    // elements in it
    void* ys;         // device pointer: int4[count] (Grotto: uint8_t[count]); valid until the set's next call
    float ms;         // this member's kernels, from events on its stream (uploads and relayout are outside)
  };
  struct Result {
    std::vector<Shard> shards;
    float max_ms = 0.f;  // the job: the slowest member
    template <typename T>
    std::span<T> Span(size_t r) const { return std::span<T>(static_cast<T*>(shards[r].ys), static_cast<size_t>(shards[r].count)); }
  };

  // One member per entry of `devices`, each with a context made from `scheme`'s PRG (keys / nonce), output-group
  // modulus and hash key -- the calls below must be made with schemes of that same key material.
  template <typename Scheme>
  DeviceSet(const Scheme& scheme, const std::vector<int>& devices) {
    if (devices.empty()) throw Error("DeviceSet: no devices", FSS_AMD_EINVAL);
    const detail::Spec spec = SpecOfScheme(scheme);
    int prev = 0;
    CheckHip(hipGetDevice(&prev), "hipGetDevice");
    try {
      for (int d : devices) {
        CheckHip(hipSetDevice(d), "hipSetDevice");
        auto m = std::make_unique<Member>();
        m->device = d;
        Check(fss_amd_ctx_create(&m->ctx, d, spec.aes, spec.hash_key), "fss_amd_ctx_create");
        if (spec.prg == FSS_AMD_PRG_CHACHA) Check(fss_amd_ctx_set_prg(m->ctx, spec.prg, spec.nonce, spec.rounds), "fss_amd_ctx_set_prg");
        if (spec.mod) Check(fss_amd_ctx_set_uint_modulus(m->ctx, spec.mod), "fss_amd_ctx_set_uint_modulus");
        Check(fss_amd_ctx_set_points_plan(m->ctx, FSS_AMD_POINTS_AUTO), "fss_amd_ctx_set_points_plan");  // as the value types' contexts
        CheckHip(hipStreamCreate(&m->stream), "hipStreamCreate");
        CheckHip(hipEventCreate(&m->ev0), "hipEventCreate");
        CheckHip(hipEventCreate(&m->ev1), "hipEventCreate");
        members_.push_back(std::move(m));
      }
    } catch (...) {
      Destroy();
      (void)hipSetDevice(prev);
      (void)hipGetLastError();  // e.g. a device that does not exist: do not leave the error for the caller's next launch check
      throw;
    }
    CheckHip(hipSetDevice(prev), "hipSetDevice");
  }
  // every GPU the process can see, once each
  template <typename Scheme>
  static DeviceSet AllDevices(const Scheme& scheme) {
    int n = 0;
    CheckHip(hipGetDeviceCount(&n), "hipGetDeviceCount");
    std::vector<int> d(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) d[static_cast<size_t>(i)] = i;
    return DeviceSet(scheme, d);
  }
  DeviceSet(const DeviceSet&) = delete;
  DeviceSet& operator=(const DeviceSet&) = delete;
  DeviceSet(DeviceSet&& o) noexcept : members_(std::move(o.members_)) {}
  ~DeviceSet() { Destroy(); }
  int size() const { return static_cast<int>(members_.size()); }
  // FSS_AMD_POINTS_PLAIN / _AUTO for every member (fss_amd_ctx_set_points_plan)
  void SetPointsPlan(int plan) {
    for (auto& m : members_) Check(fss_amd_ctx_set_points_plan(m->ctx, plan), "fss_amd_ctx_set_points_plan");
  }

  // ---- one key, a batch of points cut by BatchSlice: ys[i] = Eval(b, s0, cws, xs[i]) -------------------------------
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalPointsSharded(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                           const typename Dpf<in_bits, Group, Prg, In, pd>::Cw* cws, const In* xs, int64_t n) {
    return Points(FSS_AMD_DPF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, xs, sizeof(In), n);
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalPointsSharded(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, int4 s0,
                           const typename Dcf<in_bits, Group, Prg, In, pred, pd>::Cw* cws, const In* xs, int64_t n) {
    return Points(FSS_AMD_DCF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, xs, sizeof(In), n);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalPointsSharded(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                           const typename HalfTreeDpf<in_bits, Group, Prg, In, pd>::Cw* cws, int4 ocw, const In* xs, int64_t n) {
    return Points(FSS_AMD_HT, in_bits, Group::kId, in_bits, b, s0, cws, &ocw, xs, sizeof(In), n);
  }

  // ---- the same with the points resident: UploadPointsSharded cuts a batch by BatchSlice and leaves member r's slice on
  // its device (until the next upload); EvalResidentPointsSharded then evaluates a key on it -- only the key (about a
  // KiB) crosses PCIe per call: many keys over one set of points.
  template <typename In>
  void UploadPointsSharded(const In* xs, int64_t n) {
    UploadPoints(xs, sizeof(In), n);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalResidentPointsSharded(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                                   const typename Dpf<in_bits, Group, Prg, In, pd>::Cw* cws) {
    return Points(FSS_AMD_DPF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, nullptr, sizeof(In), 0);
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalResidentPointsSharded(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, int4 s0,
                                   const typename Dcf<in_bits, Group, Prg, In, pred, pd>::Cw* cws) {
    return Points(FSS_AMD_DCF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, nullptr, sizeof(In), 0);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalResidentPointsSharded(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                                   const typename HalfTreeDpf<in_bits, Group, Prg, In, pd>::Cw* cws, int4 ocw) {
    return Points(FSS_AMD_HT, in_bits, Group::kId, in_bits, b, s0, cws, &ocw, nullptr, sizeof(In), 0);
  }

  // ---- independent keys cut by BatchSlice: ys[k] = Eval(b, seeds[k], key k, xs[k]) ---------------------------------
  // seeds: the evaluating party's seed of every key; cws: key-major, as Gen writes them.  Each member relays its slice
  // level-major on its device (outside the timed events, as src/bench_gpu.cu:616-627 excludes relayout).
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalKeysSharded(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, const int4* seeds,
                         const typename Dpf<in_bits, Group, Prg, In, pd>::Cw* cws, const In* xs, int64_t nkeys) {
    return Keys(FSS_AMD_DPF, in_bits, Group::kId, in_bits + 1, b, seeds, cws, nullptr, xs, sizeof(In), nkeys);
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalKeysSharded(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, const int4* seeds,
                         const typename Dcf<in_bits, Group, Prg, In, pred, pd>::Cw* cws, const In* xs, int64_t nkeys) {
    return Keys(FSS_AMD_DCF, in_bits, Group::kId, in_bits + 1, b, seeds, cws, nullptr, xs, sizeof(In), nkeys);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalKeysSharded(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, const int4* seeds,
                         const typename HalfTreeDpf<in_bits, Group, Prg, In, pd>::Cw* cws, const int4* ocws, const In* xs,
                         int64_t nkeys) {
    return Keys(FSS_AMD_HT, in_bits, Group::kId, in_bits, b, seeds, cws, ocws, xs, sizeof(In), nkeys);
  }
  // Grotto's path walk (GrottoDcf::EvalKeysWalk): one byte per key
  template <int in_bits, typename Prg, typename In, int pd>
  Result EvalKeysSharded(const GrottoDcf<in_bits, Prg, In, pd>&, bool b, const int4* seeds,
                         const typename GrottoDcf<in_bits, Prg, In, pd>::Cw* cws, const In* xs, int64_t nkeys) {
    static_assert(in_bits <= 64);
    return Keys(FSS_AMD_GROTTO, in_bits, group::Bytes::kId, in_bits + 1, b, seeds, cws, nullptr, xs, sizeof(In), nkeys);
  }

  // ---- inputs that already live on the members' devices -------------------------------------------------------------
  // The reference's launchers take device pointers (include/fss/point_eval_gpu.cuh:448-460: seeds, cw_s, extra, out_cw,
  // xs, ys all in device memory): a caller whose own kernels produced the points -- or the level-major keys -- ON the
  // GPUs hands every member its part in place.  Entry r of `per_member` belongs to member r and must point into memory
  // of THAT member's device (checked: FSS_AMD_EINVAL otherwise); nothing is uploaded but the kilobyte of a shared key.
  // `ys` null: the member's own output buffer (valid until the set's next call), else the caller's, written in place.
  // Shard.first is the running sum of the members' counts.
  template <typename In>
  struct PointsOn {
    const In* xs = nullptr;  // device: n points
    int64_t n = 0;
    int4* ys = nullptr;      // device: n outputs, or null
  };
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalPointsShardedOn(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                             const typename Dpf<in_bits, Group, Prg, In, pd>::Cw* cws, std::span<const PointsOn<In>> per_member) {
    return PointsOnMembers(FSS_AMD_DPF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, Erase(per_member), sizeof(In));
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalPointsShardedOn(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, int4 s0,
                             const typename Dcf<in_bits, Group, Prg, In, pred, pd>::Cw* cws, std::span<const PointsOn<In>> per_member) {
    return PointsOnMembers(FSS_AMD_DCF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr, Erase(per_member), sizeof(In));
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalPointsShardedOn(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                             const typename HalfTreeDpf<in_bits, Group, Prg, In, pd>::Cw* cws, int4 ocw,
                             std::span<const PointsOn<In>> per_member) {
    return PointsOnMembers(FSS_AMD_HT, in_bits, Group::kId, in_bits, b, s0, cws, &ocw, Erase(per_member), sizeof(In));
  }
  // Independent keys, level-major as *RelayoutGpu / fss_amd_*_gen_level_major leave them (the arrays the reference's
  // {Dpf,Dcf,HalfTreeDpf}EvalPointGpu take, point_eval_gpu.cuh:416-491), one point per key.
  template <typename In>
  struct KeysOn {
    const int4* seeds = nullptr;  // the evaluating party's seed of every key
    const int4* cw_s = nullptr;   // [in_bits][nkeys]
    const int4* cw_v = nullptr;   // DCF: [in_bits][nkeys]; otherwise unused
    const void* tr_bits = nullptr;  // DPF / Grotto: uint64_t tr bits per key (two planes when in_bits > 64); Half-Tree: uint32_t extra
    const int4* out_cw = nullptr;   // DPF / DCF: cws[in_bits].s (.v); Half-Tree: ocws; Grotto: unused
    const In* xs = nullptr;
    int64_t nkeys = 0;
    void* ys = nullptr;  // int4[nkeys] (Grotto: uint8_t[nkeys]), or null
  };
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalKeysShardedOn(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, std::span<const KeysOn<In>> per_member) {
    return KeysOnMembers(FSS_AMD_DPF, in_bits, Group::kId, b, EraseKeys(per_member), sizeof(In));
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalKeysShardedOn(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, std::span<const KeysOn<In>> per_member) {
    return KeysOnMembers(FSS_AMD_DCF, in_bits, Group::kId, b, EraseKeys(per_member), sizeof(In));
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalKeysShardedOn(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, std::span<const KeysOn<In>> per_member) {
    return KeysOnMembers(FSS_AMD_HT, in_bits, Group::kId, b, EraseKeys(per_member), sizeof(In));
  }
  template <int in_bits, typename Prg, typename In, int pd>
  Result EvalKeysShardedOn(const GrottoDcf<in_bits, Prg, In, pd>&, bool b, std::span<const KeysOn<In>> per_member) {
    static_assert(in_bits <= 64);
    return KeysOnMembers(FSS_AMD_GROTTO, in_bits, group::Bytes::kId, b, EraseKeys(per_member), sizeof(In));
  }

  // ---- dealer and evaluator on the members: keys generated where they are evaluated --------------------------------
  // s0s: both parties' seeds, two per key (as Gen takes them); alphas; betas (Grotto: none).  Member r generates keys
  // [lo, hi) = BatchSlice(nkeys, r, size()) level-major on its device and keeps them (until the next GenKeysSharded).
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  void GenKeysSharded(const Dpf<in_bits, Group, Prg, In, pd>&, const int4* s0s, const In* alphas, const int4* betas, int64_t nkeys) {
    GenKeys(FSS_AMD_DPF, in_bits, Group::kId, 0, s0s, alphas, sizeof(In), betas, nkeys);
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  void GenKeysSharded(const Dcf<in_bits, Group, Prg, In, pred, pd>&, const int4* s0s, const In* alphas, const int4* betas, int64_t nkeys) {
    GenKeys(FSS_AMD_DCF, in_bits, Group::kId, static_cast<int>(pred), s0s, alphas, sizeof(In), betas, nkeys);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  void GenKeysSharded(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, const int4* s0s, const In* alphas, const int4* betas, int64_t nkeys) {
    GenKeys(FSS_AMD_HT, in_bits, Group::kId, 0, s0s, alphas, sizeof(In), betas, nkeys);
  }
  template <int in_bits, typename Prg, typename In, int pd>
  void GenKeysSharded(const GrottoDcf<in_bits, Prg, In, pd>&, const int4* s0s, const In* alphas, int64_t nkeys) {
    static_assert(in_bits <= 64);
    GenKeys(FSS_AMD_GROTTO, in_bits, group::Bytes::kId, 0, s0s, alphas, sizeof(In), nullptr, nkeys);
  }
  // ys[k] = Eval(b, s0s[2k + b], key k, xs[k]) on the resident keys (xs: host, one point per key; Grotto: the path walk,
  // one byte per key).  The scheme must be the one the keys were generated for.
  template <typename Scheme, typename In>
  Result EvalResidentKeysSharded(const Scheme&, bool b, const In* xs) {
    return EvalResident(b, xs, sizeof(In));
  }

  // ---- the full domain cut by DomainBlock (the set's size must be a power of two): member r holds leaves
  // [r 2^in_bits / world, (r + 1) 2^in_bits / world) --------------------------------------------------------------------
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalAllSharded(const Dpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                        const typename Dpf<in_bits, Group, Prg, In, pd>::Cw* cws) {
    return All(FSS_AMD_DPF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr);
  }
  template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int pd>
  Result EvalAllSharded(const Dcf<in_bits, Group, Prg, In, pred, pd>&, bool b, int4 s0,
                        const typename Dcf<in_bits, Group, Prg, In, pred, pd>::Cw* cws) {
    return All(FSS_AMD_DCF, in_bits, Group::kId, in_bits + 1, b, s0, cws, nullptr);
  }
  template <int in_bits, typename Group, typename Prg, typename In, int pd>
  Result EvalAllSharded(const HalfTreeDpf<in_bits, Group, Prg, In, pd>&, bool b, int4 s0,
                        const typename HalfTreeDpf<in_bits, Group, Prg, In, pd>::Cw* cws, int4 ocw) {
    return All(FSS_AMD_HT, in_bits, Group::kId, in_bits, b, s0, cws, &ocw);
  }
  // GrottoDcf::EvalAll: one byte per leaf; the carries are applied before the call returns (their kernels are not in `ms`)
  template <int in_bits, typename Prg, typename In, int pd>
  Result EvalAllSharded(const GrottoDcf<in_bits, Prg, In, pd>&, bool b, int4 s0,
                        const typename GrottoDcf<in_bits, Prg, In, pd>::Cw* cws) {
    return All(FSS_AMD_GROTTO, in_bits, group::Bytes::kId, in_bits + 1, b, s0, cws, nullptr);
  }

  // Copy a member's outputs to the host (blocking).
  template <typename T>
  void ToHost(const Shard& s, T* dst) const {
    int prev = 0;
    CheckHip(hipGetDevice(&prev), "hipGetDevice");
    CheckHip(hipSetDevice(s.device), "hipSetDevice");
    const hipError_t e = hipMemcpy(dst, s.ys, static_cast<size_t>(s.count) * sizeof(T), hipMemcpyDeviceToHost);
    (void)hipSetDevice(prev);
    CheckHip(e, "hipMemcpy (shard to host)");
  }

 private:
  struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    void* Need(size_t want) {  // the member's device is current
      if (want > bytes || !p) {
        if (p) CheckHip(hipFree(p), "hipFree");
        p = nullptr;
        bytes = 0;
        CheckHip(hipMalloc(&p, want < 256 ? 256 : want), "hipMalloc");
        bytes = want < 256 ? 256 : want;
      }
      return p;
    }
  };
  struct Member {
    int device = 0;
    fss_amd_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Buf in, aux, out;
    // resident keys of GenKeysSharded: seeds (both parties) | cw_s | cw_v (DCF) | tr_bits / extra | out_cw / ocws
    Buf keys;
    int k_scheme = -1, k_in_bits = 0, k_group = 0;
    int64_t k_lo = 0, k_n = 0;
    size_t k_s = 0, k_v = 0, k_t = 0, k_o = 0;  // offsets of the arrays in `keys` (seeds at 0)
    // resident points of UploadPointsSharded
    Buf pts;
    int64_t p_lo = 0, p_n = -1;
    size_t p_xb = 0;
  };
  std::vector<std::unique_ptr<Member>> members_;

  template <typename Scheme>
  static detail::Spec SpecOfScheme(const Scheme& s) {
    detail::Spec spec;
    s.prg.Describe(spec);
    if constexpr (requires { typename Scheme::GroupType; }) spec.mod = Scheme::GroupType::kMod;
    if constexpr (requires { s.hash_key; }) std::memcpy(spec.hash_key, &s.hash_key, sizeof spec.hash_key);
    return spec;
  }
  void Destroy() {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto& m : members_) {
      if (!m) continue;
      (void)hipSetDevice(m->device);
      if (m->stream) (void)hipStreamSynchronize(m->stream);
      for (Buf* b : {&m->in, &m->aux, &m->out, &m->keys, &m->pts})
        if (b->p) (void)hipFree(b->p);
      if (m->ev0) (void)hipEventDestroy(m->ev0);
      if (m->ev1) (void)hipEventDestroy(m->ev1);
      if (m->ctx) fss_amd_ctx_destroy(m->ctx);
      if (m->stream) (void)hipStreamDestroy(m->stream);
    }
    members_.clear();
    (void)hipSetDevice(prev);
  }
  static size_t Up(size_t v) { return (v + 255) & ~size_t(255); }

  // One host thread per member: the device made current first, `fn(member, rank)` enqueues the member's work and returns
  // its shard; the thread then waits for the stream and reads the events.  The first failure is rethrown.
  template <typename F>
  Result Run(F fn) {
    const int world = size();
    Result res;
    res.shards.resize(static_cast<size_t>(world));
    std::vector<std::string> errs(static_cast<size_t>(world));
    std::vector<int> codes(static_cast<size_t>(world), FSS_AMD_OK);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) {
      th.emplace_back([&, r] {
        Member& m = *members_[static_cast<size_t>(r)];
        try {
          CheckHip(hipSetDevice(m.device), "hipSetDevice");
          Shard s = fn(m, r);
          CheckHip(hipStreamSynchronize(m.stream), "hipStreamSynchronize");
          s.device = m.device;
          s.ms = 0.f;
          if (s.count > 0) CheckHip(hipEventElapsedTime(&s.ms, m.ev0, m.ev1), "hipEventElapsedTime");
          res.shards[static_cast<size_t>(r)] = s;
        } catch (const Error& e) {  // the member's own code (an EINVAL refusal stays an EINVAL)
          errs[static_cast<size_t>(r)] = e.what()[0] ? e.what() : "error";
          codes[static_cast<size_t>(r)] = e.code;
        } catch (const std::exception& e) {
          errs[static_cast<size_t>(r)] = e.what()[0] ? e.what() : "error";
          codes[static_cast<size_t>(r)] = FSS_AMD_EHIP;
        }
      });
    }
    for (auto& t : th) t.join();
    for (int r = 0; r < world; ++r)
      if (!errs[static_cast<size_t>(r)].empty()) throw MemberError(r, errs[static_cast<size_t>(r)], codes[static_cast<size_t>(r)]);
    for (const Shard& s : res.shards) res.max_ms = s.ms > res.max_ms ? s.ms : res.max_ms;
    return res;
  }
  // the first failing member's message and CODE (Error's constructor would append the code's text a second time)
  static Error MemberError(int r, const std::string& what, int code) {
    Error e("", code);
    static_cast<std::runtime_error&>(e) = std::runtime_error("DeviceSet member " + std::to_string(r) + ": " + what);
    return e;
  }
  static void H2D(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes) CheckHip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync (slice to device)");
  }

  void UploadPoints(const void* xs, size_t xb, int64_t n) {
    if (n < 0 || (n > 0 && !xs)) throw Error("DeviceSet::UploadPointsSharded: bad arguments", FSS_AMD_EINVAL);
    const int world = size();
    (void)Run([=](Member& m, int r) {
      const Slice sl = BatchSlice(n, r, world);
      const int64_t cnt = sl.hi - sl.lo;
      void* d = m.pts.Need(static_cast<size_t>(cnt) * xb);
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      H2D(d, static_cast<const char*>(xs) + static_cast<size_t>(sl.lo) * xb, static_cast<size_t>(cnt) * xb, m.stream);
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      m.p_lo = sl.lo; m.p_n = cnt; m.p_xb = xb;
      return Shard{m.device, sl.lo, cnt, nullptr, 0.f};
    });
  }

  // xs == nullptr: the members' resident points (UploadPointsSharded; `n` is then theirs)
  Result Points(int scheme, int in_bits, int group, int slots, bool b, int4 s0, const void* cws, const int4* ocw,
                const void* xs, size_t xb, int64_t n) {
    const int world = size();
    return Run([=](Member& m, int r) {
      if (!xs && (m.p_n < 0 || m.p_xb != xb))
        throw Error("DeviceSet::EvalResidentPointsSharded before UploadPointsSharded (or with another point type)", FSS_AMD_EINVAL);
      const Slice sl = xs ? BatchSlice(n, r, world) : Slice{m.p_lo, m.p_lo + m.p_n};
      const int64_t cnt = sl.hi - sl.lo;
      const size_t key_bytes = 32 + size_t(32) * slots + 32;
      char* in = static_cast<char*>(m.in.Need(Up(key_bytes) + (xs ? static_cast<size_t>(cnt) * xb : 0)));
      int4* d_s0 = reinterpret_cast<int4*>(in);
      void* d_cws = in + 32;
      int4* d_ocw = reinterpret_cast<int4*>(in + 32 + 32 * slots);
      void* d_xs = xs ? static_cast<void*>(in + Up(key_bytes)) : m.pts.p;
      void* d_ys = m.out.Need(static_cast<size_t>(cnt) * 16);
      H2D(d_s0, &s0, 16, m.stream);
      H2D(d_cws, cws, size_t(32) * slots, m.stream);
      if (ocw) H2D(d_ocw, ocw, 16, m.stream);
      if (xs) H2D(d_xs, static_cast<const char*>(xs) + static_cast<size_t>(sl.lo) * xb, static_cast<size_t>(cnt) * xb, m.stream);
      CheckHip(hipStreamSynchronize(m.stream), "hipStreamSynchronize");  // s0 / ocw are stack values of this call
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      int rc;
      if (scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, d_xs, static_cast<int>(xb), cnt, d_ys, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, d_xs, static_cast<int>(xb), cnt, d_ys, m.stream);
      else rc = fss_amd_ht_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, d_ocw, d_xs, static_cast<int>(xb), cnt, d_ys, m.stream);
      Check(rc, "fss_amd_*_eval_points");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, sl.lo, cnt, d_ys, 0.f};
    });
  }

  Result Keys(int scheme, int in_bits, int group, int slots, bool b, const int4* seeds, const void* cws, const int4* ocws,
              const void* xs, size_t xb, int64_t nkeys) {
    const int world = size();
    return Run([=](Member& m, int r) {
      const Slice sl = BatchSlice(nkeys, r, world);
      const int64_t cnt = sl.hi - sl.lo;
      const size_t nk = static_cast<size_t>(cnt), key_bytes = size_t(32) * slots;
      const bool grotto = scheme == FSS_AMD_GROTTO;
      // in: seeds | key-major slots | ocws | xs        aux: cw_s | cw_v (DCF) | tr_bits / extra | out_cw
      const size_t o_seeds = 0, o_cws = Up(nk * 16), o_ocw = o_cws + Up(nk * key_bytes), o_xs = o_ocw + Up(nk * 16);
      char* in = static_cast<char*>(m.in.Need(o_xs + Up(nk * xb)));
      const size_t planes = in_bits > 64 ? 2 : 1;
      const size_t a_s = 0, a_v = Up(nk * 16 * in_bits), a_t = a_v + (scheme == FSS_AMD_DCF ? Up(nk * 16 * in_bits) : 0),
                   a_o = a_t + Up(nk * 8 * planes);
      char* aux = static_cast<char*>(m.aux.Need(a_o + Up(nk * 16)));
      void* d_ys = m.out.Need(grotto ? nk : nk * 16);
      if (cnt == 0) return Shard{m.device, sl.lo, 0, d_ys, 0.f};
      H2D(in + o_seeds, seeds + sl.lo, nk * 16, m.stream);
      H2D(in + o_cws, static_cast<const char*>(cws) + static_cast<size_t>(sl.lo) * key_bytes, nk * key_bytes, m.stream);
      if (ocws) H2D(in + o_ocw, ocws + sl.lo, nk * 16, m.stream);
      H2D(in + o_xs, static_cast<const char*>(xs) + static_cast<size_t>(sl.lo) * xb, nk * xb, m.stream);
      int rc;
      if (scheme == FSS_AMD_DPF || grotto) rc = fss_amd_dpf_relayout(m.ctx, in_bits, in + o_cws, cnt, aux + a_s, aux + a_t, aux + a_o, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_relayout(m.ctx, in_bits, in + o_cws, cnt, aux + a_s, aux + a_v, aux + a_o, m.stream);
      else rc = fss_amd_ht_relayout(m.ctx, in_bits, in + o_cws, cnt, aux + a_s, aux + a_t, m.stream);
      Check(rc, "fss_amd_*_relayout");
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      const int xbi = static_cast<int>(xb);
      if (scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_keys(m.ctx, in_bits, group, b, in + o_seeds, aux + a_s, aux + a_t, aux + a_o, in + o_xs, xbi, cnt, d_ys, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_keys(m.ctx, in_bits, group, b, in + o_seeds, aux + a_s, aux + a_v, aux + a_o, in + o_xs, xbi, cnt, d_ys, m.stream);
      else if (scheme == FSS_AMD_HT) rc = fss_amd_ht_eval_keys(m.ctx, in_bits, group, b, in + o_seeds, aux + a_s, aux + a_t, in + o_ocw, in + o_xs, xbi, cnt, d_ys, m.stream);
      else rc = fss_amd_grotto_eval_keys(m.ctx, in_bits, b, in + o_seeds, aux + a_s, aux + a_t, in + o_xs, xbi, cnt, d_ys, m.stream);
      Check(rc, "fss_amd_*_eval_keys");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, sl.lo, cnt, d_ys, 0.f};
    });
  }

  struct RawPoints { const void* xs; int64_t n; void* ys; };
  struct RawKeys { const void *seeds, *cw_s, *cw_v, *tr_bits, *out_cw, *xs; int64_t nkeys; void* ys; };
  template <typename In>
  static std::vector<RawPoints> Erase(std::span<const PointsOn<In>> v) {
    std::vector<RawPoints> o;
    for (const auto& p : v) o.push_back(RawPoints{p.xs, p.n, p.ys});
    return o;
  }
  template <typename In>
  static std::vector<RawKeys> EraseKeys(std::span<const KeysOn<In>> v) {
    std::vector<RawKeys> o;
    for (const auto& k : v) o.push_back(RawKeys{k.seeds, k.cw_s, k.cw_v, k.tr_bits, k.out_cw, k.xs, k.nkeys, k.ys});
    return o;
  }
  // `p` (null allowed) is device memory of the member's device -- a pointer into another GPU's memory would be read
  // across xGMI at a fraction of the rate, or fault: refused instead
  static void OnDevice(const Member& m, const void* p, const char* what) {
    if (!p) return;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
      (void)hipGetLastError();
      throw Error((std::string(what) + ": not a device pointer").c_str(), FSS_AMD_EINVAL);
    }
    if (at.type == hipMemoryTypeHost) throw Error((std::string(what) + ": host memory (use the host-input calls)").c_str(), FSS_AMD_EINVAL);
    if (at.device != m.device) throw Error((std::string(what) + ": memory of device " + std::to_string(at.device) + ", member runs on " + std::to_string(m.device)).c_str(), FSS_AMD_EINVAL);
  }

  Result PointsOnMembers(int scheme, int in_bits, int group, int slots, bool b, int4 s0, const void* cws, const int4* ocw,
                         std::vector<RawPoints> pm, size_t xb) {
    if (static_cast<int>(pm.size()) != size()) throw Error("DeviceSet::EvalPointsShardedOn: one entry per member", FSS_AMD_EINVAL);
    std::vector<int64_t> first(pm.size() + 1, 0);
    for (size_t r = 0; r < pm.size(); ++r) {
      if (pm[r].n < 0 || (pm[r].n > 0 && !pm[r].xs)) throw Error("DeviceSet::EvalPointsShardedOn: bad entry", FSS_AMD_EINVAL);
      first[r + 1] = first[r] + pm[r].n;
    }
    const int4 ocw_v = ocw ? *ocw : int4{0, 0, 0, 0};
    const bool has_ocw = ocw != nullptr;
    return Run([=, &pm, &first](Member& m, int r) {
      const RawPoints& p = pm[static_cast<size_t>(r)];
      OnDevice(m, p.xs, "EvalPointsShardedOn xs");
      OnDevice(m, p.ys, "EvalPointsShardedOn ys");
      const size_t key_bytes = 32 + size_t(32) * slots + 32;
      char* in = static_cast<char*>(m.in.Need(Up(key_bytes)));
      int4* d_s0 = reinterpret_cast<int4*>(in);
      void* d_cws = in + 32;
      int4* d_ocw = reinterpret_cast<int4*>(in + 32 + 32 * slots);
      void* d_ys = p.ys ? p.ys : m.out.Need(static_cast<size_t>(p.n) * 16);
      H2D(d_s0, &s0, 16, m.stream);
      H2D(d_cws, cws, size_t(32) * slots, m.stream);
      if (has_ocw) H2D(d_ocw, &ocw_v, 16, m.stream);
      CheckHip(hipStreamSynchronize(m.stream), "hipStreamSynchronize");  // s0 / ocw are values of this call
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      int rc;
      const int xbi = static_cast<int>(xb);
      if (scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, p.xs, xbi, p.n, d_ys, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, p.xs, xbi, p.n, d_ys, m.stream);
      else rc = fss_amd_ht_eval_points(m.ctx, in_bits, group, b, d_s0, d_cws, d_ocw, p.xs, xbi, p.n, d_ys, m.stream);
      Check(rc, "fss_amd_*_eval_points");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, first[static_cast<size_t>(r)], p.n, d_ys, 0.f};
    });
  }

  Result KeysOnMembers(int scheme, int in_bits, int group, bool b, std::vector<RawKeys> km, size_t xb) {
    if (static_cast<int>(km.size()) != size()) throw Error("DeviceSet::EvalKeysShardedOn: one entry per member", FSS_AMD_EINVAL);
    std::vector<int64_t> first(km.size() + 1, 0);
    const bool grotto = scheme == FSS_AMD_GROTTO;
    for (size_t r = 0; r < km.size(); ++r) {
      const RawKeys& k = km[r];
      const bool need_v = scheme == FSS_AMD_DCF, need_t = scheme != FSS_AMD_DCF, need_o = !grotto;
      if (k.nkeys < 0 || (k.nkeys > 0 && (!k.seeds || !k.cw_s || !k.xs || (need_v && !k.cw_v) || (need_t && !k.tr_bits) || (need_o && !k.out_cw))))
        throw Error("DeviceSet::EvalKeysShardedOn: bad entry", FSS_AMD_EINVAL);
      first[r + 1] = first[r] + k.nkeys;
    }
    return Run([=, &km, &first](Member& m, int r) {
      const RawKeys& k = km[static_cast<size_t>(r)];
      for (const void* p : {k.seeds, k.cw_s, k.cw_v, k.tr_bits, k.out_cw, k.xs, static_cast<const void*>(k.ys)}) OnDevice(m, p, "EvalKeysShardedOn");
      const size_t nk = static_cast<size_t>(k.nkeys);
      void* d_ys = k.ys ? k.ys : m.out.Need(grotto ? nk : nk * 16);
      if (k.nkeys == 0) return Shard{m.device, first[static_cast<size_t>(r)], 0, d_ys, 0.f};
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      const int xbi = static_cast<int>(xb);
      int rc;
      if (scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_keys(m.ctx, in_bits, group, b, k.seeds, k.cw_s, k.tr_bits, k.out_cw, k.xs, xbi, k.nkeys, d_ys, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_keys(m.ctx, in_bits, group, b, k.seeds, k.cw_s, k.cw_v, k.out_cw, k.xs, xbi, k.nkeys, d_ys, m.stream);
      else if (scheme == FSS_AMD_HT) rc = fss_amd_ht_eval_keys(m.ctx, in_bits, group, b, k.seeds, k.cw_s, k.tr_bits, k.out_cw, k.xs, xbi, k.nkeys, d_ys, m.stream);
      else rc = fss_amd_grotto_eval_keys(m.ctx, in_bits, b, k.seeds, k.cw_s, k.tr_bits, k.xs, xbi, k.nkeys, d_ys, m.stream);
      Check(rc, "fss_amd_*_eval_keys");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, first[static_cast<size_t>(r)], k.nkeys, d_ys, 0.f};
    });
  }

  void GenKeys(int scheme, int in_bits, int group, int pred, const int4* s0s, const void* alphas, size_t xb, const int4* betas,
               int64_t nkeys) {
    const int world = size();
    (void)Run([=](Member& m, int r) {
      const Slice sl = BatchSlice(nkeys, r, world);
      const int64_t cnt = sl.hi - sl.lo;
      const size_t nk = static_cast<size_t>(cnt), planes = in_bits > 64 ? 2 : 1;
      m.k_scheme = scheme; m.k_in_bits = in_bits; m.k_group = group; m.k_lo = sl.lo; m.k_n = cnt;
      m.k_s = Up(nk * 32);
      m.k_v = m.k_s + Up(nk * 16 * in_bits);
      m.k_t = m.k_v + (scheme == FSS_AMD_DCF ? Up(nk * 16 * in_bits) : 0);
      m.k_o = m.k_t + Up(nk * 8 * planes);
      char* keys = static_cast<char*>(m.keys.Need(m.k_o + Up(nk * 16)));
      // uploads: alphas | betas (the seeds go straight into their resident place)
      const size_t o_b = Up(nk * xb);
      char* in = static_cast<char*>(m.in.Need(o_b + Up(nk * 16)));
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      if (cnt > 0) {
        H2D(keys, s0s + 2 * sl.lo, nk * 32, m.stream);
        H2D(in, static_cast<const char*>(alphas) + static_cast<size_t>(sl.lo) * xb, nk * xb, m.stream);
        if (betas) H2D(in + o_b, betas + sl.lo, nk * 16, m.stream);
        const void* d_betas = betas ? in + o_b : nullptr;
        const int xbi = static_cast<int>(xb);
        int rc;
        if (scheme == FSS_AMD_DPF || scheme == FSS_AMD_GROTTO)
          rc = fss_amd_dpf_gen_level_major(m.ctx, in_bits, group, keys, in, xbi, d_betas, cnt, keys + m.k_s, keys + m.k_t, keys + m.k_o, m.stream);
        else if (scheme == FSS_AMD_DCF)
          rc = fss_amd_dcf_gen_level_major(m.ctx, in_bits, group, pred, keys, in, xbi, d_betas, cnt, keys + m.k_s, keys + m.k_v, keys + m.k_o, m.stream);
        else
          rc = fss_amd_ht_gen_level_major(m.ctx, in_bits, group, keys, in, xbi, d_betas, cnt, keys + m.k_s, keys + m.k_t, keys + m.k_o, m.stream);
        Check(rc, "fss_amd_*_gen_level_major");
      }
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, sl.lo, cnt, nullptr, 0.f};
    });
  }

  Result EvalResident(bool b, const void* xs, size_t xb) {
    return Run([=](Member& m, int) {
      if (m.k_scheme < 0) throw Error("DeviceSet::EvalResidentKeysSharded before GenKeysSharded", FSS_AMD_EINVAL);
      const int64_t cnt = m.k_n;
      const size_t nk = static_cast<size_t>(cnt);
      const bool grotto = m.k_scheme == FSS_AMD_GROTTO;
      const size_t o_x = Up(nk * 16);
      char* in = static_cast<char*>(m.in.Need(o_x + Up(nk * xb)));  // party b's seeds, contiguous | xs
      char* keys = static_cast<char*>(m.keys.p);
      void* d_ys = m.out.Need(grotto ? nk : nk * 16);
      if (cnt == 0) return Shard{m.device, m.k_lo, 0, d_ys, 0.f};
      CheckHip(hipMemcpy2DAsync(in, 16, keys + (b ? 16 : 0), 32, 16, nk, hipMemcpyDeviceToDevice, m.stream), "hipMemcpy2DAsync (seeds)");
      H2D(in + o_x, static_cast<const char*>(xs) + static_cast<size_t>(m.k_lo) * xb, nk * xb, m.stream);
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      const int xbi = static_cast<int>(xb), n = m.k_in_bits, g = m.k_group;
      int rc;
      if (m.k_scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_keys(m.ctx, n, g, b, in, keys + m.k_s, keys + m.k_t, keys + m.k_o, in + o_x, xbi, cnt, d_ys, m.stream);
      else if (m.k_scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_keys(m.ctx, n, g, b, in, keys + m.k_s, keys + m.k_v, keys + m.k_o, in + o_x, xbi, cnt, d_ys, m.stream);
      else if (m.k_scheme == FSS_AMD_HT) rc = fss_amd_ht_eval_keys(m.ctx, n, g, b, in, keys + m.k_s, keys + m.k_t, keys + m.k_o, in + o_x, xbi, cnt, d_ys, m.stream);
      else rc = fss_amd_grotto_eval_keys(m.ctx, n, b, in, keys + m.k_s, keys + m.k_t, in + o_x, xbi, cnt, d_ys, m.stream);
      Check(rc, "fss_amd_*_eval_keys");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, m.k_lo, cnt, d_ys, 0.f};
    });
  }

  Result All(int scheme, int in_bits, int group, int slots, bool b, int4 s0, const void* cws, const int4* ocw) {
    const int world = size();
    const bool grotto = scheme == FSS_AMD_GROTTO;
    Result res = Run([=](Member& m, int r) {
      const Block bl = DomainBlock(in_bits, r, world);
      const size_t key_bytes = 32 + size_t(32) * slots + 32;
      char* in = static_cast<char*>(m.in.Need(Up(key_bytes)));
      int4* d_s0 = reinterpret_cast<int4*>(in);
      void* d_cws = in + 32;
      int4* d_ocw = reinterpret_cast<int4*>(in + 32 + 32 * slots);
      const size_t ws = fss_amd_eval_all_workspace(scheme, in_bits, bl.nleaves);
      if (ws == 0) throw Error("fss_amd_eval_all_workspace: unsupported block", FSS_AMD_EINVAL);
      void* d_ws = m.aux.Need(ws);
      void* d_ys = m.out.Need(static_cast<size_t>(bl.nleaves) * (grotto ? 1 : 16));
      H2D(d_s0, &s0, 16, m.stream);
      H2D(d_cws, cws, size_t(32) * slots, m.stream);
      if (ocw) H2D(d_ocw, ocw, 16, m.stream);
      CheckHip(hipStreamSynchronize(m.stream), "hipStreamSynchronize");
      CheckHip(hipEventRecord(m.ev0, m.stream), "hipEventRecord");
      int rc;
      if (scheme == FSS_AMD_DPF) rc = fss_amd_dpf_eval_all(m.ctx, in_bits, group, b, d_s0, d_cws, bl.first, bl.nleaves, d_ys, d_ws, ws, m.stream);
      else if (scheme == FSS_AMD_DCF) rc = fss_amd_dcf_eval_all(m.ctx, in_bits, group, b, d_s0, d_cws, bl.first, bl.nleaves, d_ys, d_ws, ws, m.stream);
      else if (scheme == FSS_AMD_HT) rc = fss_amd_ht_eval_all(m.ctx, in_bits, group, b, d_s0, d_cws, d_ocw, bl.first, bl.nleaves, d_ys, d_ws, ws, m.stream);
      else rc = fss_amd_grotto_eval_all(m.ctx, in_bits, b, d_s0, d_cws, bl.first, bl.nleaves, d_ys, d_ws, ws, m.stream);
      Check(rc, "fss_amd_*_eval_all");
      CheckHip(hipEventRecord(m.ev1, m.stream), "hipEventRecord");
      return Shard{m.device, bl.first, bl.nleaves, d_ys, 0.f};
    });
    if (grotto && world > 1) {  // the running XOR crosses the blocks: one byte per member through the host
      std::vector<uint8_t> last(static_cast<size_t>(world));
      int prev = 0;
      CheckHip(hipGetDevice(&prev), "hipGetDevice");
      for (int r = 0; r < world; ++r) {
        const Shard& s = res.shards[static_cast<size_t>(r)];
        CheckHip(hipSetDevice(s.device), "hipSetDevice");
        CheckHip(hipMemcpy(&last[static_cast<size_t>(r)], static_cast<const uint8_t*>(s.ys) + s.count - 1, 1, hipMemcpyDeviceToHost), "hipMemcpy");
      }
      const std::vector<uint8_t> carry = GrottoCarries(last);
      for (int r = 0; r < world; ++r) {
        const Shard& s = res.shards[static_cast<size_t>(r)];
        Member& m = *members_[static_cast<size_t>(r)];
        CheckHip(hipSetDevice(s.device), "hipSetDevice");
        Check(fss_amd_grotto_apply_carry(m.ctx, s.ys, s.count, carry[static_cast<size_t>(r)], m.stream), "fss_amd_grotto_apply_carry");
      }
      for (int r = 0; r < world; ++r) {
        CheckHip(hipSetDevice(res.shards[static_cast<size_t>(r)].device), "hipSetDevice");
        CheckHip(hipStreamSynchronize(members_[static_cast<size_t>(r)]->stream), "hipStreamSynchronize");
      }
      CheckHip(hipSetDevice(prev), "hipSetDevice");
    }
    return res;
  }
};

}  // namespace gpu
}  // namespace fss
