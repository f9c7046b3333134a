// fss_amd.hpp -- the myl7/fss C++ surface (scheme templates, PRG hook, fss::gpu launchers) over the C ABI
// of libfss_amd.so (fss_amd.h).  Header-only, C++20 (the reference is C++20 too); compile with hipcc and
// link with -lfss_amd.  A caller of the reference changes its includes, cuda* -> hip* runtime names and
// cuda::std::{array,span} -> std::{array,span}; the fss:: spellings below are the reference's own:
//
//   Prgable / Groupable (global concepts)  <- include/fss/prg.cuh:20-23, include/fss/group.cuh:39-45
//   fss::util::Xor/SetLsb/GetLsb/Pack      <- include/fss/util.cuh:16-64
//   fss::group::Bytes / Uint<T, mod>       <- include/fss/group/bytes.cuh:19-43, group/uint.cuh:27-88
//   fss::prg::Aes128Mmo<mul>               <- include/fss/prg/aes128_mmo.cuh:27-95: span constructor, CreateCtxs,
//                                             FreeCtxs, Gen(int4) -> array<int4, mul>; behind the EVP_CIPHER_CTX* handles
//                                             are the AES key bytes (AesKeyCtx), not OpenSSL state
//   fss::prg::ChaCha<mul, rounds>          <- include/fss/prg/chacha.cuh:24-128 (constructor from the nonce pointer)
//   fss::Dpf / Dcf / HalfTreeDpf / GrottoDcf
//                                          <- include/fss/dpf.cuh:61-303, dcf.cuh:74-385, half_tree_dpf.cuh:39-354,
//                                             grotto_dcf.cuh:45-238: same template parameters (incl. par_depth, accepted
//                                             and unused: the GPU expands the whole tree in parallel), public `prg`
//                                             (and `hash_key`) members, aggregate initialisation, 32-byte `Cw`, and the
//                                             reference's Gen / Eval / EvalAll signatures.  Gen and Eval are callable
//                                             from host and from device code like the reference's (here a __host__
//                                             and a __device__ overload of each).  Called on the HOST they are
//                                             computed by the engine's kernels (buffers staged through a context-owned
//                                             pool; a device pointer is used in place).  Called from a DEVICE function
//                                             -- one key per thread inside the caller's own kernel, the way
//                                             samples/dpf_dcf_gpu.cu:39-80 uses them -- they run the per-thread walk of
//                                             fss_amd_thread.hpp.  Next to them are the batched device forms the
//                                             engine is built for (GenBatch, GenLevelMajor, EvalPoints, EvalAllBlock).
//   fss::gpu::*RelayoutGpu / *EvalPointGpu / *EvalAllGpu[Batch]
//                                          <- include/fss/point_eval_gpu.cuh:324-491, include/fss/eval_all_gpu.cuh:451-535:
//                                             the reference's template-parameter order (leading `unroll`, `z, b1, bs`;
//                                             accepted and ignored: the engine plans its own launches), uint32_t `extra`
//                                             and uint32_t `xs` for in_bits <= 32, the seed by value and no workspace
//                                             argument; plus overloads with uint64_t tr bits and `In` points for
//                                             in_bits > 32, where the reference's launchers stop.
// User-defined Groupable / Prgable types (members __host__ __device__ and trivially copyable, as the reference
// requires of anything passed to a kernel): the prebuilt kernels cannot execute them, so Gen / Eval / EvalAll /
// GenBatch / EvalPoints instantiate the per-thread walk in generic kernels of this header, compiled with the caller's
// types -- still on the GPU, one thread per key or point, not the tuned path.  So do the fss::gpu launchers (the
// reference's are templates over any PRG passed by value, point_eval_gpu.cuh:448-460, eval_all_gpu.cuh:478-488): on the
// level-major arrays, one key per thread, coalesced.  GenLevelMajor and EvalAllBlock need the built-in types
// (static_assert).  Nothing in this header computes on the CPU.
// Errors throw fss::Error (the reference asserts).
#pragma once
#define FSS_AMD_HPP_INCLUDED_ 1
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <concepts>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <span>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "fss_amd.h"
#include "fss_amd_thread.hpp"

// include/fss/prg.cuh:20-23
template <typename Prg, int mul>
concept Prgable = requires(Prg prg, int4 seed) {
  { prg.Gen(seed) } -> std::same_as<std::array<int4, mul>>;
};

// include/fss/hash.cuh:15-30
template <typename Hash>
concept Hashable = requires(Hash hash, std::span<const int4, 4> msg) {
  { hash.Hash(msg) } -> std::same_as<std::array<int4, 2>>;
};
template <typename Hash>
concept XorHashable = requires(Hash hash, std::tuple<int4, const int4> msg) {
  { hash.Hash(msg) } -> std::same_as<std::array<int4, 4>>;
};
static_assert(Hashable<fss::hash::Blake3> && XorHashable<fss::hash::Blake3> && Hashable<fss::hash::Sha256> && XorHashable<fss::hash::Sha256>);

// include/fss/group.cuh:39-45
template <typename Group>
concept Groupable = std::is_default_constructible_v<Group> && requires(Group lhs, Group rhs, int4 buf) {
  { lhs + rhs } -> std::same_as<Group>;
  { -lhs } -> std::same_as<Group>;
  { Group::From(buf) } -> std::same_as<Group>;
  { lhs.Into() } -> std::same_as<int4>;
};

// The reference's Aes128Mmo hands out OpenSSL cipher contexts and its callers spell their type (`cuda::std::array<
// EVP_CIPHER_CTX*, 2> ctxs` in src/dpf_test.cu:164, src/bench_cpu.cu:27).  OpenSSL declares that type as an opaque struct, and
// so does this header -- the identical declaration, should <openssl/evp.h> be included as well -- so that such call sites
// compile as they stand.  No OpenSSL is included, linked or called: behind the pointers CreateCtxs hands out are
// fss::prg::AesKeyCtx objects (the key bytes), and a pointer that is not one of them is refused (fss::Error).
extern "C" {
typedef struct evp_cipher_ctx_st EVP_CIPHER_CTX;
}

namespace fss {

struct Error : std::runtime_error {
  int code;
  Error(const char* what, int c) : std::runtime_error(std::string(what) + ": " + fss_amd_strerror(c)), code(c) {}
};
inline void Check(int rc, const char* what) {
  if (rc != FSS_AMD_OK) throw Error(what, rc);
}
inline void CheckHip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error((std::string(what) + ": " + hipGetErrorString(e)).c_str(), FSS_AMD_EHIP);
}

// ---------------------------------------------------------------- util (include/fss/util.cuh:16-64)
namespace util {
__host__ __device__ inline int4 Xor(int4 a, int4 b) { return int4{a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
template <size_t N>
__host__ __device__ inline std::array<int4, N> Xor(std::span<const int4, N> a, std::span<const int4, N> b) {
  std::array<int4, N> r;
  for (size_t i = 0; i < N; ++i) r[i] = Xor(a[i], b[i]);
  return r;
}
// the control ("clamped") bit is bit 0 of w
__host__ __device__ inline int4 SetLsb(int4 v, bool bit) {
  v.w = (v.w & ~1) | (bit ? 1 : 0);
  return v;
}
__host__ __device__ inline bool GetLsb(int4 v) { return v.w & 1; }
// util.cuh:40-45: par_depth as given, or ceil(log2(host threads)) for -1.  The reference asks OpenMP for the thread count;
// nothing here runs on OpenMP (EvalAll is one GPU launch), so the value only matters to callers that use it themselves.
inline int ResolveParDepth(int par_depth) {
  if (par_depth >= 0) return par_depth;
  const unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  int d = 0;
  while ((1u << d) < threads) ++d;
  return d;
}
template <typename In>
__host__ __device__ inline int4 Pack(In val) {
  const unsigned __int128 v = static_cast<unsigned __int128>(val);
  int4 r{static_cast<int>(static_cast<uint32_t>(v)), 0, 0, 0};
  if constexpr (sizeof(In) > 4) r.y = static_cast<int>(static_cast<uint32_t>(v >> 32));
  if constexpr (sizeof(In) > 8) {
    r.z = static_cast<int>(static_cast<uint32_t>(v >> 64));
    r.w = static_cast<int>(static_cast<uint32_t>(v >> 96));
  }
  return r;
}
}  // namespace util

// ---------------------------------------------------------------- groups
namespace group {
// include/fss/group/bytes.cuh:19-43: 16 bytes under XOR
struct Bytes {
  int4 val;
  __host__ __device__ Bytes() : val{0, 0, 0, 0} {}
  __host__ __device__ Bytes operator+(Bytes o) const { return Bytes(util::Xor(val, o.val)); }
  __host__ __device__ Bytes operator-() const { return *this; }
  __host__ __device__ static Bytes From(int4 buf) { return Bytes(buf); }
  __host__ __device__ int4 Into() const { return val; }
  static constexpr int kId = FSS_AMD_GROUP_BYTES;
  static constexpr uint64_t kMod = 0;

 private:
  __host__ __device__ explicit Bytes(int4 b) : val(b) {}
};

// include/fss/group/uint.cuh:27-88: the low sizeof(T) bytes of the buffer as an integer, wrapping or modulo `mod`;
// T = __uint128_t keeps 127 bits (bit 0 of w is the clamped bit, so the top word is stored shifted left by one).
// The engine's name for a Uint group -- enum fss_amd_group + the context's modulus (fss_amd_ctx_set_uint_modulus) -- for the
// groups its prebuilt kernels compute in: every T up to 64 bits with any modulus, and __uint128_t modulo 2^127.  A 128-bit
// group with another modulus (the reference allows 0 < mod <= 2^127, group/uint.cuh:28-30; src/group_test.cu:18 uses 2^126)
// is a complete value type without that name: schemes over it run the per-thread code in the generic kernels, like a
// caller-defined Groupable.
template <typename T, T mod, bool kKnown = (sizeof(T) < 16 || mod == (static_cast<T>(1) << 127))>
struct UintEngineName {};
template <typename T, T mod>
struct UintEngineName<T, mod, true> {
  static constexpr int kId = sizeof(T) == 1   ? FSS_AMD_GROUP_U8
                             : sizeof(T) == 2 ? FSS_AMD_GROUP_U16
                             : sizeof(T) == 4 ? FSS_AMD_GROUP_U32
                             : sizeof(T) == 8 ? FSS_AMD_GROUP_U64
                                              : FSS_AMD_GROUP_U128_M127;
  static constexpr uint64_t kMod = sizeof(T) == 16 ? 0 : static_cast<uint64_t>(mod);
};
template <typename T, T mod = 0>
  requires((std::is_unsigned_v<T> || std::is_same_v<T, unsigned __int128>) && sizeof(T) <= 16 &&
           (sizeof(T) < 16 || (mod > 0 && mod <= static_cast<T>(1) << 127)))
struct Uint : UintEngineName<T, mod> {
  T val;
  __host__ __device__ Uint() : val(0) {}
  __host__ __device__ Uint operator+(Uint o) const {
    const T wrapped = static_cast<T>(val + o.val);
    if constexpr (mod == 0) return Uint(wrapped);
    else return Uint(val >= static_cast<T>(mod - o.val) ? static_cast<T>(wrapped - mod) : wrapped);
  }
  __host__ __device__ Uint operator-() const {
    if constexpr (mod == 0) return Uint(static_cast<T>(0 - val));
    else return Uint(val == 0 ? T(0) : static_cast<T>(mod - val));
  }
  __host__ __device__ static Uint From(int4 buf) {
    const uint32_t w[4] = {static_cast<uint32_t>(buf.x), static_cast<uint32_t>(buf.y), static_cast<uint32_t>(buf.z),
                           static_cast<uint32_t>(buf.w)};
    T v;
    if constexpr (sizeof(T) <= 4) v = static_cast<T>(w[0]);
    else if constexpr (sizeof(T) == 8) v = static_cast<T>(w[0]) | static_cast<T>(w[1]) << 32;
    else v = static_cast<T>(w[0]) | static_cast<T>(w[1]) << 32 | static_cast<T>(w[2]) << 64 | static_cast<T>(w[3] >> 1) << 96;
    if constexpr (mod != 0) v = static_cast<T>(v % mod);
    return Uint(v);
  }
  __host__ __device__ int4 Into() const {
    int4 r{0, 0, 0, 0};
    if constexpr (sizeof(T) <= 4) {
      r.x = static_cast<int>(val);
    } else {
      r.x = static_cast<int>(static_cast<uint32_t>(val));
      r.y = static_cast<int>(static_cast<uint32_t>(val >> 32));
      if constexpr (sizeof(T) == 16) {
        r.z = static_cast<int>(static_cast<uint32_t>(val >> 64));
        r.w = static_cast<int>(static_cast<uint32_t>(val >> 96) << 1);
      }
    }
    return r;
  }

 private:
  __host__ __device__ explicit Uint(T v) : val(v) {}
};
static_assert(Groupable<Bytes> && Groupable<Uint<uint8_t>> && Groupable<Uint<uint32_t>> && Groupable<Uint<uint64_t>> &&
              Groupable<Uint<unsigned __int128, static_cast<unsigned __int128>(1) << 127>>);
}  // namespace group

enum class DcfPred { kLt = FSS_AMD_PRED_LT, kGt = FSS_AMD_PRED_GT };  // include/fss/dcf.cuh:58-61

// ---------------------------------------------------------------- engine contexts behind the value types
namespace detail {
// Everything a fss_amd_ctx is made from.  The reference's PRG / scheme objects are small values that are copied
// around freely (point_eval_gpu.cuh:455 passes them to kernels by value); here they stay values too, and the engine
// context they stand for is looked up -- created on first use -- by this description.
struct Spec {
  int device = 0;
  int prg = FSS_AMD_PRG_AES128_MMO;
  uint8_t aes[64] = {};
  int32_t hash_key[4] = {};
  int32_t nonce[2] = {};
  int rounds = 20;
  uint64_t mod = 0;
  int has_iv = 0;   // 1: keyed BLAKE3 with iv[0..7] (fss::hash::Blake3); 2: SHA-256 keyed with iv[0..3] (fss::hash::Sha256)
  int32_t iv[8] = {};
};
inline std::string KeyOf(const Spec& s) {
  std::string k;
  auto put = [&k](const void* p, size_t n) { k.append(static_cast<const char*>(p), n); };
  put(&s.device, sizeof s.device); put(&s.prg, sizeof s.prg); put(s.aes, sizeof s.aes);
  put(s.hash_key, sizeof s.hash_key); put(s.nonce, sizeof s.nonce); put(&s.rounds, sizeof s.rounds);
  put(&s.mod, sizeof s.mod); put(&s.has_iv, sizeof s.has_iv); put(s.iv, sizeof s.iv);
  return k;
}
struct Registry {
  std::mutex mu;
  std::map<std::string, fss_amd_ctx*> ctxs;
  std::mutex stage_mu;  // the host-buffer methods share one staging area per context: one at a time
  // The asynchronous launchers enqueue "small arguments, then the kernel" as one unit PER (context, stream): launches on
  // different streams (or contexts) do not take each other's launch lock.  (Inside ONE context a launcher whose
  // workspace has to grow still holds the context's scratch lock across hipFree / hipMalloc -- a device-wide wait --
  // and launchers of that context on other streams queue behind it for that long; steady-state launches never grow.)
  // The per-thread streams behind hipStreamPerThread are different streams (keyed by thread).
  // Lock order: `mu` and a launch lock are never held together (Engine() takes `mu` under a launch lock, so nothing may
  // take a launch lock under `mu`); launch_map_mu is a leaf.
  using LaunchKey = std::tuple<fss_amd_ctx*, hipStream_t, std::thread::id>;
  std::mutex launch_map_mu;
  std::map<LaunchKey, std::shared_ptr<std::mutex>> launch_mus;
  static LaunchKey KeyFor(fss_amd_ctx* ctx, hipStream_t stream) {
    return LaunchKey(ctx, stream, stream == hipStreamPerThread ? std::this_thread::get_id() : std::thread::id());
  }
  // shared ownership: an entry may be dropped (ForgetLaunchMutex) while a launcher still holds or waits for the mutex
  std::shared_ptr<std::mutex> LaunchMutex(fss_amd_ctx* ctx, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(launch_map_mu);
    auto& m = launch_mus[KeyFor(ctx, stream)];
    if (!m) m = std::make_shared<std::mutex>();
    return m;
  }
  void ForgetLaunchMutex(fss_amd_ctx* ctx, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(launch_map_mu);
    launch_mus.erase(KeyFor(ctx, stream));
  }
};
inline Registry& Reg() {
  static Registry* r = new Registry;  // never destroyed: contexts must not outlive the HIP runtime's own teardown
  return *r;
}
inline fss_amd_ctx* Engine(Spec s) {
  CheckHip(hipGetDevice(&s.device), "hipGetDevice");
  const std::string key = KeyOf(s);
  Registry& r = Reg();
  std::lock_guard<std::mutex> lk(r.mu);
  auto it = r.ctxs.find(key);
  if (it != r.ctxs.end()) return it->second;
  fss_amd_ctx* c = nullptr;
  Check(fss_amd_ctx_create(&c, s.device, s.aes, s.hash_key), "fss_amd_ctx_create");
  int rc = s.prg == FSS_AMD_PRG_CHACHA ? fss_amd_ctx_set_prg(c, s.prg, s.nonce, s.rounds) : FSS_AMD_OK;
  if (rc == FSS_AMD_OK && s.mod) rc = fss_amd_ctx_set_uint_modulus(c, s.mod);
  if (rc == FSS_AMD_OK && s.has_iv == 1) rc = fss_amd_ctx_set_hash_iv(c, s.iv);
  if (rc == FSS_AMD_OK && s.has_iv == 2) rc = fss_amd_ctx_set_hash_sha256(c, s.iv);
  // the value types stage and allocate for their callers already: their contexts opt in to the shared-prefix plan for
  // large one-key batches (the C ABI's own default is FSS_AMD_POINTS_PLAIN: nothing allocated inside an eval call)
  if (rc == FSS_AMD_OK) rc = fss_amd_ctx_set_points_plan(c, FSS_AMD_POINTS_AUTO);
  if (rc != FSS_AMD_OK) {
    fss_amd_ctx_destroy(c);
    Check(rc, "fss_amd_ctx setup");
  }
  r.ctxs.emplace(key, c);
  return c;
}
// Drop every context (and its scratch) the value types created so far, e.g. before hipDeviceReset.  The registry
// holds one context (about 10 KiB of device memory plus its scratch) per distinct (device, PRG keys / nonce, hash
// key, modulus) ever used and never evicts by itself: a program that rotates key material calls this between epochs.
inline void ReleaseEngines() {
  Registry& r = Reg();
  std::lock_guard<std::mutex> lk(r.mu);
  for (auto& kv : r.ctxs) fss_amd_ctx_destroy(kv.second);
  r.ctxs.clear();
  std::lock_guard<std::mutex> lm(r.launch_map_mu);
  r.launch_mus.clear();  // no launcher may be running on a context that is being destroyed
}
// Free the scratch the asynchronous launchers keep for `stream` (every context); call before destroying the stream.
// Waits for the device (hipFree).
inline void ReleaseStreamScratch(hipStream_t stream) {
  Registry& r = Reg();
  std::vector<fss_amd_ctx*> ctxs;
  {  // the contexts under `mu`, the launch locks after it is dropped: a launcher holds its launch lock while it resolves
     // its context (Engine() takes `mu`), so taking them the other way round here could deadlock against it
    std::lock_guard<std::mutex> lk(r.mu);
    for (auto& kv : r.ctxs) ctxs.push_back(kv.second);
  }
  for (fss_amd_ctx* c : ctxs) {
    {
      // the (context, stream) launch lock: no launcher of another thread sits between "small arguments" and "kernel" on
      // this stream while its buffer goes away
      const std::shared_ptr<std::mutex> m = r.LaunchMutex(c, stream);
      std::lock_guard<std::mutex> launch(*m);
      Check(fss_amd_ctx_scratch_release(c, stream), "fss_amd_ctx_scratch_release");
    }
    r.ForgetLaunchMutex(c, stream);  // the stream is going away: its entry with it (a later launch starts a new one)
  }
}

template <typename In>
constexpr int XBytes() {
  static_assert(sizeof(In) == 1 || sizeof(In) == 2 || sizeof(In) == 4 || sizeof(In) == 8 || sizeof(In) == 16,
                "In must be an unsigned integer of 1, 2, 4 or 8 bytes or __uint128_t");
  return static_cast<int>(sizeof(In));
}
template <typename In>
concept InType = std::is_unsigned_v<In> || std::is_same_v<In, unsigned __int128>;

// Is p device memory?  Inputs that are not (keys, seeds: a few KiB) are staged.
inline bool OnDevice(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an unregistered host pointer is reported as an error: not one
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeUnified;
}
// The address a kernel can use for p: p itself for device memory, the device mapping of pinned host memory
// (hipHostMalloc / hipHostRegister), nullptr for pageable host memory.  Large OUTPUTS in pinned memory are written by
// the kernels themselves across PCIe (no staging buffer, no copy afterwards; fss_amd_eval_points_host is the same idea
// for a batch of points).
template <typename T>
inline T* DeviceView(T* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (a.type == hipMemoryTypeHost) return static_cast<T*>(a.devicePointer);
  if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeUnified) return p;
  return nullptr;
}

// Staging area for the host-buffer (reference-signature) methods: slices of a scratch buffer of the context that is
// the stage's alone (its own key: the asynchronous launchers' scratch of the null stream is a different buffer, so a
// launcher running on stream 0 in another thread never sees its slots overwritten by a staged call).  One staged
// call at a time (stage_mu); blocking copies on the null stream keep the reference's synchronous semantics.
inline char kStageKey;  // its address is the scratch key
class Stage {
 public:
  Stage(fss_amd_ctx* ctx, size_t bytes) : lock_(Reg().stage_mu), ctx_(ctx) {
    Check(fss_amd_ctx_scratch(ctx, &kStageKey, bytes + 256, &base_), "fss_amd_ctx_scratch");  // + alignment slack
  }
  // the next `bytes` of the area (32-byte aligned)
  void* Take(size_t bytes) {
    void* p = static_cast<char*>(base_) + off_;
    off_ += (bytes + 31) & ~size_t(31);
    return p;
  }
  // device copy of host data, or the pointer itself when it already is device memory
  template <typename T>
  const T* Load(const T* src, size_t count) {
    if (OnDevice(src)) return src;
    void* d = Take(count * sizeof(T));
    Put(d, src, count * sizeof(T));
    return static_cast<const T*>(d);
  }
  template <typename T>
  const T* Value(const T& v) {
    void* d = Take(sizeof(T));
    Put(d, &v, sizeof(T));
    return static_cast<const T*>(d);
  }
  // device buffer for an output: the destination itself when a kernel can write it (device memory, pinned host memory)
  template <typename T>
  T* Out(T* dst, size_t count) {
    if (T* v = DeviceView(dst)) return v;
    return static_cast<T*>(Take(count * sizeof(T)));
  }
  // ... for an output that later kernels of the same call read back or update in place (Grotto's running XOR, its
  // parity tree): only device memory is used in place; a pinned host buffer is staged like a pageable one (reads and
  // read-modify-writes across PCIe would cost more than the one copy they save)
  template <typename T>
  T* OutReadBack(T* dst, size_t count) {
    if (OnDevice(dst)) return dst;
    return static_cast<T*>(Take(count * sizeof(T)));
  }
  template <typename T>
  void Back(T* dst, const T* dev, size_t count) {
    if (static_cast<const void*>(DeviceView(dst)) == static_cast<const void*>(dev)) {
      CheckHip(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
      return;
    }
    CheckHip(hipMemcpy(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy (stage out)");
  }

 private:
  // seeds, points and single keys (at most a few KiB) go up as kernel arguments: stream-ordered, no blocking copy
  void Put(void* d, const void* src, size_t bytes) {
    if (bytes <= (size_t(1) << 16)) Check(fss_amd_upload(ctx_, d, src, bytes, nullptr), "fss_amd_upload");
    else CheckHip(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice), "hipMemcpy (stage in)");
  }
  std::lock_guard<std::mutex> lock_;
  fss_amd_ctx* ctx_;
  void* base_ = nullptr;
  size_t off_ = 0;
};

// Scratch of (ctx, stream) for the asynchronous launchers: [0, 256) small by-value arguments, then the workspace.
// One buffer per stream (per thread for hipStreamPerThread).  A launcher holds the (context, stream) launch lock from
// taking the buffer until its kernel is enqueued, so two host threads launching on the same stream (the default stream = 0 in particular)
// enqueue "upload, kernel" pairs whole: stream order then keeps one call's small arguments and workspace intact until
// its kernel has read them.
struct Scratch {
  std::shared_ptr<std::mutex> mu_;
  std::lock_guard<std::mutex> lock_;
  char* base;
  Scratch(fss_amd_ctx* ctx, hipStream_t stream, size_t ws_bytes) : mu_(Reg().LaunchMutex(ctx, stream)), lock_(*mu_) {
    void* p = nullptr;
    Check(fss_amd_ctx_scratch(ctx, stream, ws_bytes + 256, &p), "fss_amd_ctx_scratch");
    base = static_cast<char*>(p);
  }
  void* Small(int slot) const { return base + 32 * slot; }
  void* Workspace() const { return base + 256; }
};
inline const int4* PutInt4(fss_amd_ctx* ctx, const Scratch& s, int slot, int4 v, hipStream_t stream) {
  Check(fss_amd_upload(ctx, s.Small(slot), &v, sizeof v, stream), "fss_amd_upload");
  return static_cast<const int4*>(s.Small(slot));
}
}  // namespace detail

using detail::ReleaseEngines;
using detail::ReleaseStreamScratch;

namespace detail {
template <typename P>
auto GenOne(const P& prg, int4 seed) -> std::array<int4, P::kMul>;  // one PRG call on the engine (below)
}

// ---------------------------------------------------------------- PRG hook
namespace prg {
// What stands behind the EVP_CIPHER_CTX* that CreateCtxs hands out (the reference's are OpenSSL contexts holding a key
// schedule, aes128_mmo.cuh:49-64): the key itself, and a tag by which a pointer that did not come from CreateCtxs -- a real
// OpenSSL context, whose key cannot be read back -- is told apart.
struct AesKeyCtx {
  static constexpr uint64_t kTag = 0x7874634b73656141ull;  // "AesKctx"
  uint64_t tag;
  unsigned char key[16];
  static EVP_CIPHER_CTX* Handle(AesKeyCtx* c) { return reinterpret_cast<EVP_CIPHER_CTX*>(c); }
  static AesKeyCtx* Of(EVP_CIPHER_CTX* h) {
    AesKeyCtx* c = reinterpret_cast<AesKeyCtx*>(h);
    if (h == nullptr || c->tag != kTag)
      throw Error("Aes128Mmo: not a context from CreateCtxs (the key of an OpenSSL context cannot be read back)", FSS_AMD_EINVAL);
    return c;
  }
};

// AES-128 in Matyas-Meyer-Oseas mode as the PRG: out[i] = AES_{key i}(seed) ^ seed (aes128_mmo.cuh:72-93).
// mul outputs per call (Half-Tree 1, DPF / Grotto 2, DCF 4).  A plain value (the key bytes); copies are free.
template <int mul>
class Aes128Mmo {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  using Schedules = thread::AesSchedules<mul, false, true>;
  Schedules ks_;  // raw keys + round keys (little-endian words), for Gen called on the device
  mutable bool wave_tables_ready_ = false;  // device: this thread has seen its wave's LDS copy of the tables filled (WaveTables)

  __host__ __device__ void SetKeys(const unsigned char* const* keys) {
    ks_.Set(keys, [](uint32_t b) { return static_cast<uint32_t>(thread::kAes.sbox[b]); });
  }

 public:
  // ctxs: from CreateCtxs(); only read here, the caller frees them with FreeCtxs()  (aes128_mmo.cuh:40-42)
  Aes128Mmo(std::span<EVP_CIPHER_CTX*, mul> ctxs) {
    const unsigned char* keys[mul];
    for (int i = 0; i < mul; ++i) keys[i] = AesKeyCtx::Of(ctxs[i])->key;
    SetKeys(keys);
  }
  // the key bytes directly (no context objects to manage); also constructible inside a kernel
  __host__ __device__ explicit Aes128Mmo(const unsigned char* const keys[mul]) { SetKeys(keys); }
  static std::array<EVP_CIPHER_CTX*, mul> CreateCtxs(const unsigned char* keys[mul]) {  // aes128_mmo.cuh:49
    std::array<EVP_CIPHER_CTX*, mul> ctxs;
    for (int i = 0; i < mul; ++i) {
      AesKeyCtx* c = new AesKeyCtx;
      c->tag = AesKeyCtx::kTag;
      std::memcpy(c->key, keys[i], 16);
      ctxs[i] = AesKeyCtx::Handle(c);
    }
    return ctxs;
  }
  static void FreeCtxs(std::span<EVP_CIPHER_CTX*, mul> ctxs) {  // aes128_mmo.cuh:66
    for (EVP_CIPHER_CTX* h : ctxs) {
      AesKeyCtx* c = AesKeyCtx::Of(h);
      c->tag = 0;
      delete c;
    }
  }
  // Host: one seed through the engine (a batch of one on the GPU; fss_amd_prg_gen is the batched form).
  // Device: computed by the calling thread (the reference's OpenSSL-backed Gen traps there, aes128_mmo.cuh:75-77)
  // on its workgroup's LDS copy of the round table, round keys in scalar registers (one key) or in the wave's LDS slot
  // (two, four) (fss_amd_thread.hpp).
  // (always inlined: out of line the call takes `this`, and the whole scheme object moves from the kernel's arguments to
  // the stack -- 784 bytes of scratch per lane in a DCF Gen kernel)
  __device__ __attribute__((always_inline)) std::array<int4, mul> Gen(int4 seed) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.Gen(thread::WaveTables(wave_tables_ready_), seed);
#else
    (void)seed;
    return {};
#endif
  }
  __host__ std::array<int4, mul> Gen(int4 seed) const {
    return detail::GenOne(*this, seed);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // Device only: this PRG with its per-call checks made once, for a walk that calls it level after level
  // (thread::PreparedAes; the per-thread Eval takes it by itself).
  __device__ thread::PreparedAes<mul, Schedules, thread::BankedLdsRoundTable> Prepared() const {
    const thread::BankedLdsRoundTable tab = thread::WaveTables(wave_tables_ready_);
    return thread::PreparedAes<mul, Schedules, thread::BankedLdsRoundTable>{ks_, tab, ks_.ReadyKeys(tab)};
  }
#endif
  // Device only: the outputs a tree level takes for one side -- {Gen(seed)[right]} for mul = 2, {Gen(seed)[2 right],
  // Gen(seed)[2 right + 1]} for mul = 4 -- at one block per output instead of two (the per-thread Eval uses it).
  __device__ __attribute__((always_inline)) std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.GenSide(thread::WaveTables(wave_tables_ready_), seed, right);
#else
    (void)seed;
    (void)right;
    return {};
#endif
  }

  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_AES128_MMO;
    for (int i = 0; i < mul; ++i) ks_.KeyBytes(i, s.aes + 16 * i);
  }
};

// fss::prg::Aes128MmoWide<mul>: the opt-in wide form of Aes128Mmo for callers who write their own kernels -- the engine's
// own four-table, rotate-free LDS layout (128 KiB, fss_amd_thread.hpp: WideAesTables) behind the same Gen(int4) /
// GenSide, the same outputs bit for bit, 1.3 - 1.45 x the rate of the default 32 KiB one-table form on key-major keys
// (DPF n = 32, 2^21 keys: 2.5 G evals/s one key per thread, 2.6 G in a grid-stride loop, against 1.9 G; DCF 1.45 - 1.5 G
// against 1.0 G; profiles/r6/dropin_device_test.txt).  The default types stay what they are because this one dictates the
// caller's launch:
//   * workgroups of exactly 1024 threads (128 KiB of LDS = one workgroup per CU; anything else is refused at run time:
//     the kernel traps);
//   * the kernel calls  fss::prg::Aes128MmoWide<mul>::InitWorkgroup();  once, from EVERY thread, in uniform control flow,
//     before the first Gen / Eval: the tables are filled cooperatively behind one workgroup barrier (a Gen that does
//     not find them traps);
//   * the object is built on the HOST and reaches the kernel by value, as a kernel argument (alone or inside its scheme:
//     `kernel<<<n / 1024, 1024>>>(dpf, ...)` with `dpf.Eval(...)` called on the parameter itself): its round keys are
//     read with scalar loads from where the object lives, a round at a time.  A copy in a local variable still works but
//     reads them from the copy's memory with vector loads.
// Its walks also read a key-major key a 128-byte line at a time (thread::KeyMajorLines: every line fetched once instead of
// 1.8 times), which takes 20 - 32 more registers.  Best shape: as many workgroups as the chip has CUs and a grid-stride
// loop over the keys (the tables are filled once per workgroup and no CU waits for a workgroup's last wave before the next
// workgroup may start).
// Pick it when the kernel is yours to shape (block size 1024, 110 registers per thread to spare); keep Aes128Mmo when the
// PRG has to work inside kernels with other block sizes, their own LDS, or divergent first calls.
// Call-site shape: src/bench_gpu.cu:100-138 (tables prepared by the block, then one Eval per thread).
template <int mul>
class Aes128MmoWide {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  static constexpr int kPairs = mul == 4 ? 2 : 1;
  static constexpr int kPerRound = mul == 1 ? 4 : 8;
  uint32_t keyw_[mul][4];                 // the raw keys (little-endian words), for Describe
  uint32_t sel_[kPairs][11 * kPerRound];  // per round A[4] (, D[4] = (A ^ B)[4]): mul 2: {k0, k0^k1}; mul 4: {k0, k0^k2}, {k1, k1^k3}
  // (no mutable members: the object is read in place from the kernel-argument segment, which a written-to copy would end)

  void SetKeys(const unsigned char* const* keys) {
    uint32_t rk[mul][44];
    for (int i = 0; i < mul; ++i) {
      for (int j = 0; j < 4; ++j)
        keyw_[i][j] = static_cast<uint32_t>(keys[i][4 * j]) | static_cast<uint32_t>(keys[i][4 * j + 1]) << 8 |
                      static_cast<uint32_t>(keys[i][4 * j + 2]) << 16 | static_cast<uint32_t>(keys[i][4 * j + 3]) << 24;
      thread::AesExpandKeyWords<false>(keys[i], rk[i], [](uint32_t b) { return static_cast<uint32_t>(thread::kAes.sbox[b]); });
    }
    for (int r = 0; r < 11; ++r)
      for (int j = 0; j < 4; ++j) {
        if constexpr (mul == 1) {
          sel_[0][4 * r + j] = rk[0][4 * r + j];
        } else if constexpr (mul == 2) {
          sel_[0][8 * r + j] = rk[0][4 * r + j];
          sel_[0][8 * r + 4 + j] = rk[0][4 * r + j] ^ rk[1][4 * r + j];
        } else {
          sel_[0][8 * r + j] = rk[0][4 * r + j];
          sel_[0][8 * r + 4 + j] = rk[0][4 * r + j] ^ rk[2][4 * r + j];
          sel_[1][8 * r + j] = rk[1][4 * r + j];
          sel_[1][8 * r + 4 + j] = rk[1][4 * r + j] ^ rk[3][4 * r + j];
        }
      }
  }

 public:
  Aes128MmoWide(std::span<EVP_CIPHER_CTX*, mul> ctxs) {
    const unsigned char* keys[mul];
    for (int i = 0; i < mul; ++i) keys[i] = AesKeyCtx::Of(ctxs[i])->key;
    SetKeys(keys);
  }
  explicit Aes128MmoWide(const unsigned char* const keys[mul]) { SetKeys(keys); }
  static std::array<EVP_CIPHER_CTX*, mul> CreateCtxs(const unsigned char* keys[mul]) { return Aes128Mmo<mul>::CreateCtxs(keys); }
  static void FreeCtxs(std::span<EVP_CIPHER_CTX*, mul> ctxs) { Aes128Mmo<mul>::FreeCtxs(ctxs); }

  // Device, every thread of the 1024-thread workgroup, uniform control flow, once per kernel before the first Gen / Eval.
  __device__ static void InitWorkgroup() {
#if defined(__HIP_DEVICE_COMPILE__)
    thread::WideTablesInit();
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __device__ __attribute__((always_inline)) thread::PreparedWide<mul> Prepared() const {
    return thread::PreparedWide<mul>{thread::WideTab::Make(), sel_[0], sel_[kPairs - 1], thread::PreparedWide<mul>::PublishedLevels()};
  }
#endif
  __device__ __attribute__((always_inline)) std::array<int4, mul> Gen(int4 seed) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return Prepared().Gen(seed);
#else
    (void)seed;
    return {};
#endif
  }
  __host__ std::array<int4, mul> Gen(int4 seed) const { return detail::GenOne(*this, seed); }
  __device__ __attribute__((always_inline)) std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return Prepared().GenSide(seed, right);
#else
    (void)seed;
    (void)right;
    return {};
#endif
  }
  // the per-thread walks read key-major keys a cache line at a time with this PRG (thread::KeyMajorLines)
  static constexpr bool kKeyMajorLines = true;
  static constexpr bool kWideTables = true;  // needs the caller's own kernel shape: refused by this header's generic kernels
  // The engine runs the same function with these keys: schemes over it take the batched entry points and launchers too.
  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_AES128_MMO;
    for (int i = 0; i < mul; ++i)
      for (int j = 0; j < 16; ++j) s.aes[16 * i + j] = static_cast<uint8_t>(keyw_[i][j >> 2] >> (8 * (j & 3)));
  }
};

// fss::prg::Aes128SoftWide<mul>: the wide layout for PRG objects that a kernel constructs PER THREAD from key bytes, the
// shape of the reference's AES kernels (src/bench_gpu.cu:100-138: the block prepares the tables, every thread builds its
// `Aes128Soft<2> prg(kKeys, s_te0, s_sbox)` and evaluates):
//     fss::prg::Aes128SoftWide<2>::InitWorkgroup();          // every thread of the 1024-thread workgroup, before anything else
//     fss::prg::Aes128SoftWide<2> prg(kKeys);  fss::Dpf<...> dpf{prg};  ys[tid] = dpf.Eval(...);
// Same tables, same launch rules as Aes128MmoWide; an object built on the device has no memory home for scalar key loads,
// so its round keys live in the wave's LDS key slot like Aes128Mmo's (44 more LDS reads per block than Aes128MmoWide:
// about 15 % slower; still the rotate-free rounds).  Host-built objects work too.
template <int mul>
class Aes128SoftWide {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  using Schedules = thread::AesSchedules<mul, false, true>;
  Schedules ks_;

 public:
  __host__ __device__ explicit Aes128SoftWide(const uint8_t keys[][16]) {
    const unsigned char* kp[mul];
    for (int i = 0; i < mul; ++i) kp[i] = keys[i];
    ks_.Set(kp, [](uint32_t b) { return static_cast<uint32_t>(thread::kAes.sbox[b]); });
  }
  __device__ static void InitWorkgroup() {
#if defined(__HIP_DEVICE_COMPILE__)
    thread::WideTablesInit();
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __device__ __attribute__((always_inline)) thread::PreparedAes<mul, Schedules, thread::WideTab> Prepared() const {
    const thread::WideTab tab = thread::WideTab::Make();
    return thread::PreparedAes<mul, Schedules, thread::WideTab>{ks_, tab, ks_.ReadyKeys(tab)};
  }
#endif
  __device__ __attribute__((always_inline)) std::array<int4, mul> Gen(int4 seed) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.Gen(thread::WideTab::Make(), seed);
#else
    (void)seed;
    return {};
#endif
  }
  __host__ std::array<int4, mul> Gen(int4 seed) const { return detail::GenOne(*this, seed); }
  __device__ __attribute__((always_inline)) std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.GenSide(thread::WideTab::Make(), seed, right);
#else
    (void)seed;
    (void)right;
    return {};
#endif
  }
  // (key-major keys a line at a time, like Aes128MmoWide -- except with four keys, where the 32 registers of the DCF line
  // buffer beside two ciphers with keys from LDS spill)
  static constexpr bool kKeyMajorLines = mul != 4;
  static constexpr bool kWideTables = true;
  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_AES128_MMO;
    for (int i = 0; i < mul; ++i) ks_.KeyBytes(i, s.aes + 16 * i);
  }
};

// fss::prg::Aes128MmoRaw<mul>(const uint8_t keys[][16]) (include/fss/prg/aes128_mmo_raw.cuh:39-110): the reference's
// host-only AES-NI form of the same function, constructed from the key bytes directly.  Here it IS Aes128Mmo with that
// constructor: Gen on the host goes to the engine, schemes over it are engine schemes, and on the device it works too.
template <int mul>
class Aes128MmoRaw : public Aes128Mmo<mul> {
  static std::array<const unsigned char*, mul> Ptrs(const uint8_t keys[][16]) {
    std::array<const unsigned char*, mul> p;
    for (int i = 0; i < mul; ++i) p[i] = keys[i];
    return p;
  }

 public:
  Aes128MmoRaw(const uint8_t keys[][16]) : Aes128Mmo<mul>(Ptrs(keys).data()) {}
};

// ---------------------------------------------------------------- software AES-128-MMO with caller-held tables
// fss::prg::aes_detail and fss::prg::Aes128Soft<mul> (include/fss/prg/aes128_mmo_soft.cuh:32-218): the PRG the
// reference's GPU kernels construct per thread from tables in shared memory,
//     __shared__ uint32_t s_te0[256]; __shared__ uint8_t s_sbox[256];   // filled with ComputeTe0 / Sbox
//     fss::prg::Aes128Soft<2> prg(keys, s_te0, s_sbox);  fss::Dpf<...> dpf{prg};  ys[tid] = dpf.Eval(...);
// (src/bench_gpu.cu:100-138).  Same names, same arguments, same table contents (big-endian Te0, te0[0] = c66363a5).
// Behind them: ds_read lookups on a bank-replicated LDS copy of the round table (the host side reads the caller's tables),
// wave-uniform round keys (scalar registers or the wave's LDS slot) instead of a 176-byte-per-key array per thread, all nine table rounds unrolled, and -- through GenSide, which Dpf / Dcf::Eval use
// -- only the child a level takes.
namespace aes_detail {
constexpr int kNb = 4;
constexpr int kNk = 4;
constexpr int kNr = 10;
constexpr int kRoundKeySize = kNb * (kNr + 1) * 4;  // 176 bytes

__host__ __device__ inline uint8_t Sbox(uint8_t idx) { return thread::kAes.sbox[idx]; }
__host__ __device__ inline uint8_t Rcon(int idx) {  // 0x8d, then the powers of x in GF(2^8)
  uint8_t r = 0x8d;
  for (int i = 0; i < idx; ++i) r = thread::Times2(r);
  return r;
}
// Te0[idx] as a big-endian column (2 S, S, S, 3 S)
__host__ __device__ inline uint32_t ComputeTe0(uint8_t idx) { return thread::ByteSwap(thread::kAes.te0[idx]); }
__host__ __device__ inline void InitTe0(uint32_t* dst) {
  for (int i = 0; i < 256; ++i) dst[i] = ComputeTe0(static_cast<uint8_t>(i));
}
__host__ __device__ inline void InitSbox(uint8_t* dst) {
  for (int i = 0; i < 256; ++i) dst[i] = Sbox(static_cast<uint8_t>(i));
}
__host__ __device__ inline uint32_t RotWord8(uint32_t x) { return thread::AesRot<true>(x, 1); }
__host__ __device__ inline uint32_t RotWord16(uint32_t x) { return thread::AesRot<true>(x, 2); }
__host__ __device__ inline uint32_t RotWord24(uint32_t x) { return thread::AesRot<true>(x, 3); }
__host__ __device__ inline uint32_t LoadBE32(const uint8_t* p) {
  return static_cast<uint32_t>(p[0]) << 24 | static_cast<uint32_t>(p[1]) << 16 | static_cast<uint32_t>(p[2]) << 8 | p[3];
}
__host__ __device__ inline void StoreBE32(uint8_t* p, uint32_t v) {
  for (int i = 0; i < 4; ++i) p[i] = static_cast<uint8_t>(v >> (24 - 8 * i));
}
// 176 bytes of round keys from a 16-byte key, S-box lookups through `sbox`
__host__ __device__ inline void KeyExpansion(uint8_t* round_key, const uint8_t* key, const uint8_t* sbox) {
  uint32_t w[44];
  thread::AesExpandKeyWords<true>(key, w, [sbox](uint32_t b) { return static_cast<uint32_t>(sbox[b]); });
  for (int i = 0; i < 44; ++i) StoreBE32(round_key + 4 * i, w[i]);
}
// buf = AES-128(buf) in place, round keys as KeyExpansion wrote them
__host__ __device__ inline void Encrypt(uint8_t* buf, const uint8_t* round_key, const uint32_t* te0, const uint8_t* sbox) {
  uint32_t rk[44], in[4], out[4];
  for (int i = 0; i < 44; ++i) rk[i] = LoadBE32(round_key + 4 * i);
  for (int j = 0; j < 4; ++j) in[j] = LoadBE32(buf + 4 * j);
  thread::AesEncrypt<true>(thread::PlainTables{te0, sbox}, thread::AesKeyWords{rk}, in, out);
  for (int j = 0; j < 4; ++j) StoreBE32(buf + 4 * j, out[j]);
}
// The bank-replicated form of the round table (Aes128SoftBanked below): kBankedTe0Words words, entry i repeated in
// words 32 i .. 32 i + 31.  InitTe0Banked is called by every thread of the workgroup (its linear index and the
// workgroup's size), followed by a barrier; on the host, InitTe0Banked(dst, 0, 1).
constexpr int kBankedTe0Words = 256 * 32;
__host__ __device__ inline void InitTe0Banked(uint32_t* dst, int tid, int nthreads) {
  for (int w = tid; w < kBankedTe0Words; w += nthreads) dst[w] = ComputeTe0(static_cast<uint8_t>(w >> 5));
}
}  // namespace aes_detail

template <int mul>
class Aes128Soft {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  using Schedules = thread::AesSchedules<mul, true, true>;
  Schedules ks_;
  const uint32_t* te0_;
  const uint8_t* sbox_;
  mutable bool wave_tables_ready_ = false;  // device: this thread has seen its wave's LDS copy of the round table filled

 public:
  // keys: mul 16-byte AES-128 keys; te0: the 256-entry round table (InitTe0 / ComputeTe0); sbox: the S-box (InitSbox /
  // Sbox).  On the device both tables belong in __shared__ memory, filled by the workgroup before any thread
  // constructs its PRG (aes128_mmo_soft.cuh:191-207); on the host, arrays of the caller.
  // Device: the constructor expands the keys through the caller's S-box; the cipher itself does NOT read the caller's 1 KiB
  // table -- 64 random lookups of a wave instruction into 256 words share banks four or five deep, which is what bounded
  // the reference-shaped kernel (src/bench_gpu.cu:100-138) at 1.96 G evals/s -- but the workgroup's bank-replicated copy of
  // the same table (thread::WaveTablesT<true>: 32 KiB of static LDS, every wave fills it itself at its first call, no
  // barrier; the S-box is byte 1 of its entries), as Aes128Mmo's device side does.  Same function, same bytes; the
  // caller's kernel is unchanged.  (A caller who wants to own the banked table: Aes128SoftBanked below.)
  __host__ __device__ Aes128Soft(const uint8_t keys[][16], const uint32_t* te0, const uint8_t* sbox)
      : te0_(te0), sbox_(sbox) {
    const unsigned char* kp[mul];
    for (int i = 0; i < mul; ++i) kp[i] = keys[i];
    ks_.Set(kp, [sbox](uint32_t b) { return static_cast<uint32_t>(sbox[b]); });
  }

  __host__ __device__ std::array<int4, mul> Gen(int4 seed) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.Gen(thread::WaveTablesT<true>(wave_tables_ready_), seed);
#else
    return ks_.Gen(thread::PlainTables{te0_, sbox_}, seed);
#endif
  }
  // the outputs of one side of a tree level (see Aes128Mmo::GenSide)
  __host__ __device__ std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return ks_.GenSide(thread::WaveTablesT<true>(wave_tables_ready_), seed, right);
#else
    return ks_.GenSide(thread::PlainTables{te0_, sbox_}, seed, right);
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // Device only: this PRG with its per-call checks made once, for a walk that calls it level after level (see Aes128Mmo)
  __device__ thread::PreparedAes<mul, Schedules, thread::BankedLdsRoundTable> Prepared() const {
    const thread::BankedLdsRoundTable tab = thread::WaveTablesT<true>(wave_tables_ready_);
    return thread::PreparedAes<mul, Schedules, thread::BankedLdsRoundTable>{ks_, tab, ks_.ReadyKeys(tab)};
  }
#endif

  // The engine runs the same function (AES-128-MMO with these keys): schemes over Aes128Soft take the batched entry
  // points and the fss::gpu launchers like schemes over Aes128Mmo.
  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_AES128_MMO;
    for (int i = 0; i < mul; ++i) ks_.KeyBytes(i, s.aes + 16 * i);
  }
};

// Aes128Soft on the bank-replicated table: the opt-in for callers who write their own kernels and want the per-thread
// form without LDS bank conflicts -- 32 KiB of LDS per workgroup instead of 1 KiB,
//     __shared__ uint32_t s_te0[fss::prg::aes_detail::kBankedTe0Words]; __shared__ uint8_t s_sbox[256];
//     fss::prg::aes_detail::InitTe0Banked(s_te0, threadIdx.x, blockDim.x);  // + the S-box as before, then __syncthreads()
//     fss::prg::Aes128SoftBanked<2> prg(keys, s_te0, s_sbox);
// and everything else as with Aes128Soft (same function, same schemes, same engine routing).
template <int mul>
class Aes128SoftBanked {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  thread::AesSchedules<mul, true> ks_;
  const uint32_t* te0_;
  const uint8_t* sbox_;

 public:
  __host__ __device__ Aes128SoftBanked(const uint8_t keys[][16], const uint32_t* te0_banked, const uint8_t* sbox)
      : te0_(te0_banked), sbox_(sbox) {
    const unsigned char* kp[mul];
    for (int i = 0; i < mul; ++i) kp[i] = keys[i];
    ks_.Set(kp, [sbox](uint32_t b) { return static_cast<uint32_t>(sbox[b]); });
  }
  __host__ __device__ std::array<int4, mul> Gen(int4 seed) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_is_shared(te0_) && __builtin_amdgcn_is_shared(sbox_)) return ks_.Gen(Lds(), seed);
#endif
    return ks_.Gen(thread::BankedPlainTables{te0_, sbox_}, seed);
  }
  __host__ __device__ std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_is_shared(te0_) && __builtin_amdgcn_is_shared(sbox_)) return ks_.GenSide(Lds(), seed, right);
#endif
    return ks_.GenSide(thread::BankedPlainTables{te0_, sbox_}, seed, right);
  }
  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_AES128_MMO;
    for (int i = 0; i < mul; ++i) ks_.KeyBytes(i, s.aes + 16 * i);
  }

 private:
#if defined(__HIP_DEVICE_COMPILE__)
  __device__ thread::BankedLdsTables Lds() const {
    return thread::BankedLdsTables{(const __attribute__((address_space(3))) uint32_t*)te0_ + (__lane_id() & 31),
                                   (const __attribute__((address_space(3))) uint8_t*)sbox_};
  }
#endif
};

// ChaCha as the PRG (chacha.cuh:24-128): one permutation of [constant | seed | seed | 0, 0, nonce] per Gen.
template <int mul, int rounds = 20>
  requires(rounds % 2 == 0 && rounds > 0 && (mul == 1 || mul == 2 || mul == 4))
class ChaCha {
  int nonce_[2];

 public:
  // nonce: two ints, read here (the reference keeps the pointer, chacha.cuh:93): a pointer readable on the side
  // that constructs the object -- __constant__ / device memory inside a kernel, host memory on the host
  __host__ __device__ ChaCha(const int* nonce) : nonce_{nonce[0], nonce[1]} {}
  __device__ std::array<int4, mul> Gen(int4 seed) const {
    return thread::ChaChaGen<mul, rounds>(nonce_[0], nonce_[1], seed);
  }
  __host__ std::array<int4, mul> Gen(int4 seed) const {
    return detail::GenOne(*this, seed);
  }

  static constexpr int kMul = mul;
  void Describe(detail::Spec& s) const {
    s.prg = FSS_AMD_PRG_CHACHA;
    s.nonce[0] = nonce_[0];
    s.nonce[1] = nonce_[1];
    s.rounds = rounds;
  }
};
}  // namespace prg

namespace detail {
template <typename P>
concept EnginePrg = requires(const P& p, Spec& s) {
  { p.Describe(s) };
  { P::kMul } -> std::convertible_to<int>;
};
template <typename G>
concept EngineGroup = Groupable<G> && requires {
  { G::kId } -> std::convertible_to<int>;
  { G::kMod } -> std::convertible_to<uint64_t>;
};

template <typename Prg>
fss_amd_ctx* CtxOf(const Prg& prg, uint64_t mod = 0, const int4* hash_key = nullptr, const int32_t* iv = nullptr, int hash_kind = 0) {
  static_assert(EnginePrg<Prg>, "the engine runs fss::prg::Aes128Mmo and fss::prg::ChaCha; other Prgable types cannot "
                                "be executed by its prebuilt kernels");
  Spec s;
  prg.Describe(s);
  s.mod = mod;
  if (hash_key) std::memcpy(s.hash_key, hash_key, sizeof s.hash_key);
  if (iv) {
    s.has_iv = hash_kind == 1 ? 2 : 1;
    std::memcpy(s.iv, iv, hash_kind == 1 ? 16 : 32);
  }
  return Engine(s);
}

template <typename Prg>
auto GenOne(const Prg& prg, int4 seed) -> std::array<int4, Prg::kMul> {
  fss_amd_ctx* ctx = CtxOf(prg);
  Stage st(ctx, 16 + 16 * Prg::kMul);
  const int4* d_seed = st.Value(seed);
  int4* d_out = static_cast<int4*>(st.Take(16 * Prg::kMul));
  Check(fss_amd_prg_gen(ctx, Prg::kMul, d_seed, 1, d_out, nullptr), "fss_amd_prg_gen");
  std::array<int4, Prg::kMul> out;
  st.Back(out.data(), static_cast<const int4*>(d_out), Prg::kMul);
  return out;
}
}  // namespace detail

namespace prg {
static_assert(Prgable<Aes128Mmo<1>, 1> && Prgable<Aes128Mmo<2>, 2> && Prgable<Aes128Mmo<4>, 4>);
static_assert(Prgable<Aes128Soft<1>, 1> && Prgable<Aes128Soft<2>, 2> && Prgable<Aes128Soft<4>, 4>);
static_assert(Prgable<Aes128MmoRaw<1>, 1> && Prgable<Aes128MmoRaw<2>, 2> && Prgable<Aes128MmoRaw<4>, 4>);
static_assert(Prgable<Aes128MmoWide<1>, 1> && Prgable<Aes128MmoWide<2>, 2> && Prgable<Aes128MmoWide<4>, 4>);
static_assert(Prgable<Aes128SoftWide<1>, 1> && Prgable<Aes128SoftWide<2>, 2> && Prgable<Aes128SoftWide<4>, 4>);
static_assert(Prgable<Aes128SoftBanked<1>, 1> && Prgable<Aes128SoftBanked<2>, 2> && Prgable<Aes128SoftBanked<4>, 4>);
static_assert(Prgable<ChaCha<1>, 1> && Prgable<ChaCha<2>, 2> && Prgable<ChaCha<4>, 4>);
}  // namespace prg

// ---------------------------------------------------------------- generic kernels for caller-defined PRG / group types
// One thread per key (Gen) or per point (Eval) running the scheme's own device-side member; instantiated in the
// caller's translation unit with the caller's types.  The built-in types never come here: they go to the engine.
namespace detail {
// scratch and uploads need no key material: any context of the current device will do
inline fss_amd_ctx* AnyCtx() { return Engine(Spec{}); }

template <typename S, typename In>
__global__ void ThreadGenKernel(S sch, typename S::Cw* cws, const int4* s0s, const In* alphas, const int4* betas,
                                int64_t nkeys) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k < nkeys) sch.Gen(cws + k * S::kSlots, s0s + 2 * k, alphas[k], betas ? betas[k] : int4{0, 0, 0, 0});
}
template <typename S, typename In>
__global__ void ThreadGenOcwKernel(S sch, typename S::Cw* cws, int4* ocws, const int4* s0s, const In* alphas,
                                   const int4* betas, int64_t nkeys) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k < nkeys) sch.Gen(cws + k * S::kSlots, ocws[k], s0s + 2 * k, alphas[k], betas[k]);
}
// xs == nullptr: the whole domain, point i = i
template <typename S, typename In>
__global__ void ThreadEvalKernel(S sch, bool b, const int4* s0, const typename S::Cw* cws, const In* xs, int4* ys,
                                 int64_t n) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < n) ys[i] = sch.Eval(b, *s0, cws, xs ? xs[i] : static_cast<In>(i));
}
template <typename S, typename In>
__global__ void ThreadEvalOcwKernel(S sch, bool b, const int4* s0, const typename S::Cw* cws, const int4* ocw,
                                    const In* xs, int4* ys, int64_t n) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < n) ys[i] = sch.Eval(b, *s0, cws, *ocw, xs ? xs[i] : static_cast<In>(i));
}
// Level-major keys (what *RelayoutGpu writes), one key per thread: the fss::gpu launchers for caller-defined types.
template <int in_bits, typename Group, typename In, typename Prg, typename XT>
__global__ void DpfLevelMajorEvalKernel(Prg prg, bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra,
                                        const uint64_t* tr_bits, const int4* out_cw, const XT* xs, int4* ys, int64_t n) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k >= n) return;
  thread::DpfLevelMajorView key{cw_s, out_cw, extra ? uint64_t(extra[k]) : tr_bits[k],
                                (!extra && in_bits > 64) ? tr_bits[n + k] : uint64_t(0), n, k, in_bits};
  ys[k] = thread::DpfEvalOn<in_bits, Group, In>(prg, b, seeds[k], key, static_cast<In>(xs[k]));
}
template <int in_bits, typename Group, typename In, typename Prg, typename XT>
__global__ void DcfLevelMajorEvalKernel(Prg prg, bool b, const int4* seeds, const int4* cw_s, const int4* cw_v,
                                        const int4* out_cw, const XT* xs, int4* ys, int64_t n) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k >= n) return;
  thread::DcfLevelMajorView key{cw_s, cw_v, out_cw, n, k, in_bits};
  ys[k] = thread::DcfEvalOn<in_bits, Group, In>(prg, b, seeds[k], key, static_cast<In>(xs[k]));
}
template <int in_bits, typename Group, typename In, typename Prg, typename XT>
__global__ void HtLevelMajorEvalKernel(Prg prg, int4 hash_key, bool b, const int4* seeds, const int4* cw_s,
                                       const uint32_t* extra, const int4* ocws, const XT* xs, int4* ys, int64_t n) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k >= n) return;
  thread::HtLevelMajorView key{cw_s, extra[k], n, k};
  ys[k] = thread::HtEvalOn<in_bits, Group, In>(prg, hash_key, b, seeds[k], key, ocws[k], static_cast<In>(xs[k]));
}
// the full domain of every key of a key-major batch, one thread per (key, x)
template <typename S, typename In, int in_bits>
__global__ void ThreadEvalAllBatchKernel(S sch, bool b, const int4* s0s, const typename S::Cw* cws, const int4* ocws,
                                         int64_t nkeys, int4* ys) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= (nkeys << in_bits)) return;
  const int64_t k = i >> in_bits;
  const In x = static_cast<In>(i & ((int64_t(1) << in_bits) - 1));
  if constexpr (requires { sch.hash_key; }) ys[i] = sch.Eval(b, s0s[k], cws + k * S::kSlots, ocws[k], x);
  else ys[i] = sch.Eval(b, s0s[k], cws + k * S::kSlots, x);
}
// a wide PRG (Aes128MmoWide / Aes128SoftWide), alone or as a scheme's `prg`: only for kernels the caller shapes himself
template <typename T>
constexpr bool UsesWideTables() {
  if constexpr (requires { T::kWideTables; }) return true;
  else if constexpr (requires(const T& t) { t.prg; }) return requires { std::remove_cvref_t<decltype(std::declval<const T&>().prg)>::kWideTables; };
  else return false;
}
template <typename K, typename... Args>
void LaunchThreads(K kernel, int64_t n, hipStream_t stream, Args... args) {
  static_assert(!(UsesWideTables<Args>() || ...), "fss::prg::Aes128MmoWide / Aes128SoftWide need 1024-thread workgroups and InitWorkgroup(): they "
                                                  "work in the caller's own kernels and through the engine (built-in groups), not in this header's "
                                                  "generic 128-thread kernels for caller-defined groups");
  if (n <= 0) return;
  static_assert((std::is_trivially_copyable_v<Args> && ...), "scheme, PRG and group objects are passed to kernels by "
                                                              "value: they must be trivially copyable");
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>((n + 127) / 128)), dim3(128), 0, stream, args...);
  CheckHip(hipGetLastError(), "generic per-thread kernel");
}
}  // namespace detail

// ---------------------------------------------------------------- DPF (include/fss/dpf.cuh:61-303)
template <int in_bits, typename Group, typename Prg, typename In = uint32_t, int par_depth = -1>
  requires(detail::InType<In> && in_bits >= 1 && in_bits <= (int)sizeof(In) * 8 && Groupable<Group> && Prgable<Prg, 2>)
class Dpf {
 public:
  using GroupType = Group;
  Prg prg;
  struct alignas(32) Cw {  // dpf.cuh:76-81
    int4 s;
    bool tr;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits + 1;
  // built-in PRG and group: the engine's kernels; anything else: the generic per-thread kernels of this header
  static constexpr bool kEngine = detail::EnginePrg<Prg> && detail::EngineGroup<Group>;
  fss_amd_ctx* handle() const {
    static_assert(kEngine, "this form needs fss::prg::Aes128Mmo / ChaCha and fss::group::Bytes / Uint<T, mod>");
    return detail::CtxOf(prg, Group::kMod);
  }
  fss_amd_ctx* StageCtx() const {
    if constexpr (kEngine) return handle();
    else return detail::AnyCtx();
  }

  // ---- the reference's signatures.  Host: buffers on the host (or the device), computed on the GPU, synchronous.
  // Device: the calling thread walks the key itself (fss_amd_thread.hpp).
  // dpf.cuh:93: cws[in_bits + 1] out; s0s: the two party seeds; a: alpha; b_buf: beta (clamped)
  __device__ void Gen(Cw cws[], const int4 s0s[2], In a, int4 b_buf) const {
    thread::DpfGen<in_bits, Group, In>(prg, reinterpret_cast<int4*>(cws), s0s, a, b_buf);
  }
  __host__ void Gen(Cw cws[], const int4 s0s[2], In a, int4 b_buf) const {
    detail::Stage st(StageCtx(), 32 + 32 + 32 + sizeof(Cw) * kSlots);
    const int4* d_s0s = st.Load(s0s, 2);
    const In* d_a = st.Value(a);
    const int4* d_b = st.Value(b_buf);
    Cw* d_cws = st.Out(cws, kSlots);
    GenBatch(d_cws, d_s0s, d_a, d_b, 1);
    st.Back(cws, static_cast<const Cw*>(d_cws), kSlots);
  }
  // dpf.cuh:170: party b's share of f(x)
  __device__ int4 Eval(bool b, int4 s0, const Cw cws[], In x) const {
    return thread::DpfEval<in_bits, Group, In>(prg, b, s0, reinterpret_cast<const int4*>(cws), x);
  }
  __host__ int4 Eval(bool b, int4 s0, const Cw cws[], In x) const {
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 32 + 32);
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    const In* d_x = st.Value(x);
    int4* d_y = static_cast<int4*>(st.Take(sizeof(int4)));
    EvalPoints(b, d_s0, d_cws, d_x, d_y, 1);
    int4 y;
    st.Back(&y, static_cast<const int4*>(d_y), 1);
    return y;
  }
  // dpf.cuh:232: ys[x] = Eval(b, s0, cws, x) for every x of the domain, in index order
  void EvalAll(bool b, int4 s0, const Cw cws[], int4 ys[]) const {
    static_assert(in_bits <= 40, "a full-domain output of 2^in_bits int4s");
    const size_t n = size_t(1) << in_bits;
    size_t ws = 0;
    if constexpr (kEngine) ws = EvalAllWorkspace();
    const bool direct = detail::DeviceView(ys) != nullptr;
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + ws + 32 + (direct ? 0 : n * sizeof(int4)));
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    void* d_ws = st.Take(ws);
    int4* d_ys = st.Out(ys, n);
    if constexpr (kEngine) EvalAllBlock(b, d_s0, d_cws, d_ys, d_ws, ws);
    else EvalPoints(b, d_s0, d_cws, nullptr, d_ys, static_cast<int64_t>(n));  // one thread per leaf
    st.Back(ys, static_cast<const int4*>(d_ys), n);
  }

  // ---- batched device forms (every pointer a device pointer; asynchronous on `stream`)
  // nkeys keys at once, one per GPU lane: s0s[2k], s0s[2k+1] the party seeds of key k; cws key-major, kSlots per key.
  void GenBatch(Cw* d_cws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
                hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_dpf_gen(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys, d_cws,
                            stream), "fss_amd_dpf_gen");
    else
      detail::LaunchThreads(detail::ThreadGenKernel<Dpf, In>, nkeys, stream, *this, d_cws, d_s0s, d_alphas, d_betas,
                            nkeys);
  }
  // The same keys written level-major (what gpu::DpfRelayoutGpu would make of d_cws), no intermediate.
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* out_cw, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen_level_major(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                                      nkeys, cw_s, tr_bits, out_cw, stream), "fss_amd_dpf_gen_level_major");
  }
  // ys[i] = Eval(b, *d_s0, d_cws, xs[i]): one key, a whole batch of points
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_dpf_eval_points(handle(), in_bits, Group::kId, b, d_s0, d_cws, d_xs, detail::XBytes<In>(), n, d_ys,
                                    stream), "fss_amd_dpf_eval_points");
    else
      detail::LaunchThreads(detail::ThreadEvalKernel<Dpf, In>, n, stream, *this, b, d_s0, d_cws, d_xs, d_ys, n);
  }
  // The same on a batch in PINNED host memory: the walk reads xs_host and writes ys_host across PCIe itself, one launch,
  // no staging copies (fss_amd_eval_points_host); d_s0 / d_cws stay device pointers; asynchronous on `stream`.
  void EvalPointsHost(bool b, const int4* d_s0, const Cw* d_cws, const In* xs_host, int4* ys_host, int64_t n,
                      hipStream_t stream = 0) const {
    Check(fss_amd_eval_points_host(handle(), FSS_AMD_DPF, in_bits, Group::kId, b, d_s0, d_cws, nullptr, xs_host,
                                   detail::XBytes<In>(), n, ys_host, stream), "fss_amd_eval_points_host");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_DPF, in_bits, nleaves);
  }
  // ys[j] = Eval(b, *d_s0, d_cws, first_leaf + j): an aligned power-of-two block of the domain (a GPU's share)
  void EvalAllBlock(bool b, const int4* d_s0, const Cw* d_cws, int4* d_ys, void* d_workspace, size_t ws_bytes,
                    int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_eval_all(handle(), in_bits, Group::kId, b, d_s0, d_cws, first_leaf, nleaves, d_ys, d_workspace,
                               ws_bytes, stream), "fss_amd_dpf_eval_all");
  }
};

// ---------------------------------------------------------------- DCF (include/fss/dcf.cuh:74-385)
template <int in_bits, typename Group, typename Prg, typename In = uint32_t, DcfPred pred = DcfPred::kLt,
          int par_depth = -1>
  requires(detail::InType<In> && in_bits >= 1 && in_bits <= (int)sizeof(In) * 8 && Groupable<Group> && Prgable<Prg, 4>)
class Dcf {
 public:
  using GroupType = Group;
  Prg prg;
  struct alignas(32) Cw {  // dcf.cuh:91-96
    int4 s;
    int4 v;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits + 1;
  static constexpr bool kEngine = detail::EnginePrg<Prg> && detail::EngineGroup<Group>;
  fss_amd_ctx* handle() const {
    static_assert(kEngine, "this form needs fss::prg::Aes128Mmo / ChaCha and fss::group::Bytes / Uint<T, mod>");
    return detail::CtxOf(prg, Group::kMod);
  }
  fss_amd_ctx* StageCtx() const {
    if constexpr (kEngine) return handle();
    else return detail::AnyCtx();
  }

  __device__ void Gen(Cw cws[], const int4 s0s[2], In a, int4 b_buf) const {  // dcf.cuh:108
    thread::DcfGen<in_bits, Group, In, pred == DcfPred::kGt>(prg, reinterpret_cast<int4*>(cws), s0s, a, b_buf);
  }
  __host__ void Gen(Cw cws[], const int4 s0s[2], In a, int4 b_buf) const {
    detail::Stage st(StageCtx(), 32 + 32 + 32 + sizeof(Cw) * kSlots);
    const int4* d_s0s = st.Load(s0s, 2);
    const In* d_a = st.Value(a);
    const int4* d_b = st.Value(b_buf);
    Cw* d_cws = st.Out(cws, kSlots);
    GenBatch(d_cws, d_s0s, d_a, d_b, 1);
    st.Back(cws, static_cast<const Cw*>(d_cws), kSlots);
  }
  __device__ int4 Eval(bool b, int4 s0, const Cw cws[], In x) const {  // dcf.cuh:205
    return thread::DcfEval<in_bits, Group, In>(prg, b, s0, reinterpret_cast<const int4*>(cws), x);
  }
  __host__ int4 Eval(bool b, int4 s0, const Cw cws[], In x) const {
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 32 + 32);
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    const In* d_x = st.Value(x);
    int4* d_y = static_cast<int4*>(st.Take(sizeof(int4)));
    EvalPoints(b, d_s0, d_cws, d_x, d_y, 1);
    int4 y;
    st.Back(&y, static_cast<const int4*>(d_y), 1);
    return y;
  }
  void EvalAll(bool b, int4 s0, const Cw cws[], int4 ys[]) const {  // dcf.cuh:294
    static_assert(in_bits <= 40, "a full-domain output of 2^in_bits int4s");
    const size_t n = size_t(1) << in_bits;
    size_t ws = 0;
    if constexpr (kEngine) ws = EvalAllWorkspace();
    const bool direct = detail::DeviceView(ys) != nullptr;
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + ws + 32 + (direct ? 0 : n * sizeof(int4)));
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    void* d_ws = st.Take(ws);
    int4* d_ys = st.Out(ys, n);
    if constexpr (kEngine) EvalAllBlock(b, d_s0, d_cws, d_ys, d_ws, ws);
    else EvalPoints(b, d_s0, d_cws, nullptr, d_ys, static_cast<int64_t>(n));  // one thread per leaf
    st.Back(ys, static_cast<const int4*>(d_ys), n);
  }

  void GenBatch(Cw* d_cws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
                hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_dcf_gen(handle(), in_bits, Group::kId, (int)pred, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                            nkeys, d_cws, stream), "fss_amd_dcf_gen");
    else
      detail::LaunchThreads(detail::ThreadGenKernel<Dcf, In>, nkeys, stream, *this, d_cws, d_s0s, d_alphas, d_betas,
                            nkeys);
  }
  void GenLevelMajor(int4* cw_s, int4* cw_v, int4* out_cw, const int4* d_s0s, const In* d_alphas, const int4* d_betas,
                     int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dcf_gen_level_major(handle(), in_bits, Group::kId, (int)pred, d_s0s, d_alphas, detail::XBytes<In>(),
                                      d_betas, nkeys, cw_s, cw_v, out_cw, stream), "fss_amd_dcf_gen_level_major");
  }
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_dcf_eval_points(handle(), in_bits, Group::kId, b, d_s0, d_cws, d_xs, detail::XBytes<In>(), n, d_ys,
                                    stream), "fss_amd_dcf_eval_points");
    else
      detail::LaunchThreads(detail::ThreadEvalKernel<Dcf, In>, n, stream, *this, b, d_s0, d_cws, d_xs, d_ys, n);
  }
  // The same on a batch in PINNED host memory: the walk reads xs_host and writes ys_host across PCIe itself, one launch,
  // no staging copies (fss_amd_eval_points_host); d_s0 / d_cws stay device pointers; asynchronous on `stream`.
  void EvalPointsHost(bool b, const int4* d_s0, const Cw* d_cws, const In* xs_host, int4* ys_host, int64_t n,
                      hipStream_t stream = 0) const {
    Check(fss_amd_eval_points_host(handle(), FSS_AMD_DCF, in_bits, Group::kId, b, d_s0, d_cws, nullptr, xs_host,
                                   detail::XBytes<In>(), n, ys_host, stream), "fss_amd_eval_points_host");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_DCF, in_bits, nleaves);
  }
  void EvalAllBlock(bool b, const int4* d_s0, const Cw* d_cws, int4* d_ys, void* d_workspace, size_t ws_bytes,
                    int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_dcf_eval_all(handle(), in_bits, Group::kId, b, d_s0, d_cws, first_leaf, nleaves, d_ys, d_workspace,
                               ws_bytes, stream), "fss_amd_dcf_eval_all");
  }
};

// ---------------------------------------------------------------- Half-Tree DPF (include/fss/half_tree_dpf.cuh:39-354)
template <int in_bits, typename Group, typename Prg, typename In = uint32_t, int par_depth = -1>
  requires(detail::InType<In> && in_bits >= 1 && in_bits <= (int)sizeof(In) * 8 && Groupable<Group> && Prgable<Prg, 1>)
class HalfTreeDpf {
 public:
  using GroupType = Group;
  Prg prg;
  int4 hash_key;           // half_tree_dpf.cuh:45
  struct alignas(32) Cw {  // half_tree_dpf.cuh:53-57
    int4 s;
    bool extra;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits;
  static constexpr bool kEngine = detail::EnginePrg<Prg> && detail::EngineGroup<Group>;
  fss_amd_ctx* handle() const {
    static_assert(kEngine, "this form needs fss::prg::Aes128Mmo / ChaCha and fss::group::Bytes / Uint<T, mod>");
    return detail::CtxOf(prg, Group::kMod, &hash_key);
  }
  fss_amd_ctx* StageCtx() const {
    if constexpr (kEngine) return handle();
    else return detail::AnyCtx();
  }

  __device__ void Gen(Cw cws[], int4& ocw, const int4 s0s[2], In a, int4 b_buf) const {  // half_tree_dpf.cuh:68
    thread::HtGen<in_bits, Group, In>(prg, hash_key, reinterpret_cast<int4*>(cws), ocw, s0s, a, b_buf);
  }
  __host__ void Gen(Cw cws[], int4& ocw, const int4 s0s[2], In a, int4 b_buf) const {
    detail::Stage st(StageCtx(), 32 + 32 + 32 + sizeof(Cw) * kSlots + 32);
    const int4* d_s0s = st.Load(s0s, 2);
    const In* d_a = st.Value(a);
    const int4* d_b = st.Value(b_buf);
    Cw* d_cws = st.Out(cws, kSlots);
    int4* d_ocw = static_cast<int4*>(st.Take(sizeof(int4)));
    GenBatch(d_cws, d_ocw, d_s0s, d_a, d_b, 1);
    st.Back(cws, static_cast<const Cw*>(d_cws), kSlots);
    st.Back(&ocw, static_cast<const int4*>(d_ocw), 1);
  }
  __device__ int4 Eval(bool b, int4 s0, const Cw cws[], int4 ocw, In x) const {  // half_tree_dpf.cuh:187
    return thread::HtEval<in_bits, Group, In>(prg, hash_key, b, s0, reinterpret_cast<const int4*>(cws), ocw, x);
  }
  __host__ int4 Eval(bool b, int4 s0, const Cw cws[], int4 ocw, In x) const {
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 32 + 32 + 32);
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    const int4* d_ocw = st.Value(ocw);
    const In* d_x = st.Value(x);
    int4* d_y = static_cast<int4*>(st.Take(sizeof(int4)));
    EvalPoints(b, d_s0, d_cws, d_ocw, d_x, d_y, 1);
    int4 y;
    st.Back(&y, static_cast<const int4*>(d_y), 1);
    return y;
  }
  void EvalAll(bool b, int4 s0, const Cw cws[], int4 ocw, int4 ys[]) const {  // half_tree_dpf.cuh:246
    static_assert(in_bits <= 40, "a full-domain output of 2^in_bits int4s");
    const size_t n = size_t(1) << in_bits;
    size_t ws = 0;
    if constexpr (kEngine) ws = EvalAllWorkspace();
    const bool direct = detail::DeviceView(ys) != nullptr;
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 32 + ws + 32 + (direct ? 0 : n * sizeof(int4)));
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    const int4* d_ocw = st.Value(ocw);
    void* d_ws = st.Take(ws);
    int4* d_ys = st.Out(ys, n);
    if constexpr (kEngine) EvalAllBlock(b, d_s0, d_cws, d_ocw, d_ys, d_ws, ws);
    else EvalPoints(b, d_s0, d_cws, d_ocw, nullptr, d_ys, static_cast<int64_t>(n));  // one thread per leaf
    st.Back(ys, static_cast<const int4*>(d_ys), n);
  }

  void GenBatch(Cw* d_cws, int4* d_ocws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
                hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_ht_gen(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys, d_cws,
                           d_ocws, stream), "fss_amd_ht_gen");
    else
      detail::LaunchThreads(detail::ThreadGenOcwKernel<HalfTreeDpf, In>, nkeys, stream, *this, d_cws, d_ocws, d_s0s,
                            d_alphas, d_betas, nkeys);
  }
  void GenLevelMajor(int4* cw_s, uint32_t* extra, int4* d_ocws, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_ht_gen_level_major(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                                     nkeys, cw_s, extra, d_ocws, stream), "fss_amd_ht_gen_level_major");
  }
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_ocw, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_ht_eval_points(handle(), in_bits, Group::kId, b, d_s0, d_cws, d_ocw, d_xs, detail::XBytes<In>(), n,
                                   d_ys, stream), "fss_amd_ht_eval_points");
    else
      detail::LaunchThreads(detail::ThreadEvalOcwKernel<HalfTreeDpf, In>, n, stream, *this, b, d_s0, d_cws, d_ocw, d_xs,
                            d_ys, n);
  }
  // The same on a batch in PINNED host memory (fss_amd_eval_points_host): x read and y written across PCIe by the walk.
  void EvalPointsHost(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_ocw, const In* xs_host, int4* ys_host,
                      int64_t n, hipStream_t stream = 0) const {
    Check(fss_amd_eval_points_host(handle(), FSS_AMD_HT, in_bits, Group::kId, b, d_s0, d_cws, d_ocw, xs_host,
                                   detail::XBytes<In>(), n, ys_host, stream), "fss_amd_eval_points_host");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_HT, in_bits, nleaves);
  }
  void EvalAllBlock(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_ocw, int4* d_ys, void* d_workspace,
                    size_t ws_bytes, int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits,
                    hipStream_t stream = 0) const {
    Check(fss_amd_ht_eval_all(handle(), in_bits, Group::kId, b, d_s0, d_cws, d_ocw, first_leaf, nleaves, d_ys,
                              d_workspace, ws_bytes, stream), "fss_amd_ht_eval_all");
  }
};

// ---------------------------------------------------------------- Grotto DCF (include/fss/grotto_dcf.cuh:45-238)
// Keys are DPF keys over Bytes with beta = 0; shares are bools.
template <int in_bits, typename Prg, typename In = uint32_t, int par_depth = -1>
  requires(detail::InType<In> && in_bits >= 1 && in_bits <= (int)sizeof(In) * 8 && Prgable<Prg, 2>)
class GrottoDcf {
  using DpfType = Dpf<in_bits, group::Bytes, Prg, In, par_depth>;
  static_assert(sizeof(bool) == 1);

 public:
  using Cw = typename DpfType::Cw;
  Prg prg;
  static constexpr int kSlots = in_bits + 1;
  static constexpr bool kEngine = detail::EnginePrg<Prg>;
  fss_amd_ctx* handle() const {
    static_assert(kEngine, "the parity table, EvalAll and the batched forms need fss::prg::Aes128Mmo / ChaCha");
    return detail::CtxOf(prg);
  }

  // grotto_dcf.cuh:63-67
  __host__ __device__ void Gen(Cw cws[], const int4 s0s[2], In a) const {
    DpfType{prg}.Gen(cws, s0s, a, int4{0, 0, 0, 0});
  }

  // grotto_dcf.cuh:76-84: p = 2N-1 bools, level order (root p[0], leaf x at p[x + N - 1]); b = the party
  struct ParityTree {
    bool* p;
    bool b;
  };
  static constexpr size_t ParityTreeBytes() { return (size_t(2) << in_bits) - 1; }

  // grotto_dcf.cuh:94-104.  pt.p: 2N-1 bools on the host or on the device, pt.b set by the caller.
  void Preprocess(ParityTree& pt, int4 s0, const Cw cws[]) const {
    static_assert(in_bits <= 40, "a parity table of 2^(in_bits + 1) bytes");
    fss_amd_ctx* ctx = handle();
    const size_t ws = PreprocessWorkspace();
    const bool direct = detail::OnDevice(pt.p);
    detail::Stage st(ctx, 32 + sizeof(Cw) * kSlots + ws + 64 + (direct ? 0 : ParityTreeBytes()));
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    void* d_ws = st.Take(ws);
    bool* d_pt = st.OutReadBack(pt.p, ParityTreeBytes());
    PreprocessDevice(pt.b, reinterpret_cast<uint8_t*>(d_pt), d_s0, d_cws, d_ws, ws);
    st.Back(pt.p, static_cast<const bool*>(d_pt), ParityTreeBytes());
  }
  // grotto_dcf.cuh:116-135: the share of 1[alpha <= x] -- no PRG call, a walk down the finished table: XOR of the
  // left children where e = x + 1 has a 1 bit.  Runs where it is called, on the table it is given.
  __host__ __device__ static bool Eval(const ParityTree& pt, In x) {
    const unsigned __int128 e = static_cast<unsigned __int128>(x) + 1;
    if (e >> in_bits) return pt.p[0];  // x = N - 1: the whole domain
    bool share = false;
    size_t node = 0;
    for (int level = in_bits - 1; level >= 0; --level) {
      const size_t left = 2 * node + 1;
      if ((e >> level) & 1) {
        share ^= pt.p[left];
        node = left + 1;
      } else {
        node = left;
      }
    }
    return share;
  }
  // grotto_dcf.cuh:151-163: ys[x] = party b's share of 1[alpha <= x] for the whole domain
  void EvalAll(bool b, int4 s0, const Cw cws[], bool ys[]) const {
    static_assert(in_bits <= 40, "a full-domain output of 2^in_bits bools");
    const size_t n = size_t(1) << in_bits, ws = EvalAllWorkspace();
    fss_amd_ctx* ctx = handle();
    const bool direct = detail::OnDevice(ys);
    detail::Stage st(ctx, 32 + sizeof(Cw) * kSlots + ws + 64 + (direct ? 0 : n));
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws, kSlots);
    void* d_ws = st.Take(ws);
    bool* d_ys = st.OutReadBack(ys, n);
    EvalAllBlock(b, d_s0, d_cws, reinterpret_cast<uint8_t*>(d_ys), d_ws, ws);
    st.Back(ys, static_cast<const bool*>(d_ys), n);
  }

  // ---- batched device forms
  void GenBatch(Cw* d_cws, const int4* d_s0s, const In* d_alphas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen(handle(), in_bits, group::Bytes::kId, d_s0s, d_alphas, detail::XBytes<In>(), nullptr, nkeys,
                          d_cws, stream), "fss_amd_dpf_gen");
  }
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* out_cw, const int4* d_s0s, const In* d_alphas, int64_t nkeys,
                     hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen_level_major(handle(), in_bits, group::Bytes::kId, d_s0s, d_alphas, detail::XBytes<In>(),
                                      nullptr, nkeys, cw_s, tr_bits, out_cw, stream), "fss_amd_dpf_gen_level_major");
  }
  static size_t PreprocessWorkspace() {
    return fss_amd_eval_all_workspace(FSS_AMD_GROTTO, in_bits, int64_t(1) << in_bits) + (size_t(1) << in_bits);
  }
  void PreprocessDevice(bool b, uint8_t* d_pt, const int4* d_s0, const Cw* d_cws, void* d_workspace, size_t ws_bytes,
                        hipStream_t stream = 0) const {
    Check(fss_amd_grotto_preprocess(handle(), in_bits, b, d_s0, d_cws, d_pt, d_workspace, ws_bytes, stream),
          "fss_amd_grotto_preprocess");
  }
  // GrottoDcf::Eval for a batch of x against one parity table on the device
  void EvalPoints(const uint8_t* d_pt, const In* d_xs, uint8_t* d_ys, int64_t n, hipStream_t stream = 0) const {
    Check(fss_amd_grotto_eval_table(handle(), in_bits, d_pt, d_xs, detail::XBytes<In>(), n, d_ys, stream),
          "fss_amd_grotto_eval_table");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_GROTTO, in_bits, nleaves);
  }
  // a sub-block needs the carry of the blocks before it (fss_amd/shard.py: grotto_fix_shard)
  void EvalAllBlock(bool b, const int4* d_s0, const Cw* d_cws, uint8_t* d_ys, void* d_workspace, size_t ws_bytes,
                    int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_grotto_eval_all(handle(), in_bits, b, d_s0, d_cws, first_leaf, nleaves, d_ys, d_workspace, ws_bytes,
                                  stream), "fss_amd_grotto_eval_all");
  }
  // Path-walk evaluator on level-major keys, one point per key, for domains with no table (in_bits up to 64).
  // Reconstructs the predicate, but a party's share is NOT the table share of Eval / EvalAll: both parties must
  // use this evaluator for a given (key, x) (see fss_amd.h).
  void EvalKeysWalk(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const In* xs, uint8_t* ys,
                    int64_t nkeys, hipStream_t stream = 0) const {
    static_assert(in_bits <= 64);
    Check(fss_amd_grotto_eval_keys(handle(), in_bits, b, seeds, cw_s, tr_bits, xs, detail::XBytes<In>(), nkeys, ys,
                                   stream), "fss_amd_grotto_eval_keys");
  }
};

// ---------------------------------------------------------------- verifiable DPF (include/fss/vdpf.cuh:57-404)
// (fss::hash::Blake3 / Sha256: include/fss_amd_thread.hpp)
namespace detail {
// hash types the engine's kernels implement: keyed BLAKE3 / keyed SHA-256, one key for H and H' per context
template <typename H>
concept EngineHash = requires(const H& h) {
  { H::kEngineHashKind } -> std::convertible_to<int>;
  { h.SameKey(h) } -> std::convertible_to<bool>;
};
inline const int32_t* HashKeyWords(const hash::Blake3& h) { return h.iv; }
inline const int32_t* HashKeyWords(const hash::Sha256& h) { return &h.key.x; }

// generic per-thread kernels for caller-defined PRG / group / hash types (one key, or one point, per thread)
template <typename S, typename In>
__global__ void ThreadVdpfGenKernel(S sch, typename S::Cw* cws, std::array<int4, 4>* cs, int4* ocws, const int4* s0s, const In* alphas,
                                    const int4* betas, int64_t nkeys, int32_t* retry) {
  const int64_t k = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (k >= nkeys) return;
  const int r = sch.Gen(cws + k * S::kSlots, cs[k], ocws[k], std::span<const int4, 2>(s0s + 2 * k, 2), alphas[k], betas[k]);
  if (retry) retry[k] = r;
}
template <typename S, typename In>
__global__ void ThreadVdpfEvalKernel(S sch, bool b, const int4* s0, const typename S::Cw* cws, const int4* cs, const int4* ocw,
                                     const In* xs, int4* ys, std::array<int4, 4>* pis, int64_t n) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  int4 y;
  pis[i] = sch.Eval(b, *s0, std::span<const typename S::Cw>(cws, S::kSlots), std::span<const int4, 4>(cs, 4), *ocw,
                    xs ? xs[i] : static_cast<In>(i), y);
  ys[i] = y;
}
}  // namespace detail

// The reference's class, parameter for parameter: `fss::Vdpf<in_bits, Group, Prg, XorHash, Hash, In, par_depth> v{prg,
// xor_hash, hash}` with public members and Gen / Eval / Prove / Verify / EvalAll in its signatures (cuda::std::array /
// span / tuple spelled std::).  A key = in_bits Cw slots + cs (4 int4) + ocw.
//  * Host forms stage onto the engine (fss_amd_vdpf_*) when Prg, Group and both hashes are engine types and the two hash
//    objects hold the same key (a context has one hash pair; the reference's tests and benches construct both from one
//    IV); other types run the per-thread code in a generic kernel.  Prove folds on the host (a sequential hash chain:
//    nothing for a GPU to do) unless the hashes it is given already live in device memory.
//  * Device forms: the calling thread walks the key itself (include/fss_amd_thread.hpp), as src/bench_gpu.cu:172-205.
//  * GenBatch / GenLevelMajor / EvalPoints / ProveChain / ProveKeys: the batched device forms.
template <int in_bits, typename Group, typename Prg, typename XorHash = hash::Blake3, typename Hash = XorHash, typename In = uint32_t,
          int par_depth = -1>
  requires(detail::InType<In> && in_bits >= 1 && in_bits <= (int)sizeof(In) * 8 && Groupable<Group> && Prgable<Prg, 2> &&
           XorHashable<XorHash> && Hashable<Hash>)
class Vdpf {
 public:
  using GroupType = Group;
  Prg prg;
  XorHash xor_hash;
  Hash hash;
  struct alignas(32) Cw {  // vdpf.cuh:77-82
    int4 s;
    bool tr;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits;
  static constexpr bool kEngine = detail::EnginePrg<Prg> && detail::EngineGroup<Group> && detail::EngineHash<XorHash> && std::is_same_v<XorHash, Hash>;
  fss_amd_ctx* handle() const {
    static_assert(kEngine, "this form needs fss::prg::Aes128Mmo / ChaCha, fss::group::Bytes / Uint<T, mod> and fss::hash::Blake3 / Sha256 for both hashes");
    if (!xor_hash.SameKey(hash))
      throw Error("fss::Vdpf: xor_hash and hash hold different keys (an engine context has one hash key for H and H')", FSS_AMD_EINVAL);
    return detail::CtxOf(prg, Group::kMod, nullptr, detail::HashKeyWords(xor_hash), XorHash::kEngineHashKind);
  }
  fss_amd_ctx* StageCtx() const {
    if constexpr (kEngine) return handle();
    else return detail::AnyCtx();
  }

  // ---- the reference's signatures ----------------------------------------------------------------------------------------
  // vdpf.cuh:103: cws[in_bits], cs and ocw out; returns 1 when the leaf control bits agree (resample the seeds, retry)
  __device__ int Gen(Cw cws[], std::array<int4, 4>& cs, int4& ocw, std::span<const int4, 2> s0s, In a, int4 b_buf) const {
    const int4 seeds[2] = {s0s[0], s0s[1]};
    return thread::VdpfGen<in_bits, Group, In>(prg, xor_hash, reinterpret_cast<int4*>(cws), cs, ocw, seeds, a, b_buf);
  }
  __host__ int Gen(Cw cws[], std::array<int4, 4>& cs, int4& ocw, std::span<const int4, 2> s0s, In a, int4 b_buf) const {
    detail::Stage st(StageCtx(), 32 + 32 + 32 + sizeof(Cw) * kSlots + 64 + 32 + 32);
    const int4* d_s0s = st.Load(s0s.data(), 2);
    const In* d_a = st.Value(a);
    const int4* d_b = st.Value(b_buf);
    Cw* d_cws = st.Out(cws, kSlots);
    int4* d_cs = static_cast<int4*>(st.Take(64));
    int4* d_ocw = static_cast<int4*>(st.Take(16));
    int32_t* d_retry = static_cast<int32_t*>(st.Take(4));
    GenBatch(d_cws, d_cs, d_ocw, d_s0s, d_a, d_b, 1, d_retry);
    st.Back(cws, static_cast<const Cw*>(d_cws), kSlots);
    int32_t retry = 0;
    int4 new_ocw;
    st.Back(cs.data(), static_cast<const int4*>(d_cs), 4);
    st.Back(&new_ocw, static_cast<const int4*>(d_ocw), 1);
    st.Back(&retry, static_cast<const int32_t*>(d_retry), 1);
    if (retry == 0) ocw = new_ocw;  // (the reference leaves ocw untouched when it asks for a retry, vdpf.cuh:171-178)
    return retry;
  }
  // vdpf.cuh:204: y = party b's share of f(x); returns the corrected per-point hash for Prove
  __device__ std::array<int4, 4> Eval(bool b, int4 s0, std::span<const Cw> cws, std::span<const int4, 4> cs, int4 ocw, In x, int4& y) const {
    return thread::VdpfEval<in_bits, Group, In>(prg, xor_hash, b, s0, reinterpret_cast<const int4*>(cws.data()), cs.data(), ocw, x, y);
  }
  __host__ std::array<int4, 4> Eval(bool b, int4 s0, std::span<const Cw> cws, std::span<const int4, 4> cs, int4 ocw, In x, int4& y) const {
    if (cws.size() < static_cast<size_t>(kSlots)) throw Error("fss::Vdpf::Eval: cws must hold in_bits correction words", FSS_AMD_EINVAL);
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 64 + 32 + 32 + 32 + 64);
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws.data(), kSlots);
    const int4* d_cs = st.Load(cs.data(), 4);
    const int4* d_ocw = st.Value(ocw);
    const In* d_x = st.Value(x);
    int4* d_y = static_cast<int4*>(st.Take(16));
    int4* d_pi = static_cast<int4*>(st.Take(64));
    EvalPoints(b, d_s0, d_cws, d_cs, d_ocw, d_x, d_y, d_pi, 1);
    std::array<int4, 4> pi;
    st.Back(&y, static_cast<const int4*>(d_y), 1);
    st.Back(pi.data(), static_cast<const int4*>(d_pi), 4);
    return pi;
  }
  // vdpf.cuh:268: pi = cs; for every per-point hash: pi[0..1] ^= H'(pi ^ pi_tilde)
  void Prove(std::span<const std::array<int4, 4>> pi_tildes, std::span<const int4, 4> cs, std::array<int4, 4>& pi) const {
    if constexpr (kEngine) {
      if (!pi_tildes.empty() && detail::OnDevice(pi_tildes.data())) {  // the hashes are where EvalPoints left them: fold there
        fss_amd_ctx* ctx = handle();
        detail::Stage st(ctx, 64 + 64 + 64);
        const int4* d_cs = st.Load(cs.data(), 4);
        int4* d_pi = static_cast<int4*>(st.Take(64));
        ProveChain(reinterpret_cast<const int4*>(pi_tildes.data()), static_cast<int64_t>(pi_tildes.size()), d_cs, d_pi);
        st.Back(pi.data(), static_cast<const int4*>(d_pi), 4);
        return;
      }
    }
    Hash h = hash;
    pi = {cs[0], cs[1], cs[2], cs[3]};
    for (const std::array<int4, 4>& pt : pi_tildes) {
      const std::array<int4, 4> in = util::Xor(std::span<const int4, 4>(pi), std::span<const int4, 4>(pt));
      const std::array<int4, 2> out = h.Hash(std::span<const int4, 4>(in));
      pi[0] = util::Xor(pi[0], out[0]);
      pi[1] = util::Xor(pi[1], out[1]);
    }
  }
  // vdpf.cuh:287: accept when the two parties' proofs agree
  __host__ __device__ static bool Verify(std::span<const int4, 4> pi0, std::span<const int4, 4> pi1) {
    bool same = true;
    for (int i = 0; i < 4; ++i)
      same = same && pi0[i].x == pi1[i].x && pi0[i].y == pi1[i].y && pi0[i].z == pi1[i].z && pi0[i].w == pi1[i].w;
    return same;
  }
  static bool Verify(const int4 pi0[4], const int4 pi1[4]) { return Verify(std::span<const int4, 4>(pi0, 4), std::span<const int4, 4>(pi1, 4)); }
  // vdpf.cuh:313: every x of the domain in index order, and the proof over all of them
  void EvalAll(bool b, int4 s0, std::span<const Cw> cws, std::span<const int4, 4> cs, int4 ocw, std::span<int4> ys, std::array<int4, 4>& pi) const {
    static_assert(in_bits <= 26, "a full-domain VDPF evaluation keeps 80 B per leaf on the device");
    const size_t n = size_t(1) << in_bits;
    if (cws.size() < static_cast<size_t>(kSlots) || ys.size() < n) throw Error("fss::Vdpf::EvalAll: cws / ys too small", FSS_AMD_EINVAL);
    std::vector<In> xs(n);
    for (size_t i = 0; i < n; ++i) xs[i] = static_cast<In>(i);
    const bool direct = detail::DeviceView(ys.data()) != nullptr;
    detail::Stage st(StageCtx(), 32 + sizeof(Cw) * kSlots + 64 + 32 + n * sizeof(In) + 32 + n * 64 + (direct ? 0 : n * 16) + 64);
    const int4* d_s0 = st.Value(s0);
    const Cw* d_cws = st.Load(cws.data(), kSlots);
    const int4* d_cs = st.Load(cs.data(), 4);
    const int4* d_ocw = st.Value(ocw);
    const In* d_xs = st.Load(xs.data(), n);
    int4* d_ys = st.Out(ys.data(), n);
    int4* d_pt = static_cast<int4*>(st.Take(n * 64));
    EvalPoints(b, d_s0, d_cws, d_cs, d_ocw, d_xs, d_ys, d_pt, static_cast<int64_t>(n));
    if constexpr (kEngine) {
      int4* d_pi = static_cast<int4*>(st.Take(64));
      ProveChain(d_pt, static_cast<int64_t>(n), d_cs, d_pi);
      st.Back(pi.data(), static_cast<const int4*>(d_pi), 4);
    } else {
      std::vector<std::array<int4, 4>> pts(n);
      CheckHip(hipMemcpy(pts.data(), d_pt, n * 64, hipMemcpyDeviceToHost), "hipMemcpy (per-point hashes)");
      Prove(std::span<const std::array<int4, 4>>(pts), cs, pi);
    }
    st.Back(ys.data(), static_cast<const int4*>(d_ys), n);
  }

  // ---- batched device forms (every pointer a device pointer; asynchronous on `stream`) --------------------------------
  // Vdpf::Gen for nkeys keys; d_retry[k] = its return value (may be null).  d_cs: 4 int4 per key.
  void GenBatch(Cw* d_cws, int4* d_cs, int4* d_ocws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
                int32_t* d_retry = nullptr, hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_vdpf_gen(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys, d_cws, d_cs, d_ocws,
                             d_retry, stream), "fss_amd_vdpf_gen");
    else
      detail::LaunchThreads(detail::ThreadVdpfGenKernel<Vdpf, In>, nkeys, stream, *this, d_cws, reinterpret_cast<std::array<int4, 4>*>(d_cs),
                            d_ocws, d_s0s, d_alphas, d_betas, nkeys, d_retry);
  }
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* d_cs, int4* d_ocws, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, int32_t* d_retry = nullptr, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_gen_level_major(handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                                       nkeys, cw_s, tr_bits, d_cs, d_ocws, d_retry, stream),
          "fss_amd_vdpf_gen_level_major");
  }
  // Vdpf::Eval of one key at n points: ys[i], pi_tildes[4i..4i+3]
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_cs, const int4* d_ocw, const In* d_xs,
                  int4* d_ys, int4* d_pi_tildes, int64_t n, hipStream_t stream = 0) const {
    if constexpr (kEngine)
      Check(fss_amd_vdpf_eval_points(handle(), in_bits, Group::kId, b, d_s0, d_cws, d_cs, d_ocw, d_xs, detail::XBytes<In>(), n, d_ys,
                                     d_pi_tildes, stream), "fss_amd_vdpf_eval_points");
    else
      detail::LaunchThreads(detail::ThreadVdpfEvalKernel<Vdpf, In>, n, stream, *this, b, d_s0, d_cws, d_cs, d_ocw, d_xs, d_ys,
                            reinterpret_cast<std::array<int4, 4>*>(d_pi_tildes), n);
  }
  // Vdpf::Prove over the n hashes of one key, on the device (a sequential chain run by a single lane)
  void ProveChain(const int4* d_pi_tildes, int64_t n, const int4* d_cs, int4* d_pi, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_prove(handle(), d_pi_tildes, n, d_cs, d_pi, stream), "fss_amd_vdpf_prove");
  }
  // ... and for nkeys keys with one evaluation each
  void ProveKeys(const int4* d_pi_tildes, const int4* d_cs, int64_t nkeys, int4* d_pis, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_prove_keys(handle(), d_pi_tildes, d_cs, nkeys, d_pis, stream), "fss_amd_vdpf_prove_keys");
  }
};

// ---------------------------------------------------------------- fss::gpu launchers
// Device pointers owned by the caller, asynchronous on `stream`, nothing synchronises (the reference's contract,
// point_eval_gpu.cuh / eval_all_gpu.cuh).  `unroll`, `z`, `b1`, `bs` are accepted so existing call sites compile;
// the engine sizes its own launches.  A failed launch throws fss::Error where the reference would leave it to
// cudaPeekAtLastError.
namespace gpu {
namespace detail_gpu {
using detail::AnyCtx;  // relayout needs no key material
}  // namespace detail_gpu

// point_eval_gpu.cuh:324-331
template <int in_bits, typename Group, typename Prg, typename In>
void HalfTreeDpfRelayoutGpu(const typename HalfTreeDpf<in_bits, Group, Prg, In>::Cw* cws, int nkeys, int4* cw_s,
                            uint32_t* extra, hipStream_t stream = 0) {
  Check(fss_amd_ht_relayout(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, extra, stream), "fss_amd_ht_relayout");
}
// point_eval_gpu.cuh:346-353: extra[k] bit i = tr of level i (in_bits <= 32, as the reference)
template <int in_bits, typename Group, typename Prg, typename In>
void DpfRelayoutGpu(const typename Dpf<in_bits, Group, Prg, In>::Cw* cws, int nkeys, int4* cw_s, uint32_t* extra,
                    int4* out_cw, hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t extra holds 32 levels; pass uint64_t* tr_bits for in_bits > 32");
  Check(fss_amd_dpf_relayout_u32(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, extra, out_cw, stream),
        "fss_amd_dpf_relayout_u32");
}
// ... with 64 tr bits per key (two planes of nkeys for in_bits > 64): any in_bits
template <int in_bits, typename Group, typename Prg, typename In>
void DpfRelayoutGpu(const typename Dpf<in_bits, Group, Prg, In>::Cw* cws, int64_t nkeys, int4* cw_s, uint64_t* tr_bits,
                    int4* out_cw, hipStream_t stream = 0) {
  Check(fss_amd_dpf_relayout(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, tr_bits, out_cw, stream),
        "fss_amd_dpf_relayout");
}
// point_eval_gpu.cuh:368-375
template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred = DcfPred::kLt>
void DcfRelayoutGpu(const typename Dcf<in_bits, Group, Prg, In, pred>::Cw* cws, int nkeys, int4* cw_s, int4* cw_v,
                    int4* out_cw, hipStream_t stream = 0) {
  Check(fss_amd_dcf_relayout(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, cw_v, out_cw, stream),
        "fss_amd_dcf_relayout");
}

// point_eval_gpu.cuh:416-429  ys[k] = HalfTreeDpf::Eval(b, seeds[k], key k, ocws[k], xs[k])
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void HalfTreeDpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra, const int4* ocws,
                             const uint32_t* xs, int4* ys, int nkeys,
                             const HalfTreeDpf<in_bits, Group, Prg, In, par_depth>& dpf, hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t xs cover in_bits <= 32; pass const In* xs");
  using S = HalfTreeDpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_ht_eval_keys(dpf.handle(), in_bits, Group::kId, b, seeds, cw_s, extra, ocws, xs, 4, nkeys, ys, stream),
          "fss_amd_ht_eval_keys");
  else  // caller-defined PRG / group: one key per thread on the same level-major arrays
    detail::LaunchThreads(detail::HtLevelMajorEvalKernel<in_bits, Group, In, Prg, uint32_t>, nkeys, stream, dpf.prg,
                          dpf.hash_key, b, seeds, cw_s, extra, ocws, xs, ys, int64_t(nkeys));
}
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, int par_depth>
  requires(!std::is_same_v<In, uint32_t>)
void HalfTreeDpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra, const int4* ocws,
                             const In* xs, int4* ys, int64_t nkeys,
                             const HalfTreeDpf<in_bits, Group, Prg, In, par_depth>& dpf, hipStream_t stream = 0) {
  using S = HalfTreeDpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_ht_eval_keys(dpf.handle(), in_bits, Group::kId, b, seeds, cw_s, extra, ocws, xs, detail::XBytes<In>(),
                               nkeys, ys, stream), "fss_amd_ht_eval_keys");
  else
    detail::LaunchThreads(detail::HtLevelMajorEvalKernel<in_bits, Group, In, Prg, In>, nkeys, stream, dpf.prg,
                          dpf.hash_key, b, seeds, cw_s, extra, ocws, xs, ys, nkeys);
}
// point_eval_gpu.cuh:448-460  ys[k] = Dpf::Eval(b, seeds[k], key k, xs[k])
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void DpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra, const int4* out_cw,
                     const uint32_t* xs, int4* ys, int nkeys, const Dpf<in_bits, Group, Prg, In, par_depth>& dpf,
                     hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t extra / xs cover in_bits <= 32; pass uint64_t* tr_bits and const In* xs");
  using S = Dpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_dpf_eval_keys_u32(dpf.handle(), in_bits, Group::kId, b, seeds, cw_s, extra, out_cw, xs, 4, nkeys, ys,
                                    stream), "fss_amd_dpf_eval_keys_u32");
  else
    detail::LaunchThreads(detail::DpfLevelMajorEvalKernel<in_bits, Group, In, Prg, uint32_t>, nkeys, stream, dpf.prg, b,
                          seeds, cw_s, extra, static_cast<const uint64_t*>(nullptr), out_cw, xs, ys, int64_t(nkeys));
}
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void DpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const int4* out_cw,
                     const In* xs, int4* ys, int64_t nkeys, const Dpf<in_bits, Group, Prg, In, par_depth>& dpf,
                     hipStream_t stream = 0) {
  using S = Dpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_dpf_eval_keys(dpf.handle(), in_bits, Group::kId, b, seeds, cw_s, tr_bits, out_cw, xs,
                                detail::XBytes<In>(), nkeys, ys, stream), "fss_amd_dpf_eval_keys");
  else
    detail::LaunchThreads(detail::DpfLevelMajorEvalKernel<in_bits, Group, In, Prg, In>, nkeys, stream, dpf.prg, b, seeds,
                          cw_s, static_cast<const uint32_t*>(nullptr), tr_bits, out_cw, xs, ys, nkeys);
}
// point_eval_gpu.cuh:479-491
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int par_depth>
void DcfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const int4* cw_v, const int4* out_cw,
                     const uint32_t* xs, int4* ys, int nkeys, const Dcf<in_bits, Group, Prg, In, pred, par_depth>& dcf,
                     hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t xs cover in_bits <= 32; pass const In* xs");
  using S = Dcf<in_bits, Group, Prg, In, pred, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_dcf_eval_keys(dcf.handle(), in_bits, Group::kId, b, seeds, cw_s, cw_v, out_cw, xs, 4, nkeys, ys, stream),
          "fss_amd_dcf_eval_keys");
  else
    detail::LaunchThreads(detail::DcfLevelMajorEvalKernel<in_bits, Group, In, Prg, uint32_t>, nkeys, stream, dcf.prg, b,
                          seeds, cw_s, cw_v, out_cw, xs, ys, int64_t(nkeys));
}
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int par_depth>
  requires(!std::is_same_v<In, uint32_t>)
void DcfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const int4* cw_v, const int4* out_cw, const In* xs,
                     int4* ys, int64_t nkeys, const Dcf<in_bits, Group, Prg, In, pred, par_depth>& dcf,
                     hipStream_t stream = 0) {
  using S = Dcf<in_bits, Group, Prg, In, pred, par_depth>;
  if constexpr (S::kEngine)
    Check(fss_amd_dcf_eval_keys(dcf.handle(), in_bits, Group::kId, b, seeds, cw_s, cw_v, out_cw, xs, detail::XBytes<In>(),
                                nkeys, ys, stream), "fss_amd_dcf_eval_keys");
  else
    detail::LaunchThreads(detail::DcfLevelMajorEvalKernel<in_bits, Group, In, Prg, In>, nkeys, stream, dcf.prg, b, seeds,
                          cw_s, cw_v, out_cw, xs, ys, nkeys);
}

// The VDPF point launchers in the reference's own form (point_eval_gpu.cuh:389-396, :514-526): template arguments
// <in_bits, Group, Prg, XorHash, Hash, In>, `uint32_t* extra` tr bits (in_bits <= 32), `const uint32_t* xs`,
// `const std::array<int4, 4>* cs`, `int nkeys`.
template <int in_bits, typename Group, typename Prg, typename XorHash, typename Hash, typename In>
void VdpfRelayoutGpu(const typename Vdpf<in_bits, Group, Prg, XorHash, Hash, In>::Cw* cws, int nkeys, int4* cw_s, uint32_t* extra,
                     hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t extra holds 32 tr bits (as in the reference); use the uint64_t* tr_bits form");
  Check(fss_amd_vdpf_relayout_u32(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, extra, stream), "fss_amd_vdpf_relayout_u32");
}
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename XorHash, typename Hash, typename In, int pd>
void VdpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra, const std::array<int4, 4>* cs, const int4* ocws,
                      const uint32_t* xs, int4* ys, int4* pi, int nkeys, const Vdpf<in_bits, Group, Prg, XorHash, Hash, In, pd>& vdpf,
                      hipStream_t stream = 0) {
  static_assert(in_bits <= 32, "uint32_t extra / xs (as in the reference); use the uint64_t* tr_bits form for wider keys");
  Check(fss_amd_vdpf_eval_keys_u32(vdpf.handle(), in_bits, Group::kId, b, seeds, cw_s, extra, cs, ocws, xs, 4, nkeys, ys, pi, stream),
        "fss_amd_vdpf_eval_keys_u32");
}
// ... and the wide forms: 64 tr bits per key (two planes above 64), xs of the scheme's own In, 64-bit key counts
template <int in_bits, typename Group, typename Prg, typename XorHash, typename Hash, typename In>
void VdpfRelayoutGpu(const typename Vdpf<in_bits, Group, Prg, XorHash, Hash, In>::Cw* cws, int64_t nkeys, int4* cw_s, uint64_t* tr_bits,
                     hipStream_t stream = 0) {
  Check(fss_amd_vdpf_relayout(detail_gpu::AnyCtx(), in_bits, cws, nkeys, cw_s, tr_bits, stream), "fss_amd_vdpf_relayout");
}
template <int unroll = 4, int in_bits, typename Group, typename Prg, typename XorHash, typename Hash, typename In, int pd>
void VdpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const int4* cs,
                      const int4* ocws, const In* xs, int4* ys, int4* pi, int64_t nkeys,
                      const Vdpf<in_bits, Group, Prg, XorHash, Hash, In, pd>& vdpf, hipStream_t stream = 0) {
  Check(fss_amd_vdpf_eval_keys(vdpf.handle(), in_bits, Group::kId, b, seeds, cw_s, tr_bits, cs, ocws, xs,
                               detail::XBytes<In>(), nkeys, ys, pi, stream), "fss_amd_vdpf_eval_keys");
}

// eval_all_gpu.cuh:451-461: ys[x] = HalfTreeDpf::Eval(b, s0, cws, ocw, x) for every x; cws / ys device pointers,
// s0 and ocw by value, workspace from the context's pool for `stream`
template <int z = -1, int b1 = 8, int bs = 256, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void HalfTreeDpfEvalAllGpu(bool b, int4 s0, const typename HalfTreeDpf<in_bits, Group, Prg, In, par_depth>::Cw* cws,
                           int4 ocw, int4* ys, const HalfTreeDpf<in_bits, Group, Prg, In, par_depth>& dpf,
                           hipStream_t stream = 0) {
  using S = HalfTreeDpf<in_bits, Group, Prg, In, par_depth>;
  fss_amd_ctx* ctx = dpf.StageCtx();
  size_t ws = 0;
  if constexpr (S::kEngine) ws = dpf.EvalAllWorkspace();
  const detail::Scratch sc(ctx, stream, ws);
  const int4* d_s0 = detail::PutInt4(ctx, sc, 0, s0, stream);
  const int4* d_ocw = detail::PutInt4(ctx, sc, 1, ocw, stream);
  if constexpr (S::kEngine) dpf.EvalAllBlock(b, d_s0, cws, d_ocw, ys, sc.Workspace(), ws, 0, int64_t(1) << in_bits, stream);
  else dpf.EvalPoints(b, d_s0, cws, d_ocw, static_cast<const In*>(nullptr), ys, int64_t(1) << in_bits, stream);  // one thread per leaf
}
// eval_all_gpu.cuh:478-488
template <int z = -1, int b1 = 8, int bs = 256, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void DpfEvalAllGpu(bool b, int4 s0, const typename Dpf<in_bits, Group, Prg, In, par_depth>::Cw* cws, int4* ys,
                   const Dpf<in_bits, Group, Prg, In, par_depth>& dpf, hipStream_t stream = 0) {
  using S = Dpf<in_bits, Group, Prg, In, par_depth>;
  fss_amd_ctx* ctx = dpf.StageCtx();
  size_t ws = 0;
  if constexpr (S::kEngine) ws = dpf.EvalAllWorkspace();
  const detail::Scratch sc(ctx, stream, ws);
  const int4* d_s0 = detail::PutInt4(ctx, sc, 0, s0, stream);
  if constexpr (S::kEngine) dpf.EvalAllBlock(b, d_s0, cws, ys, sc.Workspace(), ws, 0, int64_t(1) << in_bits, stream);
  else dpf.EvalPoints(b, d_s0, cws, static_cast<const In*>(nullptr), ys, int64_t(1) << in_bits, stream);
}
// eval_all_gpu.cuh:511-520: nkeys keys, the full domain of each; ys[k * 2^in_bits + x]
template <int z = -1, int b1 = 8, int bs = 256, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void DpfEvalAllGpuBatch(bool b, const int4* s0s, const typename Dpf<in_bits, Group, Prg, In, par_depth>::Cw* cws,
                        int nkeys, int4* ys, const Dpf<in_bits, Group, Prg, In, par_depth>& dpf, hipStream_t stream = 0) {
  using S = Dpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine) {
    fss_amd_ctx* ctx = dpf.handle();
    const size_t ws = fss_amd_eval_all_batch_workspace(FSS_AMD_DPF, in_bits, nkeys);
    const detail::Scratch sc(ctx, stream, ws);
    Check(fss_amd_dpf_eval_all_batch(ctx, in_bits, Group::kId, b, s0s, cws, nkeys, ys, sc.Workspace(), ws, stream),
          "fss_amd_dpf_eval_all_batch");
  } else {
    detail::LaunchThreads(detail::ThreadEvalAllBatchKernel<S, In, in_bits>, int64_t(nkeys) << in_bits, stream, dpf, b, s0s,
                          cws, static_cast<const int4*>(nullptr), int64_t(nkeys), ys);
  }
}
// eval_all_gpu.cuh:526-535
template <int z = -1, int b1 = 8, int bs = 256, int in_bits, typename Group, typename Prg, typename In, int par_depth>
void HalfTreeDpfEvalAllGpuBatch(bool b, const int4* s0s,
                                const typename HalfTreeDpf<in_bits, Group, Prg, In, par_depth>::Cw* cws, const int4* ocws,
                                int nkeys, int4* ys, const HalfTreeDpf<in_bits, Group, Prg, In, par_depth>& dpf,
                                hipStream_t stream = 0) {
  using S = HalfTreeDpf<in_bits, Group, Prg, In, par_depth>;
  if constexpr (S::kEngine) {
    fss_amd_ctx* ctx = dpf.handle();
    const size_t ws = fss_amd_eval_all_batch_workspace(FSS_AMD_HT, in_bits, nkeys);
    const detail::Scratch sc(ctx, stream, ws);
    Check(fss_amd_ht_eval_all_batch(ctx, in_bits, Group::kId, b, s0s, cws, ocws, nkeys, ys, sc.Workspace(), ws, stream),
          "fss_amd_ht_eval_all_batch");
  } else {
    detail::LaunchThreads(detail::ThreadEvalAllBatchKernel<S, In, in_bits>, int64_t(nkeys) << in_bits, stream, dpf, b, s0s,
                          cws, ocws, int64_t(nkeys), ys);
  }
}
// DCF full domain on the device (the reference has no such launcher; same shape as DpfEvalAllGpu)
template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred, int par_depth>
void DcfEvalAllGpu(bool b, int4 s0, const typename Dcf<in_bits, Group, Prg, In, pred, par_depth>::Cw* cws, int4* ys,
                   const Dcf<in_bits, Group, Prg, In, pred, par_depth>& dcf, hipStream_t stream = 0) {
  using S = Dcf<in_bits, Group, Prg, In, pred, par_depth>;
  fss_amd_ctx* ctx = dcf.StageCtx();
  size_t ws = 0;
  if constexpr (S::kEngine) ws = dcf.EvalAllWorkspace();
  const detail::Scratch sc(ctx, stream, ws);
  const int4* d_s0 = detail::PutInt4(ctx, sc, 0, s0, stream);
  if constexpr (S::kEngine) dcf.EvalAllBlock(b, d_s0, cws, ys, sc.Workspace(), ws, 0, int64_t(1) << in_bits, stream);
  else dcf.EvalPoints(b, d_s0, cws, static_cast<const In*>(nullptr), ys, int64_t(1) << in_bits, stream);
}

}  // namespace gpu
}  // namespace fss

#include "fss_amd_multi.hpp"  // fss::gpu::DeviceSet: one job across the GPUs of a node
