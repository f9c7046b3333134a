// fss_amd.hpp -- C++ host layer over the C ABI (fss_amd.h), spelled like the reference's
// template surface so that code written against myl7/fss's batched GPU launchers ports by
// changing includes.  Header-only; link with libfss_amd.so and the HIP runtime.
//
//   fss::prg::Aes128Mmo<mul>          <- include/fss/prg/aes128_mmo.cuh:27-95 (key material; here it
//                                        owns the engine context instead of OpenSSL EVP contexts)
//   fss::prg::ChaCha<mul, rounds>     <- include/fss/prg/chacha.cuh:24-128 (nonce in the context)
//   fss::group::Bytes / Uint<T, mod>  <- include/fss/group/bytes.cuh, uint.cuh (type tags: the group
//                                        arithmetic runs in the kernels)
//   fss::Dpf / Dcf / HalfTreeDpf      <- include/fss/dpf.cuh:61-81, dcf.cuh:74-96,
//                                        half_tree_dpf.cuh:39-57 (same template parameters, `prg`
//                                        member, 32-byte `Cw`; methods are the batched device forms)
//   fss::GrottoDcf                    <- include/fss/grotto_dcf.cuh:46-238
//   fss::Vdpf + fss::hash::Blake3     <- include/fss/vdpf.cuh:57-404, include/fss/hash/blake3.cuh
//   fss::gpu::*RelayoutGpu, *EvalPointGpu, *EvalAllGpu
//                                     <- include/fss/point_eval_gpu.cuh:324-491,
//                                        include/fss/eval_all_gpu.cuh:451-535
// Differences from the reference, all forced by the engine: every pointer is a device pointer;
// the DPF tr bits are uint64_t per key (the reference packs 32 levels into uint32_t); EvalAll
// takes a workspace; errors throw fss::Error instead of being ignored.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include "fss_amd.h"

namespace fss {

struct Error : std::runtime_error {
  int code;
  Error(const char* what, int c) : std::runtime_error(std::string(what) + ": " + fss_amd_strerror(c)), code(c) {}
};
inline void Check(int rc, const char* what) {
  if (rc != FSS_AMD_OK) throw Error(what, rc);
}

namespace group {
struct Bytes {
  static constexpr int kId = FSS_AMD_GROUP_BYTES;
};
template <typename T, unsigned __int128 mod = 0>
struct Uint {
  static_assert(std::is_unsigned_v<T> || std::is_same_v<T, unsigned __int128>);
  static_assert(sizeof(T) < 16 || mod == (static_cast<unsigned __int128>(1) << 127),
                "Uint<__uint128_t> is built for mod = 2^127 only");
  // mod != 0 below 128 bits: fss_amd_ctx_set_uint_modulus(prg.handle(), kMod) before using the scheme object
  static constexpr uint64_t kMod = sizeof(T) == 16 ? 0 : (uint64_t)mod;
  static constexpr int kId = sizeof(T) == 1   ? FSS_AMD_GROUP_U8
                             : sizeof(T) == 2 ? FSS_AMD_GROUP_U16
                             : sizeof(T) == 4 ? FSS_AMD_GROUP_U32
                             : sizeof(T) == 8 ? FSS_AMD_GROUP_U64
                                              : FSS_AMD_GROUP_U128_M127;
};
}  // namespace group

enum class DcfPred { kLt = FSS_AMD_PRED_LT, kGt = FSS_AMD_PRED_GT };  // include/fss/dcf.cuh:58-61

namespace prg {
// AES-128-MMO key material.  mul = number of PRG outputs a scheme uses (DPF/Grotto 2, DCF 4,
// Half-Tree 1); key i produces output i.  Copyable handle (shared ownership of the context).
template <int mul>
class Aes128Mmo {
  static_assert(mul == 1 || mul == 2 || mul == 4);

 public:
  // keys: mul pointers to 16-byte AES keys, as Aes128Mmo::CreateCtxs takes them (aes128_mmo.cuh:49).
  explicit Aes128Mmo(const unsigned char* const keys[mul], int device = 0, const int32_t hash_key[4] = nullptr) {
    uint8_t k[64] = {};
    for (int i = 0; i < mul; ++i) std::memcpy(k + 16 * i, keys[i], 16);
    fss_amd_ctx* c = nullptr;
    Check(fss_amd_ctx_create(&c, device, k, hash_key), "fss_amd_ctx_create");
    ctx_ = std::shared_ptr<fss_amd_ctx>(c, fss_amd_ctx_destroy);
  }
  fss_amd_ctx* handle() const { return ctx_.get(); }
  // out[i * mul + j] = AES_{key j}(seeds[i]) ^ seeds[i]   (Aes128Mmo::Gen, aes128_mmo.cuh:72-93)
  void Gen(const int4* d_seeds, int64_t n, int4* d_out, hipStream_t stream = 0) const {
    Check(fss_amd_prg_gen(handle(), mul, d_seeds, n, d_out, stream), "fss_amd_prg_gen");
  }

 private:
  std::shared_ptr<fss_amd_ctx> ctx_;
};

// ChaCha PRG (include/fss/prg/chacha.cuh:24-128): one permutation of [constant | seed | seed | 0,0,nonce]
// per Gen.  The nonce the reference keeps in __constant__ memory (chacha.cuh:93) lives in the context.
template <int mul, int rounds = 20>
class ChaCha {
  static_assert(mul == 1 || mul == 2 || mul == 4);
  static_assert(rounds > 0 && rounds % 2 == 0);

 public:
  explicit ChaCha(const int32_t nonce[2], int device = 0, const int32_t hash_key[4] = nullptr) {
    const uint8_t unused_aes_keys[64] = {};
    fss_amd_ctx* c = nullptr;
    Check(fss_amd_ctx_create(&c, device, unused_aes_keys, hash_key), "fss_amd_ctx_create");
    ctx_ = std::shared_ptr<fss_amd_ctx>(c, fss_amd_ctx_destroy);
    Check(fss_amd_ctx_set_prg(c, FSS_AMD_PRG_CHACHA, nonce, rounds), "fss_amd_ctx_set_prg");
  }
  fss_amd_ctx* handle() const { return ctx_.get(); }
  // out[i * mul + j] = output j of ChaCha::Gen(seeds[i])   (chacha.cuh:95-127)
  void Gen(const int4* d_seeds, int64_t n, int4* d_out, hipStream_t stream = 0) const {
    Check(fss_amd_prg_gen(handle(), mul, d_seeds, n, d_out, stream), "fss_amd_prg_gen");
  }

 private:
  std::shared_ptr<fss_amd_ctx> ctx_;
};
}  // namespace prg

namespace detail {
template <typename In>
constexpr int XBytes() {
  static_assert(sizeof(In) == 1 || sizeof(In) == 2 || sizeof(In) == 4 || sizeof(In) == 8 || sizeof(In) == 16,
                "In must be an unsigned integer of 1, 2, 4 or 8 bytes or __uint128_t");
  return (int)sizeof(In);
}
}  // namespace detail

// ---------------------------------------------------------------- DPF
template <int in_bits, typename Group, typename Prg, typename In = uint32_t>
class Dpf {
  static_assert(in_bits >= 1 && in_bits <= (int)sizeof(In) * 8);

 public:
  Prg prg;
  struct alignas(32) Cw {  // include/fss/dpf.cuh:76-81
    int4 s;
    bool tr;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits + 1;

  // nkeys keys at once: s0s[2k], s0s[2k+1] are the party seeds of key k; cws is key-major,
  // kSlots slots per key (Dpf::Gen, dpf.cuh:93-159, one key per GPU lane).
  void Gen(Cw* d_cws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
           hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys,
                          d_cws, stream), "fss_amd_dpf_gen");
  }
  // The same keys written level-major (what gpu::DpfRelayoutGpu would make of d_cws), no intermediate.
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* out_cw, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen_level_major(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(),
                                      d_betas, nkeys, cw_s, tr_bits, out_cw, stream), "fss_amd_dpf_gen_level_major");
  }
  // ys[i] = Eval(b, *d_s0, d_cws, xs[i])   (Dpf::Eval, dpf.cuh:170-214, for a whole batch of x)
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    Check(fss_amd_dpf_eval_points(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, d_xs, detail::XBytes<In>(), n,
                                  d_ys, stream), "fss_amd_dpf_eval_points");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_DPF, in_bits, nleaves);
  }
  // ys[j] = Eval(b, *d_s0, d_cws, first_leaf + j)   (Dpf::EvalAll, dpf.cuh:232-303)
  void EvalAll(bool b, const int4* d_s0, const Cw* d_cws, int4* d_ys, void* d_workspace, size_t ws_bytes,
               int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_eval_all(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, first_leaf, nleaves, d_ys,
                               d_workspace, ws_bytes, stream), "fss_amd_dpf_eval_all");
  }
};

// ---------------------------------------------------------------- DCF
template <int in_bits, typename Group, typename Prg, typename In = uint32_t, DcfPred pred = DcfPred::kLt>
class Dcf {
  static_assert(in_bits >= 1 && in_bits <= (int)sizeof(In) * 8);

 public:
  Prg prg;
  struct alignas(32) Cw {  // include/fss/dcf.cuh:91-96
    int4 s;
    int4 v;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits + 1;

  void Gen(Cw* d_cws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
           hipStream_t stream = 0) const {
    Check(fss_amd_dcf_gen(prg.handle(), in_bits, Group::kId, (int)pred, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                          nkeys, d_cws, stream), "fss_amd_dcf_gen");
  }
  void GenLevelMajor(int4* cw_s, int4* cw_v, int4* out_cw, const int4* d_s0s, const In* d_alphas, const int4* d_betas,
                     int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dcf_gen_level_major(prg.handle(), in_bits, Group::kId, (int)pred, d_s0s, d_alphas,
                                      detail::XBytes<In>(), d_betas, nkeys, cw_s, cw_v, out_cw, stream),
          "fss_amd_dcf_gen_level_major");
  }
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    Check(fss_amd_dcf_eval_points(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, d_xs, detail::XBytes<In>(), n,
                                  d_ys, stream), "fss_amd_dcf_eval_points");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_DCF, in_bits, nleaves);
  }
  void EvalAll(bool b, const int4* d_s0, const Cw* d_cws, int4* d_ys, void* d_workspace, size_t ws_bytes,
               int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_dcf_eval_all(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, first_leaf, nleaves, d_ys,
                               d_workspace, ws_bytes, stream), "fss_amd_dcf_eval_all");
  }
};

// ---------------------------------------------------------------- Half-Tree DPF
template <int in_bits, typename Group, typename Prg, typename In = uint32_t>
class HalfTreeDpf {
  static_assert(in_bits >= 1 && in_bits <= (int)sizeof(In) * 8);

 public:
  Prg prg;  // its context carries hash_key (HalfTreeDpf::hash_key, half_tree_dpf.cuh:45)
  struct alignas(32) Cw {  // include/fss/half_tree_dpf.cuh:53-57
    int4 s;
    bool extra;
  };
  static_assert(sizeof(Cw) == 32);
  static constexpr int kSlots = in_bits;

  void Gen(Cw* d_cws, int4* d_ocws, const int4* d_s0s, const In* d_alphas, const int4* d_betas, int64_t nkeys,
           hipStream_t stream = 0) const {
    Check(fss_amd_ht_gen(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys,
                         d_cws, d_ocws, stream), "fss_amd_ht_gen");
  }
  void GenLevelMajor(int4* cw_s, uint32_t* extra, int4* d_ocws, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_ht_gen_level_major(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas,
                                     nkeys, cw_s, extra, d_ocws, stream), "fss_amd_ht_gen_level_major");
  }
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_ocw, const In* d_xs, int4* d_ys, int64_t n,
                  hipStream_t stream = 0) const {
    Check(fss_amd_ht_eval_points(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, d_ocw, d_xs, detail::XBytes<In>(),
                                 n, d_ys, stream), "fss_amd_ht_eval_points");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_HT, in_bits, nleaves);
  }
  void EvalAll(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_ocw, int4* d_ys, void* d_workspace,
               size_t ws_bytes, int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits,
               hipStream_t stream = 0) const {
    Check(fss_amd_ht_eval_all(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, d_ocw, first_leaf, nleaves, d_ys,
                              d_workspace, ws_bytes, stream), "fss_amd_ht_eval_all");
  }
};

// ---------------------------------------------------------------- Grotto DCF
// include/fss/grotto_dcf.cuh:46-238: keys are DPF keys over Bytes with beta = 0; shares are bools (1 byte).
template <int in_bits, typename Prg, typename In = uint32_t>
class GrottoDcf {
  static_assert(in_bits >= 1 && in_bits <= 64 && in_bits <= (int)sizeof(In) * 8);

 public:
  Prg prg;
  using Cw = typename Dpf<in_bits, group::Bytes, Prg, In>::Cw;
  static constexpr int kSlots = in_bits + 1;

  // GrottoDcf::Gen (grotto_dcf.cuh:63-67), nkeys keys at once.
  void Gen(Cw* d_cws, const int4* d_s0s, const In* d_alphas, int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen(prg.handle(), in_bits, group::Bytes::kId, d_s0s, d_alphas, detail::XBytes<In>(), nullptr,
                          nkeys, d_cws, stream), "fss_amd_dpf_gen");
  }
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* out_cw, const int4* d_s0s, const In* d_alphas, int64_t nkeys,
                     hipStream_t stream = 0) const {
    Check(fss_amd_dpf_gen_level_major(prg.handle(), in_bits, group::Bytes::kId, d_s0s, d_alphas, detail::XBytes<In>(),
                                      nullptr, nkeys, cw_s, tr_bits, out_cw, stream), "fss_amd_dpf_gen_level_major");
  }
  // ParityTree::p = 2N-1 bytes (grotto_dcf.cuh:76-84); workspace >= PreprocessWorkspace() bytes.
  static size_t ParityTreeBytes() { return (size_t(2) << in_bits) - 1; }
  static size_t PreprocessWorkspace() {
    return fss_amd_eval_all_workspace(FSS_AMD_GROTTO, in_bits, int64_t(1) << in_bits) + (size_t(1) << in_bits);
  }
  // GrottoDcf::Preprocess (grotto_dcf.cuh:94-104)
  void Preprocess(bool b, uint8_t* d_pt, const int4* d_s0, const Cw* d_cws, void* d_workspace, size_t ws_bytes,
                  hipStream_t stream = 0) const {
    Check(fss_amd_grotto_preprocess(prg.handle(), in_bits, b, d_s0, d_cws, d_pt, d_workspace, ws_bytes, stream),
          "fss_amd_grotto_preprocess");
  }
  // GrottoDcf::Eval (grotto_dcf.cuh:116-135) for a batch of x against one parity tree
  void Eval(const uint8_t* d_pt, const In* d_xs, uint8_t* d_ys, int64_t n, hipStream_t stream = 0) const {
    Check(fss_amd_grotto_eval_table(prg.handle(), in_bits, d_pt, d_xs, detail::XBytes<In>(), n, d_ys, stream),
          "fss_amd_grotto_eval_table");
  }
  static size_t EvalAllWorkspace(int64_t nleaves = int64_t(1) << in_bits) {
    return fss_amd_eval_all_workspace(FSS_AMD_GROTTO, in_bits, nleaves);
  }
  // GrottoDcf::EvalAll (grotto_dcf.cuh:151-163); a sub-block needs the carry of the blocks before it
  // (fss_amd/shard.py: grotto_fix_shard).
  void EvalAll(bool b, const int4* d_s0, const Cw* d_cws, uint8_t* d_ys, void* d_workspace, size_t ws_bytes,
               int64_t first_leaf = 0, int64_t nleaves = int64_t(1) << in_bits, hipStream_t stream = 0) const {
    Check(fss_amd_grotto_eval_all(prg.handle(), in_bits, b, d_s0, d_cws, first_leaf, nleaves, d_ys, d_workspace,
                                  ws_bytes, stream), "fss_amd_grotto_eval_all");
  }
  // Path-walk evaluator on level-major keys, one point per key (domains with no table, in_bits up to 64):
  // reconstructs the predicate; the per-party share is not the table share (see fss_amd.h).
  void EvalKeysWalk(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const In* xs, uint8_t* ys,
                    int64_t nkeys, hipStream_t stream = 0) const {
    Check(fss_amd_grotto_eval_keys(prg.handle(), in_bits, b, seeds, cw_s, tr_bits, xs, detail::XBytes<In>(), nkeys, ys,
                                   stream), "fss_amd_grotto_eval_keys");
  }
};

// ---------------------------------------------------------------- verifiable DPF
namespace hash {
// fss::hash::Blake3 (include/fss/hash/blake3.cuh:13-172): the 8-word key of the keyed BLAKE3 hashes.
// A tag carrying the IV; the hashing runs in the kernels (Vdpf installs the IV in its PRG's context).
struct Blake3 {
  int32_t iv[8];
  explicit Blake3(const int4 iv2[2]) { std::memcpy(iv, iv2, sizeof(iv)); }
};
}  // namespace hash

// include/fss/vdpf.cuh:57-404 with XorHash = Hash = hash::Blake3 (one IV for both, as the reference's
// tests and benches use them).  A key = kSlots Cw slots + cs (4 int4) + ocw.
template <int in_bits, typename Group, typename Prg, typename In = uint32_t>
class Vdpf {
  static_assert(in_bits >= 1 && in_bits <= (int)sizeof(In) * 8);

 public:
  Prg prg;
  Vdpf(Prg p, const hash::Blake3& h) : prg(std::move(p)) {
    Check(fss_amd_ctx_set_hash_iv(prg.handle(), h.iv), "fss_amd_ctx_set_hash_iv");
  }
  using Cw = typename Dpf<in_bits, Group, Prg, In>::Cw;  // vdpf.cuh:77-82
  static constexpr int kSlots = in_bits;

  // Vdpf::Gen (vdpf.cuh:103-189) for nkeys keys; d_retry[k] = its return value (may be null).
  void Gen(Cw* d_cws, int4* d_cs /* 4 per key */, int4* d_ocws, const int4* d_s0s, const In* d_alphas,
           const int4* d_betas, int64_t nkeys, int32_t* d_retry = nullptr, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_gen(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(), d_betas, nkeys,
                           d_cws, d_cs, d_ocws, d_retry, stream), "fss_amd_vdpf_gen");
  }
  void GenLevelMajor(int4* cw_s, uint64_t* tr_bits, int4* d_cs, int4* d_ocws, const int4* d_s0s, const In* d_alphas,
                     const int4* d_betas, int64_t nkeys, int32_t* d_retry = nullptr, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_gen_level_major(prg.handle(), in_bits, Group::kId, d_s0s, d_alphas, detail::XBytes<In>(),
                                       d_betas, nkeys, cw_s, tr_bits, d_cs, d_ocws, d_retry, stream),
          "fss_amd_vdpf_gen_level_major");
  }
  // Vdpf::Eval (vdpf.cuh:204-257) of one key at n points: ys[i], pi_tildes[4i..4i+3]
  void EvalPoints(bool b, const int4* d_s0, const Cw* d_cws, const int4* d_cs, const int4* d_ocw, const In* d_xs,
                  int4* d_ys, int4* d_pi_tildes, int64_t n, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_eval_points(prg.handle(), in_bits, Group::kId, b, d_s0, d_cws, d_cs, d_ocw, d_xs,
                                   detail::XBytes<In>(), n, d_ys, d_pi_tildes, stream), "fss_amd_vdpf_eval_points");
  }
  // Vdpf::Prove (vdpf.cuh:268-278) over the n hashes of one key (a sequential chain)
  void Prove(const int4* d_pi_tildes, int64_t n, const int4* d_cs, int4* d_pi, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_prove(prg.handle(), d_pi_tildes, n, d_cs, d_pi, stream), "fss_amd_vdpf_prove");
  }
  // ... and for nkeys keys with one evaluation each
  void ProveKeys(const int4* d_pi_tildes, const int4* d_cs, int64_t nkeys, int4* d_pis, hipStream_t stream = 0) const {
    Check(fss_amd_vdpf_prove_keys(prg.handle(), d_pi_tildes, d_cs, nkeys, d_pis, stream), "fss_amd_vdpf_prove_keys");
  }
  // Vdpf::Verify (vdpf.cuh:287-292), host copies of the two proofs
  static bool Verify(const int4 pi0[4], const int4 pi1[4]) { return std::memcmp(pi0, pi1, 4 * sizeof(int4)) == 0; }
};

// ---------------------------------------------------------------- fss::gpu launchers
namespace gpu {

// include/fss/point_eval_gpu.cuh:346-353 (extra -> 64-bit tr_bits)
template <int in_bits, typename Group, typename Prg, typename In>
void DpfRelayoutGpu(const typename Dpf<in_bits, Group, Prg, In>::Cw* cws, int64_t nkeys, int4* cw_s, uint64_t* tr_bits,
                    int4* out_cw, const Dpf<in_bits, Group, Prg, In>& dpf, hipStream_t stream = 0) {
  Check(fss_amd_dpf_relayout(dpf.prg.handle(), in_bits, cws, nkeys, cw_s, tr_bits, out_cw, stream), "dpf_relayout");
}
// :368-375
template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred>
void DcfRelayoutGpu(const typename Dcf<in_bits, Group, Prg, In, pred>::Cw* cws, int64_t nkeys, int4* cw_s, int4* cw_v,
                    int4* out_cw, const Dcf<in_bits, Group, Prg, In, pred>& dcf, hipStream_t stream = 0) {
  Check(fss_amd_dcf_relayout(dcf.prg.handle(), in_bits, cws, nkeys, cw_s, cw_v, out_cw, stream), "dcf_relayout");
}
// :324-331
template <int in_bits, typename Group, typename Prg, typename In>
void HalfTreeDpfRelayoutGpu(const typename HalfTreeDpf<in_bits, Group, Prg, In>::Cw* cws, int64_t nkeys, int4* cw_s,
                            uint32_t* extra, const HalfTreeDpf<in_bits, Group, Prg, In>& dpf, hipStream_t stream = 0) {
  Check(fss_amd_ht_relayout(dpf.prg.handle(), in_bits, cws, nkeys, cw_s, extra, stream), "ht_relayout");
}

// :448-460  ys[k] = Dpf::Eval(b, seeds[k], key k, xs[k])
template <int in_bits, typename Group, typename Prg, typename In>
void DpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const int4* out_cw,
                     const In* xs, int4* ys, int64_t nkeys, const Dpf<in_bits, Group, Prg, In>& dpf,
                     hipStream_t stream = 0) {
  Check(fss_amd_dpf_eval_keys(dpf.prg.handle(), in_bits, Group::kId, b, seeds, cw_s, tr_bits, out_cw, xs,
                              detail::XBytes<In>(), nkeys, ys, stream), "dpf_eval_keys");
}
// :479-491
template <int in_bits, typename Group, typename Prg, typename In, DcfPred pred>
void DcfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const int4* cw_v, const int4* out_cw, const In* xs,
                     int4* ys, int64_t nkeys, const Dcf<in_bits, Group, Prg, In, pred>& dcf, hipStream_t stream = 0) {
  Check(fss_amd_dcf_eval_keys(dcf.prg.handle(), in_bits, Group::kId, b, seeds, cw_s, cw_v, out_cw, xs,
                              detail::XBytes<In>(), nkeys, ys, stream), "dcf_eval_keys");
}
// :416-429
template <int in_bits, typename Group, typename Prg, typename In>
void HalfTreeDpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint32_t* extra, const int4* ocws,
                             const In* xs, int4* ys, int64_t nkeys, const HalfTreeDpf<in_bits, Group, Prg, In>& dpf,
                             hipStream_t stream = 0) {
  Check(fss_amd_ht_eval_keys(dpf.prg.handle(), in_bits, Group::kId, b, seeds, cw_s, extra, ocws, xs,
                             detail::XBytes<In>(), nkeys, ys, stream), "ht_eval_keys");
}

// :389-396 (extra -> 64-bit tr_bits)
template <int in_bits, typename Group, typename Prg, typename In>
void VdpfRelayoutGpu(const typename Vdpf<in_bits, Group, Prg, In>::Cw* cws, int64_t nkeys, int4* cw_s, uint64_t* tr_bits,
                     const Vdpf<in_bits, Group, Prg, In>& vdpf, hipStream_t stream = 0) {
  Check(fss_amd_vdpf_relayout(vdpf.prg.handle(), in_bits, cws, nkeys, cw_s, tr_bits, stream), "vdpf_relayout");
}
// :514-526  ys[k], pi[4k..4k+3] = Vdpf::Eval(b, seeds[k], key k, cs[k], ocws[k], xs[k])
template <int in_bits, typename Group, typename Prg, typename In>
void VdpfEvalPointGpu(bool b, const int4* seeds, const int4* cw_s, const uint64_t* tr_bits, const int4* cs,
                      const int4* ocws, const In* xs, int4* ys, int4* pi, int64_t nkeys,
                      const Vdpf<in_bits, Group, Prg, In>& vdpf, hipStream_t stream = 0) {
  Check(fss_amd_vdpf_eval_keys(vdpf.prg.handle(), in_bits, Group::kId, b, seeds, cw_s, tr_bits, cs, ocws, xs,
                               detail::XBytes<In>(), nkeys, ys, pi, stream), "vdpf_eval_keys");
}

// include/fss/eval_all_gpu.cuh:478-488 (s0 and cws on the device; workspace from Dpf::EvalAllWorkspace)
template <int in_bits, typename Group, typename Prg, typename In>
void DpfEvalAllGpu(bool b, const int4* d_s0, const typename Dpf<in_bits, Group, Prg, In>::Cw* cws, int4* ys,
                   void* workspace, size_t ws_bytes, const Dpf<in_bits, Group, Prg, In>& dpf, hipStream_t stream = 0) {
  dpf.EvalAll(b, d_s0, cws, ys, workspace, ws_bytes, 0, int64_t(1) << in_bits, stream);
}
// :451-461
template <int in_bits, typename Group, typename Prg, typename In>
void HalfTreeDpfEvalAllGpu(bool b, const int4* d_s0, const typename HalfTreeDpf<in_bits, Group, Prg, In>::Cw* cws,
                           const int4* d_ocw, int4* ys, void* workspace, size_t ws_bytes,
                           const HalfTreeDpf<in_bits, Group, Prg, In>& dpf, hipStream_t stream = 0) {
  dpf.EvalAll(b, d_s0, cws, d_ocw, ys, workspace, ws_bytes, 0, int64_t(1) << in_bits, stream);
}

// :511-520  ys[k * 2^in_bits + x] = Dpf::Eval(b, s0s[k], key k, x); workspace from EvalAllBatchWorkspace
template <int in_bits, typename Group, typename Prg, typename In>
void DpfEvalAllGpuBatch(bool b, const int4* d_s0s, const typename Dpf<in_bits, Group, Prg, In>::Cw* cws, int64_t nkeys,
                        int4* ys, void* workspace, size_t ws_bytes, const Dpf<in_bits, Group, Prg, In>& dpf,
                        hipStream_t stream = 0) {
  Check(fss_amd_dpf_eval_all_batch(dpf.prg.handle(), in_bits, Group::kId, b, d_s0s, cws, nkeys, ys, workspace, ws_bytes,
                                   stream), "dpf_eval_all_batch");
}
// :526-535
template <int in_bits, typename Group, typename Prg, typename In>
void HalfTreeDpfEvalAllGpuBatch(bool b, const int4* d_s0s, const typename HalfTreeDpf<in_bits, Group, Prg, In>::Cw* cws,
                                const int4* d_ocws, int64_t nkeys, int4* ys, void* workspace, size_t ws_bytes,
                                const HalfTreeDpf<in_bits, Group, Prg, In>& dpf, hipStream_t stream = 0) {
  Check(fss_amd_ht_eval_all_batch(dpf.prg.handle(), in_bits, Group::kId, b, d_s0s, cws, d_ocws, nkeys, ys, workspace,
                                  ws_bytes, stream), "ht_eval_all_batch");
}
inline size_t EvalAllBatchWorkspace(int scheme, int in_bits, int64_t nkeys) {
  return fss_amd_eval_all_batch_workspace(scheme, in_bits, nkeys);
}

}  // namespace gpu
}  // namespace fss
