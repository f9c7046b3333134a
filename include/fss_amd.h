/*
 * fss_amd.h -- C ABI of libfss_amd.so, the MI355X (gfx950) batched DPF/DCF
 * evaluation engine.  This is the drop-in boundary for the reference's GPU
 * evaluation path: every entry point names the myl7/fss interface it replaces
 * (paths relative to the reference checkout).  Plain pointers and sizes only.
 *
 * Conventions
 *  - int4 = 16 bytes = four little-endian int32 {x,y,z,w}; the control bit is
 *    bit 0 of w (include/fss/util.cuh:30-38).
 *  - "Cw slot" = the reference's 32-byte correction word, raw bytes:
 *      Dpf::Cw         {int4 s; bool tr;  pad}  include/fss/dpf.cuh:76-81
 *      Dcf::Cw         {int4 s; int4 v}         include/fss/dcf.cuh:91-96
 *      HalfTreeDpf::Cw {int4 s; bool extra;pad} include/fss/half_tree_dpf.cuh:53-57
 *    Only byte 16 of a Dpf/HalfTree slot's second half is read.
 *  - Every `*_dev` / unnamed data pointer is a DEVICE pointer owned by the
 *    caller; nothing is allocated, freed or synchronised by the eval calls:
 *    they enqueue kernels on `stream` (a hipStream_t passed as void*, NULL =
 *    default stream) and return, like the reference launchers
 *    (include/fss/point_eval_gpu.cuh:448-460).  Two exceptions, both named where they are declared:
 *    fss_amd_eval_points_host takes its points and outputs in PINNED HOST memory, and a context that has been
 *    switched to FSS_AMD_POINTS_AUTO (opt-in; a fresh context is FSS_AMD_POINTS_PLAIN) keeps a per-stream node
 *    table for its one-key point evaluations that grows on demand, stream-ordered.
 *  - Alignment: every int4 / Cw array (seeds, cws, cw_s, cw_v, out_cw, ocws, cs, betas, ys, pi, workspaces)
 *    must be 16-byte aligned (the reference's int4 is; its Cw is alignas(32)), tr_bits 8-byte, extra 4-byte,
 *    xs / alphas sizeof(In)-aligned.  A misaligned pointer is FSS_AMD_EINVAL, never a device fault.
 *  - Return value: 0 on success, a negative FSS_AMD_E* code otherwise.  No
 *    exceptions cross the ABI.  There is no CPU fallback: without a usable HIP
 *    device fss_amd_ctx_create fails with FSS_AMD_EHIP.
 *  - x values: `x_bytes` = sizeof(In): 1, 2, 4 or 8 (uint8_t .. uint64_t, in_bits <= 8 * x_bytes)
 *    or 16 (__uint128_t as two little-endian uint64_t, in_bits <= 128) -- the
 *    reference's `In` types (dpf.cuh:61-63, fss_crypto/_jit.py:57-62).  Bits are consumed
 *    MSB-first (dpf.cuh:196).  EvalAll, the shared-prefix entry points and the
 *    Grotto calls need in_bits <= 62 / 64 / 64.
 */
#ifndef FSS_AMD_H_
#define FSS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSS_AMD_OK 0
#define FSS_AMD_EINVAL (-1)       /* bad argument (range, NULL, unsupported combination) */
#define FSS_AMD_EHIP (-2)         /* HIP runtime error; see fss_amd_last_hip_error() */
#define FSS_AMD_EUNSUPPORTED (-3) /* valid in the reference but not built here yet */

/* Output groups: fss::group::Bytes, fss::group::Uint<uint8_t|uint16_t|uint32_t|uint64_t>,
 * fss::group::Uint<__uint128_t, 1 << 127> (include/fss/group/bytes.cuh, uint.cuh). */
enum fss_amd_group {
  FSS_AMD_GROUP_BYTES = 0,
  FSS_AMD_GROUP_U8 = 1,
  FSS_AMD_GROUP_U16 = 2,
  FSS_AMD_GROUP_U32 = 3,
  FSS_AMD_GROUP_U64 = 4,
  FSS_AMD_GROUP_U128_M127 = 5
};

/* fss::DcfPred (include/fss/dcf.cuh:58-61); only matters to key generation. */
enum fss_amd_pred { FSS_AMD_PRED_LT = 0, FSS_AMD_PRED_GT = 1 };

typedef struct fss_amd_ctx fss_amd_ctx;

/* ---- context: the PRG hook -------------------------------------------------
 * Replaces fss::prg::Aes128Mmo<mul>::CreateCtxs / FreeCtxs
 * (include/fss/prg/aes128_mmo.cuh:49-70) and the HalfTreeDpf::hash_key member
 * (include/fss/half_tree_dpf.cuh:45): four AES-128 keys (key i produces PRG
 * output i; DPF/Grotto use 0-1, DCF 0-3, Half-Tree 0) and the Half-Tree hash
 * key, expanded once and kept resident on `device`.
 * Threading: a context is not synchronised.  Calls that only launch work (every eval / gen / relayout call)
 * may be issued from several host threads at once (they record nothing but the name of the last kernel, which
 * is then unspecified); fss_amd_ctx_set_prg / set_hash_iv / destroy must not run concurrently with anything else
 * on the same context.  All work is asynchronous on the given stream; nothing is allocated, freed or
 * synchronised inside a call after creation. */
int fss_amd_ctx_create(fss_amd_ctx **out, int device, const uint8_t aes_keys[64], const int32_t hash_key[4]);
void fss_amd_ctx_destroy(fss_amd_ctx *ctx);
int fss_amd_ctx_device(const fss_amd_ctx *ctx);
/* Select the context's PRG.  FSS_AMD_PRG_AES128_MMO (default): fss::prg::Aes128Mmo<mul> with the keys given
 * at creation.  FSS_AMD_PRG_CHACHA: fss::prg::ChaCha<mul, rounds>(nonce) (include/fss/prg/chacha.cuh:24-128;
 * nonce = 2 ints as the reference's constructor takes them, rounds even, 20 in the reference's default);
 * mul follows the scheme (Half-Tree 1, DPF/Grotto 2, DCF 4).  Every entry point below then uses it. */
#define FSS_AMD_PRG_AES128_MMO 0
#define FSS_AMD_PRG_CHACHA 1
int fss_amd_ctx_set_prg(fss_amd_ctx *ctx, int prg, const int32_t nonce[2], int rounds);
/* Modulus of the FSS_AMD_GROUP_U8 .. U64 groups: fss::group::Uint<T, mod> with mod != 0 (uint.cuh:11-31:
 * values reduced by From, modular add and negate).  0 (default) = the wrapping group Uint<T>.  Must fit T
 * (checked by the calls that take a group); Bytes and Uint<__uint128_t, 2^127> ignore it. */
int fss_amd_ctx_set_uint_modulus(fss_amd_ctx *ctx, uint64_t mod);
const char *fss_amd_strerror(int code);
/* Last hipError_t seen by this library on the calling thread (0 = hipSuccess). */
int fss_amd_last_hip_error(void);

/* ---- raw PRG (for KATs): out[i*mul + j] = AES_{key j}(seeds[i]) ^ seeds[i], j < mul.
 * fss::prg::Aes128Mmo<mul>::Gen, include/fss/prg/aes128_mmo.cuh:72-93. */
int fss_amd_prg_gen(fss_amd_ctx *ctx, int mul, const void *seeds, int64_t n, void *out, void *stream);

/* Scheme ids of the entry points that take one. */
enum fss_amd_scheme { FSS_AMD_DPF = 0, FSS_AMD_DCF = 1, FSS_AMD_HT = 2, FSS_AMD_GROTTO = 3 };

/* ---- point evaluation, ONE key shared by all points --------------------------
 * ys[i] = Scheme::Eval(party, *seed, cws, xs[i]).
 * Replaces a user kernel calling fss::Dpf::Eval / fss::Dcf::Eval /
 * fss::HalfTreeDpf::Eval per thread on one key (include/fss/dpf.cuh:170-214,
 * include/fss/dcf.cuh:205-276, include/fss/half_tree_dpf.cuh:187-231;
 * samples/dpf_dcf_gpu.cu:51-59) and the Python binding's single-point
 * dpf_eval / dcf_eval (fss_crypto/_csrc/dpf_binding_impl.cuh:58-101).
 * seed: device int4.  cws: device, in_bits+1 raw Cw slots (in_bits for Half-Tree).
 * ocw (Half-Tree): device int4.  ys: device int4[npoints]. */
/* Plan of these three calls (context setting; a fresh context is FSS_AMD_POINTS_PLAIN):
 *   FSS_AMD_POINTS_PLAIN  every point walks from the root: in_bits PRG levels per point; nothing is allocated, freed or
 *                         waited for -- the contract of the reference's launchers.  The default.
 *   FSS_AMD_POINTS_AUTO   opt-in (fss_amd_ctx_set_points_plan; the Python Engine and the C++ value types, which stage
 *                         and allocate for their callers anyway, opt in at creation): batches of at least 2^18 points
 *                         (in_bits <= 64) take the shared-prefix plan below with
 *                         prefix_bits = fss_amd_points_prefix_bits(): identical outputs, about in_bits - log2(npoints) + 3
 *                         AES levels per point instead of in_bits (DPF kInBits=32, 2^24 points: 3.6 -> 10.8 G evals/s).
 *                         Its node table lives in a buffer of the context, one per stream, grown on demand by the call
 *                         that needs more -- hipMallocAsync / hipFreeAsync on the CALLER'S stream (stream-ordered: no
 *                         device-wide wait, nothing that breaks a capture another thread has open): 2^prefix_bits x 16 B,
 *                         DCF x 32 B, prefix_bits <= 26, i.e. at most 1 / 2 GiB per stream.  The buffer is keyed by the
 *                         stream HANDLE: call fss_amd_ctx_scratch_release(ctx, stream) before destroying a stream that
 *                         evaluated large batches (fss_amd_ctx_destroy frees whatever is left).  Calls inside a capture
 *                         of their own stream, host-resident batches, and calls the device has no room for, walk plainly.
 * fss_amd_points_prefix_bits: the prefix_bits AUTO uses for that batch, 0 where it walks plainly. */
#define FSS_AMD_POINTS_AUTO 0
#define FSS_AMD_POINTS_PLAIN 1
int fss_amd_ctx_set_points_plan(fss_amd_ctx *ctx, int plan);
int fss_amd_points_prefix_bits(int scheme, int in_bits, int64_t npoints);
int fss_amd_dpf_eval_points(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                            const void *xs, int x_bytes, int64_t npoints, void *ys, void *stream);
int fss_amd_dcf_eval_points(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                            const void *xs, int x_bytes, int64_t npoints, void *ys, void *stream);
int fss_amd_ht_eval_points(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                           const void *ocw, const void *xs, int x_bytes, int64_t npoints, void *ys, void *stream);

/* ---- the same on a batch that lives in HOST memory (the PCIe-inclusive path) -------------
 * An entry point the reference has no counterpart for (its binding evaluates host tensors on the CPU and CUDA tensors in
 * place, fss_crypto/_csrc/dpf_binding_impl.cuh:58-101; its launchers take device pointers only): it exists because
 * SURVEY.md 8(d) asks for the PCIe-inclusive rate beside the kernel rate.  xs_host and ys_host are pinned host memory
 * (hipHostMalloc, or hipHostRegister'ed) and the walk itself reads x from and writes y to them across PCIe, 4 (8, 16)
 * bytes up and 16 down per evaluation, in ONE launch -- no staging buffers, no copies.  The batch then runs at
 * the PCIe rate (2^24 points of DPF kInBits=32: 6.1 ms = 55 GB/s both directions summed, against 4.7 ms resident and
 * 11.9 ms for copy, launch, copy on one stream; profiles/r4/e2e_trace.txt).  Outputs are those of
 * fss_amd_*_eval_points (always the plain walk: the batch is PCIe-bound either way).  scheme: FSS_AMD_DPF / _DCF / _HT;
 * seed, cws, ocw (Half-Tree) are DEVICE pointers as above; y is complete when `stream` reaches the point after this
 * call (hipStreamSynchronize / an event).  Pageable host memory is refused (FSS_AMD_EINVAL): no kernel can address it.
 * Device pointers are accepted and make this the resident plain walk. */
int fss_amd_eval_points_host(fss_amd_ctx *ctx, int scheme, int in_bits, int group, int party, const void *seed,
                             const void *cws, const void *ocw, const void *xs_host, int x_bytes, int64_t npoints,
                             void *ys_host, void *stream);

/* ---- the same, with the shared top of the tree expanded once per launch ------------------
 * All points of one key walk the same top levels.  These entry points first expand the top
 * `prefix_bits` levels of the key's tree (2^prefix_bits nodes, 2 AES each for DPF -- the
 * EvalAll expansion) into `workspace`, then every point starts from its depth-prefix_bits
 * ancestor and walks only the remaining in_bits - prefix_bits levels.  Outputs are identical
 * to fss_amd_*_eval_points; AES work drops from in_bits to about
 * (in_bits - prefix_bits) + 2^(prefix_bits+1) / npoints per point (prefix_bits ~ log2(npoints)
 * is the sweet spot).  1 <= prefix_bits <= min(in_bits, 26) (Half-Tree: <= in_bits - 1).
 * workspace: device scratch of at least fss_amd_eval_points_workspace() bytes. */
size_t fss_amd_eval_points_workspace(int scheme, int in_bits, int prefix_bits);
int fss_amd_dpf_eval_points_shared(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed,
                                   const void *cws, const void *xs, int x_bytes, int64_t npoints, void *ys,
                                   int prefix_bits, void *workspace, size_t workspace_bytes, void *stream);
int fss_amd_dcf_eval_points_shared(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed,
                                   const void *cws, const void *xs, int x_bytes, int64_t npoints, void *ys,
                                   int prefix_bits, void *workspace, size_t workspace_bytes, void *stream);
int fss_amd_ht_eval_points_shared(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed,
                                  const void *cws, const void *ocw, const void *xs, int x_bytes, int64_t npoints,
                                  void *ys, int prefix_bits, void *workspace, size_t workspace_bytes, void *stream);

/* ---- key relayout: key-major Cw slots -> level-major arrays --------------------
 * Replaces fss::gpu::{Dpf,Dcf,HalfTreeDpf}RelayoutGpu
 * (include/fss/point_eval_gpu.cuh:324-375).  cws: device, key-major, stride
 * in_bits+1 slots per key (in_bits for Half-Tree).  Outputs (device):
 *   cw_s  int4[in_bits][nkeys]          cw_s[i*nkeys + k] = cws[k][i].s
 *   cw_v  int4[in_bits][nkeys]          (DCF)
 *   tr_bits uint64[nkeys]               (DPF) bit i = cws[k][i].tr  -- 64 bits, the
 *                                        reference packs 32 (point_eval_gpu.cuh:57-62);
 *                                        in_bits > 64: uint64[2][nkeys], plane 1 = levels 64..127
 *   out_cw int4[nkeys]                  cws[k][in_bits].s (DPF) / .v (DCF)
 *   extra uint32[nkeys]                 (Half-Tree) bit 0 = cws[k][in_bits-1].extra */
int fss_amd_dpf_relayout(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *tr_bits,
                         void *out_cw, void *stream);
int fss_amd_dcf_relayout(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *cw_v,
                         void *out_cw, void *stream);
int fss_amd_ht_relayout(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *extra,
                        void *stream);

/* The inverse: level-major arrays -> key-major raw Cw slots ((in_bits+1) per key, in_bits for Half-Tree; padding
 * zero), the format the reference's Gen writes and its CPU Eval / wire transfer read (dpf.cuh:76-81,147-153). */
int fss_amd_dpf_key_major(fss_amd_ctx *ctx, int in_bits, const void *cw_s, const void *tr_bits, const void *out_cw,
                          int64_t nkeys, void *cws, void *stream);
int fss_amd_dcf_key_major(fss_amd_ctx *ctx, int in_bits, const void *cw_s, const void *cw_v, const void *out_cw,
                          int64_t nkeys, void *cws, void *stream);
int fss_amd_ht_key_major(fss_amd_ctx *ctx, int in_bits, const void *cw_s, const void *extra, int64_t nkeys, void *cws,
                         void *stream);

/* ---- point evaluation, independent keys: ys[k] = Eval(party, seeds[k], key k, xs[k]) --
 * Replaces fss::gpu::{Dpf,Dcf,HalfTreeDpf}EvalPointGpu
 * (include/fss/point_eval_gpu.cuh:416-491) on the level-major arrays above. */
int fss_amd_dpf_eval_keys(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                          const void *tr_bits, const void *out_cw, const void *xs, int x_bytes, int64_t nkeys,
                          void *ys, void *stream);
int fss_amd_dcf_eval_keys(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                          const void *cw_v, const void *out_cw, const void *xs, int x_bytes, int64_t nkeys, void *ys,
                          void *stream);
int fss_amd_ht_eval_keys(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                         const void *extra, const void *ocws, const void *xs, int x_bytes, int64_t nkeys, void *ys,
                         void *stream);

/* ---- full-domain evaluation ----------------------------------------------------
 * ys[j] = Eval(party, *seed, cws, first_leaf + j) for j < nleaves, leaves in index
 * order.  Replaces fss::gpu::{Dpf,HalfTreeDpf}EvalAllGpu (include/fss/eval_all_gpu.cuh:451-488),
 * host fss::Dpf::EvalAll / fss::Dcf::EvalAll / fss::HalfTreeDpf::EvalAll
 * (dpf.cuh:232-303, dcf.cuh:294-385, half_tree_dpf.cuh:246-316) and the binding's
 * dpf_eval_all (fss_crypto/_csrc/dpf_binding_impl.cuh:107-131).
 * [first_leaf, first_leaf + nleaves) must be an aligned power-of-two block of the
 * domain (a subtree): this is how the domain is sharded across GPUs.  `workspace`:
 * device scratch of at least fss_amd_eval_all_workspace(scheme, in_bits, nleaves)
 * bytes.  64-bit indexing throughout (the reference's int indices overflow at
 * in_bits >= 31, eval_all_gpu.cuh:257,287). */
size_t fss_amd_eval_all_workspace(int scheme, int in_bits, int64_t nleaves);
int fss_amd_dpf_eval_all(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                         int64_t first_leaf, int64_t nleaves, void *ys, void *workspace, size_t workspace_bytes,
                         void *stream);
int fss_amd_dcf_eval_all(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                         int64_t first_leaf, int64_t nleaves, void *ys, void *workspace, size_t workspace_bytes,
                         void *stream);
int fss_amd_ht_eval_all(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                        const void *ocw, int64_t first_leaf, int64_t nleaves, void *ys, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Many keys, the full domain of each: fss::gpu::DpfEvalAllGpuBatch / HalfTreeDpfEvalAllGpuBatch
 * (eval_all_gpu.cuh:511-535; DCF added).  seeds: int4[nkeys]; cws: key-major raw slots, (in_bits+1) per key
 * (in_bits for Half-Tree); ocws: int4[nkeys]; ys[k * 2^in_bits + x] = Eval(party, seeds[k], key k, x).
 * in_bits <= 40 and nkeys * 2^in_bits <= 2^40; workspace from fss_amd_eval_all_batch_workspace. */
size_t fss_amd_eval_all_batch_workspace(int scheme, int in_bits, int64_t nkeys);
int fss_amd_dpf_eval_all_batch(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cws,
                               int64_t nkeys, void *ys, void *workspace, size_t workspace_bytes, void *stream);
int fss_amd_dcf_eval_all_batch(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cws,
                               int64_t nkeys, void *ys, void *workspace, size_t workspace_bytes, void *stream);
int fss_amd_ht_eval_all_batch(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cws,
                              const void *ocws, int64_t nkeys, void *ys, void *workspace, size_t workspace_bytes,
                              void *stream);

/* ---- Grotto DCF (include/fss/grotto_dcf.cuh) -------------------------------------
 * Keys are DPF keys over Bytes with beta = 0 (grotto_dcf.cuh:63-67).
 * eval_all: ys[x] = share of 1[alpha <= x], one byte per point, = GrottoDcf::EvalAll
 * (grotto_dcf.cuh:151-163) -- bit-exact per party.
 * eval_keys: path-walk evaluator for domains with no table (in_bits up to 64):
 * XOR of the control bits of the left siblings along the path of x+1.  It
 * reconstructs the same predicate; its per-party share is NOT the table-based
 * GrottoDcf::Eval share (grotto_dcf.cuh:116-135), see DESIGN.md. */
/* GrottoDcf::Preprocess (grotto_dcf.cuh:94-104): pt = 2N-1 bytes (ParityTree::p, N = 2^in_bits): the leaf
 * control bits in pt[N-1 .. 2N-2] and the parity segment tree above them.  workspace: at least
 * fss_amd_eval_all_workspace(FSS_AMD_GROTTO, in_bits, N) + N bytes.
 * GrottoDcf::Eval (grotto_dcf.cuh:116-135): ys[i] = prefix parity of pt over [0, xs[i]]; bit-exact with the
 * reference's table walk per party. */
int fss_amd_grotto_preprocess(fss_amd_ctx *ctx, int in_bits, int party, const void *seed, const void *cws, void *pt,
                              void *workspace, size_t workspace_bytes, void *stream);
int fss_amd_grotto_eval_table(fss_amd_ctx *ctx, int in_bits, const void *pt, const void *xs, int x_bytes,
                              int64_t npoints, void *ys_bytes, void *stream);
int fss_amd_grotto_eval_all(fss_amd_ctx *ctx, int in_bits, int party, const void *seed, const void *cws,
                            int64_t first_leaf, int64_t nleaves, void *ys_bytes, void *workspace,
                            size_t workspace_bytes, void *stream);
int fss_amd_grotto_eval_keys(fss_amd_ctx *ctx, int in_bits, int party, const void *seeds, const void *cw_s,
                             const void *tr_bits, const void *xs, int x_bytes, int64_t nkeys, void *ys_bytes,
                             void *stream);

/* ---- key generation on the device (dealer side) ------------------------------------
 * One key per lane; writes key-major raw Cw slots, identical to the reference's Gen
 * for the same seeds / alpha / beta: fss::Dpf::Gen (dpf.cuh:93-159), fss::Dcf::Gen
 * (dcf.cuh:108-194), fss::HalfTreeDpf::Gen (half_tree_dpf.cuh:68-175).
 * s0s: int4[nkeys][2]; alphas: x_bytes-wide; betas: int4[nkeys], or NULL for beta = 0
 * (GrottoDcf::Gen, grotto_dcf.cuh:63-67, is fss_amd_dpf_gen over Bytes with NULL betas);
 * cws: (in_bits+1)*nkeys slots (in_bits for Half-Tree), padding bytes zero;
 * ocws (Half-Tree): int4[nkeys]. */
int fss_amd_dpf_gen(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas, int x_bytes,
                    const void *betas, int64_t nkeys, void *cws, void *stream);
int fss_amd_dcf_gen(fss_amd_ctx *ctx, int in_bits, int group, int pred, const void *s0s, const void *alphas,
                    int x_bytes, const void *betas, int64_t nkeys, void *cws, void *stream);
int fss_amd_ht_gen(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas, int x_bytes,
                   const void *betas, int64_t nkeys, void *cws, void *ocws, void *stream);

/* The same Gen writing the level-major arrays of fss_amd_*_relayout directly (what
 * fss::gpu::*RelayoutGpu, point_eval_gpu.cuh:324-375, would make of the key-major slots):
 * the dealer-to-evaluator hand-off with no key-major intermediate and no relayout pass
 * (src/bench_gpu.cu:72-83 + :616-619 fused).  Array shapes as in fss_amd_*_relayout. */
int fss_amd_dpf_gen_level_major(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas,
                                int x_bytes, const void *betas, int64_t nkeys, void *cw_s, void *tr_bits,
                                void *out_cw, void *stream);
int fss_amd_dcf_gen_level_major(fss_amd_ctx *ctx, int in_bits, int group, int pred, const void *s0s,
                                const void *alphas, int x_bytes, const void *betas, int64_t nkeys, void *cw_s,
                                void *cw_v, void *out_cw, void *stream);
int fss_amd_ht_gen_level_major(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas,
                               int x_bytes, const void *betas, int64_t nkeys, void *cw_s, void *extra, void *ocws,
                               void *stream);

/* ---- verifiable DPF (include/fss/vdpf.cuh; de Castro-Polychroniadou, EUROCRYPT'22) -------------
 * The DPF walk plus, per evaluation, the verification hash pi~ = H(Pack(x), s) [^ cs when t], with
 * H = H' = fss::hash::Blake3 keyed with the context's 8-word IV (blake3.cuh:13-172), or fss::hash::Sha256
 * after fss_amd_ctx_set_hash_sha256.  A key is in_bits raw Cw slots (vdpf.cuh:77-82, same 32-byte {s, tr}
 * slot as the DPF, no output slot), cs = int4[4] and ocw = int4.
 *   gen            Vdpf::Gen (vdpf.cuh:103-189), one key per lane; retry[k] = 1 where the reference
 *                  returns 1 (t0 == t1; resample the seeds), may be NULL.
 *   relayout       fss::gpu::VdpfRelayoutGpu (point_eval_gpu.cuh:389-396); tr_bits as for the DPF.
 *   eval_points    Vdpf::Eval (vdpf.cuh:204-257) of ONE key at npoints points: ys int4[n], pi_tildes int4[n][4].
 *   eval_keys      fss::gpu::VdpfEvalPointGpu (point_eval_gpu.cuh:514-526): nkeys keys, one point each.
 *   prove          Vdpf::Prove (vdpf.cuh:268-278) over the npoints hashes of one key: a sequential
 *                  hash chain by definition, run by a single lane.
 *   prove_keys     the same fold for nkeys keys with one evaluation each, one key per lane.
 * Vdpf::Verify (vdpf.cuh:287-292) is a 64-byte comparison of the two parties' pi and is left to the caller. */
int fss_amd_ctx_set_hash_iv(fss_amd_ctx *ctx, const int32_t iv[8]);
/* The reference's other hash pair, fss::hash::Sha256(key) for both H and H' (include/fss/hash/sha256.cuh:15-62:
 * SHA-256 of the 16-byte key followed by the message; host-only in the reference, on the device here).  This is
 * the pair the reference's CPU benchmarks combine with the AES-128-MMO PRG (README.md:354-355). */
int fss_amd_ctx_set_hash_sha256(fss_amd_ctx *ctx, const int32_t key[4]);
int fss_amd_vdpf_gen(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas, int x_bytes,
                     const void *betas, int64_t nkeys, void *cws, void *cs, void *ocws, int32_t *retry, void *stream);
int fss_amd_vdpf_gen_level_major(fss_amd_ctx *ctx, int in_bits, int group, const void *s0s, const void *alphas,
                                 int x_bytes, const void *betas, int64_t nkeys, void *cw_s, void *tr_bits, void *cs,
                                 void *ocws, int32_t *retry, void *stream);
int fss_amd_vdpf_relayout(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *tr_bits,
                          void *stream);
int fss_amd_vdpf_eval_points(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seed, const void *cws,
                             const void *cs, const void *ocw, const void *xs, int x_bytes, int64_t npoints, void *ys,
                             void *pi_tildes, void *stream);
int fss_amd_vdpf_eval_keys(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                           const void *tr_bits, const void *cs, const void *ocws, const void *xs, int x_bytes,
                           int64_t nkeys, void *ys, void *pi_tildes, void *stream);
int fss_amd_vdpf_prove(fss_amd_ctx *ctx, const void *pi_tildes, int64_t npoints, const void *cs, void *pi, void *stream);
int fss_amd_vdpf_prove_keys(fss_amd_ctx *ctx, const void *pi_tildes, const void *cs, int64_t nkeys, void *pis,
                            void *stream);

/* ---- the reference's 32-bit `extra` packing of the DPF tr bits ---------------------------
 * fss::gpu::DpfRelayoutGpu / DpfEvalPointGpu take `uint32_t* extra` (bit i = tr of level i,
 * include/fss/point_eval_gpu.cuh:57-62,346-353,448-460), which stops at in_bits = 32.  These two are
 * fss_amd_dpf_relayout / fss_amd_dpf_eval_keys on that layout (extra: uint32[nkeys], 4-byte aligned;
 * in_bits <= 32), so a caller's existing `extra` buffers keep their type and size. */
int fss_amd_dpf_relayout_u32(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *extra,
                             void *out_cw, void *stream);
int fss_amd_dpf_eval_keys_u32(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                              const void *extra, const void *out_cw, const void *xs, int x_bytes, int64_t nkeys,
                              void *ys, void *stream);
/* ... and the VDPF launchers on it: fss::gpu::VdpfRelayoutGpu(cws, int nkeys, cw_s, uint32_t* extra, stream) and
 * VdpfEvalPointGpu(b, seeds, cw_s, const uint32_t* extra, const array<int4,4>* cs, ocws, const uint32_t* xs, ys, pi,
 * int nkeys, vdpf, stream) as the reference declares them (point_eval_gpu.cuh:389-396, :514-526); in_bits <= 32. */
int fss_amd_vdpf_relayout_u32(fss_amd_ctx *ctx, int in_bits, const void *cws, int64_t nkeys, void *cw_s, void *extra,
                              void *stream);
int fss_amd_vdpf_eval_keys_u32(fss_amd_ctx *ctx, int in_bits, int group, int party, const void *seeds, const void *cw_s,
                               const void *extra, const void *cs, const void *ocws, const void *xs, int x_bytes,
                               int64_t nkeys, void *ys, void *pi_tildes, void *stream);

/* ---- one job across the GPUs of a node --------------------------------------------------------
 * The evaluation batch (points x keys) and the full domain split without any exchange (the
 * reference's launchers, include/fss/point_eval_gpu.cuh:448-460 / eval_all_gpu.cuh:478-488, as
 * src/bench_gpu.cu:604-627 calls them, are what each device runs on its share): device `rank` of `world`
 * evaluates its slice with the calls above on a context of its own (fss_amd_ctx_create(.., device, ..)) and
 * keeps the outputs.  These are the ONE definition of the split, for C, C++ (fss::gpu::DeviceSet,
 * include/fss_amd_multi.hpp) and Python (fss_amd/shard.py):
 *   batch_slice   [*lo, *hi) of an n-element batch (points of one key, or independent keys): contiguous,
 *                 sizes differ by at most one, earlier ranks take the larger shares.
 *   domain_block  rank's aligned block of a 2^in_bits domain for fss_amd_*_eval_all: world a power of two no
 *                 larger than the domain; rank owns the subtree under its log2(world)-bit prefix.
 *   grotto_carries  Grotto's EvalAll is a running XOR over the leaves, the one result that crosses block
 *                 boundaries: every rank evaluates its block with carry-in 0, shard_last[r] = the last byte
 *                 of rank r's block (the block's parity); carry_in[r] = XOR of shard_last[0 .. r-1] -- `world`
 *                 bytes through the host, no collective.
 *   grotto_apply_carry  ys_bytes[i] ^= 1 for the nleaves bytes of a block whose carry_in is 1 (device
 *                 pointer, asynchronous on `stream`); carry = 0 does nothing.
 * Pure host arithmetic apart from the last; FSS_AMD_EINVAL for rank / world out of range. */
int fss_amd_batch_slice(int64_t n, int rank, int world, int64_t *lo, int64_t *hi);
int fss_amd_domain_block(int in_bits, int rank, int world, int64_t *first_leaf, int64_t *nleaves);
int fss_amd_grotto_carries(const uint8_t *shard_last, int world, uint8_t *carry_in);
int fss_amd_grotto_apply_carry(fss_amd_ctx *ctx, void *ys_bytes, int64_t nleaves, int carry, void *stream);

/* ---- host conveniences for layers that keep the reference's call signatures ------------------
 * The reference's launchers take the party seed BY VALUE and no workspace
 * (fss::gpu::DpfEvalAllGpu(bool b, int4 s0, const Cw* cws, int4* ys, const Dpf&, cudaStream_t),
 * include/fss/eval_all_gpu.cuh:478-488); its scheme methods take host buffers (dpf.cuh:93,170,232).
 *   fss_amd_ctx_scratch  device scratch owned by the context, one grow-only buffer per stream: *out is valid
 *                        until a later call for the same (ctx, stream) asks for more, or the context is destroyed.
 *                        Growing frees the old buffer with hipFree (which waits for the device first).
 *                        This is the only call after creation that allocates.  Thread-safe.  `stream` is only the
 *                        key here (any pointer value may serve as one); for hipStreamPerThread the calling thread is
 *                        part of the key, as for the per-stream work-distribution counters of the eval calls.
 *   fss_amd_ctx_scratch_release  frees the buffer of (ctx, stream) and hands the stream's work-distribution counters on
 *                        (waits for the device), e.g. before the stream is destroyed or a thread that used
 *                        hipStreamPerThread ends; later calls for the same key start afresh.  A context serves up to
 *                        1024 live streams with the launch-wide tail queue; beyond that launches go without it.
 *   fss_amd_upload       stream-ordered copy of a small host value (<= 1 MiB; meant for seeds and single keys) to
 *                        device memory.  The bytes travel as kernel arguments, so `src_host` may be pageable and
 *                        may be reused as soon as the call returns; nothing synchronises. */
int fss_amd_ctx_scratch(fss_amd_ctx *ctx, void *stream, size_t bytes, void **out);
int fss_amd_ctx_scratch_release(fss_amd_ctx *ctx, void *stream);
int fss_amd_upload(fss_amd_ctx *ctx, void *dst, const void *src_host, size_t bytes, void *stream);

/* ---- tuning / introspection -------------------------------------------------------- */
/* Name and launch geometry of the kernel the last eval call on this ctx launched
 * (for bench.py / profiles).  Diagnostic: written without a lock, so with several threads
 * launching on one ctx it names one of their launches. */
const char *fss_amd_last_kernel(const fss_amd_ctx *ctx, int *grid, int *block);
/* The 256-entry round table the AES kernels fill their LDS copies from, read back from the device: entry x holds
 * the bytes (2 S(x), S(x), S(x), 3 S(x)) in memory order, i.e. the reference's Te0 (ComputeTe0,
 * include/fss/prg/aes128_mmo_soft.cuh:55-60) stored little-endian.  For known-answer tests.  Synchronous. */
int fss_amd_ctx_read_te0(fss_amd_ctx *ctx, uint32_t out_host[256]);

#ifdef __cplusplus
}
#endif
#endif
