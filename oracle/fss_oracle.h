/*
 * fss_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the myl7/fss DPF / DCF / Half-Tree DPF / Grotto DCF
 * Gen, Eval and EvalAll algorithms with the AES-128-MMO PRG.  It is the checker for
 * the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the
 * product (fss_amd/, libfss_amd.so) never links, imports or calls anything here.
 *
 * PARITY UNPINNED against reference-produced outputs: this repository holds no vector the reference produced under its
 * own control (the known answers below are a transcription of SURVEY.md 8(c), not regenerable here) and cannot run the
 * reference.  What narrows the gap:
 * the reference's own tests hold no golden vectors (every seed is
 * random per run), and the reference cannot be compiled in this
 * image without writing stand-ins for libcu++ <cuda/std/array|span|tuple> (prg.cuh:10,
 * util.cuh:10-11; no package here provides them), so no oracle/_ref build exists.  The oracle is pinned against
 *   (1) the known-answer values recorded from the reference's CPU OpenSSL path in
 *       SURVEY.md section 8(c) (PRG outputs, DPF/DCF n=8 CWs, Eval outputs and
 *       EvalAll SHA-256, DPF n=32 Uint<u64>, Half-Tree n=64) -- tests/test_oracle_kat.py,
 *   (2) FIPS-197 appendix C.1 for the AES core, and OpenSSL EVP_aes_128_ecb (the
 *       reference's actual PRG dependency, libcrypto 3.0.2 in this image) block by block,
 *   (3) the reference's property tests (share reconstruction) restated in tests/.
 * For the ChaCha PRG and the VDPF / BLAKE3 rows not even transcribed values exist -- none are recorded and none can be
 * produced here: they are pinned to the
 * published algorithms only (RFC 7539 for the ChaCha permutation, the official BLAKE3 test vectors for the
 * compression function) and to the reference's source for the constructions around them.
 *
 * Every function cites the reference file:line (under /root/reference) it follows.
 */
#ifndef FSS_ORACLE_H_
#define FSS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CUDA int4: four little-endian int32 {x,y,z,w}; control bit = bit 0 of w
 * (include/fss/util.cuh:30-38). */
typedef struct {
  int32_t x, y, z, w;
} fo_i4;

/* One 32-byte correction-word slot, raw bytes as the reference lays it out:
 *   Dpf::Cw        {int4 s; bool tr;  pad}   include/fss/dpf.cuh:76-81
 *   Dcf::Cw        {int4 s; int4 v;}         include/fss/dcf.cuh:91-96
 *   HalfTreeDpf::Cw{int4 s; bool extra; pad} include/fss/half_tree_dpf.cuh:53-57
 * Padding bytes are written as zero here (the reference leaves cws[in_bits]'s
 * padding indeterminate, dpf.cuh:158; compare with the mask in tests). */
typedef struct {
  fo_i4 s;
  fo_i4 v; /* Dcf: v.  Dpf/HalfTree: v.x = tr/extra (0 or 1), rest 0. */
} fo_cw;

/* Output groups (include/fss/group/bytes.cuh, include/fss/group/uint.cuh). */
enum {
  FO_GROUP_BYTES = 0,
  FO_GROUP_U8 = 1,
  FO_GROUP_U16 = 2,
  FO_GROUP_U32 = 3,
  FO_GROUP_U64 = 4,
  FO_GROUP_U128_M127 = 5 /* Uint<__uint128_t, 1<<127> */
};

/* Modulus of Uint<T, mod> for the U8..U64 groups (uint.cuh:11-31); 0 = wrapping.  Process-wide. */
void fo_set_uint_modulus(uint64_t mod);

enum { FO_PRED_LT = 0, FO_PRED_GT = 1 }; /* include/fss/dcf.cuh:58-61 */

/* AES-128-MMO PRG, include/fss/prg/aes128_mmo.cuh:27-95. */
typedef struct {
  int mul;
  int use_evp;          /* 1: OpenSSL EVP exactly as the reference calls it; 2: AES-NI (Aes128MmoRaw) */
  uint8_t key[4][16];
  uint32_t rk[4][44];   /* own AES: little-endian round-key words */
  void *evp[4];         /* EVP_CIPHER_CTX* when use_evp */
  int kind;             /* 0: AES-128-MMO (above); 1: ChaCha (include/fss/prg/chacha.cuh:24-128) */
  int rounds;           /* ChaCha rounds (20) */
  int32_t nonce[2];     /* ChaCha nonce */
} fo_prg;

int fo_prg_init(fo_prg *p, const uint8_t *keys /* mul*16 bytes */, int mul, int use_evp);
int fo_have_aesni(void);
/* fss::prg::ChaCha<mul, rounds>(nonce), include/fss/prg/chacha.cuh:93. */
int fo_prg_init_chacha(fo_prg *p, int mul, int rounds, const int32_t nonce[2]);
/* The bare ChaCha permutation (`rounds` rounds, no feed-forward) on 16 words, for the RFC 7539 KAT. */
void fo_chacha_permute(uint32_t state[16], int rounds);
void fo_prg_free(fo_prg *p);
void fo_prg_gen(const fo_prg *p, fo_i4 seed, fo_i4 *out /* mul */);
/* Raw AES-128 single block (FIPS-197), for KATs. */
void fo_aes128_encrypt(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]);

/* ---- DPF (include/fss/dpf.cuh) ---- */
void fo_dpf_gen(const fo_prg *p, int in_bits, int group, fo_cw *cws /* in_bits+1 */, const fo_i4 s0s[2],
                uint64_t a_lo, uint64_t a_hi, fo_i4 beta);
fo_i4 fo_dpf_eval(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, uint64_t x_lo,
                  uint64_t x_hi);
void fo_dpf_eval_all(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys);
/* fo_dpf_eval_all in the reference's execution shape: depth-first, one OpenMP task per child above
 * ceil(log2(nthreads)) levels (dpf.cuh:232-303, util.cuh:40-45) */
void fo_dpf_eval_all_tasks(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys,
                           int nthreads);
/* reps calls of fo_dpf_eval_all_tasks (nthreads >= 1) or fo_dpf_eval_all (nthreads = 0) */
void fo_dpf_eval_all_repeat(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys,
                            int nthreads, int reps);

/* ---- DCF (include/fss/dcf.cuh) ---- */
void fo_dcf_gen(const fo_prg *p, int in_bits, int group, int pred, fo_cw *cws /* in_bits+1 */, const fo_i4 s0s[2],
                uint64_t a_lo, uint64_t a_hi, fo_i4 beta);
fo_i4 fo_dcf_eval(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, uint64_t x_lo,
                  uint64_t x_hi);
void fo_dcf_eval_all(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys);

/* ---- Half-Tree DPF (include/fss/half_tree_dpf.cuh) ---- */
void fo_ht_gen(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, fo_cw *cws /* in_bits */, fo_i4 *ocw,
               const fo_i4 s0s[2], uint64_t a_lo, uint64_t a_hi, fo_i4 beta);
fo_i4 fo_ht_eval(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                 fo_i4 ocw, uint64_t x_lo, uint64_t x_hi);
void fo_ht_eval_all(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                    fo_i4 ocw, fo_i4 *ys);

/* ---- Grotto DCF (include/fss/grotto_dcf.cuh) ---- */
void fo_grotto_gen(const fo_prg *p, int in_bits, fo_cw *cws /* in_bits+1 */, const fo_i4 s0s[2], uint64_t a_lo,
                   uint64_t a_hi);
/* pt: 2N-1 bytes (ParityTree::p); b is ParityTree::b. */
void fo_grotto_preprocess(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint8_t *pt);
int fo_grotto_eval(int in_bits, const uint8_t *pt, uint64_t x);
void fo_grotto_eval_all(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint8_t *ys);
/* Path-walk evaluator: XOR of the control bits of the left
 * siblings on the path of e = x+1; the only evaluator that exists for in_bits = 64.
 * Not a reference function: it reconstructs the same predicate but its per-party
 * share differs from fo_grotto_eval's by a party-independent mask. */
int fo_grotto_eval_walk(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint64_t x);

/* ---- BLAKE3 compression and VDPF (include/fss/hash/blake3.cuh, include/fss/vdpf.cuh) ----
 * Parity pinning: the compression function against the official BLAKE3 test vectors (empty input, one
 * zero byte, keyed empty input; tests/test_oracle_kat.py), the construction around it against the source. */
void fo_blake3_compress(const uint32_t h[8], const uint32_t msg[16], uint64_t counter, uint32_t block_len,
                        uint32_t flags, uint32_t out[16]);
/* SHA-256 (FIPS 180-4) and the hash-pair switch: 0 = keyed BLAKE3 (default), 1 = fss::hash::Sha256
 * (include/fss/hash/sha256.cuh:15-62; key = first 4 words of `iv`).  fo_blake3_hash / fo_blake3_xorhash and the
 * VDPF functions follow the switch. */
void fo_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void fo_set_hash_kind(int kind);
void fo_blake3_hash(const int32_t iv[8], const fo_i4 msg[4], fo_i4 out[2]);
void fo_blake3_xorhash(const int32_t iv[8], fo_i4 a, fo_i4 b, fo_i4 out[4]);
int fo_vdpf_gen(const fo_prg *p, const int32_t iv[8], int in_bits, int group, fo_cw *cws /* in_bits */, fo_i4 cs[4],
                fo_i4 *ocw, const fo_i4 s0s[2], uint64_t a_lo, uint64_t a_hi, fo_i4 beta);
fo_i4 fo_vdpf_eval(const fo_prg *p, const int32_t iv[8], int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                   const fo_i4 cs[4], fo_i4 ocw, uint64_t x_lo, uint64_t x_hi, fo_i4 pi_tilde[4]);
void fo_vdpf_prove(const int32_t iv[8], const fo_i4 *pi_tildes /* 4 per point */, size_t n, const fo_i4 cs[4],
                   fo_i4 pi[4]);

/* ---- batched drivers (loops over the functions above; nthreads>1 uses OpenMP,
 * each thread with its own EVP contexts when use_evp) ---- */
/* scheme: 0 DPF, 1 DCF, 2 Half-Tree, 3 Grotto walk (ys then holds 0/1 in .x). */
enum { FO_DPF = 0, FO_DCF = 1, FO_HT = 2, FO_GROTTO = 3 };
/* one key, many points */
/* xs_hi / alphas_hi: high 64 bits of every input (NULL = 0), for in_bits > 64. */
void fo_eval_points(const fo_prg *p, fo_i4 hash_key, int scheme, int in_bits, int group, int b, fo_i4 s0,
                    const fo_cw *cws, fo_i4 ocw, const uint64_t *xs, const uint64_t *xs_hi, size_t n, fo_i4 *ys,
                    int nthreads);
/* many keys (key-major cws with `stride` slots per key), one point each */
void fo_eval_keys(const fo_prg *p, fo_i4 hash_key, int scheme, int in_bits, int group, int b, const fo_i4 *s0s,
                  const fo_cw *cws, size_t stride, const fo_i4 *ocws, const uint64_t *xs, const uint64_t *xs_hi,
                  size_t n, fo_i4 *ys, int nthreads);
/* many keys generated at once; s0s has 2 seeds per key; cws key-major with `stride`. */
void fo_gen_keys(const fo_prg *p, fo_i4 hash_key, int scheme, int in_bits, int group, int pred, const fo_i4 *s0s,
                 const uint64_t *alphas, const uint64_t *alphas_hi, const fo_i4 *betas, size_t n, fo_cw *cws,
                 size_t stride, fo_i4 *ocws, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
