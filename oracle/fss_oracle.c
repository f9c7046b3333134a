/*
 * fss_oracle.c -- TEST INFRASTRUCTURE ONLY (see fss_oracle.h for the pinning statement).
 *
 * Plain-C restatement of the reference's CPU algorithms.  All file:line citations
 * are into /root/reference.  Nothing here is used by the shipped HIP path.
 */
#include "fss_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef FO_HAVE_OPENSSL
#include <openssl/evp.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ util */
/* include/fss/util.cuh:16-38 */
static inline fo_i4 xor4(fo_i4 a, fo_i4 b) {
  fo_i4 r = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
  return r;
}
static inline fo_i4 set_lsb(fo_i4 v, int bit) {
  if (bit) v.w |= 1;
  else v.w &= ~1;
  return v;
}
static inline int get_lsb(fo_i4 v) { return (v.w & 1) != 0; }
static const fo_i4 kZero4 = {0, 0, 0, 0};

static inline u128 mk128(uint64_t lo, uint64_t hi) { return ((u128)hi << 64) | lo; }
static inline int bit_of(u128 v, int pos) { return (int)((v >> pos) & 1); }

/* ------------------------------------------------------------------ AES-128 (FIPS-197) */
static uint8_t g_sbox[256];
static uint32_t g_te[4][256]; /* little-endian T-tables: byte r of the word = row r */
static int g_aes_ready = 0;

static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

static void aes_tables_init(void) {
  if (g_aes_ready) return;
  /* S-box from the multiplicative inverse + affine map (FIPS-197 5.1.1). */
  uint8_t p = 1, q = 1;
  do {
    p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1b : 0));
    q ^= (uint8_t)(q << 1);
    q ^= (uint8_t)(q << 2);
    q ^= (uint8_t)(q << 4);
    if (q & 0x80) q ^= 0x09;
    uint8_t x = (uint8_t)(q ^ ((q << 1) | (q >> 7)) ^ ((q << 2) | (q >> 6)) ^ ((q << 3) | (q >> 5)) ^
                          ((q << 4) | (q >> 4)));
    g_sbox[p] = (uint8_t)(x ^ 0x63);
  } while (p != 1);
  g_sbox[0] = 0x63;
  for (int i = 0; i < 256; ++i) {
    uint8_t s = g_sbox[i], s2 = xtime(s), s3 = (uint8_t)(s2 ^ s);
    /* column (2,1,1,3)*s, row 0 in the low byte */
    uint32_t t0 = (uint32_t)s2 | ((uint32_t)s << 8) | ((uint32_t)s << 16) | ((uint32_t)s3 << 24);
    g_te[0][i] = t0;
    g_te[1][i] = (t0 << 8) | (t0 >> 24);
    g_te[2][i] = (t0 << 16) | (t0 >> 16);
    g_te[3][i] = (t0 << 24) | (t0 >> 8);
  }
  g_aes_ready = 1;
}

static void aes_expand(const uint8_t key[16], uint32_t rk[44]) {
  aes_tables_init();
  uint8_t w[176];
  memcpy(w, key, 16);
  uint8_t rcon = 1;
  for (int i = 16; i < 176; i += 4) {
    uint8_t t[4] = {w[i - 4], w[i - 3], w[i - 2], w[i - 1]};
    if (i % 16 == 0) {
      uint8_t t0 = t[0];
      t[0] = (uint8_t)(g_sbox[t[1]] ^ rcon);
      t[1] = g_sbox[t[2]];
      t[2] = g_sbox[t[3]];
      t[3] = g_sbox[t0];
      rcon = xtime(rcon);
    }
    for (int j = 0; j < 4; ++j) w[i + j] = (uint8_t)(w[i - 16 + j] ^ t[j]);
  }
  for (int i = 0; i < 44; ++i)
    rk[i] = (uint32_t)w[4 * i] | ((uint32_t)w[4 * i + 1] << 8) | ((uint32_t)w[4 * i + 2] << 16) |
            ((uint32_t)w[4 * i + 3] << 24);
}

/* state columns as little-endian words c[0..3] (= the int4's x,y,z,w) */
static void aes_encrypt_words(const uint32_t rk[44], const uint32_t in[4], uint32_t out[4]) {
  uint32_t s0 = in[0] ^ rk[0], s1 = in[1] ^ rk[1], s2 = in[2] ^ rk[2], s3 = in[3] ^ rk[3];
  for (int r = 1; r < 10; ++r) {
    uint32_t t0 = g_te[0][s0 & 0xff] ^ g_te[1][(s1 >> 8) & 0xff] ^ g_te[2][(s2 >> 16) & 0xff] ^ g_te[3][s3 >> 24] ^
                  rk[4 * r + 0];
    uint32_t t1 = g_te[0][s1 & 0xff] ^ g_te[1][(s2 >> 8) & 0xff] ^ g_te[2][(s3 >> 16) & 0xff] ^ g_te[3][s0 >> 24] ^
                  rk[4 * r + 1];
    uint32_t t2 = g_te[0][s2 & 0xff] ^ g_te[1][(s3 >> 8) & 0xff] ^ g_te[2][(s0 >> 16) & 0xff] ^ g_te[3][s1 >> 24] ^
                  rk[4 * r + 2];
    uint32_t t3 = g_te[0][s3 & 0xff] ^ g_te[1][(s0 >> 8) & 0xff] ^ g_te[2][(s1 >> 16) & 0xff] ^ g_te[3][s2 >> 24] ^
                  rk[4 * r + 3];
    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
  }
#define SB(x) ((uint32_t)g_sbox[(x) & 0xff])
  out[0] = (SB(s0) | (SB(s1 >> 8) << 8) | (SB(s2 >> 16) << 16) | (SB(s3 >> 24) << 24)) ^ rk[40];
  out[1] = (SB(s1) | (SB(s2 >> 8) << 8) | (SB(s3 >> 16) << 16) | (SB(s0 >> 24) << 24)) ^ rk[41];
  out[2] = (SB(s2) | (SB(s3 >> 8) << 8) | (SB(s0 >> 16) << 16) | (SB(s1 >> 24) << 24)) ^ rk[42];
  out[3] = (SB(s3) | (SB(s0 >> 8) << 8) | (SB(s1 >> 16) << 16) | (SB(s2 >> 24) << 24)) ^ rk[43];
#undef SB
}

void fo_aes128_encrypt(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]) {
  uint32_t rk[44], a[4], b[4];
  aes_expand(key, rk);
  memcpy(a, in, 16); /* little-endian host */
  aes_encrypt_words(rk, a, b);
  memcpy(out, b, 16);
}

/* ------------------------------------------------------------------ PRG */
/* include/fss/prg/aes128_mmo.cuh:49-64 (CreateCtxs), :72-93 (Gen) */
static int evp_make(fo_prg *p) {
#ifdef FO_HAVE_OPENSSL
  for (int i = 0; i < p->mul; ++i) {
    EVP_CIPHER_CTX *c = EVP_CIPHER_CTX_new();
    if (!c) return -1;
    if (EVP_EncryptInit_ex2(c, EVP_aes_128_ecb(), p->key[i], NULL, NULL) != 1) return -1;
    if (EVP_CIPHER_CTX_set_padding(c, 0) != 1) return -1;
    p->evp[i] = c;
  }
  return 0;
#else
  (void)p;
  return -1;
#endif
}

int fo_have_aesni(void);
int fo_prg_init(fo_prg *p, const uint8_t *keys, int mul, int use_evp) {
  memset(p, 0, sizeof(*p));
  if (mul < 1 || mul > 4) return -1;
  p->mul = mul;
  p->use_evp = use_evp;
  for (int i = 0; i < mul; ++i) {
    memcpy(p->key[i], keys + 16 * i, 16);
    aes_expand(p->key[i], p->rk[i]);
  }
  if (use_evp == 2) return fo_have_aesni() ? 0 : -1;
  if (use_evp) return evp_make(p);
  return 0;
}

void fo_prg_free(fo_prg *p) {
#ifdef FO_HAVE_OPENSSL
  for (int i = 0; i < 4; ++i)
    if (p->evp[i]) EVP_CIPHER_CTX_free((EVP_CIPHER_CTX *)p->evp[i]);
#endif
  memset(p->evp, 0, sizeof(p->evp));
}

/* ---- ChaCha PRG, include/fss/prg/chacha.cuh ---- */
static inline uint32_t rotl_u32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
#define FO_QR(a, b, c, d)                                   \
  do {                                                      \
    a += b; d ^= a; d = rotl_u32(d, 16);                    \
    c += d; b ^= c; b = rotl_u32(b, 12);                    \
    a += b; d ^= a; d = rotl_u32(d, 8);                     \
    c += d; b ^= c; b = rotl_u32(b, 7);                     \
  } while (0)

/* chacha.cuh:46-60 (Rounds): x[row] is an int4, so word (row, col) = state[4*row + col]. */
void fo_chacha_permute(uint32_t x[16], int rounds) {
  for (int i = 0; i < rounds; i += 2) {
    FO_QR(x[0], x[4], x[8], x[12]);
    FO_QR(x[1], x[5], x[9], x[13]);
    FO_QR(x[2], x[6], x[10], x[14]);
    FO_QR(x[3], x[7], x[11], x[15]);
    FO_QR(x[0], x[5], x[10], x[15]);
    FO_QR(x[1], x[6], x[11], x[12]);
    FO_QR(x[2], x[7], x[8], x[13]);
    FO_QR(x[3], x[4], x[9], x[14]);
  }
}

int fo_prg_init_chacha(fo_prg *p, int mul, int rounds, const int32_t nonce[2]) {
  memset(p, 0, sizeof(*p));
  if ((mul != 1 && mul != 2 && mul != 4) || rounds < 2 || (rounds & 1)) return -1;
  p->mul = mul;
  p->kind = 1;
  p->rounds = rounds;
  p->nonce[0] = nonce[0];
  p->nonce[1] = nonce[1];
  return 0;
}

/* chacha.cuh:95-127 (Gen) */
static void chacha_gen(const fo_prg *p, fo_i4 seed, fo_i4 *out) {
  static const uint32_t k16[4] = {0x61707865, 0x3120646e, 0x79622d36, 0x6b206574}; /* "expand 16-byte k" */
  static const uint32_t k32[4] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574}; /* "expand 32-byte k" */
  const uint32_t *kc = p->mul <= 2 ? k16 : k32;
  uint32_t sd[4], x[16];
  memcpy(sd, &seed, 16);
  for (int j = 0; j < 4; ++j) { x[j] = kc[j]; x[4 + j] = sd[j]; x[8 + j] = sd[j]; } /* ExpandKey: x[2] = x[1] */
  x[12] = 0; x[13] = 0; x[14] = (uint32_t)p->nonce[0]; x[15] = (uint32_t)p->nonce[1];
  fo_chacha_permute(x, p->rounds);
  uint32_t r[4][4];
  for (int j = 0; j < 4; ++j) {
    r[0][j] = x[j] ^ kc[j];
    r[1][j] = x[4 + j] ^ sd[j];
    r[2][j] = x[8 + j] ^ sd[j];
    r[3][j] = x[12 + j];
  }
  r[3][2] ^= (uint32_t)p->nonce[0];
  r[3][3] ^= (uint32_t)p->nonce[1];
  if (p->mul == 1) memcpy(&out[0], r[1], 16);
  else if (p->mul == 2) { memcpy(&out[0], r[0], 16); memcpy(&out[1], r[1], 16); }
  else for (int i = 0; i < 4; ++i) memcpy(&out[i], r[i], 16);
}

/* out[i] = AES128-ECB_{key_i}(seed) ^ seed -- aes128_mmo.cuh:79-89 */
/* AES-NI variant, the reference's fss::prg::Aes128MmoRaw<mul> (include/fss/prg/aes128_mmo_raw.cuh:62-87):
 * same function as the OpenSSL path, the "best CPU" line of the baseline (README.md:341).  Uses the round
 * keys of aes_expand (little-endian words = the key bytes in memory order). */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("aes,sse2"))) static void aesni_gen(const fo_prg *p, fo_i4 seed, fo_i4 *out) {
  __m128i s = _mm_loadu_si128((const __m128i *)&seed);
  for (int i = 0; i < p->mul; ++i) {
    const __m128i *rk = (const __m128i *)p->rk[i];
    __m128i b = _mm_xor_si128(s, _mm_loadu_si128(rk));
    for (int r = 1; r < 10; ++r) b = _mm_aesenc_si128(b, _mm_loadu_si128(rk + r));
    b = _mm_aesenclast_si128(b, _mm_loadu_si128(rk + 10));
    _mm_storeu_si128((__m128i *)&out[i], _mm_xor_si128(b, s));
  }
}
int fo_have_aesni(void) { return __builtin_cpu_supports("aes") ? 1 : 0; }
#else
static void aesni_gen(const fo_prg *p, fo_i4 seed, fo_i4 *out) { (void)p; (void)seed; (void)out; }
int fo_have_aesni(void) { return 0; }
#endif

void fo_prg_gen(const fo_prg *p, fo_i4 seed, fo_i4 *out) {
  if (p->kind == 1) { chacha_gen(p, seed, out); return; }
  if (p->use_evp == 2) { aesni_gen(p, seed, out); return; }
  for (int i = 0; i < p->mul; ++i) {
#ifdef FO_HAVE_OPENSSL
    if (p->use_evp == 1) {
      int len = 0;
      EVP_EncryptUpdate((EVP_CIPHER_CTX *)p->evp[i], (unsigned char *)&out[i], &len, (const unsigned char *)&seed, 16);
      out[i] = xor4(out[i], seed);
      continue;
    }
#endif
    uint32_t a[4], b[4];
    memcpy(a, &seed, 16);
    aes_encrypt_words(p->rk[i], a, b);
    memcpy(&out[i], b, 16);
    out[i] = xor4(out[i], seed);
  }
}

/* ------------------------------------------------------------------ groups */
/* include/fss/group/bytes.cuh:19-43, include/fss/group/uint.cuh:27-88 */
typedef struct {
  u128 v;
} gel;
#define M127 (((u128)1) << 127)

/* Uint<T, mod> with mod != 0 for T up to 64 bits (uint.cuh:11-31, :64): one process-wide modulus for the
 * U8..U64 groups, 0 = the wrapping group.  Test infrastructure: set before a run, read-only during it. */
static uint64_t g_uint_mod = 0;
void fo_set_uint_modulus(uint64_t mod) { g_uint_mod = mod; }
static inline int g_is_mod(int g) { return g_uint_mod != 0 && g >= FO_GROUP_U8 && g <= FO_GROUP_U64; }

static inline u128 gmask(int g) {
  switch (g) {
    case FO_GROUP_U8: return 0xff;
    case FO_GROUP_U16: return 0xffff;
    case FO_GROUP_U32: return 0xffffffffu;
    case FO_GROUP_U64: return ~(uint64_t)0;
    default: return ~(u128)0;
  }
}

static gel g_from(int g, fo_i4 b) {
  gel r;
  u128 x = (uint32_t)b.x, y = (uint32_t)b.y, z = (uint32_t)b.z, w = (uint32_t)b.w;
  switch (g) {
    case FO_GROUP_BYTES: r.v = x | (y << 32) | (z << 64) | (w << 96); break;
    case FO_GROUP_U8: r.v = x & 0xff; break;      /* uint.cuh:53 */
    case FO_GROUP_U16: r.v = x & 0xffff; break;   /* uint.cuh:53 */
    case FO_GROUP_U32: r.v = x; break;            /* uint.cuh:55 */
    case FO_GROUP_U64: r.v = x | (y << 32); break; /* uint.cuh:56-57 */
    default:                                      /* uint.cuh:58-62: clamp bit dropped from w */
      r.v = (x | (y << 32) | (z << 64) | ((w >> 1) << 96)) % M127;
      break;
  }
  if (g_is_mod(g)) r.v %= g_uint_mod; /* uint.cuh:64 */
  return r;
}

static gel g_add(int g, gel a, gel b) {
  gel r;
  if (g == FO_GROUP_BYTES) r.v = a.v ^ b.v;
  else if (g == FO_GROUP_U128_M127) { /* uint.cuh:33-35 */
    if (a.v >= M127 - b.v) r.v = a.v + b.v - M127;
    else r.v = a.v + b.v;
  } else if (g_is_mod(g)) { /* uint.cuh:20-24 */
    if (a.v >= (u128)g_uint_mod - b.v) r.v = a.v + b.v - g_uint_mod;
    else r.v = a.v + b.v;
  } else r.v = (a.v + b.v) & gmask(g);
  return r;
}

static gel g_neg(int g, gel a) {
  gel r;
  if (g == FO_GROUP_BYTES) r.v = a.v;
  else if (g == FO_GROUP_U128_M127) r.v = a.v == 0 ? 0 : M127 - a.v; /* uint.cuh:41-42 */
  else if (g_is_mod(g)) r.v = a.v == 0 ? 0 : (u128)g_uint_mod - a.v; /* uint.cuh:26-31 */
  else r.v = (0 - a.v) & gmask(g);
  return r;
}

static fo_i4 g_into(int g, gel a) {
  fo_i4 b = {0, 0, 0, 0};
  switch (g) {
    case FO_GROUP_BYTES:
      b.x = (int32_t)(uint32_t)a.v; b.y = (int32_t)(uint32_t)(a.v >> 32);
      b.z = (int32_t)(uint32_t)(a.v >> 64); b.w = (int32_t)(uint32_t)(a.v >> 96);
      break;
    case FO_GROUP_U8: case FO_GROUP_U16: case FO_GROUP_U32: b.x = (int32_t)(uint32_t)a.v; break; /* uint.cuh:72 */
    case FO_GROUP_U64: b.x = (int32_t)(uint32_t)a.v; b.y = (int32_t)(uint32_t)(a.v >> 32); break; /* :73-75 */
    default: /* uint.cuh:76-82 */
      b.x = (int32_t)(uint32_t)a.v; b.y = (int32_t)(uint32_t)(a.v >> 32);
      b.z = (int32_t)(uint32_t)(a.v >> 64); b.w = (int32_t)(((uint32_t)(a.v >> 96)) << 1);
      break;
  }
  return b;
}
static gel g_zero(void) { gel r; r.v = 0; return r; }

/* ------------------------------------------------------------------ DPF */
static inline fo_cw mk_cw(fo_i4 s, int flag) {
  fo_cw c;
  c.s = s;
  c.v = kZero4;
  c.v.x = flag ? 1 : 0; /* dpf.cuh:151-153: cw_slot[1] = make_int4(tr_cw,0,0,0) */
  return c;
}
static inline int cw_flag(const fo_cw *c) { return (c->v.x & 0xff) != 0; } /* bool at byte 16 */

/* dpf.cuh:93-159 */
void fo_dpf_gen(const fo_prg *p, int in_bits, int group, fo_cw *cws, const fo_i4 s0s[2], uint64_t a_lo, uint64_t a_hi,
                fo_i4 beta) {
  u128 a = mk128(a_lo, a_hi);
  fo_i4 s0 = set_lsb(s0s[0], 0), s1 = set_lsb(s0s[1], 0);
  int t0 = 0, t1 = 1;
  beta = set_lsb(beta, 0);
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 g0[2], g1[2];
    fo_prg_gen(p, s0, g0);
    fo_prg_gen(p, s1, g1);
    int t0l = get_lsb(g0[0]), t0r = get_lsb(g0[1]), t1l = get_lsb(g1[0]), t1r = get_lsb(g1[1]);
    fo_i4 s0l = set_lsb(g0[0], 0), s0r = set_lsb(g0[1], 0), s1l = set_lsb(g1[0], 0), s1r = set_lsb(g1[1], 0);
    int a_bit = bit_of(a, in_bits - 1 - i);
    fo_i4 s_cw = !a_bit ? xor4(s0r, s1r) : xor4(s0l, s1l);
    int tl_cw = t0l ^ t1l ^ a_bit ^ 1;
    int tr_cw = t0r ^ t1r ^ a_bit;
    if (!a_bit) {
      s0 = s0l; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1l; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0l ^ tl_cw) : t0l;
      t1 = t1 ? (t1l ^ tl_cw) : t1l;
    } else {
      s0 = s0r; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1r; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0r ^ tr_cw) : t0r;
      t1 = t1 ? (t1r ^ tr_cw) : t1r;
    }
    cws[i] = mk_cw(set_lsb(s_cw, tl_cw), tr_cw);
  }
  gel v = g_add(group, g_add(group, g_from(group, beta), g_neg(group, g_from(group, s0))), g_from(group, s1));
  if (t1) v = g_neg(group, v);
  cws[in_bits] = mk_cw(g_into(group, v), 0);
}

/* one level of dpf.cuh:175-205 / 266-289: node (s, t) -> both children (s, t) */
static inline void dpf_expand(const fo_prg *p, fo_i4 s, int t, const fo_cw *cw, fo_i4 *sl, int *tl, fo_i4 *sr,
                              int *tr) {
  fo_i4 s_cw = cw->s;
  int tl_cw = get_lsb(s_cw);
  s_cw = set_lsb(s_cw, 0);
  int tr_cw = cw_flag(cw);
  fo_i4 g[2];
  fo_prg_gen(p, s, g);
  *tl = get_lsb(g[0]); *sl = set_lsb(g[0], 0);
  *tr = get_lsb(g[1]); *sr = set_lsb(g[1], 0);
  if (t) {
    *sl = xor4(*sl, s_cw); *sr = xor4(*sr, s_cw);
    *tl ^= tl_cw; *tr ^= tr_cw;
  }
}

static inline fo_i4 dpf_leaf(int group, int b, fo_i4 s, int t, const fo_cw *cw_last) {
  gel y = g_from(group, s); /* dpf.cuh:207-213 */
  if (t) y = g_add(group, y, g_from(group, cw_last->s));
  if (b) y = g_neg(group, y);
  return g_into(group, y);
}

/* dpf.cuh:170-214 */
fo_i4 fo_dpf_eval(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, uint64_t x_lo,
                  uint64_t x_hi) {
  u128 x = mk128(x_lo, x_hi);
  fo_i4 s = set_lsb(s0, 0);
  int t = b;
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 sl, sr; int tl, tr;
    dpf_expand(p, s, t, &cws[i], &sl, &tl, &sr, &tr);
    if (!bit_of(x, in_bits - 1 - i)) { s = sl; t = tl; } else { s = sr; t = tr; }
  }
  return dpf_leaf(group, b, s, t, &cws[in_bits]);
}

/* dpf.cuh:232-303; the recursion is flattened to an in-place breadth-first
 * expansion (children of node j at level i land at 2j, 2j+1), which visits the
 * same nodes and writes the same leaves in the same index order. */
void fo_dpf_eval_all(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys) {
  ys[0] = set_lsb(s0, b); /* dpf.cuh:233-235: t packed into the control bit */
  for (int i = 0; i < in_bits; ++i) {
    size_t cnt = (size_t)1 << i;
    for (size_t j = cnt; j-- > 0;) {
      fo_i4 st = ys[j], sl, sr; int tl, tr;
      dpf_expand(p, set_lsb(st, 0), get_lsb(st), &cws[i], &sl, &tl, &sr, &tr);
      ys[2 * j] = set_lsb(sl, tl);
      ys[2 * j + 1] = set_lsb(sr, tr);
    }
  }
  size_t n = (size_t)1 << in_bits;
  for (size_t j = 0; j < n; ++j) ys[j] = dpf_leaf(group, b, set_lsb(ys[j], 0), get_lsb(ys[j]), &cws[in_bits]);
}

/* The same evaluation in the reference's own execution shape (dpf.cuh:232-303): a depth-first recursion over
 * [l, r) that spawns one OpenMP task per child above par_depth = ceil(log2(threads)) (util.cuh:40-45,
 * dpf.cuh:293-298) and runs sequentially below it.  Used as the CPU baseline of the full-domain workloads
 * (bench.py); the leaves are the ones fo_dpf_eval_all writes (tests/test_oracle_properties.py).  The reference
 * shares its EVP contexts between the tasks; here every thread has its own set (prgs[thread]). */
static void dpf_eval_tree(const fo_prg *prgs, int in_bits, int group, int b, fo_i4 st, const fo_cw *cws, fo_i4 *ys,
                          size_t l, size_t r, int i, int par_depth) {
  const int t = get_lsb(st);
  const fo_i4 s = set_lsb(st, 0);
  if (i == in_bits) { /* dpf.cuh:253-263 */
    ys[l] = dpf_leaf(group, b, s, t, &cws[in_bits]);
    return;
  }
#ifdef _OPENMP
  const fo_prg *p = &prgs[omp_get_thread_num()];
#else
  const fo_prg *p = &prgs[0];
#endif
  fo_i4 sl, sr; int tl, tr;
  dpf_expand(p, s, t, &cws[i], &sl, &tl, &sr, &tr);
  const fo_i4 stl = set_lsb(sl, tl), str = set_lsb(sr, tr);
  const size_t mid = (l + r) / 2;
  if (i < par_depth) {
#pragma omp task
    dpf_eval_tree(prgs, in_bits, group, b, stl, cws, ys, l, mid, i + 1, par_depth);
#pragma omp task
    dpf_eval_tree(prgs, in_bits, group, b, str, cws, ys, mid, r, i + 1, par_depth);
#pragma omp taskwait
  } else {
    dpf_eval_tree(prgs, in_bits, group, b, stl, cws, ys, l, mid, i + 1, par_depth);
    dpf_eval_tree(prgs, in_bits, group, b, str, cws, ys, mid, r, i + 1, par_depth);
  }
}

static void prg_thread_copy(const fo_prg *src, fo_prg *dst);

/* `reps` evaluations on ONE set of per-thread PRG copies: the reference builds its EVP contexts once (CreateCtxs,
 * aes128_mmo.cuh:49-64) and then calls EvalAll as often as it likes, so context setup is outside the repetition. */
static void dpf_eval_all_tasks_n(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys,
                                 int nthreads, int reps) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  int par_depth = 0; /* util.cuh:40-45 (ResolveParDepth): ceil(log2(threads)) */
  while ((1 << par_depth) < nthreads) ++par_depth;
  if (par_depth > in_bits) par_depth = in_bits;
  fo_prg *prgs = (fo_prg *)malloc((size_t)nthreads * sizeof(fo_prg));
  for (int k = 0; k < nthreads; ++k) prg_thread_copy(p, &prgs[k]);
  for (int r = 0; r < reps; ++r) {
    /* one parallel region per EvalAll, as dpf.cuh:244-246 opens one per call */
#pragma omp parallel num_threads(nthreads)
#pragma omp single
    dpf_eval_tree(prgs, in_bits, group, b, set_lsb(s0, b), cws, ys, 0, (size_t)1 << in_bits, 0, par_depth);
  }
  for (int k = 0; k < nthreads; ++k) fo_prg_free(&prgs[k]);
  free(prgs);
}

void fo_dpf_eval_all_tasks(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys,
                           int nthreads) {
  dpf_eval_all_tasks_n(p, in_bits, group, b, s0, cws, ys, nthreads, 1);
}

/* `reps` back-to-back calls (timing small domains without a foreign-function call per evaluation) */
void fo_dpf_eval_all_repeat(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys,
                            int nthreads, int reps) {
  if (nthreads >= 1) {
    dpf_eval_all_tasks_n(p, in_bits, group, b, s0, cws, ys, nthreads, reps);
    return;
  }
  for (int r = 0; r < reps; ++r) fo_dpf_eval_all(p, in_bits, group, b, s0, cws, ys);
}

/* ------------------------------------------------------------------ DCF */
/* dcf.cuh:108-194 */
void fo_dcf_gen(const fo_prg *p, int in_bits, int group, int pred, fo_cw *cws, const fo_i4 s0s[2], uint64_t a_lo,
                uint64_t a_hi, fo_i4 beta) {
  u128 a = mk128(a_lo, a_hi);
  fo_i4 s0 = set_lsb(s0s[0], 0), s1 = set_lsb(s0s[1], 0);
  int t0 = 0, t1 = 1;
  gel v = g_zero();
  beta = set_lsb(beta, 0);
  gel gb = g_from(group, beta);
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 g0[4], g1[4];
    fo_prg_gen(p, s0, g0); /* [sl, vl, sr, vr] dcf.cuh:119 */
    fo_prg_gen(p, s1, g1);
    int t0l = get_lsb(g0[0]), t0r = get_lsb(g0[2]), t1l = get_lsb(g1[0]), t1r = get_lsb(g1[2]);
    fo_i4 s0l = set_lsb(g0[0], 0), s0r = set_lsb(g0[2], 0), s1l = set_lsb(g1[0], 0), s1r = set_lsb(g1[2], 0);
    gel v0l = g_from(group, set_lsb(g0[1], 0)), v0r = g_from(group, set_lsb(g0[3], 0));
    gel v1l = g_from(group, set_lsb(g1[1], 0)), v1r = g_from(group, set_lsb(g1[3], 0));
    int a_bit = bit_of(a, in_bits - 1 - i);
    fo_i4 s_cw = !a_bit ? xor4(s0r, s1r) : xor4(s0l, s1l);

    gel v_cw = g_neg(group, v); /* dcf.cuh:146-154 */
    if (!a_bit) {
      v_cw = g_add(group, g_add(group, v_cw, v1r), g_neg(group, v0r));
      if (pred == FO_PRED_GT) v_cw = g_add(group, v_cw, gb);
    } else {
      v_cw = g_add(group, g_add(group, v_cw, v1l), g_neg(group, v0l));
      if (pred == FO_PRED_LT) v_cw = g_add(group, v_cw, gb);
    }
    if (t1) v_cw = g_neg(group, v_cw);

    if (!a_bit) v = g_add(group, g_add(group, v, g_neg(group, v1l)), v0l); /* dcf.cuh:156-159 */
    else v = g_add(group, g_add(group, v, g_neg(group, v1r)), v0r);
    if (t1) v = g_add(group, v, g_neg(group, v_cw));
    else v = g_add(group, v, v_cw);

    int tl_cw = t0l ^ t1l ^ a_bit ^ 1;
    int tr_cw = t0r ^ t1r ^ a_bit;
    if (!a_bit) {
      s0 = s0l; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1l; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0l ^ tl_cw) : t0l;
      t1 = t1 ? (t1l ^ tl_cw) : t1l;
    } else {
      s0 = s0r; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1r; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0r ^ tr_cw) : t0r;
      t1 = t1 ? (t1r ^ tr_cw) : t1r;
    }
    cws[i].s = set_lsb(s_cw, tl_cw); /* dcf.cuh:185-188 */
    cws[i].v = set_lsb(g_into(group, v_cw), tr_cw);
  }
  gel vn = g_add(group, g_add(group, g_from(group, s1), g_neg(group, g_from(group, s0))), g_neg(group, v));
  if (t1) vn = g_neg(group, vn);
  cws[in_bits].s = kZero4; /* dcf.cuh:193 */
  cws[in_bits].v = g_into(group, vn);
}

typedef struct {
  fo_i4 sl, sr;
  int tl, tr;
  gel vl, vr, v_cw;
} dcf_step;

/* dcf.cuh:211-242: one PRG call and the correction of s/t (v handled by callers) */
static inline void dcf_expand(const fo_prg *p, int group, fo_i4 s, int t, const fo_cw *cw, dcf_step *o) {
  fo_i4 s_cw = cw->s;
  int tl_cw = get_lsb(s_cw);
  s_cw = set_lsb(s_cw, 0);
  fo_i4 v_cw_buf = cw->v;
  int tr_cw = get_lsb(v_cw_buf);
  v_cw_buf = set_lsb(v_cw_buf, 0);
  o->v_cw = g_from(group, v_cw_buf);
  fo_i4 g[4];
  fo_prg_gen(p, s, g);
  o->tl = get_lsb(g[0]); o->sl = set_lsb(g[0], 0); o->vl = g_from(group, set_lsb(g[1], 0));
  o->tr = get_lsb(g[2]); o->sr = set_lsb(g[2], 0); o->vr = g_from(group, set_lsb(g[3], 0));
  if (t) {
    o->sl = xor4(o->sl, s_cw); o->sr = xor4(o->sr, s_cw);
    o->tl ^= tl_cw; o->tr ^= tr_cw;
  }
}

/* dcf.cuh:205-276 */
fo_i4 fo_dcf_eval(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, uint64_t x_lo,
                  uint64_t x_hi) {
  u128 x = mk128(x_lo, x_hi);
  fo_i4 s = set_lsb(s0, 0);
  gel v = g_zero();
  int t = b;
  for (int i = 0; i < in_bits; ++i) {
    dcf_step o;
    dcf_expand(p, group, s, t, &cws[i], &o);
    int x_bit = bit_of(x, in_bits - 1 - i);
    if (b) { /* dcf.cuh:244-252 */
      v = g_add(group, v, g_neg(group, x_bit ? o.vr : o.vl));
      if (t) v = g_add(group, v, g_neg(group, o.v_cw));
    } else {
      v = g_add(group, v, x_bit ? o.vr : o.vl);
      if (t) v = g_add(group, v, o.v_cw);
    }
    if (!x_bit) { s = o.sl; t = o.tl; } else { s = o.sr; t = o.tr; }
  }
  gel vn = g_from(group, cws[in_bits].v); /* dcf.cuh:263-273 */
  if (b) {
    v = g_add(group, v, g_neg(group, g_from(group, s)));
    if (t) v = g_add(group, v, g_neg(group, vn));
  } else {
    v = g_add(group, v, g_from(group, s));
    if (t) v = g_add(group, v, vn);
  }
  return g_into(group, v);
}

/* dcf.cuh:294-385, flattened breadth-first like fo_dpf_eval_all; the running
 * group value of every node is kept in a side array. */
void fo_dcf_eval_all(const fo_prg *p, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws, fo_i4 *ys) {
  size_t n = (size_t)1 << in_bits;
  gel *vs = (gel *)malloc(n * sizeof(gel));
  ys[0] = set_lsb(s0, b);
  vs[0] = g_zero();
  for (int i = 0; i < in_bits; ++i) {
    size_t cnt = (size_t)1 << i;
    for (size_t j = cnt; j-- > 0;) {
      fo_i4 st = ys[j];
      int t = get_lsb(st);
      gel v = vs[j];
      dcf_step o;
      dcf_expand(p, group, set_lsb(st, 0), t, &cws[i], &o);
      gel vl = o.vl, vr = o.vr;
      if (t) { vl = g_add(group, vl, o.v_cw); vr = g_add(group, vr, o.v_cw); } /* dcf.cuh:348-355 */
      if (b) { vl = g_neg(group, vl); vr = g_neg(group, vr); }
      vl = g_add(group, vl, v);
      vr = g_add(group, vr, v);
      ys[2 * j] = set_lsb(o.sl, o.tl); vs[2 * j] = vl;
      ys[2 * j + 1] = set_lsb(o.sr, o.tr); vs[2 * j + 1] = vr;
    }
  }
  gel vn = g_from(group, cws[in_bits].v);
  for (size_t j = 0; j < n; ++j) { /* dcf.cuh:318-329 */
    gel term = g_from(group, set_lsb(ys[j], 0));
    if (get_lsb(ys[j])) term = g_add(group, term, vn);
    if (b) term = g_neg(group, term);
    ys[j] = g_into(group, g_add(group, vs[j], term));
  }
  free(vs);
}

/* ------------------------------------------------------------------ Half-Tree DPF */
static inline fo_i4 ht_hash(const fo_prg *p, fo_i4 hash_key, fo_i4 node) {
  fo_i4 out[4]; /* only output 0 is used: prg.Gen(hash_key ^ node)[0] */
  fo_prg_gen(p, xor4(hash_key, node), out);
  return out[0];
}

/* half_tree_dpf.cuh:68-175 */
void fo_ht_gen(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, fo_cw *cws, fo_i4 *ocw, const fo_i4 s0s[2],
               uint64_t a_lo, uint64_t a_hi, fo_i4 beta) {
  u128 a = mk128(a_lo, a_hi);
  beta = set_lsb(beta, 0);
  fo_i4 node0 = set_lsb(s0s[0], 0), node1 = set_lsb(s0s[1], 1);
  fo_i4 delta = xor4(node0, node1);
  for (int i = 0; i < in_bits - 1; ++i) {
    fo_i4 h0 = ht_hash(p, hash_key, node0), h1 = ht_hash(p, hash_key, node1);
    int a_bit = bit_of(a, in_bits - 1 - i);
    fo_i4 cw = xor4(h0, h1);
    if (!a_bit) cw = xor4(cw, delta);
    cws[i] = mk_cw(cw, 0);
    int t0 = get_lsb(node0), t1 = get_lsb(node1);
    fo_i4 n0 = xor4(xor4(h0, a_bit ? node0 : kZero4), t0 ? cw : kZero4);
    fo_i4 n1 = xor4(xor4(h1, a_bit ? node1 : kZero4), t1 ? cw : kZero4);
    node0 = n0; node1 = n1;
    delta = xor4(node0, node1);
  }
  {
    int a_n = bit_of(a, 0);
    int t0 = get_lsb(node0), t1 = get_lsb(node1);
    fo_i4 h0_0 = ht_hash(p, hash_key, set_lsb(node0, 0)), h0_1 = ht_hash(p, hash_key, set_lsb(node0, 1));
    fo_i4 h1_0 = ht_hash(p, hash_key, set_lsb(node1, 0)), h1_1 = ht_hash(p, hash_key, set_lsb(node1, 1));
    fo_i4 high0_0 = set_lsb(h0_0, 0), high0_1 = set_lsb(h0_1, 0), high1_0 = set_lsb(h1_0, 0),
          high1_1 = set_lsb(h1_1, 0);
    int low0_0 = get_lsb(h0_0), low0_1 = get_lsb(h0_1), low1_0 = get_lsb(h1_0), low1_1 = get_lsb(h1_1);
    fo_i4 HCW = a_n ? xor4(high0_0, high1_0) : xor4(high0_1, high1_1); /* :123-125 */
    int LCW_0 = low0_0 ^ low1_0 ^ !a_n;                                  /* :132-133 */
    int LCW_1 = low0_1 ^ low1_1 ^ a_n;
    cws[in_bits - 1] = mk_cw(set_lsb(HCW, LCW_0), LCW_1); /* :139-141 */
    fo_i4 leaf0, leaf1;
    if (a_n) { leaf0 = set_lsb(high0_1, low0_1); leaf1 = set_lsb(high1_1, low1_1); }
    else { leaf0 = set_lsb(high0_0, low0_0); leaf1 = set_lsb(high1_0, low1_0); }
    int lcw_an = a_n ? LCW_1 : LCW_0;
    fo_i4 leaf_cw = set_lsb(HCW, lcw_an);
    if (t0) leaf0 = xor4(leaf0, leaf_cw);
    if (t1) leaf1 = xor4(leaf1, leaf_cw);
    gel v = g_add(group, g_add(group, g_from(group, beta), g_neg(group, g_from(group, set_lsb(leaf0, 0)))),
                  g_from(group, set_lsb(leaf1, 0)));
    if (get_lsb(leaf1)) v = g_neg(group, v);
    *ocw = g_into(group, v);
  }
}

/* half_tree_dpf.cuh:205-229 / 325-354: last level + output conversion for sigma */
static inline fo_i4 ht_convert(const fo_prg *p, fo_i4 hash_key, int group, int b, fo_i4 parent, int sigma,
                               const fo_cw *cw_last, fo_i4 ocw) {
  int t = get_lsb(parent);
  fo_i4 h = ht_hash(p, hash_key, set_lsb(parent, sigma));
  fo_i4 hcw = set_lsb(cw_last->s, 0);
  int lcw = sigma ? cw_flag(cw_last) : get_lsb(cw_last->s);
  fo_i4 high = set_lsb(h, 0);
  int low = get_lsb(h);
  if (t) { high = xor4(high, hcw); low ^= lcw; }
  gel y = g_from(group, high);
  if (low) y = g_add(group, y, g_from(group, ocw));
  if (b) y = g_neg(group, y);
  return g_into(group, y);
}

/* half_tree_dpf.cuh:187-231 */
fo_i4 fo_ht_eval(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                 fo_i4 ocw, uint64_t x_lo, uint64_t x_hi) {
  u128 x = mk128(x_lo, x_hi);
  fo_i4 node = set_lsb(s0, b);
  for (int i = 0; i < in_bits - 1; ++i) {
    int x_bit = bit_of(x, in_bits - 1 - i);
    int t = get_lsb(node);
    fo_i4 h = ht_hash(p, hash_key, node);
    node = xor4(xor4(h, x_bit ? node : kZero4), t ? cws[i].s : kZero4);
  }
  return ht_convert(p, hash_key, group, b, node, bit_of(x, 0), &cws[in_bits - 1], ocw);
}

/* half_tree_dpf.cuh:246-316 (EvalTree flattened breadth-first, then the backward
 * in-place last level :277-280) */
void fo_ht_eval_all(const fo_prg *p, fo_i4 hash_key, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                    fo_i4 ocw, fo_i4 *ys) {
  ys[0] = set_lsb(s0, b);
  for (int i = 0; i < in_bits - 1; ++i) {
    size_t cnt = (size_t)1 << i;
    for (size_t j = cnt; j-- > 0;) {
      fo_i4 node = ys[j];
      fo_i4 h = ht_hash(p, hash_key, node);
      fo_i4 left = xor4(h, get_lsb(node) ? cws[i].s : kZero4); /* :299 */
      fo_i4 right = xor4(left, node);                          /* :301 */
      ys[2 * j] = left;
      ys[2 * j + 1] = right;
    }
  }
  size_t half = (size_t)1 << (in_bits - 1);
  for (size_t j = half; j-- > 0;) {
    fo_i4 parent = ys[j];
    ys[2 * j] = ht_convert(p, hash_key, group, b, parent, 0, &cws[in_bits - 1], ocw);
    ys[2 * j + 1] = ht_convert(p, hash_key, group, b, parent, 1, &cws[in_bits - 1], ocw);
  }
}

/* ------------------------------------------------------------------ Grotto DCF */
/* grotto_dcf.cuh:63-67 */
void fo_grotto_gen(const fo_prg *p, int in_bits, fo_cw *cws, const fo_i4 s0s[2], uint64_t a_lo, uint64_t a_hi) {
  fo_dpf_gen(p, in_bits, FO_GROUP_BYTES, cws, s0s, a_lo, a_hi, kZero4);
}

/* grotto_dcf.cuh:175-238 (ExpandTree): leaf control bits, in index order */
static void grotto_expand(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint8_t *t) {
  size_t n = (size_t)1 << in_bits;
  fo_i4 *nodes = (fo_i4 *)malloc(n * sizeof(fo_i4));
  nodes[0] = set_lsb(s0, b);
  for (int i = 0; i < in_bits; ++i) {
    size_t cnt = (size_t)1 << i;
    for (size_t j = cnt; j-- > 0;) {
      fo_i4 st = nodes[j], sl, sr; int tl, tr;
      dpf_expand(p, set_lsb(st, 0), get_lsb(st), &cws[i], &sl, &tl, &sr, &tr);
      nodes[2 * j] = set_lsb(sl, tl);
      nodes[2 * j + 1] = set_lsb(sr, tr);
    }
  }
  for (size_t j = 0; j < n; ++j) t[j] = (uint8_t)get_lsb(nodes[j]);
  free(nodes);
}

/* grotto_dcf.cuh:94-104 */
void fo_grotto_preprocess(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint8_t *pt) {
  size_t N = (size_t)1 << in_bits;
  grotto_expand(p, in_bits, b, s0, cws, pt + (N - 1));
  for (size_t j = N - 2; j < N - 1; --j) pt[j] = pt[2 * j + 1] ^ pt[2 * j + 2];
}

/* grotto_dcf.cuh:116-135 */
int fo_grotto_eval(int in_bits, const uint8_t *pt, uint64_t x) {
  u128 N = (u128)1 << in_bits;
  u128 e = (u128)x + 1;
  if (e == N) return pt[0]; /* :121 (e == 0 is the same case when in_bits is the width of In) */
  int pi = 0;
  size_t cur = 0;
  for (int i = 0; i < in_bits; ++i) {
    if (bit_of(e, in_bits - 1 - i)) { pi ^= pt[2 * cur + 1]; cur = 2 * cur + 2; }
    else cur = 2 * cur + 1;
  }
  return pi;
}

/* grotto_dcf.cuh:151-163 */
void fo_grotto_eval_all(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint8_t *ys) {
  size_t N = (size_t)1 << in_bits;
  grotto_expand(p, in_bits, b, s0, cws, ys);
  for (size_t x = 1; x < N; ++x) ys[x] ^= ys[x - 1];
}

/* Path walk (not in the reference). */
int fo_grotto_eval_walk(const fo_prg *p, int in_bits, int b, fo_i4 s0, const fo_cw *cws, uint64_t x) {
  u128 N = (u128)1 << in_bits; /* in_bits <= 64 */
  u128 e = (u128)x + 1;
  if (e == N) return b; /* whole domain: the root's control bit */
  fo_i4 s = set_lsb(s0, 0);
  int t = b, pi = 0;
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 sl, sr; int tl, tr;
    dpf_expand(p, s, t, &cws[i], &sl, &tl, &sr, &tr);
    if (bit_of(e, in_bits - 1 - i)) { pi ^= tl; s = sr; t = tr; }
    else { s = sl; t = tl; }
  }
  return pi;
}

/* ------------------------------------------------------------------ BLAKE3 + VDPF
 * fss::hash::Blake3 (include/fss/hash/blake3.cuh:13-172): the BLAKE3 compression function (7 rounds)
 * in keyed-hash mode with the 8-word key `iv`, one block, flags CHUNK_START|CHUNK_END|ROOT|KEYED_HASH. */
static const uint32_t kBlake3Iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                      0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static inline uint32_t rotr_u32(uint32_t v, int n) { return (v >> n) | (v << (32 - n)); }
#define B3_G(a, b, c, d, x, y)                                           \
  do {                                                                   \
    a += b + (x); d = rotr_u32(d ^ a, 16); c += d; b = rotr_u32(b ^ c, 12); \
    a += b + (y); d = rotr_u32(d ^ a, 8);  c += d; b = rotr_u32(b ^ c, 7);  \
  } while (0)

/* blake3.cuh:82-110 (Compress); out = the 16-word extended output. */
void fo_blake3_compress(const uint32_t h[8], const uint32_t msg[16], uint64_t counter, uint32_t block_len,
                        uint32_t flags, uint32_t out[16]) {
  static const int perm[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8}; /* blake3.cuh:44 */
  uint32_t v[16], m[16], t[16];
  for (int i = 0; i < 8; ++i) v[i] = h[i];
  for (int i = 0; i < 4; ++i) v[8 + i] = kBlake3Iv[i];
  v[12] = (uint32_t)counter; v[13] = (uint32_t)(counter >> 32); v[14] = block_len; v[15] = flags;
  for (int i = 0; i < 16; ++i) m[i] = msg[i];
  for (int r = 0; r < 7; ++r) {
    B3_G(v[0], v[4], v[8], v[12], m[0], m[1]);
    B3_G(v[1], v[5], v[9], v[13], m[2], m[3]);
    B3_G(v[2], v[6], v[10], v[14], m[4], m[5]);
    B3_G(v[3], v[7], v[11], v[15], m[6], m[7]);
    B3_G(v[0], v[5], v[10], v[15], m[8], m[9]);
    B3_G(v[1], v[6], v[11], v[12], m[10], m[11]);
    B3_G(v[2], v[7], v[8], v[13], m[12], m[13]);
    B3_G(v[3], v[4], v[9], v[14], m[14], m[15]);
    if (r < 6) {
      for (int i = 0; i < 16; ++i) t[i] = m[perm[i]];
      for (int i = 0; i < 16; ++i) m[i] = t[i];
    }
  }
  for (int i = 0; i < 8; ++i) { out[i] = v[i] ^ v[8 + i]; out[8 + i] = v[8 + i] ^ h[i]; }
}

enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_ROOT = 8, B3_KEYED = 16 };

static inline void i4_words(fo_i4 v, uint32_t *w) {
  w[0] = (uint32_t)v.x; w[1] = (uint32_t)v.y; w[2] = (uint32_t)v.z; w[3] = (uint32_t)v.w;
}
static inline fo_i4 words_i4(const uint32_t *w) {
  fo_i4 v = {(int32_t)w[0], (int32_t)w[1], (int32_t)w[2], (int32_t)w[3]};
  return v;
}

/* ---- SHA-256 (FIPS 180-4), the reference's other hash pair: fss::hash::Sha256 (include/fss/hash/sha256.cuh:15-62)
 * = EVP_Digest(key | message) with a 16-byte key.  Own implementation, checked against OpenSSL in the tests. */
static const uint32_t kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
  uint32_t w[64], a[8];
  for (int i = 0; i < 16; ++i)
    w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    uint32_t s0 = rotr_u32(w[i - 15], 7) ^ rotr_u32(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr_u32(w[i - 2], 17) ^ rotr_u32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  for (int i = 0; i < 8; ++i) a[i] = h[i];
  for (int i = 0; i < 64; ++i) {
    uint32_t S1 = rotr_u32(a[4], 6) ^ rotr_u32(a[4], 11) ^ rotr_u32(a[4], 25);
    uint32_t ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
    uint32_t t1 = a[7] + S1 + ch + kSha256K[i] + w[i];
    uint32_t S0 = rotr_u32(a[0], 2) ^ rotr_u32(a[0], 13) ^ rotr_u32(a[0], 22);
    uint32_t mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
    uint32_t t2 = S0 + mj;
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (int i = 0; i < 8; ++i) h[i] += a[i];
}
void fo_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint8_t blk[64];
  size_t off = 0;
  for (; off + 64 <= len; off += 64) sha256_block(h, msg + off);
  size_t rem = len - off;
  memset(blk, 0, 64);
  memcpy(blk, msg + off, rem);
  blk[rem] = 0x80;
  if (rem >= 56) { sha256_block(h, blk); memset(blk, 0, 64); }
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; ++i) blk[63 - i] = (uint8_t)(bits >> (8 * i));
  sha256_block(h, blk);
  for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

/* Which hash pair the VDPF functions below use: 0 = keyed BLAKE3 (8-word iv), 1 = SHA-256 (the first 4 words of
 * `iv` are the 16-byte key).  Process-wide test-infrastructure switch. */
static int g_hash_kind = 0;
void fo_set_hash_kind(int kind) { g_hash_kind = kind; }

static void sha_hash(const int32_t key[4], const fo_i4 msg[4], fo_i4 out[2]) { /* sha256.cuh:24-39: 80 bytes */
  uint8_t buf[80];
  memcpy(buf, key, 16); memcpy(buf + 16, msg, 64);
  fo_sha256(buf, 80, (uint8_t *)out);
}
static void sha_xorhash(const int32_t key[4], fo_i4 a, fo_i4 b, fo_i4 out[4]) { /* sha256.cuh:41-60: 2 x 48 bytes */
  uint8_t buf[48];
  fo_i4 a0 = set_lsb(a, 0), a1 = set_lsb(a, 1);
  memcpy(buf, key, 16); memcpy(buf + 16, &a0, 16); memcpy(buf + 32, &b, 16);
  fo_sha256(buf, 48, (uint8_t *)out);
  memcpy(buf + 16, &a1, 16);
  fo_sha256(buf, 48, (uint8_t *)out + 32);
}

/* Hashable: 64-byte message -> 2 int4 (blake3.cuh:151-155). */
void fo_blake3_hash(const int32_t iv[8], const fo_i4 msg[4], fo_i4 out[2]) {
  if (g_hash_kind == 1) { sha_hash(iv, msg, out); return; }
  uint32_t m[16], o[16];
  for (int i = 0; i < 4; ++i) i4_words(msg[i], m + 4 * i);
  fo_blake3_compress((const uint32_t *)iv, m, 0, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT | B3_KEYED, o);
  out[0] = words_i4(o); out[1] = words_i4(o + 4);
}

/* XorHashable: (a, b) -> 4 int4 = H(a with lsb 0 | b) | H(a with lsb 1 | b), block_len 32 (blake3.cuh:157-169). */
void fo_blake3_xorhash(const int32_t iv[8], fo_i4 a, fo_i4 b, fo_i4 out[4]) {
  if (g_hash_kind == 1) { sha_xorhash(iv, a, b, out); return; }
  uint32_t m[16] = {0}, o[16];
  const uint32_t fl = B3_CHUNK_START | B3_CHUNK_END | B3_ROOT | B3_KEYED;
  i4_words(set_lsb(a, 0), m); i4_words(b, m + 4);
  fo_blake3_compress((const uint32_t *)iv, m, 0, 32, fl, o);
  out[0] = words_i4(o); out[1] = words_i4(o + 4);
  i4_words(set_lsb(a, 1), m);
  fo_blake3_compress((const uint32_t *)iv, m, 0, 32, fl, o);
  out[2] = words_i4(o); out[3] = words_i4(o + 4);
}

static inline fo_i4 pack_in(u128 x) { /* util.cuh:47-64 (Pack), as the 128-bit In */
  fo_i4 v = {(int32_t)(uint32_t)x, (int32_t)(uint32_t)(x >> 32), (int32_t)(uint32_t)(x >> 64), (int32_t)(uint32_t)(x >> 96)};
  return v;
}

/* fss::Vdpf::Gen (include/fss/vdpf.cuh:103-189): DPF levels, then cs = H(a, s0) ^ H(a, s1) and the output
 * correction word.  cws: in_bits slots.  Returns 1 when t0 == t1 at the end (vdpf.cuh:177). */
int fo_vdpf_gen(const fo_prg *p, const int32_t iv[8], int in_bits, int group, fo_cw *cws, fo_i4 cs[4], fo_i4 *ocw,
                const fo_i4 s0s[2], uint64_t a_lo, uint64_t a_hi, fo_i4 beta) {
  u128 a = mk128(a_lo, a_hi);
  fo_i4 s0 = set_lsb(s0s[0], 0), s1 = set_lsb(s0s[1], 0);
  int t0 = 0, t1 = 1;
  beta = set_lsb(beta, 0);
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 g0[2], g1[2];
    fo_prg_gen(p, s0, g0);
    fo_prg_gen(p, s1, g1);
    int t0l = get_lsb(g0[0]), t0r = get_lsb(g0[1]), t1l = get_lsb(g1[0]), t1r = get_lsb(g1[1]);
    fo_i4 s0l = set_lsb(g0[0], 0), s0r = set_lsb(g0[1], 0), s1l = set_lsb(g1[0], 0), s1r = set_lsb(g1[1], 0);
    int a_bit = bit_of(a, in_bits - 1 - i);
    fo_i4 s_cw = !a_bit ? xor4(s0r, s1r) : xor4(s0l, s1l);
    int tl_cw = t0l ^ t1l ^ a_bit ^ 1;
    int tr_cw = t0r ^ t1r ^ a_bit;
    if (!a_bit) {
      s0 = s0l; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1l; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0l ^ tl_cw) : t0l;
      t1 = t1 ? (t1l ^ tl_cw) : t1l;
    } else {
      s0 = s0r; if (t0) s0 = xor4(s0, s_cw);
      s1 = s1r; if (t1) s1 = xor4(s1, s_cw);
      t0 = t0 ? (t0r ^ tr_cw) : t0r;
      t1 = t1 ? (t1r ^ tr_cw) : t1r;
    }
    cws[i] = mk_cw(set_lsb(s_cw, tl_cw), tr_cw);
  }
  fo_i4 a_buf = pack_in(a), p0[4], p1[4];
  fo_blake3_xorhash(iv, a_buf, s0, p0);
  fo_blake3_xorhash(iv, a_buf, s1, p1);
  for (int j = 0; j < 4; ++j) cs[j] = xor4(p0[j], p1[j]);
  if (t0 == t1) return 1;
  gel v = g_add(group, g_add(group, g_from(group, beta), g_neg(group, g_from(group, s0))), g_from(group, s1));
  if (t1) v = g_neg(group, v);
  *ocw = g_into(group, v);
  return 0;
}

/* fss::Vdpf::Eval (vdpf.cuh:204-257): y and the corrected verification hash pi_tilde (4 int4). */
fo_i4 fo_vdpf_eval(const fo_prg *p, const int32_t iv[8], int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                   const fo_i4 cs[4], fo_i4 ocw, uint64_t x_lo, uint64_t x_hi, fo_i4 pi_tilde[4]) {
  u128 x = mk128(x_lo, x_hi);
  fo_i4 s = set_lsb(s0, 0);
  int t = b;
  for (int i = 0; i < in_bits; ++i) {
    fo_i4 sl, sr; int tl, tr;
    dpf_expand(p, s, t, &cws[i], &sl, &tl, &sr, &tr);
    if (!bit_of(x, in_bits - 1 - i)) { s = sl; t = tl; } else { s = sr; t = tr; }
  }
  gel y = g_from(group, s);
  if (t) y = g_add(group, y, g_from(group, ocw));
  if (b) y = g_neg(group, y);
  fo_blake3_xorhash(iv, pack_in(x), s, pi_tilde);
  if (t) for (int j = 0; j < 4; ++j) pi_tilde[j] = xor4(pi_tilde[j], cs[j]);
  return g_into(group, y);
}

/* fss::Vdpf::Prove (vdpf.cuh:268-278): pi = cs; for every pi_tilde: pi[0..1] ^= H'(pi ^ pi_tilde). */
void fo_vdpf_prove(const int32_t iv[8], const fo_i4 *pi_tildes, size_t n, const fo_i4 cs[4], fo_i4 pi[4]) {
  for (int j = 0; j < 4; ++j) pi[j] = cs[j];
  for (size_t i = 0; i < n; ++i) {
    fo_i4 in[4], h[2];
    for (int j = 0; j < 4; ++j) in[j] = xor4(pi[j], pi_tildes[4 * i + j]);
    fo_blake3_hash(iv, in, h);
    pi[0] = xor4(pi[0], h[0]);
    pi[1] = xor4(pi[1], h[1]);
  }
}

/* ------------------------------------------------------------------ batched drivers */
static void prg_thread_copy(const fo_prg *src, fo_prg *dst) {
  *dst = *src;
  memset(dst->evp, 0, sizeof(dst->evp));
  if (src->use_evp == 1) evp_make(dst); /* one EVP ctx set per thread */
}

static fo_i4 eval_one(const fo_prg *p, fo_i4 hk, int scheme, int in_bits, int group, int b, fo_i4 s0,
                      const fo_cw *cws, fo_i4 ocw, uint64_t x, uint64_t xh) {
  switch (scheme) {
    case FO_DPF: return fo_dpf_eval(p, in_bits, group, b, s0, cws, x, xh);
    case FO_DCF: return fo_dcf_eval(p, in_bits, group, b, s0, cws, x, xh);
    case FO_HT: return fo_ht_eval(p, hk, in_bits, group, b, s0, cws, ocw, x, xh);
    default: {
      fo_i4 r = {fo_grotto_eval_walk(p, in_bits, b, s0, cws, x), 0, 0, 0};
      return r;
    }
  }
}

void fo_eval_points(const fo_prg *p, fo_i4 hk, int scheme, int in_bits, int group, int b, fo_i4 s0, const fo_cw *cws,
                    fo_i4 ocw, const uint64_t *xs, const uint64_t *xs_hi, size_t n, fo_i4 *ys, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    fo_prg tp;
    prg_thread_copy(p, &tp);
#pragma omp for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
      ys[i] = eval_one(&tp, hk, scheme, in_bits, group, b, s0, cws, ocw, xs[i], xs_hi ? xs_hi[i] : 0);
    fo_prg_free(&tp);
  }
}

void fo_eval_keys(const fo_prg *p, fo_i4 hk, int scheme, int in_bits, int group, int b, const fo_i4 *s0s,
                  const fo_cw *cws, size_t stride, const fo_i4 *ocws, const uint64_t *xs, const uint64_t *xs_hi,
                  size_t n, fo_i4 *ys, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    fo_prg tp;
    prg_thread_copy(p, &tp);
#pragma omp for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
      ys[i] = eval_one(&tp, hk, scheme, in_bits, group, b, s0s[i], cws + (size_t)i * stride, ocws ? ocws[i] : kZero4,
                       xs[i], xs_hi ? xs_hi[i] : 0);
    fo_prg_free(&tp);
  }
}

void fo_gen_keys(const fo_prg *p, fo_i4 hk, int scheme, int in_bits, int group, int pred, const fo_i4 *s0s,
                 const uint64_t *alphas, const uint64_t *alphas_hi, const fo_i4 *betas, size_t n, fo_cw *cws,
                 size_t stride, fo_i4 *ocws, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    fo_prg tp;
    prg_thread_copy(p, &tp);
#pragma omp for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
      fo_cw *c = cws + (size_t)i * stride;
      const fo_i4 *ss = s0s + 2 * (size_t)i;
      const uint64_t ah = alphas_hi ? alphas_hi[i] : 0;
      switch (scheme) {
        case FO_DPF: fo_dpf_gen(&tp, in_bits, group, c, ss, alphas[i], ah, betas[i]); break;
        case FO_DCF: fo_dcf_gen(&tp, in_bits, group, pred, c, ss, alphas[i], ah, betas[i]); break;
        case FO_HT: fo_ht_gen(&tp, hk, in_bits, group, c, &ocws[i], ss, alphas[i], ah, betas[i]); break;
        default: fo_grotto_gen(&tp, in_bits, c, ss, alphas[i], ah); break;
      }
    }
    fo_prg_free(&tp);
  }
}
