// Quad-lane walks: one PRG-tree walk per four adjacent lanes, a state column each (DPP quad_perm exchanges).
// Used where a walk is a latency chain run by lone waves: the first frontier of EvalAll (frontier_quad_kernel,
// fss_evalall.hpp) and small batches of points or keys (points_quad_kernel, keys_quad_kernel, below).
#pragma once
#include <cstddef>
#include <type_traits>
#include "fss_dev.hpp"
#include "fss_point.hpp"

namespace fssamd {

// -- one PRG call per quad --------------------------------------------------------------------------------
// A short walk is a latency chain: `depth` PRG calls one after the other, and a lone wave issues a VALU op only
// every ~7 clk, so an AES block costs ~2 us (a ChaCha permutation ~3) however few walks there
// are.  Here four adjacent lanes share one walk and lane q holds word q of every 16-byte value.
//   AES: a state column per lane.  Per round a lane does the four table lookups of ITS column's bytes (T0[b0],
//   T1[b1], T2[b2], T3[b3]) and column q of the next state is A(q) ^ B(q+1) ^ C(q+2) ^ D(q+3), the three foreign
//   terms fetched with DPP quad_perm moves -- ~10 VALU and 4 lookups per round on the critical path instead of
//   ~36 and 16.  Same lookups in total, a quarter of the chain.  The lane's column of the two-key schedule
//   {rkA, rkA ^ rkB} (DevCtx::sel01) sits in 22 VGPRs.
//   ChaCha: the 4 x 4 state a column per lane.  The column round is one quarter-round per lane, the diagonal round
//   rotates rows 1..3 by 1..3 lanes, does the quarter-round and rotates back: 30 VALU per double round
//   instead of 96.  Word q of output row r is lane q's r-th word, which is what the walk wants.
#define FSS_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL>
__device__ __forceinline__ uint32_t quad_get(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
// Column q of AES_k(in) for the quad's block; SEL: k = key A where m == 0, key B where m == ~0 (ka = rkA,
// kd = rkA ^ rkB, this lane's column of every round).
template <bool SEL>
__device__ __forceinline__ uint32_t quad_aes(const Lds& L, uint32_t in, uint32_t m, const uint32_t ka[11], const uint32_t kd[11]) {
  uint32_t s = SEL ? xor3(in, ka[0], m & kd[0]) : (in ^ ka[0]);
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    const uint32_t A = lk<0, 0>(L, s), B = lk<1, 1>(L, s), C = lk<2, 2>(L, s), D = lk<3, 3>(L, s);
    const uint32_t x = xor3(A, quad_get<FSS_QP(1, 2, 3, 0)>(B), quad_get<FSS_QP(2, 3, 0, 1)>(C));
    s = xor3(x, quad_get<FSS_QP(3, 0, 1, 2)>(D), SEL ? and_xor(m, kd[r], ka[r]) : ka[r]);
  }
  // last round: the tables that hold S in the output byte position (aes_mmo)
  const uint32_t A = lk<0, 2>(L, s), B = quad_get<FSS_QP(1, 2, 3, 0)>(lk<1, 3>(L, s)),
                 C = quad_get<FSS_QP(2, 3, 0, 1)>(lk<2, 0>(L, s)), D = quad_get<FSS_QP(3, 0, 1, 2)>(lk<3, 1>(L, s));
  const uint32_t pcol = (A & 0xffu) | (B & 0xff00u) | (C & 0xff0000u) | (D & 0xff000000u);
  return pcol ^ (SEL ? and_xor(m, kd[10], ka[10]) : ka[10]);
}

// Two blocks of the same input under two keys (DCF: seed and value of the taken side; both children of a node),
// their rounds interleaved so that the lookups of one fly during the exchanges of the other.
// SEL = false: plain keys ka / kb (kd, ke unused).
template <bool SEL>
__device__ __forceinline__ void quad_aes_pair(const Lds& L, uint32_t in, uint32_t m, const uint32_t ka[11], const uint32_t kd[11],
                                              const uint32_t kb[11], const uint32_t ke[11], uint32_t& oa, uint32_t& ob) {
  uint32_t s = SEL ? xor3(in, ka[0], m & kd[0]) : (in ^ ka[0]), t = SEL ? xor3(in, kb[0], m & ke[0]) : (in ^ kb[0]);
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    const uint32_t A = lk<0, 0>(L, s), B = lk<1, 1>(L, s), C = lk<2, 2>(L, s), D = lk<3, 3>(L, s);
    const uint32_t E = lk<0, 0>(L, t), F = lk<1, 1>(L, t), G = lk<2, 2>(L, t), H = lk<3, 3>(L, t);
    const uint32_t x = xor3(A, quad_get<FSS_QP(1, 2, 3, 0)>(B), quad_get<FSS_QP(2, 3, 0, 1)>(C));
    const uint32_t y = xor3(E, quad_get<FSS_QP(1, 2, 3, 0)>(F), quad_get<FSS_QP(2, 3, 0, 1)>(G));
    s = xor3(x, quad_get<FSS_QP(3, 0, 1, 2)>(D), SEL ? and_xor(m, kd[r], ka[r]) : ka[r]);
    t = xor3(y, quad_get<FSS_QP(3, 0, 1, 2)>(H), SEL ? and_xor(m, ke[r], kb[r]) : kb[r]);
  }
  const uint32_t A = lk<0, 2>(L, s), B = quad_get<FSS_QP(1, 2, 3, 0)>(lk<1, 3>(L, s)),
                 C = quad_get<FSS_QP(2, 3, 0, 1)>(lk<2, 0>(L, s)), D = quad_get<FSS_QP(3, 0, 1, 2)>(lk<3, 1>(L, s));
  const uint32_t E = lk<0, 2>(L, t), F = quad_get<FSS_QP(1, 2, 3, 0)>(lk<1, 3>(L, t)),
                 G = quad_get<FSS_QP(2, 3, 0, 1)>(lk<2, 0>(L, t)), H = quad_get<FSS_QP(3, 0, 1, 2)>(lk<3, 1>(L, t));
  oa = ((A & 0xffu) | (B & 0xff00u) | (C & 0xff0000u) | (D & 0xff000000u)) ^ (SEL ? and_xor(m, kd[10], ka[10]) : ka[10]);
  ob = ((E & 0xffu) | (F & 0xff00u) | (G & 0xff0000u) | (H & 0xff000000u)) ^ (SEL ? and_xor(m, ke[10], kb[10]) : kb[10]);
}

// What a quad walk of SCHEME needs from the PRG, per table type (Lds = AES-128-MMO, ChaChaL = ChaCha): word q
// of   take(in, m): Gen(in)[m ? 1 : 0] (DPF / Grotto);  both: Gen(in)[0], Gen(in)[1];  take_sv: DCF's seed and
// value of the side m ([sl, vl, sr, vr], dcf.cuh:223);  hash: Gen(in)[0] with mul = 1 (Half-Tree).
template <class LT, int SCHEME>
struct QuadPrg;

// DCF on the AES tables, few walks (a batch of at most an eighth of the chip's lanes: points_quad_kernel / keys_quad_kernel
// <.., OCT = true>, chosen by the host): EIGHT lanes per walk.  A level costs a seed block and a value block of the same input; one quad
// interleaving both (quad_aes_pair) issues twice the instructions, and a lone wave's level is bound by the instructions it
// issues (1.8 us per level against the DPF walk's 0.95).  In an octet the four lanes after the walk's quad compute the value
// block while the quad computes the seed block -- the same instructions, other keys (is_v) -- and only the seed quad's result
// feeds the next level; the value quad fetches the level's node from it (one DPP row shift) and keeps the running value.
// DCF n = 32, 1 .. 2^15 points: 58 -> 37 - 42 us per call.  Where every lane has work anyway (the 2^15 walks of an EvalAll
// frontier, 2^16 points and more) the pair form stays: octets there only double the waves that contend for a SIMD
// (frontier walks 26 -> 32 us, 2^16 points 58 -> 77 us; profiles/r6/dcf_octet_walks.txt).
// lane i <- lane i - 4 of its row of sixteen: the value quad of an octet reads its seed quad
__device__ __forceinline__ uint32_t quad_from_seed_quad(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
}

template <int SCHEME>
struct QuadPrg<Lds, SCHEME> {
  const Lds* L;
  uint32_t ka[11], kd[11];
  // DCF: the value block's keys 1 / 3; DPF: key 1 alone, for both-children expansions
  uint32_t kb[SCHEME == S_HT ? 1 : 11], ke[SCHEME == S_DCF ? 11 : 1];
  uint32_t hk;  // Half-Tree: this lane's word of hash_key
  bool oct = false, is_v = false;  // DCF: a walk per octet; this lane belongs to the octet's value quad (keys 1 / 3 in ka / kd)
  __device__ __forceinline__ void init(const Lds& l, const DevCtx* __restrict__ c, int q, bool octets = false) {
    L = &l;
    oct = SCHEME == S_DCF && octets;
    is_v = oct && ((threadIdx.x >> 2) & 1) != 0;
    const uint32_t* gk = c->self_dev + (SCHEME == S_DPF ? offsetof(DevCtx, sel01)
                                        : (SCHEME == S_DCF ? (is_v ? offsetof(DevCtx, sel13) : offsetof(DevCtx, sel02)) : offsetof(DevCtx, rk))) / 4;
    const uint32_t* gv = c->self_dev + offsetof(DevCtx, sel13) / 4;
#pragma unroll
    for (int r = 0; r < 11; ++r) {
      ka[r] = SCHEME == S_HT ? gk[4 * r + q] : gk[8 * r + q];  // Half-Tree hashes with key 0 (half_tree_dpf.cuh:195)
      kd[r] = SCHEME == S_HT ? 0u : gk[8 * r + 4 + q];
      if constexpr (SCHEME == S_DCF) { kb[r] = gv[8 * r + q]; ke[r] = gv[8 * r + 4 + q]; }
      if constexpr (SCHEME == S_DPF) kb[r] = ka[r] ^ kd[r];
    }
    hk = SCHEME == S_HT ? (c->self_dev + offsetof(DevCtx, hash_key) / 4)[q] : 0u;
  }
  __device__ __forceinline__ uint32_t take(uint32_t in, uint32_t m) const { return quad_aes<true>(*L, in, m, ka, kd) ^ in; }
  __device__ __forceinline__ void both(uint32_t in, uint32_t& l, uint32_t& r) const {
    quad_aes_pair<false>(*L, in, 0u, ka, kd, kb, ke, l, r);
    l ^= in; r ^= in;
  }
  __device__ __forceinline__ void take_sv(uint32_t in, uint32_t m, uint32_t& os, uint32_t& ov) const {
    quad_aes_pair<true>(*L, in, m, ka, kd, kb, ke, os, ov);
    os ^= in; ov ^= in;
  }
  __device__ __forceinline__ uint32_t hash(uint32_t in) const { return quad_aes<false>(*L, in, 0u, ka, kd) ^ in; }
};

template <int SCHEME>
struct QuadPrg<ChaChaL, SCHEME> {
  uint32_t k16, k32;  // this lane's word of the 16-byte-key / 32-byte-key constant (chacha_rows)
  uint32_t r3;        // this lane's word of row 3: 0, 0, nonce
  uint32_t hk;
  int rounds;
  __device__ __forceinline__ void init(const ChaChaL&, const DevCtx* __restrict__ c, int q) {
    k16 = q == 0 ? 0x61707865u : (q == 1 ? 0x3120646eu : (q == 2 ? 0x79622d36u : 0x6b206574u));
    k32 = q == 0 ? 0x61707865u : (q == 1 ? 0x3320646eu : (q == 2 ? 0x79622d32u : 0x6b206574u));
    r3 = q == 2 ? c->chacha_nonce[0] : (q == 3 ? c->chacha_nonce[1] : 0u);
    hk = SCHEME == S_HT ? (c->self_dev + offsetof(DevCtx, hash_key) / 4)[q] : 0u;
    rounds = c->chacha_rounds;
  }
  // word q of the four output rows of ChaCha<mul>(in) (chacha_rows<WIDE>)
  template <bool WIDE>
  __device__ __forceinline__ void rows(uint32_t in, uint32_t& a, uint32_t& b, uint32_t& c_, uint32_t& d) const {
    const uint32_t k = WIDE ? k32 : k16;
    a = k; b = in; c_ = in; d = r3;
#pragma unroll 1
    for (int i = 0; i < rounds; i += 2) {
      chacha_qr(a, b, c_, d);  // column round: x[q], x[4+q], x[8+q], x[12+q]
      b = quad_get<FSS_QP(1, 2, 3, 0)>(b);  // diagonal round: x[q], x[4 + (q+1)%4], x[8 + (q+2)%4], x[12 + (q+3)%4]
      c_ = quad_get<FSS_QP(2, 3, 0, 1)>(c_);
      d = quad_get<FSS_QP(3, 0, 1, 2)>(d);
      chacha_qr(a, b, c_, d);
      b = quad_get<FSS_QP(3, 0, 1, 2)>(b);
      c_ = quad_get<FSS_QP(2, 3, 0, 1)>(c_);
      d = quad_get<FSS_QP(1, 2, 3, 0)>(d);
    }
    a ^= k; b ^= in; c_ ^= in; d ^= r3;
  }
  __device__ __forceinline__ uint32_t take(uint32_t in, uint32_t m) const {
    uint32_t a, b, c_, d;
    rows<false>(in, a, b, c_, d);
    return __builtin_amdgcn_bitop3_b32(m, b, a, kSelect);
  }
  __device__ __forceinline__ void both(uint32_t in, uint32_t& l, uint32_t& r) const {
    uint32_t c_, d;
    rows<false>(in, l, r, c_, d);
  }
  __device__ __forceinline__ void take_sv(uint32_t in, uint32_t m, uint32_t& os, uint32_t& ov) const {
    uint32_t a, b, c_, d;
    rows<true>(in, a, b, c_, d);
    os = __builtin_amdgcn_bitop3_b32(m, c_, a, kSelect);
    ov = __builtin_amdgcn_bitop3_b32(m, d, b, kSelect);
  }
  __device__ __forceinline__ uint32_t hash(uint32_t in) const {
    uint32_t a, b, c_, d;
    rows<false>(in, a, b, c_, d);
    return b;  // mul = 1: row 1 (chacha.cuh:116-118)
  }
};

// One level of a quad's walk (DpfWalk / DcfWalk / HtWalk ::level, a word per lane): m = the path bit (0 / ~0,
// the same in the four lanes), st = this lane's word of the node, v = DCF's running value (whole in every lane).
// The level's correction word: w = this lane's word of cw.s; aux = the word whose bit 0 is tr_cw (DPF: the tr
// byte, DCF: v.w); vcw = DCF's value words.
template <class LT, int SCHEME>
__device__ __forceinline__ void quad_level_cw(const QuadPrg<LT, SCHEME>& P, const Grp& g, uint32_t w, uint32_t aux,
                                              const uint32_t* vcw_in, uint32_t m, uint32_t is3, uint32_t& st, Acc& v) {
  const uint32_t tm = lsb_mask(quad_get<FSS_QP(3, 3, 3, 3)>(st));
  if constexpr (SCHEME == S_DPF) {
    const uint32_t o = P.take(st & ~(is3 & 1u), m);
    // correction (DpfWalk::apply): word 3 carries tl_cw / tr_cw in bit 0
    const uint32_t cwq = w ^ (is3 & m & ((w ^ aux) & 1u));
    st = and_xor(tm, cwq, o);
  } else if constexpr (SCHEME == S_DCF) {  // DcfWalk::level
    if constexpr (std::is_same<LT, Lds>::value) {
      if (P.oct) {  // (wave-uniform) an octet per walk: see QuadPrg<Lds, SCHEME>
        // both quads start from the level's node (the value quad takes it from the seed quad) and run ONE block each: the
        // seed quad's result is the next node, the value quad's is added to the running value it keeps.  What the other
        // quad computes in the same registers (a value out of seed blocks, a node out of value blocks) is never read.
        const uint32_t from_seed_quad = quad_from_seed_quad(st);  // every lane executes the shift: a DPP read of a masked-off lane is no read
        const uint32_t node = P.is_v ? from_seed_quad : st;
        const uint32_t tmn = lsb_mask(quad_get<FSS_QP(3, 3, 3, 3)>(node));
        const uint32_t o = P.take(node & ~(is3 & 1u), m);
        uint32_t ovo[4] = {quad_get<FSS_QP(0, 0, 0, 0)>(o), quad_get<FSS_QP(1, 1, 1, 1)>(o), quad_get<FSS_QP(2, 2, 2, 2)>(o),
                           quad_get<FSS_QP(3, 3, 3, 3)>(o) & ~1u};
        v.add(g, ovo);
        uint32_t vcwo[4] = {vcw_in[0], vcw_in[1], vcw_in[2], aux & ~1u};
        v.add_if(g, tmn, vcwo);
        const uint32_t cwo = w ^ (is3 & m & ((w ^ aux) & 1u));  // word 3: tl_cw in s.w, tr_cw in v.w
        st = and_xor(tmn, cwo, o);
        return;
      }
    }
    uint32_t os, ovq;
    P.take_sv(st & ~(is3 & 1u), m, os, ovq);
    uint32_t ov[4] = {quad_get<FSS_QP(0, 0, 0, 0)>(ovq), quad_get<FSS_QP(1, 1, 1, 1)>(ovq), quad_get<FSS_QP(2, 2, 2, 2)>(ovq),
                      quad_get<FSS_QP(3, 3, 3, 3)>(ovq) & ~1u};
    v.add(g, ov);
    uint32_t vcw[4] = {vcw_in[0], vcw_in[1], vcw_in[2], aux & ~1u};
    v.add_if(g, tm, vcw);
    const uint32_t cwq = w ^ (is3 & m & ((w ^ aux) & 1u));  // word 3: tl_cw in s.w, tr_cw in v.w
    st = and_xor(tm, cwq, os);
  } else {
    const uint32_t h = P.hash(st ^ P.hk);  // HtWalk::level
    st = and_xor(m, st, and_xor(tm, w, h));
  }
}

// The same with the correction word read from a key's raw 32-byte slots (kc = the key, level i).
template <class LT, int SCHEME>
__device__ __forceinline__ void quad_level(const QuadPrg<LT, SCHEME>& P, const Grp& g, const uint32_t* __restrict__ kc, int i,
                                           uint32_t m, int q, uint32_t is3, uint32_t& st, Acc& v) {
  const uint32_t* p = kc + 8 * i;
  if constexpr (SCHEME == S_DCF) {
    const uint32_t vcw[3] = {p[4], p[5], p[6]};
    quad_level_cw(P, g, p[q], p[7], vcw, m, is3, st, v);
  } else {
    quad_level_cw(P, g, p[q], SCHEME == S_DPF ? p[4] : 0u, nullptr, m, is3, st, v);
  }
}

// The leaf of a quad's walk: the four words gathered into every lane, the scheme's own finish on them, lane q
// keeps word q of y.
template <int SCHEME, class LT, class OCW>
__device__ __forceinline__ uint32_t quad_finish(const LT& L, const DevCtx* __restrict__ c, const Grp& g, int party, int q, uint32_t st,
                                                const Acc& v, const X128& x, const uint32_t last_cw[4], uint32_t extra,
                                                const OCW& oc) {
  const uint4 node = make_uint4(quad_get<FSS_QP(0, 0, 0, 0)>(st), quad_get<FSS_QP(1, 1, 1, 1)>(st), quad_get<FSS_QP(2, 2, 2, 2)>(st),
                                quad_get<FSS_QP(3, 3, 3, 3)>(st));
  uint4 y;
  if constexpr (SCHEME == S_DPF) {
    DpfWalk w;
    w.resume(node, x);
    y = w.finish(g, party, oc);
  } else if constexpr (SCHEME == S_DCF) {
    DcfWalk w;
    w.resume(node, make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]), x);
    y = w.finish(g, party, oc);
  } else {
    HtWalk w;
    w.resume(node, x);
    y = w.finish(L, c, g, party, last_cw, extra, oc);
  }
  return q == 0 ? y.x : (q == 1 ? y.y : (q == 2 ? y.z : y.w));
}

// -- small point batches: a quad walk from the root to the leaf -------------------------------------------
// Below ~2^17 points a point evaluation is the frontier's situation: a chain of in_bits PRG calls run by lone
// waves.  One walk per quad of lanes runs a level in ~0.9 us instead of 2.2 - 3.  Shared key; launched by
// the *_eval_points entry points for batches that leave most of the chip idle.
template <class LT, int SCHEME, bool OCT = false>  // OCT: DCF / AES, an octet of lanes per walk (ignored for every other pair)
__global__ __launch_bounds__(1024) void points_quad_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                          const uint32_t* __restrict__ k_seed,
                                                          const uint32_t* __restrict__ k_cws,
                                                          const uint32_t* __restrict__ k_ocw, EvalArgs a) {
  static_assert(SCHEME == S_DPF || SCHEME == S_DCF || SCHEME == S_HT, "quad point walk: DPF / DCF / Half-Tree");
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(LT, te0);
  const int q = threadIdx.x & 3;
  const uint32_t is3 = q == 3 ? ~0u : 0u;
  QuadPrg<LT, SCHEME> P;
  constexpr bool oct = OCT && SCHEME == S_DCF && std::is_same<LT, Lds>::value;  // eight lanes per walk (QuadPrg<Lds, SCHEME>)
  if constexpr (std::is_same<LT, Lds>::value) P.init(L, c, q, oct);
  else P.init(L, c, q);
  const Grp g = make_grp(a.group, c);
  const int n = a.in_bits, depth = SCHEME == S_HT ? n - 1 : n;  // Half-Tree: the last level is the conversion
  constexpr int sh = oct ? 3 : 2;
  const int64_t nquads = ((int64_t)gridDim.x * blockDim.x) >> sh;
  uint32_t* __restrict__ ys = reinterpret_cast<uint32_t*>(a.ys);
  for (int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> sh; j < a.n; j += nquads) {
    const X128 x = load_x(a.xs, a.x_bytes, j);
    uint32_t st = k_seed[q];
    st = (st & ~(is3 & 1u)) | (is3 & (uint32_t)a.party);
    Acc v;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) quad_level(P, g, k_cws, i, bit_mask(x, n - 1 - i), q, is3, st, v);
    if constexpr (oct) {  // the value quad holds the running value: it takes the leaf node too, and finishes alone
      st = quad_from_seed_quad(st);
      if (((threadIdx.x >> 2) & 1) == 0) continue;
    }
    const uint32_t* p = k_cws + 8 * (SCHEME == S_HT ? n - 1 : n) + (SCHEME == S_DCF ? 4 : 0);
    const uint32_t cwl[4] = {p[0], p[1], p[2], p[3]};  // DPF / DCF: the output correction word; Half-Tree: the last level's
    if constexpr (SCHEME == S_HT) {
      const uint32_t oc[4] = {k_ocw[0], k_ocw[1], k_ocw[2], k_ocw[3]};
      ys[4 * j + q] = quad_finish<SCHEME>(L, c, g, a.party, q, st, v, x, cwl, p[4], oc);
    } else {
      ys[4 * j + q] = quad_finish<SCHEME>(L, c, g, a.party, q, st, v, x, cwl, 0u, cwl);
    }
  }
}

// Small batches of independent level-major keys (fss::gpu::*EvalPointGpu's inputs): a quad per key.
template <class LT, int SCHEME, bool OCT = false>
__global__ __launch_bounds__(1024) void keys_quad_kernel(const DevCtx K, const uint32_t* __restrict__ te0, EvalArgs a) {
  static_assert(SCHEME == S_DPF || SCHEME == S_DCF || SCHEME == S_HT, "quad key walk: DPF / DCF / Half-Tree");
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(LT, te0);
  const int q = threadIdx.x & 3;
  const uint32_t is3 = q == 3 ? ~0u : 0u;
  QuadPrg<LT, SCHEME> P;
  constexpr bool oct = OCT && SCHEME == S_DCF && std::is_same<LT, Lds>::value;
  if constexpr (std::is_same<LT, Lds>::value) P.init(L, c, q, oct);
  else P.init(L, c, q);
  const Grp g = make_grp(a.group, c);
  const int n = a.in_bits, depth = SCHEME == S_HT ? n - 1 : n;
  constexpr int sh = oct ? 3 : 2;
  const int64_t nk = a.n, nquads = ((int64_t)gridDim.x * blockDim.x) >> sh;
  uint32_t* __restrict__ ys = reinterpret_cast<uint32_t*>(a.ys);
  for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> sh; k < nk; k += nquads) {
    const X128 x = load_x(a.xs, a.x_bytes, k);
    uint32_t st = reinterpret_cast<const uint32_t*>(a.seeds + k)[q];
    st = (st & ~(is3 & 1u)) | (is3 & (uint32_t)a.party);
    uint64_t trb = 0, trbh = 0;
    if constexpr (SCHEME == S_DPF) { trb = load_tr(a.tr_bits, a.tr32, k); trbh = n > 64 ? a.tr_bits[nk + k] : 0ull; }
    Acc v;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
      const uint32_t m = bit_mask(x, n - 1 - i);
      const uint32_t w = reinterpret_cast<const uint32_t*>(a.cw_s + (int64_t)i * nk + k)[q];
      if constexpr (SCHEME == S_DCF) {
        const uint4 cv = a.cw_v[(int64_t)i * nk + k];
        const uint32_t vcw[3] = {cv.x, cv.y, cv.z};
        quad_level_cw(P, g, w, cv.w, vcw, m, is3, st, v);
      } else {
        const uint32_t tr = SCHEME == S_DPF ? (uint32_t)(((i < 64 ? trb : trbh) >> (i & 63)) & 1ull) : 0u;
        quad_level_cw(P, g, w, tr, nullptr, m, is3, st, v);
      }
    }
    if constexpr (oct) {
      st = quad_from_seed_quad(st);
      if (((threadIdx.x >> 2) & 1) == 0) continue;
    }
    const uint4 u = a.out_cw[k];  // DPF / DCF: cws[in_bits]; Half-Tree: the key's ocw
    const uint32_t oc[4] = {u.x, u.y, u.z, u.w};
    if constexpr (SCHEME == S_HT) {
      const uint4 t = a.cw_s[(int64_t)(n - 1) * nk + k];
      const uint32_t cw[4] = {t.x, t.y, t.z, t.w};
      ys[4 * k + q] = quad_finish<SCHEME>(L, c, g, a.party, q, st, v, x, cw, a.extra[k], oc);
    } else {
      ys[4 * k + q] = quad_finish<SCHEME>(L, c, g, a.party, q, st, v, x, oc, 0u, oc);
    }
  }
}

}  // namespace fssamd
