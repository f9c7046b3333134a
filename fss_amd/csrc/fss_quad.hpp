// Quad-lane walks: one PRG-tree walk per four adjacent lanes, an AES state column each (DPP quad_perm
// exchanges).  Used where a walk is a latency chain run by lone waves: the first frontier of EvalAll
// (frontier_quad_kernel, fss_evalall.hpp) and small point batches (points_quad_kernel, below).  AES tables only.
#pragma once
#include <cstddef>
#include "fss_dev.hpp"
#include "fss_point.hpp"

namespace fssamd {

// -- one AES block per quad ------------------------------------------------------------------------------
// A short walk is a latency chain: `depth` blocks one after the other, and a lone wave issues a VALU op only
// every ~7 clk, so a block costs ~2 us however few walks there are.  Here four adjacent lanes
// share one walk, a state column each: per round a lane does the four table lookups of ITS column's bytes
// (T0[b0], T1[b1], T2[b2], T3[b3]) and column q of the next state is A(q) ^ B(q+1) ^ C(q+2) ^ D(q+3), the
// three foreign terms fetched with DPP quad_perm moves -- ~10 VALU and 4 lookups per round on the critical
// path instead of ~36 and 16.  Same lookups in total, a quarter of the chain.  The lane's column of the
// two-key schedule {rkA, rkA ^ rkB} (DevCtx::sel01) sits in 22 VGPRs.
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

// Two blocks of the same input under two selected keys (DCF: seed and value of the taken side), their
// rounds interleaved so that the lookups of one fly during the exchanges of the other.
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

// One level of a quad's walk (DpfWalk / DcfWalk / HtWalk ::level, a column per lane): m = the path bit (0 / ~0,
// the same in the four lanes), st = this lane's word of the node, v = DCF's running value (whole in every lane).
// The level's correction word: w = this lane's word of cw.s; aux = the word whose bit 0 is tr_cw (DPF: the tr
// byte, DCF: v.w); vcw = DCF's value words.
template <int SCHEME>
__device__ __forceinline__ void quad_level_cw(const Lds& L, const Grp& g, uint32_t w, uint32_t aux, const uint32_t* vcw_in,
                                              uint32_t m, uint32_t is3, const uint32_t ka[11], const uint32_t kd[11],
                                              const uint32_t* kb, const uint32_t* ke, uint32_t hk, uint32_t& st, Acc& v) {
  const uint32_t tm = lsb_mask(quad_get<FSS_QP(3, 3, 3, 3)>(st));
  if constexpr (SCHEME == S_DPF) {
    const uint32_t trw = aux;
    const uint32_t in = st & ~(is3 & 1u);
    const uint32_t o = quad_aes<true>(L, in, m, ka, kd) ^ in;
    // correction (DpfWalk::apply): word 3 carries tl_cw / tr_cw in bit 0
    const uint32_t cwq = w ^ (is3 & m & ((w ^ trw) & 1u));
    st = and_xor(tm, cwq, o);
  } else if constexpr (SCHEME == S_DCF) {  // DcfWalk::level
    const uint32_t in = st & ~(is3 & 1u);
    uint32_t os, ovq;
    quad_aes_pair<true>(L, in, m, ka, kd, kb, ke, os, ovq);
    os ^= in; ovq ^= in;
    uint32_t ov[4] = {quad_get<FSS_QP(0, 0, 0, 0)>(ovq), quad_get<FSS_QP(1, 1, 1, 1)>(ovq), quad_get<FSS_QP(2, 2, 2, 2)>(ovq),
                      quad_get<FSS_QP(3, 3, 3, 3)>(ovq) & ~1u};
    v.add(g, ov);
    const uint32_t v3 = aux;
    uint32_t vcw[4] = {vcw_in[0], vcw_in[1], vcw_in[2], v3 & ~1u};
    v.add_if(g, tm, vcw);
    const uint32_t cwq = w ^ (is3 & m & ((w ^ v3) & 1u));  // word 3: tl_cw in s.w, tr_cw in v.w
    st = and_xor(tm, cwq, os);
  } else {
    const uint32_t in = st ^ hk;
    const uint32_t h = quad_aes<false>(L, in, 0u, ka, kd) ^ in;  // HtWalk::level
    st = and_xor(m, st, and_xor(tm, w, h));
  }
}

// The same with the correction word read from a key's raw 32-byte slots (kc = the key, level i).
template <int SCHEME>
__device__ __forceinline__ void quad_level(const Lds& L, const Grp& g, const uint32_t* __restrict__ kc, int i, uint32_t m, int q,
                                           uint32_t is3, const uint32_t ka[11], const uint32_t kd[11], const uint32_t* kb,
                                           const uint32_t* ke, uint32_t hk, uint32_t& st, Acc& v) {
  const uint32_t* p = kc + 8 * i;
  if constexpr (SCHEME == S_DCF) {
    const uint32_t vcw[3] = {p[4], p[5], p[6]};
    quad_level_cw<SCHEME>(L, g, p[q], p[7], vcw, m, is3, ka, kd, kb, ke, hk, st, v);
  } else {
    quad_level_cw<SCHEME>(L, g, p[q], SCHEME == S_DPF ? p[4] : 0u, nullptr, m, is3, ka, kd, kb, ke, hk, st, v);
  }
}

// This lane's column of the key schedules a quad walk of SCHEME needs (see frontier_quad_kernel).
template <int SCHEME>
__device__ __forceinline__ void quad_keys(const DevCtx* __restrict__ c, int q, uint32_t ka[11], uint32_t kd[11], uint32_t* kb,
                                          uint32_t* ke, uint32_t& hk) {
  const uint32_t* gk = c->self_dev + (SCHEME == S_DPF ? offsetof(DevCtx, sel01) : (SCHEME == S_DCF ? offsetof(DevCtx, sel02) : offsetof(DevCtx, rk))) / 4;
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

// -- small point batches: the same quad walk from the root to the leaf ------------------------------------
// Below a few thousand points a point evaluation is the frontier's situation: a chain of in_bits blocks run by
// lone waves (2.2 us per level).  One walk per quad of lanes runs it at ~0.9 us per level; at the leaf the four
// words are gathered into every lane and the scheme's own finish (group conversion, Half-Tree's last hash)
// runs on them, lane q storing word q of y.  Shared key only; launched by the *_eval_points entry points for
// batches that leave most of the chip idle.
template <int SCHEME>
__global__ __launch_bounds__(1024) void points_quad_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                         const uint32_t* __restrict__ k_seed,
                                                         const uint32_t* __restrict__ k_cws,
                                                         const uint32_t* __restrict__ k_ocw, EvalArgs a) {
  static_assert(SCHEME == S_DPF || SCHEME == S_DCF || SCHEME == S_HT, "quad point walk: DPF / DCF / Half-Tree");
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(Lds, te0);
  const int q = threadIdx.x & 3;
  const uint32_t is3 = q == 3 ? ~0u : 0u;
  uint32_t ka[11], kd[11], hk;
  uint32_t kb[SCHEME == S_HT ? 1 : 11], ke[SCHEME == S_DCF ? 11 : 1];
  quad_keys<SCHEME>(c, q, ka, kd, kb, ke, hk);
  const Grp g = make_grp(a.group, c);
  const int n = a.in_bits, depth = SCHEME == S_HT ? n - 1 : n;  // Half-Tree: the last level is the conversion
  const int64_t nquads = ((int64_t)gridDim.x * blockDim.x) >> 2;
  uint32_t* __restrict__ ys = reinterpret_cast<uint32_t*>(a.ys);
  for (int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; j < a.n; j += nquads) {
    const X128 x = load_x(a.xs, a.x_bytes, j);
    uint32_t st = k_seed[q];
    st = (st & ~(is3 & 1u)) | (is3 & (uint32_t)a.party);
    Acc v;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) quad_level<SCHEME>(L, g, k_cws, i, bit_mask(x, n - 1 - i), q, is3, ka, kd, kb, ke, hk, st, v);
    const uint32_t node[4] = {quad_get<FSS_QP(0, 0, 0, 0)>(st), quad_get<FSS_QP(1, 1, 1, 1)>(st), quad_get<FSS_QP(2, 2, 2, 2)>(st),
                              quad_get<FSS_QP(3, 3, 3, 3)>(st)};
    uint4 y;
    if constexpr (SCHEME == S_DPF) {
      DpfWalk w;
      w.resume(make_uint4(node[0], node[1], node[2], node[3]), x);
      const uint32_t* p = k_cws + 8 * n;
      const uint32_t oc[4] = {p[0], p[1], p[2], p[3]};
      y = w.finish(g, a.party, oc);
    } else if constexpr (SCHEME == S_DCF) {
      DcfWalk w;
      w.resume(make_uint4(node[0], node[1], node[2], node[3]), make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]), x);
      const uint32_t* p = k_cws + 8 * n + 4;
      const uint32_t oc[4] = {p[0], p[1], p[2], p[3]};
      y = w.finish(g, a.party, oc);
    } else {
      HtWalk w;
      w.resume(make_uint4(node[0], node[1], node[2], node[3]), x);
      const uint32_t* p = k_cws + 8 * (n - 1);
      const uint32_t cw[4] = {p[0], p[1], p[2], p[3]};
      const uint32_t oc[4] = {k_ocw[0], k_ocw[1], k_ocw[2], k_ocw[3]};
      y = w.finish(L, c, g, a.party, cw, p[4], oc);
    }
    ys[4 * j + q] = q == 0 ? y.x : (q == 1 ? y.y : (q == 2 ? y.z : y.w));
  }
}

// Small batches of independent level-major keys (fss::gpu::*EvalPointGpu's inputs): a quad per key.
template <int SCHEME>
__global__ __launch_bounds__(1024) void keys_quad_kernel(const DevCtx K, const uint32_t* __restrict__ te0, EvalArgs a) {
  static_assert(SCHEME == S_DPF || SCHEME == S_DCF || SCHEME == S_HT, "quad key walk: DPF / DCF / Half-Tree");
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(Lds, te0);
  const int q = threadIdx.x & 3;
  const uint32_t is3 = q == 3 ? ~0u : 0u;
  uint32_t ka[11], kd[11], hk;
  uint32_t kb[SCHEME == S_HT ? 1 : 11], ke[SCHEME == S_DCF ? 11 : 1];
  quad_keys<SCHEME>(c, q, ka, kd, kb, ke, hk);
  const Grp g = make_grp(a.group, c);
  const int n = a.in_bits, depth = SCHEME == S_HT ? n - 1 : n;
  const int64_t nk = a.n, nquads = ((int64_t)gridDim.x * blockDim.x) >> 2;
  uint32_t* __restrict__ ys = reinterpret_cast<uint32_t*>(a.ys);
  for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; k < nk; k += nquads) {
    const X128 x = load_x(a.xs, a.x_bytes, k);
    uint32_t st = reinterpret_cast<const uint32_t*>(a.seeds + k)[q];
    st = (st & ~(is3 & 1u)) | (is3 & (uint32_t)a.party);
    uint64_t trb = 0, trbh = 0;
    if constexpr (SCHEME == S_DPF) { trb = a.tr_bits[k]; trbh = n > 64 ? a.tr_bits[nk + k] : 0ull; }
    Acc v;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
      const uint32_t m = bit_mask(x, n - 1 - i);
      const uint32_t w = reinterpret_cast<const uint32_t*>(a.cw_s + (int64_t)i * nk + k)[q];
      if constexpr (SCHEME == S_DCF) {
        const uint4 cv = a.cw_v[(int64_t)i * nk + k];
        const uint32_t vcw[3] = {cv.x, cv.y, cv.z};
        quad_level_cw<SCHEME>(L, g, w, cv.w, vcw, m, is3, ka, kd, kb, ke, hk, st, v);
      } else {
        const uint32_t tr = SCHEME == S_DPF ? (uint32_t)(((i < 64 ? trb : trbh) >> (i & 63)) & 1ull) : 0u;
        quad_level_cw<SCHEME>(L, g, w, tr, nullptr, m, is3, ka, kd, kb, ke, hk, st, v);
      }
    }
    const uint32_t node[4] = {quad_get<FSS_QP(0, 0, 0, 0)>(st), quad_get<FSS_QP(1, 1, 1, 1)>(st), quad_get<FSS_QP(2, 2, 2, 2)>(st),
                              quad_get<FSS_QP(3, 3, 3, 3)>(st)};
    uint4 y;
    if constexpr (SCHEME == S_HT) {
      HtWalk w;
      w.resume(make_uint4(node[0], node[1], node[2], node[3]), x);
      const uint4 t = a.cw_s[(int64_t)(n - 1) * nk + k], u = a.out_cw[k];
      const uint32_t cw[4] = {t.x, t.y, t.z, t.w}, oc[4] = {u.x, u.y, u.z, u.w};
      y = w.finish(L, c, g, a.party, cw, a.extra[k], oc);
    } else {
      const uint4 u = a.out_cw[k];
      const uint32_t oc[4] = {u.x, u.y, u.z, u.w};
      if constexpr (SCHEME == S_DPF) {
        DpfWalk w;
        w.resume(make_uint4(node[0], node[1], node[2], node[3]), x);
        y = w.finish(g, a.party, oc);
      } else {
        DcfWalk w;
        w.resume(make_uint4(node[0], node[1], node[2], node[3]), make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]), x);
        y = w.finish(g, a.party, oc);
      }
    }
    ys[4 * k + q] = q == 0 ? y.x : (q == 1 ? y.y : (q == 2 ? y.z : y.w));
  }
}

}  // namespace fssamd
