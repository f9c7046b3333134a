// Verifiable DPF (include/fss/vdpf.cuh): the DPF walk plus one keyed-BLAKE3 XorHash per evaluation.
#pragma once
#include "fss_dev.hpp"
#include "fss_gen.hpp"
#include "fss_point.hpp"

namespace fssamd {

struct VdpfArgs {
  int in_bits, party, group, x_bytes;
  int64_t n;
  const void* xs;
  uint4* ys;
  uint4* pis;               // 4 per evaluation: the corrected verification hash (pi tilde)
  // independent keys (level-major)
  const uint4* seeds;
  const uint4* cw_s;
  const uint64_t* tr_bits;
  const uint4* cs;          // 4 per key
  const uint4* ocws;
  int dyn;                  // WaveQ: chunks of 64 * ILP evaluations from the workgroup's LDS counter
  int tr32;                 // tr_bits is the reference's uint32_t `extra` (in_bits <= 32)
};

// fss::Vdpf::Eval (vdpf.cuh:204-257) for a batch; fss::gpu::VdpfEvalPointKernel (point_eval_gpu.cuh:254-308).
// SHARED: one key (k_seed, k_cws = in_bits raw slots, k_cs = 16 words, k_ocw = 4 words, all wave-uniform).
template <class LT, bool SHARED, int ILP, int BS>
__global__ __launch_bounds__(BS) void vdpf_points_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                         const uint32_t* __restrict__ k_seed,
                                                         const uint32_t* __restrict__ k_cws,
                                                         const uint32_t* __restrict__ k_cs,
                                                         const uint32_t* __restrict__ k_ocw, VdpfArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const int in_bits = a.in_bits, party = a.party;
  const Grp g = make_grp(a.group, c);
  const int64_t n = a.n;
  constexpr int64_t SJ = 64;
  WaveQ q(a.dyn, &waveq_ctr_, n, ILP);
  for (int64_t base; q.next(base);) {
    int64_t idx[ILP];
    DpfWalk w[ILP];
    uint64_t trb[ILP], trbh[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      int64_t e = base + (int64_t)j * SJ;
      idx[j] = e < n ? e : n - 1;
      uint32_t s0[4];
      if constexpr (SHARED) { s0[0] = k_seed[0]; s0[1] = k_seed[1]; s0[2] = k_seed[2]; s0[3] = k_seed[3]; }
      else { uint4 t = a.seeds[idx[j]]; s0[0] = t.x; s0[1] = t.y; s0[2] = t.z; s0[3] = t.w; }
      w[j].init(s0, party, load_x(a.xs, a.x_bytes, idx[j]));
      if constexpr (!SHARED) {
        trb[j] = load_tr(a.tr_bits, a.tr32, idx[j]);
        trbh[j] = in_bits > 64 ? a.tr_bits[n + idx[j]] : 0ull;
      }
    }
#pragma unroll 1
    for (int i = 0; i < in_bits; ++i) {
      CwDpf cw[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if constexpr (SHARED) {
          const uint32_t* p = k_cws + 8 * i;
          cw[j].x = p[0]; cw[j].y = p[1]; cw[j].z = p[2]; cw[j].wl = p[3]; cw[j].tr = p[4];
        } else {
          uint4 t = FSS_LD_STREAM(a.cw_s + (int64_t)i * n + idx[j]);
          cw[j].x = t.x; cw[j].y = t.y; cw[j].z = t.z; cw[j].wl = t.w;
          cw[j].tr = (uint32_t)((i >= 64 ? trbh[j] : trb[j]) >> (i & 63));
        }
      }
      if constexpr (ILP == 2) DpfWalk::level2(L, c, in_bits, i, w[0], cw[0], w[1], cw[1]);
      else w[0].level(L, c, in_bits, i, cw[0]);
    }
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      uint32_t oc[4];
      if constexpr (SHARED) { oc[0] = k_ocw[0]; oc[1] = k_ocw[1]; oc[2] = k_ocw[2]; oc[3] = k_ocw[3]; }
      else { uint4 t = a.ocws[idx[j]]; oc[0] = t.x; oc[1] = t.y; oc[2] = t.z; oc[3] = t.w; }
      const uint4 y = w[j].finish(g, party, oc);  // vdpf.cuh:241-246 = the DPF leaf
      // vdpf.cuh:248-256: pi~ = H(Pack(x), s) [^ cs when t]
      const uint32_t xb[4] = {(uint32_t)w[j].x.lo, (uint32_t)(w[j].x.lo >> 32), (uint32_t)w[j].x.hi, (uint32_t)(w[j].x.hi >> 32)};
      const uint32_t s[4] = {w[j].st[0], w[j].st[1], w[j].st[2], w[j].st[3] & ~1u};
      const uint32_t tm = lsb_mask(w[j].st[3]);
      uint32_t h[16];
      vdpf_xorhash(c, xb, s, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t cq[4];
        if constexpr (SHARED) { cq[0] = k_cs[4 * q]; cq[1] = k_cs[4 * q + 1]; cq[2] = k_cs[4 * q + 2]; cq[3] = k_cs[4 * q + 3]; }
        else { uint4 t = a.cs[4 * idx[j] + q]; cq[0] = t.x; cq[1] = t.y; cq[2] = t.z; cq[3] = t.w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) h[4 * q + r] ^= tm & cq[r];
      }
      if (base + (int64_t)j * SJ < n) {
        a.ys[idx[j]] = y;
#pragma unroll
        for (int q = 0; q < 4; ++q) a.pis[4 * idx[j] + q] = make_uint4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
      }
    }
  }
}

struct VdpfGenArgs {
  int in_bits, group, x_bytes;
  int64_t nkeys;
  const uint4* s0s;
  const void* alphas;
  const uint4* betas;
  uint4* cws;        // key-major, in_bits slots per key
  uint4* cw_s;       // level-major
  uint64_t* tr_bits;
  uint4* cs;         // 4 per key
  uint4* ocws;
  int32_t* retry;    // optional: 1 where t0 == t1 at the end (vdpf.cuh:177)
  int dyn;           // WaveQ: 64-key chunks from the workgroup's LDS counter
};

// fss::Vdpf::Gen (vdpf.cuh:103-189): the DPF levels of gen_kernel, then cs and the output correction word.
template <class LT, bool LM>
__global__ __launch_bounds__(1024) void vdpf_gen_kernel(const DevCtx K, const uint32_t* __restrict__ te0, VdpfGenArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const int n = a.in_bits;
  const Grp g = make_grp(a.group, c);
  WaveQ q(a.dyn, &waveq_ctr_, a.nkeys, 1);
  for (int64_t k; q.next(k);) {
    if (k >= a.nkeys) continue;
    uint32_t s0[4], s1[4];
    ld4(s0, a.s0s[2 * k]); ld4(s1, a.s0s[2 * k + 1]);
    s0[3] &= ~1u; s1[3] &= ~1u;
    uint32_t t0 = 0u, t1 = ~0u;
    const X128 alpha = load_x(a.alphas, a.x_bytes, k);
    uint4* slot = LM ? nullptr : a.cws + k * (int64_t)n * 2;
    unsigned long long tr_lo = 0ull, tr_hi = 0ull;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      const uint32_t am = bit_mask(alpha, n - 1 - i);
      uint32_t s0l[4], s0r[4], s1l[4], s1r[4], scw[4];
      prg_gen2(L, c, s0, s0l, s0r);
      prg_gen2(L, c, s1, s1l, s1r);
      const uint32_t t0l = s0l[3] & 1u, t0r = s0r[3] & 1u, t1l = s1l[3] & 1u, t1r = s1r[3] & 1u;
      s0l[3] &= ~1u; s0r[3] &= ~1u; s1l[3] &= ~1u; s1r[3] &= ~1u;
      const uint32_t ab = am & 1u;
#pragma unroll
      for (int q = 0; q < 4; ++q) scw[q] = am ? (s0l[q] ^ s1l[q]) : (s0r[q] ^ s1r[q]);
      const uint32_t tl_cw = t0l ^ t1l ^ ab ^ 1u, tr_cw = t0r ^ t1r ^ ab;
      const uint32_t keep_cw = am ? tr_cw : tl_cw;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t k0 = am ? s0r[q] : s0l[q], k1 = am ? s1r[q] : s1l[q];
        s0[q] = k0 ^ (t0 & scw[q]);
        s1[q] = k1 ^ (t1 & scw[q]);
      }
      {
        uint32_t k0 = am ? t0r : t0l, k1 = am ? t1r : t1l;
        t0 = 0u - ((k0 ^ ((t0 & 1u) & keep_cw)) & 1u);
        t1 = 0u - ((k1 ^ ((t1 & 1u) & keep_cw)) & 1u);
      }
      const uint4 s_slot = make_uint4(scw[0], scw[1], scw[2], (scw[3] & ~1u) | tl_cw);
      if (LM) {
        a.cw_s[(int64_t)i * a.nkeys + k] = s_slot;
        if (i < 64) tr_lo |= (unsigned long long)tr_cw << i;
        else tr_hi |= (unsigned long long)tr_cw << (i - 64);
      } else {
        slot[2 * i] = s_slot;
        slot[2 * i + 1] = make_uint4(tr_cw, 0, 0, 0);
      }
    }
    if (LM) {
      a.tr_bits[k] = tr_lo;
      if (n > 64) a.tr_bits[a.nkeys + k] = tr_hi;
    }
    // cs = H(a, s0) ^ H(a, s1)   (vdpf.cuh:170-174)
    const uint32_t ab4[4] = {(uint32_t)alpha.lo, (uint32_t)(alpha.lo >> 32), (uint32_t)alpha.hi, (uint32_t)(alpha.hi >> 32)};
    uint32_t h0[16], h1[16];
    vdpf_xorhash(c, ab4, s0, h0);
    vdpf_xorhash(c, ab4, s1, h1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a.cs[4 * k + q] = make_uint4(h0[4 * q] ^ h1[4 * q], h0[4 * q + 1] ^ h1[4 * q + 1], h0[4 * q + 2] ^ h1[4 * q + 2],
                                   h0[4 * q + 3] ^ h1[4 * q + 3]);
    if (a.retry) a.retry[k] = (t0 == t1) ? 1 : 0;
    uint32_t beta[4];
    ld4(beta, a.betas ? a.betas[k] : make_uint4(0, 0, 0, 0));
    beta[3] &= ~1u;
    Acc y;  // beta - s0 + s1, negated when t1 (vdpf.cuh:180-182)
    y.add(g, beta); y.sub(g, s0); y.add(g, s1);
    a.ocws[k] = y.finish(g, t1 != 0u);
  }
}

// fss::Vdpf::Prove (vdpf.cuh:268-278).  CHAIN = false: one evaluation per key, pi[k] = Prove({pi~[k]}, cs[k]),
// one key per lane.  CHAIN = true: the sequential fold of n evaluations of ONE key, a single lane (each
// step hashes the previous pi: there is no parallelism to use).
template <bool CHAIN>
__global__ __launch_bounds__(256) void vdpf_prove_kernel(const DevCtx K, const uint4* __restrict__ pis,
                                                         const uint4* __restrict__ cs, int64_t n, uint4* __restrict__ out) {
  const DevCtx* c = first_kernarg(K);
  auto ld = [](uint32_t* d, uint4 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
  if constexpr (CHAIN) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t pi[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) ld(pi + 4 * q, cs[q]);
    for (int64_t i = 0; i < n; ++i) {
      uint32_t m[16], h[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) ld(m + 4 * q, pis[4 * i + q]);
#pragma unroll
      for (int q = 0; q < 16; ++q) m[q] ^= pi[q];
      vdpf_hash(c, m, h);
#pragma unroll
      for (int q = 0; q < 8; ++q) pi[q] ^= h[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = make_uint4(pi[4 * q], pi[4 * q + 1], pi[4 * q + 2], pi[4 * q + 3]);
  } else {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
      uint32_t pi[16], m[16], h[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { ld(pi + 4 * q, cs[4 * k + q]); ld(m + 4 * q, pis[4 * k + q]); }
#pragma unroll
      for (int q = 0; q < 16; ++q) m[q] ^= pi[q];
      vdpf_hash(c, m, h);
#pragma unroll
      for (int q = 0; q < 8; ++q) pi[q] ^= h[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) out[4 * k + q] = make_uint4(pi[4 * q], pi[4 * q + 1], pi[4 * q + 2], pi[4 * q + 3]);
    }
  }
}

}  // namespace fssamd
