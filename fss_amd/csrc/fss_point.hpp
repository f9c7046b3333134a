// Per-point PRG-tree walks (one evaluation per lane, ILP evaluations per thread).
//
// Only the child the walk takes is computed: Aes128Mmo output i is
// AES_{key_i}(s) ^ s independently per key (include/fss/prg/aes128_mmo.cuh:79-89),
// and Eval keeps only (sl,tl) or (sr,tr) (include/fss/dpf.cuh:198-204), so the
// per-lane round-key choice halves the AES work with bit-identical outputs.
#pragma once
#include "fss_dev.hpp"

// Correction words of independent keys are read exactly once: FSS_NT_CW = 1 (default) marks those loads
// non-temporal (+1 % on the 2^24-key walks, which stream 1.7 TB/s and run power-limited).
#ifndef FSS_NT_CW
#define FSS_NT_CW 1
#endif
#if FSS_NT_CW
typedef uint32_t fss_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 fss_ld_stream(const uint4* p) {
  fss_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fss_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
#define FSS_LD_STREAM(p) fss_ld_stream(p)
#else
#define FSS_LD_STREAM(p) (*(p))
#endif

namespace fssamd {

struct EvalArgs {
  int in_bits;
  int party;
  int group;
  int x_bytes;
  int64_t n;        // evaluations
  const void* xs;   // uint32 or uint64 per evaluation
  void* ys;         // uint4 per evaluation (Grotto: uint8)
  // independent keys, level-major
  const uint4* seeds;
  const uint4* cw_s;
  const uint4* cw_v;        // DCF
  const uint64_t* tr_bits;  // DPF / Grotto
  const uint4* out_cw;      // DPF / DCF: cws[in_bits]; Half-Tree: ocw per key
  const uint32_t* extra;    // Half-Tree
  // Shared-prefix start (one shared key only): the 2^start_level nodes of tree level
  // start_level, expanded once per launch; evaluation i starts from node xs[i] >> (in_bits - start_level)
  // instead of the root.  start_level = 0 / pre_nodes = nullptr: plain root-to-leaf walk.
  int start_level;
  const uint4* pre_nodes;
  const uint4* pre_v;       // DCF: running group value of every node
  int dyn;                  // waves take chunks from the workgroup's LDS counter (WaveQ, fss_dev.hpp)
  int prio_slack;           // rank_prio: levels a wave may be ahead of another before it yields
  int prio_every;           // rank_prio: exchange when (levels done & prio_every) == 0
  int rot_prio;             // wave priority per level: 1 = the waves of a SIMD take turns (rotate_prio), 2 = by progress (rank_prio)
  int tr32;                 // tr_bits is the reference's uint32_t `extra` array (point_eval_gpu.cuh:57-62; in_bits <= 32)
  uint64_t* stamps;         // FSS_DIAG_STAMPS builds only (tools/keys_clock.py, bench.py clk): 4 x uint64 per workgroup; unused otherwise
  uint32_t* gq;             // WaveQ's launch-wide tail: {counter, waves done} of this stream, or nullptr
  int64_t static_chunks;    // chunks (of the one-per-lane region) dealt out per workgroup before the launch-wide tail starts
  int64_t n_body;           // two-per-lane kernels: evaluations [0, n_body) run two per lane, the rest one (a multiple of 128, or == n: two per lane throughout)
  int tail_on_body;         // two-per-lane kernels without a one-per-lane region: the launch-wide tail deals two-per-lane chunks
};

// Diagnostic build (-DFSS_DIAG_STAMPS=1, a separate library: in the product no stamp executes): lane 0 of every
// workgroup records s_memtime (shader clock) and s_memrealtime (100 MHz) around the walk loop, so that
// in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
#ifndef FSS_DIAG_STAMPS
#define FSS_DIAG_STAMPS 0
#endif
// records of the context's stamp buffer (one per wave of a launch; 256 CUs x 2 workgroups x 16 waves fit twice)
constexpr int kDiagStampRecords = 16384;

// tr bits of key i: uint64_t per key, or the reference's 32-bit packing (wave-uniform choice)
__device__ __forceinline__ uint64_t load_tr(const uint64_t* tr_bits, int tr32, int64_t i) {
  return tr32 ? (uint64_t)reinterpret_cast<const uint32_t*>(tr_bits)[i] : tr_bits[i];
}

// An evaluation point / alpha of up to 128 bits (In = uint32_t, uint64_t or __uint128_t in the reference).
struct X128 {
  uint64_t lo, hi;
};
__device__ __forceinline__ uint32_t bit_mask(uint64_t x, int pos) {  // 0 or ~0, pos < 64
  uint32_t w = pos >= 32 ? (uint32_t)(x >> 32) : (uint32_t)x;
  return (uint32_t)__builtin_amdgcn_sbfe(w, pos & 31, 1);
}
__device__ __forceinline__ uint32_t bit_mask(const X128& x, int pos) {  // pos < 128 (wave-uniform)
  return bit_mask(pos >= 64 ? x.hi : x.lo, pos & 63);
}
__device__ __forceinline__ uint32_t lsb_mask(uint32_t w) { return (uint32_t)__builtin_amdgcn_sbfe(w, 0, 1); }

__device__ __forceinline__ X128 load_x(const void* xs, int x_bytes, int64_t i) {
  X128 x;
  x.hi = 0;
  if (x_bytes == 1) x.lo = reinterpret_cast<const uint8_t*>(xs)[i];
  else if (x_bytes == 2) x.lo = reinterpret_cast<const uint16_t*>(xs)[i];
  else if (x_bytes == 4) x.lo = reinterpret_cast<const uint32_t*>(xs)[i];
  else if (x_bytes == 8) x.lo = reinterpret_cast<const uint64_t*>(xs)[i];
  else { x.lo = reinterpret_cast<const uint64_t*>(xs)[2 * i]; x.hi = reinterpret_cast<const uint64_t*>(xs)[2 * i + 1]; }
  return x;
}

// Correction word of one level as the walk consumes it.
struct CwDpf {
  uint32_t x, y, z;
  uint32_t wl;  // s.w with tl_cw in bit 0
  uint32_t tr;  // tr_cw in bit 0
};

// ---------------------------------------------------------------- DPF (dpf.cuh:170-214)
struct DpfWalk {
  uint32_t st[4];  // seed with the control bit t packed into bit 0 of w (dpf.cuh:233-235)
  X128 x;

  __device__ __forceinline__ void init(const uint32_t s0[4], int party, X128 x_) {
    st[0] = s0[0]; st[1] = s0[1]; st[2] = s0[2];
    st[3] = (s0[3] & ~1u) | (uint32_t)party;  // dpf.cuh:171-173
    x = x_;
  }
  __device__ __forceinline__ void resume(uint4 node, X128 x_) {
    st[0] = node.x; st[1] = node.y; st[2] = node.z; st[3] = node.w;
    x = x_;
  }
  template <class LT>
  __device__ __forceinline__ void level(const LT& L, const DevCtx* __restrict__ c, int in_bits, int i, const CwDpf& cw) {
    uint32_t m = bit_mask(x, in_bits - 1 - i);  // dpf.cuh:196
    uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
    uint32_t o[4];
    prg_take2(L, c, m, in, o);  // dpf.cuh:182, taken child only
    apply(m, o, cw);
  }
  // Two walks, one level, the two AES blocks interleaved (aes_mmo2).
  template <class LT>
  static __device__ __forceinline__ void level2(const LT& L, const DevCtx* __restrict__ c, int in_bits, int i,
                                                DpfWalk& a, const CwDpf& cwa, DpfWalk& b, const CwDpf& cwb) {
    uint32_t ma = bit_mask(a.x, in_bits - 1 - i), mb = bit_mask(b.x, in_bits - 1 - i);
    uint32_t ia[4] = {a.st[0], a.st[1], a.st[2], a.st[3] & ~1u}, ib[4] = {b.st[0], b.st[1], b.st[2], b.st[3] & ~1u};
    uint32_t oa[4], ob[4];
    prg_take2x2(L, c, ma, mb, ia, ib, oa, ob);
    a.apply(ma, oa, cwa);
    b.apply(mb, ob, cwb);
  }
  __device__ __forceinline__ void apply(uint32_t m, const uint32_t o[4], const CwDpf& cw) {
    uint32_t tm = lsb_mask(st[3]);
    // bit 0 of o[3] is tl/tr; bit 0 of the word below is tl_cw/tr_cw: one XOR does
    // both the seed and the control-bit correction (dpf.cuh:189-194).
    uint32_t cww = cw.wl ^ (m & ((cw.wl ^ cw.tr) & 1u));
    st[0] = and_xor(tm, cw.x, o[0]);
    st[1] = and_xor(tm, cw.y, o[1]);
    st[2] = and_xor(tm, cw.z, o[2]);
    st[3] = and_xor(tm, cww, o[3]);
  }
  __device__ __forceinline__ uint4 finish(const Grp& g, int party, const uint32_t out_cw[4]) const {
    Acc y;  // dpf.cuh:207-213
    uint32_t s[4] = {st[0], st[1], st[2], st[3] & ~1u};
    y.add(g, s);
    y.add_if(g, lsb_mask(st[3]), out_cw);
    return y.finish(g, party != 0);
  }
};

// ---------------------------------------------------------------- DCF (dcf.cuh:205-276)
struct CwDcf {
  uint32_t s[4];  // s.w has tl_cw in bit 0
  uint32_t v[4];  // v.w has tr_cw in bit 0
};

struct DcfWalk {
  uint32_t st[4];
  X128 x;
  Acc v;  // dcf.cuh:208, zero

  __device__ __forceinline__ void init(const uint32_t s0[4], int party, X128 x_) {
    st[0] = s0[0]; st[1] = s0[1]; st[2] = s0[2];
    st[3] = (s0[3] & ~1u) | (uint32_t)party;
    x = x_;
  }
  __device__ __forceinline__ void resume(uint4 node, uint4 val, X128 x_) {
    st[0] = node.x; st[1] = node.y; st[2] = node.z; st[3] = node.w;
    v.v[0] = val.x; v.v[1] = val.y; v.v[2] = val.z; v.v[3] = val.w;
    x = x_;
  }
  template <class LT>
  __device__ __forceinline__ void level(const LT& L, const DevCtx* __restrict__ c, const Grp& g, int in_bits, int i,
                                        const CwDcf& cw) {
    uint32_t m = bit_mask(x, in_bits - 1 - i);
    uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
    uint32_t os[4], ov[4];
    // sl or sr, and vl or vr (dcf.cuh:223); AES: keys 0/2 and 1/3, two interleaved blocks
    prg_take4(L, c, m, in, os, ov);
    uint32_t tm = lsb_mask(st[3]);
    ov[3] &= ~1u;  // dcf.cuh:227,232
    v.add(g, ov);  // dcf.cuh:244-252 (signs applied once in finish)
    uint32_t vcw[4] = {cw.v[0], cw.v[1], cw.v[2], cw.v[3] & ~1u};
    v.add_if(g, tm, vcw);
    uint32_t cww = cw.s[3] ^ (m & ((cw.s[3] ^ cw.v[3]) & 1u));
    st[0] = and_xor(tm, cw.s[0], os[0]);
    st[1] = and_xor(tm, cw.s[1], os[1]);
    st[2] = and_xor(tm, cw.s[2], os[2]);
    st[3] = and_xor(tm, cww, os[3]);
  }
  __device__ __forceinline__ uint4 finish(const Grp& g, int party, const uint32_t out_cw[4]) {
    uint32_t s[4] = {st[0], st[1], st[2], st[3] & ~1u};  // dcf.cuh:263-273
    v.add(g, s);
    v.add_if(g, lsb_mask(st[3]), out_cw);
    return v.finish(g, party != 0);
  }
};

// ---------------------------------------------------------------- Half-Tree DPF (half_tree_dpf.cuh:187-231)
struct HtWalk {
  uint32_t node[4];  // control bit kept in bit 0 of w (half_tree_dpf.cuh:188)
  X128 x;

  __device__ __forceinline__ void init(const uint32_t s0[4], int party, X128 x_) {
    node[0] = s0[0]; node[1] = s0[1]; node[2] = s0[2];
    node[3] = (s0[3] & ~1u) | (uint32_t)party;
    x = x_;
  }
  __device__ __forceinline__ void resume(uint4 n, X128 x_) {
    node[0] = n.x; node[1] = n.y; node[2] = n.z; node[3] = n.w;
    x = x_;
  }
  template <class LT>
  static __device__ __forceinline__ void level2(const LT& L, const DevCtx* __restrict__ c, int in_bits, int i,
                                                HtWalk& a, const uint32_t cwa[4], HtWalk& b, const uint32_t cwb[4]) {
    uint32_t ma = bit_mask(a.x, in_bits - 1 - i), mb = bit_mask(b.x, in_bits - 1 - i);
    uint32_t ta = lsb_mask(a.node[3]), tb = lsb_mask(b.node[3]);
    uint32_t ia[4], ib[4], ha[4], hb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ia[j] = a.node[j] ^ c->hash_key[j]; ib[j] = b.node[j] ^ c->hash_key[j]; }
    prg_hash2(L, c, ia, ib, ha, hb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a.node[j] = and_xor(ma, a.node[j], and_xor(ta, cwa[j], ha[j]));
      b.node[j] = and_xor(mb, b.node[j], and_xor(tb, cwb[j], hb[j]));
    }
  }
  // levels 0 .. in_bits-2
  template <class LT>
  __device__ __forceinline__ void level(const LT& L, const DevCtx* __restrict__ c, int in_bits, int i, const uint32_t cw[4]) {
    uint32_t m = bit_mask(x, in_bits - 1 - i);
    uint32_t tm = lsb_mask(node[3]);
    uint32_t in[4] = {node[0] ^ c->hash_key[0], node[1] ^ c->hash_key[1], node[2] ^ c->hash_key[2],
                      node[3] ^ c->hash_key[3]};
    uint32_t h[4];
    prg_hash(L, c, in, h);  // prg.Gen(hash_key ^ node)[0], :195
#pragma unroll
    for (int j = 0; j < 4; ++j) node[j] = and_xor(m, node[j], and_xor(tm, cw[j], h[j]));  // :197-201
  }
  // last level + output conversion, :205-229
  template <class LT>
  __device__ __forceinline__ uint4 finish(const LT& L, const DevCtx* __restrict__ c, const Grp& g, int party, const uint32_t cw[4],
                                          uint32_t extra, const uint32_t ocw[4]) const {
    uint32_t xn = (uint32_t)x.lo & 1u;
    uint32_t tm = lsb_mask(node[3]);
    uint32_t in[4] = {node[0] ^ c->hash_key[0], node[1] ^ c->hash_key[1], node[2] ^ c->hash_key[2],
                      ((node[3] & ~1u) | xn) ^ c->hash_key[3]};
    uint32_t h[4];
    prg_hash(L, c, in, h);
    uint32_t lcw = xn ? (extra & 1u) : (cw[3] & 1u);
    uint32_t low = (h[3] ^ (tm & lcw)) & 1u;
    uint32_t high[4] = {and_xor(tm, cw[0], h[0]), and_xor(tm, cw[1], h[1]), and_xor(tm, cw[2], h[2]),
                        and_xor(tm, cw[3], h[3]) & ~1u};
    Acc y;
    y.add(g, high);
    y.add_if(g, 0u - low, ocw);
    return y.finish(g, party != 0);
  }
};

// ---------------------------------------------------------------- kernels
// SHARED: every evaluation uses the same key (correction words are wave-uniform,
// fetched with scalar loads from the raw key-major slots).  Otherwise evaluation k
// uses key k from the level-major arrays (one coalesced 16-byte load per lane and
// level, as include/fss/point_eval_gpu.cuh:144).
//
// K (by value), and the shared key's seed / cws / ocw as __restrict__ kernel
// arguments, are wave-uniform and provably not clobbered by the ys stores, so they
// are read with scalar loads.
// PREFIX: start every walk from the level-`start_level` node table (shared-prefix entry points);
// a separate instantiation so profiles tell the two paths apart.
//
// ILPT = 2, "two, then one": evaluations [0, n_body) are walked two per lane -- two interleaved AES streams (aes_mmo2)
// keep the LDS busier than one: the stamped wave spends 2.5 % of the launch above the lookup floor instead of 4.3 % --
// and the rest, [n_body, n), one per lane, so that the launch ends on chunks half as long (the idle tail of waves that
// are done was 125-150 us of a 4.8 ms launch with two per lane throughout, 65-70 us with one).  Measured round 3:
// profiles/r3/points_2e24_ilp_raw.txt (each alone), DESIGN.md 4.4 (combined).  n_body = n: two per lane throughout.
template <class LT, int SCHEME, bool SHARED, int ILPT, int BS, bool PREFIX = false>
__global__ __launch_bounds__(BS) void eval_points_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                         const uint32_t* __restrict__ k_seed,  // uint32[4]
                                                         const uint32_t* __restrict__ k_cws,   // raw Cw slots
                                                         const uint32_t* __restrict__ k_ocw,   // Half-Tree
                                                         EvalArgs a) {
  const DevCtx* c = first_kernarg(K);
#if FSS_DIAG_STAMPS
  const uint64_t stamp_re = __builtin_amdgcn_s_memrealtime();  // the wave's first instruction: before the table fill
#endif
  FSS_PROGRESS_PROLOGUE();
  __shared__ uint32_t waveq_ctr2_;  // chunk counter of the one-per-lane region (zeroed before FSS_WAVEQ_PROLOGUE's barrier)
  if (threadIdx.x == 0) waveq_ctr2_ = 0;
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  uint32_t my_prog = 0;

  const int in_bits = a.in_bits, party = a.party;
  const Grp g = make_grp(a.group, c);
  const int lvl0 = PREFIX ? a.start_level : 0;
  const int64_t n = a.n;
  uint4* __restrict__ ys = reinterpret_cast<uint4*>(a.ys);

  constexpr int64_t SJ = 64;  // a lane's ILP evaluations are 64 apart: a chunk is 64 * ILP consecutive evaluations
#if FSS_DIAG_STAMPS
  const uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // the points of one chunk (64 * ILP consecutive evaluations from `base` on, ILP per lane)
  auto load_chunk_x = [&](auto ilp_tag, const int64_t base, X128* xv) {
    constexpr int ILP = decltype(ilp_tag)::value;
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      const int64_t e = base + (int64_t)j * SJ;
      xv[j] = load_x(a.xs, a.x_bytes, e < n ? e : n - 1);
    }
  };
  // Level at which a shared-prefix walk requests the NEXT chunk's start nodes (mid_hook): half way down, so the
  // points those nodes are indexed by (requested before this chunk started) have had half a walk to arrive and the
  // nodes have the other half.
  const int mid_level = lvl0 + ((in_bits - lvl0) >> 1);
  // one chunk; its points arrive in xv and, for the shared-prefix walks, its start nodes in pn / pv (run_region)
  auto walk_chunk = [&](auto ilp_tag, const int64_t base, const X128* xv, const uint4* pn, const uint4* pv, auto&& mid_hook) {
    constexpr int ILP = decltype(ilp_tag)::value;
    (void)pn; (void)pv; (void)mid_hook;
    int64_t idx[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      int64_t e = base + (int64_t)j * SJ;
      idx[j] = e < n ? e : n - 1;  // clamp: keeps the tail lanes converged, the store is masked
    }

    if constexpr (SCHEME == S_DPF) {
      DpfWalk w[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if constexpr (PREFIX) {
          w[j].resume(pn[j], xv[j]);  // start_level >= 1 (make_prefix_plan): the root never starts a shared-prefix walk
        } else {
          uint32_t s0[4];
          if constexpr (SHARED) { s0[0] = k_seed[0]; s0[1] = k_seed[1]; s0[2] = k_seed[2]; s0[3] = k_seed[3]; }
          else { uint4 t = a.seeds[idx[j]]; s0[0] = t.x; s0[1] = t.y; s0[2] = t.z; s0[3] = t.w; }
          w[j].init(s0, party, xv[j]);
        }
      }
      uint64_t trb[ILP], trbh[ILP];  // tr bits of levels 0..63 and 64..127 (second plane at + n)
      if constexpr (!SHARED) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
          trb[j] = load_tr(a.tr_bits, a.tr32, idx[j]);
          trbh[j] = in_bits > 64 ? a.tr_bits[n + idx[j]] : 0ull;
        }
      }
      if constexpr (PREFIX) { if (mid_level >= in_bits) mid_hook(); }  // no level left to walk: no iteration below
#pragma unroll 1
      for (int i = lvl0; i < in_bits; ++i) {
        if (a.rot_prio == 1) rotate_prio(i);
        else if (a.rot_prio == 2) rank_prio(wave_prog_, my_prog, (uint32_t)a.prio_slack, (uint32_t)a.prio_every);
        if constexpr (PREFIX) { if (i == mid_level) mid_hook(); }
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
        if constexpr (SHARED) { const uint32_t* p = k_cws + 8 * in_bits; oc[0] = p[0]; oc[1] = p[1]; oc[2] = p[2]; oc[3] = p[3]; }
        else { uint4 t = a.out_cw[idx[j]]; oc[0] = t.x; oc[1] = t.y; oc[2] = t.z; oc[3] = t.w; }
        uint4 y = w[j].finish(g, party, oc);
        if (base + (int64_t)j * SJ < n) ys[idx[j]] = y;
      }
    } else if constexpr (SCHEME == S_DCF) {
      DcfWalk w[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if constexpr (PREFIX) {
          w[j].resume(pn[j], pv[j], xv[j]);
        } else {
          uint32_t s0[4];
          if constexpr (SHARED) { s0[0] = k_seed[0]; s0[1] = k_seed[1]; s0[2] = k_seed[2]; s0[3] = k_seed[3]; }
          else { uint4 t = a.seeds[idx[j]]; s0[0] = t.x; s0[1] = t.y; s0[2] = t.z; s0[3] = t.w; }
          w[j].init(s0, party, xv[j]);
        }
      }
      if constexpr (PREFIX) { if (mid_level >= in_bits) mid_hook(); }  // no level left to walk: no iteration below
#pragma unroll 1
      for (int i = lvl0; i < in_bits; ++i) {
        if (a.rot_prio == 1) rotate_prio(i);
        else if (a.rot_prio == 2) rank_prio(wave_prog_, my_prog, (uint32_t)a.prio_slack, (uint32_t)a.prio_every);
        if constexpr (PREFIX) { if (i == mid_level) mid_hook(); }
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
          CwDcf cw;
          if constexpr (SHARED) {
            const uint32_t* p = k_cws + 8 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) { cw.s[q] = p[q]; cw.v[q] = p[4 + q]; }
          } else {
            uint4 t = FSS_LD_STREAM(a.cw_s + (int64_t)i * n + idx[j]);
            uint4 u = FSS_LD_STREAM(a.cw_v + (int64_t)i * n + idx[j]);
            cw.s[0] = t.x; cw.s[1] = t.y; cw.s[2] = t.z; cw.s[3] = t.w;
            cw.v[0] = u.x; cw.v[1] = u.y; cw.v[2] = u.z; cw.v[3] = u.w;
          }
          w[j].level(L, c, g, in_bits, i, cw);
        }
      }
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        uint32_t oc[4];
        if constexpr (SHARED) { const uint32_t* p = k_cws + 8 * in_bits + 4; oc[0] = p[0]; oc[1] = p[1]; oc[2] = p[2]; oc[3] = p[3]; }
        else { uint4 t = a.out_cw[idx[j]]; oc[0] = t.x; oc[1] = t.y; oc[2] = t.z; oc[3] = t.w; }
        uint4 y = w[j].finish(g, party, oc);
        if (base + (int64_t)j * SJ < n) ys[idx[j]] = y;
      }
    } else if constexpr (SCHEME == S_HT) {
      HtWalk w[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if constexpr (PREFIX) {
          w[j].resume(pn[j], xv[j]);  // start_level >= 1 (make_prefix_plan): the root never starts a shared-prefix walk
        } else {
          uint32_t s0[4];
          if constexpr (SHARED) { s0[0] = k_seed[0]; s0[1] = k_seed[1]; s0[2] = k_seed[2]; s0[3] = k_seed[3]; }
          else { uint4 t = a.seeds[idx[j]]; s0[0] = t.x; s0[1] = t.y; s0[2] = t.z; s0[3] = t.w; }
          w[j].init(s0, party, xv[j]);
        }
      }
      // (a Half-Tree key with one walked level left has no loop iteration: its hook runs before the last level)
      if constexpr (PREFIX) { if (mid_level >= in_bits - 1) mid_hook(); }
#pragma unroll 1
      for (int i = lvl0; i < in_bits - 1; ++i) {
        if (a.rot_prio == 1) rotate_prio(i);
        else if (a.rot_prio == 2) rank_prio(wave_prog_, my_prog, (uint32_t)a.prio_slack, (uint32_t)a.prio_every);
        if constexpr (PREFIX) { if (i == mid_level) mid_hook(); }
        uint32_t cw[ILP][4];
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
          if constexpr (SHARED) { const uint32_t* p = k_cws + 8 * i; cw[j][0] = p[0]; cw[j][1] = p[1]; cw[j][2] = p[2]; cw[j][3] = p[3]; }
          else { uint4 t = FSS_LD_STREAM(a.cw_s + (int64_t)i * n + idx[j]); cw[j][0] = t.x; cw[j][1] = t.y; cw[j][2] = t.z; cw[j][3] = t.w; }
        }
        if constexpr (ILP == 2) HtWalk::level2(L, c, in_bits, i, w[0], cw[0], w[1], cw[1]);
        else w[0].level(L, c, in_bits, i, cw[0]);
      }
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        uint32_t cw[4], oc[4], extra;
        if constexpr (SHARED) {
          const uint32_t* p = k_cws + 8 * (in_bits - 1);
          cw[0] = p[0]; cw[1] = p[1]; cw[2] = p[2]; cw[3] = p[3]; extra = p[4];
          oc[0] = k_ocw[0]; oc[1] = k_ocw[1]; oc[2] = k_ocw[2]; oc[3] = k_ocw[3];
        } else {
          uint4 t = a.cw_s[(int64_t)(in_bits - 1) * n + idx[j]];
          cw[0] = t.x; cw[1] = t.y; cw[2] = t.z; cw[3] = t.w;
          extra = a.extra[idx[j]];
          uint4 u = a.out_cw[idx[j]];
          oc[0] = u.x; oc[1] = u.y; oc[2] = u.z; oc[3] = u.w;
        }
        uint4 y = w[j].finish(L, c, g, party, cw, extra, oc);
        if (base + (int64_t)j * SJ < n) ys[idx[j]] = y;
      }
    }
  };
  // two per lane over [0, n_body): this workgroup's chunks through its LDS counter (or the wave's fixed share) ...
  const int64_t n_body = ILPT == 2 ? a.n_body : 0;
  // The chunks of one region.  The shared-key DPF walk requests each chunk's points a chunk ahead of their use
  // (WaveQ::peek): +0.8-1.0 % at n = 16 / 32 / 64 x 2^24 points, +0.3 % at 2^20, three interleaved rounds on one box; the
  // same form costs DCF, Half-Tree and every independent-key walk 0.2-0.4 % (their levels carry more live state), so
  // those load their points when the chunk starts (profiles/r4/prefetch_x_ab.txt).
  // The shared-prefix walks (PREFIX) go one step further: a chunk there is only in_bits - start_level levels long and
  // starts with a dependent chain -- point, then a 16-byte gather from the node table at an address the point decides --
  // so besides the points (a chunk ahead) the NEXT chunk's start nodes are requested half way down the current chunk
  // (mid_hook) and the gather's latency runs under the second half of the walk.
  constexpr bool kAheadX = SCHEME == S_DPF && SHARED;
  auto gather_nodes = [&](auto ilp_tag, const X128* xv, uint4* pn, uint4* pv) {
    constexpr int ILP = decltype(ilp_tag)::value;
    (void)pv;
    if constexpr (PREFIX) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        const uint64_t q = (xv[j].lo >> (in_bits - lvl0)) & ((1ull << lvl0) - 1);
        pn[j] = a.pre_nodes[q];
        if constexpr (SCHEME == S_DCF) pv[j] = a.pre_v[q];
      }
    }
  };
  auto run_region = [&](auto ilp_tag, WaveQ& wq, const int64_t off) {
    constexpr int ILP = decltype(ilp_tag)::value;
    auto no_hook = [] {};
    if constexpr (PREFIX) {
      X128 xc[ILP], xn[ILP];
      uint4 nc[ILP], nn[ILP], vc[ILP], vn[ILP];
      int64_t base, nb;
      if (!wq.peek(base)) return;
      load_chunk_x(ilp_tag, off + base, xc);
      gather_nodes(ilp_tag, xc, nc, vc);
      while (wq.next(base)) {
        const bool more = wq.peek(nb);
        if (more) load_chunk_x(ilp_tag, off + nb, xn);
        walk_chunk(ilp_tag, off + base, xc, nc, vc, [&] { if (more) gather_nodes(ilp_tag, xn, nn, vn); });
#pragma unroll
        for (int j = 0; j < ILP; ++j) { xc[j] = xn[j]; nc[j] = nn[j]; vc[j] = vn[j]; }
      }
    } else if constexpr (kAheadX) {
      X128 xc[ILP], xn[ILP];
      int64_t base, nb;
      if (!wq.peek(base)) return;
      load_chunk_x(ilp_tag, off + base, xc);
      while (wq.next(base)) {
        const bool more = wq.peek(nb);
        if (more) load_chunk_x(ilp_tag, off + nb, xn);
        walk_chunk(ilp_tag, off + base, xc, nullptr, nullptr, no_hook);
#pragma unroll
        for (int j = 0; j < ILP; ++j) xc[j] = xn[j];
      }
    } else {
      for (int64_t base; wq.next(base);) {
        X128 xc[ILP];
        load_chunk_x(ilp_tag, off + base, xc);
        walk_chunk(ilp_tag, off + base, xc, nullptr, nullptr, no_hook);
      }
    }
  };
  if constexpr (ILPT == 2) {
    WaveQ qb(a.dyn, &waveq_ctr_, n_body, 2, a.tail_on_body ? a.gq : nullptr, a.static_chunks);
    run_region(std::integral_constant<int, 2>{}, qb, 0);
    qb.done();
  }
  // ... then one per lane over [n_body, n): the first static_chunks chunks per workgroup, the rest launch-wide
  WaveQ q(a.dyn, ILPT == 2 ? &waveq_ctr2_ : &waveq_ctr_, n - n_body, 1, (ILPT == 2 && a.tail_on_body) ? nullptr : a.gq,
          a.static_chunks);
  run_region(std::integral_constant<int, 1>{}, q, n_body);
#if FSS_DIAG_STAMPS
  if ((threadIdx.x & 63) == 0 && a.stamps) {  // one record per wave, taken as the wave leaves its loop (before done()'s barrier)
    const int64_t rec = (int64_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
    uint64_t* o = a.stamps + 4 * (rec < kDiagStampRecords ? rec : 0);
    o[0] = stamp_t0; o[1] = stamp_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    a.stamps[4 * (int64_t)kDiagStampRecords + (rec < kDiagStampRecords ? rec : 0)] = stamp_re;  // second region: entry times
  }
#endif
  q.done();
}

// ---------------------------------------------------------------- Grotto DCF, path walk
// Share of 1[alpha <= x] for domains with no parity table (in_bits up to 64):
// XOR of the control bits of the left siblings where e = x + 1 has a 1 bit
// (the whole-domain case e = 2^n is the root's control bit).  Keys are DPF keys
// over Bytes (include/fss/grotto_dcf.cuh:63-67); the per-level expansion is
// ExpandTreeRec's (grotto_dcf.cuh:201-222).  Independent keys, level-major.
template <class LT, int BS>
__global__ __launch_bounds__(BS) void grotto_walk_kernel(const DevCtx K, const uint32_t* __restrict__ te0, EvalArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const int n = a.in_bits;
  uint8_t* __restrict__ ys = reinterpret_cast<uint8_t*>(a.ys);
  WaveQ q(a.dyn, &waveq_ctr_, a.n, 1, a.gq, a.static_chunks);
  for (int64_t k; q.next(k);) {
    if (k >= a.n) continue;
    const uint64_t x = load_x(a.xs, a.x_bytes, k).lo;
    const uint64_t e = x + 1;
    const bool whole = n == 64 ? (e == 0) : (e == (1ull << n));
    uint4 s4 = a.seeds[k];
    uint32_t st[4] = {s4.x, s4.y, s4.z, (s4.w & ~1u) | (uint32_t)a.party};
    const uint64_t trb = load_tr(a.tr_bits, a.tr32, k);
    uint32_t pi = 0;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      uint4 t = FSS_LD_STREAM(a.cw_s + (int64_t)i * a.n + k);
      const uint32_t m = bit_mask(e, n - 1 - i);
      uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
      const uint32_t tm = lsb_mask(st[3]);
      uint32_t l[4], r[4];
      prg_gen2(L, c, in, l, r);
      const uint32_t wl = t.w, wr = (t.w & ~1u) | ((uint32_t)(trb >> i) & 1u);
      l[0] = and_xor(tm, t.x, l[0]); l[1] = and_xor(tm, t.y, l[1]); l[2] = and_xor(tm, t.z, l[2]); l[3] = and_xor(tm, wl, l[3]);
      r[0] = and_xor(tm, t.x, r[0]); r[1] = and_xor(tm, t.y, r[1]); r[2] = and_xor(tm, t.z, r[2]); r[3] = and_xor(tm, wr, r[3]);
      pi ^= m & l[3] & 1u;
#pragma unroll
      for (int q = 0; q < 4; ++q) st[q] = m ? r[q] : l[q];
    }
    ys[k] = (uint8_t)(whole ? (uint32_t)a.party : pi);
  }
  q.done();
}

// The same walk with the siblings deferred (AES tables only).  A level needs the taken child of every lane
// but the left sibling -- and of it only the control bit -- just where e's bit is 1: computing both children
// everywhere is 2n blocks for the 1.5n the walk needs.  So a level computes the taken child only (per-lane
// key select, two keys per lane interleaved) and lanes whose bit is 1 append the parent seed to a per-wave
// queue in LDS; whenever 64 jobs are queued the wave runs one dense AES over them (key 0, every lane busy).
// The 64 control bits go back to their owners without touching the LDS again: a lane remembers WHICH queue
// positions it filled (a 64-bit mask per walk, `mine`), the dense block's bits are gathered into one wave-wide
// ballot, and every owner XORs the parity of (ballot & mine) into its accumulator -- XOR commutes, so the order
// of delivery is free.  (Round 4 delivered each bit with an LDS atomic XOR into a per-key word: 20.4 M bank-conflict
// cycles per launch from jobs of one owner meeting in one word, profiles/r4/pmc_workloads.json.)  The correction of
// the sibling's bit (t & cw.tl) is the owner's own.  A short queue is flushed at the end of the chunk (one partly
// filled AES per 128 keys x n levels); the lanes past its end run the block on stale seeds whose bits nobody owns.
template <int BS>
__global__ __launch_bounds__(BS) void grotto_walk_defer_kernel(const DevCtx K, const uint32_t* __restrict__ te0, EvalArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_PROGRESS_PROLOGUE();
  FSS_WAVEQ_PROLOGUE();
  uint32_t my_prog = 0;
  FSS_PRG_PROLOGUE(Lds, te0);
  constexpr int NW = BS / 64;
  __shared__ uint4 q_seed[NW][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#if FSS_DIAG_STAMPS
  const uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  q_seed[wv][lane] = make_uint4(0u, 0u, 0u, 0u);  // a short first flush reads every slot
  const int n = a.in_bits;
  uint8_t* __restrict__ ys = reinterpret_cast<uint8_t*>(a.ys);
  // one dense AES over the queued jobs of this wave: bit p of the result = control bit of the left child of job p
  auto flush = [&]() -> uint64_t {
    wave_lds_sync();
    const uint4 s4 = q_seed[wv][lane];
    uint32_t in[4] = {s4.x, s4.y, s4.z, s4.w};
    // key 0 = the left child (aes128_mmo.cuh:79-89); in[3]'s control bit is clear; control bit only
    const uint64_t bits = __ballot((prg_tbit_left(L, c, in) & 1u) != 0u);
    wave_lds_sync();
    return bits;
  };
  WaveQ q(a.dyn, &waveq_ctr_, a.n, 2, a.gq, a.static_chunks);
  for (int64_t base; q.next(base);) {
    int64_t k[2];
    bool valid[2], whole[2];
    uint64_t e[2], trb[2];
    uint64_t mine[2] = {0ull, 0ull};  // queue positions this lane filled for walk j since the last flush
    uint32_t st[2][4], pi[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      valid[j] = base + 64 * j < a.n;
      k[j] = valid[j] ? base + 64 * j : 0;  // idle lanes walk key 0 and drop the result
      e[j] = load_x(a.xs, a.x_bytes, k[j]).lo + 1;
      whole[j] = n == 64 ? (e[j] == 0) : (e[j] == (1ull << n));
      const uint4 s4 = a.seeds[k[j]];
      st[j][0] = s4.x; st[j][1] = s4.y; st[j][2] = s4.z; st[j][3] = (s4.w & ~1u) | (uint32_t)a.party;
      trb[j] = load_tr(a.tr_bits, a.tr32, k[j]);
    }
    auto deliver = [&](uint64_t bits) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        pi[j] ^= (uint32_t)__popcll(bits & mine[j]) & 1u;
        mine[j] = 0ull;
      }
    };
    int pending = 0;  // jobs queued by this wave (wave-uniform)
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      if (a.rot_prio == 2) rank_prio(wave_prog_, my_prog, 0u, (uint32_t)a.prio_every);
      uint4 t[2];
      uint32_t m[2], tm[2], in[2][4], o[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t[j] = FSS_LD_STREAM(a.cw_s + (int64_t)i * a.n + k[j]);
        m[j] = bit_mask(e[j], n - 1 - i);
        tm[j] = lsb_mask(st[j][3]);
        in[j][0] = st[j][0]; in[j][1] = st[j][1]; in[j][2] = st[j][2]; in[j][3] = st[j][3] & ~1u;
      }
      prg_take2x2(L, c, m[0], m[1], in[0], in[1], o[0], o[1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t wl = t[j].w, wr = (t[j].w & ~1u) | ((uint32_t)(trb[j] >> i) & 1u);
        st[j][0] = and_xor(tm[j], t[j].x, o[j][0]);
        st[j][1] = and_xor(tm[j], t[j].y, o[j][1]);
        st[j][2] = and_xor(tm[j], t[j].z, o[j][2]);
        st[j][3] = and_xor(tm[j], m[j] ? wr : wl, o[j][3]);
        pi[j] ^= m[j] & tm[j] & wl & 1u;  // the sibling's correction; its PRG bit arrives through the queue
        const bool has = valid[j] && m[j] != 0u;
        const uint64_t bal = __ballot(has);
        const int cnt = __builtin_amdgcn_readfirstlane(__popcll(bal));
        const int pos = pending + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (has && pos < 64) {
          q_seed[wv][pos] = make_uint4(in[j][0], in[j][1], in[j][2], in[j][3]);
          mine[j] |= 1ull << pos;
        }
        if (pending + cnt >= 64) {
          deliver(flush());
          if (has && pos >= 64) {
            q_seed[wv][pos - 64] = make_uint4(in[j][0], in[j][1], in[j][2], in[j][3]);
            mine[j] |= 1ull << (pos - 64);
          }
          pending += cnt - 64;
        } else {
          pending += cnt;
        }
      }
    }
    if (pending > 0) deliver(flush());
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (valid[j]) ys[k[j]] = (uint8_t)(whole[j] ? (uint32_t)a.party : pi[j]);
  }
#if FSS_DIAG_STAMPS
  if (lane == 0 && a.stamps) {  // one record per wave, as eval_points_kernel's
    const int64_t rec = (int64_t)blockIdx.x * NW + wv;
    uint64_t* o = a.stamps + 4 * (rec < kDiagStampRecords ? rec : 0);
    o[0] = stamp_t0; o[1] = stamp_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  q.done();
}

}  // namespace fssamd
