// Full-domain evaluation (EvalAll) as a short chain of expansion stages.
//
// The evaluated block [first_leaf, first_leaf + 2^r) is the subtree under the
// depth-p node whose path is `prefix` (p = in_bits - r): this is the unit a GPU
// owns when the domain is sharded.  The frontier kernel walks to 2^top nodes
// (point-eval style, a quad of lanes per walk), then every stage lets one lane take
// one node and expand it D levels in registers into 2^D children (each node is
// expanded exactly once: 2 AES per DPF node, 4 per DCF node, 1 per Half-Tree node,
// as the reference's EvalTree).  The last stage converts leaves to group outputs and
// writes them in index order with 64-bit indexing.  Intermediate frontiers live in
// the caller's workspace (nodes of 16 B; 1/16 of the leaves for the last frontier).
//
// The whole plan of a single key (and of an AES batch) is ONE launch (evalall_fused_kernel
// below): every workgroup takes its own share of the walks through all the stages to the
// leaves; the separate frontier / stage kernels remain for ChaCha and DCF batches and the
// shared-prefix point walks, and as the A/B reference (FSS_AMD_FUSED=0).
//
// Follows: fss::Dpf::EvalTree (include/fss/dpf.cuh:250-303), fss::Dcf::EvalTree
// (include/fss/dcf.cuh:315-385), fss::HalfTreeDpf::EvalTree / ConvertLastLevel
// (include/fss/half_tree_dpf.cuh:284-354), GrottoDcf::ExpandTreeRec
// (include/fss/grotto_dcf.cuh:190-238); replaces the kernels of
// include/fss/eval_all_gpu.cuh:207-268, :62-150.
#pragma once
#include <cstddef>
#include "fss_dev.hpp"
#include "fss_point.hpp"
#include "fss_quad.hpp"

// FSS_HT_PAIR = 1: the Half-Tree stages hash two nodes (or the two leaves of one parent) per call as two interleaved
// AES streams; 0: one hash at a time (A/B).
#ifndef FSS_HT_PAIR
#define FSS_HT_PAIR 1
#endif
// DCF stages that store NODES (intermediate stages, the node table of the shared-prefix plan) holding a half's two nodes and
// running values for 64-byte stores, as the leaf stage does with its leaves: OFF.  It pays on big trees (node table of 2^23
// nodes 0.45 -> 0.42 ms, 2^28 leaves 10.0 -> 9.85 ms) and costs on small ones -- its 24 registers put the one-launch kernel
// back at 128 + 124 B of scratch, some of it in the frontier walk (2^24 leaves: walks 27 -> 34 us, 0.649 -> 0.669 ms; 2^20
// leaves 0.089 -> 0.093 ms), which is where configs[2]'s 2^19-node table lives (profiles/r6/dcf_evalall_stores.txt).
#ifndef FSS_DCF_FRONTIER_OCTETS
#define FSS_DCF_FRONTIER_OCTETS 1
#endif
#ifndef FSS_DCF_HOLD_NODES
#define FSS_DCF_HOLD_NODES 0
#endif
#ifndef FSS_LEAF_PAIR_LINES
#define FSS_LEAF_PAIR_LINES 1
#endif
// FSS_HT_HOLD = 1: the Half-Tree leaf stage stores a lane's four leaves together (A/B: 0 = as they are converted).
#ifndef FSS_HT_HOLD
#define FSS_HT_HOLD 1
#endif

namespace fssamd {

struct ExpandArgs {
  int scheme;
  int in_bits;
  int party;
  int group;
  int level0;         // tree level of the input nodes
  int64_t n_in;       // number of input nodes
  uint64_t prefix;      // walk_kernel: path of the subtree root (level0 bits)
  const uint4* in_nodes;
  const uint4* in_v;    // DCF: running group value per node
  uint4* out_nodes;
  uint4* out_v;
  void* ys;             // leaf stage: uint4 per leaf (Grotto: uint8 control bit per leaf)
  int dyn;              // expand_kernel: waves take 64-node chunks from the workgroup's LDS counter (WaveQ)
  uint32_t* gq;         // WaveQ's launch-wide tail: this stream's {counter, waves done}, or nullptr
  int64_t static_chunks;
  uint64_t* stamps;     // FSS_DIAG_STAMPS builds: per-wave {memtime, realtime} x {start, end} of the leaf stage
  // BATCH kernels (fss::gpu::*EvalAllGpuBatch, eval_all_gpu.cuh:511-535): nkeys trees side by side.  Node f of a
  // stage whose inputs are at level l belongs to key f >> l; keys are key-major raw slots, `key_slots` per key.
  int64_t nkeys;
  int key_slots;
  const uint4* seeds;   // one per key
};

// -- one-node expansions ------------------------------------------------------
// DPF / Grotto node -> both children, control bits packed (dpf.cuh:266-289).
// OW < 4 (leaf level only): the children feed a narrow output group -- words [0, OW) and the control bit are
// computed, the rest of left / right is unspecified (aes_fin).
template <class LT, int OW = 4>
__device__ __forceinline__ void dpf_expand(const LT& L, const DevCtx* __restrict__ c, const uint32_t* __restrict__ cw,
                                           const uint32_t st[4], uint32_t left[4], uint32_t right[4]) {
  uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
  uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(st[3], 0, 1);
  if constexpr (OW < 4) prg_gen2_narrow<OW>(L, c, in, left, right);
  else prg_gen2(L, c, in, left, right);  // both children (AES: two interleaved blocks)
  uint32_t wl = cw[3], wr = (cw[3] & ~1u) | (cw[4] & 1u);
  left[0] = and_xor(tm, cw[0], left[0]);   right[0] = and_xor(tm, cw[0], right[0]);
  left[1] = and_xor(tm, cw[1], left[1]);   right[1] = and_xor(tm, cw[1], right[1]);
  left[2] = and_xor(tm, cw[2], left[2]);   right[2] = and_xor(tm, cw[2], right[2]);
  left[3] = and_xor(tm, wl, left[3]);      right[3] = and_xor(tm, wr, right[3]);
}

// Grotto, last level: only the children's control bits (bit 0 of tl / tr; grotto_dcf.cuh:194-198).
template <class LT>
__device__ __forceinline__ void grotto_leaf_bits(const LT& L, const DevCtx* __restrict__ c, const uint32_t* __restrict__ cw,
                                                 const uint32_t st[4], uint32_t& tl, uint32_t& tr) {
  uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
  uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(st[3], 0, 1);
  prg_tbits2(L, c, in, tl, tr);
  tl = and_xor(tm, cw[3], tl);
  tr = and_xor(tm, cw[4], tr);
}

// DCF node (dcf.cuh:331-366): children plus their running values (positive terms;
// the party sign is applied at the leaf).
template <class LT>
__device__ __forceinline__ void dcf_expand(const LT& L, const DevCtx* __restrict__ c, const Grp& g,
                                           const uint32_t* __restrict__ cw, const uint32_t st[4], const Acc& v,
                                           uint32_t left[4], Acc& vl, uint32_t right[4], Acc& vr) {
  uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
  uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(st[3], 0, 1);
  uint32_t bl[4], br[4];
  prg_gen4(L, c, in, left, bl, right, br);
  bl[3] &= ~1u; br[3] &= ~1u;
  uint32_t vcw[4] = {cw[4], cw[5], cw[6], cw[7] & ~1u};
  vl = v; vl.add(g, bl); vl.add_if(g, tm, vcw);
  vr = v; vr.add(g, br); vr.add_if(g, tm, vcw);
  uint32_t wl = cw[3], wr = (cw[3] & ~1u) | (cw[7] & 1u);
  left[0] = and_xor(tm, cw[0], left[0]);   right[0] = and_xor(tm, cw[0], right[0]);
  left[1] = and_xor(tm, cw[1], left[1]);   right[1] = and_xor(tm, cw[1], right[1]);
  left[2] = and_xor(tm, cw[2], left[2]);   right[2] = and_xor(tm, cw[2], right[2]);
  left[3] = and_xor(tm, wl, left[3]);      right[3] = and_xor(tm, wr, right[3]);
}

// One child of a DCF node: its seed and value block as ONE cipher pair (keys 0, 1 for the left child, 2, 3 for the right:
// [sl, vl, sr, vr], dcf.cuh:223), corrected and accumulated as dcf_expand does.  `right` is uniform.  The stages take
// the children one at a time -- a child's eight words are consumed (stored, converted or walked into) before the other
// child's pair runs, so a pair never runs with the other child's outputs waiting: with both children expanded at once
// (dcf_expand) the leaf kernels kept 7 - 13 words per lane in scratch.
template <class LT>
__device__ __forceinline__ void dcf_child(const LT& L, const DevCtx* __restrict__ c, const Grp& g,
                                          const uint32_t* __restrict__ cw, const uint32_t st[4], const Acc& v, bool right,
                                          uint32_t child[4], Acc& vc) {
  const uint32_t in[4] = {st[0], st[1], st[2], st[3] & ~1u};
  const uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(st[3], 0, 1);
  uint32_t b[4];
  prg_gen4_side(L, c, in, right, child, b);
  b[3] &= ~1u;
  const uint32_t vcw[4] = {cw[4], cw[5], cw[6], cw[7] & ~1u};
  vc = v; vc.add(g, b); vc.add_if(g, tm, vcw);
  const uint32_t w3 = right ? ((cw[3] & ~1u) | (cw[7] & 1u)) : cw[3];
  child[0] = and_xor(tm, cw[0], child[0]);
  child[1] = and_xor(tm, cw[1], child[1]);
  child[2] = and_xor(tm, cw[2], child[2]);
  child[3] = and_xor(tm, w3, child[3]);
}

// Half-Tree inner node (half_tree_dpf.cuh:293-302): one hash, right = left ^ parent.
template <class LT>
__device__ __forceinline__ void ht_expand(const LT& L, const DevCtx* __restrict__ c, const uint32_t* __restrict__ cw,
                                          const uint32_t node[4], uint32_t left[4], uint32_t right[4]) {
  uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(node[3], 0, 1);
  uint32_t in[4] = {node[0] ^ c->hash_key[0], node[1] ^ c->hash_key[1], node[2] ^ c->hash_key[2],
                    node[3] ^ c->hash_key[3]};
  uint32_t h[4];
  prg_hash(L, c, in, h);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    left[j] = and_xor(tm, cw[j], h[j]);
    right[j] = left[j] ^ node[j];
  }
}

// Two Half-Tree inner nodes at once: their hashes run as two interleaved AES streams (prg_hash2), so one
// stream's table lookups fly under the other's VALU work (fss_dev.hpp, aes_mmo2).
template <class LT>
__device__ __forceinline__ void ht_expand2(const LT& L, const DevCtx* __restrict__ c, const uint32_t* __restrict__ cw,
                                           const uint32_t na[4], const uint32_t nb[4], uint32_t la[4], uint32_t ra[4],
                                           uint32_t lb[4], uint32_t rb[4]) {
  const uint32_t tma = (uint32_t)__builtin_amdgcn_sbfe(na[3], 0, 1), tmb = (uint32_t)__builtin_amdgcn_sbfe(nb[3], 0, 1);
  uint32_t ia[4], ib[4], ha[4], hb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { ia[j] = na[j] ^ c->hash_key[j]; ib[j] = nb[j] ^ c->hash_key[j]; }
  prg_hash2(L, c, ia, ib, ha, hb);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    la[j] = and_xor(tma, cw[j], ha[j]); ra[j] = la[j] ^ na[j];
    lb[j] = and_xor(tmb, cw[j], hb[j]); rb[j] = lb[j] ^ nb[j];
  }
}

// Half-Tree last level, both leaves of one parent (sigma = 0, 1; half_tree_dpf.cuh:325-354), the two hashes interleaved.
template <class LT, int OW = 4>
__device__ __forceinline__ void ht_convert2(const LT& L, const DevCtx* __restrict__ c, const Grp& g, int party,
                                            const uint32_t* __restrict__ cw, const uint32_t* __restrict__ ocw,
                                            const uint32_t parent[4], uint4& y0, uint4& y1) {
  const uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(parent[3], 0, 1);
  uint32_t in0[4] = {parent[0] ^ c->hash_key[0], parent[1] ^ c->hash_key[1], parent[2] ^ c->hash_key[2],
                     (parent[3] & ~1u) ^ c->hash_key[3]};
  uint32_t in1[4] = {in0[0], in0[1], in0[2], in0[3] ^ 1u};
  uint32_t h0[4], h1[4];
  if constexpr (OW < 4) prg_hash2_narrow<OW>(L, c, in0, in1, h0, h1);
  else prg_hash2(L, c, in0, in1, h0, h1);
  const uint32_t oc[4] = {ocw[0], ocw[1], ocw[2], ocw[3]};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint32_t* h = s ? h1 : h0;
    const uint32_t lcw = s ? (cw[4] & 1u) : (cw[3] & 1u);
    const uint32_t low = (h[3] ^ (tm & lcw)) & 1u;
    uint32_t high[4] = {and_xor(tm, cw[0], h[0]), and_xor(tm, cw[1], h[1]), and_xor(tm, cw[2], h[2]),
                        and_xor(tm, cw[3], h[3]) & ~1u};
    Acc y;
    y.add(g, high);
    y.add_if(g, 0u - low, oc);
    (s ? y1 : y0) = y.finish(g, party != 0);
  }
}

// Half-Tree last level for one sigma (half_tree_dpf.cuh:325-354).
template <class LT>
__device__ __forceinline__ uint4 ht_convert(const LT& L, const DevCtx* __restrict__ c, const Grp& g, int party,
                                            const uint32_t* __restrict__ cw, const uint32_t* __restrict__ ocw,
                                            const uint32_t parent[4], uint32_t sigma) {
  uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(parent[3], 0, 1);
  uint32_t in[4] = {parent[0] ^ c->hash_key[0], parent[1] ^ c->hash_key[1], parent[2] ^ c->hash_key[2],
                    ((parent[3] & ~1u) | sigma) ^ c->hash_key[3]};
  uint32_t h[4];
  prg_hash(L, c, in, h);
  uint32_t lcw = sigma ? (cw[4] & 1u) : (cw[3] & 1u);
  uint32_t low = (h[3] ^ (tm & lcw)) & 1u;
  uint32_t high[4] = {and_xor(tm, cw[0], h[0]), and_xor(tm, cw[1], h[1]), and_xor(tm, cw[2], h[2]),
                      and_xor(tm, cw[3], h[3]) & ~1u};
  uint32_t oc[4] = {ocw[0], ocw[1], ocw[2], ocw[3]};
  Acc y;
  y.add(g, high);
  y.add_if(g, 0u - low, oc);
  return y.finish(g, party != 0);
}

// -- frontier: lane j walks prefix . j from the root (taken child only) ----------
// Produces the 2^top nodes at depth level0 = p + top under the subtree root whose
// path is `prefix` (p bits).  Same walk as point evaluation, but the node (and the
// DCF running value) is stored instead of converted.
template <class LT, int SCHEME, bool BATCH = false>
__global__ __launch_bounds__(1024) void frontier_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                       const uint32_t* __restrict__ k_seed,
                                                       const uint32_t* __restrict__ k_cws, ExpandArgs a, int top) {
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(LT, te0);
  const int depth = a.level0;  // p + top
  const int64_t nout = (BATCH ? a.nkeys : (int64_t)1) << top;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nout; j += (int64_t)gridDim.x * blockDim.x) {
    // BATCH: lane j walks node j & (2^top - 1) of key j >> top; its correction words are that key's slots
    const int64_t key = BATCH ? (j >> top) : 0;
    const uint64_t path = BATCH ? (uint64_t)(j & (((int64_t)1 << top) - 1)) : ((top >= 64 ? 0ull : (a.prefix << top)) | (uint64_t)j);
    const uint32_t* kc = BATCH ? k_cws + 8 * key * a.key_slots : k_cws;
    uint32_t s0[4];
    if constexpr (BATCH) { uint4 t = a.seeds[key]; s0[0] = t.x; s0[1] = t.y; s0[2] = t.z; s0[3] = t.w; }
    else { s0[0] = k_seed[0]; s0[1] = k_seed[1]; s0[2] = k_seed[2]; s0[3] = k_seed[3]; }
    if constexpr (SCHEME == S_DCF) {
      DcfWalk w;
      w.init(s0, a.party, X128{path, 0});
      for (int i = 0; i < depth; ++i) {
        const uint32_t* p = kc + 8 * i;
        CwDcf cw;
#pragma unroll
        for (int q = 0; q < 4; ++q) { cw.s[q] = p[q]; cw.v[q] = p[4 + q]; }
        w.level(L, c, make_grp(a.group, c), depth, i, cw);
      }
      a.out_nodes[j] = make_uint4(w.st[0], w.st[1], w.st[2], w.st[3]);
      a.out_v[j] = make_uint4(w.v.v[0], w.v.v[1], w.v.v[2], w.v.v[3]);
    } else if constexpr (SCHEME == S_HT) {
      HtWalk w;
      w.init(s0, a.party, X128{path, 0});
      for (int i = 0; i < depth; ++i) {
        const uint32_t* p = kc + 8 * i;
        uint32_t cw[4] = {p[0], p[1], p[2], p[3]};
        w.level(L, c, depth, i, cw);
      }
      a.out_nodes[j] = make_uint4(w.node[0], w.node[1], w.node[2], w.node[3]);
    } else {
      DpfWalk w;
      w.init(s0, a.party, X128{path, 0});
      for (int i = 0; i < depth; ++i) {
        const uint32_t* p = kc + 8 * i;
        CwDpf cw{p[0], p[1], p[2], p[3], p[4]};
        w.level(L, c, depth, i, cw);
      }
      a.out_nodes[j] = make_uint4(w.st[0], w.st[1], w.st[2], w.st[3]);
    }
  }
}

// -- frontier with the AES tables: one walk per quad of lanes (fss_quad.hpp) ---------------------------------
// Lanes per frontier walk: a quad; DCF on the AES tables an octet (QuadPrg<Lds, SCHEME>).
template <class LT, int SCHEME>
__host__ __device__ constexpr bool frontier_octets() {
  return FSS_DCF_FRONTIER_OCTETS && SCHEME == S_DCF && std::is_same<LT, Lds>::value;
}
template <class LT, int SCHEME>
__host__ __device__ constexpr int frontier_lanes_log2() {
  return frontier_octets<LT, SCHEME>() ? 3 : 2;
}
// The walks j_first, j_first + j_stride, ... < j_end of this lane's quad (j counts walks: quads of lanes).
template <class LT, int SCHEME, bool BATCH>  // SCHEME: S_DPF (also Grotto), S_DCF or S_HT
__device__ __forceinline__ void quad_frontier_walks(const LT& L, const DevCtx* __restrict__ c,
                                                    const uint32_t* __restrict__ k_seed,
                                                    const uint32_t* __restrict__ k_cws, const ExpandArgs& a, int top,
                                                    int tail, int64_t j_first, int64_t j_end, int64_t j_stride) {
  // `top` levels are produced per key: top - tail walked, then (tail = 2, DPF / Grotto / Half-Tree) the quad expands
  // its node two levels in registers and stores the four grandchildren -- the first, thinnest expansion stage
  // without its launch.  a.level0 = depth of the stored nodes.
  static_assert(SCHEME == S_DPF || SCHEME == S_DCF || SCHEME == S_HT, "quad frontier: DPF / Grotto / DCF / Half-Tree");
  const int q = threadIdx.x & 3;
  const uint32_t is3 = q == 3 ? ~0u : 0u;
  QuadPrg<LT, SCHEME> P;
  // DCF on the AES tables: an octet of lanes per walk (fss_quad.hpp, QuadPrg<Lds, SCHEME>; the callers count walks per
  // eight lanes, kFrontierLanesLog2, and the planner walks one level less so that the octets still take only half of a
  // workgroup's waves -- with all sixteen walking, the waves of a SIMD queue for its issue slot and the walk gets slower)
  constexpr bool kOct = frontier_octets<LT, SCHEME>();
  if constexpr (std::is_same<LT, Lds>::value) P.init(L, c, q, kOct);
  else P.init(L, c, q);
  const Grp g = make_grp(a.group, c);
  const int depth = a.level0 - tail;  // levels walked: p + top - tail
  top -= tail;
  // one node -> both children at tree level lv (dpf_expand / ht_expand), a word per lane
  auto expand = [&](const uint32_t* kc, int lv, uint32_t node, uint32_t& left, uint32_t& right) {
    const uint32_t w = kc[8 * lv + q];
    const uint32_t tm = lsb_mask(quad_get<FSS_QP(3, 3, 3, 3)>(node));
    if constexpr (SCHEME == S_DPF) {
      const uint32_t trw = kc[8 * lv + 4];
      uint32_t l, r;
      P.both(node & ~(is3 & 1u), l, r);
      left = and_xor(tm, w, l);
      right = and_xor(tm, w ^ (is3 & ((w ^ trw) & 1u)), r);
    } else if constexpr (SCHEME == S_HT) {
      left = and_xor(tm, w, P.hash(node ^ P.hk));
      right = left ^ node;
    } else {
      left = right = node;  // no tail for DCF
    }
  };
  for (int64_t j = j_first; j < j_end; j += j_stride) {
    const int64_t key = BATCH ? (j >> top) : 0;
    const uint64_t path = BATCH ? (uint64_t)(j & (((int64_t)1 << top) - 1)) : ((top >= 64 ? 0ull : (a.prefix << top)) | (uint64_t)j);
    const uint32_t* kc = BATCH ? k_cws + 8 * key * a.key_slots : k_cws;
    uint32_t st = BATCH ? reinterpret_cast<const uint32_t*>(a.seeds + key)[q] : k_seed[q];
    st = (st & ~(is3 & 1u)) | (is3 & (uint32_t)a.party);  // dpf.cuh:171-173, half_tree_dpf.cuh:188
    Acc v;  // DCF: the running value, kept whole (and identical) in the four lanes
#pragma unroll 1
    for (int i = 0; i < depth; ++i) quad_level(P, g, kc, i, bit_mask(path, depth - 1 - i), q, is3, st, v);
    uint32_t* out = reinterpret_cast<uint32_t*>(a.out_nodes);
    if (tail == 2 && SCHEME != S_DCF) {
      uint32_t c1[2];
      expand(kc, depth, st, c1[0], c1[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t g0, g1;
        expand(kc, depth + 1, c1[h], g0, g1);
        out[4 * ((j << 2) + 2 * h) + q] = g0;
        out[4 * ((j << 2) + 2 * h + 1) + q] = g1;
      }
    } else {
      if constexpr (kOct) {  // the node is the seed quad's, the running value the value quad's
        if (P.is_v) reinterpret_cast<uint32_t*>(a.out_v)[4 * j + q] = v.v[q];
        else out[4 * j + q] = st;
      } else {
        out[4 * j + q] = st;
        if constexpr (SCHEME == S_DCF) reinterpret_cast<uint32_t*>(a.out_v)[4 * j + q] = v.v[q];
      }
    }
  }
}

template <class LT, int SCHEME, bool BATCH>
__global__ __launch_bounds__(1024) void frontier_quad_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                            const uint32_t* __restrict__ k_seed,
                                                            const uint32_t* __restrict__ k_cws, ExpandArgs a, int top,
                                                            int tail) {
  const DevCtx* c = first_kernarg(K);
  FSS_PRG_PROLOGUE(LT, te0);
  const int64_t nout = (BATCH ? a.nkeys : (int64_t)1) << (top - tail);  // walks
  quad_frontier_walks<LT, SCHEME, BATCH>(L, c, k_seed, k_cws, a, top, tail,
                                         ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> frontier_lanes_log2<LT, SCHEME>(), nout,
                                         ((int64_t)gridDim.x * blockDim.x) >> frontier_lanes_log2<LT, SCHEME>());
}

// -- expansion stage ------------------------------------------------------------
// Lane j takes input node j at level `level0` and writes its descendants E + D levels down at
// (j << (E + D)) + k.  The last D levels are expanded in registers (all 2^D nodes live); the E levels
// above them are walked depth-first by rolled loops (one pair of children live per level), which keeps the
// code and the register count of a deep stage close to those of a D-level one.  LEAF: the outputs are
// the tree's leaves and are converted (dpf.cuh:255-263, dcf.cuh:318-329, half_tree_dpf.cuh
// ConvertLastLevel, grotto_dcf.cuh:194-198) instead of stored as nodes.
// kLdsLevel: the rolled level (counted like E: the outermost of a stage is E = the stage's rolled depth) whose right child
// waits in LDS (`park`, this lane's 16 bytes) instead of in four registers while the left subtree is walked -- the fourth
// rolled level of a DPF leaf stage, which otherwise spills (128 registers, 36 B of scratch per lane): one 16-byte LDS
// write and read per 64 AES blocks.
template <class LT, int SCHEME, int E, int kLdsLevel = 0>
struct OuterLevels {
  template <class Body>
  static __device__ __forceinline__ void run(const LT& L, const DevCtx* __restrict__ c, const Grp& g,
                                             const uint32_t* __restrict__ k_cws, const uint32_t* __restrict__ k_ocw,
                                             int lvl, const uint32_t node[4], const Acc& v, int64_t ob, int sub_log2,
                                             Body& body, int half = -1, uint4* park = nullptr) {
    // half: which child of the innermost rolled level this subtree is (0 / 1), -1 with no rolled level above the body
    if constexpr (E == 0) {
      body(k_cws, k_ocw, node, v, lvl, ob, half);
    } else {
      const uint32_t* cw = k_cws + 8 * lvl;
      if constexpr (SCHEME == S_DCF && std::is_same<LT, Lds>::value) {  // AES: a child at a time -- the parent (node, running value)
                                                                          // waits, not the sibling's outputs (ChaCha: one permutation
                                                                          // yields all four blocks, so both children at once below)
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          uint32_t ch[4];
          Acc cv;
          dcf_child(L, c, g, cw, node, v, h != 0, ch, cv);
          // (`half` here: the subtree's index among those of the rolled levels walked so far, 0 .. 2^levels - 1)
          OuterLevels<LT, SCHEME, E - 1, kLdsLevel>::run(L, c, g, k_cws, k_ocw, lvl + 1, ch, cv, ob + ((int64_t)h << (sub_log2 - 1)),
                                                         sub_log2 - 1, body, (half < 0 ? 0 : 2 * half) + h, park);
        }
        return;
      }
      uint32_t l[4], r[4];
      Acc vl, vr;
      if constexpr (SCHEME == S_DCF) dcf_expand(L, c, g, cw, node, v, l, vl, r, vr);
      else if constexpr (SCHEME == S_HT) ht_expand(L, c, cw, node, l, r);
      else dpf_expand(L, c, cw, node, l, r);
      if constexpr (E == kLdsLevel && SCHEME != S_DCF) {
        *park = make_uint4(r[0], r[1], r[2], r[3]);
        OuterLevels<LT, SCHEME, E - 1, kLdsLevel>::run(L, c, g, k_cws, k_ocw, lvl + 1, l, v, ob, sub_log2 - 1, body, 0, park);
        const uint4 t = *park;
        const uint32_t rr[4] = {t.x, t.y, t.z, t.w};
        OuterLevels<LT, SCHEME, E - 1, kLdsLevel>::run(L, c, g, k_cws, k_ocw, lvl + 1, rr, v, ob + ((int64_t)1 << (sub_log2 - 1)), sub_log2 - 1,
                                                       body, 1, park);
      } else {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          uint32_t ch[4];
          Acc cv;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ch[q] = h ? r[q] : l[q];
            if constexpr (SCHEME == S_DCF) cv.v[q] = h ? vr.v[q] : vl.v[q];
          }
          OuterLevels<LT, SCHEME, E - 1, kLdsLevel>::run(L, c, g, k_cws, k_ocw, lvl + 1, ch, cv, ob + ((int64_t)h << (sub_log2 - 1)),
                                                         sub_log2 - 1, body, h, park);
        }
      }
    }
  }
};

// OW (leaf stages of DPF / Half-Tree with the AES tables): 4 = any output group; 2 = the group reads only the low
// 8 bytes of a leaf seed (Uint<uint8_t .. uint64_t>, uint.cuh:49-68), so the leaf-level blocks compute words 0, 1 and
// the control bit only -- 7 of their last round's 16 lookups vanish (aes_fin), about 2 % of a DPF tree's lookups.
//
// StageBody: the D levels expanded in registers, from one node at level `lvl` to outputs ob .. ob + 2^D - 1
// (a.out_nodes / a.out_v, or the converted leaves in a.ys).
template <class LT, int SCHEME, int D, bool LEAF, int OW>
struct StageBody {
  const LT& L;
  const DevCtx* __restrict__ c;
  const Grp& g;
  const ExpandArgs& a;
  uint32_t* prog;  // fixed shares with progress-ranked wave priorities (rank_prio), or nullptr
  uint32_t mine;
  // DPF leaves under a rolled level: the 64 bytes of the first half are held back and stored together with the
  // second half -- the lane's eight 16-byte stores then fall on one 128-byte line within a few cycles instead of half
  // a line now and half after six more AES blocks (FSS_LEAF_PAIR_LINES, A/B).  The same for DCF (two leaves per
  // half) measured flat to slightly worse and is not taken.
  static constexpr bool kHold = FSS_LEAF_PAIR_LINES && LEAF && (SCHEME == S_DPF || (SCHEME == S_HT && FSS_HT_PAIR && FSS_HT_HOLD)) && D == 2;
  // DCF on the AES tables (one level in registers, a child at a time): the two leaves of the innermost rolled level's first
  // half wait for the two of its second half, and a lane stores 64 contiguous bytes at once.  With 32 bytes per body the
  // 2^28-leaf domain wrote 14.3 GiB for its 4 GiB of leaves and ran at 2.07 GHz under that traffic; 64 bytes: 7.5 GiB,
  // 10.65 -> 9.7 ms.  All eight leaves of a lane held for whole 128-byte lines (a shift register of six more) wrote
  // 6.7 GiB but cost more in moves and spills than it saved: 10.0 ms (profiles/r6/dcf_evalall_stores.txt).
  static constexpr bool kHoldDcf = FSS_LEAF_PAIR_LINES && LEAF && SCHEME == S_DCF && D == 1 && std::is_same<LT, Lds>::value;
  uint4 held[kHold ? 4 : (kHoldDcf ? 3 : 1)];
  static constexpr bool kHoldDcfNodes = FSS_DCF_HOLD_NODES && !LEAF && SCHEME == S_DCF && D == 1 && std::is_same<LT, Lds>::value;
  uint4 heldn[kHoldDcfNodes ? 3 : 1], heldv[kHoldDcfNodes ? 3 : 1];
  __device__ __forceinline__ void operator()(const uint32_t* __restrict__ kc, const uint32_t* __restrict__ koc,
                                             const uint32_t nin[4], const Acc& vin, int lvl, int64_t ob, int half) {
    constexpr int NN = 1 << D;
    const int party = a.party;
    if (prog != nullptr) rank_prio(prog, mine, 0u, 0u);  // a step = one body: the last D levels of one subtree
    if constexpr (SCHEME == S_DCF && D == 1 && std::is_same<LT, Lds>::value) {
      // DCF on the AES tables expands one level in registers: the two children one after the other, each stored (a node and
      // its running value) or converted (a leaf) before the other's cipher pair runs
      const uint32_t* cw = kc + 8 * lvl;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t ch[4];
        Acc cv;
        dcf_child(L, c, g, cw, nin, vin, h != 0, ch, cv);
        if constexpr (!LEAF) {
          const uint4 nn = make_uint4(ch[0], ch[1], ch[2], ch[3]), vv = make_uint4(cv.v[0], cv.v[1], cv.v[2], cv.v[3]);
          if constexpr (kHoldDcfNodes) {  // (FSS_DCF_HOLD_NODES, off) as the leaves below: the first half's two nodes and values
                                          // wait for the second half's, 64 contiguous bytes per array at once
            const int side = half < 0 ? -1 : (half & 1);
            if (side == 0) {
              heldn[h] = nn; heldv[h] = vv;
            } else if (h == 0) {
              heldn[2] = nn; heldv[2] = vv;
            } else {
              if (side == 1) {
                a.out_nodes[ob - 2] = heldn[0]; a.out_nodes[ob - 1] = heldn[1];
              }
              a.out_nodes[ob] = heldn[2];
              a.out_nodes[ob + 1] = nn;
              if (side == 1) {
                a.out_v[ob - 2] = heldv[0]; a.out_v[ob - 1] = heldv[1];
              }
              a.out_v[ob] = heldv[2];
              a.out_v[ob + 1] = vv;
            }
          } else {
            a.out_nodes[ob + h] = nn;
            a.out_v[ob + h] = vv;
          }
        } else {
          const uint32_t* ocp = kc + 8 * a.in_bits + 4;
          const uint32_t oc[4] = {ocp[0], ocp[1], ocp[2], ocp[3]};
          const uint32_t sl[4] = {ch[0], ch[1], ch[2], ch[3] & ~1u};
          const uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(ch[3], 0, 1);
          Acc y = cv;
          y.add(g, sl);
          y.add_if(g, tm, oc);
          const uint4 yy = y.finish(g, party != 0);
          uint4* ys = reinterpret_cast<uint4*>(a.ys);
          if constexpr (kHoldDcf) {
            // `half` counts the subtrees of the rolled levels; its low bit is the innermost level's side
            const int side = half < 0 ? -1 : (half & 1);
            if (side == 0) {
              held[h] = yy;
            } else if (h == 0) {
              held[2] = yy;
            } else {  // the last of the four (or, with no rolled level above, of the two): everything out together
              if (side == 1) { ys[ob - 2] = held[0]; ys[ob - 1] = held[1]; }
              ys[ob] = held[2];
              ys[ob + 1] = yy;
            }
          } else {
            ys[ob + h] = yy;
          }
        }
      }
      return;
    }
    uint32_t nodes[NN][4];
    Acc vs[SCHEME == S_DCF ? NN : 1];
    nodes[0][0] = nin[0]; nodes[0][1] = nin[1]; nodes[0][2] = nin[2]; nodes[0][3] = nin[3];
    if constexpr (SCHEME == S_DCF) vs[0] = vin;
    // Half-Tree: the tree's last level is the conversion, not an expansion.
    constexpr int DE = (SCHEME == S_HT && LEAF) ? D - 1 : D;
#pragma unroll
    for (int lev = 0; lev < DE; ++lev) {
      const uint32_t* cw = kc + 8 * (lvl + lev);
#pragma unroll
      for (int k = (1 << lev) - 1; k >= 0; --k) {
        uint32_t l[4], r[4];
        if constexpr (SCHEME == S_HT && FSS_HT_PAIR) {
          if (lev > 0) {  // nodes k and k - 1 of this level together (k odd first: their outputs land above k - 1's)
            if (k & 1) {
              uint32_t l2[4], r2[4];
              ht_expand2(L, c, cw, nodes[k], nodes[k - 1], l, r, l2, r2);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                nodes[2 * k][q] = l[q]; nodes[2 * k + 1][q] = r[q];
                nodes[2 * k - 2][q] = l2[q]; nodes[2 * k - 1][q] = r2[q];
              }
            }
            continue;
          }
        }
        if constexpr (SCHEME == S_DCF) {
          Acc vl, vr;
          dcf_expand(L, c, g, cw, nodes[k], vs[k], l, vl, r, vr);
          vs[2 * k] = vl; vs[2 * k + 1] = vr;
        } else if constexpr (SCHEME == S_HT) {
          ht_expand(L, c, cw, nodes[k], l, r);
        } else if constexpr (SCHEME == S_GROTTO && LEAF) {
          if (lev == DE - 1) {  // the leaves are control bits: partial blocks
            l[0] = l[1] = l[2] = r[0] = r[1] = r[2] = 0u;
            grotto_leaf_bits(L, c, cw, nodes[k], l[3], r[3]);
          } else {
            dpf_expand(L, c, cw, nodes[k], l, r);
          }
        } else {
          if (LEAF && SCHEME == S_DPF && lev == DE - 1) dpf_expand<LT, OW>(L, c, cw, nodes[k], l, r);
          else dpf_expand(L, c, cw, nodes[k], l, r);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { nodes[2 * k][q] = l[q]; nodes[2 * k + 1][q] = r[q]; }
      }
    }
    if constexpr (!LEAF) {
#pragma unroll
      for (int k = 0; k < NN; ++k) {
        a.out_nodes[ob + k] = make_uint4(nodes[k][0], nodes[k][1], nodes[k][2], nodes[k][3]);
        if constexpr (SCHEME == S_DCF) a.out_v[ob + k] = make_uint4(vs[k].v[0], vs[k].v[1], vs[k].v[2], vs[k].v[3]);
      }
    } else if constexpr (SCHEME == S_HT) {
      const uint32_t* cw = kc + 8 * (a.in_bits - 1);
      uint4* ys = reinterpret_cast<uint4*>(a.ys);
#if FSS_HT_PAIR && FSS_HT_HOLD
      if constexpr (DE == 1) {  // the lane's four leaves (64 bytes) stored together
        uint4 y[4];
        ht_convert2<LT, OW>(L, c, g, party, cw, koc, nodes[0], y[0], y[1]);
        ht_convert2<LT, OW>(L, c, g, party, cw, koc, nodes[1], y[2], y[3]);
        if (half == 0) {  // under a rolled level: with the other half, one 128-byte line at once
#pragma unroll
          for (int k = 0; k < 4; ++k) held[k] = y[k];
          return;
        }
        if (half == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) ys[ob - 4 + k] = held[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ys[ob + k] = y[k];
        return;
      }
#endif
#pragma unroll
      for (int k = 0; k < (1 << DE); ++k) {
#if FSS_HT_PAIR
        uint4 y0, y1;
        ht_convert2<LT, OW>(L, c, g, party, cw, koc, nodes[k], y0, y1);
        ys[ob + 2 * k] = y0;
        ys[ob + 2 * k + 1] = y1;
#else
        ys[ob + 2 * k] = ht_convert(L, c, g, party, cw, koc, nodes[k], 0u);
        ys[ob + 2 * k + 1] = ht_convert(L, c, g, party, cw, koc, nodes[k], 1u);
#endif
      }
    } else if constexpr (SCHEME == S_GROTTO) {
      uint8_t* ys = reinterpret_cast<uint8_t*>(a.ys);
      if constexpr (NN >= 4) {
#pragma unroll
        for (int k = 0; k < NN; k += 4) {
          uint32_t w = (nodes[k][3] & 1u) | ((nodes[k + 1][3] & 1u) << 8) | ((nodes[k + 2][3] & 1u) << 16) |
                       ((nodes[k + 3][3] & 1u) << 24);
          *reinterpret_cast<uint32_t*>(ys + ob + k) = w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < NN; ++k) ys[ob + k] = (uint8_t)(nodes[k][3] & 1u);
      }
    } else {
      const uint32_t* ocp = kc + 8 * a.in_bits + (SCHEME == S_DCF ? 4 : 0);
      uint32_t oc[4] = {ocp[0], ocp[1], ocp[2], ocp[3]};
      uint4* ys = reinterpret_cast<uint4*>(a.ys);
#pragma unroll
      for (int k = 0; k < NN; ++k) {
        uint32_t s[4] = {nodes[k][0], nodes[k][1], nodes[k][2], nodes[k][3] & ~1u};
        uint32_t tm = (uint32_t)__builtin_amdgcn_sbfe(nodes[k][3], 0, 1);
        Acc y;
        if constexpr (SCHEME == S_DCF) y = vs[k];
        y.add(g, s);
        y.add_if(g, tm, oc);
        if constexpr (kHold) {
          const uint4 yy = y.finish(g, party != 0);
          if (half == 0) {
            held[k] = yy;
          } else {
            if (half == 1) ys[ob - NN + k] = held[k];
            ys[ob + k] = yy;
          }
        } else {
          ys[ob + k] = y.finish(g, party != 0);
        }
      }
    }
  }
};

// One stage over the input nodes [j_lo, j_hi) handed out 64 at a time by `next` (WaveQ / LocalQ): lane j loads
// node j, walks E levels depth-first and expands the last D in registers.
template <class LT, int SCHEME, int D, bool LEAF, int E, bool BATCH, int OW, bool UNI = false, class Queue>
__device__ __forceinline__ void run_stage(const LT& L, const DevCtx* __restrict__ c, const Grp& g, const ExpandArgs& a,
                                          const uint32_t* __restrict__ k_cws, const uint32_t* __restrict__ k_ocw,
                                          Queue& q, int64_t j_hi, uint32_t* prog = nullptr, uint4* park = nullptr) {
  constexpr int PK = (E >= 4 && SCHEME != S_DCF) ? E : 0;  // a fourth rolled level parks its right child in LDS (OuterLevels)
  StageBody<LT, SCHEME, D, LEAF, OW> body{L, c, g, a, prog, 0u, {}};
  for (int64_t j; q.next(j);) {
    if (j >= j_hi) continue;  // ragged last chunk
    uint32_t n0[4];
    Acc v0;
    {
      uint4 t = a.in_nodes[j];
      n0[0] = t.x; n0[1] = t.y; n0[2] = t.z; n0[3] = t.w;
      if constexpr (SCHEME == S_DCF) {
        uint4 u = a.in_v[j];
        v0.v[0] = u.x; v0.v[1] = u.y; v0.v[2] = u.z; v0.v[3] = u.w;
      }
    }
    if constexpr (BATCH && UNI) {
      // Node j belongs to key j >> level0.  UNI (the host's promise, launch_fused_evalall_batch): every stage has at least
      // 64 nodes per key and every workgroup's node range starts on a multiple of 64, so the 64 nodes of a chunk share
      // their key: its slots (and the Half-Tree ocw) are addressed from scalar registers and read with scalar loads, as
      // for a single key.  Per-lane addresses (the form below) make every correction word of every level a vector load
      // -- five per expansion, 5 % of a batch's time against a single tree of the same size (profiles/r5).
      const int64_t key = j >> a.level0;
      const int64_t key0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(key >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)key);
      const uint32_t* kc = k_cws + 8 * key0 * a.key_slots;
      const uint32_t* koc = k_ocw ? k_ocw + 4 * key0 : k_ocw;
      OuterLevels<LT, SCHEME, E, PK>::run(L, c, g, kc, koc, a.level0, n0, v0, j << (E + D), E + D, body, -1, park);
    } else if constexpr (BATCH) {
      const int64_t key = j >> a.level0;
      const uint32_t* kc = k_cws + 8 * key * a.key_slots;
      const uint32_t* koc = k_ocw ? k_ocw + 4 * key : k_ocw;
      OuterLevels<LT, SCHEME, E, PK>::run(L, c, g, kc, koc, a.level0, n0, v0, j << (E + D), E + D, body, -1, park);
    } else {
      OuterLevels<LT, SCHEME, E, PK>::run(L, c, g, k_cws, k_ocw, a.level0, n0, v0, j << (E + D), E + D, body, -1, park);
    }
  }
}

template <class LT, int SCHEME, int D, bool LEAF, int BS, int E = 0, bool BATCH = false, int OW = 4>
__global__ __launch_bounds__(BS) void expand_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                    const uint32_t* __restrict__ k_cws,  // raw Cw slots of the key
                                                    const uint32_t* __restrict__ k_ocw,  // Half-Tree
                                                    ExpandArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const Grp g = make_grp(a.group, c);

#if FSS_DIAG_STAMPS
  const uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  WaveQ q(a.dyn, &waveq_ctr_, a.n_in, 1, a.gq, a.static_chunks);
  run_stage<LT, SCHEME, D, LEAF, E, BATCH, OW>(L, c, g, a, k_cws, k_ocw, q, a.n_in);
#if FSS_DIAG_STAMPS
  if (LEAF && (threadIdx.x & 63) == 0 && a.stamps) {
    const int64_t rec = (int64_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
    uint64_t* o = a.stamps + 4 * (rec < kDiagStampRecords ? rec : 0);
    o[0] = stamp_t0; o[1] = stamp_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  q.done();
}

// -- the whole chain in one launch ------------------------------------------------------------------------------
// Every workgroup owns a contiguous share of the frontier walks and takes it all the way down by itself: its walks,
// its intermediate stages (through a private slice of the two workspace buffers), its leaves.  No workgroup ever
// reads what another wrote, so the phases are separated by workgroup barriers only -- one launch, one table fill and
// no drain / ramp between the stages of a plan (2^25-leaf block: four launches of 25 + 12 + 42 + 585 us become one;
// the thin stages were 12 % of the time for 6 % of the blocks).  The nodes a workgroup reads back were written by its
// own CU, so they also come from its own XCD's L2.
constexpr int kFusedStampSlots = 16;  // FSS_DIAG_STAMPS builds: stamps per wave of evalall_fused_kernel
struct FusedArgs {
  ExpandArgs a;       // scheme, in_bits, party, group, prefix, ys; level0 / in / out are set per phase
  int p;              // depth of the block's root
  int top, tail;      // frontier depth below p (tail levels of it expanded in registers after the walk)
  int nmid;           // intermediate stages
  int mid_t[6];       // their depths, 1 .. 3: up to two levels in registers (DCF: one), the rest rolled above them
  uint4* buf[2];      // workspace halves; workgroup b owns [b * wg_cap[i], (b + 1) * wg_cap[i])
  uint4* bufv[2];     // DCF: the running values of those nodes, same slices
  int64_t wg_cap[2];
  int rank;           // 1: the stages hand a workgroup's chunks to its waves in fixed shares, priorities by progress
};

// The workgroup's own 64-node chunks of [lo, hi), handed to its waves through the LDS counter.
// fixed = true: wave w takes chunks w, w + 16, ... (equal shares; run with rank_prio so that they also end together).
struct LocalQ {
  uint32_t* ctr;
  int64_t lo, n, cur;
  bool fixed;
  __device__ __forceinline__ int64_t fetch() {
    if (fixed) return cur + (blockDim.x >> 6);
    uint32_t v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(ctr, 1u);
    return (int64_t)__builtin_amdgcn_readfirstlane(v);
  }
  __device__ __forceinline__ LocalQ(uint32_t* lds_ctr, int64_t lo_, int64_t hi_, bool fixed_ = false)
      : ctr(lds_ctr), lo(lo_), n(hi_ - lo_), fixed(fixed_) {
    cur = (int64_t)(threadIdx.x >> 6) - (blockDim.x >> 6);
    cur = fetch();
  }
  __device__ __forceinline__ bool next(int64_t& j) {
    if (cur * 64 >= n) return false;
    j = lo + cur * 64 + (threadIdx.x & 63);
    cur = fetch();
    return true;
  }
};

// BATCH: nkeys trees side by side (a.nkeys, a.seeds, a.key_slots; walk j belongs to key j >> (top - tail)).
// OW = 0: the last stage stores the NODES of its level (and the DCF running values) at their global index in
// f.a.out_nodes / f.a.out_v instead of converting leaves -- the node table of the shared-prefix point walks
// (fss_amd_*_eval_points_shared), whose top `prefix_bits` levels are this kernel's whole tree.
template <class LT, int SCHEME, int LD, int LE, int OW, bool BATCH = false, bool UNI = false>
__global__ __launch_bounds__(1024) void evalall_fused_kernel(const DevCtx K, const uint32_t* __restrict__ te0,
                                                            const uint32_t* __restrict__ k_seed,
                                                            const uint32_t* __restrict__ k_cws,
                                                            const uint32_t* __restrict__ k_ocw, FusedArgs f) {
  static_assert(SCHEME == S_DPF || SCHEME == S_GROTTO || SCHEME == S_HT || SCHEME == S_DCF);
  constexpr int WS = SCHEME == S_GROTTO ? S_DPF : SCHEME;
  constexpr int DM = SCHEME == S_DCF ? 1 : 2;  // levels a stage expands in registers
  const DevCtx* c = first_kernarg(K);
#if FSS_DIAG_STAMPS
  // per-wave phase stamps (s_memrealtime, 100 MHz): entry | tables filled | own walks done | then per stage {start
  // (after the barrier), own chunks done}; tools/evalall_phases.py prints the table (kFusedStampSlots per wave)
  // slots 0 .. 12: the phase stamps; 13 = s_memrealtime at the wave's end, 14 / 15 = s_memtime (shader clock) at its
  // entry / end: in-kernel clock = (slot 15 - slot 14) / (slot 13 - slot 0) x 100 MHz (bench.py's `clk`).  A launch
  // with more waves than the context's buffer has records for (FSS_AMD_FUSED_GRID above 320 workgroups) goes unstamped.
  const int64_t stamp_rec = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint64_t* const stamp_o = (f.a.stamps && (stamp_rec + 1) * kFusedStampSlots <= (int64_t)kDiagStampRecords * 5)
                                ? f.a.stamps + (int64_t)kFusedStampSlots * stamp_rec : nullptr;
  int stamp_k = 0;
  if (stamp_o && (threadIdx.x & 63) == 0) stamp_o[14] = __builtin_amdgcn_s_memtime();
#define FSS_FUSED_STAMP()                                                                                          \
  do {                                                                                                             \
    if (stamp_o && (threadIdx.x & 63) == 0 && stamp_k < kFusedStampSlots - 3) stamp_o[stamp_k] = __builtin_amdgcn_s_memrealtime(); \
    ++stamp_k;                                                                                                     \
  } while (0)
#else
#define FSS_FUSED_STAMP() do { } while (0)
#endif
  FSS_FUSED_STAMP();
  FSS_PROGRESS_PROLOGUE();
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  FSS_FUSED_STAMP();
  const Grp g = make_grp(f.a.group, c);
  uint32_t* const prog = f.rank ? wave_prog_ : nullptr;
  // this workgroup's walks [w_lo, w_hi) of the 2^(top - tail) per key
  const int64_t walks = (BATCH ? f.a.nkeys : (int64_t)1) << (f.top - f.tail);
  const int64_t w_lo = walks * blockIdx.x / gridDim.x, w_hi = walks * (blockIdx.x + 1) / gridDim.x;
  uint4* const mine[2] = {f.buf[0] + (int64_t)blockIdx.x * f.wg_cap[0], f.buf[1] + (int64_t)blockIdx.x * f.wg_cap[1]};
  uint4* const minev[2] = {f.bufv[0] + (int64_t)blockIdx.x * f.wg_cap[0], f.bufv[1] + (int64_t)blockIdx.x * f.wg_cap[1]};
  int64_t lo = w_lo << f.tail, hi = w_hi << f.tail;  // node range of the current level (indices within the block)
  int level = f.p + f.top;
  ExpandArgs a = f.a;
  {  // ---- frontier: nodes [lo, hi) at depth p + top into mine[0]
    a.level0 = level;
    a.out_nodes = mine[0] - lo;
    a.out_v = minev[0] - lo;
    quad_frontier_walks<LT, WS, BATCH>(L, c, k_seed, k_cws, a, f.top, f.tail, w_lo + (threadIdx.x >> frontier_lanes_log2<LT, WS>()), w_hi,
                                       blockDim.x >> frontier_lanes_log2<LT, WS>());
  }
  FSS_FUSED_STAMP();
  int cur = 0;
  for (int s = 0; s < f.nmid; ++s) {  // ---- intermediate stages: mine[cur] -> mine[cur ^ 1]
    __syncthreads();  // the nodes of the phase before are in place (same CU: same L1), the chunk counter is idle
    if (threadIdx.x == 0) waveq_ctr_ = 0;
    if (threadIdx.x < 16) wave_prog_[threadIdx.x] = 0;
    __syncthreads();
    FSS_FUSED_STAMP();
    const int t = f.mid_t[s];
    a.level0 = level;
    a.in_nodes = mine[cur] - lo;
    a.in_v = minev[cur] - lo;
    a.out_nodes = mine[cur ^ 1] - (lo << t);
    a.out_v = minev[cur ^ 1] - (lo << t);
    LocalQ q(&waveq_ctr_, lo, hi, f.rank != 0);
    if (t == 1) run_stage<LT, SCHEME, 1, false, 0, BATCH, 4, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog);
    else if (t == 2) run_stage<LT, SCHEME, DM, false, 2 - DM, BATCH, 4, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog);
    else run_stage<LT, SCHEME, DM, false, 3 - DM, BATCH, 4, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog);
    lo <<= t; hi <<= t; level += t;
    cur ^= 1;
    FSS_FUSED_STAMP();
  }
  __syncthreads();
  if (threadIdx.x == 0) waveq_ctr_ = 0;
  if (threadIdx.x < 16) wave_prog_[threadIdx.x] = 0;
  __syncthreads();
  FSS_FUSED_STAMP();
  {  // ---- leaf stage: mine[cur] -> ys (OW = 0: -> the node table)
    a.level0 = level;
    a.in_nodes = mine[cur] - lo;
    a.in_v = minev[cur] - lo;
    LocalQ q(&waveq_ctr_, lo, hi, f.rank != 0);
    if constexpr (OW == 0) {
      a.out_nodes = f.a.out_nodes;
      a.out_v = f.a.out_v;
      run_stage<LT, SCHEME, LD, false, LE, BATCH, 4, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog);
    } else if constexpr (LE >= 4) {
      __shared__ uint4 park_[1024];  // 16 KiB beside the 128 KiB of tables: one right child per lane (OuterLevels)
      run_stage<LT, SCHEME, LD, true, LE, BATCH, OW, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog, &park_[threadIdx.x]);
    } else {
      run_stage<LT, SCHEME, LD, true, LE, BATCH, OW, UNI>(L, c, g, a, k_cws, k_ocw, q, hi, prog);
    }
  }
  FSS_FUSED_STAMP();
#if FSS_DIAG_STAMPS
  if (stamp_o && (threadIdx.x & 63) == 0) {
    stamp_o[13] = __builtin_amdgcn_s_memrealtime();
    stamp_o[15] = __builtin_amdgcn_s_memtime();
  }
#endif
#undef FSS_FUSED_STAMP
}

#ifndef FSS_EVALALL_TEMPLATES_ONLY  // the plain (non-template) kernels below belong to one unit: fss_abi_evalall.hip
// Grotto EvalAll phase 2 (grotto_dcf.cuh:158-162): in-place running XOR over the
// leaf control bits.  Three passes: per-chunk parity, scan of the chunk parities,
// per-chunk running XOR seeded with the carry.
constexpr int kScanChunk = 4096;  // bytes per lane-chunk group: one workgroup of 256 lanes x 16 B

__global__ __launch_bounds__(256) void grotto_chunk_parity_kernel(const uint8_t* __restrict__ t, int64_t n,
                                                                 uint8_t* __restrict__ parity) {
  __shared__ uint32_t red[4];
  int64_t base = (int64_t)blockIdx.x * kScanChunk + threadIdx.x * 16;
  uint32_t x = 0;
  if (base + 16 <= n) {
    uint4 v = *reinterpret_cast<const uint4*>(t + base);
    x = v.x ^ v.y ^ v.z ^ v.w;
  } else {
    for (int q = 0; q < 16; ++q)
      if (base + q < n) x ^= t[base + q];
  }
  x ^= x >> 16; x ^= x >> 8; x &= 1u;
  uint64_t b = __ballot(x != 0);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __popcll(b) & 1;
  __syncthreads();
  if (threadIdx.x == 0) parity[blockIdx.x] = (uint8_t)((red[0] ^ red[1] ^ red[2] ^ red[3]) & 1u);
}

// Serial exclusive scan of chunk parities by one wave (nchunks is small: n / 4096).
__global__ __launch_bounds__(64) void grotto_carry_kernel(uint8_t* parity, int64_t nchunks) {
  // lane-strided blocks of the parity array; wave-level exclusive XOR scan of block parities.
  int64_t per = (nchunks + 63) / 64;
  int64_t b0 = (int64_t)threadIdx.x * per, b1 = b0 + per < nchunks ? b0 + per : nchunks;
  uint32_t p = 0;
  for (int64_t i = b0; i < b1; ++i) p ^= parity[i];
  uint64_t ball = __ballot(p & 1u);
  uint64_t below = threadIdx.x == 0 ? 0ull : (ball & ((~0ull) >> (64 - threadIdx.x)));
  uint32_t carry = __popcll(below) & 1;
  for (int64_t i = b0; i < b1; ++i) {
    uint32_t v = parity[i];
    parity[i] = (uint8_t)carry;  // exclusive: parity of everything before chunk i
    carry ^= v;
  }
}

__global__ __launch_bounds__(256) void grotto_chunk_scan_kernel(uint8_t* __restrict__ t, int64_t n,
                                                               const uint8_t* __restrict__ carry_in) {
  __shared__ uint32_t wsum[4];
  int64_t base = (int64_t)blockIdx.x * kScanChunk + threadIdx.x * 16;
  alignas(16) uint8_t b[16];
  if (base + 16 <= n) {
    *reinterpret_cast<uint4*>(b) = *reinterpret_cast<const uint4*>(t + base);
  } else {
    for (int q = 0; q < 16; ++q) b[q] = base + q < n ? t[base + q] : 0;
  }
  uint32_t run = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { run ^= b[q] & 1u; b[q] = (uint8_t)run; }
  // exclusive prefix of lane totals inside the wave, then across the 4 waves
  uint64_t ball = __ballot(run != 0);
  uint32_t lane = threadIdx.x & 63;
  uint64_t below = lane == 0 ? 0ull : (ball & ((~0ull) >> (64 - lane)));
  uint32_t pre = __popcll(below) & 1;
  if (lane == 0) wsum[threadIdx.x >> 6] = __popcll(ball) & 1;
  __syncthreads();
  uint32_t w = threadIdx.x >> 6;
  for (uint32_t q = 0; q < w; ++q) pre ^= wsum[q];
  pre ^= carry_in[blockIdx.x] & 1u;
  if (pre) {
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] ^= 1;
  }
  if (base + 16 <= n) {
    *reinterpret_cast<uint4*>(t + base) = *reinterpret_cast<uint4*>(b);
  } else {
    for (int q = 0; q < 16; ++q)
      if (base + q < n) t[base + q] = b[q];
  }
}

// GrottoDcf::Preprocess phase 2a (grotto_dcf.cuh:100-103): one tree level, p[j] = p[2j+1] ^ p[2j+2].
__global__ __launch_bounds__(256) void grotto_tree_level_kernel(uint8_t* __restrict__ p, int64_t first, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    int64_t j = first + i;
    p[j] = p[2 * j + 1] ^ p[2 * j + 2];
  }
}

// GrottoDcf::Eval (grotto_dcf.cuh:116-135): prefix-parity walk over the table, one point per lane.
__global__ __launch_bounds__(256) void grotto_table_eval_kernel(const uint8_t* __restrict__ p, int in_bits,
                                                               const void* __restrict__ xs, int x_bytes, int64_t n,
                                                               uint8_t* __restrict__ ys) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint64_t x = x_bytes == 1   ? (uint64_t) reinterpret_cast<const uint8_t*>(xs)[i]
                 : x_bytes == 2 ? (uint64_t) reinterpret_cast<const uint16_t*>(xs)[i]
                 : x_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t*>(xs)[i]
                                : reinterpret_cast<const uint64_t*>(xs)[i];
    uint64_t e = x + 1;
    uint32_t pi;
    if (e == (1ull << in_bits)) {
      pi = p[0];
    } else {
      pi = 0;
      uint64_t cur = 0;
      for (int l = 0; l < in_bits; ++l) {
        if ((e >> (in_bits - 1 - l)) & 1) { pi ^= p[2 * cur + 1]; cur = 2 * cur + 2; }
        else cur = 2 * cur + 1;
      }
    }
    ys[i] = (uint8_t)(pi & 1u);
  }
}

#endif  // FSS_EVALALL_TEMPLATES_ONLY

}  // namespace fssamd
