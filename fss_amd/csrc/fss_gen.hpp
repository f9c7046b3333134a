// Dealer-side kernels: raw PRG, key relayout, key generation (one key per lane).
#pragma once
#include <type_traits>
#include "fss_dev.hpp"
#include "fss_point.hpp"

namespace fssamd {

// out[i*mul + j] = AES_{key j}(seeds[i]) ^ seeds[i]   (include/fss/prg/aes128_mmo.cuh:79-89)
template <class LT>
__global__ __launch_bounds__(1024) void prg_gen_kernel(const DevCtx K, const uint32_t* __restrict__ te0, int mul,
                                                      const uint4* __restrict__ seeds, int64_t n,
                                                      uint4* __restrict__ out, int dyn) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  WaveQ q(dyn, &waveq_ctr_, n, 1);
  for (int64_t i; q.next(i);) {
    if (i >= n) continue;
    uint4 s = seeds[i];
    uint32_t in[4] = {s.x, s.y, s.z, s.w};
    // one path per mul (wave-uniform), each with its own named outputs: a shared o[4][4] that the three paths fill to
    // different depths stayed a stack object (16 B of scratch per lane with AES, 40 B with ChaCha)
    auto put = [&](int j, const uint32_t o[4]) { out[i * mul + j] = make_uint4(o[0], o[1], o[2], o[3]); };
    if (mul == 1) {
      uint32_t o0[4];
      if constexpr (std::is_same_v<LT, Lds>) aes_mmo<false>(L, c->rk[0], 0, in, o0);
      else prg_hash(L, c, in, o0);
      put(0, o0);
    } else if (mul == 2) {
      uint32_t o0[4], o1[4];
      prg_gen2(L, c, in, o0, o1);
      put(0, o0); put(1, o1);
    } else {
      uint32_t o0[4], o1[4], o2[4], o3[4];
      prg_gen4(L, c, in, o0, o1, o2, o3);
      put(0, o0); put(1, o1); put(2, o2); put(3, o3);
    }
  }
}

// Key-major raw Cw slots -> level-major arrays (include/fss/point_eval_gpu.cuh:39-91).
// A 256-lane workgroup owns 64 keys; its four waves take levels i = w, w+4, ... so
// the 128-byte lines (4 consecutive slots of one key) are shared through L1.
template <int SCHEME>
__global__ __launch_bounds__(256) void relayout_kernel(int in_bits, int stride, const uint4* __restrict__ cws,
                                                       int64_t nkeys, uint4* __restrict__ cw_s,
                                                       uint4* __restrict__ cw_v, uint64_t* __restrict__ tr_bits,
                                                       uint4* __restrict__ out_cw, uint32_t* __restrict__ extra,
                                                       int tr32) {
  __shared__ unsigned long long bits[2][64];  // levels 0..63 / 64..127
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.x * 64 + lane;
  if (wave < 2) bits[wave][lane] = 0ull;
  __syncthreads();
  if (k < nkeys) {
    const uint4* slot = cws + k * (int64_t)stride * 2;
    unsigned long long mine = 0ull, mine_hi = 0ull;
    for (int i = wave; i < in_bits; i += 4) {
      uint4 s = slot[2 * i], v = slot[2 * i + 1];
      cw_s[(int64_t)i * nkeys + k] = s;
      if (SCHEME == S_DCF) cw_v[(int64_t)i * nkeys + k] = v;
      if (SCHEME == S_DPF) {
        if (i < 64) mine |= (unsigned long long)(v.x & 1u) << i;
        else mine_hi |= (unsigned long long)(v.x & 1u) << (i - 64);
      }
    }
    if (SCHEME == S_DPF && mine) atomicOr(&bits[0][lane], mine);
    if (SCHEME == S_DPF && mine_hi) atomicOr(&bits[1][lane], mine_hi);
  }
  __syncthreads();
  if (wave == 0 && k < nkeys) {
    const uint4* slot = cws + k * (int64_t)stride * 2;
    if (SCHEME == S_DPF) {
      if (tr32) reinterpret_cast<uint32_t*>(tr_bits)[k] = (uint32_t)bits[0][lane];  // the reference's `extra`
      else tr_bits[k] = bits[0][lane];
      if (in_bits > 64) tr_bits[nkeys + k] = bits[1][lane];
      if (out_cw) out_cw[k] = slot[2 * in_bits];  // VDPF keys (stride in_bits) have no such slot
    } else if (SCHEME == S_DCF) {
      out_cw[k] = slot[2 * in_bits + 1];
    } else {
      extra[k] = slot[2 * (in_bits - 1) + 1].x & 1u;
    }
  }
}

// The same transposition staged through LDS, the mirror image of key_major_kernel below: a wave owns 64 keys
// and moves them four levels at a time; on the key-major side eight lanes read the 128 contiguous bytes (four
// slots) of one key, on the level-major side a store instruction is 64 keys x 16 B of one level.
template <int SCHEME>
__global__ __launch_bounds__(256) void relayout_tiled_kernel(int in_bits, int stride, const uint4* __restrict__ cws,
                                                             int64_t nkeys, uint4* __restrict__ cw_s,
                                                             uint4* __restrict__ cw_v, uint64_t* __restrict__ tr_bits,
                                                             uint4* __restrict__ out_cw, uint32_t* __restrict__ extra,
                                                             int tr32) {
  __shared__ uint4 tile[4][8][65];  // [wave][level-in-group * 2 + half][key], padded: the eight lanes of a key hit different banks
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ntiles = (nkeys + 63) / 64;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wv; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t k0 = t * 64, k = k0 + lane;
    const bool mine = k < nkeys;
    unsigned long long lo = 0ull, hi = 0ull;
    for (int ig = 0; ig < stride; ig += 4) {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int kk = pass * 8 + (lane >> 3), part = lane & 7, i = ig + (part >> 1);
        if (k0 + kk < nkeys && i < stride) tile[wv][part][kk] = cws[(k0 + kk) * (int64_t)stride * 2 + 2 * ig + part];
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ig + j;
        if (!mine || i >= stride) continue;
        const uint4 s4 = tile[wv][2 * j][lane], v4 = tile[wv][2 * j + 1][lane];
        if (i < in_bits) {
          cw_s[(int64_t)i * nkeys + k] = s4;
          if (SCHEME == S_DCF) cw_v[(int64_t)i * nkeys + k] = v4;
          if (SCHEME == S_DPF) {
            if (i < 64) lo |= (unsigned long long)(v4.x & 1u) << i;
            else hi |= (unsigned long long)(v4.x & 1u) << (i - 64);
          }
          if (SCHEME == S_HT && i == in_bits - 1) extra[k] = v4.x & 1u;
        } else {  // slot in_bits of a DPF / DCF key
          if (SCHEME == S_DPF && out_cw) out_cw[k] = s4;
          if (SCHEME == S_DCF) out_cw[k] = v4;
        }
      }
      wave_lds_sync();
    }
    if (SCHEME == S_DPF && mine) {
      if (tr32) reinterpret_cast<uint32_t*>(tr_bits)[k] = (uint32_t)lo;  // the reference's `extra`
      else tr_bits[k] = lo;
      if (in_bits > 64) tr_bits[nkeys + k] = hi;
    }
  }
}


// ---- the same two transpositions with whole lines on BOTH sides ("flat") ---------------------------------------
// A wave owns 8 keys at a time.  Eight keys are one 128-byte line per level on the level-major side (8 x 16 B), and
// on the key-major side their slots are one contiguous region of 8 x slots x 32 B that starts on a line boundary
// (8 x 32 x slots is a multiple of 128) -- so the wave moves that region as a flat run of 16-byte pieces, 1 KiB per
// instruction, every line whole and touched by this wave only, whatever the slot count (33-slot DPF / DCF keys
// start 32 bytes off a line, which cost the 4-level-tiled kernels above half of their bandwidth).  The tile lives in
// LDS as [slot * 2 + half][key], rows padded to 9 entries so that both access patterns are conflict-free within the
// 8-lane groups a 16-byte LDS access is served in.
constexpr int kFlatKeys = 8;
constexpr int kFlatRow = 9;  // padded keys per LDS row
inline size_t flat_lds_bytes(int slots, int waves) { return (size_t)waves * slots * 2 * kFlatRow * sizeof(uint4); }
inline int flat_block(int slots) { return slots <= 34 ? 256 : (slots <= 66 ? 128 : 64); }  // <= ~40 KiB of LDS per workgroup

__device__ __forceinline__ void flat_next(int& key, int& r, int step, int per_key) {
  r += step;
  while (r >= per_key) { r -= per_key; ++key; }
}

template <int SCHEME>
__global__ void relayout_flat_kernel(int in_bits, int slots, const uint4* __restrict__ cws, int64_t nkeys,
                                     uint4* __restrict__ cw_s, uint4* __restrict__ cw_v, uint64_t* __restrict__ tr_bits,
                                     uint4* __restrict__ out_cw, uint32_t* __restrict__ extra, int tr32) {
  extern __shared__ uint4 flat_lds[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  uint4* tile = flat_lds + (size_t)wv * slots * 2 * kFlatRow;
  const int key = lane & 7, l8 = lane >> 3, per_key = 2 * slots;
  const int64_t ntiles = (nkeys + kFlatKeys - 1) / kFlatKeys;
  for (int64_t t = (int64_t)blockIdx.x * nw + wv; t < ntiles; t += (int64_t)gridDim.x * nw) {
    const int64_t k0 = t * kFlatKeys, k = k0 + key;
    const int nk = nkeys - k0 < kFlatKeys ? (int)(nkeys - k0) : kFlatKeys;
    const int npieces = nk * per_key;
    const uint4* src = cws + k0 * per_key;
    {  // flat read of the key-major region
      int kf = 0, r = 0;
      flat_next(kf, r, lane, per_key);
      for (int f = lane; f < npieces; f += 64) {
        tile[r * kFlatRow + kf] = src[f];
        flat_next(kf, r, 64, per_key);
      }
    }
    wave_lds_sync();
    unsigned long long lo = 0ull, hi = 0ull;
    for (int s0 = 0; s0 < slots; s0 += 8) {
      const int slot = s0 + l8;
      if (slot >= slots || key >= nk) continue;
      const uint4 s4 = tile[(2 * slot) * kFlatRow + key], v4 = tile[(2 * slot + 1) * kFlatRow + key];
      if (slot < in_bits) {
        cw_s[(int64_t)slot * nkeys + k] = s4;
        if (SCHEME == S_DCF) cw_v[(int64_t)slot * nkeys + k] = v4;
        if (SCHEME == S_DPF) {
          if (slot < 64) lo |= (unsigned long long)(v4.x & 1u) << slot;
          else hi |= (unsigned long long)(v4.x & 1u) << (slot - 64);
        }
        if (SCHEME == S_HT && slot == in_bits - 1) extra[k] = v4.x & 1u;
      } else {  // slot in_bits of a DPF / DCF key
        if (SCHEME == S_DPF && out_cw) out_cw[k] = s4;
        if (SCHEME == S_DCF) out_cw[k] = v4;
      }
    }
    if (SCHEME == S_DPF) {  // the eight lanes of a key (l8 = 0..7) each saw every eighth level
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {
        lo |= __shfl_xor(lo, m);
        hi |= __shfl_xor(hi, m);
      }
      if (l8 == 0 && key < nk) {
        if (tr32) reinterpret_cast<uint32_t*>(tr_bits)[k] = (uint32_t)lo;
        else tr_bits[k] = lo;
        if (in_bits > 64) tr_bits[nkeys + k] = hi;
      }
    }
    wave_lds_sync();
  }
}

template <int SCHEME>
__global__ void key_major_flat_kernel(int in_bits, int slots, const uint4* __restrict__ cw_s, const uint4* __restrict__ cw_v,
                                      const uint64_t* __restrict__ tr_bits, const uint4* __restrict__ out_cw,
                                      const uint32_t* __restrict__ extra, int64_t nkeys, uint4* __restrict__ cws) {
  extern __shared__ uint4 flat_lds[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  uint4* tile = flat_lds + (size_t)wv * slots * 2 * kFlatRow;
  const int key = lane & 7, l8 = lane >> 3, per_key = 2 * slots;
  const int64_t ntiles = (nkeys + kFlatKeys - 1) / kFlatKeys;
  for (int64_t t = (int64_t)blockIdx.x * nw + wv; t < ntiles; t += (int64_t)gridDim.x * nw) {
    const int64_t k0 = t * kFlatKeys, k = k0 + key;
    const int nk = nkeys - k0 < kFlatKeys ? (int)(nkeys - k0) : kFlatKeys;
    const bool mine = key < nk;
    unsigned long long lo = 0ull, hi = 0ull;
    uint32_t ex = 0;
    if (mine) {
      if (SCHEME == S_DPF) { lo = tr_bits[k]; hi = in_bits > 64 ? tr_bits[nkeys + k] : 0ull; }
      if (SCHEME == S_HT) ex = extra[k] & 1u;
    }
    for (int s0 = 0; s0 < slots; s0 += 8) {
      const int slot = s0 + l8;
      if (slot >= slots) continue;
      uint4 s4 = make_uint4(0, 0, 0, 0), v4 = make_uint4(0, 0, 0, 0);
      if (mine && slot < in_bits) {
        s4 = cw_s[(int64_t)slot * nkeys + k];
        if (SCHEME == S_DCF) v4 = cw_v[(int64_t)slot * nkeys + k];
        else if (SCHEME == S_DPF) v4.x = (uint32_t)(((slot < 64 ? lo : hi) >> (slot & 63)) & 1ull);
        else if (slot == in_bits - 1) v4.x = ex;
      } else if (mine && slot == in_bits) {  // cws[in_bits]: DPF {out_cw, 0}, DCF {0, out_cw}
        if (SCHEME == S_DCF) v4 = out_cw[k];
        else s4 = out_cw[k];
      }
      tile[(2 * slot) * kFlatRow + key] = s4;
      tile[(2 * slot + 1) * kFlatRow + key] = v4;
    }
    wave_lds_sync();
    {  // flat write of the key-major region
      const int npieces = nk * per_key;
      uint4* dst = cws + k0 * per_key;
      int kf = 0, r = 0;
      flat_next(kf, r, lane, per_key);
      for (int f = lane; f < npieces; f += 64) {
        dst[f] = tile[r * kFlatRow + kf];
        flat_next(kf, r, 64, per_key);
      }
    }
    wave_lds_sync();
  }
}

// Host side.  Keys whose slots fill whole lines (slots % 4 == 0: Half-Tree n = 64, 2 KiB per key) take the
// 4-level-tiled kernel (5.1 TB/s there against 4.3 for the flat one, which runs fewer waves per CU on long keys);
// every other shape -- the 33-slot DPF / DCF keys first of all -- takes the flat kernel (DPF n = 32: relayout 4.4 ->
// 5.4 TB/s, key_major 2.9 -> 5.4 TB/s).  FSS_AMD_RELAYOUT = 2 / 1 / 0 forces flat / tiled / one key per lane (A/B).
inline int relayout_mode(int slots) {
  static const int forced = [] { const char* e = tune_env("FSS_AMD_RELAYOUT"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced;
  return slots % 4 == 0 ? 1 : 2;
}
template <int SCHEME, class Ctx>
void launch_relayout(Ctx* ctx, hipStream_t st, int in_bits, int stride, const uint4* cws, int64_t nkeys, uint4* cw_s,
                     uint4* cw_v, uint64_t* tr_bits, uint4* out_cw, uint32_t* extra, int tr32 = 0) {
  const int mode = relayout_mode(stride);
  if (mode >= 2) {
    const int bs = flat_block(stride), nw = bs / 64;
    const int64_t want = ((nkeys + kFlatKeys - 1) / kFlatKeys + nw - 1) / nw, cap = (int64_t)ctx->num_cu * 8;
    hipLaunchKernelGGL(relayout_flat_kernel<SCHEME>, dim3((unsigned)(want < cap ? want : cap)), dim3(bs),
                       flat_lds_bytes(stride, nw), st, in_bits, stride, cws, nkeys, cw_s, cw_v, tr_bits, out_cw, extra, tr32);
  } else if (mode == 1) {
    const int64_t want = (nkeys + 255) / 256, cap = (int64_t)ctx->num_cu * 4;
    hipLaunchKernelGGL(relayout_tiled_kernel<SCHEME>, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, in_bits,
                       stride, cws, nkeys, cw_s, cw_v, tr_bits, out_cw, extra, tr32);
  } else {
    hipLaunchKernelGGL(relayout_kernel<SCHEME>, dim3((unsigned)((nkeys + 63) / 64)), dim3(256), 0, st, in_bits, stride, cws,
                       nkeys, cw_s, cw_v, tr_bits, out_cw, extra, tr32);
  }
}

// Level-major arrays -> key-major raw Cw slots: the inverse of relayout_kernel, for shipping keys generated
// level-major in the reference's 32-byte-slot format; whole slots are written (both 16-byte halves, padding
// zero) as the reference's Gen does (dpf.cuh:147-153).  A wave owns 64 keys and moves them four levels at a
// time through LDS: the level-major side is read as whole lines (64 keys x 16 B per level), and on the
// key-major side eight lanes write the 128 contiguous bytes (four slots) of one key, so a store instruction
// is 8 x 128 B instead of 64 scattered 16-byte pieces (2 -> 4 TB/s).
template <int SCHEME>
__global__ __launch_bounds__(256) void key_major_kernel(int in_bits, const uint4* __restrict__ cw_s,
                                                        const uint4* __restrict__ cw_v,
                                                        const uint64_t* __restrict__ tr_bits,
                                                        const uint4* __restrict__ out_cw,
                                                        const uint32_t* __restrict__ extra, int64_t nkeys,
                                                        uint4* __restrict__ cws) {
  constexpr bool kV = SCHEME == S_DCF;
  __shared__ uint4 tile_s[4][4][65];  // padded: lanes reading different levels of one key hit different banks
  __shared__ uint4 tile_v[kV ? 4 : 1][4][65];
  __shared__ unsigned long long bits[4][2][64];  // DPF: tr bits (two planes); Half-Tree: extra
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int stride = SCHEME == S_HT ? in_bits : in_bits + 1;  // slots per key
  const int64_t ntiles = (nkeys + 63) / 64;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t k0 = tile * 64, k = k0 + lane;
    const bool mine = k < nkeys;
    if (SCHEME == S_DPF) {
      bits[wv][0][lane] = mine ? tr_bits[k] : 0ull;
      bits[wv][1][lane] = (mine && in_bits > 64) ? tr_bits[nkeys + k] : 0ull;
    } else if (SCHEME == S_HT) {
      bits[wv][0][lane] = mine ? (unsigned long long)(extra[k] & 1u) : 0ull;
    }
    for (int ig = 0; ig < stride; ig += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ig + j;
        uint4 s4 = make_uint4(0, 0, 0, 0), v4 = make_uint4(0, 0, 0, 0);
        if (mine && i < in_bits) {
          s4 = cw_s[(int64_t)i * nkeys + k];
          if (kV) v4 = cw_v[(int64_t)i * nkeys + k];
        } else if (mine && i == in_bits && SCHEME != S_HT) {  // cws[in_bits]: DPF {out_cw, 0}, DCF {0, out_cw}
          if (kV) v4 = out_cw[k];
          else s4 = out_cw[k];
        }
        tile_s[wv][j][lane] = s4;
        if (kV) tile_v[wv][j][lane] = v4;
      }
      wave_lds_sync();
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int kk = pass * 8 + (lane >> 3), j = (lane & 7) >> 1, h = lane & 1, i = ig + j;
        if (k0 + kk < nkeys && i < stride) {
          uint4 val;
          if (h == 0) {
            val = tile_s[wv][j][kk];
          } else if (kV) {
            val = tile_v[wv][j][kk];
          } else if (SCHEME == S_DPF) {
            const unsigned long long b = i < 64 ? bits[wv][0][kk] : bits[wv][1][kk];
            val = make_uint4(i < in_bits ? (uint32_t)((b >> (i & 63)) & 1ull) : 0u, 0, 0, 0);
          } else {
            val = make_uint4(i == in_bits - 1 ? (uint32_t)bits[wv][0][kk] : 0u, 0, 0, 0);
          }
          cws[(k0 + kk) * (int64_t)stride * 2 + 2 * i + h] = val;
        }
      }
      wave_lds_sync();
    }
  }
}

struct GenArgs {
  int in_bits, group, pred, x_bytes;
  int64_t nkeys;
  const uint4* s0s;     // [nkeys][2]
  const void* alphas;
  const uint4* betas;
  uint4* cws;           // key-major raw slots
  uint4* ocws;          // Half-Tree
  // level-major outputs (LM kernels): the arrays relayout_kernel would produce
  uint4* cw_s;
  uint4* cw_v;
  uint64_t* tr_bits;
  uint4* out_cw;
  uint32_t* extra;
  int dyn;              // waves take 64-key chunks from the workgroup's LDS counter (WaveQ, fss_dev.hpp)
};

__device__ __forceinline__ void ld4(uint32_t d[4], uint4 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
__device__ __forceinline__ uint4 mk4(const uint32_t d[4]) { return make_uint4(d[0], d[1], d[2], d[3]); }

// fss::Dpf::Gen (include/fss/dpf.cuh:93-159) and fss::Dcf::Gen (include/fss/dcf.cuh:108-194).
// LM: write the level-major arrays of point_eval_gpu.cuh:39-91 directly (lane k's 16-byte store
// at level i lands next to lane k+1's: whole 1 KiB lines per wave) instead of key-major slots.
template <class LT, int SCHEME, bool LM>
__global__ __launch_bounds__(1024) void gen_kernel(const DevCtx K, const uint32_t* __restrict__ te0, GenArgs a) {
  // K is the first kernel argument = offset 0 of the kernarg segment.  Taking &K instead makes the
  // DCF/AES instantiation copy all 1.8 KiB of it to scratch and fetch round keys from there.
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const int n = a.in_bits;
  const Grp g = make_grp(a.group, c);
  WaveQ q(a.dyn, &waveq_ctr_, a.nkeys, 1);
  for (int64_t k; q.next(k);) {
    if (k >= a.nkeys) continue;
    uint32_t s0[4], s1[4], beta[4];
    ld4(s0, a.s0s[2 * k]); ld4(s1, a.s0s[2 * k + 1]);
    ld4(beta, a.betas ? a.betas[k] : make_uint4(0, 0, 0, 0));  // betas == nullptr: beta = 0 (Grotto keys)
    s0[3] &= ~1u; s1[3] &= ~1u; beta[3] &= ~1u;
    uint32_t t0 = 0u, t1 = ~0u;  // masks
    const X128 alpha = load_x(a.alphas, a.x_bytes, k);
    uint4* slot = LM ? nullptr : a.cws + k * (int64_t)(n + 1) * 2;
    unsigned long long tr_lo = 0ull, tr_hi = 0ull;
    Acc v;  // DCF running value (dcf.cuh:115)
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      const uint32_t am = bit_mask(alpha, n - 1 - i);  // ~0 when a_bit
      uint32_t scw[4], vcw_buf[4], tl_cw, tr_cw;
      if constexpr (SCHEME == S_DCF && std::is_same<LT, Lds>::value) {
        // AES: four cipher pairs per level, each the SAME block of both parties side by side (seed blocks of the lose side,
        // value blocks of the lose side, value blocks of the keep side, seed blocks of the keep side), every pair folded
        // into the level's state before the next one starts: between two pairs only the two seeds, the seed correction
        // word, the value correction word and the running value are live -- 17 words, no output words of a cipher among them.  (By party instead -- seed and value block of one party, then
        // of the other -- one party's eight output words waited through the other party's pair: 16 - 17 words, and the
        // kernel kept 27 of its words in scratch; all children at once spilled 2 KiB, round 1.)
        uint32_t lose_cw, vcw[4];
        {
          uint32_t ls0[4], ls1[4];
          aes_mmo2<true>(L, &c->sel02[0][0], &c->sel02[0][0], ~am, ~am, s0, s1, ls0, ls1);
          lose_cw = (ls0[3] ^ ls1[3]) & 1u;
#pragma unroll
          for (int q = 0; q < 4; ++q) scw[q] = ls0[q] ^ ls1[q];
          scw[3] &= ~1u;
        }
        {
          uint32_t lv0[4], lv1[4];
          aes_mmo2<true>(L, &c->sel13[0][0], &c->sel13[0][0], ~am, ~am, s0, s1, lv0, lv1);
          lv0[3] &= ~1u; lv1[3] &= ~1u;
          // v_cw = -v + (v1_lose - v0_lose) [+ beta]   (dcf.cuh:146-154); negated when t1
          Acc vc;
          vc.add(g, lv1); vc.sub(g, lv0);
          const bool add_beta = am ? (a.pred == 0) : (a.pred == 1);
          if (add_beta) {  // reloaded per level (an L2 hit) rather than held across the AES calls
            uint32_t bb[4];
            ld4(bb, a.betas ? a.betas[k] : make_uint4(0, 0, 0, 0));
            bb[3] &= ~1u;
            vc.add(g, bb);
          }
          {
            uint4 vb = v.finish(g, false);
            uint32_t vbuf[4] = {vb.x, vb.y, vb.z, vb.w};
            vc.sub(g, vbuf);
          }
          // v = v - v1_keep + v0_keep + (t1 ? -v_cw : v_cw)   (dcf.cuh:156-159): the last term now (the group's addition commutes)
          const uint4 vcw4 = vc.finish(g, t1 != 0u);
          vcw[0] = vcw4.x; vcw[1] = vcw4.y; vcw[2] = vcw4.z; vcw[3] = vcw4.w;
          if (t1) v.sub(g, vcw); else v.add(g, vcw);
        }
        {
          uint32_t kv0[4], kv1[4];
          aes_mmo2<true>(L, &c->sel13[0][0], &c->sel13[0][0], am, am, s0, s1, kv0, kv1);
          kv0[3] &= ~1u; kv1[3] &= ~1u;
          v.sub(g, kv1); v.add(g, kv0);
        }
        uint32_t keep_cw;
        {
          uint32_t ks0[4], ks1[4];
          aes_mmo2<true>(L, &c->sel02[0][0], &c->sel02[0][0], am, am, s0, s1, ks0, ks1);
          const uint32_t kt0 = ks0[3] & 1u, kt1 = ks1[3] & 1u;
          keep_cw = (kt0 ^ kt1 ^ 1u) & 1u;
          ks0[3] &= ~1u; ks1[3] &= ~1u;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s0[q] = ks0[q] ^ (t0 & scw[q]);
            s1[q] = ks1[q] ^ (t1 & scw[q]);
          }
          t0 = 0u - ((kt0 ^ ((t0 & 1u) & keep_cw)) & 1u);
          t1 = 0u - ((kt1 ^ ((t1 & 1u) & keep_cw)) & 1u);
        }
        tl_cw = am ? lose_cw : keep_cw;
        tr_cw = am ? keep_cw : lose_cw;
        vcw_buf[0] = vcw[0]; vcw_buf[1] = vcw[1]; vcw_buf[2] = vcw[2]; vcw_buf[3] = (vcw[3] & ~1u) | tr_cw;
      } else {
        uint32_t s0l[4], s0r[4], s1l[4], s1r[4];
        uint32_t v0l[4], v0r[4], v1l[4], v1r[4];
        if (SCHEME == S_DCF) {
          // two blocks at a time: eight interleaved AES states do not fit 128 VGPRs
          prg_gen4(L, c, s0, s0l, v0l, s0r, v0r);
          prg_gen4(L, c, s1, s1l, v1l, s1r, v1r);
          v0l[3] &= ~1u; v0r[3] &= ~1u; v1l[3] &= ~1u; v1r[3] &= ~1u;
        } else {
          prg_gen2(L, c, s0, s0l, s0r);
          prg_gen2(L, c, s1, s1l, s1r);
        }
        const uint32_t t0l = s0l[3] & 1u, t0r = s0r[3] & 1u, t1l = s1l[3] & 1u, t1r = s1r[3] & 1u;
        s0l[3] &= ~1u; s0r[3] &= ~1u; s1l[3] &= ~1u; s1r[3] &= ~1u;
        const uint32_t ab = am & 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) scw[q] = am ? (s0l[q] ^ s1l[q]) : (s0r[q] ^ s1r[q]);  // lose side
        tl_cw = t0l ^ t1l ^ ab ^ 1u; tr_cw = t0r ^ t1r ^ ab;

        vcw_buf[0] = tr_cw; vcw_buf[1] = 0; vcw_buf[2] = 0; vcw_buf[3] = 0;
        if (SCHEME == S_DCF) {
          // v_cw = -v + (v1_lose - v0_lose) [+ beta]   (dcf.cuh:146-154); negated when t1
          Acc vc;
          uint32_t lose1[4], lose0[4], keep1[4], keep0[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            lose1[q] = am ? v1l[q] : v1r[q]; lose0[q] = am ? v0l[q] : v0r[q];
            keep1[q] = am ? v1r[q] : v1l[q]; keep0[q] = am ? v0r[q] : v0l[q];
          }
          vc.add(g, lose1); vc.sub(g, lose0);
          const bool add_beta = am ? (a.pred == 0) : (a.pred == 1);
          if (add_beta) vc.add(g, beta);
          // vc -= v
          {
            uint4 vb = v.finish(g, false);
            uint32_t vbuf[4] = {vb.x, vb.y, vb.z, vb.w};
            vc.sub(g, vbuf);
          }
          uint4 vcw4 = vc.finish(g, t1 != 0u);
          uint32_t vcw[4] = {vcw4.x, vcw4.y, vcw4.z, vcw4.w};
          // v = v - v1_keep + v0_keep + (t1 ? -v_cw : v_cw)   (dcf.cuh:156-159)
          v.sub(g, keep1); v.add(g, keep0);
          if (t1) v.sub(g, vcw); else v.add(g, vcw);
          vcw_buf[0] = vcw[0]; vcw_buf[1] = vcw[1]; vcw_buf[2] = vcw[2]; vcw_buf[3] = (vcw[3] & ~1u) | tr_cw;
        }
        // keep side, corrected (dpf.cuh:124-144)
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
      }
      if (LM) {
        a.cw_s[(int64_t)i * a.nkeys + k] = make_uint4(scw[0], scw[1], scw[2], (scw[3] & ~1u) | tl_cw);
        if (SCHEME == S_DCF) a.cw_v[(int64_t)i * a.nkeys + k] = mk4(vcw_buf);
        else if (i < 64) tr_lo |= (unsigned long long)tr_cw << i;
        else tr_hi |= (unsigned long long)tr_cw << (i - 64);
      } else {
        slot[2 * i] = make_uint4(scw[0], scw[1], scw[2], (scw[3] & ~1u) | tl_cw);
        slot[2 * i + 1] = mk4(vcw_buf);
      }
    }
    Acc y;
    if (SCHEME == S_DCF) {
      // v_cw_np1 = s1 - s0 - v, negated when t1 (dcf.cuh:191-193)
      uint4 vb = v.finish(g, false);
      uint32_t vbuf[4] = {vb.x, vb.y, vb.z, vb.w};
      y.add(g, s1); y.sub(g, s0); y.sub(g, vbuf);
      if (LM) {
        a.out_cw[k] = y.finish(g, t1 != 0u);
      } else {
        slot[2 * n] = make_uint4(0, 0, 0, 0);
        slot[2 * n + 1] = y.finish(g, t1 != 0u);
      }
    } else {
      // beta - s0 + s1, negated when t1 (dpf.cuh:156-158)
      y.add(g, beta); y.sub(g, s0); y.add(g, s1);
      if (LM) {
        a.out_cw[k] = y.finish(g, t1 != 0u);
        a.tr_bits[k] = tr_lo;
        if (n > 64) a.tr_bits[a.nkeys + k] = tr_hi;
      } else {
        slot[2 * n] = y.finish(g, t1 != 0u);
        slot[2 * n + 1] = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// fss::HalfTreeDpf::Gen (include/fss/half_tree_dpf.cuh:68-175).
template <class LT, bool LM>
__global__ __launch_bounds__(1024) void ht_gen_kernel(const DevCtx K, const uint32_t* __restrict__ te0, GenArgs a) {
  const DevCtx* c = first_kernarg(K);
  FSS_WAVEQ_PROLOGUE();
  FSS_PRG_PROLOGUE(LT, te0);
  const int n = a.in_bits;
  const Grp g = make_grp(a.group, c);
  // the two parties' hashes of a level side by side: two interleaved cipher streams (aes_mmo2) keep the LDS pipe busy
  // through each other's arithmetic (one block at a time: 20.1 ms for 2^24 keys of n = 64; paired: 19.8)
  auto hash2 = [&](const uint32_t na[4], const uint32_t nb[4], uint32_t ha[4], uint32_t hb[4]) {
    uint32_t ia[4] = {na[0] ^ c->hash_key[0], na[1] ^ c->hash_key[1], na[2] ^ c->hash_key[2], na[3] ^ c->hash_key[3]};
    uint32_t ib[4] = {nb[0] ^ c->hash_key[0], nb[1] ^ c->hash_key[1], nb[2] ^ c->hash_key[2], nb[3] ^ c->hash_key[3]};
    prg_hash2(L, c, ia, ib, ha, hb);
  };
  WaveQ q(a.dyn, &waveq_ctr_, a.nkeys, 1);
  for (int64_t k; q.next(k);) {
    if (k >= a.nkeys) continue;
    uint32_t n0[4], n1[4], beta[4];
    ld4(n0, a.s0s[2 * k]); ld4(n1, a.s0s[2 * k + 1]);
    ld4(beta, a.betas ? a.betas[k] : make_uint4(0, 0, 0, 0));
    n0[3] &= ~1u; n1[3] |= 1u; beta[3] &= ~1u;
    const X128 alpha = load_x(a.alphas, a.x_bytes, k);
    uint4* slot = LM ? nullptr : a.cws + k * (int64_t)n * 2;
#pragma unroll 1
    for (int i = 0; i < n - 1; ++i) {
      const uint32_t am = bit_mask(alpha, n - 1 - i);
      uint32_t h0[4], h1[4], cw[4];
      hash2(n0, n1, h0, h1);
      const uint32_t t0 = lsb_mask(n0[3]), t1 = lsb_mask(n1[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cw[q] = h0[q] ^ h1[q] ^ (~am & (n0[q] ^ n1[q]));  // :84-85
        uint32_t a0 = h0[q] ^ (am & n0[q]) ^ (t0 & cw[q]);   // :93-101
        uint32_t a1 = h1[q] ^ (am & n1[q]) ^ (t1 & cw[q]);
        n0[q] = a0; n1[q] = a1;
      }
      if (LM) {
        a.cw_s[(int64_t)i * a.nkeys + k] = mk4(cw);
      } else {
        slot[2 * i] = mk4(cw);
        slot[2 * i + 1] = make_uint4(0, 0, 0, 0);
      }
    }
    {
      const uint32_t an = (uint32_t)alpha.lo & 1u;
      const uint32_t t0 = lsb_mask(n0[3]), t1 = lsb_mask(n1[3]);
      uint32_t h00[4], h01[4], h10[4], h11[4];
      uint32_t pz[4], po[4];  // the node with its control bit cleared / set (half_tree_dpf.cuh:108-121)
      pz[0] = po[0] = n0[0]; pz[1] = po[1] = n0[1]; pz[2] = po[2] = n0[2];
      pz[3] = n0[3] & ~1u; po[3] = n0[3] | 1u;
      hash2(pz, po, h00, h01);
      pz[0] = po[0] = n1[0]; pz[1] = po[1] = n1[1]; pz[2] = po[2] = n1[2];
      pz[3] = n1[3] & ~1u; po[3] = n1[3] | 1u;
      hash2(pz, po, h10, h11);
      const uint32_t l00 = h00[3] & 1u, l01 = h01[3] & 1u, l10 = h10[3] & 1u, l11 = h11[3] & 1u;
      h00[3] &= ~1u; h01[3] &= ~1u; h10[3] &= ~1u; h11[3] &= ~1u;
      uint32_t hcw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) hcw[q] = an ? (h00[q] ^ h10[q]) : (h01[q] ^ h11[q]);  // :123-125
      const uint32_t lcw0 = l00 ^ l10 ^ (an ^ 1u), lcw1 = l01 ^ l11 ^ an;                // :132-133
      if (LM) {
        a.cw_s[(int64_t)(n - 1) * a.nkeys + k] = make_uint4(hcw[0], hcw[1], hcw[2], (hcw[3] & ~1u) | lcw0);
        a.extra[k] = lcw1;
      } else {
        slot[2 * (n - 1)] = make_uint4(hcw[0], hcw[1], hcw[2], (hcw[3] & ~1u) | lcw0);
        slot[2 * (n - 1) + 1] = make_uint4(lcw1, 0, 0, 0);
      }
      uint32_t leaf0[4], leaf1[4];
      const uint32_t lcw_an = an ? lcw1 : lcw0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        leaf0[q] = an ? h01[q] : h00[q];
        leaf1[q] = an ? h11[q] : h10[q];
      }
      leaf0[3] |= an ? l01 : l00;
      leaf1[3] |= an ? l11 : l10;
      uint32_t lc[4] = {hcw[0], hcw[1], hcw[2], (hcw[3] & ~1u) | lcw_an};
#pragma unroll
      for (int q = 0; q < 4; ++q) { leaf0[q] ^= t0 & lc[q]; leaf1[q] ^= t1 & lc[q]; }
      const bool neg = (leaf1[3] & 1u) != 0;
      leaf0[3] &= ~1u; leaf1[3] &= ~1u;
      Acc y;  // beta - leaf0 + leaf1, negated when leaf1's control bit (:168-172)
      y.add(g, beta); y.sub(g, leaf0); y.add(g, leaf1);
      a.ocws[k] = y.finish(g, neg);
    }
  }
}

}  // namespace fssamd
