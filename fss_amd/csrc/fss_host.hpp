// Host-side plumbing shared by the translation units of libfss_amd.so (fss_amd.hip, fss_abi_evalall.hip,
// fss_abi_gen.hip, fss_abi_fused*.hip): the context, error recording, launch geometry and work-distribution heuristics,
// PRG dispatch.  The library is split so that hipcc compiles the kernel families in parallel.
#pragma once
#include "../../include/fss_amd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "fss_dev.hpp"
#include "fss_point.hpp"
#include "fss_quad.hpp"

using namespace fssamd;

// ------------------------------------------------------------------ context
// What per-stream state (scratch buffers, WaveQ's tail counters) is keyed by.  A stream handle names one real stream
// except hipStreamPerThread, one handle value that stands for a different stream in every host thread: there the
// calling thread is part of the key, so two threads launching on "their" per-thread stream never share a counter pair
// or a scratch buffer.  (The null stream is one real stream for the whole process -- this library is not compiled
// with -fgpu-default-stream=per-thread -- so launches on it run one after the other and may share.)
using StreamKey = std::pair<void*, std::thread::id>;
inline StreamKey stream_key(void* stream) {
  return StreamKey(stream, stream == (void*)hipStreamPerThread ? std::this_thread::get_id() : std::thread::id());
}

struct fss_amd_ctx {
  int device;
  int num_cu;
  uint32_t* te0;  // device: Te0 table, 1 KiB, followed by the device copy of `host`
  DevCtx host;    // round keys etc., passed to kernels by value
  int prg;        // FSS_AMD_PRG_AES128_MMO or FSS_AMD_PRG_CHACHA
  const char* last_kernel;
  int last_grid, last_block;
  // fss_amd_ctx_scratch: one grow-only device buffer per stream that asked for one
  struct Scratch { void* p = nullptr; size_t bytes = 0; };
  std::mutex scratch_mu;
  std::map<StreamKey, Scratch> scratch;
  // fss_amd_*_eval_points under FSS_AMD_POINTS_AUTO (opt-in: fss_amd_ctx_set_points_plan; a fresh context walks
  // plainly and its eval calls allocate nothing): the node table + stage buffers of the shared-prefix plan, one
  // grow-only buffer per stream, NOT the buffer fss_amd_ctx_scratch hands out (a caller may have parked the very seed it
  // passes in that one).  auto_mu is held from taking the buffer until the plan's last kernel is enqueued: two host
  // threads on one stream then enqueue "table, walk" pairs whole.
  int points_plan = FSS_AMD_POINTS_PLAIN;
  std::mutex auto_mu;
  std::map<StreamKey, Scratch> prefix_ws;
  // (auto_mu held) the stream's buffer grown to `bytes`; nullptr when the device has no room (the caller walks plainly).
  // Stream-ordered (hipMallocAsync / hipFreeAsync on the caller's stream): the old buffer goes away after the work already
  // queued on it, the new one exists before the kernels enqueued next -- no device-wide wait inside an eval call, and
  // nothing a capture another thread has open in global mode would be invalidated by (hipMalloc / hipFree are).
  void* prefix_buffer(void* stream, size_t bytes) {
    Scratch& s = prefix_ws[stream_key(stream)];
    if (s.bytes < bytes || !s.p) {
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
      if (s.p && hipFreeAsync(s.p, st) != hipSuccess) (void)hipGetLastError();
      s.p = nullptr;
      s.bytes = 0;
      if (hipMallocAsync(&s.p, bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        s.p = nullptr;
        return nullptr;
      }
      s.bytes = bytes;
    }
    return s.p;
  }
  uint64_t* diag_stamps = nullptr;  // FSS_DIAG_STAMPS builds: 4 x uint64 per workgroup of the last point-walk launch
  // WaveQ's launch-wide tail counters: one {counter, waves done} pair per real stream that launched a dynamic walk
  // (launches on one stream run one after the other, and the kernel hands its pair back zeroed; StreamKey tells the
  // per-thread streams behind hipStreamPerThread apart); nothing is allocated after creation, so a first launch may
  // sit inside a stream capture
  static constexpr int kTailPairs = 1024;
  uint32_t* tail_pairs = nullptr;   // device: kTailPairs x 2 words, zero
  std::mutex tail_mu;
  std::map<StreamKey, int> tail_slot;
  std::vector<int> tail_free;  // pairs given back by fss_amd_ctx_scratch_release (a stream that is going away)
  int tail_next = 0;
  uint32_t* tail_pair(void* stream) {
    const StreamKey key = stream_key(stream);
    std::lock_guard<std::mutex> lk(tail_mu);
    auto it = tail_slot.find(key);
    if (it == tail_slot.end()) {
      int idx;
      if (!tail_free.empty()) { idx = tail_free.back(); tail_free.pop_back(); }
      else if (tail_next < kTailPairs) idx = tail_next++;
      else return nullptr;  // more live streams than pairs: those launch without a tail
      it = tail_slot.emplace(key, idx).first;
    }
    return tail_pairs + 2 * it->second;
  }
  // The stream (this thread's, for hipStreamPerThread) is going away and the device is idle: its pair can serve another.
  void tail_release(void* stream) {
    std::lock_guard<std::mutex> lk(tail_mu);
    auto it = tail_slot.find(stream_key(stream));
    if (it == tail_slot.end()) return;
    tail_free.push_back(it->second);
    tail_slot.erase(it);
  }
};

inline thread_local int g_last_hip = 0;  // one instance across the translation units

#define HIP_TRY(expr)                    \
  do {                                   \
    hipError_t e_ = (expr);              \
    if (e_ != hipSuccess) {              \
      g_last_hip = (int)e_;              \
      return FSS_AMD_EHIP;               \
    }                                    \
  } while (0)

namespace {


// The entry points run on the context's device and leave the caller's current device as they found it.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    else if (err == hipSuccess) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define FSS_ON_DEVICE(ctx)             \
  DeviceGuard guard_((ctx)->device);   \
  HIP_TRY(guard_.err)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// group id in range and, for Uint<T, mod> with a context modulus, the modulus representable in T
bool group_ok(const fss_amd_ctx* ctx, int g) {
  if (g < 0 || g > 5) return false;
  const uint64_t m = ctx->host.uint_mod;
  if (m == 0 || g < G_U8 || g > G_U32) return true;
  return m <= (1ull << (g == G_U8 ? 8 : (g == G_U16 ? 16 : 32)));
}
// Every int4 array of the interface is read and written with 16-byte accesses; null (optional) pointers pass.
template <typename... P>
bool aligned16(const P*... p) {
  return (((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ...);
}
inline bool aligned_to(const void* p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(bytes - 1)) == 0; }
bool x_ok(int in_bits, int x_bytes) {
  if (in_bits < 1) return false;
  if (x_bytes == 1) return in_bits <= 8;
  if (x_bytes == 2) return in_bits <= 16;
  if (x_bytes == 4) return in_bits <= 32;
  if (x_bytes == 8) return in_bits <= 64;
  if (x_bytes == 16) return in_bits <= 128;  // two little-endian uint64 per value (__uint128_t)
  return false;
}

// Tunables (read once, and only when the process sets FSS_AMD_TUNE=1 -- tune_env, fss_dev.hpp): FSS_AMD_ILP in {1,2},
// FSS_AMD_BS in {256,512,1024}, FSS_AMD_WGS_PER_CU, ...  A default process runs the plan below whatever its environment.
struct Tune {
  int ilp = 2, bs = 1024, wgs_per_cu = 0;  // ilp 2 = two staggered AES streams per lane (aes_mmo2)
  bool ilp_set = false;  // FSS_AMD_ILP given: do not second-guess it for small batches
  int dyn = -1;  // FSS_AMD_DYN: 1 / 0 force dynamic / static work distribution; default: by batch size
  int rot_prio = -1;  // FSS_AMD_ROTPRIO: 0 none / 1 turns / 2 by progress; default: see launch_points_k
  int prio_slack = 0;  // FSS_AMD_PRIO_SLACK
  int prio_every = 3;  // FSS_AMD_PRIO_EVERY: rank_prio exchanges when (levels done & this) == 0
  int static_prio = 2;  // FSS_AMD_STATIC_PRIO: wave priorities of launches with fixed shares (0 none / 1 turns / 2 by progress)
  int64_t dyn_min = 65536;  // FSS_AMD_DYN_MIN: AES blocks per resident wave from which shares are handed out dynamically
  int tail_pct = 8;  // FSS_AMD_TAIL_PCT: share of a dynamic walk's chunks in the launch-wide tail queue (0: none)
  // "two, then one" (eval_points_kernel): what runs ONE evaluation per lane at the end of a two-per-lane launch --
  int mix_mid_pct = 6;   // FSS_AMD_MIX_MID_PCT: dynamic launches: % of the batch dealt per workgroup, before the tail_pct % launch-wide
  int mix_tail_units = 2;  // FSS_AMD_MIX_TAIL_UNITS: fixed shares: chunks of 64 per wave (0 with either: two per lane throughout)
  Tune() {
    if (const char* e = tune_env("FSS_AMD_MIX_MID_PCT")) { mix_mid_pct = atoi(e); if (mix_mid_pct < 0 || mix_mid_pct > 90) mix_mid_pct = 6; }
    if (const char* e = tune_env("FSS_AMD_MIX_TAIL_UNITS")) { mix_tail_units = atoi(e); if (mix_tail_units < 0) mix_tail_units = 2; }
    if (const char* e = tune_env("FSS_AMD_TAIL_PCT")) { tail_pct = atoi(e); if (tail_pct < 0 || tail_pct > 100) tail_pct = 8; }
    if (const char* e = tune_env("FSS_AMD_ROTPRIO")) { rot_prio = atoi(e); if (rot_prio < 0 || rot_prio > 2) rot_prio = 1; }
    if (const char* e = tune_env("FSS_AMD_DYN")) dyn = atoi(e) != 0;
    if (const char* e = tune_env("FSS_AMD_STATIC_PRIO")) { static_prio = atoi(e); if (static_prio < 0 || static_prio > 2) static_prio = 2; }
    if (const char* e = tune_env("FSS_AMD_DYN_MIN")) { dyn_min = atoll(e); if (dyn_min < 1) dyn_min = 65536; }
    if (const char* e = tune_env("FSS_AMD_PRIO_EVERY")) { prio_every = atoi(e); if (prio_every != 0 && prio_every != 1 && prio_every != 3 && prio_every != 7) prio_every = 3; }
    if (const char* e = tune_env("FSS_AMD_PRIO_SLACK")) { prio_slack = atoi(e); if (prio_slack < 0) prio_slack = 0; }
    if (const char* e = tune_env("FSS_AMD_ILP")) { ilp = atoi(e); ilp_set = true; }
    if (const char* e = tune_env("FSS_AMD_BS")) bs = atoi(e);
    if (const char* e = tune_env("FSS_AMD_WGS_PER_CU")) wgs_per_cu = atoi(e);
    if (ilp != 1 && ilp != 2) ilp = 2;
    if (bs != 256 && bs != 512 && bs != 1024) bs = 1024;
  }
};
const Tune& tune() {
  static Tune t;
  return t;
}

// Grid of a persistent launch: one workgroup per resident slot, fewer when the batch is smaller.
template <typename K>
int persistent_grid(fss_amd_ctx* ctx, K kernel, int bs, int64_t threads_needed, int* grid) {
  // resident workgroups per CU of this kernel at this block size: asked once per (kernel, block size)
  static std::mutex mu;
  static std::unordered_map<uint64_t, int> cache;
  const uint64_t key = (uint64_t)reinterpret_cast<uintptr_t>(reinterpret_cast<const void*>(kernel)) * 4099u + (uint64_t)bs;
  int occ = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) occ = it->second;
  }
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, bs, 0));
    if (occ < 1) occ = 1;
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = occ;
  }
  if (tune().wgs_per_cu > 0 && tune().wgs_per_cu < occ) occ = tune().wgs_per_cu;
  int64_t want = (threads_needed + bs - 1) / bs;
  int64_t cap = (int64_t)ctx->num_cu * occ;
  *grid = (int)(want < cap ? want : cap);
  if (*grid < 1) *grid = 1;
  return FSS_AMD_OK;
}

// Dynamic (per-workgroup LDS counter, WaveQ in fss_dev.hpp) or static distribution of `units` units taken
// `per_chunk` at a time over `grid` workgroups of `bs` lanes: dynamic when a wave gets at least 4 chunks.
int use_dyn(int64_t units, int per_chunk, int grid, int bs) {
  if (tune().dyn >= 0) return tune().dyn;
  const int64_t nchunks = (units + per_chunk - 1) / per_chunk, waves = (int64_t)grid * (bs / 64);
  return nchunks >= 4 * waves ? 1 : 0;
}

// WaveQ's launch-wide tail for a dynamic launch of `units` units in chunks of `per_chunk` over `grid` workgroups:
// the stream's counter pair and the number of chunks dealt out per workgroup before it takes over.
// `tail_chunks` < 0: tail_pct % of the chunks; otherwise that many chunks (the caller's own split).
inline void set_tail(fss_amd_ctx* ctx, hipStream_t st, int dyn, int64_t units, int64_t per_chunk, int grid, uint32_t** gq,
                     int64_t* static_chunks, int64_t tail_chunks = -1) {
  *gq = nullptr;
  *static_chunks = 0;
  if (!dyn || tune().tail_pct <= 0 || !ctx->tail_pairs) return;
  // A launch recorded into a graph may be replayed later on any stream, next to other work of the stream it was
  // captured on: its counter pair would no longer be its own.  Captured launches go without the tail.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
  if (cap != hipStreamCaptureStatusNone) return;
  const int64_t nchunks = (units + per_chunk - 1) / per_chunk;
  int64_t stat = nchunks - (tail_chunks < 0 ? nchunks * tune().tail_pct / 100 : tail_chunks);
  if (stat < 0) stat = 0;
  stat -= stat % grid;
  if (stat < nchunks && (*gq = ctx->tail_pair((void*)st)) != nullptr) *static_chunks = stat;
}

template <typename K, typename... Args>
int launch_grid(fss_amd_ctx* ctx, K kernel, const char* name, int grid, int bs, hipStream_t st, Args... args) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(bs), 0, st, args...);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = name;
  ctx->last_grid = grid;
  ctx->last_block = bs;
  return FSS_AMD_OK;
}

template <typename K, typename... Args>
int launch_persistent(fss_amd_ctx* ctx, K kernel, const char* name, int bs, int64_t threads_needed, hipStream_t st,
                      Args... args) {
  int grid = 0;
  int rc = persistent_grid(ctx, kernel, bs, threads_needed, &grid);
  if (rc != FSS_AMD_OK) return rc;
  return launch_grid(ctx, kernel, name, grid, bs, st, args...);
}

// Instantiate CALL once per PRG (LT = Lds: AES-128-MMO with LDS tables; LT = ChaChaL) and run the one
// the context is set to.
#define FSS_BY_PRG(ctx, CALL)                                  \
  ((ctx)->prg == FSS_AMD_PRG_CHACHA ? [&] { using LT = ChaChaL; return CALL; }() \
                                    : [&] { using LT = Lds; return CALL; }())

#define FSS_BY_PRG_DO(ctx, ...)                      \
  do {                                               \
    if ((ctx)->prg == FSS_AMD_PRG_CHACHA) {          \
      using LT = ChaChaL;                            \
      __VA_ARGS__;                                   \
    } else {                                         \
      using LT = Lds;                                \
      __VA_ARGS__;                                   \
    }                                                \
  } while (0)

struct SharedKey {
  const uint32_t *seed = nullptr, *cws = nullptr, *ocw = nullptr;
};

template <typename K>
int launch_points_k(fss_amd_ctx* ctx, K kernel, const char* name, int ilp, int bs, const SharedKey& k, EvalArgs a,
                    hipStream_t st, bool tail_ok, bool mix_ok, int blocks_per_eval) {
  int grid = 0;
  int rc = persistent_grid(ctx, kernel, bs, (a.n + ilp - 1) / ilp, &grid);
  if (rc != FSS_AMD_OK) return rc;
  // dynamic distribution from 65 536 AES blocks per resident wave on (DPF n = 32: 2^23 points on the whole chip; DCF
  // n = 32 and Half-Tree n = 64: 2^22); below that fixed shares with progress-ranked wave priorities (rank_prio) win:
  // the waves of a SIMD finish together and there is no ragged last chunk (DPF 2^20 points: 0.344 -> 0.307 ms, 2^21:
  // 0.624 -> 0.600, 2^22: 1.217 -> 1.194 against the earlier policy -- dynamic from 16 384 blocks, rotating priorities
  // below; profiles/r2/points_prio_ab.txt)
  a.dyn = tune().dyn >= 0 ? tune().dyn : (a.n * blocks_per_eval >= tune().dyn_min * grid * (bs / 64) ? 1 : 0);
  a.rot_prio = tune().rot_prio >= 0 ? tune().rot_prio : (a.dyn ? 0 : tune().static_prio);
  a.prio_slack = tune().prio_slack;
  a.prio_every = tune().prio_every;
  // the last tail_pct % of the chunks go to whichever workgroup asks first (WaveQ); shares stay whole rounds of the grid.
  // Only where a chunk is long enough: in the tail every chunk costs a request to one device-memory word, and those
  // serialise at a few tens of millions per second -- DPF n = 8, one evaluation per lane: 1.20 -> 1.39 ms WITH the tail;
  // from about 24 AES blocks per lane and chunk on it pays (n = 24: 3.53 -> 3.49 ms, n = 32: 4.66 -> 4.62).
  a.gq = nullptr;
  a.static_chunks = 0;
  a.n_body = 0;
  a.tail_on_body = 0;
  if (ilp == 2 && !mix_ok) {
    // two per lane throughout (short walks: a one-evaluation chunk would be too short to end the launch on)
    a.n_body = a.n;
    a.tail_on_body = 1;
    if (tail_ok) set_tail(ctx, st, a.dyn, a.n, 128, grid, &a.gq, &a.static_chunks);
  } else if (ilp == 2) {
    // two per lane over [0, n_body), one per lane over the rest (eval_points_kernel)
    const int64_t units = (a.n + 63) / 64, waves = (int64_t)grid * (bs / 64);  // units of 64 evaluations
    int64_t tail_units = 0;
    if (a.dyn) tail_units = units * (tune().mix_mid_pct + tune().tail_pct) / 100;
    else tail_units = units >= 2 * tune().mix_tail_units * waves ? tune().mix_tail_units * waves : 0;
    // FSS_AMD_MIX_MID_PCT + FSS_AMD_TAIL_PCT may exceed 100: the one-per-lane region is then the whole batch, never more
    if (tail_units > units) tail_units = units;
    if (tail_units < 0) tail_units = 0;
    int64_t body_units = units - tail_units;
    body_units -= body_units & 1;
    if (body_units < 0) body_units = 0;
    a.n_body = tail_units > 0 ? body_units * 64 : a.n;
    if (a.n_body > a.n) a.n_body = a.n;
    // of the one-per-lane region the last tail_pct % OF THE BATCH go to whichever workgroup asks first
    if (a.n_body < a.n) set_tail(ctx, st, a.dyn, a.n - a.n_body, 64, grid, &a.gq, &a.static_chunks, units * tune().tail_pct / 100);
  } else if (tail_ok) {
    set_tail(ctx, st, a.dyn, a.n, 64, grid, &a.gq, &a.static_chunks);
  }
#if FSS_DIAG_STAMPS
  a.stamps = ctx->diag_stamps;
#endif
  return launch_grid(ctx, kernel, name, grid, bs, st, ctx->host, ctx->te0, k.seed, k.cws, k.ocw, a);
}

template <int SCHEME, bool SHARED, bool PREFIX = false>
int launch_points(fss_amd_ctx* ctx, const char* name, const SharedKey& k, const EvalArgs& a, hipStream_t st) {
  if (a.n == 0) return FSS_AMD_OK;
  const int bs = PREFIX ? 1024 : tune().bs;  // the shared-prefix walk is built for 1024 only
  int ilp = tune().ilp;
  // two evaluations per lane only pay when there are enough to give every resident wave a chunk of 128
  // (2^18 points: 0.10 ms with one per lane, 0.17 ms with two and half of the waves idle)
  const int64_t resident_waves = (int64_t)ctx->num_cu * (ctx->prg == FSS_AMD_PRG_CHACHA ? 2 : 1) * (bs / 64);
  if (!tune().ilp_set && a.n < 128 * resident_waves) ilp = 1;
  // a DCF evaluation already runs two interleaved AES streams (seed and value block of the taken child): a second
  // evaluation per lane only adds registers -- one per lane is ahead at every batch size (2^20 points: 0.616 ms
  // against 0.663, 2^22: 2.43 against 2.48, 2^24 even; profiles/r2/points_ab.txt)
  if (!tune().ilp_set && SCHEME == S_DCF) ilp = 1;
  // DPF and Half-Tree: with the launch-wide tail queue the finer chunks of one evaluation per lane are worth more than
  // the second interleaved stream wherever a one-evaluation chunk is still long enough for that queue (>= 24 levels
  // walked): one shared DPF key 2^24 points 4.62 against 4.69 ms, 2^22 1.186 against 1.237, equal at 2^19 .. 2^21;
  // independent DPF keys 5.28 against 5.34 ms, Half-Tree keys 10.155 against 10.176 (profiles/r2/keys_clock.txt,
  // tail_queue_ab.txt).  Short walks (n < 24, the shared-prefix walks) keep two.
  // Round 3: not either-or.  The two-per-lane kernel walks most of the batch two per lane and ENDS on one-per-lane
  // chunks ("two, then one", eval_points_kernel): the interleaved streams' busier LDS for the body, the finer chunks
  // for the end (launch_points_k sets the split).  FSS_AMD_ILP=1 / 2 with FSS_AMD_MIX_*=0 give the pure forms back.
  const int walked = a.in_bits - (PREFIX ? a.start_level : 0);
  const int blocks_per_eval = walked * (SCHEME == S_DCF ? 2 : 1);
  static const int tail_minwork = [] { const char* e = tune_env("FSS_AMD_TAIL_MINWORK"); return e ? atoi(e) : 24; }();
  const bool tail_ok = blocks_per_eval * ilp >= tail_minwork;  // a chunk of `ilp` evaluations per lane is long enough for the launch-wide queue
  const bool mix_ok = blocks_per_eval >= tail_minwork && (tune().mix_mid_pct > 0 || tune().mix_tail_units > 0);  // ... and so is a one-evaluation chunk
  // fewer than four units of 64 per resident wave: nothing to split; one per lane as before (equal at 2^19 points, r2)
  if (!tune().ilp_set && mix_ok && a.n < 256 * resident_waves) ilp = 1;
  // ChaCha (pure VALU, no LDS to keep busy) keeps round 2's choice: one per lane from 24 walked levels on
  if (!tune().ilp_set && ctx->prg == FSS_AMD_PRG_CHACHA && walked >= 24) ilp = 1;
#define FSS_L(I, B) \
  if (ilp == I && bs == B)                                                                                      \
  return FSS_BY_PRG(ctx, (launch_points_k(ctx, eval_points_kernel<LT, SCHEME, SHARED, I, B, PREFIX>, name, I, B, k, a, st, tail_ok, mix_ok, blocks_per_eval)))
  FSS_L(1, 1024);
  FSS_L(2, 1024);
  if constexpr (!PREFIX) {  // (discarded for PREFIX: those kernels exist for 1024 lanes only)
    FSS_L(1, 512);
    FSS_L(2, 512);
    FSS_L(1, 256);
    FSS_L(2, 256);
  }
#undef FSS_L
  return FSS_AMD_EINVAL;
}

// Small shared-key batches: one walk per quad of lanes (points_quad_kernel, fss_quad.hpp) runs a level in ~0.9 us
// instead of the 2 - 3 us of a lone wave.  DPF n = 32: 30 us per launch up to 2^16 points (a per-lane walk: 85 us up
// to 2^18), 56 us at 2^17, 109 us at 2^18.  FSS_AMD_QUAD_POINTS_MAX = largest batch that takes it (0: never).
constexpr int kQuadPointsMax = 131072;
inline int64_t quad_points_max() {
  static const int64_t v = [] {
    const char* e = tune_env("FSS_AMD_QUAD_POINTS_MAX");
    return e ? (int64_t)atoll(e) : (int64_t)kQuadPointsMax;
  }();
  return v;
}
// DCF with the AES tables, a batch small enough that eight lanes per walk still fit the chip in one pass: the octet form of
// the quad kernels (QuadPrg<Lds, SCHEME>; 1 .. 2^15 walks: 58 -> 37 - 42 us at n = 32).  FSS_AMD_QUAD_OCT=0 turns it off (A/B).
inline bool quad_octets_pay(const fss_amd_ctx* ctx, int scheme, int64_t walks) {
  static const bool on = [] { const char* e = tune_env("FSS_AMD_QUAD_OCT"); return !(e && atoi(e) == 0); }();
  return on && scheme == S_DCF && ctx->prg != FSS_AMD_PRG_CHACHA && walks * 8 <= (int64_t)ctx->num_cu * 1024;
}
template <int SCHEME>
int launch_points_quad(fss_amd_ctx* ctx, const char* name, const SharedKey& k, const EvalArgs& a, hipStream_t st) {
  // a quad of lanes per walk; DCF on the AES tables: an octet while the octets fit the chip's lanes in one pass (fss_quad.hpp)
  const bool oct = quad_octets_pay(ctx, SCHEME, a.n);
  const int64_t lanes = a.n * (oct ? 8 : 4);
  const int bs = lanes >= 65536 ? 1024 : 256;
  const int64_t want = (lanes + bs - 1) / bs, cap = ctx->num_cu;
  const dim3 grid((unsigned)(want < cap ? want : cap)), block(bs);
  if (oct) FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((points_quad_kernel<LT, SCHEME, true>), grid, block, 0, st, ctx->host, ctx->te0, k.seed, k.cws, k.ocw, a));
  else FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((points_quad_kernel<LT, SCHEME>), grid, block, 0, st, ctx->host, ctx->te0, k.seed, k.cws, k.ocw, a));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = name;
  ctx->last_grid = (int)grid.x;
  ctx->last_block = bs;
  return FSS_AMD_OK;
}

}  // namespace
