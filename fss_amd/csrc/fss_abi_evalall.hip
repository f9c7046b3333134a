// libfss_amd.so, second unit: full-domain evaluation (single key, sub-blocks, batches of keys), the shared-prefix
// point walks that are built on its stages, and Grotto DCF.
#include "fss_host.hpp"
#include "fss_evalall.hpp"

// fss_abi_fused.hip, fss_abi_fused_batch.hip
int launch_fused_evalall(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st);
int launch_fused_evalall_batch(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* cws, const uint32_t* ocws,
                               const fssamd::FusedArgs& f, int grid, hipStream_t st);
int launch_fused_evalall_chacha(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                                const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st, bool batch);
// fss_abi_fused_nodes.hip
int launch_fused_nodes(fss_amd_ctx* ctx, int scheme, int last_t, const uint32_t* seed, const uint32_t* cws,
                       const fssamd::FusedArgs& f, int grid, hipStream_t st);

namespace {

// The first frontier comes from independent point walks (taken child only): cheaper than the tiny expansion
// launches they replace (each a launch, a table fill per workgroup and a drain: >= 10 us for almost no work).
// Every scheme and PRG walks it one PRG call per QUAD of lanes (frontier_quad_kernel, fss_quad.hpp): a level
// costs ~0.9 us instead of 2.2 (AES) or 1.6 (ChaCha), and the optimum is 16 levels plus the in-register tail
// (frontier_tail) -- DPF n = 20: 136 -> 44 us, 2^24 leaves +13 %; Half-Tree n = 20: 75 -> 39 us; DCF, two
// blocks and a group addition per level: 146 -> 120 us; ChaCha DPF n = 20: 88 -> 62 us.
// FSS_AMD_QUAD_FRONTIER=0 selects one walk per lane (then DPF / Grotto with the AES tables walk up to 2^18
// lanes x 18 levels, every wave of the chip busy, the rest 15).  FSS_AMD_TOPMAX overrides (A/B).  The workspace
// queries have no context and return the larger of the plans' needs.
inline bool quad_frontier() {
  static const bool v = [] { const char* e = tune_env("FSS_AMD_QUAD_FRONTIER"); return !(e && atoi(e) == 0); }();
  return v;
}
// single_key: the one-launch form of a single key's plan walks one level less -- its stages cost no launch, so the
// balance between the walk's latency and the stages' work moves (DPF n = 18 / 20 / 22 / 24: 24 / 41 / 104 / 324 ->
// 21 / 38 / 100 / 321 us; the 2^25-leaf block 0.624 -> 0.620 ms, Half-Tree 0.506 -> 0.503); batches and the
// shared-prefix plans measured flat to slightly worse with 15 and keep 16.
inline int top_max(int scheme, bool aes, bool single_key = false) {
  static int v = [] {
    const char* e = tune_env("FSS_AMD_TOPMAX");
    int t = e ? atoi(e) : -1;
    return (t < 0 || t > 20) ? -1 : t;
  }();
  if (v >= 0) return v;
  // DCF on the AES tables walks an octet of lanes per walk (fss_quad.hpp): one level less, so that the 2^14 octets of a single
  // tree still occupy half of every workgroup's waves, as the 2^15 quads of the other schemes do
  if (quad_frontier() && aes && frontier_octets<Lds, S_DCF>() && scheme == S_DCF) return single_key ? 14 : 15;
  if (quad_frontier()) return (single_key && aes) ? 15 : 16;
  if (aes && (scheme == S_DPF || scheme == S_GROTTO)) return 18;
  return 15;
}
// Levels of the frontier that its kernel expands in registers after the walk (frontier_quad_kernel's tail): the
// quad's node -> its four grandchildren, i.e. the first and thinnest expansion stage without a launch.
// `walked` levels come from the walk, `room` more levels exist above the leaf stage's reach: the tail is used
// when the walk was capped (it then replaces a launch; below the cap walking two more levels is cheaper).
inline int frontier_tail(int scheme, bool aes, bool capped, int room) {
  static const bool on = [] { const char* e = tune_env("FSS_AMD_FRONTIER_TAIL"); return !(e && atoi(e) == 0); }();
  (void)aes;
  if (!on || !quad_frontier() || scheme == S_DCF || !capped || room < 2) return 0;
  return 2;
}

struct Plan {
  int r, p, top, nstages;
  int tail;  // levels of `top` the frontier kernel expands in registers after its walk (frontier_tail)
  int D[64];
  int64_t cap_a, cap_b;  // nodes in the two frontier buffers
};

inline int dmax_for(int scheme) { return scheme == S_DCF ? 1 : 2; }  // levels expanded in registers
// Extra levels the last stage of a plan walks depth-first above those (expand_kernel's E): the deeper the
// last stage, the smaller and fewer the launches above it.  FSS_AMD_LEAF_E = 0 .. 2.
// Measured at 2^24 leaves: DPF +3 %, DCF +5 % with two extra levels; Half-Tree -7 % in round 1, +4-6 % once a lane's
// leaves were held back and stored as whole 128-byte lines (the loss had been half-written lines falling out of L2).
// A deep leaf stage needs enough input nodes to give every wave of the chip a chunk (64 nodes x 4096 waves =
// 2^18): below that the launch is a latency chain run by a fraction of the waves, and a shallower last stage
// on more nodes is faster (DPF n = 20: 136 -> 94 us with E = 0; round 2, profiles/r2).
// log2_leaves = log2 of the leaves the stage produces in total (all keys of a batch).
inline int leaf_extra(int scheme, int log2_leaves) {
  static int v = [] {
    const char* e = tune_env("FSS_AMD_LEAF_E");
    int t = e ? atoi(e) : -1;
    return (t < 0 || t > 3) ? -1 : t;
  }();
  if (v >= 0) return v;
  int e = 2;  // Half-Tree too, since its leaves go out as whole lines (round 2: 0.503 -> 0.481 ms on the 2^25-leaf block)
  const int room = log2_leaves - dmax_for(scheme) - 18;
  if (e > room) e = room < 0 ? 0 : room;
  // DPF at 2^23 and 2^24 leaves: one rolled level (0.175 / 0.320 ms against 0.177 / 0.322 with two; at 2^22 two are
  // ahead, 0.095 against 0.100; equal at 2^25, two ahead by 0.6 % at 2^26)
  if (scheme == S_DPF && (log2_leaves == 23 || log2_leaves == 24) && e > 1) e = 1;
  return e;
}

// Extra rolled levels of the stages between the frontier and the leaf stage (FSS_AMD_MID_E = 0 .. 2): fewer,
// deeper launches above the leaf stage.
inline int mid_extra(int scheme, int log2_leaves) {
  static int v = [] {
    const char* e = tune_env("FSS_AMD_MID_E");
    int t = e ? atoi(e) : -1;
    return (t < 0 || t > 2) ? -1 : t;
  }();
  if (v >= 0) return v;
  // DPF / Grotto: one rolled level per intermediate stage -- the 2^25-leaf block goes 2^18 -> 2^21 in one launch
  // instead of two (0.647 -> 0.639 ms); Half-Tree measured flat.  DCF (one level in registers): from 2^24 leaves on -- with
  // single-level stages a 2^24-leaf tree (14 levels walked, a three-level leaf stage) has eight stages, one more than the
  // one-launch form takes, and falls back to the chain of launches: 2^26 leaves 2.68 -> 2.48 ms with two levels per stage
  // (2^28: 9.74 -> 9.69; 2^20 - 2^22 are 1 - 4 % faster without; round 6, profiles/r6/dcf_evalall_stores.txt)
  if (scheme == S_DCF) return log2_leaves >= 24 ? 1 : 0;
  return (scheme == S_DPF || scheme == S_GROTTO) ? 1 : 0;
}

// `walks` plus the rounding of the workgroups' shares: grid * ceil(walks / grid) <= walks + grid - 1 with at most one
// workgroup per walk and at most 1024 of them (fused_grid; the workspace queries have no context).
inline int64_t fused_padded(int64_t walks) {
  const int64_t g = walks < 1024 ? walks : 1024;
  return walks + (g < 1 ? 0 : g - 1);
}
// Room (in nodes) the one-launch form needs in the two buffers beyond the plan's own sizes: every workgroup's private
// slice is ceil(walks / grid) walks wide; `slack_walks` covers the rounding of up to 1024 workgroups.
inline void fused_room(const Plan& pl, int64_t walks, int64_t* room_a, int64_t* room_b) {
  int sh = pl.tail;
  *room_a = walks << sh;
  *room_b = 0;
  for (int s = 0; s + 1 < pl.nstages; ++s) {
    sh += pl.D[s];
    const int64_t v = walks << sh;
    if (s % 2 == 0) { if (v > *room_b) *room_b = v; }
    else { if (v > *room_a) *room_a = v; }
  }
}

// nodes: the plan of a node table (shared-prefix point walks): every one of the r levels is an expansion -- a Half-Tree
// tree's last level too, which for leaves is the conversion (ConvertLastLevel) and not part of `top` / the stages.
// deep: rolled levels of the leaf stage beyond leaf_extra's (deep_leaf_levels: DPF leaves of a narrow output group in the
// one-launch form only -- the plan then differs from the group-agnostic one, whose workspace also covers it).
bool make_plan(int scheme, bool aes, int in_bits, int64_t first_leaf, int64_t nleaves, Plan* pl, bool nodes = false, int deep = 0) {
  if (in_bits < 1 || in_bits > 62 || nleaves < 1 || first_leaf < 0) return false;
  if (nleaves & (nleaves - 1)) return false;
  int r = 0;
  while (((int64_t)1 << r) < nleaves) ++r;
  if (r > in_bits) return false;
  if (first_leaf & (nleaves - 1)) return false;
  if (first_leaf + nleaves > ((int64_t)1 << in_bits)) return false;
  if (scheme == S_HT && r < 1 && !nodes) return false;  // the last Half-Tree level always yields two leaves
  pl->r = r;
  pl->p = in_bits - r;
  const int limit = r - ((scheme == S_HT && !nodes) ? 1 : 0);
  const int tmax = top_max(scheme, aes, true);
  int top = limit < tmax ? limit : tmax;
  pl->tail = frontier_tail(scheme, aes, top < limit, limit - top);
  top += pl->tail;
  pl->top = top;
  int rem = r - top;
  const int dm = dmax_for(scheme);
  pl->nstages = 0;
  // D[] holds the total depth of each stage (launch_stage splits it into rolled + in-register levels)
  if (rem == 0) {
    pl->D[pl->nstages++] = 0;  // leaf conversion only
  } else {
    int le = leaf_extra(scheme, r);
    if (deep > 0 && le == 2) {  // (only on top of the full two rolled levels; the same "a chunk for every wave" rule as leaf_extra)
      const int room = r - dm - 18;
      le += deep;
      if (le > room) le = room < 2 ? 2 : room;
    }
    const int leaf = rem < dm + le ? rem : dm + le;
    rem -= leaf;
    const int mid = dm + mid_extra(scheme, r);  // shallowest stage first: it runs on the fewest nodes
    if (rem % mid) pl->D[pl->nstages++] = rem % mid;
    for (int i = 0; i < rem / mid; ++i) pl->D[pl->nstages++] = mid;
    pl->D[pl->nstages++] = leaf;
  }
  int64_t f = (int64_t)1 << top;
  pl->cap_a = f;
  pl->cap_b = 1;
  for (int s = 0; s + 1 < pl->nstages; ++s) {  // outputs of every stage but the leaf stage
    f <<= pl->D[s];
    if (s % 2 == 0) { if (f > pl->cap_b) pl->cap_b = f; }
    else { if (f > pl->cap_a) pl->cap_a = f; }
  }
  // the one-launch form (evalall_fused_kernel): room for the rounding of the workgroups' slices, a few hundred KiB
  {
    int64_t fa, fb;
    fused_room(*pl, fused_padded((int64_t)1 << (top - pl->tail)), &fa, &fb);
    if (fa > pl->cap_a) pl->cap_a = fa;
    if (fb > pl->cap_b) pl->cap_b = fb;
  }
  return true;
}

// The plan as one launch (fss_abi_fused.hip): AES tables, DPF / Grotto / Half-Tree / DCF, a quad frontier with at
// a workgroup per CU (or per walk when there are fewer), stages of the depths the fused kernel is built for.  FSS_AMD_FUSED=0: the chain of
// launches (A/B).
inline bool fused_on() {
  static const bool v = [] { const char* e = tune_env("FSS_AMD_FUSED"); return !(e && atoi(e) == 0); }();
  return v;
}
constexpr int kDeepLeafDefault = 2;
// Two more rolled levels in the leaf stage of a DPF tree whose output group reads only the low 8 bytes of a leaf (OW = 2:
// the narrow leaf blocks leave twelve registers free, a rolled level takes four): the leaf stage then starts from
// 2^(r - 6) nodes instead of 2^(r - 4) and the frontier nodes written to and read back from the workspace shrink
// fourfold -- HBM counter traffic of the 2^28-leaf domain 1.163 x -> 1.04 x its leaves (profiles/r6/evalall_deep_leaf_ab.txt).
// FSS_AMD_LEAF_DEEP = 0 .. 2.
constexpr int kDeepLeafMax = 2;
inline int deep_leaf_levels(const fss_amd_ctx* ctx, int scheme, int group) {
  static const int v = [] { const char* e = tune_env("FSS_AMD_LEAF_DEEP"); const int t = e ? atoi(e) : kDeepLeafDefault; return t < 0 ? 0 : (t > kDeepLeafMax ? kDeepLeafMax : t); }();
  static const bool narrow_on = [] { const char* e = tune_env("FSS_AMD_LEAF_NARROW"); return !(e && atoi(e) == 0); }();
  if (scheme != S_DPF || ctx->prg == FSS_AMD_PRG_CHACHA || !narrow_on || !fused_on()) return 0;
  return (group >= G_U8 && group <= G_U64) ? v : 0;
}

// One workgroup per CU (the AES tables leave room for one), fewer only when there are fewer walks than CUs: a small
// domain is spread as wide as it goes (2^14 leaves on 256 workgroups 14.5 us, on 64 17, on 16 50: the stages below
// the walks are what more CUs shorten).  FSS_AMD_FUSED_GRID overrides, for tests of uneven shares.
inline int fused_grid(const fss_amd_ctx* ctx, int64_t walks) {
  static const int grid_env = [] { const char* e = tune_env("FSS_AMD_FUSED_GRID"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= 1024) ? v : 0; }();
  if (grid_env) return grid_env;
  return (int)(walks < ctx->num_cu ? (walks < 1 ? 1 : walks) : ctx->num_cu);
}
// The workgroups' private slices of the two buffers, for `nkeys` trees side by side.
inline void fused_slices(const Plan& pl, int64_t nkeys, int grid, uint4* const bufn[2], uint4* const bufv[2], FusedArgs* f) {
  f->p = pl.p; f->top = pl.top; f->tail = pl.tail;
  f->nmid = pl.nstages - 1;
  for (int s = 0; s < f->nmid; ++s) f->mid_t[s] = pl.D[s];
  const int64_t wmax = (((nkeys << (pl.top - pl.tail)) + grid - 1) / grid);
  fused_room(pl, wmax, &f->wg_cap[0], &f->wg_cap[1]);
  f->buf[0] = bufn[0]; f->buf[1] = bufn[1];
  f->bufv[0] = bufv[0]; f->bufv[1] = bufv[1];
  // fixed shares + progress-ranked priorities inside the stages: DPF block 0.619 -> 0.616 ms, n = 24 0.325 -> 0.321,
  // Half-Tree 0 .. +0.6 % (FSS_AMD_FUSED_RANK=0: the per-workgroup chunk counter)
  static const int rank = [] { const char* e = tune_env("FSS_AMD_FUSED_RANK"); return e ? (atoi(e) != 0) : 1; }();
  f->rank = rank;
}
bool fused_ok(const fss_amd_ctx* ctx, int scheme, const Plan& pl, int64_t nkeys, int deep = 0) {
  static const bool chacha_too = [] { const char* e = tune_env("FSS_AMD_FUSED_CHACHA"); return !(e && atoi(e) == 0); }();
  if (!fused_on() || !quad_frontier() || (ctx->prg == FSS_AMD_PRG_CHACHA && !chacha_too)) return false;
  if (scheme != S_DPF && scheme != S_GROTTO && scheme != S_HT && scheme != S_DCF) return false;
  if (pl.nstages < 1 || pl.nstages > 7) return false;
  (void)nkeys;
  const int leaf = pl.D[pl.nstages - 1], dm = dmax_for(scheme);
  if (leaf < 0 || leaf > dm + 2 + deep || (scheme == S_HT && leaf < 1)) return false;
  for (int s = 0; s + 1 < pl.nstages; ++s)
    if (pl.D[s] < 1 || pl.D[s] > 3) return false;
  return true;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t plan_workspace(int scheme, const Plan& pl, int64_t nleaves) {
  size_t per = (scheme == S_DCF) ? 32 : 16;
  size_t b = align256((size_t)pl.cap_a * per) + align256((size_t)pl.cap_b * per);
  if (scheme == S_GROTTO) b += align256((size_t)((nleaves + kScanChunk - 1) / kScanChunk));
  return b;
}

template <int SCHEME, int D, bool LEAF, int E = 0, bool BATCH = false>
int launch_expand(fss_amd_ctx* ctx, const SharedKey& k, const ExpandArgs& a_in, hipStream_t st) {
  constexpr int BS = 1024;
  const ExpandArgs& a0 = a_in;
  int64_t want = (a0.n_in + BS - 1) / BS;
  // workgroups per CU: the AES tables (128 KiB) admit one; ChaCha needs no LDS
  int64_t cap = (int64_t)ctx->num_cu * (ctx->prg == FSS_AMD_PRG_CHACHA ? 2 : 1);
  int grid = (int)(want < cap ? want : cap);
  ExpandArgs a = a_in;
  a.dyn = use_dyn(a.n_in, 64, grid, BS);
#if FSS_DIAG_STAMPS
  a.stamps = ctx->diag_stamps;
#endif
  // no launch-wide tail here: a stage's chunks last microseconds, and thousands of requests to one device-memory
  // word in that time serialise (Half-Tree 2^25 leaves: 0.53 -> 0.82 ms with it; the walks' chunks last 70 - 150 us)
  // narrow output groups: the leaf-level blocks skip the lookups of the seed words the group never reads
  constexpr bool kCanNarrow = LEAF && D >= 1 && (SCHEME == S_DPF || SCHEME == S_HT);
  static const bool narrow_on = [] { const char* e = tune_env("FSS_AMD_LEAF_NARROW"); return !(e && atoi(e) == 0); }();
  bool done = false;
  if constexpr (kCanNarrow) {
    if (narrow_on && ctx->prg != FSS_AMD_PRG_CHACHA && a.group >= G_U8 && a.group <= G_U64) {
      hipLaunchKernelGGL((expand_kernel<Lds, SCHEME, D, LEAF, BS, E, BATCH, 2>), dim3(grid), dim3(BS), 0, st, ctx->host,
                         ctx->te0, k.cws, k.ocw, a);
      done = true;
    }
  }
  if (!done)
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((expand_kernel<LT, SCHEME, D, LEAF, BS, E, BATCH>), dim3(grid), dim3(BS), 0, st,
                                        ctx->host, ctx->te0, k.cws, k.ocw, a));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = LEAF ? "expand_leaf" : "expand";
  ctx->last_grid = grid;
  ctx->last_block = BS;
  return FSS_AMD_OK;
}

// One stage of total depth T: the top T - dmax levels rolled (E), the rest in registers (D).
template <int SCHEME, bool BATCH = false>
int launch_stage(fss_amd_ctx* ctx, int T, bool leaf, const SharedKey& k, const ExpandArgs& a, hipStream_t st) {
  constexpr int DM = SCHEME == S_DCF ? 1 : 2;
  if (leaf) {
    if (T == 0) {
      if constexpr (SCHEME == S_HT) return FSS_AMD_EINVAL;
      else return launch_expand<SCHEME, 0, true, 0, BATCH>(ctx, k, a, st);
    }
    if (T == 1) return launch_expand<SCHEME, 1, true, 0, BATCH>(ctx, k, a, st);
    if constexpr (DM == 2) { if (T == 2) return launch_expand<SCHEME, 2, true, 0, BATCH>(ctx, k, a, st); }
    if (T == DM + 1) return launch_expand<SCHEME, DM, true, 1, BATCH>(ctx, k, a, st);
    if (T == DM + 2) return launch_expand<SCHEME, DM, true, 2, BATCH>(ctx, k, a, st);
  } else {
    if (T == 1) return launch_expand<SCHEME, 1, false, 0, BATCH>(ctx, k, a, st);
    if constexpr (DM == 2) { if (T == 2) return launch_expand<SCHEME, 2, false, 0, BATCH>(ctx, k, a, st); }
    if constexpr (!BATCH) {  // deep intermediate stages only occur in the shared-prefix plans
      if (T == DM + 1) return launch_expand<SCHEME, DM, false, 1>(ctx, k, a, st);
      if (T == DM + 2) return launch_expand<SCHEME, DM, false, 2>(ctx, k, a, st);
    }
  }
  return FSS_AMD_EINVAL;
}

// The first frontier: one walk per lane (ChaCha), or -- with the AES tables -- one walk per quad of lanes
// (frontier_quad_kernel; FSS_AMD_QUAD_FRONTIER=0 selects the per-lane kernel).
template <int SCHEME, bool BATCH>
int launch_frontier(fss_amd_ctx* ctx, hipStream_t st, int64_t nout, const uint32_t* seed, const uint32_t* cws,
                    const ExpandArgs& a, int top, int tail) {
  constexpr int WS = SCHEME == S_GROTTO ? S_DPF : SCHEME;
  if (quad_frontier()) {
    const int64_t lanes = (nout >> tail) << ((ctx->prg != FSS_AMD_PRG_CHACHA && frontier_octets<Lds, WS>()) ? 3 : 2);  // a quad per walk (DCF / AES: an octet)
    const int fbs = lanes >= 65536 ? 1024 : (lanes >= 256 ? 256 : 64);
    const int64_t want = (lanes + fbs - 1) / fbs, cap = (int64_t)ctx->num_cu * (1024 / fbs);
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((frontier_quad_kernel<LT, WS, BATCH>), dim3((unsigned)(want < cap ? want : cap)), dim3(fbs),
                                        0, st, ctx->host, ctx->te0, seed, cws, a, top, tail));
    HIP_TRY(hipGetLastError());
    return FSS_AMD_OK;
  }
  if (tail != 0) return FSS_AMD_EINVAL;  // only the quad kernel has a tail (frontier_tail)
  {
    const int fbs = nout >= 65536 ? 1024 : (nout >= 256 ? 256 : 64);  // few walks: spread them over CUs
    const int64_t want = (nout + fbs - 1) / fbs, cap = BATCH ? (int64_t)ctx->num_cu * (1024 / fbs) * 4 : want;
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((frontier_kernel<LT, WS, BATCH>), dim3((unsigned)(want < cap ? want : cap)), dim3(fbs),
                                        0, st, ctx->host, ctx->te0, seed, cws, a, top));
  }
  HIP_TRY(hipGetLastError());
  return FSS_AMD_OK;
}

template <int SCHEME>
int eval_all_impl(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                  const void* ocw, int64_t first_leaf, int64_t nleaves, void* ys, void* workspace, size_t ws_bytes,
                  hipStream_t st, bool grotto_scan = true) {
  if (!ctx || !seed || !cws || !ys || !group_ok(ctx, group) || (party != 0 && party != 1)) return FSS_AMD_EINVAL;
  if (SCHEME == S_HT && !ocw) return FSS_AMD_EINVAL;
  Plan pl;
  if (!make_plan(SCHEME, ctx->prg != FSS_AMD_PRG_CHACHA, in_bits, first_leaf, nleaves, &pl)) return FSS_AMD_EINVAL;
  size_t need = plan_workspace(SCHEME, pl, nleaves);
  if (!workspace || ws_bytes < need) return FSS_AMD_EINVAL;
  if (!aligned16(seed, cws, ocw, ys, workspace)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  // narrow DPF leaves in the one-launch form: a deeper leaf stage where the plan has room for one (deep_leaf_levels);
  // anything that keeps that plan from running as one launch falls back to the group-agnostic plan above
  int deep = deep_leaf_levels(ctx, SCHEME, group);
  if (deep > 0) {
    Plan dp;
    bool take = make_plan(SCHEME, true, in_bits, first_leaf, nleaves, &dp, false, deep) && plan_workspace(SCHEME, dp, nleaves) <= ws_bytes &&
                dp.D[dp.nstages - 1] > pl.D[pl.nstages - 1] && fused_ok(ctx, SCHEME, dp, 1, deep);
    if (take) {
      const int grid = fused_grid(ctx, (int64_t)1 << (dp.top - dp.tail));
      int64_t ra, rb;
      fused_room(dp, ((((int64_t)1 << (dp.top - dp.tail)) + grid - 1) / grid), &ra, &rb);
      take = ra * grid <= dp.cap_a && rb * grid <= dp.cap_b;
    }
    if (take) pl = dp;
    else deep = 0;
  }

  const size_t per = (SCHEME == S_DCF) ? 32 : 16;
  uint8_t* w = reinterpret_cast<uint8_t*>(workspace);
  uint4* bufn[2] = {reinterpret_cast<uint4*>(w), reinterpret_cast<uint4*>(w + align256((size_t)pl.cap_a * per))};
  uint4* bufv[2] = {bufn[0] + pl.cap_a, bufn[1] + pl.cap_b};  // DCF values behind the nodes

  ExpandArgs a{};
  a.scheme = SCHEME; a.in_bits = in_bits; a.party = party; a.group = group;
  SharedKey k;
  k.seed = reinterpret_cast<const uint32_t*>(seed);
  k.cws = reinterpret_cast<const uint32_t*>(cws);
  k.ocw = reinterpret_cast<const uint32_t*>(ocw);
  a.prefix = pl.r >= 63 ? 0ull : (uint64_t)(first_leaf >> pl.r);
  a.ys = ys;

  bool fused = false;
  if (fused_ok(ctx, SCHEME, pl, 1, deep)) {
    FusedArgs f{};
    f.a = a;
#if FSS_DIAG_STAMPS
    f.a.stamps = ctx->diag_stamps;  // kFusedStampSlots x 4096 waves fit the context's stamp buffer (the kernel checks)
#endif
    const int grid = fused_grid(ctx, (int64_t)1 << (pl.top - pl.tail));
    fused_slices(pl, 1, grid, bufn, bufv, &f);
    if (f.wg_cap[0] * grid <= pl.cap_a && f.wg_cap[1] * grid <= pl.cap_b) {
      int rc = ctx->prg == FSS_AMD_PRG_CHACHA
                   ? launch_fused_evalall_chacha(ctx, SCHEME, pl.D[pl.nstages - 1], k.seed, k.cws, k.ocw, f, grid, st, false)
                   : launch_fused_evalall(ctx, SCHEME, pl.D[pl.nstages - 1], k.seed, k.cws, k.ocw, f, grid, st);
      if (rc != FSS_AMD_OK) return rc;
      fused = true;
    }
  }
  // frontier at depth p + top
  a.level0 = pl.p + pl.top;
  a.out_nodes = bufn[0];
  a.out_v = bufv[0];
  if (!fused) {
    int rc = launch_frontier<SCHEME, false>(ctx, st, (int64_t)1 << pl.top, k.seed, k.cws, a, pl.top, pl.tail);
    if (rc != FSS_AMD_OK) return rc;
  }
  int64_t f = (int64_t)1 << pl.top;
  int level = pl.p + pl.top;
  int cur = 0;
  for (int s = 0; s < pl.nstages && !fused; ++s) {
    bool leaf = (s == pl.nstages - 1);
    a.level0 = level;
    a.n_in = f;
    a.in_nodes = bufn[cur]; a.in_v = bufv[cur];
    a.out_nodes = bufn[cur ^ 1]; a.out_v = bufv[cur ^ 1];
    int rc = launch_stage<SCHEME>(ctx, pl.D[s], leaf, k, a, st);
    if (rc != FSS_AMD_OK) return rc;
    f <<= pl.D[s];
    level += pl.D[s];
    cur ^= 1;
  }
  if (SCHEME == S_GROTTO && grotto_scan) {
    // running XOR over the leaf control bits, in place (grotto_dcf.cuh:158-162)
    uint8_t* t = reinterpret_cast<uint8_t*>(ys);
    uint8_t* par = w + align256((size_t)pl.cap_a * per) + align256((size_t)pl.cap_b * per);
    int64_t nch = (nleaves + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL(grotto_chunk_parity_kernel, dim3((unsigned)nch), dim3(256), 0, st, t, nleaves, par);
    hipLaunchKernelGGL(grotto_carry_kernel, dim3(1), dim3(64), 0, st, par, nch);
    hipLaunchKernelGGL(grotto_chunk_scan_kernel, dim3((unsigned)nch), dim3(256), 0, st, t, nleaves, par);
    HIP_TRY(hipGetLastError());
  }
  return FSS_AMD_OK;
}

// ---- many keys, full domain each (fss::gpu::{Dpf,HalfTreeDpf}EvalAllGpuBatch, eval_all_gpu.cuh:511-535) -------
// The same frontier + stage kernels over nkeys trees side by side (BATCH): node f of a level-l frontier belongs
// to key f >> l, ys[k * 2^in_bits + x].  The frontier depth shrinks with the number of keys so that the first
// launch has about 2^top_max walks in total (2^16 with the quad-lane frontier).
struct BatchPlan {
  Plan pl;
  int64_t nkeys;
};

bool make_batch_plan(int scheme, bool aes, int in_bits, int64_t nkeys, BatchPlan* bp) {
  if (nkeys < 1 || in_bits < 1 || in_bits > 40) return false;
  if (nkeys > ((int64_t)1 << 40) >> in_bits) return false;  // at most 2^40 leaves in one call
  Plan& pl = bp->pl;
  bp->nkeys = nkeys;
  pl.r = in_bits;
  pl.p = 0;
  int lk = 0;
  while (((int64_t)1 << lk) < nkeys) ++lk;
  const int limit = in_bits - (scheme == S_HT ? 1 : 0);
  int want = top_max(scheme, aes) - lk;  // about 2^top_max walks over all keys
  if (want < 0) want = 0;
  int top = limit < want ? limit : want;
  pl.tail = frontier_tail(scheme, aes, top < limit, limit - top);
  top += pl.tail;
  pl.top = top;
  int rem = in_bits - top;
  const int dm = dmax_for(scheme);
  pl.nstages = 0;
  if (rem == 0) {
    pl.D[pl.nstages++] = 0;
  } else {
    const int le = leaf_extra(scheme, in_bits + lk);
    const int leaf = rem < dm + le ? rem : dm + le;
    rem -= leaf;
    if (rem % dm) pl.D[pl.nstages++] = rem % dm;
    for (int i = 0; i < rem / dm; ++i) pl.D[pl.nstages++] = dm;
    pl.D[pl.nstages++] = leaf;
  }
  int64_t f = (int64_t)1 << top;
  pl.cap_a = f;
  pl.cap_b = 1;
  for (int s = 0; s + 1 < pl.nstages; ++s) {
    f <<= pl.D[s];
    if (s % 2 == 0) { if (f > pl.cap_b) pl.cap_b = f; }
    else { if (f > pl.cap_a) pl.cap_a = f; }
  }
  return true;
}

// Nodes in the two buffers: the plan's frontiers for all keys, or the one-launch form's slices (fused_room) if larger.
inline void batch_caps(const BatchPlan& bp, int64_t* ca, int64_t* cb) {
  int64_t fa, fb;
  fused_room(bp.pl, fused_padded(bp.nkeys << (bp.pl.top - bp.pl.tail)), &fa, &fb);
  *ca = bp.pl.cap_a * bp.nkeys;
  *cb = bp.pl.cap_b * bp.nkeys;
  if (fa > *ca) *ca = fa;
  if (fb > *cb) *cb = fb;
}
size_t batch_workspace(int scheme, const BatchPlan& bp) {
  size_t per = (scheme == S_DCF) ? 32 : 16;
  int64_t ca, cb;
  batch_caps(bp, &ca, &cb);
  return align256((size_t)ca * per) + align256((size_t)cb * per);
}

template <int SCHEME>
int eval_all_batch_impl(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cws,
                        const void* ocws, int64_t nkeys, void* ys, void* workspace, size_t ws_bytes, hipStream_t st) {
  if (!ctx || nkeys < 0 || !group_ok(ctx, group) || (party != 0 && party != 1)) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!seeds || !cws || !ys || (SCHEME == S_HT && !ocws)) return FSS_AMD_EINVAL;
  BatchPlan bp;
  if (!make_batch_plan(SCHEME, ctx->prg != FSS_AMD_PRG_CHACHA, in_bits, nkeys, &bp)) return FSS_AMD_EINVAL;
  const Plan& pl = bp.pl;
  if (!workspace || ws_bytes < batch_workspace(SCHEME, bp)) return FSS_AMD_EINVAL;
  if (!aligned16(seeds, cws, ocws, ys, workspace)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);

  const size_t per = (SCHEME == S_DCF) ? 32 : 16;
  int64_t ca, cb;
  batch_caps(bp, &ca, &cb);
  uint8_t* w = reinterpret_cast<uint8_t*>(workspace);
  uint4* bufn[2] = {reinterpret_cast<uint4*>(w), reinterpret_cast<uint4*>(w + align256((size_t)ca * per))};
  uint4* bufv[2] = {bufn[0] + ca, bufn[1] + cb};

  ExpandArgs a{};
  a.scheme = SCHEME; a.in_bits = in_bits; a.party = party; a.group = group;
  a.nkeys = nkeys;
  a.key_slots = SCHEME == S_HT ? in_bits : in_bits + 1;
  a.seeds = reinterpret_cast<const uint4*>(seeds);
  a.ys = ys;
  SharedKey k;
  k.cws = reinterpret_cast<const uint32_t*>(cws);
  k.ocw = reinterpret_cast<const uint32_t*>(ocws);

  // (ChaCha batches keep the chain: its stages run two workgroups per CU, the one-launch form one -- 64 keys x 2^20
  // 1710 -> 1770 us with it; single ChaCha keys gain 1.5 - 5 %)
  if ((SCHEME == S_DPF || SCHEME == S_HT) && ctx->prg != FSS_AMD_PRG_CHACHA && fused_ok(ctx, SCHEME, pl, nkeys)) {
    FusedArgs fa{};
    fa.a = a;
#if FSS_DIAG_STAMPS
    fa.a.stamps = ctx->diag_stamps;
#endif
    const int grid = fused_grid(ctx, nkeys << (pl.top - pl.tail));
    fused_slices(pl, nkeys, grid, bufn, bufv, &fa);
    if (fa.wg_cap[0] * grid <= ca && fa.wg_cap[1] * grid <= cb)
      return launch_fused_evalall_batch(ctx, SCHEME, pl.D[pl.nstages - 1], k.cws, k.ocw, fa, grid, st);
  }
  a.level0 = pl.top;
  a.out_nodes = bufn[0];
  a.out_v = bufv[0];
  {
    int rc = launch_frontier<SCHEME, true>(ctx, st, nkeys << pl.top, nullptr, k.cws, a, pl.top, pl.tail);
    if (rc != FSS_AMD_OK) return rc;
  }
  int64_t f = nkeys << pl.top;
  int level = pl.top, cur = 0;
  for (int s = 0; s < pl.nstages; ++s) {
    const bool leaf = (s == pl.nstages - 1);
    a.level0 = level;
    a.n_in = f;
    a.in_nodes = bufn[cur]; a.in_v = bufv[cur];
    a.out_nodes = bufn[cur ^ 1]; a.out_v = bufv[cur ^ 1];
    int rc = launch_stage<SCHEME, true>(ctx, pl.D[s], leaf, k, a, st);
    if (rc != FSS_AMD_OK) return rc;
    f <<= pl.D[s];
    level += pl.D[s];
    cur ^= 1;
  }
  return FSS_AMD_OK;
}

// ---- shared-prefix point evaluation ------------------------------------------------------
constexpr int kPrefixMax = 26;

struct PrefixPlan {
  int top, nstages, tail;
  int D[32];
  int64_t cap_a, cap_b;  // intermediate frontiers (the last stage writes the node table itself)
};

bool make_prefix_plan(int scheme, bool aes, int in_bits, int k, PrefixPlan* pl) {
  if (k < 1 || k > kPrefixMax || k > in_bits) return false;
  if (scheme == S_HT && k > in_bits - 1) return false;
  if (scheme != S_DPF && scheme != S_DCF && scheme != S_HT) return false;
  int top = k < top_max(scheme, aes) ? k : top_max(scheme, aes);
  pl->tail = frontier_tail(scheme, aes, top < k, k - top);
  top += pl->tail;
  pl->top = top;
  int rem = k - top;
  const int dm = dmax_for(scheme);
  pl->nstages = 0;
  if (rem > 0) {
    const int le = leaf_extra(scheme, k);
    const int last = rem < dm + le ? rem : dm + le;
    rem -= last;
    if (rem % dm) pl->D[pl->nstages++] = rem % dm;
    for (int i = 0; i < rem / dm; ++i) pl->D[pl->nstages++] = dm;
    pl->D[pl->nstages++] = last;
  }
  int64_t f = (int64_t)1 << top;
  pl->cap_a = pl->nstages ? f : 0;  // with no stage the frontier kernel writes the table directly
  pl->cap_b = 0;
  for (int s = 0; s + 1 < pl->nstages; ++s) {
    f <<= pl->D[s];
    if (s % 2 == 0) { if (f > pl->cap_b) pl->cap_b = f; }
    else { if (f > pl->cap_a) pl->cap_a = f; }
  }
  return true;
}

// The node table as ONE launch (evalall_fused_kernel with OW = 0, fss_abi_fused_nodes.hip): the tree of the key's top
// k levels under the one-launch EvalAll's plan, from kPrefixFusedMin levels on (below that a plan may end on a stage
// that expands nothing; those tables are a few KiB and the chain of launches builds them).  Four launches of
// 2^23 nodes: 0.219 ms as a chain (three of them thin: a table fill and a drain each), ~0.17 as one.
constexpr int kPrefixFusedMin = 12;
bool make_prefix_fused_plan(int scheme, int k, Plan* fp) {
  if (k < kPrefixFusedMin || k > kPrefixMax) return false;
  if (!make_plan(scheme, true, k, 0, (int64_t)1 << k, fp, /*nodes=*/true)) return false;
  return fp->nstages >= 1 && fp->D[fp->nstages - 1] >= 1;
}
size_t prefix_workspace(int scheme, const PrefixPlan& pl, int k) {
  size_t per = (scheme == S_DCF) ? 32 : 16;
  size_t chain = align256((size_t)pl.cap_a * per) + align256((size_t)pl.cap_b * per);
  Plan fp;
  if (make_prefix_fused_plan(scheme, k, &fp)) {
    const size_t fused = align256((size_t)fp.cap_a * per) + align256((size_t)fp.cap_b * per);
    if (fused > chain) chain = fused;
  }
  return align256(((size_t)1 << k) * per) + chain;
}

template <int SCHEME>
int points_shared_impl(fss_amd_ctx* ctx, const char* name, int in_bits, int group, int party, const void* seed,
                       const void* cws, const void* ocw, const void* xs, int x_bytes, int64_t n, void* ys, int k,
                       void* workspace, size_t ws_bytes, hipStream_t st) {
  if (!ctx || !seed || !cws || n < 0 || !group_ok(ctx, group) || (party != 0 && party != 1) || !x_ok(in_bits, x_bytes))
    return FSS_AMD_EINVAL;
  if (in_bits > 64) return FSS_AMD_EINVAL;  // the node table is indexed by the top bits of a 64-bit x
  if (n > 0 && (!xs || !ys)) return FSS_AMD_EINVAL;
  if (SCHEME == S_HT && !ocw) return FSS_AMD_EINVAL;
  PrefixPlan pl;
  if (!make_prefix_plan(SCHEME, ctx->prg != FSS_AMD_PRG_CHACHA, in_bits, k, &pl)) return FSS_AMD_EINVAL;
  if (!workspace || ws_bytes < prefix_workspace(SCHEME, pl, k)) return FSS_AMD_EINVAL;
  if (!aligned16(seed, cws, ocw, ys, workspace) || !aligned_to(xs, x_bytes)) return FSS_AMD_EINVAL;
  if (n == 0) return FSS_AMD_OK;
  if (n <= quad_points_max()) {  // too few points to amortise a node table: the plain quad-lane walk (same outputs)
    FSS_ON_DEVICE(ctx);
    EvalArgs q{};
    q.in_bits = in_bits; q.party = party; q.group = group; q.x_bytes = x_bytes; q.n = n;
    q.xs = xs; q.ys = ys;
    SharedKey sk;
    sk.seed = reinterpret_cast<const uint32_t*>(seed);
    sk.cws = reinterpret_cast<const uint32_t*>(cws);
    sk.ocw = reinterpret_cast<const uint32_t*>(ocw);
    return launch_points_quad<SCHEME>(ctx, name, sk, q, st);
  }
  FSS_ON_DEVICE(ctx);

  const size_t per = (SCHEME == S_DCF) ? 32 : 16;
  const int64_t ntab = (int64_t)1 << k;
  uint8_t* w = reinterpret_cast<uint8_t*>(workspace);
  uint4* tab_n = reinterpret_cast<uint4*>(w);
  uint4* tab_v = tab_n + ntab;  // DCF values behind the nodes
  uint8_t* w1 = w + align256((size_t)ntab * per);
  uint4* bufn[2] = {reinterpret_cast<uint4*>(w1), reinterpret_cast<uint4*>(w1 + align256((size_t)pl.cap_a * per))};
  uint4* bufv[2] = {bufn[0] + pl.cap_a, bufn[1] + pl.cap_b};

  SharedKey key;
  key.seed = reinterpret_cast<const uint32_t*>(seed);
  key.cws = reinterpret_cast<const uint32_t*>(cws);
  key.ocw = reinterpret_cast<const uint32_t*>(ocw);

  // 1. the node table of level k: one launch (the one-launch EvalAll's plan, nodes instead of leaves) ...
  ExpandArgs e{};
  e.scheme = SCHEME; e.in_bits = in_bits; e.party = party; e.group = group;
  e.prefix = 0; e.level0 = pl.top;
  bool fused = false;
  static const bool fused_nodes_on = [] { const char* v = tune_env("FSS_AMD_PREFIX_FUSED"); return !(v && atoi(v) == 0); }();
  Plan fp;
  if (fused_nodes_on && ctx->prg != FSS_AMD_PRG_CHACHA && make_prefix_fused_plan(SCHEME, k, &fp) && fused_ok(ctx, SCHEME, fp, 1)) {
    uint4* fbn[2] = {reinterpret_cast<uint4*>(w1), reinterpret_cast<uint4*>(w1 + align256((size_t)fp.cap_a * per))};
    uint4* fbv[2] = {fbn[0] + fp.cap_a, fbn[1] + fp.cap_b};
    FusedArgs fa{};
    fa.a = e;
    fa.a.out_nodes = tab_n;
    fa.a.out_v = tab_v;
    const int grid = fused_grid(ctx, (int64_t)1 << (fp.top - fp.tail));
    fused_slices(fp, 1, grid, fbn, fbv, &fa);
    if (fa.wg_cap[0] * grid <= fp.cap_a && fa.wg_cap[1] * grid <= fp.cap_b) {
      int rc = launch_fused_nodes(ctx, SCHEME, fp.D[fp.nstages - 1], key.seed, key.cws, fa, grid, st);
      if (rc != FSS_AMD_OK) return rc;
      fused = true;
    }
  }
  // ... or (ChaCha, small tables, FSS_AMD_PREFIX_FUSED=0) the frontier walk to depth `top`, then expansion stages
  e.out_nodes = pl.nstages ? bufn[0] : tab_n;
  e.out_v = pl.nstages ? bufv[0] : tab_v;
  if (!fused) {
    int rc = launch_frontier<SCHEME, false>(ctx, st, (int64_t)1 << pl.top, key.seed, key.cws, e, pl.top, pl.tail);
    if (rc != FSS_AMD_OK) return rc;
  }
  int64_t f = (int64_t)1 << pl.top;
  int level = pl.top, cur = 0;
  for (int s = 0; s < pl.nstages && !fused; ++s) {
    const bool last = (s == pl.nstages - 1);
    e.level0 = level;
    e.n_in = f;
    e.in_nodes = bufn[cur]; e.in_v = bufv[cur];
    e.out_nodes = last ? tab_n : bufn[cur ^ 1];
    e.out_v = last ? tab_v : bufv[cur ^ 1];
    int rc = launch_stage<SCHEME>(ctx, pl.D[s], false, key, e, st);
    if (rc != FSS_AMD_OK) return rc;
    f <<= pl.D[s];
    level += pl.D[s];
    cur ^= 1;
  }
  // 2. every point resumes from its level-k ancestor
  EvalArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = group; a.x_bytes = x_bytes; a.n = n;
  a.xs = xs; a.ys = ys;
  a.start_level = k; a.pre_nodes = tab_n; a.pre_v = tab_v;
  return launch_points<SCHEME, true, true>(ctx, name, key, a, st);
}

}  // namespace

extern "C" {

// The plan FSS_AMD_POINTS_AUTO takes for `npoints` points of one key: prefix_bits, or 0 for the plain walk.
// Blocks executed: npoints x (in_bits - k) walked + 2^(k+1) expanded, least at k ~ log2(npoints) - 0.5; with the
// expansion a little less efficient than the walk the measured optimum is k = floor(log2 npoints) - 1
// (profiles/r5/shared_prefix.txt), capped by the key (in_bits; Half-Tree in_bits - 1: its last level is the output
// conversion) and by the table the plan may build (2^26 nodes).  Below kAutoPrefixMinPoints (2^18: 1.3 x there, 1.8 x at
// 2^20, 2.9 x at 2^24) the plain walk is taken.
constexpr int64_t kAutoPrefixMinPoints = (int64_t)1 << 18;
int fss_amd_points_prefix_bits(int scheme, int in_bits, int64_t npoints) {
  if ((scheme != S_DPF && scheme != S_DCF && scheme != S_HT) || in_bits < 1 || in_bits > 64) return 0;
  static const int64_t min_points = [] { const char* e = tune_env("FSS_AMD_AUTO_PREFIX_MIN"); return e ? (int64_t)atoll(e) : kAutoPrefixMinPoints; }();
  static const int back = [] { const char* e = tune_env("FSS_AMD_AUTO_PREFIX_BACK"); return e ? atoi(e) : 1; }();
  if (npoints < min_points || npoints <= quad_points_max()) return 0;
  int lg = 0;
  while (((int64_t)2 << lg) <= npoints) ++lg;  // floor(log2 npoints)
  int k = lg - (lg >= 19 ? back : back + 1);  // (2^18 points: every table up to 2^18 nodes costs the same ~60 us, a launch-latency chain)
  const int cap = scheme == S_HT ? in_bits - 1 : in_bits;
  if (k > cap) k = cap;
  if (k > kPrefixMax) k = kPrefixMax;
  return k < 1 ? 0 : k;
}

size_t fss_amd_eval_points_workspace(int scheme, int in_bits, int prefix_bits) {
  PrefixPlan pl;
  size_t need = 0;
  for (int aes = 0; aes < 2; ++aes) {
    if (!make_prefix_plan(scheme, aes != 0, in_bits, prefix_bits, &pl)) return 0;
    const size_t b = prefix_workspace(scheme, pl, prefix_bits);
    if (b > need) need = b;
  }
  return need;
}

int fss_amd_dpf_eval_points_shared(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed,
                                   const void* cws, const void* xs, int x_bytes, int64_t npoints, void* ys,
                                   int prefix_bits, void* workspace, size_t workspace_bytes, void* stream) {
  return points_shared_impl<S_DPF>(ctx, "dpf_eval_points_shared", in_bits, group, party, seed, cws, nullptr, xs, x_bytes,
                                   npoints, ys, prefix_bits, workspace, workspace_bytes, S(stream));
}
int fss_amd_dcf_eval_points_shared(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed,
                                   const void* cws, const void* xs, int x_bytes, int64_t npoints, void* ys,
                                   int prefix_bits, void* workspace, size_t workspace_bytes, void* stream) {
  return points_shared_impl<S_DCF>(ctx, "dcf_eval_points_shared", in_bits, group, party, seed, cws, nullptr, xs, x_bytes,
                                   npoints, ys, prefix_bits, workspace, workspace_bytes, S(stream));
}
int fss_amd_ht_eval_points_shared(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed,
                                  const void* cws, const void* ocw, const void* xs, int x_bytes, int64_t npoints,
                                  void* ys, int prefix_bits, void* workspace, size_t workspace_bytes, void* stream) {
  return points_shared_impl<S_HT>(ctx, "ht_eval_points_shared", in_bits, group, party, seed, cws, ocw, xs, x_bytes,
                                  npoints, ys, prefix_bits, workspace, workspace_bytes, S(stream));
}

size_t fss_amd_eval_all_workspace(int scheme, int in_bits, int64_t nleaves) {
  Plan pl;
  if (scheme < 0 || scheme > 3) return 0;
  size_t need = 0;
  for (int aes = 0; aes < 2; ++aes) {
    if (!make_plan(scheme, aes != 0, in_bits, 0, nleaves, &pl)) return 0;
    const size_t b = plan_workspace(scheme, pl, nleaves);
    if (b > need) need = b;
  }
  if (scheme == S_DPF)  // the deeper leaf stage narrow output groups may take (deep_leaf_levels)
    for (int deep = 1; deep <= kDeepLeafMax; ++deep)
      if (make_plan(scheme, true, in_bits, 0, nleaves, &pl, false, deep)) {
        const size_t b = plan_workspace(scheme, pl, nleaves);
        if (b > need) need = b;
      }
  return need;
}

int fss_amd_dpf_eval_all(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                         int64_t first_leaf, int64_t nleaves, void* ys, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return eval_all_impl<S_DPF>(ctx, in_bits, group, party, seed, cws, nullptr, first_leaf, nleaves, ys, workspace,
                              workspace_bytes, S(stream));
}
int fss_amd_dcf_eval_all(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                         int64_t first_leaf, int64_t nleaves, void* ys, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return eval_all_impl<S_DCF>(ctx, in_bits, group, party, seed, cws, nullptr, first_leaf, nleaves, ys, workspace,
                              workspace_bytes, S(stream));
}
int fss_amd_ht_eval_all(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                        const void* ocw, int64_t first_leaf, int64_t nleaves, void* ys, void* workspace,
                        size_t workspace_bytes, void* stream) {
  return eval_all_impl<S_HT>(ctx, in_bits, group, party, seed, cws, ocw, first_leaf, nleaves, ys, workspace,
                             workspace_bytes, S(stream));
}
int fss_amd_grotto_eval_all(fss_amd_ctx* ctx, int in_bits, int party, const void* seed, const void* cws,
                            int64_t first_leaf, int64_t nleaves, void* ys_bytes, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return eval_all_impl<S_GROTTO>(ctx, in_bits, G_BYTES, party, seed, cws, nullptr, first_leaf, nleaves, ys_bytes,
                                 workspace, workspace_bytes, S(stream));
}

int fss_amd_grotto_preprocess(fss_amd_ctx* ctx, int in_bits, int party, const void* seed, const void* cws, void* pt,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx || !pt || !workspace || in_bits < 1 || in_bits > 32) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  const int64_t N = (int64_t)1 << in_bits;
  size_t ea = fss_amd_eval_all_workspace(S_GROTTO, in_bits, N);
  if (ea == 0 || workspace_bytes < ea + (size_t)N) return FSS_AMD_EINVAL;
  uint8_t* w = reinterpret_cast<uint8_t*>(workspace);
  uint8_t* leaves = w + ea;  // 256-byte aligned staging for the leaf control bits
  int rc = eval_all_impl<S_GROTTO>(ctx, in_bits, G_BYTES, party, seed, cws, nullptr, 0, N, leaves, w, ea, S(stream),
                                   /*grotto_scan=*/false);
  if (rc != FSS_AMD_OK) return rc;
  uint8_t* p = reinterpret_cast<uint8_t*>(pt);
  HIP_TRY(hipMemcpyAsync(p + (N - 1), leaves, (size_t)N, hipMemcpyDeviceToDevice, S(stream)));
  for (int l = in_bits - 1; l >= 0; --l) {
    int64_t count = (int64_t)1 << l, first = count - 1;
    int64_t want = (count + 255) / 256;
    int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(grotto_tree_level_kernel, dim3(grid), dim3(256), 0, S(stream), p, first, count);
  }
  HIP_TRY(hipGetLastError());
  return FSS_AMD_OK;
}

int fss_amd_grotto_eval_table(fss_amd_ctx* ctx, int in_bits, const void* pt, const void* xs, int x_bytes,
                              int64_t npoints, void* ys_bytes, void* stream) {
  if (!ctx || !pt || npoints < 0 || in_bits < 1 || in_bits > 32 || !x_ok(in_bits, x_bytes)) return FSS_AMD_EINVAL;
  if (npoints == 0) return FSS_AMD_OK;
  if (!xs || !ys_bytes || !aligned_to(xs, x_bytes)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  int64_t want = (npoints + 255) / 256;
  int grid = (int)(want < (int64_t)ctx->num_cu * 16 ? want : (int64_t)ctx->num_cu * 16);
  hipLaunchKernelGGL(grotto_table_eval_kernel, dim3(grid), dim3(256), 0, S(stream),
                     reinterpret_cast<const uint8_t*>(pt), in_bits, xs, x_bytes, npoints,
                     reinterpret_cast<uint8_t*>(ys_bytes));
  HIP_TRY(hipGetLastError());
  return FSS_AMD_OK;
}

int fss_amd_grotto_eval_keys(fss_amd_ctx* ctx, int in_bits, int party, const void* seeds, const void* cw_s,
                             const void* tr_bits, const void* xs, int x_bytes, int64_t nkeys, void* ys_bytes,
                             void* stream) {
  if (!ctx || nkeys < 0 || (party != 0 && party != 1) || !x_ok(in_bits, x_bytes) || in_bits > 64) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!seeds || !cw_s || !tr_bits || !xs || !ys_bytes) return FSS_AMD_EINVAL;
  if (!aligned16(seeds, cw_s) || !aligned_to(tr_bits, 8) || !aligned_to(xs, x_bytes)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  EvalArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = G_BYTES; a.x_bytes = x_bytes; a.n = nkeys;
  a.xs = xs; a.ys = ys_bytes;
  a.seeds = reinterpret_cast<const uint4*>(seeds);
  a.cw_s = reinterpret_cast<const uint4*>(cw_s);
  a.tr_bits = reinterpret_cast<const uint64_t*>(tr_bits);
  static const int bs = [] {
    const char* e = tune_env("FSS_AMD_GROTTO_BS");
    return (e && atoi(e) == 512) ? 512 : 1024;
  }();
  return FSS_BY_PRG(ctx, ([&] {
    auto go = [&](auto kernel) {
      int grid = 0;
      int rc = persistent_grid(ctx, kernel, bs, nkeys, &grid);
      if (rc != FSS_AMD_OK) return rc;
      a.dyn = use_dyn(nkeys, 64, grid, bs);
      if (in_bits >= 16) set_tail(ctx, S(stream), a.dyn, nkeys, 64, grid, &a.gq, &a.static_chunks);  // long chunks only
      return launch_grid(ctx, kernel, "grotto_eval_keys", grid, bs, S(stream), ctx->host, ctx->te0, a);
    };
    // AES: taken child per level, siblings deferred to dense batches (FSS_AMD_GROTTO_DEFER=0: both children per level)
    if constexpr (std::is_same_v<LT, Lds>) {
      static const bool defer = [] { const char* e = tune_env("FSS_AMD_GROTTO_DEFER"); return !(e && atoi(e) == 0); }();
      if (defer && bs == 1024) {
        int grid = 0;
        int rc = persistent_grid(ctx, grotto_walk_defer_kernel<1024>, bs, (nkeys + 1) / 2, &grid);
        if (rc != FSS_AMD_OK) return rc;
        // like the point walks (launch_points_k): fixed shares with progress-ranked priorities below 65 536 AES blocks
        // per resident wave (a key costs 1.5 in_bits blocks on average), the dynamic queue above
        const int64_t blocks = nkeys * (int64_t)(in_bits + in_bits / 2);
        a.dyn = tune().dyn >= 0 ? tune().dyn : (blocks >= tune().dyn_min * grid * (bs / 64) ? 1 : 0);
        a.rot_prio = tune().rot_prio >= 0 ? tune().rot_prio : (a.dyn ? 0 : tune().static_prio);
        a.prio_every = tune().prio_every;
        if (in_bits >= 16) set_tail(ctx, S(stream), a.dyn, nkeys, 128, grid, &a.gq, &a.static_chunks);
#if FSS_DIAG_STAMPS
        a.stamps = ctx->diag_stamps;
#endif
        return launch_grid(ctx, grotto_walk_defer_kernel<1024>, "grotto_eval_keys", grid, bs, S(stream), ctx->host, ctx->te0, a);
      }
    }
    return bs == 512 ? go(grotto_walk_kernel<LT, 512>) : go(grotto_walk_kernel<LT, 1024>);
  }()));
}

// ------------------------------------------------------------------ batched full-domain evaluation
size_t fss_amd_eval_all_batch_workspace(int scheme, int in_bits, int64_t nkeys) {
  BatchPlan bp;
  if (scheme != S_DPF && scheme != S_DCF && scheme != S_HT) return 0;
  size_t need = 0;
  for (int aes = 0; aes < 2; ++aes) {
    if (!make_batch_plan(scheme, aes != 0, in_bits, nkeys, &bp)) return 0;
    const size_t b = batch_workspace(scheme, bp);
    if (b > need) need = b;
  }
  return need;
}
int fss_amd_dpf_eval_all_batch(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cws,
                               int64_t nkeys, void* ys, void* workspace, size_t workspace_bytes, void* stream) {
  return eval_all_batch_impl<S_DPF>(ctx, in_bits, group, party, seeds, cws, nullptr, nkeys, ys, workspace,
                                    workspace_bytes, S(stream));
}
int fss_amd_dcf_eval_all_batch(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cws,
                               int64_t nkeys, void* ys, void* workspace, size_t workspace_bytes, void* stream) {
  return eval_all_batch_impl<S_DCF>(ctx, in_bits, group, party, seeds, cws, nullptr, nkeys, ys, workspace,
                                    workspace_bytes, S(stream));
}
int fss_amd_ht_eval_all_batch(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cws,
                              const void* ocws, int64_t nkeys, void* ys, void* workspace, size_t workspace_bytes,
                              void* stream) {
  return eval_all_batch_impl<S_HT>(ctx, in_bits, group, party, seeds, cws, ocws, nkeys, ys, workspace, workspace_bytes,
                                   S(stream));
}

}  // extern "C"
