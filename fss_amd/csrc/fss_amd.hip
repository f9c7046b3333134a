// libfss_amd.so -- C ABI (include/fss_amd.h) over the gfx950 kernels.  This unit: context, raw PRG, point
// evaluation (shared key and independent keys), relayout.  fss_abi_evalall.hip: full-domain evaluation,
// shared-prefix walks, Grotto.  fss_abi_gen.hip: key generation, verifiable DPF.
#include "fss_host.hpp"
#include "fss_gen.hpp"

namespace {
void make_sbox(uint8_t sbox[256]) {
  uint8_t p = 1, q = 1;
  do {
    p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1b : 0));
    q ^= (uint8_t)(q << 1);
    q ^= (uint8_t)(q << 2);
    q ^= (uint8_t)(q << 4);
    if (q & 0x80) q ^= 0x09;
    auto rl = [](uint8_t v, int n) { return (uint8_t)((v << n) | (v >> (8 - n))); };
    sbox[p] = (uint8_t)(q ^ rl(q, 1) ^ rl(q, 2) ^ rl(q, 3) ^ rl(q, 4) ^ 0x63);
  } while (p != 1);
  sbox[0] = 0x63;
}

inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

void expand_key(const uint8_t sbox[256], const uint8_t key[16], uint32_t rk[44]) {
  uint8_t w[176];
  memcpy(w, key, 16);
  uint8_t rcon = 1;
  for (int i = 16; i < 176; i += 4) {
    uint8_t t[4] = {w[i - 4], w[i - 3], w[i - 2], w[i - 1]};
    if (i % 16 == 0) {
      uint8_t t0 = t[0];
      t[0] = (uint8_t)(sbox[t[1]] ^ rcon);
      t[1] = sbox[t[2]];
      t[2] = sbox[t[3]];
      t[3] = sbox[t0];
      rcon = xtime(rcon);
    }
    for (int j = 0; j < 4; ++j) w[i + j] = (uint8_t)(w[i - 16 + j] ^ t[j]);
  }
  for (int i = 0; i < 44; ++i)
    rk[i] = (uint32_t)w[4 * i] | ((uint32_t)w[4 * i + 1] << 8) | ((uint32_t)w[4 * i + 2] << 16) |
            ((uint32_t)w[4 * i + 3] << 24);
}

}  // namespace

extern "C" {

int fss_amd_ctx_create(fss_amd_ctx** out, int device, const uint8_t aes_keys[64], const int32_t hash_key[4]) {
  if (!out || !aes_keys) return FSS_AMD_EINVAL;
  *out = nullptr;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return FSS_AMD_EINVAL;
  DeviceGuard guard_(device);
  HIP_TRY(guard_.err);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));

  fss_amd_ctx* c = new (std::nothrow) fss_amd_ctx();
  if (!c) return FSS_AMD_EINVAL;
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  c->last_kernel = "";
  c->last_grid = c->last_block = 0;
  c->prg = FSS_AMD_PRG_AES128_MMO;
  c->host.chacha_nonce[0] = c->host.chacha_nonce[1] = 0;
  c->host.chacha_rounds = 20;
  c->host.uint_mod = 0;
  c->host.hash_kind = 0;
  {  // default VDPF hash key: the reference bench's kBlake3Iv (src/bench_gpu.cu:27-30)
    const uint32_t iv[8] = {0x11223344u, 0x55667788u, 0x99aabbccu, 0xddeeff00u, 0x00112233u, 0x44556677u, 0x8899aabbu, 0xccddeeffu};
    for (int i = 0; i < 8; ++i) c->host.blake3_iv[i] = iv[i];
  }

  uint8_t sbox[256];
  uint32_t te0[256];
  make_sbox(sbox);
  for (int i = 0; i < 256; ++i) {
    uint8_t s = sbox[i], s2 = xtime(s), s3 = (uint8_t)(s2 ^ s);
    te0[i] = (uint32_t)s2 | ((uint32_t)s << 8) | ((uint32_t)s << 16) | ((uint32_t)s3 << 24);
  }
  for (int k = 0; k < 4; ++k) expand_key(sbox, aes_keys + 16 * k, c->host.rk[k]);
  for (int r = 0; r < 11; ++r)
    for (int j = 0; j < 4; ++j) {
      const uint32_t *k0 = c->host.rk[0], *k1 = c->host.rk[1], *k2 = c->host.rk[2], *k3 = c->host.rk[3];
      c->host.sel01[r][j] = k0[4 * r + j]; c->host.sel01[r][4 + j] = k0[4 * r + j] ^ k1[4 * r + j];
      c->host.sel02[r][j] = k0[4 * r + j]; c->host.sel02[r][4 + j] = k0[4 * r + j] ^ k2[4 * r + j];
      c->host.sel13[r][j] = k1[4 * r + j]; c->host.sel13[r][4 + j] = k1[4 * r + j] ^ k3[4 * r + j];
    }
  for (int i = 0; i < 4; ++i) c->host.hash_key[i] = hash_key ? (uint32_t)hash_key[i] : 0u;

  c->te0 = nullptr;
  hipError_t e = hipMalloc(&c->te0, sizeof(te0) + sizeof(DevCtx));
  if (e == hipSuccess) e = hipMemcpy(c->te0, te0, sizeof(te0), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    c->host.self_dev = c->te0 + 256;
    e = hipMemcpy(c->te0 + 256, &c->host, sizeof(DevCtx), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc(&c->tail_pairs, fss_amd_ctx::kTailPairs * 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(c->tail_pairs, 0, fss_amd_ctx::kTailPairs * 2 * sizeof(uint32_t));
#if FSS_DIAG_STAMPS
  // one record per wave of the largest grid any stamped kernel launches: two workgroups of 16 waves per CU (ChaCha)
  if (e == hipSuccess) e = hipMalloc(&c->diag_stamps, kDiagStampRecords * 5 * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMemset(c->diag_stamps, 0, kDiagStampRecords * 5 * sizeof(uint64_t));
#endif
  if (e != hipSuccess) {
    g_last_hip = (int)e;
    if (c->te0) (void)hipFree(c->te0);
    if (c->tail_pairs) (void)hipFree(c->tail_pairs);
    if (c->diag_stamps) (void)hipFree(c->diag_stamps);
    delete c;
    return FSS_AMD_EHIP;
  }
  *out = c;
  return FSS_AMD_OK;
}

#if FSS_DIAG_STAMPS
// diagnostic library only (not declared in fss_amd.h): the stamps of the last point-walk launch, 4 x uint64 per
// workgroup {memtime0, realtime0, memtime1, realtime1}; synchronises the device
int fss_amd_diag_read_stamps(fss_amd_ctx* ctx, uint64_t* out_host, int workgroups) {
  if (!ctx || !out_host || workgroups < 1 || workgroups > kDiagStampRecords) return FSS_AMD_EINVAL;  // `workgroups` = records (waves)
  FSS_ON_DEVICE(ctx);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_host, ctx->diag_stamps, (size_t)workgroups * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return FSS_AMD_OK;
}
// s_memrealtime of every wave's first instruction (before the table fill), point walks only
int fss_amd_diag_read_entry_stamps(fss_amd_ctx* ctx, uint64_t* out_host, int records) {
  if (!ctx || !out_host || records < 1 || records > kDiagStampRecords) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_host, ctx->diag_stamps + 4 * (size_t)kDiagStampRecords, (size_t)records * sizeof(uint64_t),
                    hipMemcpyDeviceToHost));
  return FSS_AMD_OK;
}
#endif

void fss_amd_ctx_destroy(fss_amd_ctx* ctx) {
  if (!ctx) return;
  if (ctx->te0) (void)hipFree(ctx->te0);
  if (ctx->diag_stamps) (void)hipFree(ctx->diag_stamps);
  if (ctx->tail_pairs) (void)hipFree(ctx->tail_pairs);
  for (auto& kv : ctx->scratch)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& kv : ctx->prefix_ws)
    if (kv.second.p) (void)hipFree(kv.second.p);
  delete ctx;
}

int fss_amd_ctx_scratch(fss_amd_ctx* ctx, void* stream, size_t bytes, void** out) {
  if (!ctx || !out) return FSS_AMD_EINVAL;
  *out = nullptr;
  FSS_ON_DEVICE(ctx);
  std::lock_guard<std::mutex> lk(ctx->scratch_mu);
  fss_amd_ctx::Scratch& s = ctx->scratch[stream_key(stream)];
  if (s.bytes < bytes || !s.p) {
    // hipFree waits for the device, so work still queued on the old buffer has finished before it goes away
    if (s.p) HIP_TRY(hipFree(s.p));
    s.p = nullptr;
    s.bytes = 0;
    size_t want = bytes < 4096 ? 4096 : bytes;
    HIP_TRY(hipMalloc(&s.p, want));
    s.bytes = want;
  }
  *out = s.p;
  return FSS_AMD_OK;
}

int fss_amd_ctx_scratch_release(fss_amd_ctx* ctx, void* stream) {
  if (!ctx) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  void *p = nullptr, *pw = nullptr;
  {
    std::lock_guard<std::mutex> lk(ctx->scratch_mu);
    auto it = ctx->scratch.find(stream_key(stream));
    if (it != ctx->scratch.end()) {
      p = it->second.p;
      ctx->scratch.erase(it);
    }
  }
  {  // the stream's shared-prefix buffer (FSS_AMD_POINTS_AUTO) goes with it
    std::lock_guard<std::mutex> lk(ctx->auto_mu);
    auto it = ctx->prefix_ws.find(stream_key(stream));
    if (it != ctx->prefix_ws.end()) {
      pw = it->second.p;
      ctx->prefix_ws.erase(it);
    }
  }
  // wait for the device either way: work still queued on the buffers (or using the stream's work-distribution
  // counters, which are handed on below) has finished
  hipError_t e = p ? hipFree(p) : hipDeviceSynchronize();
  if (pw) {
    const hipError_t e2 = hipFree(pw);
    if (e == hipSuccess) e = e2;
  }
  // the counter pair goes back to the pool whatever the device said: the buffer is already forgotten, and a pair that
  // stayed with a dead stream would be lost for the life of the context
  ctx->tail_release(stream);
  HIP_TRY(e);
  return FSS_AMD_OK;
}

}  // extern "C"
namespace {
struct UploadBlob { uint32_t w[512]; };  // 2 KiB of kernel arguments per launch
__global__ void upload_kernel(uint8_t* dst, const UploadBlob b, int bytes) {
  const uint8_t* src = reinterpret_cast<const uint8_t*>(b.w);
  if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
    for (int i = threadIdx.x; i < bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(dst)[i] = b.w[i];
    for (int i = (bytes & ~3) + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
  }
}
}  // namespace
extern "C" {

int fss_amd_upload(fss_amd_ctx* ctx, void* dst, const void* src_host, size_t bytes, void* stream) {
  if (!ctx || (bytes && (!dst || !src_host)) || bytes > (1u << 20)) return FSS_AMD_EINVAL;
  if (bytes == 0) return FSS_AMD_OK;
  FSS_ON_DEVICE(ctx);
  for (size_t off = 0; off < bytes; off += sizeof(UploadBlob)) {
    UploadBlob b;
    const size_t m = bytes - off < sizeof(UploadBlob) ? bytes - off : sizeof(UploadBlob);
    memcpy(b.w, static_cast<const uint8_t*>(src_host) + off, m);  // the bytes travel as kernel arguments
    hipLaunchKernelGGL(upload_kernel, dim3(1), dim3(256), 0, S(stream), static_cast<uint8_t*>(dst) + off, b, (int)m);
    HIP_TRY(hipGetLastError());
  }
  return FSS_AMD_OK;
}

int fss_amd_ctx_device(const fss_amd_ctx* ctx) { return ctx ? ctx->device : -1; }

int fss_amd_ctx_read_te0(fss_amd_ctx* ctx, uint32_t out_host[256]) {
  if (!ctx || !out_host) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  HIP_TRY(hipMemcpy(out_host, ctx->te0, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return FSS_AMD_OK;
}

int fss_amd_ctx_set_prg(fss_amd_ctx* ctx, int prg, const int32_t nonce[2], int rounds) {
  if (!ctx) return FSS_AMD_EINVAL;
  if (prg == FSS_AMD_PRG_AES128_MMO) {
    ctx->prg = prg;
    return FSS_AMD_OK;
  }
  if (prg != FSS_AMD_PRG_CHACHA || rounds < 2 || rounds > 64 || (rounds & 1)) return FSS_AMD_EINVAL;
  ctx->prg = prg;
  ctx->host.chacha_nonce[0] = nonce ? (uint32_t)nonce[0] : 0u;
  ctx->host.chacha_nonce[1] = nonce ? (uint32_t)nonce[1] : 0u;
  ctx->host.chacha_rounds = rounds;
  return FSS_AMD_OK;
}

int fss_amd_ctx_set_points_plan(fss_amd_ctx* ctx, int plan) {
  if (!ctx || (plan != FSS_AMD_POINTS_AUTO && plan != FSS_AMD_POINTS_PLAIN)) return FSS_AMD_EINVAL;
  ctx->points_plan = plan;
  return FSS_AMD_OK;
}

int fss_amd_ctx_set_uint_modulus(fss_amd_ctx* ctx, uint64_t mod) {
  if (!ctx) return FSS_AMD_EINVAL;
  ctx->host.uint_mod = mod;
  return FSS_AMD_OK;
}

const char* fss_amd_strerror(int code) {
  switch (code) {
    case FSS_AMD_OK: return "ok";
    case FSS_AMD_EINVAL: return "invalid argument";
    case FSS_AMD_EHIP: return "HIP runtime error";
    case FSS_AMD_EUNSUPPORTED: return "not supported by this build";
    default: return "unknown error";
  }
}

int fss_amd_last_hip_error(void) { return g_last_hip; }

const char* fss_amd_last_kernel(const fss_amd_ctx* ctx, int* grid, int* block) {
  if (!ctx) return "";
  if (grid) *grid = ctx->last_grid;
  if (block) *block = ctx->last_block;
  return ctx->last_kernel;
}

// ------------------------------------------------------------------ one job across the GPUs of a node
int fss_amd_batch_slice(int64_t n, int rank, int world, int64_t* lo, int64_t* hi) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return FSS_AMD_EINVAL;
  const int64_t base = n / world, rem = n % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
  return FSS_AMD_OK;
}

int fss_amd_domain_block(int in_bits, int rank, int world, int64_t* first_leaf, int64_t* nleaves) {
  if (in_bits < 1 || in_bits > 62 || world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || !first_leaf || !nleaves)
    return FSS_AMD_EINVAL;
  if ((int64_t)world > ((int64_t)1 << in_bits)) return FSS_AMD_EINVAL;  // more ranks than leaves
  *nleaves = ((int64_t)1 << in_bits) / world;
  *first_leaf = rank * *nleaves;
  return FSS_AMD_OK;
}

int fss_amd_grotto_carries(const uint8_t* shard_last, int world, uint8_t* carry_in) {
  if (!shard_last || !carry_in || world < 1) return FSS_AMD_EINVAL;
  uint8_t acc = 0;
  for (int r = 0; r < world; ++r) {
    carry_in[r] = acc;
    acc ^= shard_last[r] & 1u;
  }
  return FSS_AMD_OK;
}

}  // extern "C"
namespace {
__global__ void xor_bytes_kernel(uint8_t* p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;  // bytes before the first 16-byte boundary
  const int64_t h = head < n ? head : n, nv = (n - h) >> 4;
  uint4* v = reinterpret_cast<uint4*>(p + h);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    uint4 t = v[i];
    t.x ^= 0x01010101u; t.y ^= 0x01010101u; t.z ^= 0x01010101u; t.w ^= 0x01010101u;
    v[i] = t;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = threadIdx.x; i < h; i += blockDim.x) p[i] ^= 1u;
    for (int64_t i = h + (nv << 4) + threadIdx.x; i < n; i += blockDim.x) p[i] ^= 1u;
  }
}
}  // namespace
extern "C" {

int fss_amd_grotto_apply_carry(fss_amd_ctx* ctx, void* ys_bytes, int64_t nleaves, int carry, void* stream) {
  if (!ctx || nleaves < 0 || (carry != 0 && carry != 1) || (nleaves > 0 && !ys_bytes)) return FSS_AMD_EINVAL;
  if (carry == 0 || nleaves == 0) return FSS_AMD_OK;
  FSS_ON_DEVICE(ctx);
  const int64_t want = (nleaves / 16 + 255) / 256 + 1, cap = (int64_t)ctx->num_cu * 8;
  hipLaunchKernelGGL(xor_bytes_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, S(stream),
                     reinterpret_cast<uint8_t*>(ys_bytes), nleaves);
  HIP_TRY(hipGetLastError());
  return FSS_AMD_OK;
}

// ------------------------------------------------------------------ raw PRG
int fss_amd_prg_gen(fss_amd_ctx* ctx, int mul, const void* seeds, int64_t n, void* out, void* stream) {
  if (!ctx || !seeds || !out || n < 0 || (mul != 1 && mul != 2 && mul != 4)) return FSS_AMD_EINVAL;
  if (!aligned16(seeds, out)) return FSS_AMD_EINVAL;
  if (n == 0) return FSS_AMD_OK;
  FSS_ON_DEVICE(ctx);
  // one wave per workgroup for tiny batches (the LDS fill is per workgroup), 1024 lanes otherwise
  int bs = n <= 64 ? 64 : (n <= 256 ? 256 : 1024);
  int64_t want = (n + bs - 1) / bs;
  int grid = (int)(want < (int64_t)ctx->num_cu ? want : (int64_t)ctx->num_cu);
  FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL(prg_gen_kernel<LT>, dim3(grid), dim3(bs), 0, S(stream), ctx->host, ctx->te0, mul,
                                      reinterpret_cast<const uint4*>(seeds), n, reinterpret_cast<uint4*>(out),
                                      use_dyn(n, 64, grid, bs)));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "prg_gen";
  ctx->last_grid = grid;
  ctx->last_block = bs;
  return FSS_AMD_OK;
}

// ------------------------------------------------------------------ point eval, shared key
static int points_common(fss_amd_ctx* ctx, int scheme, int in_bits, int group, int party, const void* seed,
                         const void* cws, const void* ocw, const void* xs, int x_bytes, int64_t n, void* ys,
                         void* stream, bool resident = true) {
  if (!ctx || !seed || !cws || n < 0 || !group_ok(ctx, group) || (party != 0 && party != 1) || !x_ok(in_bits, x_bytes))
    return FSS_AMD_EINVAL;
  if (n > 0 && (!xs || !ys)) return FSS_AMD_EINVAL;
  if (scheme == S_HT && !ocw) return FSS_AMD_EINVAL;
  if (!aligned16(seed, cws, ocw, ys) || !aligned_to(xs, x_bytes)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  // FSS_AMD_POINTS_AUTO (fss_amd.h): a batch large enough to pay for the key's top levels once takes the shared-prefix
  // plan -- the same outputs, about a third of the AES blocks at 2^24 points.  Not for host-resident batches (they
  // run at the PCIe rate either way), nor inside a stream capture (the plan's buffer may have to be allocated).
  if (resident && ctx->points_plan == FSS_AMD_POINTS_AUTO && x_bytes <= 8) {
    const int k = fss_amd_points_prefix_bits(scheme, in_bits, n);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (k > 0 && hipStreamIsCapturing(S(stream), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
    const size_t need = k > 0 ? fss_amd_eval_points_workspace(scheme, in_bits, k) : 0;
    if (need > 0 && cap == hipStreamCaptureStatusNone) {
      std::lock_guard<std::mutex> lk(ctx->auto_mu);
      if (void* ws = ctx->prefix_buffer(stream, need)) {
        switch (scheme) {
          case S_DPF: return fss_amd_dpf_eval_points_shared(ctx, in_bits, group, party, seed, cws, xs, x_bytes, n, ys, k, ws, need, stream);
          case S_DCF: return fss_amd_dcf_eval_points_shared(ctx, in_bits, group, party, seed, cws, xs, x_bytes, n, ys, k, ws, need, stream);
          default: return fss_amd_ht_eval_points_shared(ctx, in_bits, group, party, seed, cws, ocw, xs, x_bytes, n, ys, k, ws, need, stream);
        }
      }
    }
  }
  EvalArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = group; a.x_bytes = x_bytes; a.n = n;
  a.xs = xs; a.ys = ys;
  SharedKey k;
  k.seed = reinterpret_cast<const uint32_t*>(seed);
  k.cws = reinterpret_cast<const uint32_t*>(cws);
  k.ocw = reinterpret_cast<const uint32_t*>(ocw);
  if (n > 0 && n <= quad_points_max()) {  // small batch: one walk per quad of lanes
    switch (scheme) {
      case S_DPF: return launch_points_quad<S_DPF>(ctx, "dpf_eval_points", k, a, S(stream));
      case S_DCF: return launch_points_quad<S_DCF>(ctx, "dcf_eval_points", k, a, S(stream));
      default: return launch_points_quad<S_HT>(ctx, "ht_eval_points", k, a, S(stream));
    }
  }
  switch (scheme) {
    case S_DPF: return launch_points<S_DPF, true>(ctx, "dpf_eval_points", k, a, S(stream));
    case S_DCF: return launch_points<S_DCF, true>(ctx, "dcf_eval_points", k, a, S(stream));
    default: return launch_points<S_HT, true>(ctx, "ht_eval_points", k, a, S(stream));
  }
}

int fss_amd_dpf_eval_points(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                            const void* xs, int x_bytes, int64_t npoints, void* ys, void* stream) {
  return points_common(ctx, S_DPF, in_bits, group, party, seed, cws, nullptr, xs, x_bytes, npoints, ys, stream);
}
int fss_amd_dcf_eval_points(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                            const void* xs, int x_bytes, int64_t npoints, void* ys, void* stream) {
  return points_common(ctx, S_DCF, in_bits, group, party, seed, cws, nullptr, xs, x_bytes, npoints, ys, stream);
}
int fss_amd_ht_eval_points(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                           const void* ocw, const void* xs, int x_bytes, int64_t npoints, void* ys, void* stream) {
  return points_common(ctx, S_HT, in_bits, group, party, seed, cws, ocw, xs, x_bytes, npoints, ys, stream);
}

// ------------------------------------------------------------------ point eval, shared key, host-resident batch
// A kernel-usable address of `p` when it is pinned (hipHostMalloc / hipHostRegister: mapped into the device's address
// space) or device memory; nullptr for pageable host memory, which no kernel can touch.
static void* device_view(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // an unregistered host pointer is reported as an error
    return nullptr;
  }
  if (at.type == hipMemoryTypeHost) return at.devicePointer;
  if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeUnified) return const_cast<void*>(p);
  return nullptr;
}

int fss_amd_eval_points_host(fss_amd_ctx* ctx, int scheme, int in_bits, int group, int party, const void* seed,
                             const void* cws, const void* ocw, const void* xs_host, int x_bytes, int64_t npoints,
                             void* ys_host, void* stream) {
  if (!ctx || (scheme != FSS_AMD_DPF && scheme != FSS_AMD_DCF && scheme != FSS_AMD_HT)) return FSS_AMD_EINVAL;
  if (npoints > 0 && (!xs_host || !ys_host)) return FSS_AMD_EINVAL;
  const void* xs = xs_host;
  void* ys = ys_host;
  if (npoints > 0) {
    FSS_ON_DEVICE(ctx);
    xs = device_view(xs_host);
    ys = device_view(ys_host);
    if (!xs || !ys) return FSS_AMD_EINVAL;  // pageable memory: pin it (hipHostRegister) or stage it yourself
  }
  const int s = scheme == FSS_AMD_DPF ? S_DPF : (scheme == FSS_AMD_DCF ? S_DCF : S_HT);
  return points_common(ctx, s, in_bits, group, party, seed, cws, s == S_HT ? ocw : nullptr, xs, x_bytes, npoints, ys, stream,
                       /*resident=*/false);
}

// ------------------------------------------------------------------ relayout
static int dpf_relayout_common(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* tr_bits,
                               void* out_cw, int tr32, void* stream) {
  if (!ctx || nkeys < 0 || in_bits < 1 || in_bits > (tr32 ? 32 : 128)) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!cws || !cw_s || !tr_bits || !out_cw) return FSS_AMD_EINVAL;
  if (!aligned16(cws, cw_s, out_cw) || !aligned_to(tr_bits, tr32 ? 4 : 8)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  launch_relayout<S_DPF>(ctx, S(stream), in_bits,
                     in_bits + 1, reinterpret_cast<const uint4*>(cws), nkeys, reinterpret_cast<uint4*>(cw_s),
                     (uint4*)nullptr, reinterpret_cast<uint64_t*>(tr_bits), reinterpret_cast<uint4*>(out_cw),
                     (uint32_t*)nullptr, tr32);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "relayout";
  ctx->last_grid = (int)((nkeys + 63) / 64);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}
int fss_amd_dpf_relayout(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* tr_bits,
                         void* out_cw, void* stream) {
  return dpf_relayout_common(ctx, in_bits, cws, nkeys, cw_s, tr_bits, out_cw, 0, stream);
}
int fss_amd_dpf_relayout_u32(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* extra,
                             void* out_cw, void* stream) {
  return dpf_relayout_common(ctx, in_bits, cws, nkeys, cw_s, extra, out_cw, 1, stream);
}
int fss_amd_dcf_relayout(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* cw_v,
                         void* out_cw, void* stream) {
  if (!ctx || nkeys < 0 || in_bits < 1 || in_bits > 128) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!cws || !cw_s || !cw_v || !out_cw) return FSS_AMD_EINVAL;
  if (!aligned16(cws, cw_s, cw_v, out_cw)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  launch_relayout<S_DCF>(ctx, S(stream), in_bits,
                     in_bits + 1, reinterpret_cast<const uint4*>(cws), nkeys, reinterpret_cast<uint4*>(cw_s),
                     reinterpret_cast<uint4*>(cw_v), (uint64_t*)nullptr, reinterpret_cast<uint4*>(out_cw),
                     (uint32_t*)nullptr);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "relayout";
  ctx->last_grid = (int)((nkeys + 63) / 64);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}
int fss_amd_ht_relayout(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* extra,
                        void* stream) {
  if (!ctx || nkeys < 0 || in_bits < 1 || in_bits > 128) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!cws || !cw_s || !extra) return FSS_AMD_EINVAL;
  if (!aligned16(cws, cw_s) || !aligned_to(extra, 4)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  launch_relayout<S_HT>(ctx, S(stream), in_bits,
                     in_bits, reinterpret_cast<const uint4*>(cws), nkeys, reinterpret_cast<uint4*>(cw_s),
                     (uint4*)nullptr, (uint64_t*)nullptr, (uint4*)nullptr, reinterpret_cast<uint32_t*>(extra));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "relayout";
  ctx->last_grid = (int)((nkeys + 63) / 64);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}

// ------------------------------------------------------------------ level-major -> key-major
}  // extern "C"

template <int SCHEME>
static int key_major_common(fss_amd_ctx* ctx, int in_bits, const void* cw_s, const void* cw_v, const void* tr_bits,
                            const void* out_cw, const void* extra, int64_t nkeys, void* cws, void* stream) {
  if (!ctx || nkeys < 0 || in_bits < 1 || in_bits > 128) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!cw_s || !cws || (SCHEME == S_DPF && (!tr_bits || !out_cw)) || (SCHEME == S_DCF && (!cw_v || !out_cw)) ||
      (SCHEME == S_HT && !extra))
    return FSS_AMD_EINVAL;
  if (!aligned16(cw_s, cw_v, out_cw, cws) || !aligned_to(tr_bits, 8) || !aligned_to(extra, 4)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  const int slots = SCHEME == S_HT ? in_bits : in_bits + 1;
  int64_t want = (nkeys + 255) / 256, cap = (int64_t)ctx->num_cu * 8;
  if (relayout_mode(slots) >= 2) {  // whole lines on both sides (relayout_flat_kernel's mirror image)
    const int bs = flat_block(slots), nw = bs / 64;
    want = ((nkeys + kFlatKeys - 1) / kFlatKeys + nw - 1) / nw;
    hipLaunchKernelGGL(key_major_flat_kernel<SCHEME>, dim3((unsigned)(want < cap ? want : cap)), dim3(bs),
                       flat_lds_bytes(slots, nw), S(stream), in_bits, slots, reinterpret_cast<const uint4*>(cw_s),
                       reinterpret_cast<const uint4*>(cw_v), reinterpret_cast<const uint64_t*>(tr_bits),
                       reinterpret_cast<const uint4*>(out_cw), reinterpret_cast<const uint32_t*>(extra), nkeys,
                       reinterpret_cast<uint4*>(cws));
  } else
  hipLaunchKernelGGL(key_major_kernel<SCHEME>, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, S(stream), in_bits,
                     reinterpret_cast<const uint4*>(cw_s), reinterpret_cast<const uint4*>(cw_v),
                     reinterpret_cast<const uint64_t*>(tr_bits), reinterpret_cast<const uint4*>(out_cw),
                     reinterpret_cast<const uint32_t*>(extra), nkeys, reinterpret_cast<uint4*>(cws));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "key_major";
  ctx->last_grid = (int)(want < cap ? want : cap);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}
extern "C" {

int fss_amd_dpf_key_major(fss_amd_ctx* ctx, int in_bits, const void* cw_s, const void* tr_bits, const void* out_cw,
                          int64_t nkeys, void* cws, void* stream) {
  return key_major_common<S_DPF>(ctx, in_bits, cw_s, nullptr, tr_bits, out_cw, nullptr, nkeys, cws, stream);
}
int fss_amd_dcf_key_major(fss_amd_ctx* ctx, int in_bits, const void* cw_s, const void* cw_v, const void* out_cw,
                          int64_t nkeys, void* cws, void* stream) {
  return key_major_common<S_DCF>(ctx, in_bits, cw_s, cw_v, nullptr, out_cw, nullptr, nkeys, cws, stream);
}
int fss_amd_ht_key_major(fss_amd_ctx* ctx, int in_bits, const void* cw_s, const void* extra, int64_t nkeys, void* cws,
                         void* stream) {
  return key_major_common<S_HT>(ctx, in_bits, cw_s, nullptr, nullptr, nullptr, extra, nkeys, cws, stream);
}

// ------------------------------------------------------------------ point eval, independent keys
static int keys_common(fss_amd_ctx* ctx, int scheme, int in_bits, int group, int party, const void* seeds,
                       const void* cw_s, const void* cw_v, const void* tr_bits, const void* out_cw, const void* extra,
                       const void* xs, int x_bytes, int64_t n, void* ys, void* stream, int tr32 = 0) {
  if (!ctx || n < 0 || !group_ok(ctx, group) || (party != 0 && party != 1) || !x_ok(in_bits, x_bytes)) return FSS_AMD_EINVAL;
  if (tr32 && in_bits > 32) return FSS_AMD_EINVAL;
  if (n == 0) return FSS_AMD_OK;
  if (!seeds || !cw_s || !xs || !ys || !out_cw) return FSS_AMD_EINVAL;
  if (scheme == S_DPF && !tr_bits) return FSS_AMD_EINVAL;
  if (scheme == S_DCF && !cw_v) return FSS_AMD_EINVAL;
  if (scheme == S_HT && !extra) return FSS_AMD_EINVAL;
  if (!aligned16(seeds, cw_s, cw_v, out_cw, ys) || !aligned_to(tr_bits, tr32 ? 4 : 8) || !aligned_to(extra, 4) ||
      !aligned_to(xs, x_bytes))
    return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  EvalArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = group; a.x_bytes = x_bytes; a.n = n;
  a.xs = xs; a.ys = ys;
  a.seeds = reinterpret_cast<const uint4*>(seeds);
  a.cw_s = reinterpret_cast<const uint4*>(cw_s);
  a.cw_v = reinterpret_cast<const uint4*>(cw_v);
  a.tr_bits = reinterpret_cast<const uint64_t*>(tr_bits);
  a.tr32 = tr32;
  a.out_cw = reinterpret_cast<const uint4*>(out_cw);
  a.extra = reinterpret_cast<const uint32_t*>(extra);
  if (n <= quad_points_max()) {  // small batch: a quad of lanes per key (points_common)
    const bool oct = quad_octets_pay(ctx, scheme, n);  // DCF / AES, few keys: an octet of lanes per key (fss_quad.hpp)
    const int64_t lanes = n * (oct ? 8 : 4);
    const int bs = lanes >= 65536 ? 1024 : 256;
    const int64_t want = (lanes + bs - 1) / bs, cap = ctx->num_cu;
    const dim3 grid((unsigned)(want < cap ? want : cap)), block(bs);
    switch (scheme) {
      case S_DPF: FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((keys_quad_kernel<LT, S_DPF>), grid, block, 0, S(stream), ctx->host, ctx->te0, a)); break;
      case S_DCF:
        if (oct) FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((keys_quad_kernel<LT, S_DCF, true>), grid, block, 0, S(stream), ctx->host, ctx->te0, a));
        else FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((keys_quad_kernel<LT, S_DCF>), grid, block, 0, S(stream), ctx->host, ctx->te0, a));
        break;
      default: FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((keys_quad_kernel<LT, S_HT>), grid, block, 0, S(stream), ctx->host, ctx->te0, a)); break;
    }
    HIP_TRY(hipGetLastError());
    ctx->last_kernel = scheme == S_DPF ? "dpf_eval_keys" : (scheme == S_DCF ? "dcf_eval_keys" : "ht_eval_keys");
    ctx->last_grid = (int)grid.x;
    ctx->last_block = bs;
    return FSS_AMD_OK;
  }
  switch (scheme) {
    case S_DPF: return launch_points<S_DPF, false>(ctx, "dpf_eval_keys", SharedKey{}, a, S(stream));
    case S_DCF: return launch_points<S_DCF, false>(ctx, "dcf_eval_keys", SharedKey{}, a, S(stream));
    default: return launch_points<S_HT, false>(ctx, "ht_eval_keys", SharedKey{}, a, S(stream));
  }
}

int fss_amd_dpf_eval_keys(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                          const void* tr_bits, const void* out_cw, const void* xs, int x_bytes, int64_t nkeys, void* ys,
                          void* stream) {
  return keys_common(ctx, S_DPF, in_bits, group, party, seeds, cw_s, nullptr, tr_bits, out_cw, nullptr, xs, x_bytes,
                     nkeys, ys, stream);
}
int fss_amd_dpf_eval_keys_u32(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                              const void* extra, const void* out_cw, const void* xs, int x_bytes, int64_t nkeys, void* ys,
                              void* stream) {
  return keys_common(ctx, S_DPF, in_bits, group, party, seeds, cw_s, nullptr, extra, out_cw, nullptr, xs, x_bytes,
                     nkeys, ys, stream, 1);
}
int fss_amd_dcf_eval_keys(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                          const void* cw_v, const void* out_cw, const void* xs, int x_bytes, int64_t nkeys, void* ys,
                          void* stream) {
  return keys_common(ctx, S_DCF, in_bits, group, party, seeds, cw_s, cw_v, nullptr, out_cw, nullptr, xs, x_bytes,
                     nkeys, ys, stream);
}
int fss_amd_ht_eval_keys(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                         const void* extra, const void* ocws, const void* xs, int x_bytes, int64_t nkeys, void* ys,
                         void* stream) {
  return keys_common(ctx, S_HT, in_bits, group, party, seeds, cw_s, nullptr, nullptr, ocws, extra, xs, x_bytes, nkeys,
                     ys, stream);
}

}  // extern "C"

