# ChaCha double round, W independent states per lane interleaved op-by-op (runs of 4W same-class ops)
def qr(a,b,c,d):
    return [f"v_add_u32 v{a}, v{b}, v{a}", f"v_xor_b32 v{d}, v{a}, v{d}", f"v_alignbit_b32 v{d}, v{d}, v{d}, 16",
            f"v_add_u32 v{c}, v{d}, v{c}", f"v_xor_b32 v{b}, v{c}, v{b}", f"v_alignbit_b32 v{b}, v{b}, v{b}, 20",
            f"v_add_u32 v{a}, v{b}, v{a}", f"v_xor_b32 v{d}, v{a}, v{d}", f"v_alignbit_b32 v{d}, v{d}, v{d}, 24",
            f"v_add_u32 v{c}, v{d}, v{c}", f"v_xor_b32 v{b}, v{c}, v{b}", f"v_alignbit_b32 v{b}, v{b}, v{b}, 25"]
def interleave(qs):
    out=[]
    for i in range(12):
        for q in qs: out.append(q[i])
    return out
def dround(W, merge_addxor=False):
    col=[];dia=[]
    for w in range(W):
        x=lambda i: 16*w+i
        col+=[qr(x(0),x(4),x(8),x(12)),qr(x(1),x(5),x(9),x(13)),qr(x(2),x(6),x(10),x(14)),qr(x(3),x(7),x(11),x(15))]
        dia+=[qr(x(0),x(5),x(10),x(15)),qr(x(1),x(6),x(11),x(12)),qr(x(2),x(7),x(8),x(13)),qr(x(3),x(4),x(9),x(14))]
    return interleave(col)+interleave(dia)
src=['#include <hip/hip_runtime.h>','#include <cstdio>','#include <cstdint>','#define REP 1024']
names=[]
for W in (1,2,4,6):
    ins=dround(W)
    body="\\n".join(ins)
    regs=list(range(16*W))
    clob=",".join(f'"v{r}"' for r in regs)
    inits="\\n".join(f"v_mov_b32 v{r}, %0" for r in regs)
    fin="\\n".join(f"v_xor_b32 %0, v{r}, %0" for r in regs)
    k=f"k_w{W}"
    names.append((k,len(ins),W))
    src.append(f'''__global__ __launch_bounds__(512) void {k}(uint32_t* out, uint32_t a) {{
  uint32_t t = threadIdx.x + a;
  asm volatile("{inits}\\n s_movk_i32 s20, 1024\\n 1:\\n {body}\\n s_sub_u32 s20, s20, 1\\n s_cmp_lg_u32 s20, 0\\n s_cbranch_scc1 1b\\n {fin}" : "+v"(t) : : {clob}, "s20", "scc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}}''')
src.append('''template <typename K> void run(const char* name, K kern, uint32_t* out, int blocks, int threads, int nins, int W) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double waves_per_simd = (double)blocks * threads / 64 / 1024;
  double ns_per_loop = ms * 1e6 / (waves_per_simd * REP);
  printf("%-8s %4d x %4d (%2.0f waves/SIMD): %.3f ms  %.2f clk/inst  %.0f clk per double round per state @2.36GHz\\n", name, blocks, threads, waves_per_simd, ms, ns_per_loop * 2.36 / nins, ns_per_loop * 2.36 / W);
}
int main() { uint32_t* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
  for (int blocks : {256, 512, 1024}) {''')
for k,n,W in names: src.append(f'    run("{k}", {k}, out, blocks, 512, {n}, {W});')
src.append('  }\n  return 0; }')
open("chacha_ilp.hip","w").write("\n".join(src))
