# generate chacha_rate.hip: double-round loops with hard-coded VGPR layouts
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
def dround(m):
    x=lambda i: m[i]
    col=[qr(x(0),x(4),x(8),x(12)),qr(x(1),x(5),x(9),x(13)),qr(x(2),x(6),x(10),x(14)),qr(x(3),x(7),x(11),x(15))]
    dia=[qr(x(0),x(5),x(10),x(15)),qr(x(1),x(6),x(11),x(12)),qr(x(2),x(7),x(8),x(13)),qr(x(3),x(4),x(9),x(14))]
    return interleave(col)+interleave(dia)
layouts={
 "rows": list(range(16)),                                  # x[i] in v_i: a,b,c,d of a column share a bank
 "banked": [4*(i%4)+(i//4) for i in range(16)],            # a in bank 0, b bank 1, c bank 2, d bank 3
 "compiler": [19,18,21,20, 23,14,13,12, 27,26,17,16, 25,22,24,15],
}
def noalign(ins): return [s if "alignbit" not in s else s.replace("v_alignbit_b32","v_xor_b32").rsplit(",",1)[0] for s in ins]
src=['#include <hip/hip_runtime.h>','#include <cstdio>','#include <cstdint>','#define REP 2048']
names=[]
for name,m in layouts.items():
    for variant in ("", "_noalign"):
        ins=dround(m)
        if variant: ins=noalign(ins)
        body="\\n".join(ins)
        clob=",".join(f'"v{r}"' for r in sorted(set(m)))
        k=f"k_{name}{variant}"
        names.append((k,len(ins)))
        inits="\\n".join(f"v_add_u32 v{r}, %0, {r*77+1}" if False else f"v_mov_b32 v{r}, %0" for r in m)
        fin="\\n".join(f"v_xor_b32 %0, v{r}, %0" for r in m)
        src.append(f'''__global__ __launch_bounds__(1024) void {k}(uint32_t* out, uint32_t a) {{
  uint32_t t = threadIdx.x + a;
  asm volatile("{inits}\\n s_movk_i32 s20, {2048}\\n 1:\\n {body}\\n s_sub_u32 s20, s20, 1\\n s_cmp_lg_u32 s20, 0\\n s_cbranch_scc1 1b\\n {fin}" : "+v"(t) : : {clob}, "s20", "scc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}}''')
src.append('''template <typename K> void run(const char* name, K kern, uint32_t* out, int blocks, int threads, int nins) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double waves_per_simd = (double)blocks * threads / 64 / 1024;
  double ns_per_inst = ms * 1e6 / (waves_per_simd * REP * nins);
  printf("%-22s %4d x %4d: %.3f ms  %.3f ns/wave-inst/SIMD = %.2f clk @2.36GHz (%.2f @2.1)\\n", name, blocks, threads, ms, ns_per_inst, ns_per_inst * 2.36, ns_per_inst*2.1);
}
int main() { uint32_t* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
  for (int blocks : {256, 512}) {''')
for k,n in names: src.append(f'    run("{k}", {k}, out, blocks, 1024, {n});')
src.append('  }\n  return 0; }')
open("chacha_rate.hip","w").write("\n".join(src))
