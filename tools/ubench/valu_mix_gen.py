# mix experiments: where does v_alignbit's cost come from when mixed with 2-cycle ops?
R=list(range(16))
def body_mix(kind):
    ins=[]
    if kind=="indep_xxa":      # per register: xor, add, alignbit — self-dependent only, grouped by 4 like ChaCha
        for g in range(0,16,4):
            for op in ("v_add_u32 v{r}, v{s}, v{r}","v_xor_b32 v{r}, v{s}, v{r}","v_alignbit_b32 v{r}, v{r}, v{r}, 16"):
                for r in range(g,g+4): ins.append(op.format(r=r,s=16+(r%4)))
    elif kind=="alt_xxa":      # x x a alternating, 16 independent registers
        for r in range(16):
            ins += [f"v_add_u32 v{r}, v{16+r%4}, v{r}", f"v_xor_b32 v{(r+5)%16}, v{16+r%4}, v{(r+5)%16}", f"v_alignbit_b32 v{(r+9)%16}, v{(r+9)%16}, v{(r+9)%16}, 16"]
    elif kind=="runs":         # 32 two-cycle ops, then 16 alignbits
        for r in range(16): ins.append(f"v_add_u32 v{r}, v{16+r%4}, v{r}")
        for r in range(16): ins.append(f"v_xor_b32 v{r}, v{16+r%4}, v{r}")
        for r in range(16): ins.append(f"v_alignbit_b32 v{r}, v{r}, v{r}, 16")
    elif kind=="align_only":
        for k in range(3):
            for r in range(16): ins.append(f"v_alignbit_b32 v{r}, v{r}, v{r}, 16")
    elif kind=="xor_only":
        for k in range(3):
            for r in range(16): ins.append(f"v_xor_b32 v{r}, v{16+r%4}, v{r}")
    elif kind=="rot3":         # rotate as lshl + lshr + or (three 2-cycle ops), grouped by 4
        for g in range(0,16,4):
            for r in range(g,g+4): ins.append(f"v_add_u32 v{r}, v{16+r%4}, v{r}")
            for r in range(g,g+4): ins.append(f"v_xor_b32 v{r}, v{16+r%4}, v{r}")
            for r in range(g,g+4): ins.append(f"v_lshlrev_b32 v{20+r%4}, 16, v{r}")
            for r in range(g,g+4): ins.append(f"v_lshrrev_b32 v{r}, 16, v{r}")
            for r in range(g,g+4): ins.append(f"v_or_b32 v{r}, v{20+r%4}, v{r}")
    elif kind=="perm_mix":     # v_perm (selector in VGPR) instead of alignbit
        for g in range(0,16,4):
            for op in ("v_add_u32 v{r}, v{s}, v{r}","v_xor_b32 v{r}, v{s}, v{r}","v_perm_b32 v{r}, v{r}, v{r}, v19"):
                for r in range(g,g+4): ins.append(op.format(r=r,s=16+(r%4)))
    elif kind=="bfe_mix":
        for g in range(0,16,4):
            for op in ("v_add_u32 v{r}, v{s}, v{r}","v_xor_b32 v{r}, v{s}, v{r}","v_lshl_or_b32 v{r}, v{r}, 16, v{s}"):
                for r in range(g,g+4): ins.append(op.format(r=r,s=16+(r%4)))
    elif kind=="align_nop":    # alignbit followed by s_nop
        for g in range(0,16,4):
            for op in ("v_add_u32 v{r}, v{s}, v{r}","v_xor_b32 v{r}, v{s}, v{r}","v_alignbit_b32 v{r}, v{r}, v{r}, 16"):
                for r in range(g,g+4): ins.append(op.format(r=r,s=16+(r%4)))
            ins.append("s_nop 0")
    elif kind=="align_2src":   # alignbit with two different sources (still a funnel shift)
        for g in range(0,16,4):
            for op in ("v_add_u32 v{r}, v{s}, v{r}","v_xor_b32 v{r}, v{s}, v{r}","v_alignbit_b32 v{r}, v{r}, v{s}, 16"):
                for r in range(g,g+4): ins.append(op.format(r=r,s=16+(r%4)))
    return ins
kinds=["indep_xxa","alt_xxa","runs","align_only","xor_only","rot3","perm_mix","bfe_mix","align_nop","align_2src"]
src=['#include <hip/hip_runtime.h>','#include <cstdio>','#include <cstdint>','#define REP 2048']
names=[]
for kind in kinds:
    ins=body_mix(kind)
    nv=sum(1 for s in ins if s.startswith("v_"))
    body="\\n".join(ins)
    regs=list(range(24))
    clob=",".join(f'"v{r}"' for r in regs)
    inits="\\n".join(f"v_mov_b32 v{r}, %0" for r in regs)
    fin="\\n".join(f"v_xor_b32 %0, v{r}, %0" for r in regs)
    k=f"k_{kind}"
    names.append((k,nv))
    src.append(f'''__global__ __launch_bounds__(1024) void {k}(uint32_t* out, uint32_t a) {{
  uint32_t t = threadIdx.x + a;
  asm volatile("{inits}\\n s_movk_i32 s20, 2048\\n 1:\\n {body}\\n s_sub_u32 s20, s20, 1\\n s_cmp_lg_u32 s20, 0\\n s_cbranch_scc1 1b\\n {fin}" : "+v"(t) : : {clob}, "s20", "scc");
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
  double ns_per_loop = ms * 1e6 / (waves_per_simd * REP);
  printf("%-14s %4d x %4d: %.3f ms  %3d VALU/loop  %.1f clk/loop @2.36GHz = %.2f clk/inst\\n", name, blocks, threads, ms, nins, ns_per_loop * 2.36, ns_per_loop * 2.36 / nins);
}
int main() { uint32_t* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
  for (int blocks : {512}) {''')
for k,n in names: src.append(f'    run("{k}", {k}, out, blocks, 1024, {n});')
src.append('  }\n  return 0; }')
open("valu_mix.hip","w").write("\n".join(src))
