# which pairs of "2-cycle" VALU ops keep the 2-cycle rate when interleaved?  16 independent registers.
ops={
 "xor":"v_xor_b32 v{r}, v{s}, v{r}",
 "add":"v_add_u32 v{r}, v{s}, v{r}",
 "or":"v_or_b32 v{r}, v{s}, v{r}",
 "and":"v_and_b32 v{r}, v{s}, v{r}",
 "lshl":"v_lshlrev_b32 v{r}, 3, v{r}",
 "lshr":"v_lshrrev_b32 v{r}, 3, v{r}",
 "lshlv":"v_lshlrev_b32 v{r}, v{s}, v{r}",
 "mov":"v_mov_b32 v{r}, v{s}",
 "bitop3":"v_bitop3_b32 v{r}, v{r}, v{s}, v{r} bitop3:0x96",
 "sub":"v_sub_u32 v{r}, v{s}, v{r}",
 "align":"v_alignbit_b32 v{r}, v{r}, v{r}, 16",
}
def pat(seq, dist=16):
    ins=[]
    for rep in range(48//len(seq)):
        for j,o in enumerate(seq):
            i=rep*len(seq)+j
            ins.append(ops[o].format(r=i%dist, s=16+i%4))
    return ins
tests=[("xor",["xor"]),("align",["align"]),("xor3/align",["xor"]*3+["align"]),("xor7/align",["xor"]*7+["align"]),("xor15/align",["xor"]*15+["align"]),("xor47/align",["xor"]*47+["align"]),("xor32/align16",["xor"]*32+["align"]*16),("xor40/align8",["xor"]*40+["align"]*8),("xor d1",None),("xor/align d1",None)]
_unused=[("xor",["xor"]),("add",["add"]),("lshl",["lshl"]),("lshlv",["lshlv"]),("bitop3",["bitop3"]),
 ("xor/add",["xor","add"]),("xor/or",["xor","or"]),("xor/and",["xor","and"]),("xor/lshl",["xor","lshl"]),("xor/lshlv",["xor","lshlv"]),("lshl/lshr",["lshl","lshr"]),
 ("xor/mov",["xor","mov"]),("xor/bitop3",["xor","bitop3"]),("add/sub",["add","sub"]),("add/bitop3",["add","bitop3"]),("add/lshl",["add","lshl"]),
 ("xor4/add4",["xor"]*4+["add"]*4),("xor8/add8",["xor"]*8+["add"]*8),("xor/align",["xor","align"]),("xor3/align",["xor"]*3+["align"]),
 ("xor d1",None),("xor d2",None),("xor d4",None),("xor d8",None),("add d4",None),("xor/add d4",None)]
src=['#include <hip/hip_runtime.h>','#include <cstdio>','#include <cstdint>','#define REP 2048']
names=[]
for idx,(name,seq) in enumerate(tests):
    if seq is None:
        parts=name.split(" d"); d=int(parts[1]); ins=pat(parts[0].split("/"),d)
    else: ins=pat(seq)
    body="\\n".join(ins)
    regs=list(range(24))
    clob=",".join(f'"v{r}"' for r in regs)
    inits="\\n".join(f"v_mov_b32 v{r}, %0" for r in regs)
    fin="\\n".join(f"v_xor_b32 %0, v{r}, %0" for r in regs)
    k=f"k_{idx}"
    names.append((k,name,len(ins)))
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
  printf("%-14s %4d x %4d: %.3f ms  %.2f clk/inst @2.36GHz\\n", name, blocks, threads, ms, ns_per_loop * 2.36 / nins);
}
int main() { uint32_t* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
  for (int threads : {256, 512, 1024, 2048}) { int blocks = threads > 1024 ? 512 : 256; int th = threads > 1024 ? 1024 : threads;''')
for k,name,n in names: src.append(f'    run("{name}", {k}, out, blocks, th, {n});')
src.append('  }\n  return 0; }')
open("valu_mode.hip","w").write("\n".join(src))
