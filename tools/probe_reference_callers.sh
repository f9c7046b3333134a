#!/bin/bash
# Compile-only probe, build container only: do the reference's OWN caller sources -- its five in-scope samples, its
# GPU-vs-CPU checker, its six gtest files and its two benchmark files -- compile UNCHANGED against this library's headers
# (include/fss/*.cuh -> include/fss_amd.hpp)?  Nothing is copied, linked, shipped or run: the sources are read where they
# lie under /root/reference, objects go to a temporary directory that is removed, and the three things a CUDA caller's
# translation unit expects from its toolchain rather than from the library are supplied from that same temporary
# directory for the duration of the probe only -- the cuda* runtime names as hip* macros, `cuda::std` as `std`, and empty
# declarations for gtest / Google Benchmark (neither is in the image).  They are NOT part of the library (which has no
# CUDA spellings: a caller ports those as INTEGRATION.md section 2 says) and never leave /tmp.
#   tools/probe_reference_callers.sh            ->  one line per file: ok | FAILED (+ the first errors)
# Known and not the library's: src/group_test.cu:34 is a function template partial specialisation, which clang rejects
# whatever it is compiled against; samples/dpf_dcf_gpu.cu:27 narrows 0x9abcdef0 into an int (-Wno-c++11-narrowing).
set -u
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/src" ] || { echo "no reference tree at $REF: this probe runs in the build container only"; exit 0; }
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/cuda/std" "$T/gtest" "$T/benchmark"
{
  echo '#pragma once'
  echo '#include <hip/hip_runtime.h>'
  for n in Error_t:hipError_t Success GetErrorString Malloc Free Memcpy MemcpyAsync Memset MemcpyHostToDevice MemcpyDeviceToHost \
           MemcpyDeviceToDevice Memcpy2D MemcpyToSymbol DeviceSynchronize GetLastError PeekAtLastError Stream_t:hipStream_t StreamCreate \
           StreamDestroy StreamSynchronize GetDeviceCount SetDevice Event_t:hipEvent_t EventCreate EventRecord EventSynchronize EventElapsedTime EventDestroy; do
    c=${n%%:*}; h=${n#*:}; [ "$h" = "$n" ] && h="hip$n"; echo "#define cuda$c $h"
  done
} > "$T/cuda_runtime.h"
for h in array span tuple; do printf '#pragma once\n#include <%s>\nnamespace cuda { namespace std { using namespace ::std; } }\n' $h > "$T/cuda/std/$h"; done
printf '#pragma once\n#include <array>\n#include <span>\n#include <tuple>\nnamespace cuda { namespace std { using namespace ::std; } }\n' > "$T/pre.h"
cat > "$T/gtest/gtest.h" <<'G'
#pragma once
namespace testing {
struct Test { virtual ~Test() {} virtual void SetUp() {} virtual void TearDown() {} };
template <class... T> struct Types {};
struct Sink { template <class T> Sink& operator<<(const T&) { return *this; } };
}
#define GT_CAT2(a, b) a##b
#define GT_CAT(a, b) GT_CAT2(a, b)
#define TEST_F(Fix, Name) struct GT_CAT(Fix, Name) : Fix { void Body(); }; void GT_CAT(Fix, Name)::Body()
#define TYPED_TEST_SUITE(Fix, Types) using GT_CAT(Fix, _Types) = Types
#define TYPED_TEST(Fix, Name) template <class TypeParam> struct GT_CAT(Fix, Name) : Fix<TypeParam> { void Body(); }; \
  template <class TypeParam> void GT_CAT(Fix, Name)<TypeParam>::Body()
#define EXPECT_EQ(a, b) if ((a) == (b)) ; else ::testing::Sink()
#define EXPECT_TRUE(a) if (a) ; else ::testing::Sink()
G
cat > "$T/benchmark/benchmark.h" <<'B'
#pragma once
#include <cstdint>
#include <string>
namespace benchmark {
struct State {
  struct It { int i; bool operator!=(const It& o) const { return i != o.i; } void operator++() { ++i; } int operator*() const { return i; } };
  It begin() { return {0}; } It end() { return {1}; }
  int64_t iterations() const { return 1; }
  void SetItemsProcessed(int64_t) {}
  void SetIterationTime(double) {}
};
template <class T> inline void DoNotOptimize(T&& v) { asm volatile("" : : "g"(&v) : "memory"); }
struct Bench { Bench* Name(const std::string&) { return this; } Bench* UseManualTime() { return this; } };
inline Bench* Register(void (*)(State&)) { static Bench b; return &b; }
template <class F> inline Bench* RegisterBenchmark(const char*, F) { static Bench b; return &b; }
inline void Initialize(int*, char**) {}
inline void RunSpecifiedBenchmarks() {}
inline void Shutdown() {}
}
#define BM_CAT2(a, b) a##b
#define BM_CAT(a, b) BM_CAT2(a, b)
#define BENCHMARK(...) static ::benchmark::Bench* BM_CAT(bm_reg_, __COUNTER__) = ::benchmark::Register(__VA_ARGS__)
#define BENCHMARK_MAIN() int main() { return 0; }
B
fail=0
for f in samples/dpf_dcf_cpu.cu samples/half_tree_dpf_cpu.cu samples/grotto_dcf_cpu.cu samples/dpf_dcf_gpu.cu samples/vdpf_cpu.cu \
         check_evalall_gpu.cu src/dpf_test.cu src/dcf_test.cu src/half_tree_dpf_test.cu src/grotto_dcf_test.cu src/vdpf_test.cu \
         src/group_test.cu src/bench_cpu.cu src/bench_gpu.cu third_party/fss/bench.cu; do
  if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++20 -x hip -Wno-c++11-narrowing -Wno-unused-value -include "$T/pre.h" \
       -I "$T" -I "$ROOT/include" -c "$REF/$f" -o "$T/o.o" > "$T/log" 2>&1; then
    echo "ok      $f"
  else
    echo "FAILED  $f"; grep -m3 "error" "$T/log" | sed 's/^/          /'
    [ "$f" = src/group_test.cu ] || fail=1
  fi
done
exit $fail
