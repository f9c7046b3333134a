/* include/fss_amd.h is a C header: this file is compiled as C99 with -pedantic (tests/test_abi.py), linked against
 * libfss_amd.so, and calls the entry points that need no GPU -- the slice arithmetic a multi-GPU caller uses
 * (fss_amd_batch_slice / _domain_block / _grotto_carries), the plan query, the error strings -- the way a C FFI of the
 * reference (cgo / ctypes / JNI) would bind them. */
#include "fss_amd.h"

#include <stdio.h>

int main(void) {
  int64_t lo = 0, hi = 0, at = 0, first = 0, nleaves = 0;
  uint8_t last[4] = {1, 0, 1, 1}, carry[4];
  int r;
  for (r = 0; r < 7; ++r) {
    if (fss_amd_batch_slice(100, r, 7, &lo, &hi) != FSS_AMD_OK || lo != at || hi - lo < 14 || hi - lo > 15) return 1;
    at = hi;
  }
  if (at != 100) return 2;
  if (fss_amd_batch_slice(100, 7, 7, &lo, &hi) != FSS_AMD_EINVAL) return 3;
  if (fss_amd_domain_block(28, 5, 8, &first, &nleaves) != FSS_AMD_OK || nleaves != ((int64_t)1 << 25) || first != 5 * nleaves) return 4;
  if (fss_amd_domain_block(28, 0, 3, &first, &nleaves) != FSS_AMD_EINVAL) return 5;
  if (fss_amd_grotto_carries(last, 4, carry) != FSS_AMD_OK || carry[0] != 0 || carry[1] != 1 || carry[2] != 1 || carry[3] != 0) return 6;
  if (fss_amd_points_prefix_bits(FSS_AMD_DPF, 32, (int64_t)1 << 24) != 23 || fss_amd_points_prefix_bits(FSS_AMD_HT, 12, (int64_t)1 << 24) != 11 ||
      fss_amd_points_prefix_bits(FSS_AMD_DCF, 32, 1000) != 0)
    return 7;
  if (fss_amd_eval_points_workspace(FSS_AMD_DPF, 32, 23) < ((size_t)16 << 23)) return 8;
  if (fss_amd_eval_all_workspace(FSS_AMD_DPF, 28, (int64_t)1 << 25) == 0) return 9;
  printf("%s / %s\n", fss_amd_strerror(FSS_AMD_OK), fss_amd_strerror(FSS_AMD_EINVAL));
  puts("OK");
  return 0;
}
