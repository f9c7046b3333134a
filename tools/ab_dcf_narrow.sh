#!/bin/bash
# DCF point walk with the narrow value block (312 instead of 320 lookups per level for groups of at most 64 bits)
# against the previous library, interleaved.  fss_amd/libfss_amd_prev.so = the library of the commit before.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3/dcf_narrow.txt"; mkdir -p "$(dirname "$O")"; : > "$O"
for rep in 1 2 3; do
  for lg in 20 24; do
    for lib in libfss_amd_prev.so libfss_amd.so; do
      echo -n "[$lib 2^$lg] " >> "$O"
      FSS_AMD_LIB="$R/fss_amd/$lib" python3 tools/bench_points.py --scheme dcf --n 32 --log2-points $lg --steady 2>/dev/null | tail -1 >> "$O"
    done
  done
done
for lib in libfss_amd_prev.so libfss_amd.so; do
  echo -n "[$lib keys] " >> "$O"
  FSS_AMD_LIB="$R/fss_amd/$lib" python3 tools/bench_keys.py --scheme dcf --n 32 --steady 2>/dev/null | tail -1 >> "$O"
done
cat "$O"
