#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Sweep the launch tunables of the point-eval kernels (one process per setting).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for bs in 1024 512 256; do for ilp in 1 2; do for w in 0 1; do
  echo -n "BS=$bs ILP=$ilp WGS_PER_CU=$w : "
  FSS_AMD_BS=$bs FSS_AMD_ILP=$ilp FSS_AMD_WGS_PER_CU=$w python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f Gev/s kernel_ms %.3f frac %.3f grid %d'%(d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['grid']))"
done; done; done
