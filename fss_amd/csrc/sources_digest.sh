#!/bin/sh
# The digest fss_amd/_build.py stamps libfss_amd.so with (_sources_digest: compile flags, then name and content of
# every unit and header), computed without Python so that `make` leaves a library Python does not rebuild.
# tests/test_abi.py checks that the two agree.
cd "$(dirname "$0")" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
FILES="fss_amd.hip fss_abi_evalall.hip fss_abi_gen.hip fss_abi_fused.hip fss_abi_fused_batch.hip fss_abi_fused_chacha.hip fss_abi_fused_nodes.hip \
fss_dev.hpp fss_point.hpp fss_quad.hpp fss_evalall.hpp fss_gen.hpp fss_vdpf.hpp fss_host.hpp fss_fused_launch.hpp \
../../include/fss_amd.h"
{
  printf '%s' "$FLAGS"
  for f in $FILES; do
    printf '%s' "$(basename "$f")"
    cat "$f"
  done
} | sha256sum | cut -d' ' -f1
