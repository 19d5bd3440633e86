#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
timeout 400 python -m pytest tests/test_gpu_fused_degree.py tests/test_gpu_determinism.py tests/test_gpu_degree_groups.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_two_ranks.py -m gpu -q --timeout 200 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -6
for r in 1 2 3; do
  for lib in libpna_amd_nols libpna_amd; do
    PNA_AMD_LIB=pna_amd/lib/$lib.so timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c3 $lib: /"
  done
done
export FD_V=2000000 FD_E=20000000 FD_F=128
for r in 1 2; do
  for lib in libpna_amd_nols libpna_amd; do
    PNA_AMD_LIB=pna_amd/lib/$lib.so timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c5 $lib: /"
  done
done
