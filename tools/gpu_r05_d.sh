#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05d
rm -rf $O; mkdir -p $O
cd $P
timeout 200 python -m pytest tests/test_gpu_fused_degree.py -m gpu -q --timeout 100 -k "which_path" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-600
export FD_V=2000000 FD_E=20000000 FD_F=128
for lib in libpna_amd_exp libpna_amd_w8exp; do
  PNA_AMD_LIB=pna_amd/lib/$lib.so timeout 200 python tools/fd_diag.py $O/c5_$lib.json 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib: /"
done
unset FD_V FD_E FD_F
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 200 python tools/fd_diag.py $O/c3_exp.json 2>&1 | grep -v amdgpu.ids | sed "s/^/c3 exp: /"
