#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 600 > $O/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fused.log
echo "== fd_time (production lib)"; timeout 300 python tools/fd_time.py $O/fd_time.json 2>&1 | grep "^rep"
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 600 python tools/fd_time.py $O/fd_time_exp.json 2>&1 | grep -v amdgpu.ids | grep "rep 0\|phase\|ablation\|workgroups"
