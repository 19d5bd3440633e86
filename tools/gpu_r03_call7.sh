#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c7; mkdir -p $O
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 600 python tools/fd_time.py $O/fd_time_exp.json 2>&1 | grep -v amdgpu.ids | grep "rep 0\|phase\|ablation\|workgroups"
