#!/bin/bash
# decomposition of the round's final kernel (fp16 x 2): phase timers and ablations, experiments build, C5's shape then C3
mkdir -p gpurun_out/r05dec2
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so FD_V=2000000 FD_E=20000000 FD_F=128 timeout 300 python tools/fd_diag.py gpurun_out/r05dec2/c5.json 2>&1 | grep -E "group rows|phase|ablation|workgroups" | cut -c1-200
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 200 python tools/fd_diag.py gpurun_out/r05dec2/c3.json 2>&1 | grep -E "group rows|phase|ablation|workgroups" | cut -c1-200
