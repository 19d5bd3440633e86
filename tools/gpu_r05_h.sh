#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
for r in 1 2; do
for tail in 1 4 8; do
  PNA_AMD_FUSED_DYNAMIC_TAIL=$tail timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c3 dynamic tail=$tail: /"
done
done
export FD_V=2000000 FD_E=20000000 FD_F=128
for tail in 1 4 8; do
  PNA_AMD_FUSED_DYNAMIC_TAIL=$tail timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c5 dynamic tail=$tail: /"
done
unset FD_V FD_E FD_F
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 150 python tools/fd_diag.py 2>&1 | grep "phase timers" | cut -c1-330 | sed "s/^/c3 dynamic tail=4 exp: /"
