#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05g
rm -rf $O; mkdir -p $O
cd $P
timeout 200 python -m pytest tests/test_gpu_fused_degree.py -m gpu -q --timeout 60 -x -k "balanced_tile_order" 2>&1 | grep -E "Error|passed|failed|Timeout" | cut -c1-400
for r in 1 2; do
  for ord in ascending lpt dynamic_plan_order dynamic; do
    ORDER=$ord timeout 100 python tools/tile_order_exp.py 2>&1 | grep "RESULT\|Error\|error" | sed "s/^/$r /" | tee -a $O/tile_order_time.log
  done
done
export FD_V=2000000 FD_E=20000000 FD_F=128
for bal in off lpt dynamic; do
  PNA_AMD_FUSED_BALANCE=$bal timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c5 balance=$bal: /"
done
unset FD_V FD_E FD_F
for bal in off lpt dynamic; do
  PNA_AMD_FUSED_BALANCE=$bal timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c3 balance=$bal: /"
done
PNA_AMD_FUSED_BALANCE=dynamic PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 150 python tools/fd_diag.py 2>&1 | grep "phase timers" | cut -c1-330 | sed "s/^/c3 dynamic exp: /"
