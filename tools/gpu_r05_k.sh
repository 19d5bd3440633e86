#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
for sp in 32 48 64 96; do
  PNA_AMD_FUSED_SPARE_WGS=$sp timeout 200 python bench.py --workload c5 --no-cpu-baseline --no-cold --no-power-probe --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 spare $sp: step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['ms_per_launch'],4), 'rest alone', round(d['kernel_ms']['fused_degree_rest_rows'],4))"
done
