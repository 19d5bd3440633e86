#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
for r in 1 2; do
for sp in 24 32 40; do
  PNA_AMD_FUSED_SPARE_WGS=$sp timeout 200 python bench.py --no-cpu-baseline --no-cold --no-power-probe --no-c5-leg --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c3 spare $sp: step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['ms_per_launch'],4), 'full grid', round(d['roofline']['full_grid']['ms_per_launch'],4))"
done
done
