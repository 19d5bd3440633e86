#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
timeout 400 python -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -6
timeout 300 python bench.py --no-cpu-baseline --no-c5-leg --no-power-probe 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', d['ms_per_step'], 'kernel', d['roofline']['ms_per_launch'], 'frac', d['roofline']['frac'], 'hipgraph', d['ms_per_step_hipgraph_replay'], 'contig', d['ms_per_step_contiguous_input'], 'setup', d['per_graph_setup']['degree_plan_build_ms'], d['per_graph_setup']['weight_image_pack_ms'])"
