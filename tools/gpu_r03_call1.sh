#!/bin/bash
# round 3, GPU call 1: packed-fp32 reproducer + flag bisect of the round-2 experiment, the new one-kernel layer's tests, timings, bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c1
O=gpurun_out/r03c1
echo "== pk_mfma_repro"; timeout 120 tools/ubench/pk_mfma_repro 4096 50 2>&1 | tee $O/pk_repro.log
echo "== new one-kernel layer tests"
timeout 900 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 600 > $O/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_fused.log
echo "== fd_time (production lib)"; timeout 300 python tools/fd_time.py $O/fd_time.json 2>&1 | tail -8
echo "== fd_time (experiments lib: phase timers)"; PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 300 python tools/fd_time.py $O/fd_time_exp.json 2>&1 | tail -8
echo "== bisect of the round-2 experiment (tools/ubench/degree_fused.hip)"
for v in libdegree_fused.so libdegree_fused_pk.so libdegree_fused_pk_fz.so libdegree_fused_nopk.so; do
  echo "-- $v"; DF_LIB=$v DF_DEBUG_AGG=1 timeout 400 python tools/df_check.py 2>&1 | grep -v "^$" | tail -12 | tee -a $O/df_bisect_$v.log
done
echo "== bench"; timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r03c1/bench.log').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step','ms_per_step_cold','kernel_ms','parity_check')})
print('roofline', {k:r['roofline'][k] for k in ('frac','read_only_frac','ms_per_launch','achieved')}, 'layer', r['roofline_layer']['frac'])
PY
tail -3 $O/bench.err
