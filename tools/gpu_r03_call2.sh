#!/bin/bash
# round 3, GPU call 2: 5-buffer weight pipeline, 32-edge hub segments in the rest launch, reproducer with per-class counters, kernel trace
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=gpurun_out/r03c2; mkdir -p $O
echo "== pk_mfma_repro"; timeout 120 tools/ubench/pk_mfma_repro 4096 50 2>&1 | tee $O/pk_repro.log
echo "== pk experiment with ONE workgroup per CU (round-2 claim: exact)"; DF_WGS=1 DF_LIB=libdegree_fused_pk.so DF_DEBUG_AGG=1 timeout 400 python tools/df_check.py 2>&1 | grep -v "^$" | tail -11 | tee $O/df_pk_wgs1.log
echo "== one-kernel layer tests"
timeout 900 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 600 > $O/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_fused.log
echo "== fd_time (experiments lib: phase timers)"; PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 300 python tools/fd_time.py $O/fd_time_exp.json 2>&1 | tail -8
echo "== fd_time (production lib)"; timeout 300 python tools/fd_time.py $O/fd_time.json 2>&1 | tail -3
echo "== kernel trace of the bench"
cd /tmp; rm -rf $P/$O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe > $P/$O/trace.log 2>&1; echo rc=$?
cd $P
python - <<'PY'
import csv,glob
for p in glob.glob('gpurun_out/r03c2/trace/**/bench_kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:14]:
        print(r['Name'][:90].ljust(90), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
