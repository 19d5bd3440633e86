export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $P/gpurun_out/x3w_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/x3w_trace -o k -- python $P/tools/prof_x3w.py 30 > $P/gpurun_out/x3w_trace.log 2>&1; echo rc=$?
python - <<PY
import csv, os
P = os.environ["GRAFT_REPO_ROOT"]
for r in csv.DictReader(open(P + "/gpurun_out/x3w_trace/k_kernel_stats.csv")):
    if "posttrans" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
