#!/bin/bash
# Kernel timeline of one PNASimpleLayer step at C3 (rocprofv3 --kernel-trace over tools/prof_dg.py): start offset and duration of every
# launch of the LAST step -> gpurun_out/r04_step_timeline.txt
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $P/tools/prof_dg.py 6 > /tmp/tl.log 2>&1; echo "rc=$?"
python - <<'PY' | tee $P/gpurun_out/r04_step_timeline.txt
import csv, glob
rows = []
for p in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "k_fused_degree" in r["Kernel_Name"] or "k_fused_roles" in r["Kernel_Name"]]
print(len(rows), "launches;", len(fused), "one-kernel launches")
for which in fused[-2:]:
    # the step around this launch: from the previous one-kernel launch's end to the next's start
    t0 = int(rows[which]["Start_Timestamp"])
    print("---- step around launch", which)
    for r in rows[max(0, which - 3):which + 8]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
PY
