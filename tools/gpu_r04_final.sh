#!/bin/bash
# Round-4 measurement set: bench line, its rocprofv3 kernel trace, C5 shard shape, stacked layers, training step, the other BASELINE
# configs, tower layer, PMC traffic passes over the one-kernel layer + counter-unit calibration on launches of known volume.
#   gpurun --timeout 3000 -- 'bash tools/gpu_r04_final.sh'
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r04
mkdir -p $O
cd $P
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout 300 python bench.py --layers 4 --steps 10 --warmup 3 > $O/bench_layers4.json 2> $O/bench_layers4.err; echo "layers rc=$?"
timeout 600 python tools/bench_train.py > $O/train_step.json 2> $O/train_step.err; echo "train rc=$?"
timeout 900 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
timeout 300 python tools/tower_time.py $O/tower_time.json > $O/tower_time.log 2>&1; echo "tower rc=$?"
timeout 300 python tools/dw_time.py > $O/dw_time.json 2> $O/dw_time.err; echo "dw rc=$?"
timeout 400 python tools/block_pipeline_time.py > $O/block_pipeline.log 2>&1; echo "blocks rc=$?"; cp $P/gpurun_out/r04_block_pipeline_time.json $O/ 2>/dev/null
cd /tmp; rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe > $O/trace.log 2>&1; echo "trace rc=$?"
find $O/trace -name "bench_kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -6 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf $O/trace $O/pmc_*
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_dg_$i -o k -- python $P/tools/prof_dg.py 3 > $O/pmc_dg_$i.log 2>&1; echo "pmc dg $set rc=$?"
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_cal_$i -o k -- python $P/tools/pmc_calib.py > $O/pmc_cal_$i.log 2>&1; echo "pmc cal $set rc=$?"
done
cd $P
python - <<'PY' | tee $O/pmc_traffic.txt
import csv, glob, collections, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04")
def table(prefix):
    agg = collections.defaultdict(list)
    for p in glob.glob(os.path.join(O, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
cal = table("pmc_cal_")
print("calibration launches (mean counter per launch, launches):")
for (k, c), v in sorted(cal.items()):
    if sum(v) / len(v) > 1000: print(f"  {k:72s} {c:12s} {sum(v)/len(v):14.1f} x{len(v)}")
def pick(agg, sub, ctr, big):
    best = None
    for (k, c), v in agg.items():
        if c == ctr and sub in k:
            m = sorted(v)[len(v) // 2]
            if m > big and (best is None or m > best): best = m
    return best
GiB = 1024 ** 3
out = {}
w_fill = pick(cal, "Fill", "WRITE_SIZE", 1e5)
if w_fill: out["bytes_per_WRITE_SIZE_unit_streaming_fill"] = 2 * GiB / w_fill
# the strided row copy and the contiguous copy share torch's copy kernel names: tell them apart by FETCH volume
copies_w = sorted({sorted(v)[len(v)//2] for (k, c), v in cal.items() if c == "WRITE_SIZE" and ("copy" in k.lower() or "elementwise" in k.lower()) and sorted(v)[len(v)//2] > 1e5})
copies_f = sorted({sorted(v)[len(v)//2] for (k, c), v in cal.items() if c == "FETCH_SIZE" and ("copy" in k.lower() or "elementwise" in k.lower()) and sorted(v)[len(v)//2] > 1e5})
out["copy_kernels_WRITE_SIZE"], out["copy_kernels_FETCH_SIZE"] = copies_w, copies_f
dg = table("pmc_dg_")
f = pick(dg, "k_fused_degree", "FETCH_SIZE", 1e4); w = pick(dg, "k_fused_degree", "WRITE_SIZE", 1e4)
out["k_fused_degree"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
print(json.dumps(out, indent=1))
PY
tail -c 300 $O/configs.err
