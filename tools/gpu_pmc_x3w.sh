#!/bin/bash
# PMC passes over tools/prof_x3w.py (each its own run, no tracing domains)
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/x3wpmc_*
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $P/gpurun_out/x3wpmc_$i -o k -- python $P/tools/prof_x3w.py 2 > $P/gpurun_out/x3wpmc_$i.log 2>&1; echo "pmc$i rc=$?"
done
python - <<'PY'
import csv, glob, collections, os
P = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for p in glob.glob(P + "/gpurun_out/x3wpmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_posttrans_x3w" in k: name = "x3w"
        elif "k_posttrans_x3" in k: name = "x3"
        else: continue
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({c for (_, c) in agg})
for c in names:
    print(f"{c:28s} x3={sum(agg[('x3', c)]) / max(1, len(agg[('x3', c)])):.4g}   x3w={sum(agg[('x3w', c)]) / max(1, len(agg[('x3w', c)])):.4g}")
PY
