#!/bin/bash
# PMC passes over tools/prof_dg.py (each its own run, no tracing domains) -> gpurun_out/dgpmc_summary.txt
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/dgpmc_*
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $P/gpurun_out/dgpmc_$i -o k -- python $P/tools/prof_dg.py 2 > $P/gpurun_out/dgpmc_$i.log 2>&1; echo "pmc$i rc=$?"
done
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/dgpmc_summary.txt
import csv, glob, collections, os
P = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(list)
for p in glob.glob(P + "/gpurun_out/dgpmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        # (rocprofv3 writes demangled names: "... k_posttrans_x3<1, false, 80, ...>")
        if "k_posttrans_x3<1," in k or "k_posttrans_x3ILi1E" in k: name = "grouped(S=1)"
        elif "k_posttrans_x3<3," in k or "k_posttrans_x3ILi3E" in k: name = "rest(S=3)"
        elif "k_segreduce_fast" in k: name = "gather"
        else: continue
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({c for (_, c) in agg})
kern = ["gather", "grouped(S=1)", "rest(S=3)"]
print(f"{'counter (mean per launch)':28s} " + " ".join(f"{k:>14s}" for k in kern))
for k in kern:
    assert any(kk == k for (kk, _) in agg), f"no counter rows matched kernel {k}: check the name filter"
for c in names:
    print(f"{c:28s} " + " ".join(f"{(sum(agg[(k, c)]) / len(agg[(k, c)])):14.5g}" if agg[(k, c)] else f"{'n/a':>14s}" for k in kern))
PY
