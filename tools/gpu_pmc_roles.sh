#!/bin/bash
# PMC passes over tools/prof_dg.py (the C3 simple layer): the one-kernel layer with wavefront roles (PNA_AMD_ROLES=1, default) and the
# round-3 one-kernel layer (PNA_AMD_ROLES=0), each counter set its own rocprofv3 run, --pmc only (no trace domains)
# -> gpurun_out/r04_roles_pmc.txt
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/frpmc_*
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $P/gpurun_out/frpmc_available.txt
i=0
for roles in 1 0; do
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PNA_AMD_ROLES=$roles timeout 300 rocprofv3 --pmc $set --output-format csv -d $P/gpurun_out/frpmc_$i -o k -- python $P/tools/prof_dg.py 3 > $P/gpurun_out/frpmc_$i.log 2>&1; echo "pmc$i roles=$roles rc=$?"
done
done
cd $P
python - <<'PY' | tee gpurun_out/r04_roles_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for p in glob.glob("gpurun_out/frpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_fused_roles" in k: name = "roles(groups)"
        elif "k_fused_degree" in k: name = "round3(groups)"
        else: continue
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({c for (_, c) in agg})
kern = ["roles(groups)", "round3(groups)"]
print("# available instruction-cache counters:", open("gpurun_out/frpmc_available.txt").read())
print(f"{'counter (mean per launch)':28s} " + " ".join(f"{k:>18s}" for k in kern))
for c in names:
    print(f"{c:28s} " + " ".join(f"{(sum(agg[(k, c)]) / len(agg[(k, c)])):18.6g}" if agg[(k, c)] else f"{'n/a':>18s}" for k in kern))
PY
