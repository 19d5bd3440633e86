#!/bin/bash
# PMC passes over tools/prof_dg.py (the C3 simple layer: pna_fused_degree_f32 over the group rows + the two-kernel rest path), each its own
# rocprofv3 run, --pmc only (no trace domains) -> gpurun_out/r03_fused_pmc.txt
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/fdpmc_*
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $P/gpurun_out/fdpmc_$i -o k -- python $P/tools/prof_dg.py 3 > $P/gpurun_out/fdpmc_$i.log 2>&1; echo "pmc$i rc=$?"
done
cd $P
python - <<'PY' | tee gpurun_out/r03_fused_pmc.txt
import csv, glob, collections, json
agg = collections.defaultdict(list)
for p in glob.glob("gpurun_out/fdpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "k_fused_degree" in k: name = "fused(groups)"
        elif "k_segreduce_fast" in k: name = "gather(rest)"
        elif "k_heavy_finalize" in k: name = "finalize(rest)"
        elif "k_posttrans_x3<3," in k or "k_posttrans_x3ILi3E" in k: name = "contraction(rest)"
        else: continue
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
names = sorted({c for (_, c) in agg})
kern = ["fused(groups)", "gather(rest)", "finalize(rest)", "contraction(rest)"]
print(f"{'counter (mean per launch)':28s} " + " ".join(f"{k:>18s}" for k in kern))
assert any(kk == "fused(groups)" for (kk, _) in agg), "no counter rows matched k_fused_degree: check the name filter"
for c in names:
    print(f"{c:28s} " + " ".join(f"{(sum(agg[(k, c)]) / len(agg[(k, c)])):18.6g}" if agg[(k, c)] else f"{'n/a':>18s}" for k in kern))
m = lambda k, c: sum(agg[(k, c)]) / len(agg[(k, c)]) if agg[(k, c)] else None
f, w = m("fused(groups)", "FETCH_SIZE"), m("fused(groups)", "WRITE_SIZE")
if f and w:
    unit = 2018.4212378437962      # bytes per FETCH_SIZE unit for 16-byte-per-lane reads (calibrated in round 2 on a launch of known volume: profiles/hbm_traffic.json)
    rd, wr = f * unit, w * 1024.0
    print(json.dumps({"pna_fused_degree_c3": {"fabric_read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                                              "FETCH_SIZE": f, "WRITE_SIZE": w, "bytes_per_FETCH_SIZE_unit": unit}}))
PY
