#!/bin/bash
# FETCH_SIZE / L2 hit counters of pna_fused_degree_f32 builds on ONE box (tools/build_variant.sh): which change moved the fabric reads.
#   FD_LIBS=pna_amd/lib/libpna_amd_r3.so,pna_amd/lib/libpna_amd.so gpurun -- 'bash tools/gpu_pmc_ab.sh'
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/pmc_ab
rm -rf $O; mkdir -p $O
cd /tmp
for lib in ${FD_LIBS//,/ }; do
  n=$(basename $lib .so)
  i=0
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE"; do
    i=$((i+1))
    PNA_AMD_LIB=$lib timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/${n}_$i -o k -- python $P/tools/fd_ab.py --child > $O/${n}_$i.log 2>&1; echo "$n pass $i rc=$?"
  done
done
cd $P
python - <<'PY' | tee $O/summary.txt
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_ab")
agg = collections.defaultdict(list)
for p in glob.glob(os.path.join(O, "*", "**", "*counter_collection.csv"), recursive=True):
    lib = os.path.relpath(p, O).split(os.sep)[0].rsplit("_", 1)[0]
    for r in csv.DictReader(open(p)):
        if "k_fused_degree" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            agg[(lib, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = sorted(agg[k]); print(f"{k[0]:24s} {k[1]:24s} grid {k[2]:8s} median {v[len(v)//2]:14.1f}  n={len(v)}")
PY
