#!/bin/bash
# Round 5, GPU call A: (1) the random-gather ceiling of the memory system (tools/ubench/gather_ceiling), (2) phase timers + ablations +
# PMC counters of the C5 wide instantiation, (3) the tile-order experiment at C3 with its counters (item 2).
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05a
rm -rf $O; mkdir -p $O
cd $P
GC=tools/ubench/gather_ceiling
{
  for args in "1024 512 2 4 16 8" "1024 512 2 6 18 8" "1024 512 4 4 16 8" "1024 512 4 5 15 8" "1024 512 2 4 16 4" "1024 512 4 4 16 4" \
              "1024 512 2 4 16 16" "1024 512 1 8 16 16" "1024 512 2 4 16 8 256" "4096 512 2 4 16 8" "4096 512 4 4 16 8" "4096 512 4 4 16 4" \
              "306 320 3 4 16 8" "306 320 3 4 16 4" "128 320 3 4 16 8" "64 512 4 4 16 8" "16384 512 4 4 16 8" "16384 512 2 4 16 8"; do
    timeout 60 $GC $args
  done
} > $O/gather_ceiling.jsonl 2>&1
echo "ubench done"; tail -3 $O/gather_ceiling.jsonl
# C5 shape: shipped library (time), experiments library (timers, ablations)
export FD_V=2000000 FD_E=20000000 FD_F=128
timeout 300 python tools/fd_diag.py $O/c5_time.json > $O/c5_time.log 2>&1; echo "c5 time rc=$?"
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 400 python tools/fd_diag.py $O/c5_exp.json > $O/c5_exp.log 2>&1; echo "c5 exp rc=$?"
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc/c5_$i -o k -- python $P/tools/fd_diag.py --pmc > $O/pmc_c5_$i.log 2>&1; echo "c5 pmc pass $i rc=$?"
done
unset FD_V FD_E FD_F
cd $P
# C3: tile orders -- time (one process each, round robin twice), then the L2 counters
for r in 1 2; do
  for ord in ascending idmajor idmajor_rr band4 band12; do
    ORDER=$ord timeout 200 python tools/tile_order_exp.py 2>&1 | grep RESULT | sed "s/^/$r /" >> $O/tile_order_time.log
  done
done
cat $O/tile_order_time.log
cd /tmp
for ord in ascending idmajor band4; do
  ORDER=$ord timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc/c3${ord}_1 -o k -- python $P/tools/tile_order_exp.py --pmc > $O/pmc_c3_$ord.log 2>&1; echo "c3 $ord pmc rc=$?"
done
cd $P
python tools/pmc_sum.py $O/pmc | tee $O/pmc_summary.txt
rm -rf $O/pmc/*/*/*.db 2>/dev/null
du -sh $O
