#!/bin/bash
# Round 5, GPU call B: the wide instantiation's two candidates (default library: 4 wavefronts x 2 workgroups per CU with SIX weight buffers;
# _w8: ONE 8-wavefront workgroup per CU with ten) at BASELINE configs[4]'s per-GPU shape, parity-checked; the GPU suite; the tile-order
# experiment at C3 with its L2 counters.
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05b
rm -rf $O; mkdir -p $O
cd $P
timeout 400 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest (default lib) rc=$?"; tail -4 $O/pytest_gpu.log
PNA_AMD_LIB_PATH=$P/pna_amd/lib/libpna_amd_w8.so timeout 300 python -m pytest tests/test_gpu_fused_degree.py tests/test_gpu_degree_groups.py tests/test_gpu_sharded_two_ranks.py -m gpu -x -q --timeout 200 > $O/pytest_w8.log 2>&1; echo "pytest (w8 lib) rc=$?"; tail -4 $O/pytest_w8.log
export FD_V=2000000 FD_E=20000000 FD_F=128 FD_PARITY=1
for r in 1 2; do
  for lib in libpna_amd libpna_amd_w8; do
    PNA_AMD_LIB=pna_amd/lib/$lib.so timeout 150 python tools/fd_diag.py $O/c5_${lib}_$r.json 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib $r: /"
  done
done
PNA_AMD_LIB=pna_amd/lib/libpna_amd_w8exp.so timeout 200 python tools/fd_diag.py $O/c5_w8exp.json 2>&1 | grep -v amdgpu.ids | sed "s/^/w8exp: /"
unset FD_V FD_E FD_F FD_PARITY
for r in 1 2; do
  for ord in ascending idmajor band4; do
    ORDER=$ord timeout 100 python tools/tile_order_exp.py 2>&1 | grep "RESULT\|Error\|error" | sed "s/^/$r /" | tee -a $O/tile_order_time.log
  done
done
cd /tmp
for ord in ascending idmajor; do
  ORDER=$ord timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc/c3${ord}_1 -o k -- python $P/tools/tile_order_exp.py --pmc > $O/pmc_c3_$ord.log 2>&1; echo "c3 $ord pmc rc=$?"
done
cd $P
python tools/pmc_sum.py $O/pmc | tee $O/pmc_summary.txt
rm -rf $O/pmc/*/*/*.db 2>/dev/null
