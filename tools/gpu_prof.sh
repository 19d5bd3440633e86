# rocprofv3 passes for profiles/: kernel trace of bench.py + PMC passes over tools/prof_kernels.py
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/pmc_* $P/gpurun_out/prof_trace
cd $P && timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/prof_trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/gpurun_out/prof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/gpurun_out/pmc_fetch -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_fetch.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/gpurun_out/pmc_write -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_write.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gpurun_out/pmc_sq -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_sq.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $P/gpurun_out/pmc_inst -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_inst.log 2>&1; echo "pmc4 rc=$?"
tail -3 $P/gpurun_out/pmc_inst.log
