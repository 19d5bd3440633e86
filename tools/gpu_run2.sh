export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
rocprofv3 -L > gpurun_out/counters.txt 2>&1
cd /tmp
P=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/prof_trace -o bench -- python $P/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $P/gpurun_out/prof_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/gpurun_out/pmc_fetch -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_fetch.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/gpurun_out/pmc_write -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_write.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gpurun_out/pmc_sq -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_sq.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $P/gpurun_out/pmc_inst -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_inst.log 2>&1; echo "pmc4 rc=$?"
cd $P
find gpurun_out -name "*.csv" | head -40
du -sh gpurun_out
