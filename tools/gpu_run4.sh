export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -25
timeout 600 python tools/sweep.py --tag r01d --rounds 3 > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
head -30 gpurun_out/sweep.log
cd /tmp; P=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gpurun_out/pmc_sq -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_sq.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $P/gpurun_out/pmc_inst -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc_inst.log 2>&1; echo "pmc4 rc=$?"
