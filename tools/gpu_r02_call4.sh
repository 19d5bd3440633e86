#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_posttrans_x3.py -m gpu -q --timeout 600 > gpurun_out/pytest_x3.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/pytest_x3.log | tail -8
timeout 900 python tools/exp_r02.py posttrans kscan > gpurun_out/exp4.log 2>&1; echo "exp4 rc=$?"; grep -E "^(posttrans pipeline|kscan)" gpurun_out/exp4.log | cut -c1-400
cd /tmp
rm -rf $P/gpurun_out/pmc2_*
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gpurun_out/pmc2_sq -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc2_sq.log 2>&1; echo "pmc sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $P/gpurun_out/pmc2_inst -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc2_inst.log 2>&1; echo "pmc inst rc=$?"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --output-format csv -d $P/gpurun_out/pmc2_lds -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/pmc2_lds.log 2>&1; echo "pmc lds rc=$?"
ls $P/gpurun_out/pmc2_sq | head
