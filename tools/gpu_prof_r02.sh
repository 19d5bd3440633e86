#!/bin/bash
# rocprofv3 passes for profiles/r02_*: kernel trace of bench.py + PMC passes over tools/prof_kernels.py (each its own run)
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
mkdir -p $P/gpurun_out; rm -rf $P/gpurun_out/r02pmc_* $P/gpurun_out/r02_trace
cd $P && timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.log 2>gpurun_out/r02_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r02_bench.log | cut -c1-200
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gpurun_out/r02_trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold > $P/gpurun_out/r02_trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/gpurun_out/r02pmc_fetch -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/r02pmc_fetch.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/gpurun_out/r02pmc_write -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/r02pmc_write.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/gpurun_out/r02pmc_sq -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/r02pmc_sq.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $P/gpurun_out/r02pmc_inst -o k -- python $P/tools/prof_kernels.py 3 > $P/gpurun_out/r02pmc_inst.log 2>&1; echo "pmc4 rc=$?"
timeout 300 python $P/tools/x3_timers.py > $P/gpurun_out/r02_x3_timers.txt 2>&1; echo "timers rc=$?"
timeout 600 python $P/tools/bench_configs.py > $P/gpurun_out/r02_configs.json 2>$P/gpurun_out/r02_configs.err; echo "configs rc=$?"
timeout 600 python $P/bench.py --workload c5 --no-cpu-baseline > $P/gpurun_out/r02_bench_c5.log 2>&1; echo "c5 rc=$?"
