#!/bin/bash
# Round-6 measurement set (one GPU call): GPU suite, smoke, the bench line (with its configs[4]-shape child leg), its rocprofv3 kernel trace,
# the C3 / C5-shape kernels' FETCH / WRITE / L2 counters (guarded fp16 x 2: both launches of a call), the other BASELINE configs incl. the
# 5-tower layers, the training step, the guard probes, the fuzzers.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r06_final.sh'
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd $P
timeout 700 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_n1.json | head -4
timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-cold > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout 200 python bench.py --layers 4 --steps 10 --warmup 3 > $O/bench_layers4.json 2> $O/bench_layers4.err; echo "layers rc=$?"
cd /tmp; rm -rf $O/trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold --no-power-probe --no-c5-leg > $O/trace.log 2>&1; echo "trace rc=$?"
find $O/trace -name "bench_kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -5 $O/bench_kernel_stats.csv | cut -c1-220
rm -rf $O/trace
export FD_V=2000000 FD_E=20000000 FD_F=128
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc/c5_$i -o k -- python $P/tools/fd_diag.py --pmc > $O/pmc_c5_$i.log 2>&1; echo "c5 pmc pass $i rc=$?"
done
unset FD_V FD_E FD_F
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc3/c3_$i -o k -- python $P/tools/fd_diag.py --pmc > $O/pmc_c3_$i.log 2>&1; echo "c3 pmc pass $i rc=$?"
done
cd $P
python tools/pmc_sum.py $O/pmc | tee $O/pmc_c5.txt
python tools/pmc_sum.py $O/pmc3 | tee $O/pmc_c3.txt
rm -rf $O/pmc3 $O/pmc 2>/dev/null
timeout 500 python tools/bench_train.py > $O/train_step.json 2> $O/train_step.err; echo "train rc=$?"
timeout 700 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
tail -3 $O/configs.err
timeout 200 python tools/arith_ab.py 3 > $O/arith_ab.log 2>&1; tail -1 $O/arith_ab.log
timeout 200 python tools/guard_probe.py > $O/guard_probe.log 2>&1; tail -3 $O/guard_probe.log
timeout 200 python tools/guard_probe_tower.py > $O/guard_probe_tower.log 2>&1; tail -4 $O/guard_probe_tower.log
timeout 200 python tools/graph_replay_ab.py > $O/graph_replay_ab.log 2>&1; tail -1 $O/graph_replay_ab.log
timeout 200 python tools/plan_build_time.py > $O/plan_build_time.log 2>&1; tail -3 $O/plan_build_time.log
timeout 200 python tools/multi_tower_time.py > $O/multi_tower_time.log 2>&1; tail -1 $O/multi_tower_time.log
timeout 200 python tools/fuzz_fused.py 90 61 > $O/fuzz_fused.log 2>&1; echo "fuzz fused rc=$?"; tail -3 $O/fuzz_fused.log
timeout 200 python tools/fuzz_fused.py 60 63 big > $O/fuzz_fused_big.log 2>&1; echo "fuzz fused big rc=$?"; tail -3 $O/fuzz_fused_big.log
timeout 200 python tools/fuzz_train.py 60 11 > $O/fuzz_train.log 2>&1; echo "fuzz train rc=$?"; tail -3 $O/fuzz_train.log
