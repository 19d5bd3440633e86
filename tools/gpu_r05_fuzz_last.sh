#!/bin/bash
# the fuzzers on the round's last code, fresh seeds
mkdir -p gpurun_out/r05last
timeout 150 python tools/fuzz_fused.py 75 71 > gpurun_out/r05last/fuzz_fused.log 2>&1; echo "fuzz fused rc=$?"; tail -1 gpurun_out/r05last/fuzz_fused.log
timeout 120 python tools/fuzz_fused.py 45 73 big > gpurun_out/r05last/fuzz_fused_big.log 2>&1; echo "fuzz big rc=$?"; tail -1 gpurun_out/r05last/fuzz_fused_big.log
