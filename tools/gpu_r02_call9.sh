#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python tools/tf_timers.py > gpurun_out/tf_timers.txt 2>&1; echo "timers rc=$?"; cat gpurun_out/tf_timers.txt | cut -c1-220
