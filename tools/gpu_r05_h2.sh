#!/bin/bash
# fp16 x 2 contraction of the one-kernel layer: parity suite of the kernel, then its time at C3 and C5's shape
mkdir -p gpurun_out/r05h2
timeout 600 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 300 2>&1 | tail -25 > gpurun_out/r05h2/pytest_fused.log
tail -5 gpurun_out/r05h2/pytest_fused.log
FD_PARITY=1 timeout 200 python tools/fd_diag.py gpurun_out/r05h2/c3.json 2>&1 | tail -4
FD_PARITY=1 FD_V=2000000 FD_E=20000000 FD_F=128 timeout 300 python tools/fd_diag.py gpurun_out/r05h2/c5.json 2>&1 | tail -4
