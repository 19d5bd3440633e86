#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c6; mkdir -p $O
echo "== pk sanity"; timeout 60 tools/ubench/pk_sanity
for f in 75 64; do echo "== fd_debug F=$f"; FD_F=$f timeout 300 python tools/fd_debug.py 2>&1 | grep -v amdgpu.ids | tail -4; done
timeout 900 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 600 > $O/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fused.log
echo "== fd_time (production lib)"; timeout 300 python tools/fd_time.py $O/fd_time.json 2>&1 | grep "^rep"
