#!/bin/bash
# merged inline-asm statements (no s_nop between dependent statements): same bits as the previous build? faster?  C3, C5 shape, same box, 2 rounds
mkdir -p gpurun_out/r05h2
for rep in 1 2; do
for lib in _prev ""; do
  echo "== lib$lib rep $rep"
  L=""; [ -n "$lib" ] && L="pna_amd/lib/libpna_amd$lib.so"
  PNA_AMD_LIB=$L FD_PARITY=1 timeout 200 python tools/fd_diag.py 2>&1 | grep -E "checksum|group rows"
  PNA_AMD_LIB=$L FD_PARITY=1 FD_V=2000000 FD_E=20000000 FD_F=128 timeout 300 python tools/fd_diag.py 2>&1 | grep -E "checksum|group rows"
done; done 2>&1 | tee gpurun_out/r05h2/nop_ab.log
timeout 400 python -m pytest tests/test_gpu_fused_degree.py -x -q --timeout 300 2>&1 | tail -3 | tee gpurun_out/r05h2/pytest_fused_nop.log
