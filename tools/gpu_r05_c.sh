#!/bin/bash
# Round 5, GPU call C: the whole GPU suite (all failures listed), and the one-kernel layer's finer phase timers (copy wait | first barrier of a
# pass | other barriers inside the multiply phase) at BASELINE configs[4]'s per-GPU shape (both wide candidates) and at C3.
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
O=$P/gpurun_out/r05c
rm -rf $O; mkdir -p $O
cd $P
timeout 500 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -12
export FD_V=2000000 FD_E=20000000 FD_F=128
for lib in libpna_amd_exp libpna_amd_w8exp; do
  PNA_AMD_LIB=pna_amd/lib/$lib.so timeout 200 python tools/fd_diag.py $O/c5_$lib.json 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib: /"
done
unset FD_V FD_E FD_F
PNA_AMD_LIB=pna_amd/lib/libpna_amd_exp.so timeout 200 python tools/fd_diag.py $O/c3_exp.json 2>&1 | grep -v amdgpu.ids | sed "s/^/c3 exp: /"
