#!/bin/bash
export TMPDIR=/tmp
echo "== pk sanity"; timeout 60 tools/ubench/pk_sanity
for f in 75 64 40 20; do echo "== fd_debug F=$f"; FD_F=$f timeout 300 python tools/fd_debug.py 2>&1 | grep -v amdgpu.ids | tail -4; done
echo "== fd_debug F=75 with scheduling fences"; PNA_AMD_LIB=pna_amd/lib/libpna_amd_fence.so timeout 300 python tools/fd_debug.py 2>&1 | grep -v amdgpu.ids | tail -4
