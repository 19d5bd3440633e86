#!/bin/bash
# Builds pna_amd/lib/libpna_amd_exp.so: the same sources with -DPNA_AMD_EXPERIMENTS, which enables the bench-experiment
# knobs of tune.reserved[0] (bit0: skip output stores; bits 8..: KB of dummy dynamic LDS per block = occupancy cap).
# The shipped libpna_amd.so has none of them.  tools/sweep.py loads this file when PNA_AMD_LIB points at it.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DPNA_AMD_EXPERIMENTS \
  -Iinclude -Ipna_amd/csrc pna_amd/csrc/*.hip -o pna_amd/lib/libpna_amd_exp.so
echo pna_amd/lib/libpna_amd_exp.so
