#!/bin/bash
# Development builds of the library with parts of k_posttrans_x3w compiled out (X3W_SKIP bitmask): where does the time go?
#   bash tools/x3w_variants.sh build "0 1"     (here)       bash tools/x3w_variants.sh run "0 1"     (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for v in $2; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DX3W_SKIP=$v -Iinclude -Ipna_amd/csrc pna_amd/csrc/*.hip -o pna_amd/lib/libpna_amd_v$v.so 2>/dev/null &
  done
  wait; ls pna_amd/lib/
else
  for v in $2; do
    PNA_AMD_LIB=pna_amd/lib/libpna_amd_v$v.so python tools/x3w_variant_time.py $v
  done
fi
