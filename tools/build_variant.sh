#!/bin/bash
# A/B builds: libpna_amd_<name>.so = the current sources with pna_fused_degree.hip taken from git revision <rev> (boxes of the pool differ
# by 4-5 %: variants are only comparable inside ONE gpurun call; tools/fd_time.py takes the library through PNA_AMD_LIB).
#   tools/build_variant.sh <rev> <name>
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
cp pna_amd/csrc/*.hip pna_amd/csrc/*.h $tmp/
git show $1:pna_amd/csrc/pna_fused_degree.hip > $tmp/pna_fused_degree.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -I$tmp $tmp/*.hip -o pna_amd/lib/libpna_amd_$2.so
rm -rf $tmp
echo pna_amd/lib/libpna_amd_$2.so
