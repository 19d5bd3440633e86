#!/bin/bash
# build: tools/ubench/project_variants.sh build     run (on the GPU box): tools/ubench/project_variants.sh run
cd "$(dirname "$0")/../.."
V=("b512:-DPNA_PROJECT_BLOCK=512")     # (other block sizes: before pna_project_grouped_f32 tied the file to 8-wavefront workgroups -- profiles/r06_project_variants.txt)
mkdir -p pna_amd/lib/pv
for v in "${V[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  if [ "$1" = build ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipna_amd/csrc $flags tools/ubench/project_variants.cpp pna_amd/csrc/pna_project.hip pna_amd/csrc/pna_common.hip -o pna_amd/lib/pv/$name &
  else
    pna_amd/lib/pv/$name
  fi
done
wait
