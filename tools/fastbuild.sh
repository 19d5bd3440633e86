#!/bin/bash
# Development build: one object per .hip, in parallel, re-compiled only when the source or a header is newer; links libpna_amd[<suffix>].so.
#   tools/fastbuild.sh [suffix] [extra hipcc flags...]      e.g.  tools/fastbuild.sh _exp -DPNA_AMD_EXPERIMENTS
# (__graft_entry__.build() / python -m pna_amd.build stay the one-command full build.)
set -e
cd "$(dirname "$0")/.."
SUF=$1; shift || true
OBJ=build/obj$SUF
mkdir -p $OBJ pna_amd/lib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ipna_amd/csrc $*"
newest_h=$(ls -t pna_amd/csrc/*.h include/*.h | head -1)
pids=()
for f in pna_amd/csrc/*.hip; do
  o=$OBJ/$(basename $f .hip).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ $newest_h -nt $o ]; then
    ( hipcc $FLAGS -c $f -o $o 2> $o.log || { echo "FAILED $f"; cat $o.log; exit 1; } ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# (only the objects of sources that still exist: an object left behind by a removed .hip would otherwise be linked in -- ADVICE r5)
objs=""
for f in pna_amd/csrc/*.hip; do objs="$objs $OBJ/$(basename $f .hip).o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o pna_amd/lib/libpna_amd$SUF.so.tmp
mv pna_amd/lib/libpna_amd$SUF.so.tmp pna_amd/lib/libpna_amd$SUF.so
echo pna_amd/lib/libpna_amd$SUF.so
