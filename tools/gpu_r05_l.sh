#!/bin/bash
export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT
cd $P
for r in 1 2; do
for al in 4 16 32; do
  PNA_AMD_OUT_PITCH_ALIGN=$al timeout 150 python tools/fd_diag.py 2>&1 | grep "group rows" | sed "s/^/c3 y pitch align=$al: /"
done
done
