#!/usr/bin/env python
"""Median per (pass label, counter, grid size) of the k_fused_degree launches in rocprofv3 --pmc output directories.

    python tools/pmc_sum.py <dir with one sub-directory per pass, named <label>_<n>> [kernel-name substring]
"""
import collections
import csv
import glob
import os
import re
import sys

O = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_fused_degree"
agg = collections.defaultdict(list)
for p in glob.glob(os.path.join(O, "*", "**", "*counter_collection.csv"), recursive=True):
    label = os.path.relpath(p, O).split(os.sep)[0].rsplit("_", 1)[0]
    for r in csv.DictReader(open(p)):
        if sub in r["Kernel_Name"] and int(r["Grid_Size"]) > 60000:
            # (round 6: a guarded call is two launches of the same grid -- k_fused_degree<.., ARITH = 0, ..> and its consuming
            # launch <.., ARITH = 1, ..> --: kept apart by the template arguments)
            m = re.search(r"<([^>]*)>", r["Kernel_Name"])
            tag = label + (" <" + m.group(1).replace(" ", "") + ">" if m else "")
            agg[(tag, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = sorted(agg[k])
    print(f"{k[0]:44s} {k[1]:36s} grid {k[2]:8s} median {v[len(v) // 2]:16.1f}  n={len(v)}")
