"""TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.

`oracle/` holds CPU restatements of the reference's PNA hot path
(lukecavabarrett/pna, models/dgl/* and models/pytorch/pna/*) plus the harness
that runs the reference's own source in the build container to pin them.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import anything from here, and only as the checker / reported baseline.
`pna_amd/` never imports `oracle/`.
"""
