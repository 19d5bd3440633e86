#!/usr/bin/env python
"""Launches of KNOWN HBM volume for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE units on gfx950 (the guide:
"WRITE_SIZE [is] uncalibrated: calibrate on a known byte count in your own access pattern").  Run under `rocprofv3 --pmc WRITE_SIZE`
and `--pmc FETCH_SIZE` (tools/gpu_r04_final.sh), then the summariser divides the known bytes by the counter:

  fill    2 GiB of floats written by torch's fill kernel (16-byte stores, whole lines)            -> bytes per WRITE_SIZE unit, streaming
  rows    the one-kernel layer's OUTPUT pattern: 1 M rows of 75 floats at pitch 80 (300 of every 320 bytes: partial last line),
          written by a torch strided copy                                                         -> bytes per WRITE_SIZE unit, our rows
  copy    1 GiB read + 1 GiB written by a contiguous copy                                         -> both units at once
All buffers are far beyond the 256 MiB Infinity Cache, so nothing is absorbed on-die."""
import torch

dev = torch.device("cuda:0")
n = 3
a = torch.empty(512 * 1024 * 1024, device=dev)            # 2 GiB
b = torch.empty(256 * 1024 * 1024, device=dev)            # 1 GiB
c = torch.randn(256 * 1024 * 1024, device=dev)            # 1 GiB
rows = torch.empty(4_000_000, 80, device=dev)             # 1.28 GB, 300-byte rows written at pitch 320
src = torch.randn(4_000_000, 75, device=dev)
torch.cuda.synchronize()
for _ in range(n):
    a.fill_(1.0)
    torch.cuda.synchronize()
    b.copy_(c)
    torch.cuda.synchronize()
    rows[:, :75].copy_(src)
    torch.cuda.synchronize()
print("ok")
