"""Operator registry of the DGL variant: AGGREGATORS[name](h:(n, d, F)) -> (n, F).

Same names and signatures as the reference's models/dgl/aggregators.py:54-56, but each call runs the
HIP segment-reduce kernel: a mailbox (n, d, F) is a regular CSR (rowptr = [0, d, 2d, ...]) over
edge-resident messages, so `torch.mean(h, 1)` & co. become one pna_segreduce_fwd_f32 launch.
GPU tensors only.  The layers do not go through this dict per degree bucket (that is the Python
loop the fused kernel removes); it exists so that code written against the reference's registry
keeps working.
"""
import torch

from .. import ops


def _mailbox_reduce(h, name):
    if h.dim() != 3:
        raise ValueError("mailbox must be (n, d, F)")
    n, d, F = h.shape
    x = h.contiguous().view(n * d, F)
    rowptr = torch.arange(0, (n + 1) * d, d, dtype=torch.int32, device=h.device)
    return ops.segreduce(rowptr, None, x, F, [name])


def aggregate_mean(h):
    return _mailbox_reduce(h, "mean")


def aggregate_max(h):
    return _mailbox_reduce(h, "max")


def aggregate_min(h):
    return _mailbox_reduce(h, "min")


def aggregate_std(h):
    return _mailbox_reduce(h, "std")


def aggregate_var(h):
    return _mailbox_reduce(h, "var")


def aggregate_sum(h):
    return _mailbox_reduce(h, "sum")


# moment3/4/5 of the reference (aggregators.py:29-47) reduce over the WHOLE tensor by mistake
# (torch.mean without dim, :33) and are unused by every shipped config -- not provided.
AGGREGATORS = {"mean": aggregate_mean, "sum": aggregate_sum, "max": aggregate_max, "min": aggregate_min,
               "std": aggregate_std, "var": aggregate_var}
