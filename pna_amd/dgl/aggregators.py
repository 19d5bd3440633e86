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

EPS = 1e-5  # models/dgl/aggregators.py:3


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


def aggregate_moment(h, n=3):
    """models/dgl/aggregators.py:29-36, QUIRK INCLUDED: the reference centres per node (`torch.mean(h, dim=1, keepdim=True)`, :32) but
    then takes `torch.mean(torch.pow(h - h_mean, n))` WITHOUT a dim (:33) -- the n-th central moment averaged over the whole
    mailbox tensor -- and returns the signed n-th root of that as a 0-dim tensor.  (So it cannot be concatenated in reduce_func,
    pna_layer.py:48, and no shipped config names it; it is an entry the reference's registry exports, reproduced as it behaves.)
    The per-node mean runs on the segment-reduce kernel; the whole-tensor power mean is two device-side torch reductions."""
    if h.dim() != 3:
        raise ValueError("mailbox must be (n, d, F)")
    h_mean = _mailbox_reduce(h, "mean").unsqueeze(1)
    h_n = torch.mean(torch.pow(h - h_mean, n))
    return torch.sign(h_n) * torch.pow(torch.abs(h_n) + EPS, 1. / n)


def aggregate_moment_3(h):
    return aggregate_moment(h, n=3)


def aggregate_moment_4(h):
    return aggregate_moment(h, n=4)


def aggregate_moment_5(h):
    return aggregate_moment(h, n=5)


AGGREGATORS = {"mean": aggregate_mean, "sum": aggregate_sum, "max": aggregate_max, "min": aggregate_min,
               "std": aggregate_std, "var": aggregate_var, "moment3": aggregate_moment_3, "moment4": aggregate_moment_4,
               "moment5": aggregate_moment_5}
