"""Backward passes of the fused operators.

Not implemented in this round: the forward kernels are the scoped hot path; training through them
raises instead of silently falling back to an eager implementation.
"""
import torch


class AggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *args):
        raise NotImplementedError("pna_amd: backward of the fused aggregation is not implemented yet; "
                                  "run under torch.no_grad() (forward / inference only)")


class PosttransFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *args):
        raise NotImplementedError("pna_amd: backward of the fused posttrans contraction is not implemented yet; "
                                  "run under torch.no_grad() (forward / inference only)")
