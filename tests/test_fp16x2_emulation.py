"""The arithmetic of the one-kernel layer's contraction (pna_amd/csrc/pna_x3_split.h), restated with torch on the CPU:
fp32 operands as TWO fp16 terms behind power-of-two row / column scales and THREE partial products, against bf16 x 3 (three terms, six
products: rounds 2-4, still the two-kernel path's) and a plain fp32 GEMM, all measured against float64.  The partial products are
accumulated in float64 here, so only the operand split shows -- the fp32 accumulation is common to all three on the GPU.  This pins
the accuracy CLASS the kernel's design rests on; the kernel itself is checked on the GPU (tests/test_gpu_fused_degree.py)."""
import pytest
import torch


def _split_bf16x3(x):
    def top16(t):
        return (t.view(torch.int32) & -65536).view(torch.float32)
    x0 = top16(x); r = x - x0; x1 = top16(r); x2 = top16(r - x1)
    return x0, x1, x2


def _split_h2(x):
    h0 = x.half().float()
    return h0, (x - h0).half().float()


def _scale_exp(bound):
    """2^s puts `bound` into [2^13, 2^14): s = 14 - frexp exponent (pna_x3_split.h::h2_scale_exp)."""
    return 14 - torch.frexp(bound)[1]


def _contract(A, W, dist):
    ref = A.double() @ W.double()
    floor = A.abs().double() @ W.abs().double()
    out = {"f32": ((A @ W).double() - ref).abs() / floor}
    a, w = _split_bf16x3(A), _split_bf16x3(W)
    y = sum(a[i].double() @ w[j].double() for i, j in [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)])
    out["bf16x3"] = (y - ref).abs() / floor
    # rows of A by the kernel's rule (bound = 2 x the row's largest magnitude), columns of W by the pack kernel's (the largest itself).
    # (The kernel floors the row's magnitude at 0.0032: a PNA row always holds std statistics >= sqrt(1e-5), which a generic matrix
    # like this one does not -- here the floor is only what keeps an all-zero row finite.)
    sA = _scale_exp(2.0 * A.abs().amax(1, keepdim=True).clamp(min=1e-30))
    sW = _scale_exp(W.abs().amax(0, keepdim=True).clamp(min=1e-30))
    As, Ws = torch.ldexp(A, sA), torch.ldexp(W, sW)
    assert As.abs().max() < 2.0 ** 14 and Ws.abs().max() < 2.0 ** 14 and torch.isfinite(As.half()).all() and torch.isfinite(Ws.half()).all()
    a, w = _split_h2(As), _split_h2(Ws)
    y = sum(a[i].double() @ w[j].double() for i, j in [(1, 0), (0, 1), (0, 0)])
    y = torch.ldexp(torch.ldexp(y, -sA.double().int()), -sW.double().int())
    out["fp16x2"] = (y - ref).abs() / floor
    return out


@pytest.mark.parametrize("M,K,N,dist", [(2048, 300, 75, "normal"), (2048, 512, 128, "normal"), (2048, 300, 75, "wide")])
def test_two_fp16_terms_and_three_products_are_in_the_accuracy_class_of_bf16x3(M, K, N, dist):
    torch.manual_seed(0)
    A, W = torch.randn(M, K), torch.randn(K, N) * 0.1
    if dist == "wide":                                           # six decades inside a row / column, thirty between rows
        A = A * torch.exp(torch.randn(M, K) * 3) * 10.0 ** torch.empty(M, 1).uniform_(-15, 15)
        W = W * torch.exp(torch.randn(K, N) * 2) * 10.0 ** (torch.arange(N) % 9 - 4).float()[None]
    err = _contract(A, W, dist)
    rms = {k: (v ** 2).mean().sqrt().item() for k, v in err.items()}
    mx = {k: v.max().item() for k, v in err.items()}
    # the operand split alone: both multi-term forms far below an fp32 GEMM's rounding, fp16 x 2 within 1.5x of bf16 x 3
    assert rms["fp16x2"] <= 1.5 * rms["bf16x3"] and mx["fp16x2"] <= 2.0 * mx["bf16x3"], (rms, mx)
    assert rms["fp16x2"] <= 0.5 * rms["f32"], (rms, mx)
    assert mx["fp16x2"] <= 5e-7, mx                              # (of sum_k |a_k| |w_k|: the bar the GPU tests hold the kernel to is 2e-6)


def test_scale_rule_keeps_extreme_rows_inside_fp16():
    """Rows near FLT_MAX, rows of denormal size, an all-zero row: the scaled operand stays finite in fp16 and the scale is a normal fp32."""
    A = torch.tensor([[3.0e38, -1.0e38, 1.0], [1e-40, -3e-41, 0.0], [0.0, 0.0, 0.0], [1.0, -2.0, 0.5]])
    std = torch.full((4, 1), 1e-5 ** 0.5)                          # every PNA row holds std statistics >= sqrt(eps): the kernel's floor
    A = torch.cat([A, std], dim=1)
    bound = (2.0 * A[:, :3].abs().amax(1, keepdim=True).clamp(min=0.0032)).clamp(max=torch.finfo(torch.float32).max)
    s = _scale_exp(bound)
    assert int(s.min()) >= -114 and int(s.max()) <= 21            # 2^s and 2^-s are normal fp32 numbers (the kernel keeps both)
    As = torch.ldexp(A, s)
    assert torch.isfinite(As.half()).all() and As.abs().max() < 2.0 ** 15
    assert (As[:, 3] >= 2.0 ** -14).all() or int(s.min()) < 0     # the std entries stay normal fp16 numbers unless the row is astronomically large
