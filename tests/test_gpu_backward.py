"""GPU parity of the dense-layer gradients (s2d_amd/backward.py) against torch autograd of the same
layer in float64 on the CPU (the reference's `losses.backward()`, engine/train_loop.py:720, differentiates nn.Linear /
F.conv2d with exactly these formulas)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("M,N,K", [(1000, 256, 96), (5000, 100, 256), (40000, 256, 1024), (33, 7, 5), (70000, 1024, 256)])
def test_linear_backward_vs_autograd(M, N, K):
    from s2d_amd import backward
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g)
    w = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn((N,), generator=g)
    dy = torch.randn((M, N), generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    (torch.nn.functional.linear(xd, wd, bd) * dy.double()).sum().backward()
    dx, dw, db = backward.linear_backward(x.to(DEV), w.to(DEV), dy.to(DEV))
    # fp32-class: the forward's dense tolerance (split-fp16 x3, f32 accumulation over up to 70000 terms)
    assert rel(dx.cpu().numpy(), xd.grad.numpy()) < 2e-6
    assert rel(dw.cpu().numpy(), wd.grad.numpy()) < 5e-6
    assert rel(db.cpu().numpy(), bd.grad.numpy()) < 5e-6
    # reproducible: fixed-order reduction of the contraction slices
    dw2 = backward.weight_grad(dy.to(DEV), x.to(DEV))
    assert torch.equal(dw, dw2)
    # accumulation into an existing gradient (a weight used twice / gradient accumulation over iterations)
    acc = dw.clone()
    backward.weight_grad(dy.to(DEV), x.to(DEV), out=acc, beta=1.0)
    assert rel(acc.cpu().numpy(), 2 * wd.grad.numpy()) < 5e-6


def test_transpose_odd_shapes():
    from s2d_amd import backward
    for R, C, pad in ((1, 1, None), (65, 130, None), (1000, 37, 1024), (129, 64, 160)):
        x = torch.randn((R, C), device=DEV)
        t = backward.transpose(x, pad)
        assert torch.equal(t[:, :R], x.t())
        if pad:
            assert float(t[:, R:].abs().max()) == 0.0
