"""Torch-tensor front ends of the C ABI (include/s2d_hip.h).  Tensors are plumbing only: device memory,
the current stream, and the caching allocator for outputs.  Every function launches hand-written HIP."""
import torch

from ._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or not t.is_contiguous() or t.dtype != dtype:
        raise RuntimeError(f"s2d op needs a contiguous {dtype} CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def gemm_nt(A, B, scale=None, bias=None, res=None, relu=False, out=None):
    """out[..., M, N] = act(A[..., M, K] @ B[(...), N, K]^T * scale + bias + res).
    A may be 2-D or batched 3-D; B 2-D (shared) or 3-D (per batch)."""
    for t in (A, B, scale, bias, res, out):
        _chk(t)
    batched = A.dim() == 3
    bs = A.shape[0] if batched else 1
    M, K = A.shape[-2:]
    N = B.shape[-2]
    assert B.shape[-1] == K
    if out is None:
        out = torch.empty((bs, M, N) if batched else (M, N), device=A.device, dtype=torch.float32)
    sA = M * K if batched else 0
    sB = N * K if B.dim() == 3 else 0
    sC = M * N
    lib().call("s2d_gemm_nt_f32", A, B, out, M, N, K, K, K, N, bs, sA, sB, sC, scale, bias, res, N,
               M * N if res is not None and res.dim() == 3 else 0, int(relu), _stream())
    return out


def conv2d_nhwc(x, w, stride=1, pad=0, scale=None, bias=None, res=None, relu=False):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] -> [N,Ho,Wo,Cout]."""
    for t in (x, w, scale, bias, res):
        _chk(t)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    lib().call("s2d_conv2d_nhwc_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, scale, bias, res, int(relu),
               _stream())
    return y
