"""GPU parity of the dense-contraction kernels (GEMM / implicit-GEMM conv) against the CPU oracle."""
import numpy as np
import pytest
import torch

from s2d_amd.utils import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(params=["f32", "bf16x3", "f16x3"], autouse=True)
def dense_mode(request):
    from s2d_amd import ops
    ops.set_dense_mode(request.param)
    yield request.param
    ops.set_dense_mode("f16x3")


@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (200, 256, 256, 1), (1000, 100, 256, 2), (77, 288, 1024, 1),
                                          (4096, 2048, 512, 1), (5, 2, 256, 1), (40000, 256, 256, 1), (300, 256, 1024, 3)])
def test_gemm_nt(oracle, M, N, K, batch):
    from s2d_amd import ops
    A = synth.randn(1, 1, (batch, M, K))
    B = synth.randn(1, 2, (batch, N, K))
    sc = synth.randn(1, 3, (N,)) * 0.5 + 1
    bi = synth.randn(1, 4, (N,))
    res = synth.randn(1, 5, (batch, M, N))
    ref = np.maximum(np.einsum("bmk,bnk->bmn", A.astype(np.float64), B.astype(np.float64)) * sc + bi + res, 0)
    out = ops.gemm_nt(_dev(A), _dev(B), _dev(sc), _dev(bi), _dev(res), relu=True).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # no-epilogue path, shared B
    out2 = ops.gemm_nt(_dev(A), _dev(B[0])).cpu().numpy()
    ref2 = np.einsum("bmk,nk->bmn", A.astype(np.float64), B[0].astype(np.float64))
    np.testing.assert_allclose(out2, ref2, rtol=1e-4, atol=1e-4 * np.abs(ref2).max())


@pytest.mark.parametrize("frames,S,N,K,rc", [(3, 70, 544, 256, 288), (2, 333, 288, 64, 0), (4, 129, 100, 256, 0)])
def test_gemm_row_periodic_residual(oracle, frames, S, N, K, rc):
    """res[row % S] on the first rc columns only: the pos . W^T term of the merged MSDeformAttn projection"""
    from s2d_amd import ops
    M = frames * S
    A = synth.randn(2, 1, (M, K))
    B = synth.randn(2, 2, (N, K))
    bi = synth.randn(2, 3, (N,))
    ldr = rc or N
    res = synth.randn(2, 4, (S, ldr))
    ref = np.einsum("mk,nk->mn", A.astype(np.float64), B.astype(np.float64)) + bi
    ref[:, :ldr] += np.tile(res, (frames, 1))
    out = ops.gemm_nt(_dev(A), _dev(B), bias=_dev(bi), res=_dev(res), res_rows=S, res_cols=rc).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(200, 256, 256), (200, 2048, 256), (200, 256, 2048), (200, 2, 256), (1, 768, 256), (126, 288, 256), (96, 256, 1024),
                                   (37, 70, 64), (256, 320, 128), (33, 65, 192)])
def test_gemm_few_rows_static_weights(oracle, dense_mode, M, N, K):
    """M <= 256 against static (pre-split) weights: in the split-fp16 mode the one-round-trip kernel of gemm_small.hip (the video
    decoder's query side); every epilogue form -- scale, bias, ReLU, a row-periodic residual on the first columns, an output
    wider than N -- against float64, and the result does not depend on the run (waves add their K slices in a fixed order)"""
    from s2d_amd import ops
    tol = {"f16x3": 2e-6, "bf16x3": 1e-4, "f32": 1e-4}[dense_mode]
    A = _dev(synth.randn(4, 1, (M, K)))
    W = torch.nn.Parameter(_dev(synth.randn(4, 2, (N, K)) * K ** -0.5), requires_grad=False)
    sc = _dev(synth.randn(4, 3, (N,)) * 0.5 + 1)
    bi = _dev(synth.randn(4, 4, (N,)))
    R = _dev(synth.randn(4, 5, (M, N)))
    ref = A.double() @ W.double().t()
    y = ops.gemm_nt(A, W, sc, bi, R, relu=True)
    r = torch.relu(ref * sc.double() + bi.double() + R.double())
    np.testing.assert_allclose(y.cpu().numpy(), r.cpu().numpy(), rtol=tol, atol=tol * float(r.abs().max()))
    assert torch.equal(ops.gemm_nt(A, W, sc, bi, R, relu=True), y)
    # plain product into a wider output (padded row stride); the padding stays untouched
    wide = torch.full((M, N + 3), 7.0, device="cuda")
    ops.gemm_nt(A, W, out=wide)
    np.testing.assert_allclose(wide[:, :N].cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()))
    assert bool((wide[:, N:] == 7.0).all())
    # residual of S rows repeated down the output, on the first rc columns only
    if M % 2 == 0 and N >= 8:
        S, rc = M // 2, (N // 2) // 4 * 4
        res = _dev(synth.randn(4, 6, (S, rc)))
        y2 = ops.gemm_nt(A, W, bias=bi, res=res, res_rows=S, res_cols=rc)
        r2 = ref + bi.double()
        r2[:, :rc] += res.double().repeat(2, 1)
        np.testing.assert_allclose(y2.cpu().numpy(), r2.cpu().numpy(), rtol=tol, atol=tol * float(r2.abs().max()))


@pytest.mark.parametrize("M,N,K,periodic", [(1000, 256, 64, False), (777, 512, 128, False), (900, 72, 96, True), (515, 256, 32, False)])
def test_short_k_static_weights_with_residual(oracle, dense_mode, M, N, K, periodic):
    """K <= 128 against static weights with a residual: in the split-fp16 mode the 128 x 64 kernel's variant that fetches the
    residual tile before the first k-tile (the 1 x 1 convolutions closing a res2 / res3 bottleneck); ragged M, a column count that
    is not a whole tile, a row-periodic residual"""
    from s2d_amd import ops
    tol = {"f16x3": 2e-6, "bf16x3": 1e-4, "f32": 1e-4}[dense_mode]
    A = _dev(synth.randn(6, 1, (M, K)))
    W = torch.nn.Parameter(_dev(synth.randn(6, 2, (N, K)) * K ** -0.5), requires_grad=False)
    sc = _dev(synth.randn(6, 3, (N,)) * 0.5 + 1)
    bi = _dev(synth.randn(6, 4, (N,)))
    S = M // 3 if periodic else M
    R = _dev(synth.randn(6, 5, (S, N)))
    y = ops.gemm_nt(A, W, sc, bi, R, relu=True, res_rows=S if periodic else 0)
    Rf = R.double().repeat(3, 1) if periodic else R.double()
    ref = torch.relu((A.double() @ W.double().t()) * sc.double() + bi.double() + Rf)
    np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()))
    # in place on the residual (the trunk's bottleneck output overwrites nothing it still needs: element-wise read before write)
    y2 = ops.gemm_nt(A, W, sc, bi, R if not periodic else Rf.float().contiguous(), relu=True)
    np.testing.assert_allclose(y2.cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()))


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,p", [(2, 16, 24, 64, 64, 3, 1, 1), (1, 33, 47, 4, 64, 7, 2, 3),
                                                   (2, 16, 24, 256, 128, 1, 2, 0), (1, 20, 20, 128, 128, 3, 2, 1),
                                                   (1, 8, 12, 512, 2048, 1, 1, 0)])
def test_conv_nhwc(oracle, N, H, W, Cin, Cout, k, s, p):
    from s2d_amd import ops
    x = synth.randn(2, 1, (N, Cin, H, W))
    w = synth.randn(2, 2, (Cout, Cin, k, k), 1.0 / np.sqrt(Cin * k * k))
    sc = synth.randn(2, 3, (Cout,)) * 0.2 + 1
    bi = synth.randn(2, 4, (Cout,))
    ref = oracle.conv2d(x.astype(np.float64), w.astype(np.float64), None, s, p)
    ref = ref * sc[None, :, None, None] + bi[None, :, None, None]
    res = synth.randn(2, 5, ref.shape)
    ref = np.maximum(ref + res, 0)
    y = ops.conv2d_nhwc(_dev(x.transpose(0, 2, 3, 1)), _dev(w.transpose(0, 2, 3, 1)), s, p, _dev(sc), _dev(bi),
                        _dev(res.transpose(0, 2, 3, 1)), relu=True).cpu().numpy().transpose(0, 3, 1, 2)
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_presplit_static_weights(oracle, dense_mode):
    """a static B (nn.Parameter / marked tensor) goes through the cached pre-split fp16 image: bit-identical to the
    on-the-fly split, refreshed when the parameter changes in place, never used for unmarked tensors"""
    from s2d_amd import ops
    M, N, K = 700, 288, 260
    A = _dev(synth.randn(3, 1, (M, K)))
    W = torch.nn.Parameter(_dev(synth.randn(3, 2, (N, K))), requires_grad=False)
    ops.clear_weight_cache()
    ref = ops.gemm_nt(A, W.data.clone())                  # plain tensor: split on the fly
    n0 = len(ops._SPLIT)
    out = ops.gemm_nt(A, W)
    assert torch.equal(out, ref)
    assert len(ops._SPLIT) == n0 + (1 if dense_mode in ("f16x3", "bf16x3") else 0)
    out_v = ops.gemm_nt(A, W[32:160])                      # a row view of the parameter
    assert torch.equal(out_v, ref[:, 32:160])
    with torch.no_grad():
        W.mul_(2.0)                                        # in-place update bumps the version: the image is rebuilt
    assert torch.equal(ops.gemm_nt(A, W), ops.gemm_nt(A, W.data.clone()))
    # conv weights marked static
    x = _dev(synth.randn(3, 3, (2, 20, 24, 64)))
    w = _dev(synth.randn(3, 4, (128, 3, 3, 64)) * 0.05)
    y0 = ops.conv2d_nhwc(x, w, 1, 1)
    y1 = ops.conv2d_nhwc(x, ops.mark_static(w.clone()), 1, 1)
    if dense_mode == "f16x3":          # static 3x3 weights take the input-halo kernel: same products, (channel block, tap) order
        assert torch.allclose(y0, y1, rtol=1e-5, atol=1e-5 * float(y0.abs().max()))
    else:
        assert torch.equal(y0, y1)
    ops.clear_weight_cache()


@pytest.mark.parametrize("M,N,K,with_res", [(300, 256, 1024, False), (5000, 1024, 256, True), (77, 64, 128, True)])
def test_gemm_gate_epilogue(oracle, dense_mode, M, N, K, with_res):
    """dgrad GEMM with the ReLU / dropout gate in its epilogue == GEMM followed by the separate gate pass"""
    from s2d_amd import ops, backward as Bk
    if dense_mode != "f16x3":
        pytest.skip("the gate epilogue exists in the split-fp16 kernels")
    A = synth.randn(5, 1, (M, K))
    B = synth.randn(5, 2, (N, K))
    g = synth.randn(5, 3, (M, N))
    g[g < 0.3] = 0.0                                                  # a ReLU output: exact zeros where the unit was off
    res = synth.randn(5, 4, (M, N)) if with_res else None
    ref = A.astype(np.float64) @ B.astype(np.float64).T + (res if with_res else 0.0)
    ref = np.where(g > 0, ref * 1.25, 0.0)
    out = ops.gemm_nt_gate(_dev(A), _dev(B), _dev(g), 1.25, res=_dev(res) if with_res else None)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # the path the models use: input_grad(dy, w, gate) with w [N_out, K_in] -> dx gated by the layer input's ReLU output
    dy, w = _dev(synth.randn(5, 5, (M, 96))), torch.nn.Parameter(_dev(synth.randn(5, 6, (96, N))))
    want = Bk.relu_scale_backward(Bk.input_grad(dy, w), _dev(g))
    got = Bk.input_grad(dy, w, gate=_dev(g))
    assert torch.equal(want, got)


@pytest.mark.parametrize("M,N,K,with_res,relu", [(700, 256, 1024, True, False), (4096, 1024, 256, False, True), (130, 64, 224, False, False)])
def test_gemm_presplit_a(oracle, dense_mode, M, N, K, with_res, relu):
    """A handed over as its fp16 hi/lo row image (what an upstream kernel could write instead of fp32): bit-equal to the GEMM that
    splits A itself"""
    from s2d_amd import ops
    if dense_mode != "f16x3":
        pytest.skip("the row image is the split-fp16 kernels' format")
    A = _dev(synth.randn(6, 1, (M, K)))
    W = torch.nn.Parameter(_dev(synth.randn(6, 2, (N, K))), requires_grad=False)
    b = _dev(synth.randn(6, 3, (N,)))
    R = _dev(synth.randn(6, 4, (M, N))) if with_res else None
    want = ops.gemm_nt(A, W, bias=b, res=R, relu=relu)
    got = ops.gemm_nt_presplit(ops.split_rows(A), M, K, W, bias=b, res=R, relu=relu)
    assert torch.equal(want, got)
    ref = A.double().cpu().numpy() @ W.detach().double().cpu().numpy().T + b.double().cpu().numpy()
    if with_res:
        ref = ref + R.double().cpu().numpy()
    if relu:
        ref = np.maximum(ref, 0)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("N,H,W,Cin,Cout,with_res", [(2, 16, 32, 64, 128, False), (1, 23, 40, 128, 192, True), (3, 9, 17, 32, 68, True),
                                                      (1, 46, 80, 256, 256, False), (2, 40, 48, 64, 64, True), (1, 33, 50, 64, 48, False),
                                                      (1, 16, 16, 32, 64, False)])
def test_conv3x3_halo(oracle, dense_mode, N, H, W, Cin, Cout, with_res):
    """3x3 / stride 1 / pad 1 with static weights (the input-halo kernel in the default mode): image borders, patches that hang
    over the bottom / right edge, Cout that is not a multiple of the 128-wide tile, Cout <= 64 (16 x 16 patches x 64 channels: the res2
    bottlenecks), scale + bias + residual + ReLU"""
    from s2d_amd import ops
    x = synth.randn(7, 1, (N, H, W, Cin))
    w = (synth.randn(7, 2, (Cout, 3, 3, Cin)) / np.sqrt(9.0 * Cin)).astype(np.float32)
    sc = (synth.randn(7, 3, (Cout,)) * 0.5 + 1).astype(np.float32)
    bi = synth.randn(7, 4, (Cout,))
    res = synth.randn(7, 5, (N, H, W, Cout)) if with_res else None
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xt, wt, padding=1).permute(0, 2, 3, 1).numpy() * sc + bi
    if with_res:
        ref = ref + res
    ref = np.maximum(ref, 0)
    out = ops.conv2d_nhwc(_dev(x), ops.mark_static(_dev(w)), 1, 1, scale=_dev(sc), bias=_dev(bi), res=_dev(res) if with_res else None,
                          relu=True).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("N,H,W,Cout", [(1, 33, 47, 64), (2, 64, 96, 64), (1, 70, 130, 48), (3, 32, 32, 64)])
def test_stem_conv_halo(oracle, dense_mode, N, H, W, Cout):
    """the R50 stem geometry (7 x 7 / stride 2 / pad 3, NHWC4 input) with static weights -- in the split-fp16 mode its own input-halo
    kernel: image borders on all four sides, patches overhanging the output, odd sizes, FrozenBN scale / bias + ReLU"""
    from s2d_amd import ops
    x = synth.randn(8, 1, (N, H, W, 4)); x[..., 3] = 0
    w = (synth.randn(8, 2, (Cout, 7, 7, 4)) / 12.0).astype(np.float32)
    sc = (synth.randn(8, 3, (Cout,)) * 0.5 + 1).astype(np.float32)
    bi = synth.randn(8, 4, (Cout,))
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(w).double().permute(0, 3, 1, 2)
    ref = np.maximum(torch.nn.functional.conv2d(xt, wt, stride=2, padding=3).permute(0, 2, 3, 1).numpy() * sc + bi, 0)
    out = ops.conv2d_nhwc(_dev(x), ops.mark_static(_dev(w)), 2, 3, scale=_dev(sc), bias=_dev(bi), relu=True).cpu().numpy()
    tol = {"f16x3": 2e-6, "bf16x3": 1e-4, "f32": 1e-4}[dense_mode]
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.abs(ref).max())


@pytest.mark.parametrize("switch,select", [("S2D_GEMM_WS", "gemm or conv or dropout"), ("S2D_GEMM_W128", "bf16x3 and (gemm or presplit)"),
                                           ("S2D_CONV_HALO", "f16x3 and conv")])
def test_forced_kernel_subprocess(dense_mode, switch, select):
    """kernels the default dispatch only picks for large shapes, forced onto every eligible launch (value 2 of their switch) and
    run against the same oracle cases: the opt-in wave-specialised persistent kernel (S2D_GEMM_WS) and the 128 x 64-per-wave
    split-bf16 kernel (S2D_GEMM_W128), the input-halo 3x3 convolution on every patch geometry (S2D_CONV_HALO); in a child process because the switches are read once per process"""
    import os
    import subprocess
    import sys
    if os.environ.get("S2D_GEMM_WS") or os.environ.get("S2D_GEMM_W128") or os.environ.get("S2D_CONV_HALO") or dense_mode != "f16x3":
        pytest.skip("once, from the default mode, outside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{switch: "2"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_dense.py"), os.path.join(root, "tests", "test_gpu_dropin.py"),
                        "-k", select], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_conv3x3_halo_fragment_pipeline_opt_in_is_bitwise_equal(monkeypatch):
    """S2D_CONV_HALO_PIPE=1 (fragment reads one MFMA group ahead; measured and not adopted, round 5) computes the same products in the
    same order: bit-identical outputs on both halo geometries (Cout > 64 and Cout <= 64), two channel blocks and more"""
    import torch
    from s2d_amd import ops
    g = torch.Generator().manual_seed(5)
    for (n, H, W, Ci, Co) in [(2, 24, 40, 64, 128), (1, 33, 47, 96, 64), (2, 16, 16, 256, 256)]:
        x = torch.randn((n, H, W, Ci), generator=g).cuda()
        w = torch.nn.Parameter(torch.randn((Co, 3, 3, Ci), generator=g).mul_((9 * Ci) ** -0.5).cuda(), requires_grad=False)
        b = torch.randn((Co,), generator=g).cuda()
        monkeypatch.setenv("S2D_CONV_HALO_PIPE", "0")
        y0 = ops.conv2d_nhwc(x, w, stride=1, pad=1, bias=b, relu=True)
        monkeypatch.setenv("S2D_CONV_HALO_PIPE", "1")
        y1 = ops.conv2d_nhwc(x, w, stride=1, pad=1, bias=b, relu=True)
        assert torch.equal(y0, y1)
