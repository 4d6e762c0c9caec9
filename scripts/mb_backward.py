"""backward of the R50 trunk and the pixel decoder at the metric's size (16 frames of 720p = one GPU's c4 batch, student only):
time of forward-with-tape and of the backward pass built on the forward kernels (s2d_amd/backward.py)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd.modeling import ResNet50, MSDeformAttnPixelDecoder
dev = torch.device("cuda")
torch.manual_seed(0)
F_, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 736, 1280
net, pd = ResNet50().to(dev), MSDeformAttnPixelDecoder().to(dev)
x = torch.randn((F_, H, W, 4), device=dev); x[..., 3] = 0


def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return time.perf_counter() - t0, r


with torch.no_grad():
    for it in range(2):
        for p in list(net.parameters()) + list(pd.parameters()):
            p.grad = None
        tape_b, tape_p = [], []
        t_fb, feats = sync_time(lambda: net(x, tape_b))
        t_fp, (mf, ms) = sync_time(lambda: pd.forward_features(feats, tape_p))
        d_mf = torch.randn_like(mf); d_outs = [torch.randn_like(t) for t, _ in ms]
        t_bp, grads = sync_time(lambda: pd.backward_features(tape_p[0], d_mf, d_outs))
        t_bb, _ = sync_time(lambda: net.backward(tape_b, grads))
        print(f"run {it}: {F_} frames {H}x{W}: trunk fwd {t_fb*1e3:.1f} ms, bwd {t_bb*1e3:.1f} ms | pixel decoder fwd {t_fp*1e3:.1f} ms, bwd {t_bp*1e3:.1f} ms "
              f"| peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
        del tape_b, tape_p, feats, mf, ms, grads
