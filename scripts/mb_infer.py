"""eval-side step (csrc/infer.hip) at video sizes: kernel times and achieved bandwidth of the selection, the fused
two-stage resize + threshold, and the pairwise popcount; the plain-torch equivalent of the reference's code beside it"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from s2d_amd import ops
from s2d_amd.modeling.postprocess import inference_video
dev = torch.device("cuda")
torch.manual_seed(0)


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


for (Q, K, T, hm, wm, Hp, Wp, ih, iw, oh, ow, tag) in [
        (100, 10, 36, 184, 320, 736, 1280, 720, 1280, 720, 1280, "720p x36, K=10"),
        (100, 100, 36, 184, 320, 736, 1280, 720, 1280, 720, 1280, "720p x36, K=100 + NMS"),
        (100, 10, 36, 96, 160, 384, 640, 360, 640, 720, 1280, "360p->720p x36, K=10")]:
    cls = torch.randn((Q, 2), device=dev)
    lo = torch.randn((1, Q * T, hm // 8, wm // 8), device=dev)
    ml_q = F.interpolate(lo, size=(hm, wm), mode="bilinear").view(Q, T, hm, wm) * 3
    ml = torch.zeros((T * hm * wm, 128), device=dev)
    ml[:, :Q] = ml_q.reshape(Q, -1).t()
    dims, pad, img, out = (T, hm, wm), (Hp, Wp), (ih, iw), (oh, ow)
    sc, q, lb = ops.infer_select(cls, K)
    t_sel = t(lambda: ops.infer_select(cls, K))
    t_msk = t(lambda: ops.infer_masks(ml, dims, pad, img, out, q, want_bits=True))
    m, bits = ops.infer_masks(ml, dims, pad, img, out, q, want_bits=True)
    t_pair = t(lambda: ops.mask_pair_counts(bits))
    nms = K > 10
    t_all = t(lambda: inference_video(cls, ml, dims, pad, img, out, K, nms, 0.75), 3)
    wbytes = K * T * oh * ow * (1 + 1 / 8)
    rbytes = T * hm * wm * 4 * K
    print(f"{tag}: select {t_sel*1e6:.0f} us | gather+resize {t_msk*1e3:.3f} ms = {(wbytes+2*rbytes)/t_msk/1e9:.0f} GB/s algorithmic "
          f"({wbytes/1e6:.0f} MB out) | pair counts {t_pair*1e3:.3f} ms = {K*(K+1)/2*bits.shape[1]*8/t_pair/1e9:.0f} GB/s pair-bytes "
          f"| whole call incl. D2H of masks {t_all*1e3:.1f} ms", flush=True)

    def torch_ref():
        up = F.interpolate(ml_q, size=(Hp, Wp), mode="bilinear", align_corners=False)
        s = F.softmax(cls, -1)[:, :-1].flatten()
        sv, ti = s.topk(K, sorted=True)
        pm = up[ti][:, :, :ih, :iw]
        pm = F.interpolate(pm, size=(oh, ow), mode="bilinear", align_corners=False)
        return pm > 0
    try:
        t_ref = t(torch_ref, 3)
        mm = torch_ref()
        agree = (mm.view(torch.uint8) == m).float().mean().item()
        print(f"    torch ops on the same GPU (upsample all Q, top-k, crop, resize, > 0; no NMS, no D2H): {t_ref*1e3:.1f} ms; "
              f"masks agree on {agree*100:.4f} % of pixels", flush=True)
        if nms:
            def torch_nms_pairs():
                mk = mm.flatten(1)
                n = 0
                for i in range(0, 10):                      # 10 of the K(K-1)/2 pair evaluations, as the reference does them
                    a = torch.sum(mk[0] & mk[i + 1]).float(); b = torch.sum(mk[0] | mk[i + 1]).float()
                    n += float(a / b if b > 0 else 0.0)
                return n
            print(f"    reference-style NMS pair evaluation: {t(torch_nms_pairs, 2)*1e3/10:.2f} ms per pair (x up to {K*(K-1)//2} pairs)", flush=True)
    except RuntimeError as e:
        print("    torch reference failed:", str(e)[:100])
    del ml, ml_q, m, bits
    torch.cuda.empty_cache()

# device RLE of the K=10 720p case: encode where the masks were made instead of copying them out
from s2d_amd import rle
Q, K, T, hm, wm = 100, 10, 36, 184, 320
cls = torch.randn((Q, 2), device=dev)
lo = torch.randn((1, Q * T, hm // 8, wm // 8), device=dev)
ml_q = F.interpolate(lo, size=(hm, wm), mode="bilinear").view(Q, T, hm, wm) * 3
ml = torch.zeros((T * hm * wm, 128), device=dev)
ml[:, :Q] = ml_q.reshape(Q, -1).t()
sc, q, lb = ops.infer_select(cls, K)
m, _ = ops.infer_masks(ml, (T, hm, wm), (736, 1280), (720, 1280), (720, 1280), q)
t_rle = t(lambda: rle.encode_video_predictions(m), 3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
col = torch.empty((K * T, 1280), device=dev, dtype=torch.int32); nb = torch.empty((K * T,), device=dev, dtype=torch.int32)
ar = torch.empty_like(nb); bb = torch.empty((K * T, 4), device=dev, dtype=torch.int32)
from s2d_amd._lib import lib
e0.record(); lib().call("s2d_rle_count_u8", m, K * T, 720, 1280, col, nb, ar, bb, torch.cuda.current_stream().cuda_stream); e1.record()
torch.cuda.synchronize()
segs = rle.encode_video_predictions(m)
nbytes = sum(len(s["counts"]) for inst in segs for s in inst)
print(f"device RLE of the 360 720p masks: {t_rle*1e3:.1f} ms per call incl. host string building ({nbytes/1e3:.0f} kB of RLE instead of 373 MB of bytes); "
      f"count pass alone {e0.elapsed_time(e1):.2f} ms on the device")
