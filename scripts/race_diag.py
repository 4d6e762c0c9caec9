"""Two-stream stale-read diagnosis (round 2).  Teacher forward on a side stream, student forward on the main stream (the
schedule of KDVideoMaskFormer.forward_losses); every full-map attention-mask launch is recorded together with the mask
logits it read, launched a second time right behind the first, and after a device synchronise recomputed in isolation.
Wrong words are mapped back to the mask-logit GEMM tile that produced their source pixels.

    S2D_DIAG_ATTN_MASK=0|1|2|3  S2D_DIAG_GEMM_RELEASE=0|1  S2D_DIAG_ONE_STREAM=0|1  python scripts/race_diag.py [reps] [config]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from s2d_amd import ops
from s2d_amd.modeling import build_kd_model
import s2d_amd.modeling.video_decoder as vd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = sys.argv[2] if len(sys.argv) > 2 else "c4"
one_stream = os.environ.get("S2D_DIAG_ONE_STREAM", "0") == "1"
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS[cfg]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
images = ops.normalize_pad(frames)
main = torch.cuda.current_stream()
side = main if one_stream else torch.cuda.Stream()

rec = []
active = [False]
orig_bits = ops.attn_mask_bits
orig_gemm = ops.gemm_nt
DUMP = os.environ.get("S2D_DIAG_ATTN_MASK", "0") == "4"
CLONE = os.environ.get("S2D_DIAG_CLONE", "0") == "1"
SENT = float(os.environ.get("S2D_DIAG_SENTINEL", "0"))      # != 0: the mask-logit GEMM's output is pre-filled with this value


def bits_wrap(ml, *a, **k):
    compact = k.get("compact", False) or (len(a) > 7 and a[7])
    if not active[0] or compact:
        return orig_bits(ml, *a, **k)
    src = ml.clone() if CLONE else ml             # CLONE: the consumer reads a copy written by a plain copy kernel
    dbg = None
    if DUMP:
        Bq, Qq, Tq, hm, wm, hl, wl = a[:7]
        dbg = torch.full((Bq, Tq * hl * wl, 36, 6), float("nan"), device=ml.device)
        os.environ["S2D_DIAG_DBG_PTR"] = str(dbg.data_ptr())
    r = orig_bits(src, *a, **k)
    if DUMP:
        os.environ["S2D_DIAG_DBG_PTR"] = "0"
    r2 = orig_bits(src, *a, **k)           # the same launch again, right behind the first
    rec.append((torch.cuda.current_stream() == side and not one_stream, ml, a, r[0], r2[0], dbg))
    return r


def gemm_wrap(A, Bm, *a, **kw):
    if active[0] and SENT != 0.0 and kw.get("out") is not None and A.dim() == 3:
        kw["out"].fill_(SENT)
    return orig_gemm(A, Bm, *a, **kw)


vd.ops.attn_mask_bits = bits_wrap
vd.ops.gemm_nt = gemm_wrap


def run():
    rec.clear()
    side.wait_stream(main)
    active[0] = True
    with torch.cuda.stream(side):
        t = model.teacher(images, True, aux_masks=True)
    s = model.student(images, True)
    active[0] = False
    main.wait_stream(side)
    torch.cuda.synchronize()
    return t, s


print(f"env: ATTN_MASK={os.environ.get('S2D_DIAG_ATTN_MASK', '0')} GEMM_RELEASE={os.environ.get('S2D_DIAG_GEMM_RELEASE', '0')} "
      f"ONE_STREAM={int(one_stream)} AMD_OPT_FLUSH={os.environ.get('AMD_OPT_FLUSH')} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}",
      flush=True)
run()
tot_first = tot_second = 0
shown = 0
for rep in range(reps):
    keep = run()
    bad1 = bad2 = 0
    for on_side, ml, a, b1, b2, dbg in rec:
        good = orig_bits(ml, *a)[0]
        torch.cuda.synchronize()
        if dbg is not None:                                   # every raw tap value the first launch loaded vs the logits in memory
            Bq, Qq, Tq, hm, wm, hl, wl = a[:7]
            idx = ops.attn_mask_tap_index(Tq, hm, wm, hl, wl, ml.device)          # [K*4] source rows
            f32 = torch.float32
            ys = torch.arange(hl, dtype=f32, device=ml.device); xs = torch.arange(wl, dtype=f32, device=ml.device)
            sy = torch.clamp((torch.tensor(float(hm), dtype=f32, device=ml.device) / hl) * (ys + 0.5) - 0.5, min=0.0)
            sx = torch.clamp((torch.tensor(float(wm), dtype=f32, device=ml.device) / wl) * (xs + 0.5) - 0.5, min=0.0)
            ly = (sy - sy.to(torch.int64).to(f32)); lx = (sx - sx.to(torch.int64).to(f32))
            LY = ly[None, :, None].expand(Tq, hl, wl).reshape(-1, 1); LX = lx[None, None, :].expand(Tq, hl, wl).reshape(-1, 1)
            HY, HX = 1.0 - LY, 1.0 - LX
            for bb in range(Bq):
                true = ml[bb].index_select(0, idx)[:, 64:Qq].view(-1, 4, Qq - 64).permute(0, 2, 1)     # [K, 36, 4]
                got = dbg[bb][:, :Qq - 64]
                neq = (true != got[..., :4])
                nbad = int(neq.sum())
                # the interpolated value, in the kernel's operation order (no fused multiply-add; IEEE float32 on both sides)
                t = got[..., :4]
                ev = HY * (HX * t[..., 0] + LX * t[..., 1]) + LY * (HX * t[..., 2] + LX * t[..., 3])
                vbad = (ev != got[..., 4])
                # the lane's nibble after query j: bits 0..j of sign(v) for its four queries
                sign = (got[..., 4] < 0).view(-1, 9, 4)
                en = torch.cumsum(sign.to(f32) * torch.tensor([1.0, 2.0, 4.0, 8.0], device=ml.device), -1).view(-1, 36)
                nibbad = (en != got[..., 5])
                # the stored word vs the nibbles the lanes held at the end
                fin = got[..., 5].view(-1, 9, 4)[..., 3].to(torch.int64)                # [K, 9] lanes 16..24
                w2 = sum(fin[:, i] << (4 * i) for i in range(8))
                w3 = fin[:, 8]
                st = b1[bb].to(torch.int64) & 0xFFFFFFFF
                wbad2, wbad3 = (w2 != st[:, 2]), (w3 != st[:, 3])
                print(f"  rep{rep} {'side' if on_side else 'main'} level({hl}x{wl}) b{bb}: raw taps != memory {nbad}; interpolated v != recomputed {int(vbad.sum())}; "
                      f"lane nibble != sign bits {int(nibbad.sum())}; stored word 2 != packed nibbles {int(wbad2.sum())}, word 3 {int(wbad3.sum())}", flush=True)
                for (kk, qq) in vbad.nonzero()[:6].tolist():
                    print(f"     v: key{kk} q{64 + qq}: taps {t[kk, qq].tolist()} kernel v {float(got[kk, qq, 4])!r} expected {float(ev[kk, qq])!r}", flush=True)
                for kk in wbad2.nonzero()[:6].flatten().tolist():
                    print(f"     word2: key{kk}: stored {int(st[kk, 2]):08x} packed-from-lanes {int(w2[kk]):08x} lanes' nibbles {fin[kk].tolist()}", flush=True)
        d1, d2 = (good != b1), (good != b2)
        n1, n2 = int(d1.sum()), int(d2.sum())
        bad1 += n1; bad2 += n2
        if n1 and shown < 40:
            Bq, Qq, Tq, hm, wm, hl, wl = a[:7]
            npix = Tq * hm * wm
            tiles_m = (npix + 127) // 128
            tiles_n = (Qq + 63) // 64
            per_xcd = (tiles_m * tiles_n) // 8
            for (bb, kk, ww) in d1.nonzero()[:6].tolist():
                x = kk % wl; y = (kk // wl) % hl; t = kk // (wl * hl)
                sy = max((hm / hl) * (y + 0.5) - 0.5, 0.0); sx = max((wm / wl) * (x + 0.5) - 0.5, 0.0)
                y0, x0 = int(sy), int(sx)
                rows = sorted({(t * hm + yy) * wm + xx for yy in (y0, min(y0 + 1, hm - 1)) for xx in (x0, min(x0 + 1, wm - 1))})
                tl = sorted({r // 128 for r in rows})
                ids = [tm * tiles_n + (1 if ww >= 2 else 0) for tm in tl]      # remapped tile id (tile_m-major)
                pos = [(i // per_xcd, i % per_xcd, per_xcd) for i in ids]      # (xcd chunk, index within chunk, chunk length)
                gw, bw = int(good[bb, kk, ww]) & 0xFFFFFFFF, int(b1[bb, kk, ww]) & 0xFFFFFFFF
                print(f"  rep{rep} {'side' if on_side else 'main'} level({hl}x{wl}) b{bb} key{kk} (t{t},y{y},x{x}) word{ww} xor {gw ^ bw:08x} "
                      f"second-launch {'wrong' if bool(d2[bb, kk, ww]) else 'right'}; src rows {rows} tile_m {tl} (chunk, idx, len) {pos}", flush=True)
                shown += 1
    tot_first += bad1; tot_second += bad2
    print(f"rep {rep}: {len(rec)} full-map mask launches; wrong words first launch {bad1}, second launch {bad2}", flush=True)
    del keep
print(f"TOTAL over {reps} reps: first-launch wrong words {tot_first}, second-launch wrong words {tot_second}", flush=True)
