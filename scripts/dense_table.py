"""Per-shape table of the step's dense launches (NT GEMMs, convolutions, MSDeformAttn, decoder attention): HIP-event time of every
launch in one-stream steps at BASELINE config c4, next to its floor = max(algorithmic bytes / 6.3 TB/s, 3 x flops / 2.5 PFLOP/s)
(6.3 TB/s = the copy rate measured on this part; 3 MFMA passes of the split-fp16 scheme at the dense f16 peak).
Usage: python scripts/dense_table.py [--steps 3] [--amp]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, build_kd_model, set_amp_compute
    dev = torch.device("cuda", 0)
    B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0), dropout=0.3).to(dev)
    model.train()
    frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
    bench.calibrate_teacher(model, ops.normalize_pad(frames))
    mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()
    if a.amp:
        set_amp_compute(model, True)
    model.overlap_teacher = model.overlap_criteria = False

    def step():
        images = ops.normalize_pad(frames, 32, mean, std)
        return sum(model.forward_losses(images, TargetSet.from_list(masks, device=dev)).values())

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ops.PROFILE = []
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr, ops.PROFILE = ops.PROFILE, None
    agg = collections.OrderedDict()
    for s, e, flops, tag in pr:
        if tag is None:
            continue
        k = tag[:5]
        ent = agg.setdefault(k, [0, 0.0, flops, tag[5]])
        ent[0] += 1
        ent[1] += s.elapsed_time(e)
    rows = []
    for k, (n, ms, flops, byts) in agg.items():
        per = ms / n
        fl_h = byts / 6.3e12 * 1e3
        fl_m = 3.0 * flops / 2.5e15 * 1e3
        floor = max(fl_h, fl_m)
        rows.append((ms / a.steps, k, n / a.steps, per, flops / per / 1e9, byts / per / 1e9, floor, "hbm" if fl_h >= fl_m else "mfma", per / floor if floor > 0 else float("nan")))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    totfloor = sum(r[6] * r[2] for r in rows)
    print(f"# dense launches per one-stream step: {tot:.2f} ms, sum of floors {totfloor:.2f} ms, amp={a.amp}")
    print("# ms/step | kind b M N K | launches/step | ms/launch | TFLOP/s alg | TB/s alg | floor ms (bound) | x floor | excess ms/step")
    for r in rows[: a.top]:
        ms, k, n, per, tf, tb, floor, bound, x = r
        print(f"{ms:7.3f} | {k[0]:5s} {k[1]:3d} {k[2]:7d} {k[3]:5d} {k[4]:5d} | {n:5.1f} | {per:7.4f} | {tf:6.1f} | {tb:5.2f} | {floor:7.4f} {bound:4s} | {x:5.2f} | {ms - floor * n:6.3f}")


if __name__ == "__main__":
    main()
