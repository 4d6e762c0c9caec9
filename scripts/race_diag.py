"""Two-stream wrong-value check (round 2; history and the full variant matrix: profiles/r2_two_stream_diagnosis/).
Teacher forward on a side HIP stream, student forward on the main stream (the schedule of
KDVideoMaskFormer.forward_losses); every full-map attention-mask launch is recorded with the mask logits it read, launched a
second time right behind the first, and after a device synchronise recomputed in isolation.  Prints the number of wrong
32-bit mask words per repetition and `TOTAL ...` at the end.

    [S2D_ATTN_MASK_DWORD_TAPS=1] [S2D_DIAG_ONE_STREAM=1] python scripts/race_diag.py [reps] [config]

S2D_ATTN_MASK_DWORD_TAPS=1 selects round 1's kernel source (attn_mask_kernel_dword_taps, csrc/attn.hip), which showed
~9 wrong words per repetition while the library was built with SLP vectorisation and 0 since it is not.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from s2d_amd import ops
from s2d_amd.modeling import build_kd_model
import s2d_amd.modeling.video_decoder as vd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
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


def bits_wrap(ml, *a, **k):
    compact = k.get("compact", False) or (len(a) > 7 and a[7])
    if not active[0] or compact:
        return orig_bits(ml, *a, **k)
    r = orig_bits(ml, *a, **k)
    r2 = orig_bits(ml, *a, **k)            # the same launch again, right behind the first
    rec.append((torch.cuda.current_stream() == side and not one_stream, ml, a, r[0], r2[0]))
    return r


vd.ops.attn_mask_bits = bits_wrap


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


print(f"env: DWORD_TAPS={os.environ.get('S2D_ATTN_MASK_DWORD_TAPS', '0')} ONE_STREAM={int(one_stream)}", flush=True)
run()
tot_first = tot_second = 0
for rep in range(reps):
    keep = run()
    bad1 = bad2 = 0
    for on_side, ml, a, b1, b2 in rec:
        good = orig_bits(ml, *a)[0]
        torch.cuda.synchronize()
        d1, d2 = (good != b1), (good != b2)
        bad1 += int(d1.sum()); bad2 += int(d2.sum())
        for (bb, kk, ww) in d1.nonzero()[:4].tolist():
            gw, bw = int(good[bb, kk, ww]) & 0xFFFFFFFF, int(b1[bb, kk, ww]) & 0xFFFFFFFF
            print(f"  rep{rep} {'side' if on_side else 'main'} level({a[5]}x{a[6]}) clip {bb} key {kk} word {ww}: xor {gw ^ bw:08x}; "
                  f"second launch {'wrong' if bool(d2[bb, kk, ww]) else 'right'}", flush=True)
    tot_first += bad1; tot_second += bad2
    print(f"rep {rep}: {len(rec)} full-map mask launches; wrong words first launch {bad1}, second launch {bad2}", flush=True)
    del keep
print(f"TOTAL over {reps} reps: first-launch wrong words {tot_first}, second-launch wrong words {tot_second}", flush=True)
