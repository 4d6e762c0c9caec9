"""Measured kernel time per family of a one-stream step (a rocprofv3 --kernel-trace --stats csv of bench.py --one-stream) next to a stated
floor for BASELINE config c4, the table VERDICT r2 item 2 asks for.

Floors: memory-bound families = algorithmic bytes (each tensor a launch must read or write, once) / 6.3 TB/s (the copy rate measured on this
part; the 8 TB/s peak is never reached by a copy); dense = per-launch max(bytes / 6.3 TB/s, 3 x flops / 2.5 PFLOP/s) summed (profiles/
r3_dense_table.txt, HIP events); issue-bound families (matcher, point loss) = vector instructions of the arithmetic the reference's formulas
need per element / (1024 SIMDs x 1 instruction per 4 cycles x 2.0 GHz).

usage: python scripts/family_floors.py profiles/r3_kernel_stats_one_stream.csv <steps in the trace> [profiles/r3_dense_table.txt]"""
import csv
import re
import sys

# c4 geometry: 2 clips x 8 frames per network, 2 networks (student + teacher); maps at stride 4: 184 x 320, pyramid 92x160 / 46x80 / 23x40
F, NET = 16, 2
HM, WM, S = 184 * 320, 0, 92 * 160 + 46 * 80 + 23 * 40
C = 256
Q, NL, B, T, P, NGT = 100, 10, 2, 8, 160000, 10
BW = 6.3e12
ISSUE = 1024 * 2.0e9 / 4          # wave instructions per second, whole chip


def ms_bytes(b):
    return b / BW * 1e3


def main():
    stats, steps = sys.argv[1], float(sys.argv[2])
    dense_floor = None
    if len(sys.argv) > 3:
        # rows of scripts/dense_table.py: ms/step | kind b M N K | launches/step | ms/launch | TFLOP/s | TB/s | floor ms (bound) | x floor | excess
        dense_floor = 0.0
        for line in open(sys.argv[3]):
            f = [x.strip() for x in line.split("|")]
            if len(f) == 9 and f[1].split()[0] in ("gemm", "conv"):
                dense_floor += float(f[2]) * float(f[6].split()[0])
    rows = list(csv.DictReader(open(stats)))
    per = {r["Name"]: (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps) for r in rows}

    def fam(*subs, exclude=()):
        t = n = 0.0
        for k, (ms, c) in per.items():
            if any(s in k for s in subs) and not any(e in k for e in exclude):
                t += ms
                n += c
        return t, n

    out = []
    # dense contractions (GEMM / conv kernels; the decoder's query side included)
    t, n = fam("gemm_f16x3", "conv3x3_f16x3", "conv7x7s2", "gemm_small_m")
    enc_rows = F * S                                                     # rows of one network's encoder activations
    # floors of the dense launches: from the per-shape table when given (it includes msda / attention rows: subtract those)
    out.append(("dense contractions (NT GEMM, implicit-GEMM / halo conv; the decoder's query side included)", t, n, dense_floor,
                "per launch max(bytes / 6.3 TB/s, 3 x flops / 2.5 PFLOP/s), summed over profiles/r3_dense_table.txt" if dense_floor else "see profiles/r3_dense_table.txt"))
    # MSDeformAttn gather: value read once + output written once + offsets / logits read once, 12 launches
    t, n = fam("msda_fused_kernel")
    b = 4.0 * enc_rows * (2 * C + 8 * 12 * 3)
    out.append(("MSDeformAttn gather", t, n, n * ms_bytes(b), "value + output + offsets/logits once; bound in fact by the L1 line rate (48 lines per (query, head))"))
    # LayerNorm over the encoder activations: read + write
    t, n = fam("layernorm256_kernel")
    out.append(("LayerNorm (encoder, 309 120 x 256)", t, n, n * ms_bytes(4.0 * enc_rows * C * 2), "read + write"))
    # GroupNorm: statistics (one read) + apply (read + write, + the up-sampled addend on the FPN level)
    t1, n1 = fam("gn_stats_kernel", "gn_reduce_kernel")
    t2, n2 = fam("gn_apply")
    big = F * HM * C * 4.0                                               # one stride-4 map of a network
    lvl = 4.0 * enc_rows * C
    gn_bytes = NET * (2 * big + lvl) + NET * (2 * (2 * big) + big / 16 + 2 * lvl)    # stats reads; applies (2 big maps: r+w, one with the coarse addend; 3 input_proj levels)
    out.append(("GroupNorm (statistics + apply)", t1 + t2, n1 + n2, ms_bytes(gn_bytes), "statistics: one read; apply: read + write (+ addend)"))
    # max-pool 3x3/2 after the stem: read [F,368,640,64], write [F,184,320,64]
    t, n = fam("maxpool_kernel")
    out.append(("max-pool 3x3/2", t, n, NET * ms_bytes(4.0 * F * 64 * (368 * 640 + 184 * 320)), "read + write"))
    # matcher: Q(128 padded) x T x P sampled logits per (layer, clip), 22 vector + 2 transcendental (8-cycle) instructions per 8 samples' lane ...
    t, n = fam("matcher_cost_f16_kernel")
    pairs = NL * B * T * P * 128.0                                       # (padded query, frame, point) evaluations per pass
    out.append(("matcher cost (2 passes)", t, n, n * pairs * (22 + 2 * 2) / 64 / ISSUE * 1e3,
                "22 vector + 2 transcendental (2 issue slots each) instructions per (query, frame, point) at 128 padded queries; 100 real queries: x 0.78"))
    # point loss: rows x oversampled points bilinear samples for the uncertainty (hist<0>), then the selected points
    rows_l = NL * B * NGT * T
    t, n = fam("hist_kernel<0>")
    out.append(("point loss: uncertainty of the oversampled points (hist<0>)", t, n, n * rows_l * 3.0 * P * 30 / 64 / ISSUE * 1e3,
                "30 instructions per oversampled point (point generation, 4 LDS taps, lerp, key, histogram update)"))
    t, n = fam("accumulate_stream_kernel<false>")
    out.append(("point loss: selected points (accumulate_stream)", t, n, n * rows_l * (3.0 * P * 6 + 1.0 * P * 60) / 64 / ISSUE * 1e3,
                "6 instructions per oversampled point (threshold test on the stored logit) + 60 per kept point (target taps, BCE, dice sums)"))
    t, n = fam("gather_rows_kernel", "hist_stream_kernel", "loss_finalize", "select_kernel", "row_")
    out.append(("point loss: row gather / radix-select passes", t, n, 2 * ms_bytes(rows_l * HM * 4.0 * 2 + rows_l * 3.0 * P * 4 * 3), "gather: read + write one logit map per row; select: 3 passes over the stored keys"))
    # KD targets
    t, n = fam("kd_upsample", "nonempty_kernel")
    out.append(("KD target planes + DropLoss predicate", t, n, ms_bytes(B * NGT * T * 736 * 1280 * 2.0 + 4.0 * B * T * HM * 128 + B * NGT * T * 736 * 1280), "write planes + read teacher logits; read planes"))
    # decoder attention
    t, n = fam("cross_attn_kernel", "attn_merge_kernel", "attn_mask_kernel", "self_attn")
    out.append(("decoder attention (masked cross-attention, merges, attention-mask bits)", t, n, None, "small launches (200 queries); see roofline.per_kernel.cross_attn"))
    # the rest
    named = ("gemm_f16x3", "conv3x3_f16x3", "conv7x7s2", "gemm_small_m", "msda_fused_kernel", "layernorm256_kernel", "gn_", "maxpool_kernel", "matcher_cost_f16_kernel",
             "hist_kernel<0>", "accumulate_stream_kernel<false>", "gather_rows_kernel", "hist_stream_kernel", "loss_finalize", "select_kernel", "row_",
             "kd_upsample", "nonempty_kernel", "cross_attn_kernel", "attn_merge_kernel", "attn_mask_kernel", "self_attn")
    t = sum(ms for k, (ms, c) in per.items() if not any(s in k for s in named))
    n = sum(c for k, (ms, c) in per.items() if not any(s in k for s in named))
    out.append(("everything else (small layer norms, adds, copies, sorts, fills, torch glue)", t, n, None, "launch-bound: %.0f launches of ~5 us" % n))
    tot = sum(o[1] for o in out)
    print(f"# one-stream step, kernel time {tot:.1f} ms ({stats}, {steps:g} steps)")
    print("# family | ms/step | launches/step | floor ms/step | x floor | floor =")
    for name, t, n, fl, why in out:
        print(f"{name} | {t:.2f} | {n:.0f} | {'' if fl is None else f'{fl:.2f}'} | {'' if fl is None else f'{t / fl:.2f}'} | {why}")


if __name__ == "__main__":
    main()
