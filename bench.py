#!/usr/bin/env python3
"""bench.py -- clip-frames/s of the S2D hot path on MI355X: KDVideoMaskFormer forward + distillation loss
(student fwd + teacher fwd + GT criterion + KD targets + KD criterion), R50 M2F-Video, T=8, 720p (736x1280 padded),
Q=100, P=160000, 2 clips per GPU (BASELINE.json configs[3] shapes; forward+loss as BASELINE.json:metric says), encoder
dropout 0.3 as every shipped config sets it (MODEL.MASK_FORMER.DROPOUT).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Launched by `python -m torch.distributed.run ... bench.py --gpus N ...` the ranks read
RANK / LOCAL_RANK / WORLD_SIZE; launched plainly with --gpus N > 1 this script starts those N ranks itself (child
processes, before this process touches a GPU) and exits with their status.

One "step" = one pass of the hot path over one batch (2 clips x 8 frames per GPU) of synthetic input that is already
resident in HBM.  Clips are independent units: ranks shard clips, forward+loss needs no collective (scaling: weak).
Rank 0 prints ONE JSON line: the metric, `roofline` (dense kernels, live HIP events), `hbm_roofline` (whole step),
`train_step` (one full training iteration incl. the RCCL gradient all-reduce at N > 1), `keymask` (BASELINE configs[2]
kernel set, N = 1), `cpu_baseline` (N = 1).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (B clips/GPU, T, H0, W0, Q, P, N gt instances/clip)
    "c4": (2, 8, 720, 1280, 100, 160000, 10),   # the metric's config: T=8 720p Q=100
    "c2": (1, 2, 480, 854, 100, 12544, 10),      # BASELINE configs[1]
    "tiny": (1, 2, 64, 96, 16, 256, 3),
}
# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters: dense MFMA peaks
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "bf16x3": 2500.0}
DENSE_DESC = {"f32": "fp32-input MFMA (v_mfma_f32_32x32x2_f32)",
              "f16x3": "split-fp16 x3 on v_mfma_f32_32x32x16_f16 (fp32-class accuracy: 3 MFMA flops per algorithmic flop)",
              "bf16x3": "split-bf16 x3 on v_mfma_f32_32x32x16_bf16 (3 MFMA flops per algorithmic flop)"}


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks as children (nothing in this process has touched the GPU yet)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def train_step_report(model, frames, masks, mean, std, dev, world, cdev, iters=3):
    """one FULL training iteration per rank on its batch: forward + loss (student + teacher, GT + KD) + the student's backward
    on the HIP gradient kernels + SUM all-reduce of the flat gradient arena over the ranks (RCCL at N > 1; the mean is folded
    into the optimizer's inv_scale) + full-model clip + AdamW + EMA teacher update.  Time = max over ranks."""
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet
    from s2d_amd.optim import FullModelGradientClippingAdamW, param_groups_like_reference
    # the iteration's forward on two streams (teacher forward beside the student's, GT criterion + its point-loss backward beside the KD
    # pass's; tests/test_gpu_fullsize.py: same losses and gradients bit for bit); S2D_TRAIN_ONE_STREAM=1 times the one-stream form
    model.overlap_teacher = model.overlap_criteria = os.environ.get("S2D_TRAIN_ONE_STREAM", "0") != "1"
    model.last = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    groups = param_groups_like_reference(model.student, 1e-4, 0.05)
    teach = dict(zip((id(p) for p in model.student.parameters()), model.teacher.parameters()))
    opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01, ema_params=[teach[id(g["params"][0])] for g in groups])
    from s2d_amd.optim import OverlappedAllReduce, student_parts
    # The gradient exchange: ONE SUM all-reduce of the arena behind the backward is the default (optim.allreduce_grads); the
    # per-part exchange started while the backward still runs (OverlappedAllReduce) is opt-in (S2D_OVERLAP_ALLREDUCE=1) until it has
    # run on real multi-GPU hardware -- at N > 1 this report times it too and compares its reduced arena with the one-shot exchange's.
    overlap_default = os.environ.get("S2D_OVERLAP_ALLREDUCE", "0") == "1"
    exchange = OverlappedAllReduce(opt, student_parts(model))
    losses, times, t_ar = [], [], []
    WARM = 4                                            # the caching allocator settles over the first iterations (0 hipMalloc from the fifth on)
    st0 = torch.cuda.memory_stats()

    def iteration(overlap, step=True):
        _fence(world)
        t0 = time.perf_counter()
        images = ops.normalize_pad(frames, 32, mean, std)
        targets = TargetSet.from_list(masks, device=dev)
        opt.zero_grad()
        if overlap:
            out = model.forward_backward(images, targets, grad_ready=exchange.ready)     # buckets go out while the backward still runs
            torch.cuda.synchronize(); t1 = time.perf_counter()
            inv = exchange.finish()
        else:
            out = model.forward_backward(images, targets)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            inv = opt.allreduce_grads()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if step:
            opt.step(inv_scale=inv, ema_momentum=0.999)
        tot = float(sum(out.values()))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t2 - t1, round(tot, 4)

    for i in range(iters + WARM):
        if i == WARM:
            st0 = torch.cuda.memory_stats()
        dt, dar, tot = iteration(overlap_default)
        times.append(dt); t_ar.append(dar); losses.append(tot)
    st1 = torch.cuda.memory_stats()
    other = None
    # opt-in (S2D_BENCH_EXCHANGE_CHECK=1): a collective that misbehaves would take the metric line down with it, and the overlapped
    # exchange has never run over RCCL (tests/test_gpu_multi.py runs it wherever two GPUs are visible)
    if world > 1 and os.environ.get("S2D_BENCH_EXCHANGE_CHECK", "0") == "1":
        # the other exchange mode, timed the same way, and both modes on the SAME batch and seeds without stepping: the reduced
        # arenas must agree (bitwise at 2 ranks; to summation order beyond, since a ring reduces a sub-range in another rank order)
        t_o = [iteration(not overlap_default)[0] for _ in range(iters)]
        arenas = []
        for ov in (False, True):
            model.criterion.seed = 0; model.criterion.matcher.seed = 0
            torch.manual_seed(5); ops._DROP_CALLS[0] = 0
            iteration(ov, step=False)
            arenas.append(opt.grad_arena.clone())
        d = float((arenas[0] - arenas[1]).abs().max() / (arenas[0].abs().max() + 1e-30))
        other = {"mode": "overlapped per part" if not overlap_default else "one all-reduce behind the backward",
                 "ms_per_iteration": round(1000 * _max_over_ranks(sum(t_o) / iters, world, cdev or dev), 1),
                 "reduced_arena_max_rel_diff_between_modes": d, "bitwise_equal": bool(torch.equal(arenas[0], arenas[1]))}
        del arenas
    assert all(map(lambda v: v == v and abs(v) != float("inf"), losses)) and not opt.found_inf()
    ms = 1000 * _max_over_ranks(sum(times[WARM:]) / iters, world, cdev or dev)
    ar = 1000 * _max_over_ranks(sum(t_ar[WARM:]) / iters, world, cdev or dev)
    return {"what": "one full training iteration per rank on its batch: fwd + loss (student + teacher, GT + KD) + backward of the student "
                    "(HIP gradient kernels, no autograd graph) + gradient all-reduce ("
                    + ("per part of the student, started while the backward of the remaining parts runs; allreduce_ms = what is left to wait for after the backward"
                       if overlap_default else "one SUM all-reduce of the arena behind the backward") + ") + full-model clip + AdamW + EMA teacher update; fp32, "
                    + ("teacher forward and GT criterion on a second stream" if model.overlap_teacher else "one stream"),
            "ms_per_iteration": round(ms, 1), "clip_frames_per_s": round(world * frames.shape[0] / (ms / 1000), 2),
            "allreduce_ms": round(ar, 2), "allreduce_bytes": int(opt.grad_arena.numel() * 4), "n_gpus": world,
            "iterations": iters, "warmup_iterations": WARM, "loss_per_iteration_rank0": losses,
            "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "ms_each_rank0": [round(1000 * t, 1) for t in times],
            "allocator_in_timed_iterations": {k: st1.get(k, 0) - st0.get(k, 0) for k in
                                              ("num_alloc_retries", "num_device_alloc", "num_device_free")},
            "other_exchange_mode": other}


def synth_batch(rank, B, T, H0, W0, N, device):
    """seeded synthetic clips: smooth-noise frames + sparse moving ellipses, generated on the device"""
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    coarse = torch.rand((B * T, 3, H0 // 8 + 2, W0 // 8 + 2), generator=g, device=device)
    frames = torch.nn.functional.interpolate(coarse, size=(H0, W0), mode="bilinear", align_corners=False)
    frames = (frames * 255).clamp(0, 255).to(torch.uint8).contiguous()
    Hp, Wp = (H0 + 31) // 32 * 32, (W0 + 31) // 32 * 32
    yy = torch.arange(Hp, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(Wp, device=device, dtype=torch.float32)[None, :]
    rng = np.random.default_rng(1234 + rank)
    masks = []
    for b in range(B):
        m = torch.zeros((N, T, Hp, Wp), dtype=torch.uint8, device=device)
        for i in range(N):
            cy, cx = rng.uniform(0.2 * H0, 0.8 * H0), rng.uniform(0.2 * W0, 0.8 * W0)
            ry, rx = rng.uniform(24, 160) * H0 / 720, rng.uniform(24, 160) * H0 / 720
            present = rng.random(T) >= 0.5                      # sparse: ~50 % of frames annotated (DropLoss)
            present[rng.integers(T)] = True
            for t in range(T):
                cy += rng.uniform(-8, 8); cx += rng.uniform(-8, 8)
                if present[t]:
                    e = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
                    e[H0:, :] = False; e[:, W0:] = False
                    m[i, t] = e
        masks.append(m)
    return frames, masks


def calibrate_teacher(model, images, want=10):
    """random-init teachers score ~half the queries above 0.75; shift the class bias so ~`want` queries per clip pass
    the distillation threshold ('teacher class logits biased so ~10 queries pass')."""
    was = model.teacher.training
    model.teacher.eval()                                 # calibrate on the dropout-free forward
    out = model.teacher(images, True)
    model.teacher.train(was)
    d = (out.class_logits[-1][..., 0] - out.class_logits[-1][..., 1]).flatten().sort(descending=True).values
    B = out.class_logits.shape[1]
    thr = float(d[min(want * B, d.numel() - 1)])
    need = float(np.log(0.75 / 0.25))
    with torch.no_grad():
        bias = model.teacher[1].predictor.class_embed.bias
        bias[0] += (need - thr) / 2
        bias[1] -= (need - thr) / 2


def _usable_cores():
    """host cores this process may really run on: the affinity mask, capped by the cgroup's CPU quota (a GPU box gives a 1-GPU job
    a 16-core share of a 256-thread host: sizing thread pools by os.cpu_count() oversubscribes 16x)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return max(n, 1)


def _oracle_sample(cfg_name, Ts):
    """seconds PER STAGE the CPU oracle needs for one Ts-frame clip at the workload's resolution (BASELINE.md section 4 / SURVEY 8d's
    stage list): normalise + pad, R50 trunk, pixel decoder, video decoder (cross-frame attention over the Ts frames included),
    prepare_targets, matcher (cost matrices + LSAP) and loss_labels + loss_masks over the 10 prediction layers, KD target preparation"""
    from oracle import oracle_np as O
    from s2d_amd.utils import synth
    from s2d_amd.utils.seeded import seeded_state
    from tests.test_oracle import pixel_decoder_shapes, video_decoder_shapes
    B, T, H0, W0, Q, P, N = CONFIGS[cfg_name]
    p = seeded_state([("0." + k, s) for k, s in O.r50_param_shapes()] +
                     [("1.pixel_decoder." + k, s) for k, s in pixel_decoder_shapes()] +
                     [("1.predictor." + k, s) for k, s in video_decoder_shapes(Q)], 0)
    fr = synth.smooth_frames_u8(0, 1, Ts, H0, W0)
    m, ids = synth.ellipse_targets(0, 2, N, Ts, H0, W0, sparse=0.0)
    st = {}
    clock = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        st[name] = st.get(name, 0.0) + now - clock[0]
        clock[0] = now

    x = O.normalize_pad(fr); lap("normalize_pad")
    feats = O.resnet50(p, x, "0."); lap("r50_trunk")
    mf, ms = O.pixel_decoder(p, feats, "1.pixel_decoder."); lap("pixel_decoder")
    logits, masks = O.video_decoder(p, ms, mf, Ts, "1.predictor."); lap("video_decoder")
    tg = [O.prepare_targets(m, ids, x.shape[2], x.shape[3])[0]]; lap("prepare_targets")
    rng = np.random.default_rng(0)
    NL = logits.shape[0]
    num_masks = float(max(tg[0].shape[0], 1))
    for layer in range(NL):
        coords = [rng.random((1, P, 2), dtype=np.float32)]
        clock[0] = time.perf_counter()
        idx = O.matcher(logits[layer], masks[layer], tg, coords, 0.0, 5.0, 5.0); lap("matcher_incl_lsap")
        if layer == NL - 1:
            O.loss_labels(logits[layer], idx)
        O.loss_masks(masks[layer], tg, idx, num_masks, P=P, rng=rng); lap("loss_labels_masks")
    clock[0] = time.perf_counter()
    O.kd_targets(logits[-1][0], masks[-1][0], x.shape[2], x.shape[3]); lap("kd_target_prep")
    return st


def _warm_oracle(cores, threadpool_limits):
    from oracle import oracle_np as O
    O.lib().orc_set_threads(cores)
    if threadpool_limits is None:
        _oracle_sample("tiny", 1)
    else:
        with threadpool_limits(limits=cores):
            _oracle_sample("tiny", 1)


def cpu_baseline(cfg_name):
    """the CPU oracle (`kind: "port"`: oracle/oracle_np.py, numpy + the C kernels of oracle/s2d_oracle.c) timed on a bounded sample of
    the same workload, rank 0, N = 1 only, at two thread counts as SURVEY.md 8d asks: ALL USABLE host cores (BLAS threads for the
    contractions, OpenMP threads for the oracle's C loops: MSDeformAttn gather, point sampling, bilinear resize -- sized by the
    process's affinity / cgroup share, not os.cpu_count()) on one 2-frame clip, and ONE thread on one 1-frame clip, each with its
    per-stage seconds.  Both are extrapolated, and say so: the KD step also runs the teacher forward and the KD criterion (x2)."""
    from oracle import oracle_np as O
    B, T, H0, W0, Q, P, N = CONFIGS[cfg_name]
    cores = _usable_cores()
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None

    def run(nthreads, Ts):
        omp = O.lib().orc_set_threads(nthreads)
        if threadpool_limits is None:
            return _oracle_sample(cfg_name, Ts), omp
        with threadpool_limits(limits=nthreads):
            return _oracle_sample(cfg_name, Ts), omp

    _warm_oracle(cores, threadpool_limits)      # thread-pool start-up and library loads are not part of any stage
    # one REAL clip of the workload (all T frames: the cross-frame attention over T * hw keys and the T-frame matcher / point loss are
    # timed, not extrapolated from a shorter clip); S2D_CPU_BASELINE_FRAMES shortens it for quick runs
    Ts = max(1, min(T, int(os.environ.get("S2D_CPU_BASELINE_FRAMES", T))))
    st, omp = run(cores, Ts)
    dt = sum(st.values())
    out = {"value": round(Ts / (2.0 * dt), 5), "unit": "clip-frames/s", "cores": cores, "kind": "port",
           "sample": f"oracle port: one {Ts}-frame clip {H0}x{W0} of the workload (Q={Q}, P={P}, N={N}): student fwd + 10-layer GT criterion + KD "
                     f"target prep took {dt:.1f}s on {cores} threads (BLAS and OpenMP {omp}; usable cores of {os.cpu_count()} host threads; python "
                     f"loops between the kernels single-threaded); the KD step runs the same network and criterion twice (teacher fwd + KD "
                     f"criterion: same code, same shapes) -> frames/s = {Ts}/(2*{dt:.1f})",
           "stage_seconds": {k: round(v, 3) for k, v in st.items()}}
    if cores > 1:
        st1, _ = run(1, 1)
        dt1 = sum(st1.values())
        out["single_thread"] = {"value": round(1 / (2.0 * dt1), 5), "unit": "clip-frames/s", "cores": 1,
                                "sample": f"oracle port, extrapolated x2: one 1-frame clip {H0}x{W0}, same stages, {dt1:.1f}s on 1 thread -> 1/(2*{dt1:.1f})",
                                "stage_seconds": {k: round(v, 3) for k, v in st1.items()}}
        O.lib().orc_set_threads(cores)
    return out


def keymask_report(dev, cpu=True):
    """BASELINE configs[2]: the keymask propagate-and-match kernel set (K2..K6) on a 32-frame 480p synthetic clip with 6 objects
    and 2 500 tracked points, and the K1 local-correlation kernel on a [32,120,214,128] feature map.
    HBM-bound integer/byte kernels: achieved GB/s of ALGORITHMIC bytes against the 8 TB/s peak, HIP events on the launch
    stream, the oracle's CPU time on the same arrays beside it."""
    from s2d_amd import keymask as km
    from s2d_amd.utils import synth
    T, H, W, Np, NOBJ = 32, 480, 854, 2500, 6
    m, _ = synth.ellipse_targets(12, 1, NOBJ, T, H, W, sparse=0.0, rmin=30, rmax=90)
    idm = np.zeros((T, H, W), np.int64)
    for o in range(NOBJ):
        idm[m[o] > 0] = o + 1
    rng = synth.rng_for(12, 2)
    ys, xs = np.nonzero(m[0, 0])
    sel = rng.integers(0, len(ys), Np)
    tracks = np.stack([np.stack([xs[sel], ys[sel]], -1).astype(np.float32) + rng.normal(0, 1, (Np, 2)).astype(np.float32) + 0.6 * t
                       for t in range(T)])[None]
    vis = rng.random((1, T, Np)) > 0.3
    tr_d = torch.from_numpy(tracks).to(dev)
    vis_d = torch.from_numpy(vis).to(dev)
    idmap = km.IdMap(torch.from_numpy(idm))
    Hf, Wf, C, r = H // 4, (W + 3) // 4, 128, 3
    fmap = torch.randn((T, Hf, Wf, C), device=dev)
    coords = (torch.rand((T, Np, 2), device=dev) * torch.tensor([Wf, Hf], device=dev)).contiguous()
    sup = torch.randn((Np, (2 * r + 1) ** 2, C), device=dev)
    S = (2 * r + 1) ** 2

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)       # current stream = the launch stream here
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    tm = km.pred_tracks_to_binary_masks(tr_d, H, W)[0]
    rows = {}
    ms = timed(lambda: km.visibility_curve(vis_d))
    rows["K2 visibility curve"] = (ms, T * Np * 1 + T * 4)
    ms = timed(lambda: km.pred_tracks_to_binary_masks(tr_d, H, W))
    rows["K3 tracks -> point masks"] = (ms, T * Np * 8 + T * H * W)                      # tracks read + the u8 planes written (zero fill + scatter)
    ms = timed(lambda: km.point_id_counts(tm, idmap))
    rows["K4-K6 point/mask counts (all frames x objects, one launch)"] = (ms, T * H * W * (1 + 8) + T * (NOBJ + 2) * 4)
    ms = timed(lambda: km.local_correlation(fmap, coords, sup, r), reps=5)
    rows["K1 local correlation (self-defined; parity unpinned)"] = (ms, Np * S * C * 4 + T * Np * S * S * 4 + T * Np * (2 * r + 2) ** 2 * C * 4)
    t0 = time.perf_counter()
    matches, allc = km.extract_mask_matches((H, W), tr_d, idmap, (0, T - 1))
    torch.cuda.synchronize()
    whole = 1000 * (time.perf_counter() - t0)
    out = {"workload": f"keymask_ident propagate-and-match: {T}-frame {H}x{W} clip, {NOBJ} objects, {Np} tracked points; K1 on [{T},{Hf},{Wf},{C}] features, r={r}",
           "kernels": {k: {"ms": round(v[0], 4), "algorithmic_MB": round(v[1] / 1e6, 2), "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1),
                           "frac_of_8TBps": round(v[1] / (v[0] * 1e-3) / 8e12, 4)} for k, v in rows.items()},
           "extract_mask_matches_ms_incl_one_sync_and_host_loop": round(whole, 3), "n_comparisons": len(allc), "n_matches": len(matches),
           "bound": "hbm"}
    if cpu:
        from oracle import oracle_np as O
        t0 = time.perf_counter()
        O.extract_mask_matches(tracks[0], idm, H, W, (0, T - 1))
        O.visibility_curve(vis[0])
        out["cpu_oracle_ms_K2_to_K6"] = round(1000 * (time.perf_counter() - t0), 1)
    return out


def per_kernel_report(prof, steps, dims, args):
    """north_star's per-kernel numbers: achieved HBM GB/s of the MSDeformAttn gather against its algorithmic bytes (and, from the
    committed PMC pass, against the bytes it really moved), MFMA rate / busy fraction of the mask-logit einsum and of the masked
    cross-attention.  Durations: HIP events around each launch inside this run's one-stream steps; the PMC figures come from
    profiles/r4_pmc_northstar.json (rocprofv3 --pmc passes of scripts/mb_northstar_kernels.py, stamped with their commit)."""
    B, T, Q = dims
    pmc = {}
    pp = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r5_pmc_northstar.json", "r4_pmc_northstar.json", "r3_pmc_northstar.json")) if os.path.exists(q)), "")
    if pp and args.config == "c4" and args.dense == "f16x3":
        j = json.load(open(pp))
        pmc = {k: dict(v, pmc_commit=j.get("commit"), pmc_program=j.get("program")) for k, v in j.get("kernels", {}).items()}

    def agg(sel):
        rows = [r for r in prof if sel(r[3])]
        if not rows:
            return None
        ms = sum(s.elapsed_time(e) for s, e, *_ in rows)
        return len(rows), ms, sum(r[2] for r in rows), sum(r[3][-1] for r in rows)

    out = {}
    a = agg(lambda t: t[0] == "msda")
    if a:
        n, ms, _, by = a
        e = {"launches_per_step": n // steps, "avg_launch_us": round(1000 * ms / n, 1), "bound": "hbm",
             "algorithmic_GB_per_step": round(by / steps / 1e9, 2), "GBps_algorithmic": round(by / (ms * 1e-3) / 1e9, 1),
             "frac_of_8TBps": round(by / (ms * 1e-3) / 8e12, 4)}
        p = pmc.get("msda_gather", {})
        if "hbm_bytes_per_launch_corrected" in p:
            e.update(GBps_measured=round(p["hbm_bytes_per_launch_corrected"] / (1000 * ms / n * 1e-6) / 1e9, 1),
                     hbm_bytes_per_launch_pmc=p["hbm_bytes_per_launch_corrected"], traffic_over_algorithmic=round(p["hbm_bytes_per_launch_corrected"] / (by / n), 3),
                     pmc_commit=p.get("pmc_commit"), pmc_program=p.get("pmc_program"))
        out["msda_gather"] = e
    for key, sel, desc in (("mask_einsum", lambda t: t[0] == "gemm" and t[1] == B and t[3] == Q and t[4] == 256 and t[2] > 4096, "bqc,btchw->bqthw as pixel-major GEMM"),
                           ("cross_attn", lambda t: t[0] == "xattn", "masked QK^T / softmax / AV, 3 levels")):
        a = agg(sel)
        if a:
            n, ms, fl, by = a
            tf = fl / (ms * 1e-3) / 1e12
            gbps = by / (ms * 1e-3) / 1e9
            # the einsum reads 482 MB of mask features and writes 188 MB of logits per clip for 24 GFLOP: 36 flop / byte, under the ridge of
            # the split-fp16 matrix rate -- its roofline is HBM; the cross-attention's is the matrix pipe
            hbm_bound = key == "mask_einsum"
            e = {"what": desc, "launches_per_step": n // steps, "avg_launch_us": round(1000 * ms / n, 1), "bound": "hbm" if hbm_bound else "mfma",
                 "TFLOPs_algorithmic": round(tf, 1), "frac_of_f16_peak_2500": round(tf / 2500.0, 4), "mfma_flops_per_algorithmic_flop": 3,
                 "GBps_algorithmic": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)}
            p = pmc.get(key, {})
            if "mfma_busy_frac" in p:
                e.update(mfma_busy=p["mfma_busy_frac"], pmc_commit=p.get("pmc_commit"), pmc_program=p.get("pmc_program"))
            if "hbm_bytes_per_launch_corrected" in p and hbm_bound:
                e.update(hbm_bytes_per_launch_pmc=p["hbm_bytes_per_launch_corrected"], traffic_over_algorithmic=round(p["hbm_bytes_per_launch_corrected"] / (by / n), 3))
            out[key] = e
    return out


_DIST = {"on": False}


def _fence(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(x, world, device):
    if world <= 1:
        return float(x)
    import torch.distributed as dist
    t = torch.tensor([float(x)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--dropout", type=float, default=0.3, help="encoder dropout (MODEL.MASK_FORMER.DROPOUT; 0.3 in every shipped config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true",
                    help="skip the extra report of one full training iteration (fwd + loss + backward + all-reduce + clip/AdamW/EMA, BASELINE config 4)")
    ap.add_argument("--no-keymask", action="store_true", help="skip the keymask kernel-set report (BASELINE config 3)")
    ap.add_argument("--no-amp", action="store_true", help="skip the extra `amp` report (the same step with AMP compute switched on)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--one-stream", "--no-overlap", dest="one_stream", action="store_true",
                    help="time the one-stream schedule (default: teacher forward + GT criterion on a second HIP stream; the other "
                         "schedule is timed after the metric and reported under `schedules`, with a bitwise comparison of the losses)")
    ap.add_argument("--clip-pipeline", dest="clip_pipeline", action="store_true", default=os.environ.get("S2D_CLIP_PIPELINE", "0") == "1",
                    help="process the batch clip by clip, clip b's criteria on a third stream beside clip b + 1's forwards "
                         "(KDVideoMaskFormer.pipeline_clips)")
    ap.add_argument("--no-clip-pipeline", dest="clip_pipeline", action="store_false")
    ap.add_argument("--no-other-schedule", action="store_true",
                    help="profiling runs: time only the chosen schedule (no second timing pass, no bitwise comparison), so that a kernel "
                         "trace of the process holds one schedule's launches")
    ap.add_argument("--dense-breakdown", action="store_true", help="print per-shape time of the dense launches to stderr")
    ap.add_argument("--dense", default="f16x3", choices=["f32", "f16x3", "bf16x3"],
                    help="arithmetic of the dense contractions (all three are fp32-in/fp32-out)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (spawn, rendezvous, barrier, max-over-ranks, one JSON line): no GPU "
                         "work, value null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("S2D_BENCH_BACKEND", "nccl")          # "gloo": rehearse the N > 1 control flow (CPU, or ranks sharing one GPU)
    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        if world > 1:
            dist.barrier()
        dt = _max_over_ranks(time.perf_counter() - t0, world, "cpu")
        if rank == 0:
            print(json.dumps({"metric": "clip-frames/sec fwd+loss, R50 M2F-Video T=8 720p Q=100", "value": None, "unit": "clip-frames/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_rank_seconds": round(dt, 4)}))
        if world > 1:
            dist.destroy_process_group()
        return
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    cdev = torch.device("cpu") if world > 1 and backend != "nccl" else None
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, build_kd_model
    ops.set_dense_mode(args.dense)
    B, T, H0, W0, Q, P, N = CONFIGS[args.config]
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0),
                           dropout=args.dropout).to(dev)
    model.train()            # student AND teacher in training mode, as in the reference (the teacher is never put in eval(), Appendix C)
    model.pipeline_clips = args.clip_pipeline
    frames, masks = synth_batch(rank, B, T, H0, W0, N, dev)
    gt = TargetSet.from_list(masks, device=dev)
    calibrate_teacher(model, ops.normalize_pad(frames))
    mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()

    def step():
        # everything KDVideoMaskFormer.forward does per batch on the device: normalise + pad the frames, paste the GT masks
        # into padded target planes (prepare_targets), both forwards, both criteria, the weighted sum the trainer takes
        images = ops.normalize_pad(frames, 32, mean, std)
        targets = TargetSet.from_list(masks, device=dev)
        losses = model.forward_losses(images, targets)
        return sum(losses.values())

    def seeded_step(two_streams):
        model.overlap_teacher = model.overlap_criteria = two_streams
        model.criterion.seed = 12345
        model.criterion.matcher.seed = 12345
        torch.manual_seed(777)
        ops._DROP_CALLS[0] = 0                           # same dropout masks in both schedules
        images = ops.normalize_pad(frames, 32, mean, std)
        return torch.stack(list(model.forward_losses(images, gt).values()))

    def timed(live, two_streams, steps, warmup):
        model.overlap_teacher = model.overlap_criteria = two_streams
        for _ in range(warmup):
            step()
        _fence(world)
        # step i of either schedule draws the same points and the same dropout masks (host-side counters, reset here), and its
        # loss total stays on the device: after both timed regions the totals are compared step by step (nothing is read back or
        # synchronised inside the timed region)
        model.criterion.seed = model.criterion.matcher.seed = 1000
        torch.manual_seed(777)
        ops._DROP_CALLS[0] = 0
        totals = []
        if live:
            ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(steps):
            tot = step()
            totals.append(tot)
        _fence(world)
        el = time.perf_counter() - t0
        pr, ops.PROFILE = ops.PROFILE, None
        STEP_TOTALS[two_streams] = totals
        return el, tot, pr

    STEP_TOTALS = {}

    two = not args.one_stream
    live_events = not args.no_kernel_events and not two
    dt, total, prof = timed(live_events, two, args.steps, args.warmup)
    dt = _max_over_ranks(dt, world, cdev or dev)
    assert bool(torch.isfinite(total)), "non-finite loss"
    # the other schedule, timed the same way after the metric, and a bitwise comparison of all 42 losses between the two
    # schedules on the same seeds (every kernel is deterministic, so the schedule must not change a bit)
    dt_other, prof_other, same = float("nan"), None, None
    if not args.no_other_schedule:
        dt_other, _, prof_other = timed(not args.no_kernel_events and two, not two, max(args.steps // 2, 2), 1)
        dt_other = _max_over_ranks(dt_other, world, cdev or dev)
        a = seeded_step(True)
        b = seeded_step(False)
        _fence(world)
        same = torch.equal(a, b)
        # ... and the loss totals of the timed steps themselves, step by step (the shorter region's length)
        ta, tb = STEP_TOTALS.get(True, []), STEP_TOTALS.get(False, [])
        ncmp = min(len(ta), len(tb))
        timed_same = ncmp > 0 and torch.equal(torch.stack(ta[:ncmp]), torch.stack(tb[:ncmp]))
        same = same and timed_same
        if world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1 if same else 0], device=cdev or dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = bool(flag.item())
    if prof is None or not prof:
        prof = prof_other                                 # dense-launch durations come from the one-stream steps (isolated launches)
        prof_steps = max(args.steps // 2, 2)
        events_note = ("HIP events around every dense launch of the one-stream steps timed right after the metric (in the two-stream "
                       "schedule an event pair would bracket the other stream's kernels too)")
    else:
        prof_steps = args.steps
        events_note = "HIP events around every dense launch inside the timed region"
    model.overlap_teacher = model.overlap_criteria = two

    # Extra report, never `value`: the same step with AMP COMPUTE on -- fp16-operand / f32-accumulate contractions with f32 outputs
    # (autocast-LIKE: every shipped yaml sets SOLVER.AMP.ENABLED, engine/train_loop.py:709 `with autocast():`, but real autocast also
    # rounds each output to fp16): single-pass fp16 MFMA for the modules autocast runs in fp16 (R50 trunk, video decoder linears,
    # mask-logit einsum); pixel decoder, matcher, losses as above.
    amp_res = None
    if not args.no_amp and args.dense == "f16x3":
        from s2d_amd.modeling import set_amp_compute
        set_amp_compute(model, True)
        try:
            n_amp = max(args.steps // 2, 2)
            dt_amp, tot_amp, _ = timed(False, two, n_amp, 1)
            dt_amp = _max_over_ranks(dt_amp, world, cdev or dev)
            amp_res = {"what": "same workload and schedule, AMP compute on (s2d_amd.modeling.set_amp_compute): fp16 operands / f32 accumulate / f32 outputs (autocast-like, not autocast's exact "
                               "arithmetic), one MFMA pass, in the R50 trunk, the video decoder's linear layers and the mask-logit einsum -- the modules "
                               "torch.autocast runs in fp16 in the reference; pixel decoder, matcher and losses unchanged (fp32-class)",
                       "value": round(world * B * T * n_amp / dt_amp, 3), "unit": "clip-frames/s", "ms_per_step": round(1000 * dt_amp / n_amp, 3),
                       "steps": n_amp, "loss_total_finite": bool(torch.isfinite(tot_amp)),
                       "hbm_roofline_frac": round(world * B * T * n_amp / dt_amp / world * 19.0 / 8000.0, 4) if args.config == "c4" else None}
        finally:
            set_amp_compute(model, False)

    fallback = False
    if same is False and two and dt_other == dt_other:
        # The two schedules must agree bit for bit (every kernel is deterministic).  If they do not, the two-stream timing is not
        # reported as the metric: the line carries the one-stream measurement (fewer steps, same kind) and says so.
        sys.stderr.write("bench.py: losses differ between the two-stream and the one-stream schedule -- reporting the one-stream time\n")
        fallback = True
    if rank == 0:
        frames_per_step = world * B * T
        other_ms = 1000 * dt_other / max(args.steps // 2, 2)
        if fallback:
            steps_other = max(args.steps // 2, 2)
            dt, dt_other, other_ms = dt_other * args.steps / steps_other, dt, 1000 * dt / args.steps
            two = False
        res = {"metric": "clip-frames/sec fwd+loss, R50 M2F-Video T=8 720p Q=100", "value": round(frames_per_step * args.steps / dt, 3),
               "unit": "clip-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.dense == "f32" else f"f32 ({args.dense} MFMA for dense contractions)", "data": "synthetic",
               "config": {"workload": f"KDVideoMaskFormer fwd+loss (student+teacher fwd, GT+KD VideoSetCriterion), {args.config}: "
                                      f"{B} clips/GPU x T={T} x {H0}x{W0}, Q={Q}, P={P}, N={N} sparse GT instances/clip, encoder dropout {args.dropout}",
                          "clips_per_gpu": B, "frames_per_clip": T, "parallelism": f"dp{world} (clips sharded, no collective)",
                          "streams": 2 if two else 1, "encoder_dropout": args.dropout,
                          "teacher_intermediate_masks": "full maps" if model.teacher_aux_masks else
                          "only at the pixels its own attention masks read (no loss reads them; final prediction bit-identical)",
                          "kd_targets_per_clip": model.last["kd_count"].cpu().tolist()},
               "schedules": {"timed": "two streams (teacher forward + GT criterion on a second HIP stream)" if two else "one stream",
                             "other_ms_per_step": None if args.no_other_schedule else round(other_ms, 3),
                             "other": "one stream" if two else "two streams",
                             "losses_bitwise_equal_between_schedules": None if same is None else bool(same),
                             "timed_steps_compared_bitwise": None if same is None else ncmp,
                             "fell_back_to_one_stream": fallback}}
        prof_all = prof or []
        prof = [r for r in prof_all if r[3][0] in ("gemm", "conv", "ffn")]  # the dense family (ffn: the one-launch encoder FFN, counted as its two contractions); the other tagged launches feed per_kernel
        if prof:
            ms = sum(s.elapsed_time(e) for s, e, *_ in prof)
            fl = sum(f for _, _, f, *_ in prof)
            n = len(prof)
            ach = fl / (ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dense]
            # HBM bytes per dense launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            # (scripts/pmc_traffic.py; fetch corrected x2 as MI355X_MICROARCH.md prescribes), stamped with the commit it was taken at
            traffic, traffic_src = None, None
            for name in ("r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
                tp = os.path.join(ROOT, "profiles", name)
                if args.config == "c4" and args.dense == "f16x3" and os.path.exists(tp):
                    j = json.load(open(tp))
                    traffic = j["per_kernel_family"]["gemm"]["per_launch_bytes_corrected"]
                    traffic_src = f"profiles/{name}, measured at commit {j.get('commit', 'unrecorded (round 1)')} by scripts/pmc_traffic.py (not in this run)"
                    break
            passes = 1 if args.dense == "f32" else 3
            res["roofline"] = {"bound": "mfma", "kernel": "dense NT GEMM / implicit-GEMM conv: " + DENSE_DESC[args.dense],
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": int(sum(t[-1] for *_, t in prof) / n),
                               "mfma_flops_per_algorithmic_flop": passes, "mfma_pipe_frac": round(passes * ach / peak, 4),
                               "achieved_vs_fp32_mfma_peak_157.3": round(ach / 157.3, 4),
                               # the clock the chip HOLDS inside the family's largest kernel (the encoder FFN launch), from the diagnostic
                               # build's s_memtime / s_memrealtime stamps (profiles/r5_experiments/ffn_phases.txt; not taken in this run):
                               # `peak` above is the 2.4 GHz figure of MI355X_MICROARCH.md, the matrix pipe itself runs 1.58-1.72 GHz here
                               "held_clock_MHz_in_ffn_kernel": 1661 if args.dense == "f16x3" else None,
                               "mfma_pipe_frac_at_held_clock": round(passes * ach / (peak * 1661.0 / 2400.0), 4) if args.dense == "f16x3" else None,
                               "launches_per_step": n // prof_steps, "avg_launch_us": round(1000 * ms / n, 2),
                               "kernel_ms_per_step": round(ms / prof_steps, 2),
                               "algorithmic_gflop_per_step": round(fl / prof_steps / 1e9, 1), "measured": events_note,
                               "parity_note": "R50 trunk and K1 are parity-unpinned (detectron2 / co-tracker absent from the reference tree)"}
            res["roofline"]["per_kernel"] = per_kernel_report(prof_all, prof_steps, (B, T, Q), args)
        # the north star states its target against the whole-step HBM roofline: 19.0 GB algorithmic per clip-frame
        #  at 8 TB/s
        if args.config == "c4":
            res["hbm_roofline"] = {"algorithmic_GB_per_frame": 19.0, "peak_TBps": 8.0,
                                   "frac": round(res["value"] / world * 19.0 / 8000.0, 4), "target_frac": 0.4}
            if not model.teacher_aux_masks:
                # The 19.0 GB/frame of SURVEY 8d counts ten FULL mask-logit einsums + attention-mask resizes for the teacher too.  The
                # step evaluates the teacher's intermediate heads only at the pixels its attention masks read (1/32 and 1/16 levels:
                # 4 bilinear taps per key; outputs bit-identical): those bytes are not moved, so the fraction is also given against
                # the bytes net of them.
                Hp, Wp = (H0 + 31) // 32 * 32, (W0 + 31) // 32 * 32
                hm, wm = Hp // 4, Wp // 4
                npx = T * hm * wm
                saved = 0.0
                for div in (32, 16):                        # three intermediate heads feed each level (heads 0,3,6 / 1,4,7)
                    ntap = min(4 * T * (Hp // div) * (Wp // div), npx)
                    saved += 3 * B * (4.0 * (npx - ntap) * (256 + Q)            # einsum: feature rows read + logits written
                                      + 4.0 * Q * (npx - ntap))                 # attention-mask builder: logits read
                net = 19.0 - saved / 1e9 / (B * T)
                res["hbm_roofline"].update({"skipped_teacher_mask_GB_per_step": round(saved / 1e9, 2),
                                            "algorithmic_GB_per_frame_net": round(net, 3),
                                            "frac_net_of_skipped_teacher_masks": round(res["value"] / world * net / 8000.0, 4)})
        if amp_res is not None:
            res["amp"] = amp_res
        if prof and args.dense_breakdown:
            import collections
            agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for s_, e_, f_, tag in prof:
                a_ = agg[tag[:5]]; a_[0] += 1; a_[1] += s_.elapsed_time(e_); a_[2] += f_
            for tag, (n_, t_, f_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
                print(f"{str(tag):44s} calls/step {n_/prof_steps:6.1f}  ms/step {t_/prof_steps:7.2f}  {f_/t_/1e9:7.1f} TF", file=sys.stderr)
    else:
        res = None
    if not args.no_train_step:
        # BASELINE config 4 / SURVEY.md 8d: the same batch through one FULL training iteration on every rank, gradient all-reduce
        # included.  An extra report after the metric's timed region; it never touches `value`.
        try:
            ts = train_step_report(model, frames, masks, mean, std, dev, world, cdev)
        except Exception as e:                      # the metric line must come out whatever happens here
            ts = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            res["train_step"] = ts
    if rank == 0:
        if world == 1 and not args.no_keymask:
            try:
                res["keymask"] = keymask_report(dev, cpu=not args.no_cpu_baseline)
            except Exception as e:
                res["keymask"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:     # last: its BLAS worker threads keep the host busy for a while afterwards
            res["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
