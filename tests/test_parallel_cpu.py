"""N>1 path on CPU: 2 gloo ranks, clip sharding, barrier + max-over-ranks timing (what bench.py does under RCCL)."""
import os
import socket

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s2d_amd import parallel
    w, r, _ = parallel.init("gloo")
    clips = parallel.shard_clips(5, r, w)
    parallel.fence()
    elapsed = 1.0 + 0.5 * r            # rank 1 is the straggler
    mx = parallel.max_over_ranks(elapsed)
    out.put((r, clips, mx))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    clips = [c for _, cs, _ in res for c in cs]
    assert sorted(clips) == list(range(5)) and len(res[0][1]) == 3 and len(res[1][1]) == 2   # disjoint, complete, balanced
    assert all(abs(mx - 1.5) < 1e-12 for _, _, mx in res)                                   # max over ranks
    from s2d_amd.parallel import whole_job_rate
    assert whole_job_rate(16, 5, 1.5, 2) == 16 * 2 * 5 / 1.5


def test_shard_clips_properties():
    from s2d_amd.parallel import shard_clips
    for n in (0, 1, 7, 16):
        for w in (1, 2, 4, 8):
            allc = [c for r in range(w) for c in shard_clips(n, r, w)]
            assert allc == list(range(n))
            sizes = [len(shard_clips(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def _allreduce_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    from s2d_amd.optim import allreduce_flat
    dist.init_process_group("gloo")
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    f = allreduce_flat(flat, bucket_bytes=4 * 96)          # 11 buckets, the last one partial
    q.put((rank, f, flat.clone()))
    dist.destroy_process_group()


def test_gradient_arena_allreduce_two_ranks_gloo():
    """the one exchange step of training, a bucketed SUM all-reduce of the flat gradient arena; the mean is
    folded into the optimizer's inv_scale (returned factor 1/world)"""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, f, flat in got:
        assert f == 0.5
        assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)


def _overlap_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import types
    import torch
    import torch.distributed as dist
    from s2d_amd.optim import OverlappedAllReduce
    dist.init_process_group("gloo")
    # a stand-in with the optimizer's arena bookkeeping (the real one needs device memory): 7 parameters in 3 parts, 4-aligned offsets
    sizes = [10, 3, 8, 5, 4, 9, 2]
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot); tot += (n + 3) // 4 * 4
    opt = types.SimpleNamespace(_params=params, _offs=offs, _total=tot, grad_arena=torch.arange(tot, dtype=torch.float32) * (rank + 1))
    parts = {"predictor": params[5:], "pixel_decoder": params[2:5], "backbone": params[:2]}
    ex = OverlappedAllReduce(opt, parts)
    assert ex.ranges == {"predictor": (offs[5], tot), "pixel_decoder": (offs[2], offs[5]), "backbone": (0, offs[2])}
    out = []
    for it in range(2):                                    # the exchange object is reused iteration after iteration
        opt.grad_arena.copy_(torch.arange(tot, dtype=torch.float32) * (rank + 1 + it))
        for name in ("predictor", "pixel_decoder", "backbone"):
            ex.ready(name)
        out.append((ex.finish(), opt.grad_arena.clone()))
    bad = None
    try:
        OverlappedAllReduce(opt, {"a": [params[0], params[2]], "b": [params[1]] + params[3:]})
    except ValueError as e:
        bad = str(e)
    q.put((rank, out, tot, bad))
    dist.destroy_process_group()


def test_overlapped_allreduce_per_part_two_ranks_gloo():
    """the exchange DDP overlaps with the backward (engine/defaults.py:76-85): one asynchronous all-reduce per part of the student as
    its gradients complete (predictor, pixel decoder, trunk), each a contiguous range of the arena; finish() waits and returns
    1 / world.  Same reduced arena as the one-shot exchange."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, tot, bad in got:
        for it, (f, arena) in enumerate(out):
            assert f == 0.5
            assert torch.equal(arena, torch.arange(tot, dtype=torch.float32) * (3 + 2 * it))
        assert bad and "contiguous" in bad


def test_param_groups_follow_reference_rules():
    """Trainer.build_optimizer (train_net_video.py:134-186): one group per trainable parameter, norm / embedding weight
    decay overrides, 'backbone' matched against the MODULE name"""
    import torch
    from s2d_amd.modeling import build_kd_model
    from s2d_amd.optim import param_groups_like_reference, ema_momentum_schedule
    model = build_kd_model(num_queries=8, num_frames=2, num_points=64, dec_layers=3)
    groups = param_groups_like_reference(model, 1e-4, 0.05, weight_decay_norm=0.0, weight_decay_embed=0.0, backbone_multiplier=0.1)
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert len(groups) == len(trainable) and all(len(g["params"]) == 1 for g in groups)
    by_id = {id(g["params"][0]): g for g in groups}
    dec = model.student[1].predictor
    assert by_id[id(dec.query_feat.weight)]["weight_decay"] == 0.0            # nn.Embedding
    assert by_id[id(dec.decoder_norm.weight)]["weight_decay"] == 0.0          # nn.LayerNorm
    lin = next(p for n, p in model.student[1].named_parameters() if n.endswith("linear1.weight"))
    assert by_id[id(lin)]["weight_decay"] == 0.05 and by_id[id(lin)]["lr"] == 1e-4
    # student.0.* is the backbone but the module NAME does not contain 'backbone': no multiplier, as in the reference
    bb = next(iter(model.student[0].parameters()))
    assert by_id[id(bb)]["lr"] == 1e-4
    assert not any(id(p) in by_id for p in model.teacher.parameters())        # frozen teacher
    assert abs(ema_momentum_schedule(0, 0.99, 0.9999, 1000) - 0.99) < 1e-12
    assert abs(ema_momentum_schedule(1000, 0.99, 0.9999, 1000) - 0.9999) < 1e-12


def test_bench_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself (torch.distributed.run children), they
    rendezvous on 127.0.0.1 over gloo, and rank 0 prints ONE JSON line with n_gpus == 2 (--dry-run: the control flow only, no
    GPU work -- this container has no GPU; tests/test_gpu_fullsize.py runs the real two-rank step on the GPU box)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, S2D_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["value"] is None and j["steps"] == 3
    # a launcher-provided world that contradicts --gpus is refused, not silently accepted
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
