"""N>1 path on CPU: 2 gloo ranks, clip sharding, barrier + max-over-ranks timing (what bench.py does under RCCL)."""
import os
import socket

import torch
import torch.multiprocessing as mp


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
