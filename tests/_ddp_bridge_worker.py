"""Worker of tests/test_gpu_config5.py::test_ddp_wrapped_training_step_through_the_grad_bridge (one process per rank, started
by torch.distributed.run; both ranks use cuda:0, backend gloo).  Mirrors the reference's wrapping and step:
engine/defaults.py:76-85 (`DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False)`) and
engine/train_loop.py:709-726 (`loss_dict = self.model(data)`; `losses = sum(loss_dict.values())`; `losses.backward()`)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from s2d_amd.modeling import build_kd_model
    from tests.test_gpu_config5 import _batch
    from tests.parity import seeded_load
    dev = torch.device("cuda:0")
    Q, T, P = 16, 2, 256
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(2.0, 5.0, 5.0), dropout=0.0)
    seeded_load(model.student, 3); seeded_load(model.teacher, 4)
    model = model.to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False)
    data = _batch(20 + rank, T, 64, 96, 3)                 # every rank its own clip (data_video/build.py:21-35 shards by rank)
    params = [p for p in model.student.parameters() if p.requires_grad]

    def step(sync):
        for p in params:
            p.grad = None
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        if sync:
            loss_dict = ddp(data)
            losses = sum(loss_dict.values())
            losses.backward()
        else:
            with ddp.no_sync():
                loss_dict = ddp(data)
                losses = sum(loss_dict.values())
                losses.backward()
        torch.cuda.synchronize()
        return float(losses), [p.grad.detach().clone() for p in params]

    l_local, g_local = step(False)                          # this rank's own gradient, no exchange
    l_ddp, g_ddp = step(True)                               # DDP's bucket hooks all-reduce (mean) what the bridge hands to AccumulateGrad
    assert l_local == l_ddp
    worst = 0.0
    for gl, gd in zip(g_local, g_ddp):
        mean = gl.clone()
        dist.all_reduce(mean)
        mean /= world
        scale = float(mean.abs().max()) + 1e-30
        worst = max(worst, float((mean - gd).abs().max()) / scale)
    # every rank saw the same reduced gradient
    chk = torch.stack([g.double().sum() for g in g_ddp])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "ranks disagree on the reduced gradients"
    # every gradient kernel is bitwise reproducible (the point-loss scatter accumulates in int32 fixed point, the MSDeformAttn backward
    # is the sorted atomic-free form), so the two evaluations hand DDP the same bits; what is left is DDP's own order of operations
    # (divide by the world size, then sum) against sum-then-divide here: a rounding of the last bit, exact for world = 2
    assert worst < 1e-6, worst
    n_nonzero = sum(int(g.abs().max() > 0) for g in g_ddp)
    dist.barrier()
    if rank == 0:
        print(f"DDP_BRIDGE_OK world={world} params={len(params)} nonzero={n_nonzero} worst_rel_dev_from_mean={worst:.3e} loss={l_ddp:.5f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
