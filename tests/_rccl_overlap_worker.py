"""Worker of tests/test_gpu_multi.py (one process per rank, one GPU per rank, backend nccl = RCCL; started by torch.distributed.run,
each child initialises ITS GPU only).  The per-part gradient exchange issued while the backward still runs
(optim.OverlappedAllReduce; the reference's DDP bucket hooks, engine/defaults.py:76-85) against ONE all-reduce of the arena behind
the backward (optim.allreduce_grads) on the same batch and seeds: the reduced arenas must be bit-identical at two ranks, then an
optimizer step from each must leave the same parameters."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    backend = os.environ.get("S2D_TEST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0      # gloo rehearsal: both ranks share cuda:0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, build_kd_model
    from s2d_amd.optim import FullModelGradientClippingAdamW, OverlappedAllReduce, param_groups_like_reference, student_parts
    from tests.parity import make_case, seeded_load
    Q, T, P = 16, 2, 256
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(2.0, 5.0, 5.0), dropout=0.0)
    seeded_load(model.student, 3); seeded_load(model.teacher, 4)
    model = model.to(dev).train()
    model.overlap_teacher = model.overlap_criteria = True                  # the schedule bench.py trains with
    frames, tg = make_case(20 + rank, 1, T, 64, 96, Q, P, (3,))            # every rank its own clip
    images = ops.normalize_pad(torch.from_numpy(frames).to(dev))
    Hp, Wp = images.shape[1:3]
    m, ids = tg[0]
    pad = torch.zeros((m.shape[0], T, Hp, Wp), dtype=torch.uint8)
    pad[:, :, :64, :96] = torch.from_numpy(m)
    gts = [pad[torch.from_numpy((ids != -1).any(-1))]]
    groups = param_groups_like_reference(model.student, 1e-4, 0.05)
    opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01)
    ex = OverlappedAllReduce(opt, student_parts(model))
    assert ex.active and (ex.async_ok or backend != "nccl")

    def grads(overlap):
        opt.zero_grad()
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        tgt = TargetSet.from_list(gts, device=dev)
        if overlap:
            model.forward_backward(images, tgt, grad_ready=ex.ready)
            inv = ex.finish()
        else:
            model.forward_backward(images, tgt)
            inv = opt.allreduce_grads()
        torch.cuda.synchronize()
        assert abs(inv - 1.0 / world) < 1e-12
        return opt.grad_arena.clone()

    one = grads(False)
    worst, nbit = 0.0, 0
    for rep in range(4):                                   # repeated: a missing stream ordering would be a race, not a constant
        ov = grads(True)
        nbit += int(torch.equal(ov, one))
        worst = max(worst, float((ov - one).abs().max() / (one.abs().max() + 1e-30)))
    local_only = None
    # ... and the exchange really happened: the reduced arena is the SUM of the two ranks' local arenas
    opt.zero_grad()
    model.criterion.seed = 0; model.criterion.matcher.seed = 0
    model.forward_backward(images, TargetSet.from_list(gts, device=dev))
    torch.cuda.synchronize()
    local_only = opt.grad_arena.clone()
    tot = local_only.clone()
    dist.all_reduce(tot)
    assert torch.equal(tot, one) if world == 2 else torch.allclose(tot, one, rtol=1e-5, atol=1e-8)
    assert not torch.equal(local_only, one)
    if world == 2:
        assert nbit == 4 and worst == 0.0, (nbit, worst)
    else:
        assert worst < 1e-6, worst
    chk = one.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "ranks disagree on the reduced gradients"
    dist.barrier()
    if rank == 0:
        print(f"RCCL_OVERLAP_OK backend={backend} async={ex.async_ok} world={world} arena_floats={one.numel()} bitwise={nbit}/4 worst_rel={worst:.3e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
