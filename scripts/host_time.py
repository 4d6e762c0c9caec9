"""how long the host needs to enqueue one step (no synchronisation) vs the GPU time of the step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
gt = TargetSet.from_list(masks, device=dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
for two, pipe in ((False, False), (True, False), (True, True)):
    model.overlap_teacher = model.overlap_criteria = two
    model.pipeline_clips = pipe
    for _ in range(2):
        sum(model.forward_losses(ops.normalize_pad(frames), gt).values())
    torch.cuda.synchronize()
    host = []; t0 = time.perf_counter()
    for _ in range(5):
        h0 = time.perf_counter()
        tot = sum(model.forward_losses(ops.normalize_pad(frames), gt).values())
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5
    print(f"two_streams={two} clip_pipeline={pipe}: host enqueue {1e3 * sum(host) / 5:.1f} ms/step (min {1e3 * min(host):.1f}), wall {1e3 * wall:.1f} ms/step", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
tot = sum(model.forward_losses(ops.normalize_pad(frames), gt).values())
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
