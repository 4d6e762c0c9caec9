"""Run the two-stream forward_losses of the tiny config (under AMD_LOG_LEVEL=4 the runtime logs every AQL packet header:
barrier bit, acquire / release fence scopes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS["tiny"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
two = os.environ.get("S2D_DIAG_ONE_STREAM", "0") != "1"
model.overlap_teacher = model.overlap_criteria = two
for _ in range(2):
    out = model.forward_losses(ops.normalize_pad(frames), TargetSet.from_list(masks, device=dev))
    torch.cuda.synchronize()
print("probe done", float(sum(out.values())))
