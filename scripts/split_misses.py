"""which dense launches of one step run without a cached pre-split weight image"""
import sys, os, collections
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
miss = collections.Counter(); hit = collections.Counter()
orig = ops._static_split
def probe(Bm, N_, K_, ld):
    r = orig(Bm, N_, K_, ld)
    (hit if r is not None else miss)[(tuple(Bm.shape), type(Bm).__name__, Bm._base is not None)] += 1
    return r
ops._static_split = probe
sum(model.forward_losses(ops.normalize_pad(frames), gt).values()); torch.cuda.synchronize()
print("hits", sum(hit.values()), "misses", sum(miss.values()))
for k, v in miss.most_common(12): print("miss", k, v)
