"""one network's video decoder forward at c4, eager, a few repetitions -- to be run under `rocprofv3 --kernel-trace --output-format csv`;
scripts/trace_timeline.py then prints the device timeline of the LAST repetition (start, duration, gap to the previous kernel's end)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import build_kd_model
dev = torch.device("cuda")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0), dropout=0.3).to(dev)
model.train()
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
images = ops.normalize_pad(frames)
head = model.student[1]
with torch.no_grad():
    mf, ms = head.pixel_decoder.forward_features(model.student[0](images))
    for _ in range(3): out = head.predictor(ms, mf, True, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = head.predictor(ms, mf, True, True)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
    print(f"eager: {te * 1e3:.2f} ms per decoder forward", flush=True)
    ops.zero_marker() if hasattr(ops, "zero_marker") else None
    torch.cuda.synchronize()
    time.sleep(0.05)                       # a gap in the trace in front of the repetition that is analysed
    out = head.predictor(ms, mf, True, True)
    torch.cuda.synchronize()
