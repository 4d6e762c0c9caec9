"""the video decoder's forward (~750 launches on 200 query rows) at c4: eager launches vs one hipGraph replay (torch.cuda.CUDAGraph
captures the library's launches: they go to torch's current stream and allocate through torch's caching allocator)"""
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
    def eager():
        return head.predictor(ms, mf, True, True)
    for _ in range(3): out = eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
    print(f"eager: {te * 1e3:.2f} ms per decoder forward", flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): eager()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout = eager()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    same = torch.equal(gout.mask_logits, out.mask_logits) and torch.equal(gout.class_logits, out.class_logits)
    print(f"graph replay: {tg * 1e3:.2f} ms per decoder forward; outputs equal to eager: {same}", flush=True)
