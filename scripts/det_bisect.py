"""which stage of the teacher forward differs run-to-run when it shares the GPU with the student on a second stream"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import build_kd_model
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
images = ops.normalize_pad(frames)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "both"

def run():
    side.wait_stream(main)
    out = {}
    with torch.cuda.stream(side):
        net = model.teacher
        feats = net[0](images)
        for k, v in feats.items():
            out["bb_" + k] = v
        mf, ms = net[1].pixel_decoder.forward_features(feats)
        out["mask_features"] = mf
        for i, m in enumerate(ms):
            out[f"ms{i}"] = m[0]
        o = net[1].predictor(ms, mf, True)
        out["mask_logits_last"] = o.mask_logits[-1]
        out["class_logits_last"] = o.class_logits[-1]
    if mode == "both":
        s = model.student(images, True)
    main.wait_stream(side)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in out.items()}

ref = run()
for rep in range(8):
    cur = run()
    bad = [k for k in ref if not torch.equal(ref[k], cur[k])]
    print(mode, "rep", rep, "differing:", bad, flush=True)
