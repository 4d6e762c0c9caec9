"""data-side step on the device at the shipped clip geometry: 8-frame 720p clip, 10 instances; clip augmentation (resize to
the 480-short-side bucket + flip + brightness / contrast / rotation + crop, one resampling pass per frame) and clip copy-paste"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd.data import ClipAugmentation, augment_clip, copy_and_paste_clip
dev = torch.device("cuda")
T, H0, W0, N = 8, 720, 1280, 10
g = torch.Generator(device=dev).manual_seed(0)
frames = torch.randint(0, 256, (T, 3, H0, W0), generator=g, device=dev, dtype=torch.uint8)
yy, xx = torch.meshgrid(torch.arange(H0, device=dev), torch.arange(W0, device=dev), indexing="ij")
masks = torch.stack([torch.stack([(((yy - 100 - 50 * n - 3 * t) ** 2 + (xx - 150 - 100 * n) ** 2) < (40 + 8 * n) ** 2).to(torch.uint8) for t in range(T)]) for n in range(N)])
aug = ClipAugmentation(min_size=(360, 480), random_flip="flip_by_clip", augmentations=("brightness", "contrast", "rotation"), crop=("absolute_range", (384, 600)), num_frames=T)
np.random.seed(1); random.seed(1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
P, hw = aug.sample(T, H0, W0)
dt = t(lambda: augment_clip(frames, masks, P, hw))
inb, outb = frames.numel() + masks.numel(), (3 * T + N * T) * hw[0] * hw[1]
print(f"augment_clip {T} x {H0}x{W0} -> {hw[0]}x{hw[1]}, {N} instances: {dt*1e3:.3f} ms  ({T/dt:.0f} frames/s, {(inb+outb)/dt/1e9:.0f} GB/s of bytes in + out)", flush=True)
dt = t(lambda: (aug.sample(T, H0, W0), augment_clip(frames, masks, *aug.sample(T, H0, W0))), 20)
print(f"  with the host-side parameter draw: {dt*1e3:.3f} ms", flush=True)
src_f, src_m = frames[:, :, :480, :854].contiguous(), masks[:3, :, :480, :854].contiguous()
dt = t(lambda: copy_and_paste_clip(src_f, src_m, frames, masks, rate=1.0, min_ratio=0.5, max_ratio=1.0), 10)
print(f"copy_and_paste_clip 3 source instances (480x854) onto the {H0}x{W0} clip with {N} instances: {dt*1e3:.3f} ms ({T/dt:.0f} frames/s)", flush=True)
