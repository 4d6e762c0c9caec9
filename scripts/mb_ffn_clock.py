"""In-kernel clock of the one-launch encoder FFN: the diagnostic build S2D_FFN_DBG=16 (scripts/build_ffn_dbg.sh 16)
stamps s_memtime / s_memrealtime around the chunk loop of every workgroup's wave 0; after >= 2 s of back-to-back launches on random
data the clock is median(d memtime / d memrealtime) x 100 MHz.  Prints the clock, the chunk loop's cycles per workgroup, and what the
MFMA pipe share is AT THAT CLOCK (96 MFMAs x 32 cycles per chunk and wave = one SIMD's matrix pipe).
    S2D_HIP_LIB=s2d_amd/csrc/libs2d_hip_dbg16.so python scripts/mb_ffn_clock.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
from s2d_amd._lib import LIBPATH
M = 309120
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, 256), device=dev, generator=g)
W1 = torch.nn.Parameter(torch.randn((1024, 256), device=dev, generator=g) * 0.06)
W2 = torch.nn.Parameter(torch.randn((256, 1024), device=dev, generator=g) * 0.03)
b1 = torch.randn((1024,), device=dev, generator=g) * 0.1
b2 = torch.randn((256,), device=dev, generator=g) * 0.1
raw = ctypes.CDLL(LIBPATH)
g1, be1, g2, be2 = (torch.randn((256,), device=dev, generator=g) * 0.1 + (1 if i % 2 == 0 else 0) for i in range(4))
S = M // 16
Wp = ops.mark_static(torch.randn((544, 256), device=dev, generator=g) * 0.05)
bp = torch.randn((544,), device=dev, generator=g) * 0.1
bp[:288] = 0
pos = torch.randn((S, 288), device=dev, generator=g) * 0.5
Wo = torch.nn.Parameter(torch.randn((256, 256), device=dev, generator=g) * 0.07)
bo = torch.randn((256,), device=dev, generator=g) * 0.1
samp = torch.randn((M, 256), device=dev, generator=g)
NWG = (M + 127) // 128
ONLY_FULL = os.environ.get("MB_ONLY_FULL", "") == "1"      # ablation builds: the encoder layer's launch at p = 0.3 only
for name, mk in (("FFN only", lambda p: (lambda: ops.ffn_fused(x, W1, b1, W2, b2, dropout=(p, 7, 1, 2) if p > 0 else None))),
                 ("out_proj + LN1 + FFN + LN2 + next projection (the encoder layer's launch)",
                  lambda p: (lambda: ops.ffn_fused(samp, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=(p, 7, 1, 2) if p > 0 else None,
                                                   post=(Wp, bp, pos), pre=(Wo, bo, x, 0))))):
    for p in (0.0, 0.3):
        if ONLY_FULL and (name == "FFN only" or p == 0.0):
            continue
        fn = mk(p)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < (1.0 if ONLY_FULL else 2.5):
            for _ in range(50): fn()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        buf = np.zeros(16 * 4096, np.uint64)
        assert raw.s2d_ffn_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        st = buf.reshape(4096, 16)[: min(4096, NWG)].astype(np.int64)
        dt, dr = st[:, 3] - st[:, 1], st[:, 4] - st[:, 2]
        ok = dr > 0
        clk = np.median(dt[ok] / dr[ok]) * 100.0            # MHz
        cyc = float(np.median(dt[ok]))
        mfma_cyc = 32 * 96 * 32                             # chunks x MFMAs x cycles per wave (one SIMD's matrix pipe)
        med = lambda a: float(np.median(a[ok])) / clk       # us
        print(f"{name}, p={p}: {ms:.4f} ms per launch; in-kernel clock {clk:.0f} MHz (median over {int(ok.sum())} workgroups; 10/90 percentiles "
              f"{np.percentile(dt[ok] / dr[ok], 10) * 100:.0f} / {np.percentile(dt[ok] / dr[ok], 90) * 100:.0f}); per workgroup (median, us): "
              f"entry -> chunk loop {med(st[:, 1] - st[:, 0]):.1f} | chunk loop {cyc / clk:.1f} ({cyc:.0f} cycles, MFMA pipe busy {mfma_cyc / cyc:.3f}) | "
              f"trailing pass + epilogue {med(st[:, 5] - st[:, 3]):.1f} | projection phase + store drain {med(st[:, 6] - st[:, 5]):.1f} | "
              f"whole workgroup {med(st[:, 6] - st[:, 0]):.1f}; rounds of workgroups {-(-NWG // 256)} (x whole = {-(-NWG // 256) * med(st[:, 6] - st[:, 0]) / 1000:.3f} ms)", flush=True)
        if name != "FFN only" and st[:, 7].any():
            print(f"    fine (us): entry -> tiles landed + fragments {med(st[:, 7] - st[:, 0]):.1f} | out_proj phase {med(st[:, 8] - st[:, 7]):.1f} | LN1 + fragments "
                  f"{med(st[:, 9] - st[:, 8]):.1f} | -> loop start {med(st[:, 1] - st[:, 9]):.1f} || loop end -> trailing pass done {med(st[:, 10] - st[:, 3]):.1f} | epilogue tile "
                  f"loop {med(st[:, 11] - st[:, 10]):.1f} | LN2 + Y stores issued {med(st[:, 5] - st[:, 11]):.1f} || projection: -> part 4 {med(st[:, 13] - st[:, 5]):.1f} | 4..8 "
                  f"{med(st[:, 14] - st[:, 13]):.1f} | 8..12 {med(st[:, 15] - st[:, 14]):.1f} | 12..end + drain {med(st[:, 6] - st[:, 15]):.1f}", flush=True)
