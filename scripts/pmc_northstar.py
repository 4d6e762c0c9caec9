"""Join the rocprofv3 passes of scripts/mb_northstar_kernels.py into profiles/<out>.json (per kernel: average launch time, MFMA-busy
fraction, HBM bytes):

    rocprofv3 --kernel-trace --stats --output-format csv -d D/trace -o t -- python3 scripts/mb_northstar_kernels.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d D/mfma -o t -- python3 scripts/mb_northstar_kernels.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d D/fetch -o t -- python3 scripts/mb_northstar_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d D/write -o t -- python3 scripts/mb_northstar_kernels.py
    python scripts/pmc_northstar.py D profiles/r3_pmc_northstar.json <commit>

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the counter adds every matrix instruction's pipe cycles
over all SIMDs, GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, constants table / DVFS note).
FETCH_SIZE (KB) is doubled as that guide prescribes for gfx950; WRITE_SIZE is used as is."""
import collections, csv, glob, json, sys

D, OUT = sys.argv[1], sys.argv[2]
COMMIT = sys.argv[3] if len(sys.argv) > 3 else "unrecorded"
PROGRAM = sys.argv[4] if len(sys.argv) > 4 else "scripts/mb_northstar_kernels.py (c4 shapes, kernels alone)"
# name substring [, grid sizes that single the launch out when the program is bench.py itself: the mask-logit einsum is the only dense
# launch with 2 clips x 3 680 row tiles x 2 column tiles of 256 threads (471 040 x 100 x 256 per clip on the 128 x 64 kernel)]
KERNELS = {"msda_gather": ("msda_fused", None), "mask_einsum": ("gemm_f16x3", {2 * 3680 * 2 * 256, 2 * 3680 * 2}), "cross_attn": ("cross_attn_kernel", None)}


def counters(sub):
    """{(kernel name, grid size): {counter: [values]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{D}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], int(float(r.get("Grid_Size", 0) or 0)))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def pick(agg, spec):
    sub, grids = spec
    rows = {k: v for k, v in agg.items() if sub in k[0]}
    if grids and any(k[1] in grids for k in rows):
        rows = {k: v for k, v in rows.items() if k[1] in grids}
    if not rows:
        return {}
    k = max(rows, key=lambda k: sum(len(x) for x in rows[k].values()))
    return rows[k]


stats = {}
for f in glob.glob(f"{D}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
mf, fe, wr = counters("mfma"), counters("fetch"), counters("write")
out = {"commit": COMMIT, "program": PROGRAM, "note": __doc__.split("\n\n")[-1].replace("\n", " "),
       "kernels": {}}
for name, spec in KERNELS.items():
    sub = spec
    st = [(k, v) for k, v in stats.items() if spec[0] in k]
    ent = {}
    if st:
        k, (avg, calls) = max(st, key=lambda kv: kv[1][1] * kv[1][0])
        ent.update(kernel=k[:120])
        if spec[1] is None or "bench.py" not in PROGRAM:      # the stats CSV has no grid column: a name shared with other launches has no average of its own
            ent.update(avg_launch_us=round(avg / 1e3, 2), launches_in_trace=calls)
    c = pick(mf, sub)
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        busy, gui = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"])
        ent["mfma_busy_frac"] = round(busy / (1024.0 * gui / 8.0), 4)
        ent["gui_active_cycles_per_launch"] = round(gui / 8.0 / len(c["GRBM_GUI_ACTIVE"]))
    f_, w_ = pick(fe, sub).get("FETCH_SIZE"), pick(wr, sub).get("WRITE_SIZE")
    if f_ and w_:
        ent["hbm_bytes_per_launch_corrected"] = int(2 * 1024 * sum(f_) / len(f_) + 1024 * sum(w_) / len(w_))
        ent["fetch_bytes_per_launch_raw"] = int(1024 * sum(f_) / len(f_))
        ent["write_bytes_per_launch"] = int(1024 * sum(w_) / len(w_))
    out["kernels"][name] = ent
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
