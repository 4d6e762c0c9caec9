"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<out>.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE -o run -- python3 bench.py ... (same)
    python scripts/pmc_traffic.py gpurun_out/pmc_FETCH gpurun_out/pmc_WRITE profiles/r2_pmc_traffic.json 3.5 <commit>

Counter unit is KB.  FETCH_SIZE is doubled for the "corrected" figures (MI355X_MICROARCH.md: it under-reports wide
coalesced reads 2x on gfx950); WRITE_SIZE is used as is.  Launch counts include everything bench.py runs: the loss-free
calibration forward (0.5 step), the warmup step, the 2 timed steps and, with the default two-stream schedule, the two
re-check steps of the bitwise comparison: 5.5 step-equivalents (3.5 with --no-overlap; pass the figure as 4th argument)."""
import collections, csv, glob, json, sys

STEP_EQUIV = float(sys.argv[4]) if len(sys.argv) > 4 else 5.5
COMMIT = sys.argv[5] if len(sys.argv) > 5 else "unrecorded"


def family(name):
    if "gemm" in name or "conv3x3" in name or "conv7x7" in name or "ffn_f16x3" in name: return "gemm"      # the dense family (bench.py's roofline)
    if "msda" in name: return "msda"
    if "matcher_cost" in name: return "matcher_cost"
    if any(k in name for k in ("hist_", "accumulate", "gather_rows", "select_kernel", "loss_finalize", "row_prep", "row_list")): return "loss"
    return "other"


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            fam = family(r["Kernel_Name"])
            agg[fam] += float(r["Counter_Value"]) * 1024.0
            n[fam] += 1
    return agg, n


fetch, n = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --one-stream --no-other-schedule --no-cpu-baseline "
                  "--no-train-step --no-keymask --no-kernel-events (separate passes)", "commit": COMMIT, "step_equivalents": STEP_EQUIV,
       "note": __doc__.split("\n\n")[-1].replace("\n", " "), "per_kernel_family": {}}
tot_raw = tot_cor = 0.0
for fam in sorted(fetch, key=lambda k: -fetch[k]):
    raw = (fetch[fam] + write[fam]) / STEP_EQUIV
    cor = (2 * fetch[fam] + write[fam]) / STEP_EQUIV
    tot_raw += raw; tot_cor += cor
    out["per_kernel_family"][fam] = {"launches": n[fam], "fetch_GB_raw": round(fetch[fam] / 1e9, 2), "write_GB": round(write[fam] / 1e9, 2),
                                     "per_step_GB_raw": round(raw / 1e9, 2), "per_step_GB_corrected": round(cor / 1e9, 2),
                                     "per_launch_bytes_corrected": int((2 * fetch[fam] + write[fam]) / max(n[fam], 1))}
out["whole_step"] = {"per_step_GB_raw": round(tot_raw / 1e9, 1), "per_step_GB_corrected": round(tot_cor / 1e9, 1), "algorithmic_GB_per_step_SURVEY_8d": 304}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["whole_step"]), {k: v["per_step_GB_corrected"] for k, v in out["per_kernel_family"].items()})
