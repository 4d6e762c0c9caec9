#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r2d/prof_msda -o msda -- python3 scripts/mb_msda_bwd.py > gpurun_out/r2d/mb.txt 2>&1; echo "prof rc=$?"; tail -3 gpurun_out/r2d/mb.txt
f=$(find gpurun_out/r2d/prof_msda -name "*kernel_stats.csv" | head -1); echo $f; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -q -s -k config2 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "near-tie|passed|failed" gpurun_out/r2d/pytest.log | tail -4
