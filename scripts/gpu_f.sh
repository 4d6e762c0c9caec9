#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_msda -o msda -- python3 scripts/mb_msda_bwd.py > gpurun_out/r2f/mb.txt 2>&1; echo "prof rc=$?"; grep -E "msda (backward|forward)" gpurun_out/r2f/mb.txt
f=$(find gpurun_out/r2f/prof_msda -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
timeout -k 10 900 python -m pytest tests/test_gpu_msda_glue.py tests/test_gpu_backward.py -q > gpurun_out/r2f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r2f/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for ck in 0 19320 38640 77280; do
  S2D_FFN_CHUNK_ROWS=$ck timeout -k 10 400 python bench.py --no-cpu-baseline --no-train-step --no-keymask --steps 8 --warmup 3 > gpurun_out/r2f/bench_ck$ck.json 2> gpurun_out/r2f/bench_ck$ck.err
  python3 -c "
import json; j=json.load(open('gpurun_out/r2f/bench_ck$ck.json')); print('chunk $ck:', j['ms_per_step'], 'other', j['schedules']['other_ms_per_step'], 'dense ms', j['roofline']['kernel_ms_per_step'], 'bitwise', j['schedules']['losses_bitwise_equal_between_schedules'])"
done
