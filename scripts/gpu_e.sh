#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/prof_msda -o msda -- python3 scripts/mb_msda_bwd.py > gpurun_out/r2e/mb.txt 2>&1; echo "prof rc=$?"; grep -E "msda (backward|forward)" gpurun_out/r2e/mb.txt
f=$(find gpurun_out/r2e/prof_msda -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -9 "$f" | cut -c1-150
timeout -k 10 1200 python -m pytest tests/test_gpu_msda_glue.py tests/test_gpu_backward.py tests/test_gpu_dropin.py tests/test_gpu_formats.py -q > gpurun_out/r2e/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r2e/pytest.log
