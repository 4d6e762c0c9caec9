#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 300 python scripts/mb_msda_bwd.py > gpurun_out/r2c/mb_msda_bwd.txt 2>&1; echo "mb rc=$?"; cat gpurun_out/r2c/mb_msda_bwd.txt | tail -4
timeout -k 10 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_msda_glue.py tests/test_gpu_backward.py tests/test_gpu_dropin.py -q > gpurun_out/r2c/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r2c/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"; python -c "
import json; j=json.load(open('gpurun_out/r2c/bench.json')); print(j['value'], j['ms_per_step'], j['train_step'], j.get('keymask'))"
