#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 1700 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r2b/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err; echo "bench rc=$?"; cat gpurun_out/r2b/bench_default.json | cut -c1-3000; tail -5 gpurun_out/r2b/bench_default.err
