#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
BF="--no-other-schedule --no-cpu-baseline --no-train-step --no-keymask"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 bench.py --one-stream $BF --steps 5 --warmup 2 > $O/bench_one.json 2> $O/bench_one.err; echo "one rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/two -o two -- python3 bench.py $BF --steps 5 --warmup 2 > $O/bench_two.json 2> $O/bench_two.err; echo "two rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH -o run -- python3 bench.py --one-stream $BF --no-kernel-events --steps 2 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err; echo "pmcF rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE -o run -- python3 bench.py --one-stream $BF --no-kernel-events --steps 2 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err; echo "pmcW rc=$?"
python3 scripts/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE $O/r2_pmc_traffic.json 3.5 c2279b3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 scripts/mb_train_step.py > $O/train_step.txt 2>&1; echo "train rc=$?"; grep -E "iteration|phases|forward" $O/train_step.txt
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
