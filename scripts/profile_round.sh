#!/bin/bash
# The measurement set behind profiles/ (run on the GPU box from the repo root):
#     bash scripts/profile_round.sh <out-dir under gpurun_out/> <commit the build came from>
# kernel-trace statistics of bench.py on both schedules, the two PMC passes (FETCH_SIZE, WRITE_SIZE; counters and traces
# are never combined in one rocprofv3 run), the training step under the kernel trace, and the default bench line.
# Every rocprofv3 command puts the program itself behind `--`.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-prof}; C=${2:-unrecorded}; mkdir -p $O
BF="--no-other-schedule --no-cpu-baseline --no-train-step --no-keymask --no-amp"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 bench.py --one-stream $BF --steps 5 --warmup 2 > $O/bench_one.json 2> $O/bench_one.err; echo "one rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/two -o two -- python3 bench.py $BF --steps 5 --warmup 2 > $O/bench_two.json 2> $O/bench_two.err; echo "two rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH -o run -- python3 bench.py --one-stream $BF --no-kernel-events --steps 2 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err; echo "pmcF rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE -o run -- python3 bench.py --one-stream $BF --no-kernel-events --steps 2 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err; echo "pmcW rc=$?"
python3 scripts/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE $O/pmc_traffic.json 3.5 $C
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 scripts/mb_train_step.py > $O/train_step.txt 2>&1; echo "train rc=$?"; grep -E "iteration|phases|forward" $O/train_step.txt
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
