#!/bin/bash
# copy the summaries of a scripts/profile_round.sh (and optionally profile_northstar.sh) run from gpurun_out/ into profiles/
#     bash scripts/collect_profiles.sh <round tag, e.g. r4> <gpurun_out dir of profile_round.sh> [<gpurun_out dir of profile_northstar.sh>]
set -e
R=$1; O=gpurun_out/$2; P=profiles
cp $O/one/one_kernel_stats.csv $P/${R}_kernel_stats_one_stream.csv
cp $O/two/two_kernel_stats.csv $P/${R}_kernel_stats_two_streams.csv
cp $O/pmc_traffic.json $P/${R}_pmc_traffic.json
cp $O/train/train_kernel_stats.csv $P/${R}_train_step_kernel_stats.csv
grep -E "^iteration|^forward|^phases" $O/train_step.txt > $P/${R}_train_step_c4.txt
cp $O/bench_default.json $P/${R}_bench_default.json
if [ -n "$3" ]; then cp gpurun_out/$3/pmc_northstar.json $P/${R}_pmc_northstar.json; fi
ls -la $P | grep "${R}_"
