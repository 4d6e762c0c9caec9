#!/bin/bash
# north-star per-kernel counters (gather HBM bytes, MFMA busy of the einsum / cross-attention): four rocprofv3 passes of
# scripts/mb_northstar_kernels.py joined by scripts/pmc_northstar.py.   bash scripts/profile_northstar.sh <out-dir under gpurun_out/> <commit>
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-ns}; C=${2:-unrecorded}; mkdir -p $O
# third argument "bench": the program of every pass is bench.py itself (one stream, no other schedule) instead of the kernels alone
if [ "$3" = "bench" ]; then
  P="python3 bench.py --one-stream --no-other-schedule --no-cpu-baseline --no-train-step --no-keymask --no-amp --no-kernel-events --steps 2 --warmup 1"; L="bench.py --one-stream (c4, the timed step itself)"
  for pass in "trace:--kernel-trace --stats" "mfma:--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:--pmc FETCH_SIZE" "write:--pmc WRITE_SIZE"; do
    n=${pass%%:*}; a=${pass#*:}
    timeout -k 10 400 rocprofv3 $a --output-format csv -d $O/$n -o t -- $P > $O/$n.log 2>&1; echo "$n rc=$?"
  done
  python3 scripts/pmc_northstar.py $O $O/pmc_northstar.json $C "$L"
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
  exit 0
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 scripts/mb_northstar_kernels.py > $O/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o t -- python3 scripts/mb_northstar_kernels.py > $O/mfma.log 2>&1; echo "mfma rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 scripts/mb_northstar_kernels.py > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 scripts/mb_northstar_kernels.py > $O/write.log 2>&1; echo "write rc=$?"
python3 scripts/pmc_northstar.py $O $O/pmc_northstar.json $C
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
