#!/bin/bash
# per-kernel times of the point-loss sequence under rocprofv3 for the shipped library and experiment builds:  bash scripts/mb_loss_dbg.sh <out dir> 0 1 2 ...
export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
for N in "$@"; do
  if [ "$N" = "0" ]; then unset S2D_HIP_LIB; else export S2D_HIP_LIB=$PWD/s2d_amd/csrc/libs2d_hip_ldbg$N.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$N -o t -- python3 scripts/mb_loss.py > $O/v$N.log 2>&1
  echo "== dbg=$N: $(grep ms/call $O/v$N.log)"
  python3 scripts/kernel_table.py $O/v$N/t_kernel_stats.csv 7 8 | grep -E "hist_kernel|accumulate|gather_rows|hist_stream|pack_planes|select"
  find $O -name "*kernel_trace.csv" -delete
done
