#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/race
run() {
    name=$1; shift
    env "$@" timeout -k 10 420 python scripts/race_diag.py 8 > gpurun_out/race/$name.log 2>&1
    rc=$?
    echo "== $name rc=$rc: $(grep -E '^TOTAL' gpurun_out/race/$name.log)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then return 1; fi
    return 0
}
run N_dump_stages S2D_DIAG_ATTN_MASK=4 &&
run O_scalar_alu S2D_DIAG_ATTN_MASK=6 &&
grep -h "interpolated v" gpurun_out/race/N_dump_stages.log | awk '{a+=$9; } END {print "dump lines:", NR}' ;
grep -h -E "^     (v|word2):" gpurun_out/race/N_dump_stages.log | head -20
