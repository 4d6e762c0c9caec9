#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/race
run() {
    name=$1; shift
    env "$@" timeout -k 10 420 python scripts/race_diag.py ${REPS:-10} > gpurun_out/race/$name.log 2>&1
    rc=$?
    echo "== $name rc=$rc: $(grep -E '^TOTAL' gpurun_out/race/$name.log)  raw-diff lines: $(grep -c 'raw tap values differ' gpurun_out/race/$name.log)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then return 1; fi
    return 0
}
run I_dump_sent S2D_DIAG_ATTN_MASK=4 S2D_DIAG_SENTINEL=12345 REPS=6 &&
run J_clone S2D_DIAG_ATTN_MASK=1 S2D_DIAG_CLONE=1 &&
run K_fullexec S2D_DIAG_ATTN_MASK=5 &&
run L_old_sent S2D_DIAG_ATTN_MASK=1 S2D_DIAG_SENTINEL=12345 &&
run M_new_sent S2D_DIAG_ATTN_MASK=0 S2D_DIAG_SENTINEL=12345
