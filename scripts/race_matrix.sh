#!/bin/bash
# one GPU call: the two-stream stale-read matrix (scripts/race_diag.py under different kernel variants / runtime knobs)
# plus the AQL packet headers the runtime emits for the two-stream schedule
set -o pipefail
mkdir -p gpurun_out/race
run() {
    name=$1; shift
    env "$@" timeout -k 10 420 python scripts/race_diag.py ${REPS:-12} > gpurun_out/race/$name.log 2>&1
    rc=$?
    echo "== $name rc=$rc: $(grep -E '^TOTAL' gpurun_out/race/$name.log)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then return 1; fi
    return 0
}
run A_old_two S2D_DIAG_ATTN_MASK=1 &&
run B_old_one S2D_DIAG_ATTN_MASK=1 S2D_DIAG_ONE_STREAM=1 &&
run C_old_acquire S2D_DIAG_ATTN_MASK=2 &&
run D_old_gemm_release S2D_DIAG_ATTN_MASK=1 S2D_DIAG_GEMM_RELEASE=1 &&
run E_old_sc1_loads S2D_DIAG_ATTN_MASK=3 &&
run F_old_optflush0 S2D_DIAG_ATTN_MASK=1 AMD_OPT_FLUSH=0 &&
run G_new_two S2D_DIAG_ATTN_MASK=0 &&
run H_old_one_hwq S2D_DIAG_ATTN_MASK=1 GPU_MAX_HW_QUEUES=1 &&
( AMD_LOG_LEVEL=4 timeout -k 10 300 python scripts/aql_probe.py 2> /tmp/aql_two.log | tail -1;
  grep -E "Header|ShaderName" /tmp/aql_two.log > gpurun_out/race/aql_two.txt;
  python scripts/aql_parse.py gpurun_out/race/aql_two.txt > gpurun_out/race/aql_two_summary.txt; head -40 gpurun_out/race/aql_two_summary.txt )
