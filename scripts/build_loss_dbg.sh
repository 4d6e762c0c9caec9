#!/bin/bash
# timing-experiment builds of csrc/loss.hip (S2D_LOSS_DBG bits: hist_kernel<0>: 1 no histogram atomics, 2 no LDS taps, 4 no sample store
# ): libs2d_hip_ldbgN.so next to the library, loaded through S2D_HIP_LIB.  Results of these builds are wrong by construction.
set -e
cd "$(dirname "$0")/../s2d_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v loss.hip.o)
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I. -I../../include -DS2D_LOSS_DBG=$N -c loss.hip -o /tmp/loss_dbg$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libs2d_hip_ldbg$N.so $OBJS /tmp/loss_dbg$N.o
done
