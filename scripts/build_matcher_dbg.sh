#!/bin/bash
# timing-experiment builds of csrc/matcher.hip (S2D_MATCHER_DBG bits: 1 no target gathers, 2 no staged-row DMA, 4 no tap-table setup,
# 8 no query sampling): libs2d_hip_mdbgN.so next to the library, loaded through S2D_HIP_LIB.  Results of these builds are wrong by construction.
set -e
cd "$(dirname "$0")/../s2d_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v matcher.hip.o)
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I. -I../../include -DS2D_MATCHER_DBG=$N -c matcher.hip -o /tmp/matcher_dbg$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libs2d_hip_mdbg$N.so $OBJS /tmp/matcher_dbg$N.o
done
