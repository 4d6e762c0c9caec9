#!/bin/bash
# timing-experiment builds of csrc/msda.hip (S2D_MSDA_DBG bits: fused forward gather: 1 no value gathers, 2 one corner line per sample instead of four
# ): libs2d_hip_sdbgN.so next to the library, loaded through S2D_HIP_LIB.  Results of these builds are wrong by construction.
set -e
cd "$(dirname "$0")/../s2d_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v msda.hip.o)
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I. -I../../include -DS2D_MSDA_DBG=$N -c msda.hip -o /tmp/msda_dbg$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libs2d_hip_sdbg$N.so $OBJS /tmp/msda_dbg$N.o
done
