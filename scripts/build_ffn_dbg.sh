#!/bin/bash
# timing-experiment builds of csrc/ffn.hip (S2D_FFN_DBG bits: 1 no DMA, 2 no barrier, 4 no activation work, 8 no fragment reads): libs2d_hip_dbgN.so next to
# the library, loaded through S2D_HIP_LIB by scripts/mb_ffn_dbg.py.  Results of these builds are wrong by construction.
set -e
cd "$(dirname "$0")/../s2d_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v ffn.hip.o)
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I. -I../../include -DS2D_FFN_DBG=$N -c ffn.hip -o /tmp/ffn_dbg$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libs2d_hip_dbg$N.so $OBJS /tmp/ffn_dbg$N.o
done
