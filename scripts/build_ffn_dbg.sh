#!/bin/bash
# timing-experiment builds of csrc/ffn.hip (S2D_FFN_DBG bits: 1 no DMA, 2 no barrier, 4 no activation work, 8 no fragment reads, 16 clock / phase stamps,
# 32.. memory-op ablations: see the kernel): libs2d_hip_dbgN.so next to the library, loaded through S2D_HIP_LIB by scripts/mb_ffn_dbg.py /
# mb_ffn_clock.py.  Results of builds with bits 1-8 or >= 32 are wrong by construction.  EPI=<0|1|2> in the environment also selects the
# epilogue's residual source (S2D_FFN_EPI; the library is then named ..._dbgN_eEPI.so and the python side needs S2D_FFN_EPI=<EPI> too).
set -e
cd "$(dirname "$0")/../s2d_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v ffn.hip.o)
for N in "$@"; do
  SUF=$N; DEF="-DS2D_FFN_DBG=$N"
  if [ -n "$EPI" ]; then SUF=${N}_e$EPI; DEF="$DEF -DS2D_FFN_EPI=$EPI"; fi
  if [ -n "$POSTV" ]; then SUF=${SUF}_p$POSTV; DEF="$DEF -DS2D_FFN_POSTV=$POSTV"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -I. -I../../include $DEF -c ffn.hip -o /tmp/ffn_dbg$SUF.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libs2d_hip_dbg$SUF.so $OBJS /tmp/ffn_dbg$SUF.o
done
