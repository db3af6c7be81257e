#!/bin/bash
# A/B of a compile-time switch of sq_conv_f32_l0.hip on ONE box: bash tools/r04_l0_ab.sh MACRO v1 v2 ...  (level-0 operator timing)
R=$GRAFT_REPO_ROOT; M=$1; shift
cd $R/sequitr_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D$M=$v -c sq_conv_f32_l0.hip -o ../_build/sq_conv_f32_l0.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libsequitr_hip.so ../_build/*.o || exit 1
  echo "== $M=$v"
  (cd $R && timeout -k 10 200 python tools/r03_op_bench.py level0 2>/dev/null) || exit 1
done
