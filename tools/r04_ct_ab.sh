#!/bin/bash
# A/B of a compile-time switch of sq_convt_f32_v2.hip on ONE box: bash tools/r04_ct_ab.sh MACRO v1 v2 ...
R=$GRAFT_REPO_ROOT; M=$1; shift
cd $R/sequitr_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D$M=$v -c sq_convt_f32_v2.hip -o ../_build/sq_convt_f32_v2.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libsequitr_hip.so ../_build/*.o || exit 1
  echo "== $M=$v"
  (cd $R && timeout -k 10 200 python tools/r03_op_bench.py convT 2>/dev/null) || exit 1
done
