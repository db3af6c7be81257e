#!/bin/bash
# A/B of a compile-time switch of sq_conv_f32_l0.hip on ONE box, measured INSIDE the network's step (bench.py), not stand-alone
R=$GRAFT_REPO_ROOT; M=$1; shift
cd $R/sequitr_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D$M=$v -c sq_conv_f32_l0.hip -o ../_build/sq_conv_f32_l0.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libsequitr_hip.so ../_build/*.o || exit 1
  echo -n "$M=$v: "
  (cd $R && timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-end-to-end --no-side-lines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])") || exit 1
done
