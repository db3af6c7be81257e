#!/bin/bash
# A/B of a compile-time switch on ONE box: bash tools/r02_macro_ab.sh FILE.hip MACRO v1 v2 ...   (bf16 training line, 2 runs each)
R=$GRAFT_REPO_ROOT; F=$1; M=$2; shift 2
cd $R/sequitr_amd/csrc
for v in "$@" "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D$M=$v -c $F -o ../_build/${F%.hip}.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libsequitr_hip.so ../_build/*.o || exit 1
  echo -n "$M=$v: "
  (cd $R && timeout -k 10 200 python bench.py --mode train --dtype bf16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
done
