#!/bin/bash
# wgrad ablation: rebuild the wgrad object with SQ_WG_ABLATE=k and time the deep shapes (results are wrong by construction)
R=$GRAFT_REPO_ROOT; cd $R/sequitr_amd/csrc
for k in 0 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DSQ_WG_ABLATE=$k -c sq_conv_wgrad_bf16.hip -o ../_build/sq_conv_wgrad_bf16.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libsequitr_hip.so ../_build/*.o || exit 1
  echo "== ABLATE $k"; (cd $R && python tools/wgrad_scaling.py) | grep "N="
done
