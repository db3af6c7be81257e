#!/bin/bash
# wgrad block-shape sweep: tools/wgrad_bf16_bench.py under block-shape overrides
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wsweep}; mkdir -p $O; cd $R
for s in 0,0 2,2 1,4; do
  echo "== SQ_WGRAD_BF16_K3=$s" | tee -a $O/sweep.txt
  SQ_WGRAD_BF16_K3=$s timeout -k 10 120 python tools/wgrad_bf16_bench.py 2>&1 | grep "us" | tee -a $O/sweep.txt || exit 1
done
