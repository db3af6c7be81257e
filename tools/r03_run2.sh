#!/bin/bash
set -o pipefail
TAG=${1:-r03b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export STEPS=120 EVERY=20 HEAD=45
GRID='["f32",0.003,20];["f32",0.003,40];["f32",0.002,20];["f32",0.002,40];["bf16",0.003,40];["bf16",0.002,40]' timeout -k 10 500 python tools/r03_lr_probe.py > $O/p1.log 2>&1; echo rc=$?
GRID='["f32",0.003,40];["bf16",0.003,40];["f32",0.002,40];["bf16",0.002,40]' DATA_SEED=7 EXTRA='{"seed":3}' timeout -k 10 500 python tools/r03_lr_probe.py > $O/p2.log 2>&1; echo rc=$?
grep -v trace $O/p1.log $O/p2.log | cut -c1-700
grep trace $O/p*.log | cut -c1-500
