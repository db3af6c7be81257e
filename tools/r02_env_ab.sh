#!/bin/bash
# A/B of an environment switch on the bf16 training line: bash tools/r02_env_ab.sh VAR v1 v2 ...
R=$GRAFT_REPO_ROOT; cd $R
VAR=$1; shift
for v in "$@"; do
  for rep in 1 2; do
    echo -n "$VAR=$v: "
    env $VAR=$v timeout -k 10 200 python bench.py --mode train --dtype bf16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  done
done
