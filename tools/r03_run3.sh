#!/bin/bash
# round 3: whole GPU suite + the driver's bench command
set -o pipefail
TAG=${1:-r03d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
( time timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3; echo "bench rc=$?"; cut -c1-600 $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','failed','frac','cores')}) for k,v in d.items() if k in ('value','ms_per_step','cpu_baseline','train_bf16','gan_bf16','roofline')})
PY
