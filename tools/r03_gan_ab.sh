#!/bin/bash
# GAN iteration: bf16 storage vs the mixed (f32 storage) form on one box, then a kernel profile of the storage form
O=gpurun_out/r03gan; mkdir -p $O; R=$PWD
timeout -k 10 300 python bench.py --mode gan --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gan_bf16.json 2>$O/bench_gan_bf16.err; echo "bf16 rc=$?"
timeout -k 10 300 python bench.py --mode gan --dtype mixed --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gan_mixed.json 2>$O/bench_gan_mixed.err; echo "mixed rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gan -- python $R/bench.py --mode gan --dtype bf16 --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/prof_gan.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof_gan/*/*kernel_stats.csv | head -1); cp $f $O/gan_bf16_kernel_stats.csv
python - <<'PY'
import json
for m in ("bf16", "mixed"):
    try:
        d = json.loads(open("gpurun_out/r03gan/bench_gan_%s.json" % m).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(m, d["ms_per_step"], "ms", d["value"], "Mpix/s", "alg bytes/step %.3e" % r["algorithmic_bytes_per_step"], "ops", r["ops_per_step"])
    except Exception as e:
        print(m, "failed", e)
PY
head -45 $O/gan_bf16_kernel_stats.csv | cut -c1-200
