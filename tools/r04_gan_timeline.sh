#!/bin/bash
# per-launch timeline of one GAN iteration (bf16 storage): rocprofv3 kernel trace -> tools/trace_step.py
O=gpurun_out/r04gantl; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --mode gan --dtype bf16 --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/trace.log 2>&1; echo "rocprof rc=$?"
cd $R
python - <<'PY'
import csv, glob
fn = sorted(glob.glob('gpurun_out/r04gantl/trace/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if 'adam_multi' in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-1] + 1          # two optimiser launches per iteration (D, G)
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); tot = 0
out = open('gpurun_out/r04gantl/iteration_timeline.txt', 'w')
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); tot += e - s
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    out.write("%8.1f us  +%8.1f  grid %8s  %s\n" % ((e - s) / 1e3, (s - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), name[:120]))
out.write("launches %d, kernel time %.1f us, span %.1f us\n" % (len(step), tot / 1e3, (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
tail -1 $O/iteration_timeline.txt
