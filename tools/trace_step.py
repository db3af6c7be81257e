#!/usr/bin/env python3
"""Per-launch timeline of ONE step from a rocprofv3 --kernel-trace CSV: python tools/trace_step.py <dir> [marker-kernel-substring]
The step is cut at the last-but-one occurrence of the marker kernel (default: adam_kernel) .. the last occurrence."""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
fn = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
tot = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print("%9.1f us  +%8.1f  grid %8s wg %4s  %s" % ((e - s) / 1e3, (s - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), name[:110]))
print("launches %d, kernel time %.1f us, span %.1f us" % (len(step), tot / 1e3, (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
