#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of every HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
  python tools/resource_report.py [--spills-only] [file.hip ...]
Exit status 1 when any kernel spills more than --max-spill VGPRs (default 8): a guard against the unrolled
wave-uniform-read pattern that made the multi-channel direct kernels spill 100-570 registers."""
import argparse
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(path):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                          "-c", path, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                         stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE,
                                          text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--spills-only", action="store_true")
    ap.add_argument("--max-spill", type=int, default=8)
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "sequitr_amd", "csrc", "*.hip")))
    bad = 0
    for f in files:
        for r in report(f):
            if a.spills_only and not r.get("spill"):
                continue
            print("%-24s vgpr %3d agpr %3d spill %3d occ %d lds %6d  %s" % (os.path.basename(f), r.get("vgpr", 0), r.get("agpr", 0),
                                                                           r.get("spill", 0), r.get("occ", 0), r.get("lds", 0),
                                                                           r["name"][-90:]))
            bad += r.get("spill", 0) > a.max_spill
    sys.exit(1 if bad else 0)
