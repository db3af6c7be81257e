#!/usr/bin/env python3
"""HBM bytes of ONE step (all kernels) from two rocprofv3 PMC passes of the same command (MI355X_MICROARCH.md, HBM /
rocprofv3 section: FETCH_SIZE and WRITE_SIZE in separate passes, both in KiB, FETCH_SIZE x 2 on gfx950).

    python tools/pmc_step_traffic.py <fetch dir> <write dir> --marker adam_kernel --segments 1 \
           --algorithmic-bytes 1.04e10 --command "..." > profiles/rNN_pmc_train_bf16_traffic.json

The step is the last `segments` marker-delimited runs of dispatches (the marker kernel closes a run): adam_kernel for
the U-Net trainer (one per step), adam_prepare_kernel for the GAN (one per solver step: --segments 2 = one iteration)."""
import argparse
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()[:80]


def load(d, counter):
    fn = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    rows = {}
    with open(fn) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = int(r["Dispatch_Id"])
            rows[k] = (r["Kernel_Name"], rows.get(k, ("", 0.0))[1] + float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def last_segments(rows, marker, nseg):
    idx = [i for i, (n, _) in enumerate(rows) if marker in n]
    if len(idx) < nseg + 1:
        raise SystemExit("marker %r occurs %d times: need %d" % (marker, len(idx), nseg + 1))
    return rows[idx[-nseg - 1] + 1:idx[-1] + 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--marker", default="adam_kernel")
    ap.add_argument("--segments", type=int, default=1)
    ap.add_argument("--algorithmic-bytes", type=float, default=None)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    fe = last_segments(load(a.fetch_dir, "FETCH_SIZE"), a.marker, a.segments)
    wr = last_segments(load(a.write_dir, "WRITE_SIZE"), a.marker, a.segments)
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for n, v in fe:
        per[short(n)][0] += 1
        per[short(n)][1] += v
    for n, v in wr:
        per[short(n)][2] += v
    fk, wk = sum(v for _, v in fe), sum(v for _, v in wr)
    total = (2.0 * fk + wk) * 1024.0
    out = {k: {"launches": c, "fetch_kb_raw": f, "write_kb": w, "hbm_bytes_corrected": (2 * f + w) * 1024.0}
           for k, (c, f, w) in sorted(per.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2]))}
    out["_summary"] = {
        "command": a.command, "marker": a.marker, "segments": a.segments, "launches_per_step": len(fe),
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies a 128-byte request at 64 bytes; "
                      "exact for 64-byte pieces, so the figure is an upper bound: MI355X_MICROARCH.md HBM, DESIGN 4a)",
        "hbm_bytes_per_step": total, "hbm_bytes_per_step_uncorrected": (fk + wk) * 1024.0,
        "fetch_bytes_raw": fk * 1024.0, "write_bytes": wk * 1024.0,
        "algorithmic_bytes_per_step": a.algorithmic_bytes,
        "ratio_to_algorithmic": (total / a.algorithmic_bytes) if a.algorithmic_bytes else None,
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
