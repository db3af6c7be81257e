#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python bench.py ...
    python tools/pmc_traffic.py out/fetch/*/*_counter_collection.csv out/write/*/*_counter_collection.csv \
           --steps 4 --algorithmic-bytes-per-step 6.86e9 > profiles/rNN_pmc_hbm_traffic.json

Both counters are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  `--steps` = warm-up + timed steps of the profiled command
(every launch of the run is averaged).  The summary covers every conv_mfma_f32_v2 / conv_l0 instantiation
(bench.py's roofline kernels)."""
import argparse
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def load(fn, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(fn) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--steps", type=int, required=True, help="bench steps + warm-up steps in the profiled run")
    ap.add_argument("--algorithmic-bytes-per-step", type=float, default=None)
    ap.add_argument("--kernel-regex", default=r"conv_mfma_f32_v2_kernel|conv_l0_kernel")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    fe, wr = load(a.fetch_csv, "FETCH_SIZE"), load(a.write_csv, "WRITE_SIZE")
    out = {}
    tot_b, tot_l, tot_raw = 0.0, 0, 0.0
    for k in sorted(set(fe) | set(wr)):
        nf, f = fe.get(k, (0, 0.0))
        nw, w = wr.get(k, (0, 0.0))
        n = max(nf, nw)
        if n == 0:
            continue
        fk, wk = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        b = (2.0 * fk + wk) * 1024.0
        out[k] = {"launches": n, "fetch_kb_raw": fk, "write_kb": wk, "hbm_bytes_per_launch_corrected": b}
        if re.search(a.kernel_regex, k):
            tot_b += b * n
            tot_l += n
            tot_raw += (fk + wk) * 1024.0 * n
    out["_summary"] = {
        "command": a.command,
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of a wide "
                      "coalesced read (MI355X_MICROARCH.md HBM)",
        "kernel_regex": a.kernel_regex,
        "conv_mfma_launches_per_step": tot_l / float(a.steps),
        "conv_mfma_hbm_bytes_per_step": tot_b / float(a.steps),
        "conv_mfma_hbm_bytes_per_launch_avg": tot_b / max(1, tot_l),
        "conv_mfma_algorithmic_bytes_per_step": a.algorithmic_bytes_per_step,
        "conv_mfma_hbm_bytes_per_step_uncorrected": tot_raw / float(a.steps),
        "note": "the x2 rule holds for 128-byte requests (the level-0 kernels: 64-byte pixels in contiguous rows); the "
                "16-channel chunk loads of the C >= 32 layers are 64-byte pieces at a 128+ byte stride, which FETCH_SIZE "
                "tallies exactly -- for those launches the corrected figure is an upper bound and the uncorrected one is "
                "the better estimate (DESIGN 4a)",
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
