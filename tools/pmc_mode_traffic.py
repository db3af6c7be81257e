#!/usr/bin/env python3
"""HBM bytes per step of a secondary bench mode from two rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
runs, both in KiB, FETCH x 2 on gfx950 -- MI355X_MICROARCH.md): every launch whose kernel name matches --kernel-regex is
summed and divided by --steps (timed + warm-up steps of the profiled command).

    python tools/pmc_mode_traffic.py <fetch dir> <write dir> --steps 25 --kernel-regex 'edt_' \
           --algorithmic-bytes 8.4e7 --command "..." > profiles/r02_pmc_weightmap_traffic.json"""
import argparse, csv, glob, json, re, sys


def total(d, counter, rx):
    fn = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    per = {}
    with open(fn) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and re.search(rx, r["Kernel_Name"]):
                k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60]
                a = per.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--kernel-regex", required=True)
    ap.add_argument("--algorithmic-bytes", type=float, default=None)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    fe, wr = total(a.fetch_dir, "FETCH_SIZE", a.kernel_regex), total(a.write_dir, "WRITE_SIZE", a.kernel_regex)
    out, tot = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, [0, 0.0]), wr.get(k, [0, 0.0])
        b = (2.0 * f[1] + w[1]) * 1024.0 / a.steps
        out[k] = {"launches_per_step": max(f[0], w[0]) / a.steps, "hbm_bytes_per_step": b}
        tot += b
    out["_summary"] = {"hbm_bytes_per_step": tot, "steps": a.steps, "kernel_regex": a.kernel_regex,
                       "algorithmic_bytes_per_step": a.algorithmic_bytes,
                       "ratio_to_algorithmic": (tot / a.algorithmic_bytes) if a.algorithmic_bytes else None,
                       "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate passes", "command": a.command}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
