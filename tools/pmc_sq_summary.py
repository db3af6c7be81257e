#!/usr/bin/env python3
"""Per-kernel summary of the SQ counter passes of tools/pmc_sq.sh (gpurun_out/<TAG>/g1..g4) as JSON on stdout:
  python tools/pmc_sq_summary.py gpurun_out/r01u_sq > profiles/r01u_pmc_sq_summary.json
Ratios are per launch, averaged over the launches of a kernel.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(4 * SQ_BUSY_CU_CYCLES): the MFMA counter ticks per SIMD (32 cycles per v_mfma_f32_16x16x4_f32,
MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units"), the CU-busy counter per CU with 4 SIMDs."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(root, "g*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
    out = {}
    for k, c in acc.items():
        if "conv" not in k and "convT" not in k:
            continue
        n = {name: max(1, launches[k][name]) for name in c}
        per = {name: c[name] / n[name] for name in c}
        busy = per.get("SQ_BUSY_CU_CYCLES", 0.0)
        wave = per.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": int(max(n.values()))}
        if busy:
            e["mfma_busy_frac"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), 4)
        if wave:
            e["wait_inst_any_frac_of_wave_cycles"] = round(per.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4)
        if per.get("SQ_ACTIVE_INST_LDS"):
            e["lds_bank_conflict_frac_of_lds_active"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_ACTIVE_INST_LDS"], 4)
        if per.get("SQ_ACTIVE_INST_VMEM"):
            e["vmem_ta_addr_fifo_full_frac_of_vmem_active"] = round(per.get("SQ_VMEM_TA_ADDR_FIFO_FULL", 0.0) / per["SQ_ACTIVE_INST_VMEM"], 4)
            e["vmem_wr_ta_data_fifo_full_frac_of_vmem_active"] = round(per.get("SQ_VMEM_WR_TA_DATA_FIFO_FULL", 0.0) / per["SQ_ACTIVE_INST_VMEM"], 4)
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INST_LEVEL_VMEM"):
            if name in per:
                e[name.lower() + "_per_launch"] = round(per[name], 1)
        out[k] = e
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
