#!/usr/bin/env python3
"""Per (kernel, grid) table of the SQ counters collected by tools/pmc_micro.sh: python tools/pmc_micro_summary.py gpurun_out/<TAG> [filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "g*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if flt and flt not in name:
            continue
        m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
        k = ((m.group(1) if m else name)[:60], r.get("Grid_Size", "?"))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    per = {n: acc[k][n] / max(1, cnt[k][n]) for n in acc[k]}
    busy = per.get("SQ_BUSY_CU_CYCLES", 0) or 1; wave = per.get("SQ_WAVE_CYCLES", 0) or 1
    g = lambda n: per.get(n, 0.0)
    print("%-62s grid %-9s launches %3d" % (k[0], k[1], max(cnt[k].values())))
    print("    mfma_busy %.3f  wait_inst_any/wave %.3f  wait_lds/wave %.3f  lds_active/busy_cu %.3f  bank_conflict/lds_active %.3f  lds_idx_active/busy_cu %.3f" % (
        g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * busy), g("SQ_WAIT_INST_ANY") / wave, g("SQ_WAIT_INST_LDS") / wave,
        g("SQ_ACTIVE_INST_LDS") / busy, g("SQ_LDS_BANK_CONFLICT") / (g("SQ_ACTIVE_INST_LDS") or 1), g("SQ_LDS_IDX_ACTIVE") / busy))
    print("    insts: valu %.0f mfma %.0f lds %.0f salu %.0f vmem_rd %.0f vmem_wr %.0f waves %.0f  busy_cu_cycles %.0f wave_cycles %.0f gui_active %.0f" % (
        g("SQ_INSTS_VALU"), g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"),
        g("SQ_WAVES"), g("SQ_BUSY_CU_CYCLES"), g("SQ_WAVE_CYCLES"), g("GRBM_GUI_ACTIVE")))
