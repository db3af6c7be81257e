#!/usr/bin/env python3
"""Compact trace of a kernel's main loop from hipcc -S output: runs of instruction classes between the control points
(labels, branches, barriers, waitcnt vmcnt), so one can see WHERE the VALU instructions of an item sit.
  python tools/isa_loop_trace.py file.s 'mangled-substring' [min_run]"""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_loop_count import classify, function_body


def main():
    body = function_body(open(sys.argv[1]).read().splitlines(), sys.argv[2])
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    _, a, b = max(loops)
    seg = {}
    def flush():
        if seg:
            print("      " + "  ".join("%s %d" % kv for kv in sorted(seg.items())))
            seg.clear()
    for i in range(a, b + 1):
        l = body[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            flush(); print("%5d %s" % (i, m.group(1))); continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
        if not m or l.strip().startswith((".", ";")):
            continue
        op = m.group(1)
        if op in ("s_barrier",) or op.startswith(("s_cbranch", "s_branch")) or (op == "s_waitcnt" and "vmcnt" in m.group(2)):
            flush(); print("%5d   %s %s" % (i, op, m.group(2).split(";")[0].strip())); continue
        c = classify(op)
        seg[c] = seg.get(c, 0) + 1


if __name__ == "__main__":
    main()
