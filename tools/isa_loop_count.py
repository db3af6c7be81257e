#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc -S output (no GPU needed).
  python tools/isa_loop_count.py file.s 'mangled-name-substring' [--hist]
Finds every backward branch of the function, takes the LARGEST loop body (label .. branch) and counts VALU (non-MFMA),
MFMA, SALU, LDS and VMEM instructions in it, with a histogram of the VALU opcodes.  The counts are static (one
trip through the loop body including both sides of its inner branches), which for the persistent conv kernels is one
work item.  Used to check the 'every VALU instruction is 4 cycles the SIMD does not issue an MFMA' model
(HISTORY.md round 4)."""
import collections
import re
import sys


def function_body(lines, key):
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*:", l) and key in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    return lines[start:end]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    body = function_body(open(path).read().splitlines(), key)
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
    loops.sort(reverse=True)
    print("function lines", len(body), "backward branches", len(loops))
    for size, a, b in loops[:int(sys.argv[4]) if len(sys.argv) > 4 else 1]:
        cnt, hist = collections.Counter(), collections.Counter()
        for l in body[a:b + 1]:
            m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
            if not m or l.strip().startswith((".", ";")):
                continue
            c = classify(m.group(1))
            cnt[c] += 1
            if c == "valu":
                hist[m.group(1)] += 1
        print("loop lines %d..%d: %s" % (a, b, dict(cnt)))
        if "--hist" in sys.argv:
            for k, v in hist.most_common(40):
                print("   %-28s %d" % (k, v))


if __name__ == "__main__":
    main()
