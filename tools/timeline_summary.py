"""per-kernel totals of an iteration timeline written by tools/r0x_*_timeline.sh (duration us, +offset, grid, name)"""
import collections, re, sys
rows = [l for l in open(sys.argv[1]) if ' us  +' in l]
agg = collections.defaultdict(lambda: [0, 0.0])
small = 0
for l in rows:
    us = float(l.split(' us')[0])
    name = l.split('grid', 1)[1].split(None, 1)[1].strip()
    name = re.sub(r'^wg\s+\d+\s+', '', name)
    if 'at::native' in name:
        k = 'ATen ' + re.sub(r'.*at::native::', '', name)[:50]
    else:
        m = re.match(r'(?:void )?(?:\d+)?([A-Za-z0-9_]+?)(?:I[LNDb]|<|\()', name)
        k = m.group(1) if m else name[:40]
    agg[k][0] += 1
    agg[k][1] += us
    small += us < 8.0
tot = sum(v[1] for v in agg.values())
print("%d launches, %.1f us of kernels; %d launches under 8 us (%.1f us)" % (len(rows), tot, small, sum(float(l.split(' us')[0]) for l in rows if float(l.split(' us')[0]) < 8.0)))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d %8.1f us  avg %6.1f  %s" % (n, t, t / n, k))
