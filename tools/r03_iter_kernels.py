"""per-kernel totals of ONE replayed iteration in a rocprofv3 kernel trace (iterations delimited by a marker kernel)"""
import csv, glob, collections, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/r03gan/prof_gan/*/*kernel_trace.csv')[-1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'adam_prepare'
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2           # markers per iteration
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if marker in r['Kernel_Name']]
a, b = idx[-2 * per], idx[-per]
seg = tr[a:b]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print("one iteration: %d launches, wall %.3f ms, busy %.3f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)', nm)
    k = m.group(1) if m else re.split(r'[<(]', nm)[0]
    if 'at::native' in nm:
        k = 'ATen ' + re.sub(r'.*at::native::', '', nm)[:60]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%4d %8.1f us  avg %6.1f  %s" % (n, t, t / n, k))
