"""Average durations of the per-component kernels and of the idle gaps between them from a rocprofv3
--kernel-trace CSV (last two thirds of the trace).  usage: gapstat.py <kernel_trace.csv> [label]"""
import collections, csv, sys

def short(n):
    n = n.split('(')[0].split('::')[-1]
    return n.split('<')[0]

tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None; gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
for r in tr[len(tr) // 3:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = short(r['Kernel_Name'])
    dur[n].append(e - s)
    if prev: gaps[(prev[0], n)].append(s - prev[1])
    prev = (n, e)
out = []
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:4]:
    out.append(f"{k[0][:12]}->{k[1][:12]} {sum(v)/len(v)/1e3:.2f} us (n={len(v)})")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:3]:
    out.append(f"{k[:16]} {sum(v)/len(v)/1e3:.1f} us")
print(sys.argv[2] if len(sys.argv) > 2 else '', ' | '.join(out))
