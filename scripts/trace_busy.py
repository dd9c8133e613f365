"""GPU busy time against the span of a rocprofv3 kernel trace: how much of a run is kernels, how much is gaps between them.
usage: trace_busy.py <dir with *_kernel_trace.csv> [ticks]"""
import csv, glob, sys, collections
d = sys.argv[1]; ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
span = ev[-1][1] - ev[0][0]; busy = sum(e - s for s, e, _ in ev)
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
gaps_pos = sorted(g for g in gaps if g > 0)
print(f"{len(ev)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({busy/span:.2f}), median gap {gaps_pos[len(gaps_pos)//2]/1e3:.2f} us, "
      f"gaps > 20 us: {sum(g > 20000 for g in gaps)} totalling {sum(g for g in gaps if g > 20000)/1e6:.2f} ms")
if ticks: print(f"per tick: {len(ev)/ticks:.1f} launches, {busy/ticks/1e3:.1f} us busy, {span/ticks/1e3:.1f} us span")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.replace("(anonymous namespace)::", "").split("(")[0][-60:]; by[k][0] += 1; by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:60s} {c:7d} calls {t/1e6:9.3f} ms  avg {t/c/1e3:8.2f} us" + (f"  {c/ticks:5.2f}/tick" if ticks else ""))
# idle time on the device by transition (which kernel the device waited for, after which): where the host is in the loop
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("dmx::", "").split("<")[0].split("(")[0][-28:]
tr = collections.defaultdict(list)
for i in range(1, len(ev)):
    tr[(short(ev[i - 1][2]), short(ev[i][2]))].append(max(0, ev[i][0] - ev[i - 1][1]))
print("idle before a kernel, by (previous -> next), largest totals first:")
for k, v in sorted(tr.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print(f"  {k[0]:>28s} -> {k[1]:28s} n={len(v):5d}  median {v2[len(v2) // 2] / 1e3:7.2f} us  total {sum(v) / 1e6:7.2f} ms" + (f"  {sum(v) / ticks / 1e3:6.2f} us/tick" if ticks else ""))
