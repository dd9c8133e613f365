# how many host passes a dWorldStep solve takes over a long run (DMX_LCP_TRACE=-1 prints one line per grid solve).  usage: soak_dworldstep_passes.sh [bodies=512] [ticks=1500] [--single]
N=${1:-512}; T=${2:-1500}; S=${3:-}
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/soak_trace.txt
DMX_LCP_TRACE=-1 python scripts/time_dworldstep.py --settle 900 --ticks $T --bodies $N $S --stderr-to gpurun_out/soak_trace.txt
python3 - <<'PY'
import re, collections, statistics
P = []; L = []; S = []
for l in open("gpurun_out/soak_trace.txt"):
    m = re.match(r"lcp solve: (\d+) m (\d+) nbd (\d+) passes (\d+) l2_passes (\d+) single (\d+)", l)
    if m: P.append(int(m.group(4))); L.append(int(m.group(5))); S.append(int(m.group(6)))
print(len(P), "grid solves; host passes per solve: mean %.2f median %d max %d; of them with the LDS level: mean %.2f; single flips: %d" % (statistics.mean(P), statistics.median(P), max(P), statistics.mean(L), sum(S)))
c = collections.Counter(min(p, 30) for p in P); print("histogram (30 = 30 and more):", sorted(c.items()))
print("passes in all:", sum(P), "; in solves of more than 15:", sum(p for p in P if p > 15))
PY
