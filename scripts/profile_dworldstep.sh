# where a dWorldStep tick of the reference's pen goes: rocprofv3 kernel stats of the ODE-API harness, N bodies settled with
# QuickStep, then TICKS ticks of dWorldStep at 1/120 (main.c:211-215).  usage: profile_dworldstep.sh [N=512] [single=0] [SETTLE=900] [TICKS=40] [TAG=r04_lcp]
N=${1:-512}; SINGLE=${2:-0}; SETTLE=${3:-900}; TICKS=${4:-40}; TAG=${5:-r04_lcp}
cd $GRAFT_REPO_ROOT
if [ "$SINGLE" = "1" ]; then LIB=ode_mi355_single; DEF=-DdSINGLE; P=f32; else LIB=ode_mi355; DEF=; P=f64; fi
gcc -O1 $DEF -Iinclude tests/harness/ode_tick_harness.c -o /tmp/harness_p -Lrl-ode-physics_amd -l$LIB -Wl,-rpath,$PWD/rl-ode-physics_amd -lm
python3 - $N $SETTLE $TICKS <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_ode_compat import _scene_text, pkg
n, settle, ticks = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
open("/tmp/scene_p.txt", "w").write(_scene_text(1.0/120.0, settle + ticks, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.2, 12.0))))
PY
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}_${N}_${P}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export HARNESS_STEPPER=exact HARNESS_EXACT_AFTER=$SETTLE HARNESS_TIME_FROM=$((SETTLE+4)) DMX_LCP_REPORT=1
/tmp/harness_p < /tmp/scene_p.txt > /dev/null 2> $O/plain.txt; cat $O/plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- /tmp/harness_p < /tmp/scene_p.txt > /dev/null 2> $O/err.txt
F="$(ls -t $O/k/*/*kernel_stats.csv | head -1)"
python3 - "$F" $R/gpurun_out/${TAG}_${N}_${P}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
for r in rows[:28]:
    line = f'{r["Name"][:90]:90s} calls {int(r["Calls"]):7d}  total_us {float(r["TotalDurationNs"])/1e3:12.1f}  avg_us {float(r["AverageNs"])/1e3:9.2f}  pct {float(r["Percentage"]):6.2f}'
    print(line); out.write(line + "\n")
PY
