R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ex; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ex -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --force-exchange > $O/ex.log 2>&1
cd $R
cat $O/ex/*/*kernel_stats.csv | cut -c1-170
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_ex/ex/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print one steady-state tick: find a gather_bodies dispatch late in the run and print +-8 kernels
idx=[i for i,r in enumerate(rows) if "integrate_free" in r["Kernel_Name"]]
i=idx[len(idx)//2]
t0=int(rows[i]["Start_Timestamp"])
for r in rows[i:i+40]:
    print("%8.2f us  +%7.2f us  q%s  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Queue_Id"],r["Kernel_Name"][:70]))
PY
