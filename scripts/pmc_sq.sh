R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq3 -- python3 $R/bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/sq3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/tcc2 -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/tcc2.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,statistics,collections
for tag,kern in (("sq2","integrate_free"),("sq3","step_plane"),("tcc2","integrate_free")):
    f=glob.glob(f"gpurun_out/pmc_sq/{tag}/**/*counter_collection.csv",recursive=True)
    if not f: print(tag,"no file"); continue
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if kern in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag,kern,{k:round(statistics.median(v)) for k,v in d.items()})
PY
