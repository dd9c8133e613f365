# SQ counters of np_convex_static on configs[4] (16 384 teapot hulls on the static box floor)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_c5_r04; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $R/bench.py --config 5 --steps 120 --warmup 5 --no-cpu-baseline --no-extras > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --output-format csv -d $O/b -- python3 $R/bench.py --config 5 --steps 120 --warmup 5 --no-cpu-baseline --no-extras > $O/b.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,statistics,collections
for tag in ("a","b"):
    f=glob.glob(f"gpurun_out/pmc_c5_r04/{tag}/**/*counter_collection.csv",recursive=True)
    if not f: print(tag,"no file"); continue
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "np_convex_static" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag,{k:round(statistics.median(v)) for k,v in d.items()})
PY
