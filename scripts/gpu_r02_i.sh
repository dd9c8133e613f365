R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
bash scripts/profile_r02.sh > $O/profile.log 2>&1; tail -12 $O/profile.log
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" | tee -a $O/summary.txt
python bench.py --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err; echo "bench1000 rc=$?" | tee -a $O/summary.txt
python bench.py --config 5 --no-cpu-baseline --steps 200 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" | tee -a $O/summary.txt
( time python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_g2.json 2> $O/bench_g2.err ) 2>> $O/summary.txt; echo "g2 rc=$?" | tee -a $O/summary.txt
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("unreadable", e); raise SystemExit
print({k:o[k] for k in ("value","ms_per_step","n_gpus","scaling")}, o["timing"]["blocks"])
print(" roofline", {k:o["roofline"].get(k) for k in ("frac","stream_us_per_launch","traffic","rocprof_kernel_us","evidence")})
for k in ("f64","hbm_resident","fused","weak","exchange_every_tick"):
    if k in o: print(" ",k, json.dumps(o[k])[:400])
PY
done
cat $O/summary.txt
