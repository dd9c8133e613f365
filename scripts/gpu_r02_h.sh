R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
timeout -k 10 120 scripts/_build/ubench_fma_chain > $O/fma_chain.txt 2>&1; cat $O/fma_chain.txt
timeout -k 10 300 python scripts/time_floor.py > $O/floor.txt 2>&1; grep -v "^libode" $O/floor.txt
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" | tee -a $O/summary.txt
python bench.py --config 3 --no-cpu-baseline --steps 300 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" | tee -a $O/summary.txt
python bench.py --config 5 --no-cpu-baseline --steps 200 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" | tee -a $O/summary.txt
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_g2.json 2> $O/bench_g2.err; echo "g2 rc=$?" | tee -a $O/summary.txt
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("unreadable", e); raise SystemExit
print({k:o[k] for k in ("value","ms_per_step","n_gpus","scaling")}, o["timing"]["blocks"])
print(" roofline", {k:o["roofline"].get(k) for k in ("frac","stream_us_per_launch","traffic","evidence")})
for k in ("f64","hbm_resident","fused","weak","exchange_every_tick","cpu_baseline"):
    if k in o: print(" ",k, {kk:vv for kk,vv in o[k].items() if kk in ("value","ms_per_step","blocks","error")}, o[k].get("roofline",{}).get("frac") if isinstance(o[k].get("roofline"),dict) else "")
print(" ", o["config"]["workload"][:200])
PY
done
