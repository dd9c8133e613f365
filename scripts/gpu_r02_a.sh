# round 2, first GPU trip: the whole -m gpu suite on the new ping-pong / lazy-chunk loop, then bench.py at the driver's
# flags, at the default flags, as a 2-rank rehearsal from a plain invocation, and at the strong-scaled slab size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?" | tee -a $O/summary.txt
python bench.py --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err; echo "bench1000 rc=$?" | tee -a $O/summary.txt
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_g2_20.json 2> $O/bench_g2_20.err; echo "bench g2/20 rc=$?" | tee -a $O/summary.txt
python bench.py --gpus 2 --steps 300 --warmup 50 > $O/bench_g2_300.json 2> $O/bench_g2_300.err; echo "bench g2/300 rc=$?" | tee -a $O/summary.txt
for st in 20 1000; do
python bench.py --side 364 --steps $st --warmup 20 --no-extras --no-cpu-baseline > $O/bench_small_$st.json 2> $O/bench_small_$st.err; echo "small $st rc=$?" | tee -a $O/summary.txt
done
DMX_LAZY_CHUNKS=0 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_20_sync.json 2> $O/bench_20_sync.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    o=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("unreadable", e); raise SystemExit
print({k:o[k] for k in ("value","ms_per_step","n_gpus","scaling")}, o.get("timing"))
print(" roofline", {k:o["roofline"][k] for k in ("frac","stream_us_per_launch","traffic")})
for k in ("f64","hbm_resident","fused","weak","exchange_every_tick"):
    if k in o: print(" ",k, {kk:vv for kk,vv in o[k].items() if kk in ("value","ms_per_step","blocks","error")}, o[k].get("roofline",{}).get("frac"))
print(" ", o["config"]["parallelism"][:300])
print(" ", o["config"]["collide"][-160:])
PY
done
