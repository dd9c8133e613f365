cd $GRAFT_REPO_ROOT
for dt in f32 f64; do for v in 1 2; do
  echo -n "$dt variant=$v : "
  DMX_MIN_WAVES=$v python bench.py --config 3 --dtype $dt --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us  %.2f Gbs/s'%(d['roofline']['kernel_us'], d['value']/1e9))"
done; done
