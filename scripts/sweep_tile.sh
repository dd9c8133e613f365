# integrate_free on the tiled slab: bodies per lane (DMX_VEC) x body count, f32 and f64
cd $GRAFT_REPO_ROOT
for side in 1024 2048 4096; do for vec in 1 2 4; do
  echo -n "f32 side=$side vec=$vec : "
  DMX_VEC=$vec python bench.py --side $side --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us kernel, %.2f us/tick, frac %.3f'%(d['roofline']['kernel_us'], d['ms_per_step']*1e3, d['roofline']['frac']))"
done; done
for side in 1024 2048; do for vec in 1 2; do
  echo -n "f64 side=$side vec=$vec : "
  DMX_VEC=$vec python bench.py --dtype f64 --side $side --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us kernel, %.2f us/tick, frac %.3f'%(d['roofline']['kernel_us'], d['ms_per_step']*1e3, d['roofline']['frac']))"
done; done
