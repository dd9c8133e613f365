cd $GRAFT_REPO_ROOT
for side in 1024 2048 4096; do for vec in 1 2 4; do for gyro in 0 2; do
  echo -n "f32 side=$side vec=$vec gyro=$gyro : "
  DMX_VEC=$vec python bench.py --steps 300 --warmup 30 --no-cpu-baseline --side $side --gyro $gyro 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us  %.0f GB/s  %.2f Gbs/s'%(d['roofline']['kernel_us'], d['roofline']['achieved'], d['value']/1e9))"
done; done; done
for side in 1024 2048; do for vec in 1 2; do for gyro in 0 2; do
  echo -n "f64 side=$side vec=$vec gyro=$gyro : "
  DMX_VEC=$vec python bench.py --dtype f64 --steps 300 --warmup 30 --no-cpu-baseline --side $side --gyro $gyro 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us  %.0f GB/s  %.2f Gbs/s'%(d['roofline']['kernel_us'], d['roofline']['achieved'], d['value']/1e9))"
done; done; done
