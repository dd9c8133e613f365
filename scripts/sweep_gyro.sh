# how much of integrate_free's time is arithmetic: gyro mode (0 none, 1 explicit, 2 implicit) x bodies per lane
cd $GRAFT_REPO_ROOT
for side in 1024 2048; do for gyro in 0 1 2; do for vec in 1 2; do
  echo -n "f32 side=$side gyro=$gyro vec=$vec : "
  DMX_VEC=$vec python bench.py --side $side --gyro $gyro --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us kernel, frac %.3f'%(d['roofline']['kernel_us'], d['roofline']['frac']))"
done; done; done
