cd $GRAFT_REPO_ROOT
for mw in 0 6 8; do
  echo -n "f32 vec=1 minw=$mw : "
  DMX_VEC=1 DMX_MIN_WAVES=$mw python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['roofline']['kernel_us']))"
done
for tune in 1 2 3; do
  echo -n "f32 vec=1 tune=$tune : "
  DMX_TUNE=$tune python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['roofline']['kernel_us']))"
done
for tune in 0 3; do
  echo -n "f32 side 2048 tune=$tune : "
  DMX_TUNE=$tune python bench.py --side 2048 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['roofline']['kernel_us']))"
done
