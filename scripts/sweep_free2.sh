cd $GRAFT_REPO_ROOT
for cfg in "2 4" "2 5" "2 6" "1 0" "1 6" "1 8"; do set -- $cfg
  echo -n "f32 vec=$1 minw=$2 : "
  DMX_VEC=$1 DMX_MIN_WAVES=$2 python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['roofline']['kernel_us']))"
done
for cfg in "2 0" "1 0" "1 6"; do set -- $cfg
  echo -n "f64 vec=$1 minw=$2 : "
  DMX_VEC=$1 DMX_MIN_WAVES=$2 python bench.py --dtype f64 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us'%(d['roofline']['kernel_us']))"
done
