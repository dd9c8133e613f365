cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('%-28s %8.2f us/tick  %7.2f G body-steps/s  alg %5.0f GB/s (%.2f of 8 TB/s)'%(sys.argv[1], d['ms_per_step']*1e3, d['value']/1e9, r['achieved'], r['frac']), d.get('cpu_baseline',{}).get('value',''))" "$1"; }
python bench.py --steps 1000 --warmup 100 2>/dev/null | p "configs[1] f32 1M"
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-body-collisions 2>/dev/null | p "  same, collide proof off"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --side 2048 2>/dev/null | p "configs[1] f32 4M"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --side 4096 2>/dev/null | p "configs[1] f32 16M"
python bench.py --steps 500 --warmup 50 --no-cpu-baseline --dtype f64 2>/dev/null | p "configs[1] f64 1M"
python bench.py --config 3 --steps 600 --warmup 0 --cpu-seconds 8 2>/dev/null | p "configs[2] f32 262k"
python bench.py --config 3 --steps 300 --warmup 0 --no-cpu-baseline --dtype f64 2>/dev/null | p "configs[2] f64 262k"
