cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 f32: %.2f us'%(d['roofline']['kernel_us']))"
python bench.py --dtype f64 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 f64: %.2f us'%(d['roofline']['kernel_us']))"
python bench.py --config 3 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 f32: %.2f us'%(d['roofline']['kernel_us']))"; }
echo "== contract off"; run
cd rl-ode-physics_amd/csrc && touch dmx_kernels.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function" >/dev/null 2>&1; cd ../..
echo "== contract fast (parity broken; timing only)"; run
