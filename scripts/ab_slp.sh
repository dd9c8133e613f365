cd $GRAFT_REPO_ROOT
echo "== default flags"; bash scripts/sweep_plane.sh 2>&1 | grep "variant=2\|f64 variant=1"
DMX_VEC=4 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('free f32: %.2f us'%(d['roofline']['kernel_us']))"
cd rl-ode-physics_amd/csrc && touch dmx_kernels.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize" >/dev/null 2>&1; cd ../..
echo "== -fno-slp-vectorize"; bash scripts/sweep_plane.sh 2>&1 | grep "variant=2\|f64 variant=1"
DMX_VEC=4 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('free f32: %.2f us'%(d['roofline']['kernel_us']))"
