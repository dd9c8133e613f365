R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --config 5 --no-cpu-baseline --no-extras --steps 200 > $O/bench_c5.json 2> $O/bench_c5.err; python -c "
import json; o=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('config5', o['ms_per_step'], o['value'])"
python scripts/time_floor.py 2>&1 | grep -v amdgpu.ids | tee $O/floor.txt
bash scripts/profile_config5.sh 2>&1 | grep "solve_singles\|ex_narrow_convex" 
python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles.txt
