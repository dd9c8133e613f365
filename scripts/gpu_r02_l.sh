R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
DMX_PACK_MIN_ISLANDS=1 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_ode_compat.py -m gpu -x -q > $O/pytest_pack_always.log 2>&1; echo "pytest packed-always rc=$?"; tail -3 $O/pytest_pack_always.log
DMX_PACK_MIN_ISLANDS=1 DMX_PACK_ISLANDS=8 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_pack8.log 2>&1; echo "pytest packed-8 rc=$?"; tail -2 $O/pytest_pack8.log
python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles_packed.txt
DMX_PACK_ISLANDS=1 python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles_unpacked.txt
DMX_PACK_ISLANDS=8 python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles_packed8.txt
DMX_PACK_ISLANDS=2 python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles_packed2.txt
