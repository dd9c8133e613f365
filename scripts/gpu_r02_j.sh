R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
DMX_SMALL_EXACT=0 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_legacy.log 2>&1; echo "pytest legacy-path fuzz rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest_legacy.log
DMX_SMALL_EXACT=2 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_small_always.log 2>&1; echo "pytest small-path-whenever-it-fits rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest_small_always.log
DMX_EXS_TIMING=1 python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids | tee $O/config1.txt
python scripts/time_piles_small.py 2>&1 | grep -v amdgpu.ids | tee $O/piles_small.txt
python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles.txt
