R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python scripts/time_piles.py 2>&1 | grep -v amdgpu.ids | tee $O/piles.txt
python scripts/time_piles_small.py 384 2>&1 | grep -v amdgpu.ids | tee -a $O/piles.txt
python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids | tee $O/config1.txt
bash scripts/time_compat_exact.sh 2>&1 | grep "reference scene" | tee $O/compat.txt
