R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python scripts/time_config1.py 2>&1 | grep -v amdgpu.ids | tee $O/config1.txt
bash scripts/profile_config1.sh > $O/prof_config1.log 2>&1; cp $R/gpurun_out/prof_config1/busy.txt $O/r02_config1_trace.txt; head -9 $O/r02_config1_trace.txt
