R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do
python scripts/time_config1.py 2>&1 | grep "float32 config1" | cut -c1-62 | sed 's/^/spin: /'
DMX_RECORD_SPIN=0 python scripts/time_config1.py 2>&1 | grep "float32 config1" | cut -c1-62 | sed 's/^/sync: /'
done
