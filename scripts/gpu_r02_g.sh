R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
timeout -k 10 200 python scripts/time_config1.py > $O/config1.txt 2>&1; cat $O/config1.txt
timeout -k 10 300 python scripts/time_piles.py > $O/piles.txt 2>&1; grep "piles:" $O/piles.txt
timeout -k 10 300 python scripts/time_floor.py > $O/floor.txt 2>&1; cat $O/floor.txt
bash scripts/time_compat.sh > $O/compat.txt 2>&1; tail -3 $O/compat.txt
