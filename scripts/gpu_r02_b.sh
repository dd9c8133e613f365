R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
timeout -k 10 300 python scripts/ubench_small_slab.py > $O/small_slab.txt 2>&1; cat $O/small_slab.txt
timeout -k 10 600 python scripts/sweep_hbm_resident.py 4096 float32 100 > $O/hbm_sweep_f32.txt 2>&1; cat $O/hbm_sweep_f32.txt
