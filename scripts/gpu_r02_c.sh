R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
timeout -k 10 600 python scripts/ab_hbm_paths.py 16 > $O/ab_hbm_paths.txt 2>&1; cat $O/ab_hbm_paths.txt
