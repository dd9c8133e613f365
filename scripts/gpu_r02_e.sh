R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/piles16 -- python3 $R/scripts/time_piles_small.py 16 > $O/piles16.log 2>&1
cd $R; tail -2 $O/piles16.log
f=$(ls $O/piles16/*/*kernel_stats.csv | head -1); cut -d, -f1-7 $f | head -40
