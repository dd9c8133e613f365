R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_hp; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/scripts/time_hulls_pairs.py ${1:-4.5} > $O/log.txt 2>&1; grep teapot $O/log.txt
python3 $R/scripts/trace_busy.py $O/k 360 | head -30
