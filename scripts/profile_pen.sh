# where a tick of the reference's pen scene goes (batch path): kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_pen; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/scripts/time_pen.py ${1:-400} > $O/log.txt 2>&1; grep pen $O/log.txt
python3 $R/scripts/trace_busy.py $O/k 600 | tee $O/busy.txt
