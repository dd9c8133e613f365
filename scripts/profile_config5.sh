# where a configs[4] tick goes (16 384 teapot hulls on a static box floor): kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_config5; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --config 5 --no-cpu-baseline --no-extras --steps 200 --warmup 20 > $O/log.txt 2>&1; tail -c 600 $O/log.txt
python3 $R/scripts/trace_busy.py $O/k 340 | tee $O/busy.txt
