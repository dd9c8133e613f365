# configs[2] (262 144 boxes on the ground plane): step_plane by rocprof
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04_c3; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --config 3 --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $O/bench.json 2> $O/err.txt
F="$(ls -t $O/k/*/*kernel_stats.csv 2>/dev/null | head -1)"
if [ -n "$F" ]; then head -4 "$F" | cut -c1-160; cp "$F" $R/gpurun_out/r04_c3_f32_kernel_stats.csv; fi
python3 -c "
import json; o=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', o['ms_per_step'], 'mean', o['timing']['ms_per_step_mean'])"
