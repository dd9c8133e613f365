# configs[4] (16 384 teapot hulls on the static box floor): kernel stats, a 64-body tile per workgroup (default) / a wavefront per body (DMX_HULL_WAVE_PER_BODY=1).  usage: profile_config5_r04.sh [TAG=r04_c5]
TAG=${1:-r04_c5}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for MODE in tile waves; do
  O=$R/gpurun_out/prof_${TAG}_${MODE}; mkdir -p $O
  if [ "$MODE" = "waves" ]; then export DMX_HULL_WAVE_PER_BODY=1; else unset DMX_HULL_WAVE_PER_BODY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 $R/bench.py --config 5 --no-extras --no-cpu-baseline --steps 200 --warmup 20 > $O/bench.json 2> $O/err.txt
  F="$(ls -t $O/k/*/*kernel_stats.csv | head -1)"; [ -n "$F" ] || { echo "no stats file"; tail -5 $O/err.txt; exit 1; }
  echo "== $MODE"; python3 -c "
import json; o=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', o['ms_per_step'])"
  head -6 "$F" | cut -c1-150
  cp "$F" $R/gpurun_out/${TAG}_f32_${MODE}_kernel_stats.csv
done
