#!/bin/bash
# rocprofv3 kernel stats of the static-floor scenes (run on the GPU box: bash scripts/prof_static_r03.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for sc in floor hulls_floor hulls_plane; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$sc -- python3 $R/scripts/time_floor.py $sc > $O/prof_$sc.log 2>&1 || exit 1
  cp "$(ls $O/prof_$sc/*/*kernel_stats.csv | head -1)" $O/r03_${sc}_kernel_stats.csv || exit 1
  cut -d, -f1-4 $O/r03_${sc}_kernel_stats.csv | cut -c1-150 | head -8
  grep "ms/tick" $O/prof_$sc.log
done
