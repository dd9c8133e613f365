# where a compat-path tick goes: kernel time (rocprofv3) against wall time, reference scene with 500 bodies
cd $GRAFT_REPO_ROOT
gcc -O1 -Iinclude tests/harness/ode_tick_harness.c -o /tmp/harness_d -Lrl-ode-physics_amd -lode_mi355 -Wl,-rpath,$PWD/rl-ode-physics_amd -lm
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_ode_compat.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
open("/tmp/scene500.txt", "w").write(t._scene_text(1.0/120.0, 600, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(500, seed=7, y_range=(1.5, 30.0))))
PY
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_compat; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
/usr/bin/time -v /tmp/harness_d < /tmp/scene500.txt > /dev/null 2> $O/time.txt; grep -E "Elapsed|User time|System time" $O/time.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- /tmp/harness_d < /tmp/scene500.txt > /dev/null 2> $O/err.txt
cat "$(ls -t $O/k/*/*kernel_stats.csv | head -1)" | cut -c1-160
