# where a configs[0] tick goes (1024 boxes dropped on the plane, batch path): kernel trace, busy time against span
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_config1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_c1.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from __graft_entry__ import load_package
pkg = load_package()
scene = pkg.scenes.config1().astype("float32")
w = pkg.BatchWorld(scene.n, dtype="float32"); w.load_scene(scene)
t0 = time.perf_counter(); w.step(1/60, 610); w.synchronize(); dt = time.perf_counter() - t0
print("config1 f32: %.1f us/tick" % (dt / 610 * 1e6), w.collision_stats())
PY
python3 /tmp/run_c1.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 /tmp/run_c1.py > $O/log.txt 2>&1; tail -2 $O/log.txt
python3 $R/scripts/trace_busy.py $O/k 610 | tee $O/busy.txt
