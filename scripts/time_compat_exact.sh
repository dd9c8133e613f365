# the reference's own call sequence (dSpaceCollide + dWorldStep (exact LCP) + dJointGroupEmpty, h = 1/120) in its pen, by body count
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/compat_exact; mkdir -p $O
gcc -O1 -Iinclude tests/harness/ode_tick_harness.c -o /tmp/harness_d -Lrl-ode-physics_amd -lode_mi355 -Wl,-rpath,$PWD/rl-ode-physics_amd -lm
python3 - <<'PY'
import subprocess, time, sys, os
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_ode_compat.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
def run(n, steps, stepper):
    text = t._scene_text(1.0/120.0, steps, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.5, 30.0)))
    t0 = time.perf_counter(); p = subprocess.run(["/tmp/harness_d"], input=text, capture_output=True, text=True, env={**os.environ, "HARNESS_STEPPER": stepper}); return time.perf_counter() - t0, p.returncode
for n, steps in ((16, 1200), (48, 1200), (200, 600), (500, 300)):
    open(f"/tmp/scene{n}.txt", "w").write(t._scene_text(1.0/120.0, steps, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.5, 30.0))))
    for stepper in ("quick", "exact"):
        d0, _ = run(n, 1, stepper); dt, rc = run(n, steps, stepper)
        print(f"reference scene, {n:4d} bodies, {stepper:5s}: {(dt-d0)/steps*1e3:8.3f} ms per tick   rc={rc}", flush=True)
PY
cd /tmp && export TMPDIR=/tmp
for n in 48 200; do
HARNESS_STEPPER=exact rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$n -- /tmp/harness_d < /tmp/scene$n.txt > /dev/null 2> $O/err$n.txt
steps=600; [ $n = 48 ] && steps=1200
python3 $R/scripts/trace_busy.py $O/k$n $steps | tee $O/busy$n.txt
done
