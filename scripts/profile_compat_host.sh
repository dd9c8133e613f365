# the reference scene through the ODE API (dSpaceCollide + dWorldQuickStep + dJointGroupEmpty): host profile of the library
cd $GRAFT_REPO_ROOT
gcc -O1 -Iinclude tests/harness/ode_tick_harness.c -o /tmp/harness_d -Lrl-ode-physics_amd -lode_mi355 -Wl,-rpath,$PWD/rl-ode-physics_amd -lm
python3 - <<'PY'
import subprocess, time, sys, os
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_ode_compat.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
for n, steps in ((200, 600),):
    text = t._scene_text(1.0/120.0, steps, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.5, 30.0)))
    t0 = time.perf_counter()
    p = subprocess.run(["/tmp/harness_d"], input=text, capture_output=True, text=True, env={**os.environ, "HARNESS_STEPPER": "quick", "DMX_HOST_PROFILE": "1"})
    dt = time.perf_counter() - t0
    print(f"{n} bodies, {steps} ticks: {dt*1e3:.1f} ms wall in the process"); print(p.stderr[-1500:])
PY
