cd $GRAFT_REPO_ROOT
gcc -O1 -Iinclude tests/harness/ode_tick_harness.c -o /tmp/harness_d -Lrl-ode-physics_amd -lode_mi355 -Wl,-rpath,$PWD/rl-ode-physics_amd -lm
python3 - <<'PY'
import subprocess, time, sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_ode_compat.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
for n, steps in ((48, 600), (200, 600), (500, 600)):
    text = t._scene_text(1.0/120.0, steps, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.5, 30.0)))
    t0 = time.perf_counter(); p = subprocess.run(["/tmp/harness_d"], input=text, capture_output=True, text=True, env={**__import__("os").environ, "HARNESS_STEPPER": __import__("os").environ.get("HARNESS_STEPPER", "quick")}); dt = time.perf_counter() - t0
    text0 = t._scene_text(1.0/120.0, 1, False, pkg.scenes.reference_map(), pkg.scenes.reference_spawn(n, seed=7, y_range=(1.5, 30.0)))
    t0 = time.perf_counter(); subprocess.run(["/tmp/harness_d"], input=text0, capture_output=True, text=True, env={**__import__("os").environ, "HARNESS_STEPPER": __import__("os").environ.get("HARNESS_STEPPER", "quick")}); d0 = time.perf_counter() - t0
    print(f"reference scene, {n} bodies: {(dt-d0)/steps*1e3:.3f} ms per tick (dSpaceCollide + dWorldStep + dJointGroupEmpty), rc={p.returncode}")
PY
