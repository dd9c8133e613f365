"""The reference's own tick -- dSpaceCollide + dWorldStep(1/120) + dJointGroupEmpty through the ODE API (main.c:211-215) -- timed
on the reference's pen at several body counts: QuickStep while the bodies come down, then dWorldStep for the timed ticks.
usage: python scripts/time_dworldstep.py [--bodies 48,200,400,512] [--settle 600] [--ticks 60] [--single] [--env K=V ...]"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_ode_compat import _build_harness, _scene_text, pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", default="48,200,400,512")
    ap.add_argument("--settle", type=int, default=600)
    ap.add_argument("--ticks", type=int, default=60)
    ap.add_argument("--single", action="store_true")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--stderr-to", default="", help="append the harness's stderr (the library's DMX_LCP_TRACE / REPORT lines) to this file")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp()
    exe = _build_harness(tmp, a.single)
    extra = dict(kv.split("=", 1) for kv in a.env)
    for n in [int(v) for v in a.bodies.split(",")]:
        statics = pkg.scenes.reference_map()
        bodies = pkg.scenes.reference_spawn(n, seed=a.seed, y_range=(1.2, 12.0))
        steps = a.settle + a.ticks
        env = {**os.environ, "HARNESS_STEPPER": "exact", "HARNESS_EXACT_AFTER": str(a.settle), "HARNESS_TIME_FROM": str(a.settle + 4),
               "DMX_LCP_REPORT": "1", **extra}
        p = subprocess.run([exe], input=_scene_text(1.0 / 120.0, steps, False, statics, bodies), capture_output=True, text=True, env=env)
        if a.stderr_to:
            open(a.stderr_to, "a").write(p.stderr)
        if p.returncode != 0:
            print(n, "FAILED", p.stderr[-800:])
            continue
        t = re.search(r"harness: (.*)", p.stderr)
        g = re.search(r"lcp grid: (.*)", p.stderr)
        print(f"bodies={n} {'f32' if a.single else 'f64'} {t.group(1) if t else ''} | {g.group(1) if g else 'no grid solve'}", flush=True)


if __name__ == "__main__":
    main()
