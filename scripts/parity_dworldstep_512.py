"""dWorldStep at the reference's maximum: 512 bodies (MAX_BODIES, inc/body.h:6) in the pen, settled by 900 QuickStep ticks, then 3 ticks of the
reference's own call (main.c:213) through the ODE API -- the device's grid-wide exact solve against the oracle's exact stepper (which needs seconds
per tick here: this is a script with its output under profiles/, not a test of the 100-second suite).  Both precisions; north_star's 1e-5."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
from test_ode_compat import _build_harness, _oracle_poses, _rel, _scene_text, pkg  # noqa: E402
from test_gpu_lcp_grid import _run, _stats  # noqa: E402


def main():
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(512, seed=7, y_range=(1.2, 12.0))
    dt, settle, steps = 1.0 / 120.0, 900, 903
    tmp = tempfile.mkdtemp()
    for single in (False, True):
        dtype = "float32" if single else "float64"
        exe = _build_harness(tmp, single)
        text = _scene_text(dt, steps, False, statics, bodies)
        t0 = time.time()
        got, err = _run(exe, text, env={"HARNESS_EXACT_AFTER": str(settle)})
        t1 = time.time()
        ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact_after=settle)
        t2 = time.time()
        st = _stats(err)
        rel = _rel(got.astype(ref.dtype), ref)
        print(f"{dtype}: 512 bodies, {settle} QuickStep ticks + {steps - settle} dWorldStep ticks; grid solves {st['solves']}, last island {st['last_m']} rows "
              f"({st['last_nu']} never clamp, {st['last_nbd']} bounded), pivoting rounds {st['rounds']}; contacts in the oracle's last tick {ow.n_contacts()}; "
              f"max relative difference of the 512 x 16 transform entries {rel:.3e} ({'OK' if rel <= 1e-5 else 'FAIL'} at 1e-5); "
              f"device run {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s", flush=True)
        assert rel <= 1e-5


if __name__ == "__main__":
    main()
