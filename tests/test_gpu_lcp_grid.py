"""dWorldStep at the reference's own scale (/root/reference/src/main.c:213: dWorldStep at 1/120 s; inc/body.h:6: up to 512
bodies, which pile into one island of thousands of rows): the grid-wide exact solve of large islands (csrc/dmx_lcp.hip --
unbounded rows eliminated once per tick by a blocked Cholesky on the matrix cores, block principal pivoting on the Schur
complement, active sets carried from tick to tick) against the oracle's exact stepper (oracle/orc_step.c exact_lcp), through
the ODE C API exactly as the reference calls it (tests/harness/ode_tick_harness.c).  Tolerance: north_star's 1e-5 relative on
positions and rotation matrices; the two solvers take different routes to the same unique solution."""
import os
import re
import subprocess

import numpy as np
import pytest

from __graft_entry__ import load_package
from test_ode_compat import _build_harness, _oracle_poses, _rel, _scene_text

pkg = load_package()
pytestmark = pytest.mark.gpu


def _run(exe, text, env=None, stepper="exact"):
    """-> (poses, stderr); DMX_LCP_REPORT makes the library print the grid solve's counters when the world is destroyed"""
    p = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=900,
                       env={**os.environ, "HARNESS_STEPPER": stepper, "DMX_LCP_REPORT": "1", **(env or {})})
    assert p.returncode == 0, p.stderr[-2000:]
    return np.array([[float(v) for v in line.split()] for line in p.stdout.strip().splitlines()]), p.stderr


def _stats(err):
    m = re.search(r"lcp grid: solves=(\d+) rounds=(\d+) max_rounds=(\d+) last_m=(\d+) last_nu=(\d+) last_nbd=(\d+) single=(\d+) fallback=(\d+)", err)
    assert m, err[-500:]
    return dict(zip(("solves", "rounds", "max_rounds", "last_m", "last_nu", "last_nbd", "single", "fallback"), map(int, m.groups())))


def _pyramid(base=4, side=1.0, gap=0.0):
    """a pyramid of unit boxes on the reference's floor (top at y = 0.5): layers of base^2, (base-1)^2, ... 1 boxes, each box
    resting on the four below it -- one island from the first tick, every box-box pair a face contact of four points"""
    out = []
    for layer in range(base):
        k = base - layer
        for i in range(k):
            for j in range(k):
                x = (i - (k - 1) / 2.0) * (side + gap)
                z = (j - (k - 1) / 2.0) * (side + gap)
                out.append((2, (side, side, side), (x, 0.5 + side / 2 + layer * side, z)))
    return out


@pytest.mark.parametrize("single", [False, True])
def test_pyramid_of_thirty_boxes_one_island_of_hundreds_of_rows(tmp_path, single):
    """30 boxes, one island of ~1 300 rows from the first tick (two thirds of them friction rows that never clamp): every tick is
    one grid solve; poses agree with the oracle's exact stepper"""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()[:1]
    bodies = _pyramid(4)
    dt, steps = 1.0 / 120.0, 4
    exe = _build_harness(str(tmp_path), single)
    text = _scene_text(dt, steps, False, statics, bodies)
    got, err = _run(exe, text)
    st = _stats(err)
    assert st["solves"] == steps and st["last_m"] >= 600 and st["last_nu"] == 2 * st["last_nbd"] and st["fallback"] == 0
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact=True)
    assert 3 * ow.n_contacts() == st["last_m"]
    assert _rel(got.astype(ref.dtype), ref) <= 1e-5
    assert np.max(np.abs(got[:, 13] - np.array([b[2][1] for b in bodies]))) < 2e-3        # and the pyramid stands


@pytest.mark.parametrize("single", [False, True])
def test_pen_of_400_bodies_settled_then_dworldstep(tmp_path, single):
    """the reference's pen with 400 of the spawner's boxes and spheres: QuickStep while they come down (400 ticks), then the
    reference's own call for 4 ticks -- islands of every size at once (lane-per-island, workgroup and grid solves in one tick)"""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(400, seed=7, y_range=(1.2, 12.0))
    dt, settle, steps = 1.0 / 120.0, 400, 404
    exe = _build_harness(str(tmp_path), single)
    text = _scene_text(dt, steps, False, statics, bodies)
    got, err = _run(exe, text, env={"HARNESS_EXACT_AFTER": str(settle)})
    st = _stats(err)
    assert st["solves"] >= 1 and st["last_m"] >= 192          # (ticks whose islands all fit a workgroup's LDS have no grid solve)
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact_after=settle)
    assert ow.n_contacts() > 300
    assert _rel(got.astype(ref.dtype), ref) <= 1e-5
    if not single:
        # the active set carried from tick to tick changes the route, not the destination (the solution is unique); nor does
        # switching the volatile-row solve in LDS off (every pivoting round then refactors the whole free block)
        cold, _ = _run(exe, text, env={"HARNESS_EXACT_AFTER": str(settle), "DMX_LCP_WARM": "0"})
        assert _rel(cold, ref) <= 1e-5
        plain, _ = _run(exe, text, env={"HARNESS_EXACT_AFTER": str(settle), "DMX_LCP_LEVEL2": "0"})
        assert _rel(plain, ref) <= 1e-5


@pytest.mark.parametrize("single,seed,n", [(False, 3, 250), (True, 11, 250), (False, 29, 320), (True, 5, 180)])
def test_other_piles_other_seeds(tmp_path, single, seed, n):
    """the same sequence on other draws of the spawner (other shapes, sizes and piles: other active sets, other islands at the
    boundary between the workgroup and the grid solve): settle with QuickStep, then five ticks of dWorldStep against the oracle"""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(n, seed=seed, y_range=(1.2, 10.0))
    dt, settle, steps = 1.0 / 120.0, 360, 365
    exe = _build_harness(str(tmp_path), single)
    text = _scene_text(dt, steps, False, statics, bodies)
    got, err = _run(exe, text, env={"HARNESS_EXACT_AFTER": str(settle)})
    ref, ow = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact_after=settle)
    assert ow.n_contacts() > n // 2
    assert _rel(got.astype(ref.dtype), ref) <= 1e-5


def test_every_island_through_the_grid_solve_in_the_small_pen(tmp_path):
    """DMX_LCP_GRID_ROWS=1: islands of 3 rows and up all take the grid path (one unbounded tile, one bounded tile, padding
    everywhere) -- 24 bodies dropping into the pen, 120 ticks of dWorldStep"""
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(24, seed=21, y_range=(1.2, 5.0))
    dt, steps = 1.0 / 120.0, 120
    exe = _build_harness(str(tmp_path), False)
    got, err = _run(exe, _scene_text(dt, steps, False, statics, bodies), env={"DMX_LCP_GRID_ROWS": "1"})
    assert _stats(err)["solves"] > 100
    ref, ow = _oracle_poses("float64", dt, steps, False, statics, bodies, exact=True)
    assert ow.n_contacts() > 24
    assert _rel(got, ref) <= 1e-5


def test_single_flips_only_reach_the_same_solution(tmp_path):
    """DMX_LCP_MURTY=1: every pivoting round flips one row (Murty's rule, the fallback block pivoting takes when its violation
    count stalls) -- slow, finite, same answer"""
    statics = pkg.scenes.reference_map()
    bodies = pkg.scenes.reference_spawn(24, seed=21, y_range=(1.2, 5.0))
    dt, steps = 1.0 / 120.0, 100
    exe = _build_harness(str(tmp_path), False)
    got, err = _run(exe, _scene_text(dt, steps, False, statics, bodies),
                    env={"DMX_LCP_MURTY": "1", "DMX_LCP_WARM": "0", "DMX_LCP_GRID_ROWS": "1"})
    st = _stats(err)
    assert st["single"] > 0 and st["single"] == st["rounds"] - st["solves"]
    ref, _ = _oracle_poses("float64", dt, steps, False, statics, bodies, exact=True)
    assert _rel(got, ref) <= 1e-5


@pytest.mark.parametrize("single", [False, True])
def test_island_above_the_row_limit_is_stepped_by_the_sor_with_a_warning(tmp_path, single):
    """DMX_MAX_EXACT_ROWS=100: the pyramid's island (hundreds of rows) is over the limit at every tick, so dWorldStep warns once
    and steps with QuickStep's sweeps -- bit for bit the oracle's QuickStep"""
    dtype = "float32" if single else "float64"
    statics = pkg.scenes.reference_map()[:1]
    bodies = _pyramid(4)
    dt, steps = 1.0 / 120.0, 20
    exe = _build_harness(str(tmp_path), single)
    got, err = _run(exe, _scene_text(dt, steps, False, statics, bodies), env={"DMX_MAX_EXACT_ROWS": "100"})
    assert err.count("exceeds the exact solver's limit") == 1
    ref, _ = _oracle_poses(dtype, dt, steps, False, statics, bodies, exact=False)
    assert np.array_equal(got.astype(ref.dtype), ref), np.abs(got - ref).max()
