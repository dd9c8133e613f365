"""One large island: the reference's own scene shape (static floor + walls, a pile of boxes and spheres: main.c:115-121, 502-521)
at sizes where the pile is ONE island of more rows than a wavefront holds.  Covers, bit for bit against the oracle:
  * level schedules built by a workgroup (dmx_exact.hip: levels_coop) in both pipelines;
  * the island's rows kept in registers by a workgroup for the sweeps (dmx_islands.hip: wg_island_sweeps, 2 ... 12 rows a thread);
  * past 3 072 rows, the streamed form (schedule in LDS, rows fetched two level steps ahead)."""
import numpy as np
import pytest

from tests.test_gpu_parity import H, _compare, _oracle_with_map, _orc, pkg

pytestmark = pytest.mark.gpu


def _pen(n_spawn, dtype, seed=7, y_range=(3.0, 12.0)):
    spawn = pkg.scenes.reference_spawn(n_spawn, seed=seed, y_range=y_range)
    spawn.sort(key=lambda s: -s[0])                      # boxes (type 2) first, spheres behind them
    n = len(spawn)
    nb = sum(1 for s in spawn if s[0] == pkg.scenes.GEOM_BOX)
    sc = pkg.scenes.Scene(np.array([s[2] for s in spawn], float), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 3)), np.zeros((n, 3)),
                          np.ones((n, 1)), np.ones((n, 3)), np.array([s[1] for s in spawn], float),
                          np.array([s[0] for s in spawn], np.uint8), None).astype(dtype)
    return sc, nb


def _run_both(n_spawn, dtype, steps, pipeline=None):
    sc, nb = _pen(n_spawn, dtype)
    boxes = pkg.scenes.reference_map()
    ow = _oracle_with_map(_orc(dtype), sc, boxes, spheres_from=nb)
    most = 0
    for _ in range(steps):
        ow.tick(H)
        most = max(most, ow.n_contacts())
    w = pkg.BatchWorld(sc.n, dtype=dtype)
    w.load_scene(sc)
    w.set_static_boxes(boxes)
    if pipeline is not None:
        w.set_exact_pipeline(pipeline)
    w.step(H, steps)
    _compare(w.state(), ow.state())
    st = w.collision_stats()
    w.close()
    return most, st


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_pile_of_four_hundred_in_the_pen(dtype):
    """400 bodies: one island of ~390 bodies and 1 400-2 000 rows once they are down -- 8 rows a thread in f32; in f64 (4 a thread,
    1 024 rows) the later ticks stream"""
    most, st = _run_both(400, dtype, 260)
    assert most * 3 > 1024 and st["pair_ticks"] > 100


def test_pile_of_the_reference_s_maximum():
    """512 bodies -- MAX_BODIES in the reference (inc/body.h:6): 2 000-2 600 rows once they are down, 12 rows a thread in f32"""
    most, st = _run_both(512, "float32", 280)
    assert most * 3 > 1536


def test_pile_of_a_thousand_streams_its_rows():
    """1 000 bodies: more rows than a workgroup's registers hold"""
    most, st = _run_both(1000, "float32", 200)
    assert most * 3 > 2048


@pytest.mark.parametrize("pipeline", [1, 2])           # DMX_EXACT_STAGED, DMX_EXACT_ONE_WORKGROUP
def test_pile_in_either_pipeline(pipeline):
    """the same pile with the bookkeeping as a stage per launch and as the two one-workgroup kernels (the workgroup-built level
    schedule runs inside ex_small_back there)"""
    most, st = _run_both(160, "float32", 240, pipeline=pipeline)
    assert most * 3 > 256
