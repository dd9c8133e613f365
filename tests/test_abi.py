"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/*.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import load_package, ROOT

pkg = load_package()


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmx[A-Z]\w*|d[A-Z]\w*)\s*\(", src)))


def test_library_loads_and_exports_batch_abi():
    lib = pkg._lib.load()
    names = _declared("dmx_batch.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dmx_batch.h but not exported"
    assert sorted(pkg._lib.BATCH_SYMBOLS) == names
    assert b"gfx950" in lib.dmxVersion()


def test_every_header_symbol_is_exported():
    lib = pkg._lib.load()
    headers = sorted(h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h"))
    assert headers == ["dmx_batch.h", "dmx_hull.h", "dmx_shard.h"]
    assert _declared("dmx_hull.h") == ["dmxHullBuild", "dmxHullPlanes", "dmxObjReadVertices"]
    assert _declared("dmx_shard.h") == sorted(pkg._lib.SHARD_SYMBOLS)
    for h in headers:
        for n in _declared(h):
            assert hasattr(lib, n), f"{n} declared in include/{h} but not exported"


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product refuses to run instead of falling back."""
    lib = pkg._lib.load()
    if lib.dmxDeviceCount() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.dmxBatchCreate(C.byref(h), 16, 0, 0) == -1      # DMX_ENODEVICE
    assert not h.value
    with pytest.raises(pkg.batch.DmxError):
        pkg.BatchWorld(16)


def test_product_does_not_reference_oracle():
    """The shipped library and package never link / import anything under oracle/."""
    import subprocess
    out = subprocess.run(["ldd", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rl-ode-physics_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)


def _build_c_client(tmp_path):
    import subprocess
    pkg_dir = os.path.join(ROOT, "rl-ode-physics_amd")
    exe = str(tmp_path / "batch_abi_check")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "harness", "batch_abi_check.c"), "-o", exe,
                    "-L" + pkg_dir, "-lode_mi355", "-Wl,-rpath," + pkg_dir, "-lm"], check=True)
    return exe


def test_headers_are_plain_c_and_a_c_client_links(tmp_path):
    """include/dmx_batch.h and include/dmx_hull.h compile as strict C99 and a C program links against the library; on a
    box without a GPU the program reports the missing device and stops (no fallback)."""
    import subprocess
    exe = _build_c_client(tmp_path)
    lib = pkg._lib.load()
    if lib.dmxDeviceCount() > 0:
        pytest.skip("GPU present: the run is checked by the gpu-marked test")
    p = subprocess.run([exe, "256", "10"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_c_client_drops_boxes_on_the_plane(tmp_path):
    import subprocess
    exe = _build_c_client(tmp_path)
    p = subprocess.run([exe, "4096", "240"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "4096 boxes, 240 ticks: resting heights" in p.stdout
