"""Boundary row (b): the reference's own src/main.c is COMPILED against this repository's include/ode/ode.h and
include/ode/common.h -- every call site's argument types (dReal rm[12] -> const dMatrix3 at main.c:709, the CollMask
enums -> unsigned long bits at main.c:724-725, dContact's member layout at main.c:676-687, skip = sizeof(dContact) at
main.c:678, the dNearCallback signature at main.c:212) are checked by the compiler, in both ODE precisions, with the
mismatches that matter promoted to errors.  raylib / raymath / rlgl / raygui / ENet are test-only prototype stubs under
tests/stubs (nothing is linked or run).  Then the object file's undefined d* symbols must all be exported by
libode_mi355.so: main.c links against the library as it stands.

/root/reference never travels to the GPU box: skipped when absent."""
import os
import subprocess

import pytest

from __graft_entry__ import ROOT, load_package

REF = "/root/reference"
MAIN_C = os.path.join(REF, "src", "main.c")
pytestmark = pytest.mark.skipif(not os.path.exists(MAIN_C), reason="the reference checkout is not on this machine")

FLAGS = ["-std=gnu99", "-Wall", "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
         "-Werror=implicit-function-declaration", "-Werror=implicit-int", "-Werror=return-type",
         "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + os.path.join(REF, "inc"), "-I" + os.path.join(ROOT, "include")]


@pytest.mark.parametrize("precision", ["dDOUBLE", "dSINGLE"])
def test_main_c_compiles_against_the_shim_headers(tmp_path, precision):
    obj = str(tmp_path / "main.o")
    p = subprocess.run(["gcc", *FLAGS, "-D" + precision, "-c", MAIN_C, "-o", obj], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # the only diagnostics are the reference's own unused variables / function (main.c:115, 181, 763), nothing about the ODE calls
    for line in p.stderr.splitlines():
        if "warning:" in line and "/src/main.c" in line:
            assert "unused variable" in line or "defined but not used" in line, line
    undefined = subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout.split()
    wanted = sorted(s for s in undefined if s.startswith("d") and len(s) > 1 and s[1].isupper())
    assert len(wanted) >= 30, wanted                       # the 32 ODE entry points main.c calls
    pkg = load_package()
    lib = pkg._lib.LIB_PATH if precision == "dDOUBLE" else pkg._lib.LIB_PATH.replace("libode_mi355.so", "libode_mi355_single.so")
    exported = set(subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout.split())
    missing = [s for s in wanted if s not in exported]
    assert not missing, missing


def test_dcontact_layout_as_main_c_indexes_it(tmp_path):
    """NearCallback (main.c:676-691) declares dContact contacts[MAX_CONTACTS], hands &contacts[0].geom to dCollide with a
    byte stride of sizeof(dContact), and reads .geom.depth / writes .surface.{mode,mu,bounce,bounce_vel}: the offsets the
    library's dCollide writes at must be the ones a C client compiled against the header computes."""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "ode/ode.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dContact), offsetof(dContact, geom), offsetof(dContactGeom, pos),\n'
        '  offsetof(dContactGeom, normal), offsetof(dContactGeom, depth), offsetof(dContactGeom, g1), sizeof(dReal)); return 0; }\n')
    for precision, rsize in (("dDOUBLE", 8), ("dSINGLE", 4)):
        exe = str(tmp_path / ("layout_" + precision))
        subprocess.run(["gcc", "-std=gnu99", "-D" + precision, "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
        sz, o_geom, o_pos, o_normal, o_depth, o_g1, real = map(int, subprocess.run([exe], capture_output=True, text=True).stdout.split())
        assert real == rsize
        assert o_pos == 0 and o_normal == 4 * rsize and o_depth == 8 * rsize          # dVector3 = dReal[4]
        assert o_g1 >= o_depth + rsize and o_geom > 0 and sz >= o_geom + o_g1 + 16
