// tests/harness/collider_equiv.cpp -- the product's box-box collider (csrc/dmx_collide.hpp: written for one GPU lane per
// pair, everything in registers) against the CPU oracle's sequential restatement of dBoxBox (oracle/orc_boxbox.c), on the
// HOST, bit for bit, over random box pairs in the regimes the reference produces: a small box on a floor-sized one
// (main.c:115), boxes of similar size at random attitudes (piles), near-parallel edges, touching faces, and with fewer
// contacts asked than found (the culling branch).  Compiled by tests/test_collider_equivalence.py with the product's
// floating-point flags for REAL = double and float; links liboracle_f64.so / liboracle_f32.so.  Test infrastructure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include "dmx_collide.hpp"

#ifdef ORC_SINGLE
typedef float real;
#else
typedef double real;
#endif
struct orc_contactgeom { real pos[3]; real normal[3]; real depth; int g1, g2; };
extern "C" int orc_collide_box_box(const real *p1, const real *R1, const real *side1, const real *p2, const real *R2, const real *side2,
                                   int maxc, orc_contactgeom *out);

static uint64_t rng = 88172645463325252ull;
static double urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
static double uni(double a, double b) { return a + (b - a) * urand(); }

static dmx::M3<real> rot_of(double w, double x, double y, double z)
{
    const double l = std::sqrt(w * w + x * x + y * y + z * z);
    dmx::Q4<real> q = { (real)(w / l), (real)(x / l), (real)(y / l), (real)(z / l) };
    return dmx::quat_to_R(q);
}
static void to12(const dmx::M3<real> &R, real o[12]) { for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o[4 * i + j] = R.m[i][j]; o[4 * i + 3] = 0; } }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    long hits = 0, by_count[9] = { 0 }, culled = 0, mism = 0;
    for (long it = 0; it < n; it++) {
        const int regime = (int)(it % 5);
        dmx::V3<real> p1 = { 0, 0, 0 }, p2;
        dmx::M3<real> R1, R2;
        real s1[3], s2[3];
        if (regime == 0) {                 // a small box on / in a floor-sized one, tilted a little or a lot
            s1[0] = 100; s1[1] = 1; s1[2] = 100;
            R1 = rot_of(1, uni(-0.02, 0.02), uni(-0.02, 0.02), uni(-0.02, 0.02));
            for (int a = 0; a < 3; a++) s2[a] = (real)uni(0.2, 1.0);
            const double t = urand() < 0.5 ? 0.05 : 1.0;
            R2 = rot_of(1, uni(-t, t), uni(-t, t), uni(-t, t));
            p2 = { (real)uni(-49.9, 49.9), (real)uni(0.45, 1.1), (real)uni(-49.9, 49.9) };
            if (urand() < 0.1) p2.x = (real)(50.0 + uni(-0.6, 0.6));          // across the floor's rim
        } else if (regime == 1 || regime == 2) {      // similar sizes, random attitudes, centres close
            for (int a = 0; a < 3; a++) { s1[a] = (real)uni(0.2, 1.0); s2[a] = (real)uni(0.2, 1.0); }
            R1 = rot_of(uni(-1, 1), uni(-1, 1), uni(-1, 1), uni(-1, 1));
            R2 = rot_of(uni(-1, 1), uni(-1, 1), uni(-1, 1), uni(-1, 1));
            const double d = regime == 1 ? 0.9 : 0.5;
            p2 = { (real)uni(-d, d), (real)uni(-d, d), (real)uni(-d, d) };
        } else if (regime == 3) {          // axis-aligned and nearly so: touching faces, parallel edges, the degenerate branches
            for (int a = 0; a < 3; a++) { s1[a] = (real)(0.25 * (1 + (int)(urand() * 4))); s2[a] = (real)(0.25 * (1 + (int)(urand() * 4))); }
            const double e = urand() < 0.5 ? 0.0 : 1e-4;
            R1 = rot_of(1, uni(-e, e), uni(-e, e), uni(-e, e));
            R2 = urand() < 0.5 ? rot_of(1, uni(-e, e), uni(-e, e), uni(-e, e)) : rot_of(std::sqrt(0.5), 0, std::sqrt(0.5) + uni(-e, e), 0);
            p2 = { (real)(0.125 * (int)(uni(-8, 8))), (real)(0.125 * (int)(uni(-8, 8))), (real)(0.125 * (int)(uni(-8, 8))) };
        } else {                           // a plank narrower than the box lying across it: clipped hexagons and octagons
            s1[0] = 50; s1[1] = (real)0.8; s1[2] = (real)uni(0.3, 1.2);
            R1 = rot_of(1, 0, 0, uni(-0.05, 0.05));
            s2[0] = (real)uni(0.8, 2.0); s2[1] = (real)uni(0.2, 0.6); s2[2] = (real)uni(0.8, 2.0);
            const double yaw = uni(0, 3.2);
            R2 = rot_of(std::cos(yaw / 2), uni(-0.03, 0.03), std::sin(yaw / 2), uni(-0.03, 0.03));
            p2 = { (real)uni(-5, 5), (real)(0.4 + 0.5 * s2[1] - uni(0.0, 0.05)), (real)uni(-0.6, 0.6) };
        }
        const int maxc = (it % 7 == 3) ? 1 + (int)(urand() * 7) : 8;
        dmx::ContactPoint<real> c[8];
        memset(c, 0, sizeof(c));
        const int nc = dmx::box_box<real>(p1, R1, s1, p2, R2, s2, maxc, c);
        real P1[3] = { p1.x, p1.y, p1.z }, P2[3] = { p2.x, p2.y, p2.z }, r1[12], r2[12];
        to12(R1, r1); to12(R2, r2);
        orc_contactgeom o[8];
        memset(o, 0, sizeof(o));
        const int no = orc_collide_box_box(P1, r1, s1, P2, r2, s2, maxc, o);
        bool same = nc == no;
        for (int k = 0; same && k < nc; k++) {
            const real a[7] = { c[k].pos.x, c[k].pos.y, c[k].pos.z, c[k].normal.x, c[k].normal.y, c[k].normal.z, c[k].depth };
            const real b[7] = { o[k].pos[0], o[k].pos[1], o[k].pos[2], o[k].normal[0], o[k].normal[1], o[k].normal[2], o[k].depth };
            for (int q = 0; q < 7; q++) if (!(a[q] == b[q])) same = false;       // (== : +0 and -0 agree, NaN never does)
        }
        if (!same) {
            if (mism++ < 5) printf("MISMATCH it=%ld regime=%d maxc=%d product=%d oracle=%d\n", it, regime, maxc, nc, no);
        }
        if (nc > 0) hits++;
        by_count[nc < 8 ? nc : 8]++;
        if (maxc < 8 && nc == maxc) culled++;
    }
    printf("pairs %ld colliding %ld at-maxc-below-8 %ld counts", n, hits, culled);
    for (int k = 0; k <= 8; k++) printf(" %ld", by_count[k]);
    printf(" mismatches %ld\n", mism);
    return mism == 0 ? 0 : 1;
}
