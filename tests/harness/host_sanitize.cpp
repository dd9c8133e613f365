// tests/harness/host_sanitize.cpp -- driver for the sanitizer build of the library's host-only code (hull builder, OBJ
// reader, host thread pool): compiled with -fsanitize=address,undefined (and thread) by tests/test_hull.py.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <functional>
#include <atomic>
#include "dmx_hull.h"
namespace dmx { void host_pool_run(int nt, const std::function<void(int)> &f); }
int main(int argc, char **argv) {
    // hull of random point sets, including degenerate ones
    for (int n : {4, 5, 8, 30, 500, 5000}) {
        std::vector<double> p(3 * n), out(3 * n); std::vector<int32_t> idx(n);
        for (auto &x : p) x = (double)rand() / RAND_MAX - 0.5;
        dmxHullInfo info;
        int nv = dmxHullBuild(p.data(), n, 2.0, out.data(), idx.data(), n, &info);
        printf("n=%d hull=%d faces=%d vol=%.6f\n", n, nv, info.n_faces, info.volume);
        int nv2 = dmxHullBuild(p.data(), n, 2.0, out.data(), nullptr, 3, &info);   // small capacity: must not overrun
        if (nv2 != nv) return 1;
    }
    { std::vector<double> p = {0,0,0, 1,0,0, 0,1,0, 1,1,0, 0.5,0.5,0}; dmxHullInfo info; printf("coplanar -> %d\n", dmxHullBuild(p.data(), 5, 1.0, nullptr, nullptr, 0, &info)); }
    { std::vector<double> p(3 * 100, 1.0); dmxHullInfo info; printf("all equal -> %d\n", dmxHullBuild(p.data(), 100, 1.0, nullptr, nullptr, 0, &info)); }
    if (argc > 1) {                 // an OBJ file: parse it and build its hull
        int64_t nobj = dmxObjReadVertices(argv[1], nullptr, 0);
        if (nobj < 4) return 2;
        std::vector<double> v(3 * nobj), o(3 * nobj); dmxObjReadVertices(argv[1], v.data(), nobj);
        dmxHullInfo info; printf("obj hull %d\n", dmxHullBuild(v.data(), nobj, 0.01, o.data(), nullptr, (int)nobj, &info));
    }
    if (dmxObjReadVertices("/nonexistent.obj", nullptr, 0) != -1) return 3;
    // pool stress
    std::atomic<long> sum{0};
    for (int rep = 0; rep < 2000; rep++) {
        int nt = 1 + rep % 9;
        dmx::host_pool_run(nt, [&](int t) { sum += t + 1; });
    }
    printf("pool sum %ld\n", sum.load());
    return 0;
}
