// dmx_hull.cpp -- host-side convex hull builder for convex bodies (include/dmx_hull.h): OBJ vertices ->
// quickhull -> solid-hull mass properties -> hull vertices in the principal body frame.
//
// Nothing here runs on the device and nothing here is on the per-tick path: it is the set-up step that
// dCreateConvex + a dMass computation would be in an ODE program (the reference has no such call: SURVEY.md F9).
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/dmx_hull.h"

namespace {

struct P3 { double x, y, z; };
inline P3 sub(const P3 &a, const P3 &b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline P3 crs(const P3 &a, const P3 &b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline double dt(const P3 &a, const P3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double len(const P3 &a) { return sqrt(dt(a, a)); }

// ---- quickhull ------------------------------------------------------------------------------------------
struct Face {
    int v[3];              // counter-clockwise seen from outside
    int adj[3];            // face across edge (v[e], v[(e+1)%3])
    P3 n; double d;        // unit outward normal, n.p = d on the face
    std::vector<int> out;  // points strictly outside this face (the face's conflict list)
    bool alive = true, visible = false;
};

struct Hull {
    const std::vector<P3> &p;
    std::vector<Face> f;
    double eps;
    explicit Hull(const std::vector<P3> &pts) : p(pts), eps(0) {}

    double dist(const Face &fc, int i) const { return dt(fc.n, p[(size_t)i]) - fc.d; }

    int add_face(int a, int b, int c)
    {
        Face fc;
        fc.v[0] = a; fc.v[1] = b; fc.v[2] = c;
        fc.adj[0] = fc.adj[1] = fc.adj[2] = -1;
        P3 n = crs(sub(p[(size_t)b], p[(size_t)a]), sub(p[(size_t)c], p[(size_t)a]));
        const double l = len(n);
        fc.n = l > 0 ? P3{ n.x / l, n.y / l, n.z / l } : P3{ 0, 0, 0 };
        fc.d = dt(fc.n, p[(size_t)a]);
        f.push_back(fc);
        return (int)f.size() - 1;
    }

    int edge_of(const Face &fc, int a, int b) const
    {
        for (int e = 0; e < 3; e++)
            if (fc.v[e] == a && fc.v[(e + 1) % 3] == b) return e;
        return -1;
    }

    // horizon of the faces visible from `eye`, as (a, b, face beyond) in counter-clockwise order
    struct HEdge { int a, b, beyond; };
    void visit(int fi, int enter_edge, int eye, std::vector<HEdge> &horizon, std::vector<int> &vis)
    {
        f[(size_t)fi].visible = true;
        vis.push_back(fi);
        for (int k = 0; k < 3; k++) {
            const int e = enter_edge < 0 ? k : (enter_edge + 1 + k) % 3;
            if (enter_edge >= 0 && k == 2) break;               // the edge we came in through
            const int g = f[(size_t)fi].adj[e];
            if (f[(size_t)g].visible) continue;
            if (dist(f[(size_t)g], eye) > eps) {
                const int back = edge_of(f[(size_t)g], f[(size_t)fi].v[(e + 1) % 3], f[(size_t)fi].v[e]);
                visit(g, back, eye, horizon, vis);
            } else {
                horizon.push_back({ f[(size_t)fi].v[e], f[(size_t)fi].v[(e + 1) % 3], g });
            }
        }
    }

    bool build()
    {
        const int n = (int)p.size();
        if (n < 4) return false;
        // extreme points -> initial tetrahedron
        int ext[6] = { 0, 0, 0, 0, 0, 0 };
        for (int i = 1; i < n; i++) {
            if (p[(size_t)i].x < p[(size_t)ext[0]].x) ext[0] = i;
            if (p[(size_t)i].x > p[(size_t)ext[1]].x) ext[1] = i;
            if (p[(size_t)i].y < p[(size_t)ext[2]].y) ext[2] = i;
            if (p[(size_t)i].y > p[(size_t)ext[3]].y) ext[3] = i;
            if (p[(size_t)i].z < p[(size_t)ext[4]].z) ext[4] = i;
            if (p[(size_t)i].z > p[(size_t)ext[5]].z) ext[5] = i;
        }
        double span = 0;
        for (int i = 0; i < n; i++)
            span = std::max(span, std::max(fabs(p[(size_t)i].x), std::max(fabs(p[(size_t)i].y), fabs(p[(size_t)i].z))));
        eps = 64.0 * DBL_EPSILON * (span > 0 ? span : 1.0);
        int i0 = ext[0], i1 = ext[1];
        double best = -1;
        for (int a = 0; a < 6; a++)
            for (int b = a + 1; b < 6; b++) {
                const double l = len(sub(p[(size_t)ext[a]], p[(size_t)ext[b]]));
                if (l > best) { best = l; i0 = ext[a]; i1 = ext[b]; }
            }
        if (!(best > eps)) return false;
        int i2 = -1; best = eps;
        const P3 u = sub(p[(size_t)i1], p[(size_t)i0]);
        for (int i = 0; i < n; i++) {
            const double l = len(crs(u, sub(p[(size_t)i], p[(size_t)i0]))) / len(u);
            if (l > best) { best = l; i2 = i; }
        }
        if (i2 < 0) return false;
        int i3 = -1; best = eps;
        P3 nn = crs(u, sub(p[(size_t)i2], p[(size_t)i0]));
        const double nl = len(nn);
        nn = { nn.x / nl, nn.y / nl, nn.z / nl };
        for (int i = 0; i < n; i++) {
            const double l = fabs(dt(nn, sub(p[(size_t)i], p[(size_t)i0])));
            if (l > best) { best = l; i3 = i; }
        }
        if (i3 < 0) return false;
        if (dt(nn, sub(p[(size_t)i3], p[(size_t)i0])) > 0) std::swap(i1, i2);    // i3 must lie behind (i0,i1,i2)
        const int t[4][3] = { { i0, i1, i2 }, { i0, i3, i1 }, { i1, i3, i2 }, { i2, i3, i0 } };
        for (auto &tv : t) add_face(tv[0], tv[1], tv[2]);
        for (int a = 0; a < 4; a++)
            for (int e = 0; e < 3; e++)
                for (int b = 0; b < 4; b++)
                    if (b != a && edge_of(f[(size_t)b], f[(size_t)a].v[(e + 1) % 3], f[(size_t)a].v[e]) >= 0) f[(size_t)a].adj[e] = b;
        for (int i = 0; i < n; i++) {
            if (i == i0 || i == i1 || i == i2 || i == i3) continue;
            for (int a = 0; a < 4; a++)
                if (dist(f[(size_t)a], i) > eps) { f[(size_t)a].out.push_back(i); break; }
        }
        // main loop
        std::vector<int> work = { 0, 1, 2, 3 };
        std::vector<HEdge> horizon;
        std::vector<int> vis, orphans;
        while (!work.empty()) {
            const int fi = work.back();
            work.pop_back();
            if (!f[(size_t)fi].alive || f[(size_t)fi].out.empty()) continue;
            int eye = -1; double far = -1;
            for (int i : f[(size_t)fi].out) {
                const double dd = dist(f[(size_t)fi], i);
                if (dd > far) { far = dd; eye = i; }
            }
            horizon.clear(); vis.clear(); orphans.clear();
            visit(fi, -1, eye, horizon, vis);
            for (int v : vis) {
                for (int i : f[(size_t)v].out)
                    if (i != eye) orphans.push_back(i);
                f[(size_t)v].out.clear();
                f[(size_t)v].alive = false;
            }
            const int first_new = (int)f.size();
            const int nh = (int)horizon.size();
            for (int k = 0; k < nh; k++) {
                const int nf = add_face(horizon[(size_t)k].a, horizon[(size_t)k].b, eye);
                const int g = horizon[(size_t)k].beyond;
                f[(size_t)nf].adj[0] = g;
                f[(size_t)g].adj[edge_of(f[(size_t)g], horizon[(size_t)k].b, horizon[(size_t)k].a)] = nf;
            }
            for (int k = 0; k < nh; k++) {                      // new faces around the eye, in horizon order
                f[(size_t)(first_new + k)].adj[1] = first_new + (k + 1) % nh;       // edge (b, eye)
                f[(size_t)(first_new + k)].adj[2] = first_new + (k + nh - 1) % nh;  // edge (eye, a)
            }
            for (int i : orphans)
                for (int k = 0; k < nh; k++)
                    if (dist(f[(size_t)(first_new + k)], i) > eps) { f[(size_t)(first_new + k)].out.push_back(i); break; }
            for (int k = 0; k < nh; k++) work.push_back(first_new + k);
        }
        return true;
    }
};

// ---- mass properties of a closed triangle mesh with outward normals (polyhedral integrals, Eberly 2002) ----
struct MassProps { double volume, com[3], I[3][3]; };

inline void subexpr(double w0, double w1, double w2, double &f1, double &f2, double &f3, double &g0, double &g1, double &g2)
{
    const double t0 = w0 + w1, t1 = w0 * w0, t2 = t1 + w1 * t0;
    f1 = t0 + w2;
    f2 = t2 + w2 * f1;
    f3 = w0 * t1 + w1 * t2 + w2 * f2;
    g0 = f2 + w0 * (f1 + w0);
    g1 = f2 + w1 * (f1 + w1);
    g2 = f2 + w2 * (f1 + w2);
}

MassProps mass_props(const std::vector<P3> &p, const std::vector<Face> &faces)
{
    double in[10] = { 0 };
    for (const Face &fc : faces) {
        if (!fc.alive) continue;
        const P3 &a = p[(size_t)fc.v[0]], &b = p[(size_t)fc.v[1]], &c = p[(size_t)fc.v[2]];
        const P3 d = crs(sub(b, a), sub(c, a));
        double f1x, f2x, f3x, g0x, g1x, g2x, f1y, f2y, f3y, g0y, g1y, g2y, f1z, f2z, f3z, g0z, g1z, g2z;
        subexpr(a.x, b.x, c.x, f1x, f2x, f3x, g0x, g1x, g2x);
        subexpr(a.y, b.y, c.y, f1y, f2y, f3y, g0y, g1y, g2y);
        subexpr(a.z, b.z, c.z, f1z, f2z, f3z, g0z, g1z, g2z);
        in[0] += d.x * f1x;
        in[1] += d.x * f2x; in[2] += d.y * f2y; in[3] += d.z * f2z;
        in[4] += d.x * f3x; in[5] += d.y * f3y; in[6] += d.z * f3z;
        in[7] += d.x * (a.y * g0x + b.y * g1x + c.y * g2x);
        in[8] += d.y * (a.z * g0y + b.z * g1y + c.z * g2y);
        in[9] += d.z * (a.x * g0z + b.x * g1z + c.x * g2z);
    }
    const double mult[10] = { 1.0 / 6, 1.0 / 24, 1.0 / 24, 1.0 / 24, 1.0 / 60, 1.0 / 60, 1.0 / 60, 1.0 / 120, 1.0 / 120, 1.0 / 120 };
    for (int i = 0; i < 10; i++) in[i] *= mult[i];
    MassProps m;
    m.volume = in[0];
    m.com[0] = in[1] / in[0]; m.com[1] = in[2] / in[0]; m.com[2] = in[3] / in[0];
    const double cx = m.com[0], cy = m.com[1], cz = m.com[2];
    m.I[0][0] = in[5] + in[6] - in[0] * (cy * cy + cz * cz);
    m.I[1][1] = in[4] + in[6] - in[0] * (cz * cz + cx * cx);
    m.I[2][2] = in[4] + in[5] - in[0] * (cx * cx + cy * cy);
    m.I[0][1] = m.I[1][0] = -(in[7] - in[0] * cx * cy);
    m.I[1][2] = m.I[2][1] = -(in[8] - in[0] * cy * cz);
    m.I[0][2] = m.I[2][0] = -(in[9] - in[0] * cz * cx);
    return m;
}

// cyclic Jacobi: A = V diag(w) V^T, columns of V the eigenvectors
void jacobi3(double A[3][3], double V[3][3], double w[3])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-18 * diag) break;
        for (int pq = 0; pq < 3; pq++) {
            const int pi = pq == 2 ? 1 : 0, qi = pq == 0 ? 1 : 2;
            if (A[pi][qi] == 0.0) continue;
            const double theta = (A[qi][qi] - A[pi][pi]) / (2.0 * A[pi][qi]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) {                       // A <- A J
                const double akp = A[k][pi], akq = A[k][qi];
                A[k][pi] = c * akp - s * akq; A[k][qi] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; k++) {                       // A <- J^T A
                const double apk = A[pi][k], aqk = A[qi][k];
                A[pi][k] = c * apk - s * aqk; A[qi][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; k++) {
                const double vkp = V[k][pi], vkq = V[k][qi];
                V[k][pi] = c * vkp - s * vkq; V[k][qi] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < 3; i++) w[i] = A[i][i];
}

}  // namespace

extern "C" int64_t dmxObjReadVertices(const char *path, double *out_xyz, int64_t capacity)
{
    if (!path) return -1;
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    char line[512];
    int64_t n = 0;
    while (fgets(line, sizeof line, fp)) {
        if (line[0] != 'v' || (line[1] != ' ' && line[1] != '\t')) continue;
        double x, y, z;
        if (sscanf(line + 1, "%lf %lf %lf", &x, &y, &z) != 3) continue;
        if (out_xyz && n < capacity) { out_xyz[3 * n] = x; out_xyz[3 * n + 1] = y; out_xyz[3 * n + 2] = z; }
        n++;
    }
    fclose(fp);
    return n;
}

extern "C" int32_t dmxHullBuild(const double *xyz, int64_t n, double scale, double *out_points, int32_t *out_index,
                                int32_t capacity, dmxHullInfo *info)
{
    if (!xyz || n < 4 || !(scale > 0)) return -1;
    std::vector<P3> p((size_t)n);
    for (int64_t i = 0; i < n; i++) p[(size_t)i] = { xyz[3 * i] * scale, xyz[3 * i + 1] * scale, xyz[3 * i + 2] * scale };
    Hull h(p);
    if (!h.build()) return -2;

    std::vector<char> used((size_t)n, 0);
    int nf = 0;
    double area = 0;
    for (const Face &fc : h.f) {
        if (!fc.alive) continue;
        nf++;
        for (int e = 0; e < 3; e++) used[(size_t)fc.v[e]] = 1;
        area += 0.5 * len(crs(sub(p[(size_t)fc.v[1]], p[(size_t)fc.v[0]]), sub(p[(size_t)fc.v[2]], p[(size_t)fc.v[0]])));
    }
    MassProps m = mass_props(p, h.f);
    double A[3][3], V[3][3], w[3];
    memcpy(A, m.I, sizeof A);
    jacobi3(A, V, w);
    // a deterministic frame: axis k is the eigenvector most aligned with input axis k (greedy), pointing along +k,
    // made right-handed
    int perm[3] = { 0, 1, 2 };
    bool taken[3] = { false, false, false };
    for (int k = 0; k < 3; k++) {
        int bestc = -1; double bestv = -1;
        for (int c = 0; c < 3; c++)
            if (!taken[c] && fabs(V[k][c]) > bestv) { bestv = fabs(V[k][c]); bestc = c; }
        perm[k] = bestc; taken[bestc] = true;
    }
    double R[3][3];                                     // rows = principal axes
    for (int k = 0; k < 3; k++) {
        const double sgn = V[k][perm[k]] < 0 ? -1.0 : 1.0;
        for (int j = 0; j < 3; j++) R[k][j] = sgn * V[j][perm[k]];
    }
    const P3 r0 = { R[0][0], R[0][1], R[0][2] }, r1 = { R[1][0], R[1][1], R[1][2] };
    const P3 r2 = crs(r0, r1);
    if (dt(r2, P3{ R[2][0], R[2][1], R[2][2] }) < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }

    int32_t nv = 0;
    double radius = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!used[(size_t)i]) continue;
        const P3 q = { p[(size_t)i].x - m.com[0], p[(size_t)i].y - m.com[1], p[(size_t)i].z - m.com[2] };
        const double bx = R[0][0] * q.x + R[0][1] * q.y + R[0][2] * q.z;
        const double by = R[1][0] * q.x + R[1][1] * q.y + R[1][2] * q.z;
        const double bz = R[2][0] * q.x + R[2][1] * q.y + R[2][2] * q.z;
        radius = std::max(radius, sqrt(bx * bx + by * by + bz * bz));
        if (nv < capacity) {
            if (out_points) { out_points[3 * nv] = bx; out_points[3 * nv + 1] = by; out_points[3 * nv + 2] = bz; }
            if (out_index) out_index[nv] = (int32_t)i;
        }
        nv++;
    }
    if (info) {
        info->n_vertices = nv; info->n_faces = nf;
        info->volume = m.volume; info->area = area;
        for (int k = 0; k < 3; k++) {
            info->com[k] = m.com[k];
            info->inertia[k] = w[perm[k]];
            for (int j = 0; j < 3; j++) info->axes[3 * k + j] = R[k][j];
        }
        info->radius = radius;
    }
    return nv;
}

// The faces of the hull of `xyz` as planes: unit outward normal and offset (n.x <= d inside), one per triangle of the
// quickhull's surface, in its face order.  Called on a hull's body-frame points (dmxHullBuild's out_points) it gives the
// `planes` array dCreateConvex takes; dmxBatchSetConvexHullFaces puts it on the device for the box-convex collider.
extern "C" int32_t dmxHullPlanes(const double *xyz, int64_t n, double *out_planes, int32_t capacity)
{
    if (!xyz || n < 4) return -1;
    std::vector<P3> p((size_t)n);
    for (int64_t i = 0; i < n; i++) p[(size_t)i] = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    Hull h(p);
    if (!h.build()) return -2;
    int32_t nf = 0;
    for (const Face &fc : h.f) {
        if (!fc.alive) continue;
        if (out_planes && nf < capacity) {
            out_planes[4 * nf] = fc.n.x; out_planes[4 * nf + 1] = fc.n.y; out_planes[4 * nf + 2] = fc.n.z; out_planes[4 * nf + 3] = fc.d;
        }
        nf++;
    }
    return nf;
}
