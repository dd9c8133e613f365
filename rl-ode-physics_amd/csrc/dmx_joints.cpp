// dmx_joints.cpp -- dmxBatchStepJoints: one tick driven by an explicit list of contact joints (the form the
// reference's near callback produces, /root/reference/src/main.c:683-692, followed by dWorldStep, main.c:213).
//
// Host work is bookkeeping only: group bodies into dynamics islands (union-find over joints that connect two
// dynamic bodies; static geometry does not link islands), order islands / bodies / joints canonically
// (islands by lowest slot, bodies ascending, joints in creation order), copy the arrays to the device and
// launch solve_islands.  All arithmetic of the step runs on the GPU.
#include <string.h>
#include <memory>
#include <numeric>

#include "dmx_batch_priv.hpp"
#include "dmx_lcp.hpp"

namespace {


// multi-body islands with at least this many rows get a workgroup and a level schedule (DMX_BIG_ISLAND_ROWS overrides, for
// tests).  One lane walking an island pays a dependent L2 round trip per row and sweep, so every island with rows is better
// off with a wavefront: 500-body reference scene 5.2 / 3.3 / 2.4 / 1.7 / 1.3 / 1.1 / 0.8 ms per tick at 384 / 128 / 64 / 32 / 16 /
// 8 / 4 rows (profiles/r01_big_island_threshold.txt), 0.90 -> 0.67 from 4 to 1 with the one-body islands on solve_singles
// (profiles/r02_island_threshold.txt).
// DMX_LCP_OLD=1: dWorldStep's islands below the grid threshold go to round 2's one-workgroup kernel (lcp_island_wg) -- for A/B runs
bool lcp_old_kernel()
{
    static const bool v = [] { const char *e = getenv("DMX_LCP_OLD"); return e && atoi(e) != 0; }();
    return v;
}

int big_island_rows()
{
    static const int v = [] { const char *e = getenv("DMX_BIG_ISLAND_ROWS"); return e ? atoi(e) : 1; }();
    return v;
}

// union-find over body slots on a parent array that persists between ticks (identity outside a tick: only the entries
// a tick touched are put back, so a tick costs O(bodies involved), not O(bodies of the batch))
struct UnionFind {
    std::vector<int> &p;
    explicit UnionFind(std::vector<int> &parent) : p(parent) {}
    int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) { if (a < b) p[b] = a; else p[a] = b; } }
};

int ensure_pinned(void **p, size_t *have, size_t bytes)
{
    if (bytes <= *have) return DMX_OK;
    if (*p) HIP_TRY(hipHostFree(*p));
    *p = nullptr; *have = 0;
    size_t want = bytes + bytes / 2 + 256;
    HIP_TRY(hipHostMalloc(p, want));
    *have = want;
    return DMX_OK;
}

template <class T> int step_joints_t(dmxBatch *b, double h, int64_t nj_in, const dmxContactJoint *joints,
                                     const uint8_t *include, const DevGeometry *geo)
{
    const int n = (int)b->n;
    bool exact = b->stepper_exact;          // dWorldStep: islands solved exactly (decided per tick: see the row limit below)
    // per-slot scratch that persists between ticks: parent = identity, island = -1, last = -1 outside a tick
    if ((int)b->sc_parent.size() != n) {
        b->sc_parent.resize((size_t)n); std::iota(b->sc_parent.begin(), b->sc_parent.end(), 0);
        b->sc_island.assign((size_t)n, -1);
        b->sc_last.assign((size_t)n, -1);
    }
    // previous tick's async copies read the pinned staging buffers: drain before refilling
    HIP_TRY(hipStreamSynchronize(b->stream));

    // ---- canonical joints: body1 is a live dynamic slot, normal points into it -----------------------
    std::unique_ptr<DmxPhase> ph(new DmxPhase(b, 4));
    // (all per-tick work arrays below are members of the batch, reused from tick to tick: tens of MB of fresh
    // allocations per tick would be handed back to the OS and page-faulted in again every time)
    typedef DmxCanonicalJoint CJ;
    std::vector<CJ> &cj = b->sc_cj;
    cj.clear();
    cj.reserve((size_t)nj_in);
    // `include` (optional) restricts the tick to a subset of bodies: the rest is stepped by the fused kernels
    auto live = [&](int s) { return s >= 0 && s < n && (b->h_bflags[(size_t)s] & BF_ALIVE) && (!include || include[s]); };
    for (int64_t k = 0; k < nj_in; k++) {
        const dmxContactJoint &j = joints[k];
        int b1 = live(j.body1) ? j.body1 : -1, b2 = live(j.body2) ? j.body2 : -1;
        bool rev = false;
        if (b1 < 0 && b2 >= 0) { b1 = b2; b2 = -1; rev = true; }   // dJointAttach(c, 0, body): swap + reverse
        if (b1 < 0) continue;                                       // static-static: the stepper ignores it
        if (b1 == b2) continue;
        cj.push_back({ b1, b2, &j, rev });
    }
    const int nc = (int)cj.size();

    // ---- the bodies this tick steps, ascending: the caller's subset (`include`, with its list) or every live slot ----
    std::vector<int> &slots = b->sc_slots;
    slots.clear();
    if (include && b->sc_include_list) slots.assign(b->sc_include_list, b->sc_include_list + b->sc_include_count);
    else for (int s = 0; s < n; s++) if (live(s)) slots.push_back(s);
    const int nlive = (int)slots.size();

    // ---- islands ----------------------------------------------------------------------------------------
    UnionFind uf(b->sc_parent);
    for (const CJ &c : cj) if (c.b2 >= 0) uf.unite(c.b1, c.b2);
    std::vector<int> &island_of = b->sc_island;
    int ni = 0;
    for (int s : slots) {
        const int r = uf.find(s);             // roots are the lowest slot of their component
        if (r == s) island_of[(size_t)s] = ni++;
    }
    for (int s : slots)
        if (island_of[(size_t)s] < 0) island_of[(size_t)s] = island_of[(size_t)uf.find(s)];

    ph.reset(new DmxPhase(b, 5));
    // ---- which islands are large enough for a workgroup, and their level schedules (integers only) ------------
    // contacts are not yet in island order here; gather per-island contact lists in creation order first
    std::vector<int> &con_start = b->sc_iv[0];
    con_start.assign((size_t)ni + 1, 0);
    for (const CJ &c : cj) con_start[(size_t)island_of[(size_t)c.b1] + 1]++;
    for (int i = 0; i < ni; i++) con_start[(size_t)i + 1] += con_start[(size_t)i];
    std::vector<int> &con_sorted = b->sc_iv[1];          // cj indices grouped by island, creation order inside
    con_sorted.resize((size_t)nc);
    {
        std::vector<int> &f = b->sc_iv[15];
        f.assign(con_start.begin(), con_start.end() - 1);
        for (int k = 0; k < nc; k++) con_sorted[(size_t)f[(size_t)island_of[(size_t)cj[(size_t)k].b1]]++] = k;
    }
    // ---- ODE's own row order (dmxBatchSetRowOrder): within an island ODE numbers the rows in the order its island builder
    // discovers the joints -- a depth-first walk from the newest body, each body's joint list newest first [ODE-recall
    // dxProcessIslands, head-inserted lists] -- and QuickStep re-shuffles that order with the global LCG at every 8th
    // sweep (RANDOMLY_REORDER_CONSTRAINTS).  The LCG is consumed island by island in the walk's order, so the walk is
    // replayed here; the shuffled orders are made below, once the islands' row counts are known.
    const bool ode_order = b->row_order_ode && !exact;
    std::vector<int> &proc = b->sc_ode_proc;          // islands in the order ODE steps them
    proc.clear();
    if (ode_order) {
        std::vector<int> &adj_off = b->sc_ode_iv[0], &adj = b->sc_ode_iv[1], &fill = b->sc_ode_iv[2], &stack = b->sc_ode_iv[3];
        std::vector<uint8_t> &tag_b = b->sc_ode_tag_b, &tag_j = b->sc_ode_tag_j;
        adj_off.assign((size_t)n + 1, 0);
        for (const CJ &c : cj) { adj_off[(size_t)c.b1 + 1]++; if (c.b2 >= 0) adj_off[(size_t)c.b2 + 1]++; }
        for (int s = 0; s < n; s++) adj_off[(size_t)s + 1] += adj_off[(size_t)s];
        adj.resize((size_t)adj_off[(size_t)n] + 1);
        fill.assign(adj_off.begin(), adj_off.end() - 1);
        for (int k = 0; k < nc; k++) {
            adj[(size_t)fill[(size_t)cj[(size_t)k].b1]++] = k;
            if (cj[(size_t)k].b2 >= 0) adj[(size_t)fill[(size_t)cj[(size_t)k].b2]++] = k;
        }
        tag_b.assign((size_t)n, 0); tag_j.assign((size_t)nc + 1, 0);
        stack.clear();
        for (int q = nlive - 1; q >= 0; q--) {
            const int bb = slots[(size_t)q];
            if (tag_b[(size_t)bb]) continue;
            tag_b[(size_t)bb] = 1;
            const int isl = island_of[(size_t)bb];
            int at = con_start[(size_t)isl];
            int body = bb;
            for (;;) {
                const int lo = adj_off[(size_t)body], hi = adj_off[(size_t)body + 1];
                for (int t = 0; t < hi - lo; t++) {
                    const int j = adj[(size_t)(hi - 1 - t)];
                    if (tag_j[(size_t)j]) continue;
                    tag_j[(size_t)j] = 1;
                    con_sorted[(size_t)at++] = j;
                    const int other = cj[(size_t)j].b1 == body ? cj[(size_t)j].b2 : cj[(size_t)j].b1;
                    if (other >= 0 && !tag_b[(size_t)other]) { tag_b[(size_t)other] = 1; stack.push_back(other); }
                }
                if (stack.empty()) break;
                body = stack.back(); stack.pop_back();
            }
            proc.push_back(isl);
        }
    }
    std::vector<int> &crow_h = b->sc_iv[2];           // island-relative first row of each (sorted) contact
    crow_h.assign((size_t)nc, 0);
    std::vector<int> &big_h = b->sc_iv[3], &big_list_h = b->sc_iv[4], &lev_count_h = b->sc_iv[5], &lev_off_h = b->sc_iv[6],
                     &lev_rows_h = b->sc_iv[7], &lvl_all = b->sc_iv[8];
    big_h.assign((size_t)ni, -1); big_list_h.clear(); lev_count_h.clear(); lev_off_h.clear(); lev_rows_h.clear(); lvl_all.clear();
    int big_max_bodies = 0, big_max_width = 0, big_rows_total = 0, big_max_rows = 0;
    std::vector<int> &grid_list = b->sc_grid_list;      // dWorldStep: islands for the grid-wide exact solve
    grid_list.clear();
    size_t lds_need = 0;                                // ... and what the largest of the others needs of a workgroup's LDS
    std::vector<int> &island_bodies = b->sc_iv[9];
    island_bodies.assign((size_t)ni, 0);
    for (int s : slots) island_bodies[(size_t)island_of[(size_t)s]]++;
    {
        // Islands are independent, so every per-island pass below is spread over the host's cores (dmx_parallel_for).
        // (1) rows per island and each contact's first row
        std::vector<int> &m_of = b->sc_iv[10];
        std::vector<int> &nbd_of = b->sc_nbd_of;        // rows of the island that can clamp (normal rows, friction rows with finite mu)
        m_of.assign((size_t)ni, 0);
        nbd_of.assign((size_t)ni, 0);
        dmx_parallel_for(ni, 512, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; i++) {
                int m = 0, nbd = 0;
                for (int d = con_start[(size_t)i]; d < con_start[(size_t)i + 1]; d++) {
                    crow_h[(size_t)d] = m;
                    const double mu = cj[(size_t)con_sorted[(size_t)d]].j->mu;
                    m += mu > 0 ? 3 : 1;
                    nbd += (mu > 0 && mu < __builtin_huge_val()) ? 3 : 1;
                }
                m_of[(size_t)i] = m;
                nbd_of[(size_t)i] = nbd;
            }
        });
        // (2) the islands that get a workgroup, and where their rows sit in the flat arrays
        std::vector<int> &row_base = b->sc_iv[11];
        row_base.clear();
        int rows_total = 0;
        if (exact)
            for (int i = 0; i < ni; i++)
                if (m_of[(size_t)i] > lcp_max_exact_rows()) {
                    static bool warned = false;
                    if (!warned) fprintf(stderr, "libode_mi355: dWorldStep: an island of %d constraint rows exceeds the exact solver's limit (%d); "
                                                 "such ticks are stepped with QuickStep's SOR instead\n", m_of[(size_t)i], lcp_max_exact_rows());
                    warned = true;
                    exact = false;
                    lcp_grid_count_fallback(b);
                    break;
                }
        for (int i = 0; i < ni; i++) {
            if (ode_order) break;                     // shuffled sweeps are sequential: every island takes solve_islands
            if (m_of[(size_t)i] < (exact ? 1 : big_island_rows())) continue;
            // one body with 1..8 contacts: solve_singles' / solve_singles_lds' island (one lane), never a workgroup's
            if (!exact && island_bodies[(size_t)i] == 1 && con_start[(size_t)i + 1] - con_start[(size_t)i] <= 8) continue;
            // dWorldStep, an island of hundreds of rows or more: the grid-wide solve (dmx_lcp.hip), not one workgroup
            // (small and medium ones: one workgroup, the whole solve in LDS, while it fits)
            if (exact && !lcp_old_kernel() && !lcp_lds_fits((int)sizeof(T), m_of[(size_t)i], nbd_of[(size_t)i])) { grid_list.push_back(i); continue; }
            if (exact && m_of[(size_t)i] >= lcp_grid_threshold()) { grid_list.push_back(i); continue; }
            if (exact) lds_need = std::max(lds_need, lcp_lds_need((int)sizeof(T), m_of[(size_t)i], nbd_of[(size_t)i]));
            big_list_h.push_back(i);
            row_base.push_back(rows_total);
            rows_total += m_of[(size_t)i];
            big_max_bodies = std::max(big_max_bodies, island_bodies[(size_t)i]);
            big_max_rows = std::max(big_max_rows, m_of[(size_t)i]);
        }
        const int nbig = (int)big_list_h.size();
        if (exact) {
            // no level schedules: the exact solve is not a sweep.  Per-island scratch offsets instead (2 m^2 + 3 m reals).
            b->sc_lcp_off.resize((size_t)nbig + 1);
            long long at = 0;
            for (int k = 0; k < nbig; k++) {
                b->sc_lcp_off[(size_t)k] = at;
                const long long m = m_of[(size_t)big_list_h[(size_t)k]];
                at += 2 * m * m + 3 * m;
            }
            b->sc_lcp_off[(size_t)nbig] = at;
            lev_count_h.assign((size_t)nbig, 0);
            for (int k = 0; k < nbig; k++) big_h[(size_t)big_list_h[(size_t)k]] = 0;
            for (int i : grid_list) big_h[(size_t)i] = 0;         // (not the lane-per-island kernel's either)
            rows_total = 0;
        }
        const int nbig_sched = exact ? 0 : nbig;
        // (3) row r's level = 1 + the latest level of an earlier row sharing a body with it (creation order)
        std::vector<int> &lvl = lvl_all;                // row -> level, laid out like lev_rows (island k's rows from row_base[k])
        lvl.assign((size_t)rows_total, 0);
        big_rows_total = rows_total;
        lev_count_h.assign((size_t)nbig, 0);
        std::vector<int> &last = b->sc_last;          // per slot: level of the latest row touching the body; islands own disjoint slots
        dmx_parallel_for(nbig_sched, 64, [&](int64_t lo, int64_t hi, int) {
            for (int64_t k = lo; k < hi; k++) {
                const int i = big_list_h[(size_t)k];
                int *lv_out = lvl.data() + row_base[(size_t)k];
                int nlev = 0, r = 0;
                for (int d = con_start[(size_t)i]; d < con_start[(size_t)i + 1]; d++) {
                    const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                    const int rpc = c.j->mu > 0 ? 3 : 1;
                    for (int q = 0; q < rpc; q++, r++) {
                        int lv = last[(size_t)c.b1];
                        if (c.b2 >= 0 && last[(size_t)c.b2] > lv) lv = last[(size_t)c.b2];
                        lv += 1;
                        lv_out[r] = lv;
                        last[(size_t)c.b1] = lv;
                        if (c.b2 >= 0) last[(size_t)c.b2] = lv;
                        if (lv + 1 > nlev) nlev = lv + 1;
                    }
                }
                for (int d = con_start[(size_t)i]; d < con_start[(size_t)i + 1]; d++) {      // back to the idle state
                    const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                    last[(size_t)c.b1] = -1;
                    if (c.b2 >= 0) last[(size_t)c.b2] = -1;
                }
                lev_count_h[(size_t)k] = nlev;
            }
        });
        // (4) offsets of every island's level table, then (5) its rows grouped by level (counting sort)
        std::vector<int> &off_base = b->sc_iv[12];
        off_base.assign((size_t)nbig + 1, 0);
        for (int k = 0; k < nbig_sched; k++) off_base[(size_t)k + 1] = off_base[(size_t)k] + lev_count_h[(size_t)k] + 1;
        lev_off_h.assign((size_t)off_base[(size_t)nbig_sched], 0);
        lev_rows_h.assign((size_t)rows_total, 0);
        std::vector<int> &width_of = b->sc_iv[13];
        width_of.assign((size_t)nbig, 0);
        dmx_parallel_for(nbig_sched, 64, [&](int64_t lo, int64_t hi, int) {
            std::vector<int> fill;
            for (int64_t k = lo; k < hi; k++) {
                const int i = big_list_h[(size_t)k], m = m_of[(size_t)i], nlev = lev_count_h[(size_t)k], base = row_base[(size_t)k];
                int *off = lev_off_h.data() + off_base[(size_t)k];          // [nlev + 1], absolute positions in lev_rows
                const int *lv = lvl.data() + base;
                for (int q = 0; q < m; q++) off[lv[q] + 1]++;
                int w = 0;
                for (int q = 1; q <= nlev; q++) w = std::max(w, off[q]);
                width_of[(size_t)k] = w;
                off[0] = base;
                for (int q = 0; q < nlev; q++) off[q + 1] += off[q];
                fill.assign(off, off + nlev);
                for (int q = 0; q < m; q++) lev_rows_h[(size_t)fill[(size_t)lv[q]]++] = q;
            }
        });
        for (int k = 0; k < nbig_sched; k++) {
            big_h[(size_t)big_list_h[(size_t)k]] = off_base[(size_t)k];
            big_max_width = std::max(big_max_width, width_of[(size_t)k]);
        }
    }
    const int n_big = (int)big_list_h.size();
    ph.reset(new DmxPhase(b, 6));

    // int staging: body_off[ni+1] bodies[nlive] con_off[ni+1] row_off[ni+1] cb1[nc] cb2[nc] cmode[nc] csrc[nc] crow[nc]
    //              big[ni] big_list[n_big] lev_count[n_big] lev_off[..] lev_rows[..]
    const size_t n_int = (size_t)3 * (ni + 1) + (size_t)nlive + (size_t)5 * nc + (size_t)ni + (size_t)2 * n_big +
                         lev_off_h.size() + 2 * lev_rows_h.size();      // ... lev_rows[..] row_level[..]
    // real staging: cpos[3nc] cnormal[3nc] cdepth cmu cbounce cbounce_vel csoft_erp csoft_cfm [nc each]
    const size_t n_real = (size_t)12 * nc;
    int rc;
    if ((rc = ensure_pinned(&b->jh_int, &b->jh_int_bytes, n_int * sizeof(int) + 64)) != DMX_OK) return rc;
    if ((rc = ensure_pinned(&b->jh_real, &b->jh_real_bytes, n_real * sizeof(T) + 64)) != DMX_OK) return rc;
    int *hi = (int *)b->jh_int;
    T *hr = (T *)b->jh_real;
    int *body_off = hi, *bodies = body_off + (ni + 1), *con_off = bodies + nlive, *row_off = con_off + (ni + 1);
    int *cb1 = row_off + (ni + 1), *cb2 = cb1 + nc, *cmode = cb2 + nc, *csrc = cmode + nc, *crow = csrc + nc;
    int *big = crow + nc, *big_list = big + ni, *lev_count = big_list + n_big, *lev_off = lev_count + n_big;
    int *lev_rows = lev_off + lev_off_h.size();
    if (nc) memcpy(crow, crow_h.data(), (size_t)nc * sizeof(int));
    if (ni) memcpy(big, big_h.data(), (size_t)ni * sizeof(int));
    if (n_big) { memcpy(big_list, big_list_h.data(), (size_t)n_big * sizeof(int)); memcpy(lev_count, lev_count_h.data(), (size_t)n_big * sizeof(int)); }
    if (!lev_off_h.empty()) memcpy(lev_off, lev_off_h.data(), lev_off_h.size() * sizeof(int));
    if (!lev_rows_h.empty()) {
        memcpy(lev_rows, lev_rows_h.data(), lev_rows_h.size() * sizeof(int));
        memcpy(lev_rows + lev_rows_h.size(), lvl_all.data(), lev_rows_h.size() * sizeof(int));      // row_level, same layout
    }
    T *cpos = hr, *cnormal = cpos + 3 * (size_t)nc, *cdepth = cnormal + 3 * (size_t)nc, *cmu = cdepth + nc,
      *cbounce = cmu + nc, *cbv = cbounce + nc, *cserp = cbv + nc, *cscfm = cserp + nc;

    // counting sort of bodies and joints by island (stable: ascending slots / creation order)
    memset(body_off, 0, (size_t)(ni + 1) * sizeof(int));
    memset(con_off, 0, (size_t)(ni + 1) * sizeof(int));
    for (int s : slots) body_off[island_of[(size_t)s] + 1]++;
    for (const CJ &c : cj) con_off[island_of[(size_t)c.b1] + 1]++;
    for (int i = 0; i < ni; i++) { body_off[i + 1] += body_off[i]; con_off[i + 1] += con_off[i]; }
    for (int i = 0; i <= ni; i++) row_off[i] = 3 * con_off[i];
    {
        std::vector<int> &fill = b->sc_iv[14];
        fill.assign(body_off, body_off + ni);
        for (int s : slots) bodies[fill[(size_t)island_of[(size_t)s]]++] = s;
        // contact d of the island-ordered arrays is joint con_sorted[d] (stable counting sort above): fill in parallel
        dmx_parallel_for(nc, 8192, [&](int64_t lo, int64_t hi, int) {
            for (int64_t d = lo; d < hi; d++) {
                const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                const dmxContactJoint &j = *c.j;
                cb1[d] = c.b1; cb2[d] = c.b2; cmode[d] = j.mode;
                csrc[d] = geo ? geo->src[(size_t)(c.j - joints)] : 0;
                for (int k = 0; k < 3; k++) {
                    cpos[3 * (size_t)d + k] = (T)j.pos[k];
                    const T nk = (T)j.normal[k];
                    cnormal[3 * (size_t)d + k] = c.rev ? -nk : nk;
                }
                cdepth[d] = (T)j.depth; cmu[d] = (T)j.mu; cbounce[d] = (T)j.bounce; cbv[d] = (T)j.bounce_vel;
                cserp[d] = (T)j.soft_erp; cscfm[d] = (T)j.soft_cfm;
            }
        });
    }

    for (int s : slots) { b->sc_parent[(size_t)s] = s; island_of[(size_t)s] = -1; }     // scratch back to its idle state

    // ---- device buffers -----------------------------------------------------------------------------------
    ph.reset(new DmxPhase(b, 7));
    const size_t nrows = (size_t)3 * nc;
    if ((rc = dmx_ensure_dev(b->jd_int, n_int * sizeof(int) + 64)) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_real, n_real * sizeof(T) + 64)) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_rows, (nrows + 1) * ISLAND_ROW_REALS * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_rowjb, (nrows + 1) * 2 * sizeof(int))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_bscr, ((size_t)nlive + 1) * 28 * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_local, (size_t)b->stride * sizeof(int))) != DMX_OK) return rc;
    if (n_int) HIP_TRY(hipMemcpyAsync(b->jd_int.p, hi, n_int * sizeof(int), hipMemcpyHostToDevice, b->stream));
    if (n_real) HIP_TRY(hipMemcpyAsync(b->jd_real.p, hr, n_real * sizeof(T), hipMemcpyHostToDevice, b->stream));

    IslandSet<T> I;
    int *di = (int *)b->jd_int.p;
    T *dr = (T *)b->jd_real.p;
    I.n_islands = ni;
    I.body_off = di; I.bodies = di + (ni + 1); I.con_off = I.bodies + nlive; I.row_off = I.con_off + (ni + 1);
    I.cb1 = I.row_off + (ni + 1); I.cb2 = I.cb1 + nc; I.cmode = I.cb2 + nc;
    I.csrc = geo ? I.cmode + nc : nullptr;
    I.crow = I.cmode + 2 * (size_t)nc;
    I.big = I.crow + nc; I.n_big = n_big; I.big_list = I.big + ni; I.lev_count = I.big_list + n_big;
    I.lev_off = I.lev_count + n_big; I.lev_rows = I.lev_off + lev_off_h.size();
    I.big_max_bodies = big_max_bodies; I.big_max_width = big_max_width;
    I.big_rows_total = big_rows_total; I.big_max_rows = exact ? 0 : big_max_rows;
    I.row_level = I.lev_rows + lev_rows_h.size();
    I.gpos = geo ? (const T *)geo->pos : nullptr; I.gnormal = geo ? (const T *)geo->normal : nullptr;
    I.gdepth = geo ? (const T *)geo->depth : nullptr;
    I.cpos = dr; I.cnormal = dr + 3 * (size_t)nc; I.cdepth = I.cnormal + 3 * (size_t)nc; I.cmu = I.cdepth + nc;
    I.cbounce = I.cmu + nc; I.cbounce_vel = I.cbounce + nc; I.csoft_erp = I.cbounce_vel + nc; I.csoft_cfm = I.csoft_erp + nc;
    I.rows = (T *)b->jd_rows.p; I.rowjb = (int *)b->jd_rowjb.p; I.bscr = (T *)b->jd_bscr.p; I.local = (int *)b->jd_local.p;
    I.singles = (exact || ode_order) ? 0 : 1;
    I.order = nullptr; I.order_stride = 0;
    if (ode_order) {
        // the shuffled row orders, one per 8 sweeps, islands in ODE's processing order (the LCG is global and sequential)
        const int epochs = (b->iters + 7) / 8;
        const size_t R = nrows + 1;
        std::vector<int> &ord_h = b->sc_ode_order, &cur = b->sc_ode_iv[0];
        ord_h.assign((size_t)std::max(epochs, 1) * R, 0);
        const std::vector<int> &m_of = b->sc_iv[10];
        for (int isl : proc) {
            const int m = m_of[(size_t)isl];
            if (m <= 0) continue;
            const size_t base = (size_t)3 * con_start[(size_t)isl];
            cur.resize((size_t)m);
            for (int q = 0; q < m; q++) cur[(size_t)q] = q;
            for (int e = 0; e < epochs; e++) {
                for (int q = 1; q < m; q++) {
                    b->ode_rand = 1664525u * b->ode_rand + 1013904223u;                 // dRand [ODE-recall misc.cpp]
                    const int sw = (int)(((uint64_t)b->ode_rand * (uint32_t)(q + 1)) >> 32);   // dRandInt(q + 1)
                    std::swap(cur[(size_t)q], cur[(size_t)sw]);
                }
                memcpy(ord_h.data() + (size_t)e * R + base, cur.data(), (size_t)m * sizeof(int));
            }
        }
        if ((rc = dmx_ensure_dev(b->jd_order, ord_h.size() * sizeof(int))) != DMX_OK) return rc;
        HIP_TRY(hipMemcpyAsync(b->jd_order.p, ord_h.data(), ord_h.size() * sizeof(int), hipMemcpyHostToDevice, b->stream));
        I.order = (const int *)b->jd_order.p; I.order_stride = (int)R;
    }

    StepParams<T> P = dmx_make_params<T>(b, h);
    HIP_TRY(hipMemsetAsync(b->diag_isl, 0, sizeof(StepDiag), b->stream));
    if (exact) {
        const size_t nbig = (size_t)n_big;
        if ((rc = dmx_ensure_dev(b->jd_lcp, (size_t)(b->sc_lcp_off[nbig] + 1) * sizeof(T))) != DMX_OK) return rc;
        if ((rc = dmx_ensure_dev(b->jd_lcp_off, (nbig + 1) * sizeof(long long))) != DMX_OK) return rc;
        if ((rc = dmx_ensure_dev(b->jd_lcp_int, (nrows + 1) * 3 * sizeof(int))) != DMX_OK) return rc;
        HIP_TRY(hipMemcpyAsync(b->jd_lcp_off.p, b->sc_lcp_off.data(), (nbig + 1) * sizeof(long long), hipMemcpyHostToDevice, b->stream));
        int max_rows = 0;
        for (int k = 0; k < n_big; k++) max_rows = std::max(max_rows, b->sc_iv[10][(size_t)big_list_h[(size_t)k]]);
        HIP_TRY(launch_islands_exact<T>((T *)b->slab, b->bflags, b->stride, I, P, b->diag_isl, (T *)b->jd_lcp.p,
                                        (const long long *)b->jd_lcp_off.p, (int *)b->jd_lcp_int.p, lcp_old_kernel() ? max_rows : -1, b->stream));
        if (!lcp_old_kernel()) HIP_TRY(launch_lcp_lds<T>((T *)b->slab, b->bflags, b->stride, I, P, b->diag_isl, lds_need, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));      // sc_lcp_off is pageable host memory: the copy must have read it before the next tick rewrites it
        if (!grid_list.empty()) {
            // the large islands, one after the other: per row whether it can ever clamp, and the key its active-set state is
            // remembered under -- (body pair, ordinal of the contact within the pair this tick, row of the contact)
            lcp_grid_begin_tick(b);
            LcpIslandRows R;
            std::vector<std::pair<uint64_t, int>> &ord = b->sc_pair_ord;
            for (int isl : grid_list) {
                R.isl = isl; R.m = b->sc_iv[10][(size_t)isl]; R.row_base = 3 * con_start[(size_t)isl];
                R.unbounded.assign((size_t)R.m, 0); R.key.assign((size_t)R.m, 0);
                const int d0 = con_start[(size_t)isl], d1 = con_start[(size_t)isl + 1];
                ord.clear();
                for (int d = d0; d < d1; d++) {
                    const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                    ord.push_back({ ((uint64_t)(uint32_t)(c.b1 + 1) << 32) | (uint32_t)(c.b2 + 1), d });
                }
                std::stable_sort(ord.begin(), ord.end(), [](const std::pair<uint64_t, int> &x, const std::pair<uint64_t, int> &y) { return x.first < y.first; });
                for (size_t e = 0, run = 0; e < ord.size(); e++) {
                    if (e > 0 && ord[e].first != ord[e - 1].first) run = e;
                    const int d = ord[e].second;
                    const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                    const int r = crow_h[(size_t)d], rpc = c.j->mu > 0 ? 3 : 1;
                    const uint64_t base = ((uint64_t)(uint32_t)(c.b1 + 1) << 38) | ((uint64_t)((uint32_t)(c.b2 + 1) & 0xffffffu) << 14) |
                                          ((uint64_t)((e - run) & 0xfffu) << 2);
                    for (int q = 0; q < rpc; q++) {
                        R.key[(size_t)(r + q)] = base | (uint64_t)q;
                        R.unbounded[(size_t)(r + q)] = (q > 0 && !(c.j->mu < __builtin_huge_val())) ? 1 : 0;
                    }
                }
                // every body's rows (creation order): counting sort over the island's contacts; local index = position among the
                // island's bodies, which are the ascending slots body_off lists
                const int ib0 = body_off[isl], inb = body_off[isl + 1] - ib0;
                std::vector<int> &loc = b->sc_last;         // (-1 outside this block)
                for (int k = 0; k < inb; k++) loc[(size_t)bodies[ib0 + k]] = k;
                R.boff.assign((size_t)inb + 1, 0);
                for (int d = d0; d < d1; d++) {
                    const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                    const int rpc = c.j->mu > 0 ? 3 : 1;
                    R.boff[(size_t)loc[(size_t)c.b1] + 1] += rpc;
                    if (c.b2 >= 0) R.boff[(size_t)loc[(size_t)c.b2] + 1] += rpc;
                }
                for (int k = 0; k < inb; k++) R.boff[(size_t)k + 1] += R.boff[(size_t)k];
                R.bodyrows.assign((size_t)R.boff[(size_t)inb], 0);
                {
                    std::vector<int> &fill = b->sc_iv[14];
                    fill.assign(R.boff.begin(), R.boff.end() - 1);
                    for (int d = d0; d < d1; d++) {
                        const CJ &c = cj[(size_t)con_sorted[(size_t)d]];
                        const int r = crow_h[(size_t)d], rpc = c.j->mu > 0 ? 3 : 1;
                        for (int q = 0; q < rpc; q++) {
                            R.bodyrows[(size_t)fill[(size_t)loc[(size_t)c.b1]]++] = 2 * (r + q);
                            if (c.b2 >= 0) R.bodyrows[(size_t)fill[(size_t)loc[(size_t)c.b2]]++] = 2 * (r + q) + 1;
                        }
                    }
                }
                for (int k = 0; k < inb; k++) loc[(size_t)bodies[ib0 + k]] = -1;
                if ((rc = lcp_grid_solve<T>(b, I, P, R)) != DMX_OK) return rc;
            }
            lcp_grid_end_tick(b);
            HIP_TRY(hipStreamSynchronize(b->stream));
        }
    } else {
        HIP_TRY(launch_islands<T>((T *)b->slab, b->bflags, b->stride, I, P, b->diag_isl, b->stream));
        if (ode_order) HIP_TRY(hipStreamSynchronize(b->stream));      // the order table came from pageable host memory
    }
    b->last_islands = true;
    b->ext_pending = false;
    b->stepped_with_plane = true;     // diagnostics are valid
    return DMX_OK;
}

}  // namespace

int dmx_step_joints(dmxBatch *b, double h, int64_t n_joints, const dmxContactJoint *joints, const uint8_t *include,
                    const DevGeometry *geo)
{
    return b->precision == DMX_F32 ? step_joints_t<float>(b, h, n_joints, joints, include, geo)
                                   : step_joints_t<double>(b, h, n_joints, joints, include, geo);
}

extern "C" int dmxBatchStepJoints(dmxBatchID b, double h, int64_t n_joints, const dmxContactJoint *joints)
{
    if (!b || !(h > 0) || n_joints < 0 || (n_joints > 0 && !joints)) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    { const int rc = dmx_settle(b); if (rc != DMX_OK) return rc; }
    return dmx_step_joints(b, h, n_joints, joints, nullptr, nullptr);
}
