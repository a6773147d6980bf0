// Native host preprocessing for the spiral mesh autoencoder: QSlim-style decimation, closest-point
// up-sampling coefficients, spiral orderings.  C ABI in include/sh_preprocess.h; no GPU involved.  Each routine follows
// the reference's Python step for step (file:line cited at each stage) so that it makes the same discrete decisions.
#include "../../include/sh_preprocess.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>
#include <tuple>
#include <vector>

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------------------------------------------------ QSlim
struct Quadric { double q[16]; };

// mesh_sampling.vertex_quadrics (:18-44): per face the normalised plane equation eq (|eq[0:3]| = 1), outer(eq, eq) added
// to the quadrics of its three vertices, faces in order.  The reference obtains eq as the null vector of [v 1] by SVD; the
// cross product spans the same null space (sign is irrelevant in the outer product) to rounding.
void vertex_quadrics(const double* v, int nv, const int32_t* f, int nf, std::vector<Quadric>& Q) {
    Q.assign(nv, Quadric{});
    for (int t = 0; t < nf; ++t) {
        const double* a = v + 3 * f[3 * t], *b = v + 3 * f[3 * t + 1], *c = v + 3 * f[3 * t + 2];
        const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        double eq[4];
        if (len > 0) {
            for (int k = 0; k < 3; ++k) eq[k] = n[k] / len;
            eq[3] = -(eq[0] * a[0] + eq[1] * a[1] + eq[2] * a[2]);
        } else {
            eq[0] = eq[1] = eq[2] = eq[3] = 0.0;          // degenerate face: contributes nothing
        }
        for (int k = 0; k < 3; ++k) {
            double* q = Q[f[3 * t + k]].q;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) q[4 * i + j] += eq[i] * eq[j];
        }
    }
}

// collapse_cost (:127-140): Qsum = Qv[r] + Qv[c]; destroy_c_cost = p1^T Qsum p1 (p1 = [v[r], 1]), destroy_r_cost with p2 = [v[c], 1]
struct Cost { double destroy_c, destroy_r, collapse; Quadric qsum; };
inline double quad_form(const Quadric& Q, const double* p3) {
    const double p[4] = {p3[0], p3[1], p3[2], 1.0};
    double t[4];
    for (int j = 0; j < 4; ++j) t[j] = ((p[0] * Q.q[j] + p[1] * Q.q[4 + j]) + p[2] * Q.q[8 + j]) + p[3] * Q.q[12 + j];   // p^T Q
    return ((t[0] * p[0] + t[1] * p[1]) + t[2] * p[2]) + t[3] * p[3];
}
inline Cost collapse_cost(const std::vector<Quadric>& Qv, int r, int c, const double* v) {
    Cost k;
    for (int i = 0; i < 16; ++i) k.qsum.q[i] = Qv[r].q[i] + Qv[c].q[i];
    k.destroy_c = quad_form(k.qsum, v + 3 * r);
    k.destroy_r = quad_form(k.qsum, v + 3 * c);
    k.collapse = k.destroy_r < k.destroy_c ? k.destroy_r : k.destroy_c;      // min([destroy_c_cost, destroy_r_cost])
    return k;
}

struct Entry { double cost; int r, c; };

// Python's heapq on tuples (cost, (r, c)): `a < b` compares the first element that differs
struct Heap {
    std::vector<int> h;                      // entry ids
    const std::vector<Entry>* e;
    bool lt(int a, int b) const {
        const Entry& x = (*e)[a], &y = (*e)[b];
        if (x.cost != y.cost) return x.cost < y.cost;
        if (x.r != y.r) return x.r < y.r;
        return x.c < y.c;
    }
    void siftdown(size_t start, size_t pos) {           // heapq._siftdown
        const int item = h[pos];
        while (pos > start) {
            const size_t parent = (pos - 1) >> 1;
            if (lt(item, h[parent])) { h[pos] = h[parent]; pos = parent; continue; }
            break;
        }
        h[pos] = item;
    }
    void siftup(size_t pos) {                            // heapq._siftup
        const size_t end = h.size(), start = pos;
        const int item = h[pos];
        size_t child = 2 * pos + 1;
        while (child < end) {
            const size_t right = child + 1;
            if (right < end && !lt(h[child], h[right])) child = right;
            h[pos] = h[child];
            pos = child;
            child = 2 * pos + 1;
        }
        h[pos] = item;
        siftdown(start, pos);
    }
    void push(int id) { h.push_back(id); siftdown(0, h.size() - 1); }
    int pop() {                                          // heapq.heappop
        const int last = h.back();
        h.pop_back();
        if (h.empty()) return last;
        const int top = h[0];
        h[0] = last;
        siftup(0);
        return top;
    }
};

// ------------------------------------------------------------------------------------------------------------ CPython sets
// utils_spiral.get_spirals keeps the candidate vertices / triangles of the outer rings in Python `set`s and takes
// "the first" of a set intersection (:276-345), so its output depends on CPython's hash-table layout.  This is a faithful
// model of CPython 3.8-3.12's setobject.c for the operations the traversal uses - insertion (linear probes + perturbation),
// removal (dummy entries), the resize policy, iteration in table order, and `a.intersection(b)` (iterates the smaller
// operand, inserts into a fresh set) - with CPython's hashes: hash(int) = the value, hash(tuple) = the xxHash-style mix of
// Objects/tupleobject.c.  Keys are ints (vertex ids) or face ids standing for the tuple (u, v, w).
struct PySetModel {
    struct Ent { int64_t hash; int key; char state; };       // state: 0 unused, 1 active, 2 dummy
    std::vector<Ent> table;
    size_t mask = 7, fill = 0, used = 0;
    PySetModel() { table.assign(8, Ent{0, 0, 0}); }
    static constexpr int LINEAR_PROBES = 9, PERTURB_SHIFT = 5;
    void insert_clean(std::vector<Ent>& t, size_t m, int key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & m;
        while (true) {
            size_t e = i;
            int probes = (i + LINEAR_PROBES <= m) ? LINEAR_PROBES : 0;
            do {
                if (t[e].state == 0) { t[e] = Ent{hash, key, 1}; return; }
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & m;
        }
    }
    void resize(size_t minused) {
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<Ent> nt(newsize, Ent{0, 0, 0});
        for (const Ent& e : table)
            if (e.state == 1) insert_clean(nt, newsize - 1, e.key, e.hash);
        table.swap(nt);
        mask = newsize - 1;
        fill = used;
    }
    bool contains(int key, int64_t hash) const {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        while (true) {
            size_t e = i;
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (table[e].state == 0) return false;
                if (table[e].state == 1 && table[e].hash == hash && table[e].key == key) return true;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void add(int key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        long freeslot = -1;
        while (true) {
            size_t e = i;
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (table[e].state == 0) {
                    if (freeslot >= 0) { table[freeslot] = Ent{hash, key, 1}; ++used; return; }
                    table[e] = Ent{hash, key, 1};
                    ++fill; ++used;
                    if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (table[e].state == 1 && table[e].hash == hash && table[e].key == key) return;     // already there
                if (table[e].state == 2) freeslot = (long)e;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void discard(int key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        while (true) {
            size_t e = i;
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            do {
                if (table[e].state == 0) return;
                if (table[e].state == 1 && table[e].hash == hash && table[e].key == key) { table[e].state = 2; table[e].hash = -1; --used; return; }
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    template <class Fn> void for_each(Fn&& fn) const { for (const Ent& e : table) if (e.state == 1) fn(e.key, e.hash); }
    std::vector<int> keys() const { std::vector<int> k; for_each([&](int key, int64_t) { k.push_back(key); }); return k; }
};
inline int64_t py_hash_int(int v) { return v == -1 ? -2 : (int64_t)v; }
inline int64_t py_hash_tuple3(int a, int b, int c) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    const int items[3] = {a, b, c};
    for (int k = 0; k < 3; ++k) {
        acc += (uint64_t)py_hash_int(items[k]) * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 3ULL ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796;
    return (int64_t)acc;
}
// a.intersection(b): iterate the smaller operand (b when the sizes are equal), keep what the other contains
inline PySetModel py_intersection(const PySetModel& a, const PySetModel& b) {
    const PySetModel* so = &a;
    const PySetModel* other = &b;
    if (other->used > so->used) std::swap(so, other);
    PySetModel r;
    other->for_each([&](int key, int64_t hash) { if (so->contains(key, hash)) r.add(key, hash); });
    return r;
}

// ------------------------------------------------------------------------------------------------------------ closest point
// closest point on triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5) with the feature it lies on:
// 0 interior, 1 / 2 / 3 edge ab / bc / ca, 4 / 5 / 6 vertex a / b / c (psbody-mesh's nearest_parts convention)
inline void sub(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
inline double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
int closest_on_triangle(const double* p, const double* a, const double* b, const double* c, double* out) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    sub(b, a, ab); sub(c, a, ac); sub(p, a, ap);
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { std::memcpy(out, a, 24); return 4; }
    sub(p, b, bp);
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { std::memcpy(out, b, 24); return 5; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        const double t = d1 / (d1 - d3);
        for (int k = 0; k < 3; ++k) out[k] = a[k] + t * ab[k];
        return 1;
    }
    sub(p, c, cp);
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { std::memcpy(out, c, 24); return 6; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        const double t = d2 / (d2 - d6);
        for (int k = 0; k < 3; ++k) out[k] = a[k] + t * ac[k];
        return 3;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int k = 0; k < 3; ++k) out[k] = b[k] + t * (c[k] - b[k]);
        return 2;
    }
    const double den = 1.0 / (va + vb + vc), vv = vb * den, ww = vc * den;
    for (int k = 0; k < 3; ++k) out[k] = a[k] + ab[k] * vv + ac[k] * ww;
    return 0;
}

// solve the 3x3 system [a b c] x = p (columns = triangle vertices; mesh_sampling.py:72-73 does it with lstsq)
bool solve3(const double* a, const double* b, const double* c, const double* p, double* x) {
    const double m[3][3] = {{a[0], b[0], c[0]}, {a[1], b[1], c[1]}, {a[2], b[2], c[2]}};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    double scale = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(m[i][j]));
    if (std::fabs(det) <= 1e-12 * scale * scale * scale) return false;
    for (int k = 0; k < 3; ++k) {
        double t[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) t[i][j] = j == k ? p[i] : m[i][j];
        x[k] = (t[0][0] * (t[1][1] * t[2][2] - t[1][2] * t[2][1]) - t[0][1] * (t[1][0] * t[2][2] - t[1][2] * t[2][0]) +
                t[0][2] * (t[1][0] * t[2][1] - t[1][1] * t[2][0])) / det;
    }
    return true;
}

}  // namespace

extern "C" {

const char* shp_last_error(void) { return g_err; }

int shp_qslim(const double* verts, int nv, const int32_t* faces_in, int nf, int n_target, int32_t* faces_out, int32_t* nf_out,
              int32_t* keep, int32_t* n_keep) {
    if (!verts || !faces_in || !faces_out || !nf_out || !keep || !n_keep || nv <= 0 || nf <= 0) return fail(-1, "shp_qslim: bad argument");
    for (int i = 0; i < 3 * nf; ++i)
        if (faces_in[i] < 0 || faces_in[i] >= nv) return fail(-1, "shp_qslim: face index %d out of range", faces_in[i]);
    std::vector<Quadric> Qv;
    vertex_quadrics(verts, nv, faces_in, nf, Qv);

    // vertex adjacency as (vert_adj + vert_adj.T).tocoo() lists it (:116-119): column by column, rows ascending
    std::vector<std::vector<int>> nbr(nv);
    for (int t = 0; t < nf; ++t)
        for (int k = 0; k < 3; ++k) {
            const int a = faces_in[3 * t + k], b = faces_in[3 * t + (k + 1) % 3];
            if (a != b) { nbr[a].push_back(b); nbr[b].push_back(a); }
        }
    for (auto& l : nbr) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }

    std::vector<Entry> ent;
    std::vector<char> alive;
    std::vector<std::vector<int>> first_of(nv), second_of(nv);       // queued entries by first / second end point
    Heap heap;
    heap.e = &ent;
    auto push = [&](double cost, int r, int c) {
        const int id = (int)ent.size();
        ent.push_back({cost, r, c});
        alive.push_back(1);
        first_of[r].push_back(id);
        second_of[c].push_back(id);
        heap.e = &ent;
        heap.push(id);
    };
    for (int c = 0; c < nv; ++c)                                     // :143-152
        for (int r : nbr[c]) {
            if (r > c) continue;
            push(collapse_cost(Qv, r, c, verts).collapse, r, c);
        }

    std::vector<int32_t> faces(faces_in, faces_in + 3 * nf);
    std::vector<char> f_alive(nf, 1);
    std::vector<std::vector<int>> vf(nv);
    std::vector<int> refcount(nv, 0);
    for (int t = 0; t < nf; ++t)
        for (int k = 0; k < 3; ++k) { vf[faces[3 * t + k]].push_back(t); ++refcount[faces[3 * t + k]]; }
    int nverts_total = nv;                                           // len(mesh.v) (:157), then len(unique(faces)) (:206)
    bool counted = false;

    while (nverts_total > n_target) {
        if (heap.h.empty()) return fail(-2, "shp_qslim: edge queue exhausted at %d vertices (target %d)", nverts_total, n_target);
        const int id = heap.pop();
        alive[id] = 0;
        const int r = ent[id].r, c = ent[id].c;
        const double e_cost = ent[id].cost;
        if (r == c) continue;                                        // :163-164 (does NOT recount the vertices)
        const Cost cost = collapse_cost(Qv, r, c, verts);
        if (cost.collapse > e_cost) {                                // outdated cost: re-queue (:167-170)
            push(cost.collapse, r, c);
            continue;
        }
        int to_destroy, to_keep;
        if (cost.destroy_c < cost.destroy_r) { to_destroy = c; to_keep = r; } else { to_destroy = r; to_keep = c; }
        // faces: to_destroy -> to_keep, drop degenerate faces (:183-204)
        for (int t : vf[to_destroy]) {
            if (!f_alive[t]) continue;
            for (int k = 0; k < 3; ++k)
                if (faces[3 * t + k] == to_destroy) { faces[3 * t + k] = to_keep; --refcount[to_destroy]; ++refcount[to_keep]; }
            const int a = faces[3 * t], b = faces[3 * t + 1], d = faces[3 * t + 2];
            if (a == b || b == d || d == a) {
                f_alive[t] = 0;
                --refcount[a]; --refcount[b]; --refcount[d];
            } else {
                vf[to_keep].push_back(t);
            }
        }
        vf[to_destroy].clear();
        // queued edges: rewrite in place, no re-heapify (:186-191); which1 / which2 are both taken from the queue as it was
        std::vector<int> w1, w2;
        for (int q : first_of[to_destroy]) if (alive[q] && ent[q].r == to_destroy) w1.push_back(q);
        for (int q : second_of[to_destroy]) if (alive[q] && ent[q].c == to_destroy) w2.push_back(q);
        for (int q : w1) { ent[q].r = to_keep; first_of[to_keep].push_back(q); }
        for (int q : w2) { ent[q].c = to_keep; second_of[to_keep].push_back(q); }
        first_of[to_destroy].clear();
        second_of[to_destroy].clear();
        Qv[r] = cost.qsum;
        Qv[c] = cost.qsum;
        // nverts_total = len(np.unique(faces.flatten()))
        if (!counted) {
            nverts_total = 0;
            for (int i = 0; i < nv; ++i) nverts_total += refcount[i] > 0;
            counted = true;
        } else {
            nverts_total = 0;                                       // recount lazily: cheap enough (nv ints), keeps the code obviously right
            for (int i = 0; i < nv; ++i) nverts_total += refcount[i] > 0;
        }
    }
    // _get_sparse_transform (:214-227)
    std::vector<int> mp(nv, -1);
    int nk = 0;
    for (int i = 0; i < nv; ++i)
        if (refcount[i] > 0) { mp[i] = nk; keep[nk++] = i; }
    int no = 0;
    for (int t = 0; t < nf; ++t)
        if (f_alive[t]) {
            for (int k = 0; k < 3; ++k) faces_out[3 * no + k] = mp[faces[3 * t + k]];
            ++no;
        }
    *n_keep = nk;
    *nf_out = no;
    return 0;
}

int shp_barycentric_upsample(const double* src_v, int n_src, const int32_t* src_f, int nf, const double* tgt_v, int n_tgt,
                             int32_t* cols, double* coeffs, int32_t* part) {
    if (!src_v || !src_f || !tgt_v || !cols || !coeffs || !part || n_src <= 0 || nf <= 0 || n_tgt <= 0)
        return fail(-1, "shp_barycentric_upsample: bad argument");
    // face bounding boxes for pruning; exhaustive otherwise (exact nearest feature, first minimum wins)
    std::vector<double> lo(3 * (size_t)nf), hi(3 * (size_t)nf);
    for (int t = 0; t < nf; ++t)
        for (int k = 0; k < 3; ++k) {
            const double a = src_v[3 * src_f[3 * t] + k], b = src_v[3 * src_f[3 * t + 1] + k], c = src_v[3 * src_f[3 * t + 2] + k];
            lo[3 * (size_t)t + k] = std::min(a, std::min(b, c));
            hi[3 * (size_t)t + k] = std::max(a, std::max(b, c));
        }
    for (int i = 0; i < n_tgt; ++i) {
        const double* p = tgt_v + 3 * (size_t)i;
        double best = std::numeric_limits<double>::infinity(), bp[3] = {0, 0, 0};
        int bf = 0, bpart = 0;
        for (int t = 0; t < nf; ++t) {
            double bb = 0;
            for (int k = 0; k < 3; ++k) {
                const double d = p[k] < lo[3 * (size_t)t + k] ? lo[3 * (size_t)t + k] - p[k] : (p[k] > hi[3 * (size_t)t + k] ? p[k] - hi[3 * (size_t)t + k] : 0.0);
                bb += d * d;
            }
            if (bb >= best) continue;
            double q[3];
            const int pt = closest_on_triangle(p, src_v + 3 * src_f[3 * t], src_v + 3 * src_f[3 * t + 1], src_v + 3 * src_f[3 * t + 2], q);
            const double d2 = (q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]);
            if (d2 < best) { best = d2; bf = t; bpart = pt; std::memcpy(bp, q, 24); }
        }
        const int32_t* f = src_f + 3 * bf;
        const double* a = src_v + 3 * f[0], *b = src_v + 3 * f[1], *c = src_v + 3 * f[2];
        double w[3] = {0, 0, 0};
        if (bpart == 0) {                                            // :69-73: A = [a b c], coefficients = lstsq(A, nearest point)
            if (!solve3(a, b, c, bp, w)) {                           // plane through the origin: barycentric coordinates instead
                double ab[3], ac[3], ap[3];
                sub(b, a, ab); sub(c, a, ac); sub(bp, a, ap);
                const double d00 = dot(ab, ab), d01 = dot(ab, ac), d11 = dot(ac, ac), d20 = dot(ap, ab), d21 = dot(ap, ac);
                const double den = d00 * d11 - d01 * d01;
                w[1] = (d11 * d20 - d01 * d21) / den; w[2] = (d00 * d21 - d01 * d20) / den; w[0] = 1.0 - w[1] - w[2];
            }
        } else if (bpart <= 3) {                                     // :74-79: least-squares fit of the TARGET point by the edge's end points
            const int i0 = bpart - 1, i1 = bpart % 3;
            const double* va = src_v + 3 * f[i0], *vb = src_v + 3 * f[i1];
            const double aa = dot(va, va), ab2 = dot(va, vb), bb2 = dot(vb, vb), ap2 = dot(va, p), bp2 = dot(vb, p);
            const double den = aa * bb2 - ab2 * ab2;
            if (std::fabs(den) > 0) { w[i0] = (bb2 * ap2 - ab2 * bp2) / den; w[i1] = (aa * bp2 - ab2 * ap2) / den; }
            else { w[i0] = 0.5; w[i1] = 0.5; }
        } else {
            w[bpart - 4] = 1.0;                                      // :80-82
        }
        for (int k = 0; k < 3; ++k) { cols[3 * (size_t)i + k] = f[k]; coeffs[3 * (size_t)i + k] = w[k]; }
        part[i] = bpart;
    }
    return 0;
}

int shp_spirals(const double* verts, int nv, const int32_t* faces, int nf, const int32_t* ref_points, int n_ref, int n_steps,
                int32_t* rowptr, int32_t* out, int64_t out_cap, int64_t* out_len) {
    if (!verts || !faces || !rowptr || !out_len || nv <= 0 || nf <= 0 || n_steps < 1 || (n_ref > 0 && !ref_points))
        return fail(-1, "shp_spirals: bad argument");
    for (int i = 0; i < 3 * nf; ++i)
        if (faces[i] < 0 || faces[i] >= nv) return fail(-1, "shp_spirals: face index %d out of range", faces[i]);
    // get_adj_trigs (:9-41): adj[v] = nonzero columns of the adjacency row (ascending), trig[v] = faces at v in face order
    std::vector<std::vector<int>> adj(nv), trig(nv);
    for (int t = 0; t < nf; ++t)
        for (int k = 0; k < 3; ++k) {
            const int a = faces[3 * t + k], b = faces[3 * t + (k + 1) % 3];
            adj[a].push_back(b); adj[b].push_back(a);
            trig[a].push_back(t);
        }
    for (auto& l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    auto F = [&](int t, int k) { return (int)faces[3 * t + k]; };
    auto in_face = [&](int t, int v) { return F(t, 0) == v || F(t, 1) == v || F(t, 2) == v; };
    auto in_trig = [&](int v, int t) { return std::find(trig[v].begin(), trig[v].end(), t) != trig[v].end(); };
    std::vector<int64_t> fhash(nf);                      // hash((u, v, w)) of every face tuple
    for (int t = 0; t < nf; ++t) fhash[t] = py_hash_tuple3(faces[3 * t], faces[3 * t + 1], faces[3 * t + 2]);

    // single_source_shortest_path (:101-125), once per reference point, later calls overwriting prev / dist of what they reach
    std::vector<int> heat_path(nv, -1);                  // -1 = None
    {
        typedef std::tuple<double, int, int> Item;       // (distance, vertex, predecessor) - the heap's total order
        for (int rp = 0; rp < n_ref; ++rp) {
            if (ref_points[rp] < 0 || ref_points[rp] >= nv) return fail(-1, "shp_spirals: reference point out of range");
            std::priority_queue<Item, std::vector<Item>, std::greater<Item>> q;
            std::vector<char> seen(nv, 0);
            int nseen = 0;
            q.push(Item(0.0, ref_points[rp], -1));
            while (!q.empty() && nseen < nv) {
                const Item it = q.top();
                q.pop();
                const int v = std::get<1>(it);
                if (seen[v]) continue;
                seen[v] = 1; ++nseen;
                heat_path[v] = std::get<2>(it);
                for (int w : adj[v]) {
                    if (seen[w]) continue;
                    const double dx = verts[3 * v] - verts[3 * w], dy = verts[3 * v + 1] - verts[3 * w + 1], dz = verts[3 * v + 2] - verts[3 * w + 2];
                    q.push(Item(std::get<0>(it) + std::sqrt((dx * dx + dy * dy) + dz * dz), w, v));
                }
            }
        }
    }
    auto is_source = [&](int i) { for (int k = 0; k < n_ref; ++k) if (ref_points[k] == i) return true; return false; };

    int64_t total = 0;
    std::vector<char> seen(nv, 0);
    std::vector<int> seen_list;
    auto mark = [&](int v) { if (!seen[v]) { seen[v] = 1; seen_list.push_back(v); } };
    auto is_seen = [&](int v) { return v >= 0 && seen[v]; };
    rowptr[0] = 0;
    for (int i = 0; i < nv; ++i) {
        for (int v : seen_list) seen[v] = 0;
        seen_list.clear();
        mark(i);
        std::vector<int> trig_central(trig[i]), spiral{i}, ring;
        int init_vert = -1;                               // -1 = None
        if (is_source(i)) {                               // closest neighbour (:147-153)
            double shortest = std::numeric_limits<double>::infinity();
            for (int nb : adj[i]) {
                const double dx = verts[3 * i] - verts[3 * nb], dy = verts[3 * i + 1] - verts[3 * nb + 1], dz = verts[3 * i + 2] - verts[3 * nb + 2];
                const double d = (dx * dx + dy * dy) + dz * dz;
                if (d < shortest) { shortest = d; init_vert = nb; }
            }
        } else {
            init_vert = heat_path[i];                     // on the shortest path to the reference point (:155-156)
        }
        bool orientation_0 = false, reverse_order = true;
        int v = -1;
        if (init_vert >= 0) { ring.push_back(init_vert); mark(init_vert); }
        auto remove_first = [](std::vector<int>& l, int t) { auto it = std::find(l.begin(), l.end(), t); if (it != l.end()) l.erase(it); };
        auto third_of = [&](int t, int a, int b) { for (int k = 0; k < 3; ++k) if (F(t, k) != a && F(t, k) != b) return F(t, k); return -1; };
        // ---- first ring (:166-208)
        while (!trig_central.empty() && init_vert >= 0) {
            const int cur_v = ring.back();
            std::vector<int> cur_t;
            for (int t : trig_central) if (in_trig(cur_v, t)) cur_t.push_back(t);
            if (ring.size() == 1) {
                if (cur_t.empty()) break;                 // (the reference would raise IndexError: start vertex shares no face)
                const int t0 = cur_t[0];
                orientation_0 = (F(t0, 0) == i && F(t0, 1) == cur_v) || (F(t0, 1) == i && F(t0, 2) == cur_v) || (F(t0, 2) == i && F(t0, 0) == cur_v);
                if (cur_t.size() >= 2) {
                    const int tt = orientation_0 ? cur_t[0] : cur_t[1];
                    const int third = third_of(tt, i, cur_v);
                    remove_first(trig_central, tt);
                    ring.push_back(third); mark(third);
                } else {
                    break;
                }
            } else {
                if (!cur_t.empty()) {
                    const int third = third_of(cur_t[0], cur_v, i);
                    if (!is_seen(third)) { ring.push_back(third); mark(third); }
                    remove_first(trig_central, cur_t[0]);
                } else {
                    break;
                }
            }
        }
        size_t rev_i = ring.size();
        if (init_vert >= 0) { v = init_vert; reverse_order = !(orientation_0 && ring.size() == 1); }
        bool need_padding = false;
        while (!trig_central.empty() && init_vert >= 0) {            // second half, reversed (:221-236)
            std::vector<int> cur_t;
            for (int t : trig_central) if (in_trig(v, t)) cur_t.push_back(t);
            if (cur_t.size() != 1) break;
            need_padding = true;
            const int third = third_of(cur_t[0], v, i);
            remove_first(trig_central, cur_t[0]);
            if (!is_seen(third)) {
                ring.insert(ring.begin() + rev_i, third); mark(third);
                if (!reverse_order) rev_i = ring.size();
                v = third;
            }
        }
        if (need_padding) ring.insert(ring.begin() + rev_i, -1);
        spiral.insert(spiral.end(), ring.begin(), ring.end());

        // ---- next rings (:253-413)
        for (int step = 0; step < n_steps - 1; ++step) {
            if (ring.empty()) break;
            PySetModel next_ring, next_trigs;                        // Python sets: iteration order = CPython's table order
            int base_triangle = -1;
            init_vert = -1;
            for (int w : ring)
                if (w != -1)
                    for (int u : adj[w])
                        if (!is_seen(u)) next_ring.add(u, py_hash_int(u));
            next_ring.for_each([&](int u, int64_t) {
                for (int tr : trig[u]) {
                    const int ns = (int)is_seen(F(tr, 0)) + (int)is_seen(F(tr, 1)) + (int)is_seen(F(tr, 2));
                    if (ns == 1) next_trigs.add(tr, fhash[tr]);
                    else if (ring.front() != -1 && ring.back() != -1 && in_face(tr, ring.front()) && in_face(tr, ring.back())) base_triangle = tr;
                }
            });
            auto trig_set = [&](int vtx) { PySetModel st; for (int t : trig[vtx]) st.add(t, fhash[t]); return st; };       // set(trig[v])
            auto touches_next = [&](int vtx) { return py_intersection(next_trigs, trig_set(vtx)).used > 0; };
            std::vector<int> iv;                                     // init_vert as a list; `have` distinguishes None from []
            bool have = false;
            if (base_triangle >= 0) {
                for (int k = 0; k < 3; ++k) if (F(base_triangle, k) != ring.front() && F(base_triangle, k) != ring.back()) iv.push_back(F(base_triangle, k));
                have = true;
                if (iv.empty() || !touches_next(iv[0])) { have = false; iv.clear(); }
            }
            if (!have) {
                for (size_t r = 0; r + 1 < ring.size(); ++r) {
                    if (ring[r] != -1 && ring[r + 1] != -1) {
                        for (int t : trig[ring[r]]) {
                            if (!in_trig(ring[r + 1], t)) continue;
                            iv.clear();
                            for (int k = 0; k < 3; ++k) if (!is_seen(F(t, k))) iv.push_back(F(t, k));
                            if (!iv.empty() && touches_next(iv[0])) break;
                            iv.clear();
                        }
                        if (!iv.empty() && touches_next(iv[0])) break;
                        iv.clear();
                    }
                }
            }
            if (!iv.empty()) { init_vert = iv[0]; ring.assign(1, init_vert); mark(init_vert); }
            else { init_vert = -1; ring.clear(); }

            auto remove_next = [&](int t) { next_trigs.discard(t, fhash[t]); };
            while (next_trigs.used > 0 && init_vert >= 0) {
                const int cur_v = ring.back();
                const std::vector<int> cur_t = py_intersection(next_trigs, trig_set(cur_v)).keys();     // list(next_trigs & set(trig[cur_v]))
                if (ring.size() == 1) {
                    if (cur_t.empty()) break;
                    const int t0 = cur_t[0];
                    orientation_0 = (is_seen(F(t0, 0)) && F(t0, 1) == cur_v) || (is_seen(F(t0, 1)) && F(t0, 2) == cur_v) || (is_seen(F(t0, 2)) && F(t0, 0) == cur_v);
                    if (cur_t.size() >= 2) {
                        const int tt = orientation_0 ? cur_t[0] : cur_t[1];
                        int third = -1;
                        for (int k = 0; k < 3; ++k) if (!is_seen(F(tt, k)) && F(tt, k) != cur_v) { third = F(tt, k); break; }
                        remove_next(tt);
                        if (third < 0) break;                        // (IndexError in the reference)
                        ring.push_back(third); mark(third);
                    } else {
                        break;
                    }
                } else {
                    if (!cur_t.empty()) {
                        int third = -1;
                        for (int k = 0; k < 3; ++k) if (!is_seen(F(cur_t[0], k))) { third = F(cur_t[0], k); break; }
                        remove_next(cur_t[0]);
                        if (third >= 0) { ring.push_back(third); mark(third); }
                        else break;
                    } else {
                        break;
                    }
                }
            }
            rev_i = ring.size();
            if (init_vert >= 0) { v = init_vert; reverse_order = !(orientation_0 && ring.size() == 1); }
            need_padding = false;
            while (next_trigs.used > 0 && init_vert >= 0) {
                std::vector<int> cur_t;
                next_trigs.for_each([&](int t, int64_t) { if (in_trig(v, t)) cur_t.push_back(t); });
                if (cur_t.size() != 1) break;
                need_padding = true;
                int third = -1;
                for (int k = 0; k < 3; ++k) if (F(cur_t[0], k) != v && !is_seen(F(cur_t[0], k))) { third = F(cur_t[0], k); break; }
                remove_next(cur_t[0]);
                if (third >= 0) {
                    ring.insert(ring.begin() + rev_i, third); mark(third);
                    if (!reverse_order) rev_i = ring.size();
                    v = third;
                }
            }
            if (need_padding) ring.insert(ring.begin() + rev_i, -1);
            spiral.insert(spiral.end(), ring.begin(), ring.end());
        }
        if (out && total + (int64_t)spiral.size() <= out_cap) std::copy(spiral.begin(), spiral.end(), out + total);
        total += (int64_t)spiral.size();
        if (total > std::numeric_limits<int32_t>::max()) return fail(-2, "shp_spirals: output too large");
        rowptr[i + 1] = (int32_t)total;
    }
    *out_len = total;
    if (!out || total > out_cap) return fail(-3, "shp_spirals: output needs %lld entries (capacity %lld)", (long long)total, (long long)out_cap);
    return 0;
}

}  // extern "C"
