"""Host-side (numpy) construction of the constant index tables the HIP kernels read.

The reference hands its model three kinds of per-level constants
(main.py:183-205): `spirals[l]` int64 [1, N_l+1, S_l] with -1 padding,
and dense fp32 `D[l]`, `U[l]` of shape [1, rows+1, cols+1].  The kernels never
touch those forms.  At module construction they are turned, once, into

  * gather tables   int32 [R, S]         (-1 -> the dummy row N, Appendix D-1)
  * CSR matrices    rowptr/col/val       (D: one 1.0 per row, U: <=3 nnz per row)
  * transposed ("who references me") tables for the atomic-free backward passes

All functions are pure numpy and run on the CPU; they are exercised by the
`not gpu` tests against the golden fixtures.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


# ----------------------------------------------------------------------------- spirals
def spirals_to_table(spiral_adj) -> np.ndarray:
    """[1|B, N+1, S] (any int/float dtype, -1 = padding) -> int32 [N+1, S] with
    -1 mapped to N, which is what torch's negative-index wrap does in
    reference models.py:42 (`x[batch_index, spirals_index, :]`)."""
    a = np.asarray(spiral_adj)
    if a.ndim == 3:
        a = a[0]
    n1 = a.shape[0]
    t = a.astype(np.int64)
    if t.min() < -n1 or t.max() >= n1:
        raise IndexError("spiral index out of range for %d rows" % n1)
    t = np.where(t < 0, t + n1, t)
    return np.ascontiguousarray(t.astype(np.int32))


@dataclass
class CSR:
    """Row-compressed sparse matrix, int32 indices, float32 values."""
    rows: int
    cols: int
    rowptr: np.ndarray   # int32 [rows+1]
    col: np.ndarray      # int32 [nnz]
    val: np.ndarray      # float32 [nnz]

    @property
    def nnz(self) -> int:
        return int(self.col.shape[0])

    def is_row_select(self) -> bool:
        """True when every row has exactly one entry whose value is exactly 1.0
        (the padded down-sampling matrices, mesh_sampling.py:214-227 +
        main.py:190)."""
        return (self.nnz == self.rows
                and np.array_equal(self.rowptr, np.arange(self.rows + 1, dtype=np.int32))
                and bool(np.all(self.val == np.float32(1.0))))

    def transpose(self) -> "CSR":
        """CSR of the transposed matrix; entries of each output row are in
        increasing source-row order, so the backward sums have a fixed order."""
        order = np.argsort(self.col, kind="stable")
        src_row = np.repeat(np.arange(self.rows, dtype=np.int32), np.diff(self.rowptr))
        counts = np.bincount(self.col, minlength=self.cols)
        rowptr = np.zeros(self.cols + 1, dtype=np.int32)
        np.cumsum(counts, out=rowptr[1:])
        return CSR(self.cols, self.rows, rowptr, src_row[order].astype(np.int32),
                   self.val[order].astype(np.float32))

    def todense(self) -> np.ndarray:
        m = np.zeros((self.rows, self.cols), dtype=np.float32)
        r = np.repeat(np.arange(self.rows), np.diff(self.rowptr))
        np.add.at(m, (r, self.col), self.val)
        return m


def split_identity_rows(u: CSR):
    """Split a re-sampling matrix into its identity rows and the rest.

    A row is an identity row when it stores exactly one entry and that entry is exactly 1.0 - the rows of an up-sampling
    U that belong to vertices the coarse mesh kept, and the padded dummy row (main.py:190-191): y[r] = x[col], a bit-exact
    copy.  With Z = [x ; U_b x] (the input rows followed by the non-identity rows of the product) every output row r of
    U x is row `row_map[r]` of Z, so a consumer that gathers rows - the next SpiralConv - can read Z through a composed
    table and U has to produce only the n_b blended rows.
    -> (row_map int32 [rows], u_b CSR [n_b, cols], m CSR [cols + n_b, cols] = [I ; U_b])   (u_b.rows == rows: nothing to fold)"""
    nnz = np.diff(u.rowptr)
    first = u.val[np.minimum(u.rowptr[:-1], max(u.nnz - 1, 0))] if u.nnz else np.zeros(u.rows, np.float32)
    ident = (nnz == 1) & (first == np.float32(1.0))
    bl = np.nonzero(~ident)[0]
    row_map = np.empty(u.rows, dtype=np.int32)
    row_map[ident] = u.col[u.rowptr[:-1][ident]]
    row_map[bl] = u.cols + np.arange(bl.size, dtype=np.int32)
    cnt = nnz[bl]
    rp = np.zeros(bl.size + 1, dtype=np.int32)
    np.cumsum(cnt, out=rp[1:])
    sel = np.concatenate([np.arange(u.rowptr[r], u.rowptr[r + 1]) for r in bl]) if bl.size else np.zeros(0, dtype=np.int64)
    u_b = CSR(int(bl.size), u.cols, rp, u.col[sel].astype(np.int32), u.val[sel].astype(np.float32))
    m = CSR(u.cols + int(bl.size), u.cols,
            np.concatenate([np.arange(u.cols, dtype=np.int32), u.cols + rp]).astype(np.int32),
            np.concatenate([np.arange(u.cols, dtype=np.int32), u_b.col]).astype(np.int32),
            np.concatenate([np.ones(u.cols, dtype=np.float32), u_b.val]).astype(np.float32))
    return row_map, u_b, m


def dense_to_csr(m) -> CSR:
    """Dense [1,R,C] or [R,C] matrix -> CSR keeping exact fp32 values.  Entries
    are kept in increasing column order (the order a dense dot product visits
    them), zeros are dropped."""
    a = np.asarray(m)
    if a.ndim == 3:
        a = a[0]
    a = a.astype(np.float32)
    r, c = np.nonzero(a)           # row-major order: sorted by row, then column
    rowptr = np.zeros(a.shape[0] + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=a.shape[0]), out=rowptr[1:])
    return CSR(a.shape[0], a.shape[1], rowptr, c.astype(np.int32), a[r, c].astype(np.float32))


def compose_select(table: np.ndarray, sel: np.ndarray) -> np.ndarray:
    """Fuse a row-select down-sampling into the convolution that precedes it.

    reference encode (models.py:122-127) computes y = conv(x, spirals_l) on all
    N_l+1 rows and then keeps rows `sel` (x_{l+1} = D_l y).  Because D_l is an
    exact row select, only the kept rows need computing:
        x_{l+1}[r] = conv_row(x, spirals_l[sel[r]])
    so the fused gather table is spirals_l[sel]  (int32 [N_{l+1}+1, S_l])."""
    return np.ascontiguousarray(table[sel.astype(np.int64)])


@dataclass
class GatherLists:
    """Transposed gather table, grouped by (input row u, spiral position s).

    For the backward-data pass  dX[u] = sum_s ( sum_{r : table[r,s]==u} dPre[r] ) W_s.
    `ptr` has U*S+1 entries; entries ptr[u*S+s] .. ptr[u*S+s+1] of `src` list
    the output rows r that read input row u at position s, in increasing r (a
    fixed order -> bitwise reproducible sums, no atomics)."""
    n_in: int
    S: int
    ptr: np.ndarray      # int32 [n_in*S + 1]
    src: np.ndarray      # int32 [R*S]
    max_len: int


def transpose_table(table: np.ndarray, n_in: int, skip_row: int = -1) -> GatherLists:
    """Build GatherLists for `table` int32 [R,S] whose values are in [0,n_in).
    `skip_row`: an input row whose gradient is known to be dead (the dummy row
    after a masked SpiralConv) gets empty lists, so its huge fan-in costs
    nothing."""
    R, S = table.shape
    key = table.astype(np.int64) * S + np.arange(S, dtype=np.int64)[None, :]
    key = key.ravel()
    src = np.repeat(np.arange(R, dtype=np.int32), S)
    if skip_row >= 0:
        keep = table.ravel() != skip_row
        key, src = key[keep], src[keep]
    order = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=n_in * S)
    ptr = np.zeros(n_in * S + 1, dtype=np.int32)
    np.cumsum(counts, out=ptr[1:])
    return GatherLists(n_in, S, ptr, np.ascontiguousarray(src[order]),
                       int(counts.max()) if counts.size else 0)


@dataclass
class TransposedTable:
    """Dense transposed gather table for the backward-data pass.

    table_t[u, s] is the ONE row of the (extended) dpre buffer whose W_s-product contributes to
    dx[u]:   dx[u] = sum_s dpre_ext[table_t[u,s]] . W_s.   Three cases per (u, s):
      * exactly one output row r reads input row u at position s      -> r
      * none does (or the gradient of row u is known dead)            -> none_row (a zero row of dpre)
      * several do (irregular vertices; the dummy row, ~900 readers)  -> an EXTRA row R + k that the
        caller fills beforehand with the sum of those rows: `csr2` (rows R+n1 .. R+n1+n2-1, unit
        values) sums either original rows directly (<= chunk entries) or the chunk sums produced by
        `csr1` (rows R .. R+n1-1) for very long lists.  Entries keep increasing-row order, so the
        sums have a fixed order and the whole backward pass is bitwise reproducible (no atomics).
    """
    table_t: np.ndarray          # int32 [n_in, S]
    csr1: Optional["CSR"]        # [n1, R]           chunk sums of very long lists
    csr2: Optional["CSR"]        # [n2, R + n1]      one row per multi-entry list
    n1: int
    n2: int

    @property
    def n_extra(self) -> int:
        return self.n1 + self.n2


def transpose_table_dense(table: np.ndarray, n_in: int, none_row: int, skip_row: int = -1,
                          chunk: int = 16) -> TransposedTable:
    R, S = table.shape
    gl = transpose_table(table, n_in, skip_row=skip_row)
    lengths = np.diff(gl.ptr)
    tt = np.full(n_in * S, none_row, dtype=np.int32)
    single = lengths == 1
    tt[single] = gl.src[gl.ptr[:-1][single]]
    multi = np.nonzero(lengths > 1)[0]
    if multi.size == 0:
        return TransposedTable(tt.reshape(n_in, S), None, None, 0, 0)
    rp1, col1, rp2, col2 = [0], [], [0], []
    for p in multi:
        ent = gl.src[gl.ptr[p]:gl.ptr[p + 1]]
        if ent.size <= chunk:
            col2.extend(int(e) for e in ent)
        else:
            for c in range(0, ent.size, chunk):
                col2.append(R + len(rp1) - 1)
                col1.extend(int(e) for e in ent[c:c + chunk])
                rp1.append(len(col1))
        rp2.append(len(col2))
    n1, n2 = len(rp1) - 1, len(rp2) - 1
    tt[multi] = R + n1 + np.arange(n2, dtype=np.int32)

    def ones_csr(rows, cols, rp, col):
        return CSR(rows, cols, np.asarray(rp, np.int32), np.asarray(col, np.int32), np.ones(len(col), np.float32))
    csr1 = ones_csr(n1, R, rp1, col1) if n1 else None
    csr2 = ones_csr(n2, R + n1, rp2, col2)
    return TransposedTable(tt.reshape(n_in, S), csr1, csr2, n1, n2)


def transpose_table_ragged(table: np.ndarray, n_in: int, none_row: int, skip_row: int = -1, max_len: int = 64):
    """The sources of every input row as a LIST (round 6; csrc/p3_conv.hip conv_p3r_kernel): for input row u all (output row r,
    position s) pairs with table[r, s] == u, ordered by s, then r.  Returns (rows int32 [n_in, L], pos int32 [n_in, L]) with L the
    longest list and pos == -1 (rows == none_row) behind a row's last source, or None when a list is longer than `max_len` (a
    dummy row with hundreds of readers whose gradient is needed: such a layer keeps the dense table and its pre-summed rows).
    The number of sources of a row is about the spiral length, or half of it on a down-sampling level, and never much more: the
    list form has no empty slots (8 233 of 37 906 dense slots at 3446 rows x 11 positions, more than half of them on the
    down-sampling levels) and needs no pre-summed rows (4 495 there)."""
    R, S = table.shape
    gl = transpose_table(table, n_in, skip_row=skip_row)
    per_slot = np.diff(gl.ptr).reshape(n_in, S)
    length = per_slot.sum(axis=1)
    L = int(length.max()) if length.size else 0
    if L == 0:
        L = 1
    if L > max_len:
        return None
    rows = np.full((n_in, L), none_row, dtype=np.int32)
    pos = np.full((n_in, L), -1, dtype=np.int32)
    # gl.src is ordered by (u, s) and, inside a slot, by increasing source row (stable sort): exactly the list order wanted
    start = gl.ptr[:-1].reshape(n_in, S)[:, 0]
    slot_pos = np.repeat(np.tile(np.arange(S, dtype=np.int32), n_in), per_slot.ravel())
    u_of = np.repeat(np.arange(n_in), length)
    j_of = np.arange(gl.src.size) - np.repeat(start, length)
    rows[u_of, j_of] = gl.src
    pos[u_of, j_of] = slot_pos
    return rows, pos


def _match_by_overlap(sets, weight_pairs, cap_len, can_merge):
    """Greedy maximum-weight matching: `weight_pairs` = (a, b, w) arrays over nodes with row sets `sets`; pairs are taken in order of
    decreasing shared rows (ties: smaller indices first - deterministic) when both ends are free, `can_merge(a, b)` holds and the
    union stays within `cap_len` rows.  -> list of merged node lists."""
    a, b, w = weight_pairs
    order = np.lexsort((b, a, -w))
    taken = np.zeros(len(sets), dtype=bool)
    out = []
    for k in order:
        i, j = int(a[k]), int(b[k])
        if taken[i] or taken[j] or not can_merge(i, j):
            continue
        if len(sets[i] | sets[j]) > cap_len:
            continue
        taken[i] = taken[j] = True
        out.append((i, j))
    out.extend((i,) for i in range(len(sets)) if not taken[i])
    return out


def _shared_row_pairs(owner_of_entry, row_of_entry, n_nodes, max_readers=64):
    """(a, b, shared rows) for every pair of nodes a < b that read a common row; rows with more than `max_readers` readers (a
    dummy row) say nothing about locality and are left out."""
    key = np.unique(row_of_entry.astype(np.int64) * n_nodes + owner_of_entry.astype(np.int64))       # one entry per (row, node)
    rows, nodes = key // n_nodes, key % n_nodes
    start = np.flatnonzero(np.r_[True, rows[1:] != rows[:-1]])
    cnt = np.diff(np.r_[start, rows.size])
    ok = np.repeat(cnt <= max_readers, cnt)
    rows, nodes = rows[ok], nodes[ok]
    pa, pb = [], []
    for d in range(1, int(min(cnt.max() if cnt.size else 0, max_readers))):
        same = rows[d:] == rows[:-d]
        pa.append(nodes[:-d][same])
        pb.append(nodes[d:][same])
    if not pa:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64)
    pa, pb = np.concatenate(pa), np.concatenate(pb)
    pk, w = np.unique(pa * n_nodes + pb, return_counts=True)
    return pk // n_nodes, pk % n_nodes, w


def group_lists(rows: np.ndarray, pos: np.ndarray, members: int = 4, max_len: int = 64):
    """Group output rows whose source lists overlap (round 6; csrc/p3_conv.hip conv_p3g_kernel).  `rows`, `pos` int32 [n, L]: per output
    row its sources (row of the gathered tensor, spiral position; pos == -1 behind the last one) - a forward gather table with
    pos = 0..S-1, or the ragged lists of `transpose_table_ragged`.  Neighbouring vertices' spirals share most of their rows; a group
    of up to `members` (2 or 4) output rows lists the UNION once.  Returns
        g_rows int32  [n_groups, Lg]     rows of the gathered tensor (padding: the group's first row)
        g_pos  uint32 [n_groups, Lg]     one byte per member: the position that member reads the row at, 0xFF = it does not
                                         (0xFFFFFFFF behind the group's last entry)
        g_out  int32  [n_groups, 4]      the members' output rows (-1: none)
    or None when a single list is longer than `max_len` or a position does not fit a byte.  Grouping: greedy matching on the number
    of shared rows (pairs first, then pairs of pairs), unions capped at `max_len` entries; deterministic.  A member that reads one
    row at two positions gets a second entry for it."""
    assert members in (2, 4) and rows.shape == pos.shape
    import hashlib
    key = (hashlib.sha1(np.ascontiguousarray(rows).tobytes() + np.ascontiguousarray(pos).tobytes()).hexdigest(), rows.shape, members, max_len)
    if key not in _GROUP_CACHE:                           # (the same mesh hierarchy is built many times per process: models, tests)
        if len(_GROUP_CACHE) > 256:
            _GROUP_CACHE.clear()
        _GROUP_CACHE[key] = _group_lists(rows, pos, members, max_len)
    return _GROUP_CACHE[key]


_GROUP_CACHE: dict = {}


def _group_lists(rows, pos, members, max_len):
    n, L = rows.shape
    valid = pos >= 0
    if n == 0 or int(valid.sum(axis=1).max()) > max_len or (pos.max() if pos.size else 0) >= 255:
        return None
    owner = np.repeat(np.arange(n), L)[valid.ravel()]
    rr = rows.ravel()[valid.ravel()]
    sets = [set() for _ in range(n)]
    for o, r in zip(owner.tolist(), rr.tolist()):
        sets[o].add(r)
    # a member reading one row twice needs that many entries of the group: count multiplicities in the cap through list lengths
    length = valid.sum(axis=1)
    slack = [int(length[i]) - len(sets[i]) for i in range(n)]
    pairs = _match_by_overlap(sets, _shared_row_pairs(owner, rr, n), max_len, lambda i, j: slack[i] == 0 and slack[j] == 0 or
                              len(sets[i] | sets[j]) + slack[i] + slack[j] <= max_len)
    groups = pairs
    if members == 4:
        psets = [set().union(*(sets[i] for i in g)) for g in pairs]
        pslack = [sum(slack[i] for i in g) for g in pairs]
        node_of = np.empty(n, dtype=np.int64)
        for k, g in enumerate(pairs):
            for i in g:
                node_of[i] = k
        quads = _match_by_overlap(psets, _shared_row_pairs(node_of[owner], rr, len(pairs)), max_len,
                                  lambda a, b: len(psets[a] | psets[b]) + pslack[a] + pslack[b] <= max_len)
        groups = [tuple(i for k in q for i in pairs[k]) for q in quads]
    groups.sort(key=lambda g: g[0])                       # launch order follows the rows' order (neighbouring groups share rows in L2)
    ent = []
    for g in groups:
        by_row = {}
        for m, i in enumerate(g):
            for r, q in zip(rows[i][valid[i]].tolist(), pos[i][valid[i]].tolist()):
                slots = by_row.setdefault(r, [])
                for e in slots:                            # the first entry of this row that member m has not used yet
                    if e[m] == 0xFF:
                        e[m] = q
                        break
                else:
                    e = [0xFF] * 4
                    e[m] = q
                    slots.append(e)
        # entries ordered by the smallest position that reads them, then row: every member meets its positions in ascending order
        # as far as the union allows
        flat = [(min(x for x in e if x != 0xFF), r, e) for r, slots in by_row.items() for e in slots]
        flat.sort(key=lambda t: (t[0], t[1]))
        ent.append(flat)
    Lg = max(len(f) for f in ent)
    if Lg > max_len:
        return None
    ng = len(groups)
    g_rows = np.zeros((ng, Lg), dtype=np.int32)
    g_pos = np.full((ng, Lg), 0xFFFFFFFF, dtype=np.uint32)
    g_out = np.full((ng, 4), -1, dtype=np.int32)
    for k, (g, flat) in enumerate(zip(groups, ent)):
        g_out[k, :len(g)] = g
        g_rows[k, :] = flat[0][1] if flat else 0
        for j, (_, r, e) in enumerate(flat):
            g_rows[k, j] = r
            g_pos[k, j] = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24)
    return g_rows, g_pos, g_out


# ----------------------------------------------------------------------------- U (up-sampling)
def _closest_point_barycentric(p, a, b, c):
    """Closest point to p on triangles (a,b,c) (all [M,3]); returns barycentric
    weights [M,3] and squared distance [M].  Region-based method (Ericson,
    Real-Time Collision Detection 5.1.5), vectorised."""
    ab, ac, ap = b - a, c - a, p - a
    d1 = np.einsum("ij,ij->i", ab, ap)
    d2 = np.einsum("ij,ij->i", ac, ap)
    bp = p - b
    d3 = np.einsum("ij,ij->i", ab, bp)
    d4 = np.einsum("ij,ij->i", ac, bp)
    cp = p - c
    d5 = np.einsum("ij,ij->i", ab, cp)
    d6 = np.einsum("ij,ij->i", ac, cp)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    w = np.zeros((p.shape[0], 3))
    done = np.zeros(p.shape[0], dtype=bool)

    def put(mask, wa, wb, wc):
        nonlocal done
        m = mask & ~done
        w[m, 0], w[m, 1], w[m, 2] = wa[m], wb[m], wc[m]
        done |= m

    one, zero = np.ones_like(d1), np.zeros_like(d1)
    with np.errstate(divide="ignore", invalid="ignore"):
        put((d1 <= 0) & (d2 <= 0), one, zero, zero)
        put((d3 >= 0) & (d4 <= d3), zero, one, zero)
        t = d1 / (d1 - d3)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), 1 - t, t, zero)
        put((d6 >= 0) & (d5 <= d6), zero, zero, one)
        t = d2 / (d2 - d6)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), 1 - t, zero, t)
        t = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), zero, 1 - t, t)
        denom = 1.0 / (va + vb + vc)
        v_, w_ = vb * denom, vc * denom
        put(np.ones_like(done), 1 - v_ - w_, v_, w_)
    q = w[:, :1] * a + w[:, 1:2] * b + w[:, 2:3] * c
    return w, np.sum((p - q) ** 2, axis=1)


def barycentric_upsample(src_v, src_f, tgt_v, k: int = 8) -> CSR:
    """Up-sampling operator U (tgt rows, src cols): each target vertex is
    expressed in barycentric coordinates of its closest point on the source
    (coarser) surface - the role of mesh_sampling.setup_deformation_transfer
    (mesh_sampling.py:47-95).  The reference delegates the closest-point search
    to psbody-mesh's C++ AABB tree (psbody-mesh 0.4, absent here), so
    this construction is our own: candidate triangles = those incident to the k
    nearest source vertices.  At most 3 non-zeros per row, rows sum to 1."""
    from scipy.spatial import cKDTree
    src_v = np.asarray(src_v, dtype=np.float64)
    tgt_v = np.asarray(tgt_v, dtype=np.float64)
    src_f = np.asarray(src_f, dtype=np.int64)
    nv = src_v.shape[0]
    # vertex -> incident faces (padded)
    order = np.argsort(src_f.ravel(), kind="stable")
    vert_of = src_f.ravel()[order]
    face_of = (order // 3)
    vptr = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum(np.bincount(vert_of, minlength=nv), out=vptr[1:])
    _, nn = cKDTree(src_v).query(tgt_v, k=min(k, nv))
    nn = nn.reshape(tgt_v.shape[0], -1)
    best_d = np.full(tgt_v.shape[0], np.inf)
    best_f = np.zeros(tgt_v.shape[0], dtype=np.int64)
    best_w = np.zeros((tgt_v.shape[0], 3))
    maxdeg = int(np.diff(vptr).max())
    for j in range(nn.shape[1]):
        v = nn[:, j]
        for t in range(maxdeg):
            has = (vptr[v] + t) < vptr[v + 1]
            if not has.any():
                continue
            rows = np.nonzero(has)[0]
            f = face_of[vptr[v[rows]] + t]
            tri = src_f[f]
            w, d = _closest_point_barycentric(tgt_v[rows], src_v[tri[:, 0]], src_v[tri[:, 1]], src_v[tri[:, 2]])
            better = d < best_d[rows]
            rr = rows[better]
            best_d[rr], best_f[rr], best_w[rr] = d[better], f[better], w[better]
    cols = src_f[best_f]                       # [T,3]
    vals = best_w.astype(np.float32)
    keep = vals != 0
    rowptr = np.zeros(tgt_v.shape[0] + 1, dtype=np.int32)
    np.cumsum(keep.sum(1), out=rowptr[1:])
    # entries in increasing column order inside a row (dense dot-product order)
    o = np.argsort(np.where(keep, cols, np.iinfo(np.int64).max), axis=1, kind="stable")
    cols_s = np.take_along_axis(cols, o, 1)
    vals_s = np.take_along_axis(vals, o, 1)
    keep_s = np.take_along_axis(keep, o, 1)
    return CSR(tgt_v.shape[0], nv, rowptr, cols_s[keep_s].astype(np.int32), vals_s[keep_s].astype(np.float32))


def pad_dummy(m: CSR) -> CSR:
    """Append the dummy row/column with a 1 at [-1,-1] (main.py:186-191)."""
    rowptr = np.concatenate([m.rowptr, [m.rowptr[-1] + 1]]).astype(np.int32)
    col = np.concatenate([m.col, [m.cols]]).astype(np.int32)
    val = np.concatenate([m.val, [1.0]]).astype(np.float32)
    return CSR(m.rows + 1, m.cols + 1, rowptr, col, val)
