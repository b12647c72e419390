"""ctypes binding of the CPU oracle (oracle/libhannoy_oracle.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by hannoy_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhannoy_oracle.so")

COSINE, EUCLIDEAN, MANHATTAN, HAMMING, BQ_COSINE, BQ_EUCLIDEAN, BQ_MANHATTAN = range(7)
ORDER_X86, ORDER_WAVE = 0, 1


def build_lib(force=False):
    src = os.path.join(HERE, "hannoy_oracle.cpp")
    hdr = os.path.join(HERE, "hannoy_oracle.h")
    if (force or not os.path.exists(LIB)
            or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", HERE, "-B"], stdout=subprocess.DEVNULL)
    return LIB


NONE = 0xFFFFFFFF  # by_item: Ok(None)


class QueryOpts(C.Structure):
    _fields_ = [("has_candidates", C.c_int32), ("candidates", C.c_void_p), ("n_candidates", C.c_uint64),
                ("query_items", C.c_void_p), ("linear_below", C.c_uint32),
                ("linear_below_ratio", C.c_float)]


class Opts(C.Structure):
    _fields_ = [("metric", C.c_int32), ("dim", C.c_uint32), ("M", C.c_uint32), ("M0", C.c_uint32),
                ("ef_construction", C.c_uint32), ("alpha", C.c_float), ("order", C.c_int32),
                ("threads", C.c_int32), ("batch_frac", C.c_double), ("batch_max", C.c_uint32),
                ("level_sort", C.c_int32), ("no_shuffle", C.c_int32), ("update_no_ramp", C.c_int32)]


class Items(C.Structure):
    _fields_ = [("n", C.c_uint64), ("ids", C.c_void_p), ("vectors", C.c_void_p),
                ("stride", C.c_size_t), ("headers", C.c_void_p), ("header_size", C.c_size_t),
                ("levels", C.c_void_p)]


class PrevGraph(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("rec_item", C.c_void_p), ("rec_layer", C.c_void_p),
                ("offsets", C.c_void_p), ("nbrs", C.c_void_p), ("entry_points", C.c_void_p),
                ("n_entry_points", C.c_uint32), ("max_level", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(LIB)
        L.orc_vector_bytes.restype = C.c_size_t
        L.orc_vector_bytes.argtypes = [C.c_int32, C.c_uint32]
        L.orc_header_bytes.restype = C.c_size_t
        L.orc_header_bytes.argtypes = [C.c_int32]
        L.orc_encode_vector.argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_make_header.argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.c_int32, C.c_int32, C.c_uint32] + [C.c_void_p] * 4
        for f in (L.orc_dot, L.orc_sqeuclid):
            f.restype = C.c_float
            f.argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        for f in (L.orc_dot_x86_emulated, L.orc_sqeuclid_x86_emulated):
            f.restype = C.c_float
            f.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_level_probas.restype = C.c_uint32
        L.orc_level_probas.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_draw_levels.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_build.restype = C.c_int
        L.orc_build.argtypes = [C.POINTER(Opts), C.POINTER(Items), C.POINTER(C.c_void_p)]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_build_incremental.restype = C.c_int
        L.orc_build_incremental.argtypes = [C.POINTER(Opts), C.POINTER(Items), C.c_void_p,
                                            C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.POINTER(PrevGraph), C.POINTER(C.c_void_p)]
        for name in ("orc_graph_n_records", "orc_graph_n_links", "orc_graph_n_raw",
                     "orc_graph_distance_evals", "orc_graph_links_added", "orc_graph_walk_evals"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_graph_export.argtypes = [C.c_void_p] * 5
        L.orc_graph_export_raw.argtypes = [C.c_void_p] * 4
        L.orc_graph_entry_points.restype = C.c_uint32
        L.orc_graph_entry_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_graph_max_level.restype = C.c_uint32
        L.orc_graph_max_level.argtypes = [C.c_void_p]
        L.orc_search.restype = C.c_int
        L.orc_search.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.POINTER(Items), C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t,
                                 C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
        L.orc_search_ex.restype = C.c_int
        L.orc_search_ex.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.POINTER(Items), C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32,
                                    C.POINTER(QueryOpts), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_encode_key.argtypes = [C.c_uint16, C.c_uint8, C.c_uint32, C.c_uint8, C.c_void_p]
        L.orc_roaring_serialize.restype = C.c_size_t
        L.orc_roaring_serialize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_encode_kv.restype = C.c_size_t
        L.orc_encode_kv.argtypes = [C.c_void_p, C.POINTER(Opts), C.POINTER(Items), C.c_uint16,
                                    C.c_int, C.c_void_p, C.c_size_t]
        L.orc_batch_size.restype = C.c_uint32
        L.orc_batch_size.argtypes = [C.c_double, C.c_uint32, C.c_uint64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def vector_bytes(metric, dim):
    return lib().orc_vector_bytes(metric, dim)


def header_bytes(metric):
    return lib().orc_header_bytes(metric)


def encode_vectors(metric, vecs):
    """f32 [n, dim] -> codec bytes uint8 [n, vector_bytes]."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    n, dim = vecs.shape
    vb = vector_bytes(metric, dim)
    out = np.zeros((n, vb), dtype=np.uint8)
    L = lib()
    for i in range(n):
        L.orc_encode_vector(metric, dim, _p(vecs[i]), _p(out[i]))
    return out


def make_headers(metric, dim, codes):
    n = codes.shape[0]
    hb = header_bytes(metric)
    out = np.zeros((n, hb), dtype=np.uint8)
    L = lib()
    for i in range(n):
        L.orc_make_header(metric, dim, _p(codes[i]), _p(out[i]))
    return out


def distance(metric, order, dim, pv, ph, qv, qh):
    return float(lib().orc_distance(metric, order, dim, _p(pv), _p(ph), _p(qv), _p(qh)))


def distance_pairs(ds, order, a, b, threads=8):
    """distance(item a[i], item b[i]) for many pairs of a Dataset's items (indices, not ids)"""
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.uint32)
    out = np.zeros(len(a), np.float32)
    L = lib()
    L.orc_distance_pairs.restype = None
    L.orc_distance_pairs.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.orc_distance_pairs(ds.metric, order, ds.dim, _p(ds.codes), ds.codes.shape[1], _p(ds.headers),
                         ds.headers.shape[1], len(a), _p(a), _p(b), _p(out), threads)
    return out


def dot(order, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().orc_dot(order, a.size, _p(a), _p(b)))


def sqeuclid(order, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().orc_sqeuclid(order, a.size, _p(a), _p(b)))


def dot_emulated(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().orc_dot_x86_emulated(a.size, _p(a), _p(b)))


def sqeuclid_emulated(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().orc_sqeuclid_x86_emulated(a.size, _p(a), _p(b)))


def host_threads():
    """CPU threads this process may actually run at once: os.cpu_count() clipped by the cgroup's CPU quota
    (a GPU box shows 256 logical CPUs and grants 16 CPUs' worth of time: 256 worker threads then share them)."""
    n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def level_probas(M):
    out = np.zeros(64, dtype=np.float32)
    n = lib().orc_level_probas(M, _p(out), 64)
    return out[:n].copy()


def draw_levels(M, n, seed32=None, seed_u64=0):
    """get_random_level x n from StdRng::from_seed(seed32) or ::seed_from_u64(seed_u64)."""
    out = np.zeros(n, np.uint8)
    sd = None if seed32 is None else np.ascontiguousarray(seed32, np.uint8)
    lib().orc_draw_levels(None if sd is None else _p(sd), seed_u64, M, n, _p(out))
    return out


def batch_size(frac, bmax, n_done):
    return lib().orc_batch_size(frac, bmax, n_done)


class Dataset:
    """Items as FrozenReader would hand them over (parallel.rs:33-45)."""

    def __init__(self, metric, dim, ids, codes, headers, levels):
        self.metric, self.dim = metric, dim
        self.ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert np.all(np.diff(self.ids.astype(np.int64)) > 0), "ids must be ascending"
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.headers = np.ascontiguousarray(headers, dtype=np.uint8)
        self.levels = np.ascontiguousarray(levels, dtype=np.uint8)
        self.n = len(self.ids)

    @classmethod
    def from_f32(cls, metric, vecs, levels, ids=None):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        n, dim = vecs.shape
        ids = np.arange(n, dtype=np.uint32) if ids is None else ids
        codes = encode_vectors(metric, vecs)
        return cls(metric, dim, ids, codes, make_headers(metric, dim, codes), levels)

    def items_struct(self):
        return Items(self.n, _p(self.ids).value, _p(self.codes).value, self.codes.shape[1],
                     _p(self.headers).value, self.headers.shape[1], _p(self.levels).value)


class Graph:
    def __init__(self, handle):
        L = lib()
        self._h = handle
        nrec = L.orc_graph_n_records(handle)
        nl = L.orc_graph_n_links(handle)
        self.rec_item = np.zeros(nrec, np.uint32)
        self.rec_layer = np.zeros(nrec, np.uint8)
        self.offsets = np.zeros(nrec + 1, np.uint64)
        self.nbrs = np.zeros(max(nl, 1), np.uint32)
        L.orc_graph_export(handle, _p(self.rec_item), _p(self.rec_layer), _p(self.offsets),
                           _p(self.nbrs))
        self.nbrs = self.nbrs[:nl]
        nraw = L.orc_graph_n_raw(handle)
        self.raw_offsets = np.zeros(nrec + 1, np.uint64)
        self.raw_nbrs = np.zeros(max(nraw, 1), np.uint32)
        self.raw_dists = np.zeros(max(nraw, 1), np.float32)
        L.orc_graph_export_raw(handle, _p(self.raw_offsets), _p(self.raw_nbrs), _p(self.raw_dists))
        self.raw_nbrs, self.raw_dists = self.raw_nbrs[:nraw], self.raw_dists[:nraw]
        eps = np.zeros(4096, np.uint32)
        ne = L.orc_graph_entry_points(handle, _p(eps), 4096)
        self.entry_points = eps[:ne].copy()
        self.max_level = L.orc_graph_max_level(handle)
        self.n_distance_evals = L.orc_graph_distance_evals(handle)
        self.n_links_added = L.orc_graph_links_added(handle)
        self.n_evals_walk = L.orc_graph_walk_evals(handle)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_graph_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def as_dict(self):
        """{(item, layer): [neighbour ids]}"""
        out = {}
        for r in range(len(self.rec_item)):
            a, b = int(self.offsets[r]), int(self.offsets[r + 1])
            out[(int(self.rec_item[r]), int(self.rec_layer[r]))] = self.nbrs[a:b].tolist()
        return out

    def raw_dict(self):
        out = {}
        for r in range(len(self.rec_item)):
            a, b = int(self.raw_offsets[r]), int(self.raw_offsets[r + 1])
            out[(int(self.rec_item[r]), int(self.rec_layer[r]))] = (
                self.raw_nbrs[a:b].tolist(), self.raw_dists[a:b].copy())
        return out


LEVEL_SORT_BY_ID, LEVEL_SORT_RUST = 0, 1  # orc_opts.level_sort (hnsw.rs:268)


def make_opts(metric, dim, M=16, M0=32, ef=100, alpha=1.0, order=ORDER_X86, threads=1,
              batch_frac=0.0, batch_max=0, level_sort=LEVEL_SORT_RUST, no_shuffle=False, update_no_ramp=False,
              schedule=None):
    """schedule: the product's hny_build_opts.schedule bits (HNY_SCHED_*), so that a test hands both sides the
    same word: 1 = no_shuffle, 2 = level order by id, 4 = update_no_ramp"""
    if schedule is not None:
        no_shuffle, update_no_ramp = bool(schedule & 1), bool(schedule & 4)
        level_sort = LEVEL_SORT_BY_ID if schedule & 2 else LEVEL_SORT_RUST
    return Opts(metric, dim, M, M0, ef, alpha, order, threads, batch_frac, batch_max, level_sort,
                int(bool(no_shuffle)), int(bool(update_no_ramp)))


def gen_f32(seed32, skip, n):
    """rng.gen::<f32>() x n from StdRng::from_seed(seed32) after `skip` u32 words (orc_gen_f32)"""
    out = np.zeros(n, np.float32)
    seed = (C.c_uint8 * 32).from_buffer_copy(bytes(seed32))
    L = lib()
    L.orc_gen_f32.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.orc_gen_f32(seed, skip, n, _p(out))
    return out


def draw_levels_skip(seed32, skip, M, n):
    """get_random_level x n from StdRng::from_seed(seed32) after `skip` u32 words"""
    out = np.zeros(n, np.uint8)
    seed = (C.c_uint8 * 32).from_buffer_copy(bytes(seed32))
    L = lib()
    L.orc_draw_levels_skip.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
    L.orc_draw_levels_skip(seed, skip, M, n, _p(out))
    return out


def rust_sort_levels(ids, levels):
    """hnsw.rs:268 as Rust >= 1.81 sorts (id, level) pairs by level descending (orc_rust_sort_levels)"""
    i = np.ascontiguousarray(ids, np.uint32).copy()
    lv = np.ascontiguousarray(levels, np.uint32).copy()
    L = lib()
    L.orc_rust_sort_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.orc_rust_sort_levels(_p(i), _p(lv), len(i))
    return i, lv


def build(ds, **kw):
    o = make_opts(ds.metric, ds.dim, **kw)
    it = ds.items_struct()
    h = C.c_void_p()
    rc = lib().orc_build(C.byref(o), C.byref(it), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"orc_build failed: {rc}")
    g = Graph(h)
    g.opts = o
    return g


def build_incremental(ds, prev, to_insert, insert_levels, to_delete, **kw):
    """ds: items that exist after the update; prev: graph of the previous build."""
    o = make_opts(ds.metric, ds.dim, **kw)
    it = ds.items_struct()
    ins = np.ascontiguousarray(to_insert, np.uint32)
    lv = np.ascontiguousarray(insert_levels, np.uint8)
    dl = np.ascontiguousarray(to_delete, np.uint32)
    keep = [np.ascontiguousarray(prev.rec_item, np.uint32), np.ascontiguousarray(prev.rec_layer, np.uint8),
            np.ascontiguousarray(prev.offsets, np.uint64),
            np.ascontiguousarray(prev.nbrs if len(prev.nbrs) else np.zeros(1), np.uint32),
            np.ascontiguousarray(prev.entry_points, np.uint32)]
    pg = PrevGraph(len(keep[0]), _p(keep[0]).value, _p(keep[1]).value, _p(keep[2]).value,
                   _p(keep[3]).value, _p(keep[4]).value, len(keep[4]), int(prev.max_level))
    h = C.c_void_p()
    rc = lib().orc_build_incremental(C.byref(o), C.byref(it), _p(ins), len(ins), _p(lv), _p(dl),
                                     len(dl), C.byref(pg), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"orc_build_incremental failed: {rc}")
    g = Graph(h)
    g.opts = o
    return g


def search(ds, graph, qcodes, qheaders, k=10, ef_search=100, order=ORDER_X86, threads=1,
           candidates=None, query_items=None, linear_below=1000, linear_below_ratio=1.0):
    """Reader::nns(k).ef_search(..).candidates(..).by_vector / by_item.
    graph: anything with rec_item/rec_layer/offsets/nbrs/entry_points/max_level arrays.
    query_items: item ids (by_item) instead of qcodes/qheaders; counts == NONE where the reference
    returns None."""
    it = ds.items_struct()
    if query_items is not None:
        query_items = np.ascontiguousarray(query_items, np.uint32)
        nq = len(query_items)
        qcodes = np.zeros((1, max(1, ds.codes.shape[1])), np.uint8)
        qheaders = np.zeros((1, ds.headers.shape[1]), np.uint8)
    else:
        qcodes = np.ascontiguousarray(qcodes, np.uint8)
        qheaders = np.ascontiguousarray(qheaders, np.uint8)
        nq = qcodes.shape[0]
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    counts = np.zeros(nq, np.uint32)
    rec_item = np.ascontiguousarray(graph.rec_item, np.uint32)
    rec_layer = np.ascontiguousarray(graph.rec_layer, np.uint8)
    offsets = np.ascontiguousarray(graph.offsets, np.uint64)
    nbrs = np.ascontiguousarray(graph.nbrs, np.uint32)
    eps = np.ascontiguousarray(graph.entry_points, np.uint32)
    qo = QueryOpts()
    cand = None
    if candidates is not None:
        cand = np.ascontiguousarray(candidates, np.uint32)
        qo.has_candidates, qo.candidates, qo.n_candidates = 1, cand.ctypes.data, len(cand)
    qo.query_items = query_items.ctypes.data if query_items is not None else None
    qo.linear_below, qo.linear_below_ratio = linear_below, linear_below_ratio
    rc = lib().orc_search_ex(ds.metric, order, ds.dim, C.byref(it), len(rec_item), _p(rec_item),
                             _p(rec_layer), _p(offsets), _p(nbrs), _p(eps), len(eps),
                             int(graph.max_level), nq, _p(qcodes), qcodes.shape[1], _p(qheaders), k,
                             ef_search, threads, C.byref(qo), _p(ids), _p(dists), _p(counts))
    if rc != 0:
        raise RuntimeError(f"orc_search failed: {rc}")
    return ids, dists, counts


def encode_key(index, mode, item, layer):
    out = np.zeros(8, np.uint8)
    lib().orc_encode_key(index, mode, item, layer, _p(out))
    return out.tobytes()


def roaring_serialize(ids):
    ids = np.ascontiguousarray(ids, np.uint32)
    n = lib().orc_roaring_serialize(_p(ids), len(ids), None)
    out = np.zeros(n, np.uint8)
    lib().orc_roaring_serialize(_p(ids), len(ids), _p(out))
    return out.tobytes()


def encode_kv(ds, graph, index=0, with_items=False):
    it = ds.items_struct()
    n = lib().orc_encode_kv(graph._h, C.byref(graph.opts), C.byref(it), index, int(with_items),
                            None, 0)
    out = np.zeros(n, np.uint8)
    lib().orc_encode_kv(graph._h, C.byref(graph.opts), C.byref(it), index, int(with_items),
                        _p(out), n)
    return parse_kv(out.tobytes())


def parse_kv(buf):
    recs, p = [], 0
    while p < len(buf):
        kl = int.from_bytes(buf[p:p + 4], "little")
        key = buf[p + 4:p + 4 + kl]
        p += 4 + kl
        vl = int.from_bytes(buf[p:p + 4], "little")
        recs.append((key, buf[p + 4:p + 4 + vl]))
        p += 4 + vl
    return recs
