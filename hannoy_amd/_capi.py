"""ctypes binding of the C ABI in include/hannoy_amd.h (hannoy_amd/libhannoy_amd.so).

The shared library is the product: there is no Python or CPU fallback.  Importing works without a
GPU (so that symbol/ABI checks can run anywhere); every computing call needs a gfx950 device and
raises HannoyError otherwise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HNY_LIB") or os.path.join(HERE, "libhannoy_amd.so")

COSINE, EUCLIDEAN, MANHATTAN, HAMMING, BQ_COSINE, BQ_EUCLIDEAN, BQ_MANHATTAN = range(7)
METRIC_NAMES = ["cosine", "euclidean", "manhattan", "hamming", "binary quantized cosine",
                "binary quantized euclidean", "binary quantized manhattan"]

OK, ERR_INVALID_ARG, ERR_CANCELLED, ERR_MISSING_KEY, ERR_INVALID_DIM = 0, -1, -2, -3, -4
ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_DEVICE, ERR_OOM = -5, -6, -7, -8

EXPORTED = [
    "hny_build", "hny_graph_free", "hny_builder_create", "hny_builder_reset",
    "hny_builder_next_batch", "hny_builder_search", "hny_builder_apply", "hny_builder_sync",
    "hny_builder_finish", "hny_builder_destroy", "hny_builder_set_profiling", "hny_batch_size", "hny_builder_distances",
    "hny_builder_search_knn", "hny_vector_bytes", "hny_header_bytes", "hny_encode_vectors",
    "hny_encode_kv", "hny_last_error", "hny_version", "hny_draw_levels", "hny_build_incremental",
    "hny_builder_create_incremental", "hny_builder_fill_gaps", "hny_encode_vectors_gpu",
    "hny_builder_nns", "hny_draw_levels_from_seed", "hny_builder_load",
    "hny_builder_apply_begin", "hny_builder_apply_deferred", "hny_builder_apply_merge",
    "hny_builder_exch_stride_u64", "hny_builder_stream", "hny_default_batch_max", "hny_selftest_lane_ops",
    "hny_lmdb_writer_open", "hny_lmdb_writer_put", "hny_lmdb_writer_finish", "hny_lmdb_writer_abort",
    "hny_lmdb_open", "hny_lmdb_stat_get", "hny_lmdb_get", "hny_lmdb_scan", "hny_lmdb_close",
    "hny_multi_builder_create", "hny_multi_builder_run", "hny_multi_builder_set_profiling",
    "hny_multi_builder_world", "hny_multi_builder_collectives", "hny_multi_builder_replica",
    "hny_multi_builder_destroy", "hny_abi_sizes", "hny_set_graph_cache",
]
ERR_IO = -9
NNS_NONE = 0xFFFFFFFF  # by_item: the reference returns None


CANCEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class QueryOpts(C.Structure):
    _fields_ = [("k", C.c_uint32), ("ef_search", C.c_uint32), ("has_candidates", C.c_int32),
                ("candidates", C.c_void_p), ("n_candidates", C.c_uint64), ("linear_below", C.c_uint32),
                ("linear_below_ratio", C.c_float), ("cancel", CANCEL_FN), ("cancel_ctx", C.c_void_p),
                ("did_cancel", C.POINTER(C.c_int32))]


class HannoyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hannoy_amd error {code}: {msg}")
        self.code = code


class BuildCancelled(HannoyError):
    """Error::BuildCancelled (/root/reference/src/error.rs:58-59)"""


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)
KV_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8),
                      C.c_size_t)


class LmdbStat(C.Structure):
    _fields_ = [("page_size", C.c_uint32), ("depth", C.c_uint32), ("branch_pages", C.c_uint64),
                ("leaf_pages", C.c_uint64), ("overflow_pages", C.c_uint64), ("entries", C.c_uint64),
                ("last_pgno", C.c_uint64), ("txnid", C.c_uint64), ("map_size", C.c_uint64)]


class BuildOpts(C.Structure):
    _fields_ = [("metric", C.c_int32), ("dim", C.c_uint32), ("M", C.c_uint32), ("M0", C.c_uint32),
                ("ef_construction", C.c_uint32), ("alpha", C.c_float), ("seed", C.c_uint64),
                ("cancel", CANCEL_FN), ("cancel_ctx", C.c_void_p),
                ("progress", PROGRESS_FN), ("progress_ctx", C.c_void_p),
                ("batch_frac", C.c_double), ("batch_max", C.c_uint32), ("device", C.c_int32),
                ("x86_order", C.c_int32), ("n_gpus", C.c_int32), ("devices", C.POINTER(C.c_int32)),
                ("schedule", C.c_uint32), ("reserved_", C.c_uint32)]


SCHED_NO_SHUFFLE, SCHED_LEVEL_ORDER_ID, SCHED_UPDATE_NO_RAMP = 1, 2, 4  # hny_build_opts.schedule (HNY_SCHED_*)


class Items(C.Structure):
    _fields_ = [("n", C.c_uint64), ("ids", C.c_void_p), ("vectors", C.c_void_p),
                ("stride", C.c_size_t), ("headers", C.c_void_p), ("header_size", C.c_size_t),
                ("levels", C.c_void_p)]


class GraphStruct(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("rec_item", C.POINTER(C.c_uint32)),
                ("rec_layer", C.POINTER(C.c_uint8)), ("rec_offset", C.POINTER(C.c_uint64)),
                ("neighbours", C.POINTER(C.c_uint32)), ("entry_points", C.POINTER(C.c_uint32)),
                ("n_entry_points", C.c_uint32), ("max_level", C.c_uint32),
                ("n_links_added", C.c_uint64), ("n_distance_evals", C.c_uint64),
                ("n_evals_walk", C.c_uint64), ("n_evals_prune", C.c_uint64),
                ("n_evals_apply", C.c_uint64), ("n_batches", C.c_uint64),
                ("t_upload_s", C.c_double), ("t_build_s", C.c_double), ("t_export_s", C.c_double),
                ("n_tie_pool_overflow", C.c_uint64),
                ("t_walk_kernels_s", C.c_double), ("t_prune_kernels_s", C.c_double),
                ("t_sort_kernels_s", C.c_double), ("t_apply_kernels_s", C.c_double),
                ("n_walk_launches", C.c_uint64)]


class PrevGraph(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("rec_item", C.c_void_p), ("rec_layer", C.c_void_p),
                ("rec_offset", C.c_void_p), ("neighbours", C.c_void_p), ("entry_points", C.c_void_p),
                ("n_entry_points", C.c_uint32), ("max_level", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("first", C.c_uint64), ("count", C.c_uint32), ("level", C.c_uint32),
                ("n_layers", C.c_uint32), ("sel_stride_u64", C.c_uint32)]


_lib = None


def load_library():
    """Loads the HIP extension; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.  If this library were
    # loaded first it would register its kernels with /opt/rocm's runtime and later calls could bind
    # to torch's (global scope) — "no HIP device" / invalid device function.  Importing torch first
    # (when it is installed) makes every HIP symbol resolve to the one runtime torch also uses, which
    # the multi-GPU driver needs anyway (torch tensors are handed to the C ABI as device pointers).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    # HNY_LIB: another build of this same library (python -m hannoy_amd.buildlib --out PATH), for A/B
    # measurements of two kernel variants on one box (scripts/r2_ab_lib.sh); still no CPU fallback
    lib_path = os.environ.get("HNY_LIB") or LIB_PATH
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build it with `python -m hannoy_amd.buildlib` "
            "(hipcc --offload-arch=gfx950). hannoy_amd has no CPU fallback.")
    L = C.CDLL(lib_path)
    vp = C.c_void_p
    L.hny_build.restype = C.c_int
    L.hny_build.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), C.POINTER(C.POINTER(GraphStruct))]
    L.hny_graph_free.argtypes = [C.POINTER(GraphStruct)]
    L.hny_build_incremental.restype = C.c_int
    L.hny_build_incremental.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), vp, C.c_uint64, vp,
                                        C.c_uint64, C.POINTER(PrevGraph),
                                        C.POINTER(C.POINTER(GraphStruct))]
    L.hny_builder_create_incremental.restype = C.c_int
    L.hny_builder_create_incremental.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), vp, C.c_uint64,
                                                 vp, C.c_uint64, C.POINTER(PrevGraph), C.POINTER(vp)]
    L.hny_builder_load.restype = C.c_int
    L.hny_builder_load.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), C.POINTER(PrevGraph), C.POINTER(vp)]
    L.hny_builder_fill_gaps.restype = C.c_int
    L.hny_builder_fill_gaps.argtypes = [vp]
    L.hny_builder_create.restype = C.c_int
    L.hny_builder_create.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), C.POINTER(vp)]
    for name in ("hny_builder_reset", "hny_builder_sync"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    L.hny_builder_set_profiling.restype = C.c_int
    L.hny_builder_set_profiling.argtypes = [vp, C.c_int]
    L.hny_builder_next_batch.restype = C.c_int
    L.hny_builder_next_batch.argtypes = [vp, C.POINTER(Batch)]
    L.hny_builder_search.restype = C.c_int
    L.hny_builder_search.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.hny_builder_apply.restype = C.c_int
    L.hny_builder_apply.argtypes = [vp, vp]
    L.hny_builder_apply_begin.restype = C.c_int
    L.hny_builder_apply_begin.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.hny_builder_apply_deferred.restype = C.c_int
    L.hny_builder_apply_deferred.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.hny_builder_apply_merge.restype = C.c_int
    L.hny_builder_apply_merge.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.hny_builder_exch_stride_u64.restype = C.c_uint32
    L.hny_builder_exch_stride_u64.argtypes = [vp]
    L.hny_builder_finish.restype = C.c_int
    L.hny_builder_finish.argtypes = [vp, C.POINTER(C.POINTER(GraphStruct))]
    L.hny_builder_destroy.argtypes = [vp]
    L.hny_draw_levels.restype = C.c_int
    L.hny_draw_levels.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, vp]
    L.hny_batch_size.restype = C.c_uint32
    L.hny_batch_size.argtypes = [C.c_double, C.c_uint32, C.c_uint64]
    L.hny_builder_distances.restype = C.c_int
    L.hny_builder_distances.argtypes = [vp, C.c_uint64, vp, vp, vp]
    L.hny_builder_search_knn.restype = C.c_int
    L.hny_builder_search_knn.argtypes = [vp, C.c_uint64, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32,
                                         vp, vp, vp]
    L.hny_builder_nns.restype = C.c_int
    L.hny_builder_nns.argtypes = [vp, C.POINTER(QueryOpts), C.c_uint64, vp, C.c_size_t, vp, vp, vp, vp, vp]
    L.hny_draw_levels_from_seed.restype = C.c_int
    L.hny_draw_levels_from_seed.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, vp]
    L.hny_vector_bytes.restype = C.c_size_t
    L.hny_vector_bytes.argtypes = [C.c_int32, C.c_uint32]
    L.hny_header_bytes.restype = C.c_size_t
    L.hny_header_bytes.argtypes = [C.c_int32]
    L.hny_encode_vectors.restype = C.c_int
    L.hny_encode_vectors.argtypes = [C.c_int32, C.c_uint32, C.c_uint64, vp, vp, vp]
    L.hny_encode_vectors_gpu.restype = C.c_int
    L.hny_encode_vectors_gpu.argtypes = [C.c_int32, C.c_uint32, C.c_uint64, vp, vp, vp, C.c_int32]
    L.hny_encode_kv.restype = C.c_int
    L.hny_encode_kv.argtypes = [C.POINTER(GraphStruct), C.POINTER(BuildOpts), C.POINTER(Items),
                                C.c_uint16, C.c_int, KV_SINK, vp]
    L.hny_lmdb_writer_open.restype = C.c_int
    L.hny_lmdb_writer_open.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p, C.POINTER(vp)]
    L.hny_lmdb_writer_put.restype = C.c_int
    L.hny_lmdb_writer_put.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.hny_lmdb_writer_finish.restype = C.c_int
    L.hny_lmdb_writer_finish.argtypes = [vp]
    L.hny_lmdb_writer_abort.argtypes = [vp]
    L.hny_lmdb_open.restype = C.c_int
    L.hny_lmdb_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.hny_lmdb_stat_get.restype = C.c_int
    L.hny_lmdb_stat_get.argtypes = [vp, C.POINTER(LmdbStat)]
    L.hny_lmdb_get.restype = C.c_int
    L.hny_lmdb_get.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.hny_lmdb_scan.restype = C.c_int
    L.hny_lmdb_scan.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, KV_SINK, vp]
    L.hny_lmdb_close.argtypes = [vp]
    L.hny_last_error.restype = C.c_char_p
    L.hny_version.restype = C.c_char_p
    L.hny_multi_builder_create.restype = C.c_int
    L.hny_multi_builder_create.argtypes = [C.POINTER(BuildOpts), C.POINTER(Items), C.POINTER(vp)]
    L.hny_multi_builder_run.restype = C.c_int
    L.hny_multi_builder_run.argtypes = [vp, C.POINTER(C.POINTER(GraphStruct))]
    L.hny_multi_builder_set_profiling.restype = C.c_int
    L.hny_multi_builder_set_profiling.argtypes = [vp, C.c_int]
    L.hny_multi_builder_world.restype = C.c_uint32
    L.hny_multi_builder_world.argtypes = [vp]
    L.hny_multi_builder_collectives.restype = C.c_uint64
    L.hny_multi_builder_collectives.argtypes = [vp]
    L.hny_multi_builder_replica.restype = vp
    L.hny_multi_builder_replica.argtypes = [vp, C.c_uint32]
    L.hny_multi_builder_destroy.argtypes = [vp]
    L.hny_abi_sizes.restype = C.c_uint32
    L.hny_abi_sizes.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    _check_abi(L)
    _lib = L
    return L


# the order of hny_abi_sizes (HNY_ABI_* in include/hannoy_amd.h)
ABI_STRUCTS = None


def _check_abi(L):
    """sizeof of every struct declared here against the library's own (hny_abi_sizes): a field added
    to the header and not to this binding fails the import instead of corrupting a call"""
    global ABI_STRUCTS
    ABI_STRUCTS = [("hny_build_opts", BuildOpts), ("hny_items", Items), ("hny_graph", GraphStruct),
                   ("hny_prev_graph", PrevGraph), ("hny_batch", Batch), ("hny_query_opts", QueryOpts),
                   ("hny_lmdb_stat", LmdbStat)]
    out = (C.c_uint32 * len(ABI_STRUCTS))()
    n = L.hny_abi_sizes(out, len(ABI_STRUCTS))
    if n != len(ABI_STRUCTS):
        raise ImportError(f"hannoy_amd ABI mismatch: the library knows {n} public structs, this binding {len(ABI_STRUCTS)}")
    for (name, cls), size in zip(ABI_STRUCTS, out):
        if C.sizeof(cls) != size:
            raise ImportError(f"hannoy_amd ABI mismatch: sizeof({name}) is {size} in the library, "
                              f"{C.sizeof(cls)} in hannoy_amd/_capi.py")


def abi_sizes():
    """{struct name: sizeof in the loaded library}"""
    L = load_library()
    out = (C.c_uint32 * len(ABI_STRUCTS))()
    L.hny_abi_sizes(out, len(ABI_STRUCTS))
    return {name: int(v) for (name, _), v in zip(ABI_STRUCTS, out)}


def _check(rc):
    if rc != OK:
        msg = load_library().hny_last_error().decode("utf-8", "replace")
        raise (BuildCancelled if rc == ERR_CANCELLED else HannoyError)(rc, msg)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_graph_cache(max_bytes):
    """hny_set_graph_cache: opt-in recycling of released export arrays (0 = off, the default)"""
    L = load_library()
    L.hny_set_graph_cache.restype = None
    L.hny_set_graph_cache.argtypes = [C.c_size_t]
    L.hny_set_graph_cache(int(max_bytes))


def default_batch_max(n_items):
    """what batch_max = 0 selects for an index of n_items (hny_default_batch_max)"""
    L = load_library()
    L.hny_default_batch_max.restype = C.c_uint32
    L.hny_default_batch_max.argtypes = [C.c_uint64]
    return int(L.hny_default_batch_max(int(n_items)))


def selftest_lane_ops(device=-1):
    """hny_selftest_lane_ops: per-lane mismatch masks (all zero = the cross-lane primitives agree with
    __shfl_xor); raises HannoyError if they do not"""
    L = load_library()
    L.hny_selftest_lane_ops.restype = C.c_int
    L.hny_selftest_lane_ops.argtypes = [C.c_int32, C.POINTER(C.c_uint32)]
    out = (C.c_uint32 * 64)()
    _check(L.hny_selftest_lane_ops(int(device), out))
    return list(out)


def draw_levels(seed, M, n):
    out = np.zeros(n, np.uint8)
    _check(load_library().hny_draw_levels(seed, M, n, _p(out)))
    return out


class StdRng:
    """rand 0.8.5 `StdRng` as far as a build uses it: the 32-byte seed plus how many levels were
    drawn so far (get_random_level costs one u32 each, hnsw.rs:113-119), so that one generator can
    be carried across builds like `writer.builder(&mut rng)` does."""

    def __init__(self, seed32, drawn=0):
        self.seed = bytes(seed32)
        assert len(self.seed) == 32
        self.drawn = drawn

    @classmethod
    def from_seed(cls, seed32):
        """SeedableRng::from_seed (the reference's test rng: [42; 32], src/tests/mod.rs:145-147)"""
        return cls(seed32)

    @classmethod
    def seed_from_u64(cls, state):
        """SeedableRng::seed_from_u64 (rand_core 0.6): PCG32 expansion (python.rs:261 uses 42)"""
        out = b""
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
            xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            out += (((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF).to_bytes(4, "little")
        return cls(out)

    def draw_levels(self, M, n):
        out = np.zeros(n, np.uint8)
        seed = (C.c_uint8 * 32).from_buffer_copy(self.seed)
        _check(load_library().hny_draw_levels_from_seed(seed, self.drawn, M, n, _p(out)))
        self.drawn += n
        return out


def vector_bytes(metric, dim):
    return load_library().hny_vector_bytes(metric, dim)


def header_bytes(metric):
    return load_library().hny_header_bytes(metric)


def encode_vectors(metric, vecs, gpu=False, device=-1):
    """UnalignedVectorCodec::from_slice + Distance::new_header for a [n, dim] f32 matrix
    (gpu=True: hny_encode_vectors_gpu, byte-identical)."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    n, dim = vecs.shape
    codes = np.zeros((n, vector_bytes(metric, dim)), np.uint8)
    headers = np.zeros((n, header_bytes(metric)), np.uint8)
    if gpu:
        _check(load_library().hny_encode_vectors_gpu(metric, dim, n, _p(vecs), _p(codes), _p(headers),
                                                     device))
    else:
        _check(load_library().hny_encode_vectors(metric, dim, n, _p(vecs), _p(codes), _p(headers)))
    return codes, headers


class ItemSet:
    """What FrozenReader hands to the builder (/root/reference/src/parallel.rs:33-45)."""

    def __init__(self, metric, dim, ids, codes, headers, levels=None):
        self.metric, self.dim = int(metric), int(dim)
        self.ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.headers = np.ascontiguousarray(headers, dtype=np.uint8)
        self.levels = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        self.n = len(self.ids)

    @classmethod
    def from_f32(cls, metric, vecs, ids=None, levels=None):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        codes, headers = encode_vectors(metric, vecs)
        ids = np.arange(len(vecs), dtype=np.uint32) if ids is None else ids
        return cls(metric, vecs.shape[1], ids, codes, headers, levels)

    def struct(self):
        return Items(self.n, _p(self.ids).value, _p(self.codes).value,
                     self.codes.shape[1] if self.codes.ndim == 2 else 0, _p(self.headers).value,
                     self.headers.shape[1] if self.headers.ndim == 2 else 0,
                     None if self.levels is None else _p(self.levels).value)


def make_opts(metric, dim, M=16, M0=32, ef_construction=100, alpha=1.0, seed=42, batch_frac=0.0,
              batch_max=0, device=-1, cancel=None, progress=None, x86_order=False, n_gpus=0, devices=None,
              schedule=0):
    o = BuildOpts()
    o.schedule = int(schedule)
    o.metric, o.dim, o.M, o.M0 = metric, dim, M, M0
    o.ef_construction, o.alpha, o.seed = ef_construction, alpha, seed
    o.batch_frac, o.batch_max, o.device = batch_frac, batch_max, device
    o.x86_order = int(bool(x86_order))
    keep = []
    o.n_gpus = int(n_gpus)
    if devices is not None:  # one replica per listed GPU, RCCL all-gathers between them (hny_multi.cpp)
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        o.devices = C.cast(arr, C.POINTER(C.c_int32))
        o.n_gpus = int(n_gpus) or len(devices)
        keep.append(arr)
    if cancel is not None:
        fn = CANCEL_FN(lambda _ctx: 1 if cancel() else 0)
        o.cancel = fn
        keep.append(fn)
    if progress is not None:
        fn = PROGRESS_FN(lambda _ctx, done, total: progress(done, total))
        o.progress = fn
        keep.append(fn)
    o._keep = keep
    return o


class Graph:
    """Host copy of hny_graph (records the write loop of hnsw.rs:195-213 consumes)."""

    _ARRAYS = ("rec_item", "rec_layer", "offsets", "nbrs", "entry_points")

    def __init__(self, gp, opts=None, items=None):
        g = gp.contents
        self.max_level = g.max_level
        for f in ("n_links_added", "n_distance_evals", "n_evals_walk", "n_evals_prune",
                  "n_evals_apply", "n_batches", "t_upload_s", "t_build_s", "t_export_s",
                  "n_tie_pool_overflow", "t_walk_kernels_s", "t_prune_kernels_s",
                  "t_sort_kernels_s", "t_apply_kernels_s", "n_walk_launches"):
            setattr(self, f, getattr(g, f))
        self._gp, self._opts, self._items = gp, opts, items

    def __getattr__(self, name):
        # numpy copies of the library-owned arrays, made on first use (a 1M x 768 index has 125 MB
        # of them; the product is the hny_graph itself, these are for tests and scripts)
        if name not in Graph._ARRAYS or self.__dict__.get("_gp") is None:
            raise AttributeError(name)
        g = self._gp.contents
        nrec = g.n_records
        if name == "rec_item":
            v = np.ctypeslib.as_array(g.rec_item, (max(nrec, 1),))[:nrec].copy()
        elif name == "rec_layer":
            v = np.ctypeslib.as_array(g.rec_layer, (max(nrec, 1),))[:nrec].copy()
        elif name == "offsets":
            v = np.ctypeslib.as_array(g.rec_offset, (nrec + 1,)).copy()
        elif name == "nbrs":
            nl = int(self.offsets[-1])
            v = np.ctypeslib.as_array(g.neighbours, (max(nl, 1),))[:nl].copy()
        else:
            ne = g.n_entry_points
            v = np.ctypeslib.as_array(g.entry_points, (max(ne, 1),))[:ne].copy()
        self.__dict__[name] = v
        return v

    def __del__(self):
        try:
            if getattr(self, "_gp", None) is not None:
                load_library().hny_graph_free(self._gp)
                self._gp = None
        except Exception:  # interpreter shutdown
            pass

    def as_dict(self):
        return {(int(self.rec_item[r]), int(self.rec_layer[r])):
                self.nbrs[int(self.offsets[r]):int(self.offsets[r + 1])].tolist()
                for r in range(len(self.rec_item))}

    def encode_kv(self, index=0, with_items=False):
        """Byte-exact (key, value) records in LMDB key order (hny_encode_kv)."""
        out = []

        def sink(_ctx, k, kl, v, vl):
            out.append((bytes(k[:kl]), bytes(v[:vl])))
            return 0
        cb = KV_SINK(sink)
        it = self._items.struct()
        _check(load_library().hny_encode_kv(self._gp, C.byref(self._opts), C.byref(it), index,
                                            int(with_items), cb, None))
        return out

    def write_lmdb(self, path, index=0, with_items=True, name=None, page_size=0, map_size=0):
        """The records of a fresh index straight into an LMDB `data.mdb`: hny_encode_kv with
        hny_lmdb_writer_put as its sink (no Python in the record loop)."""
        L = load_library()
        w = LmdbWriter(path, name, page_size, map_size)
        it = self._items.struct()
        sink = C.cast(L.hny_lmdb_writer_put, KV_SINK)
        try:
            _check(L.hny_encode_kv(self._gp, C.byref(self._opts), C.byref(it), index, int(with_items), sink, w.handle))
        except Exception:
            w.abort()
            raise
        w.finish()


class LmdbWriter:
    """hny_lmdb_writer_*: bulk loader of an LMDB `data.mdb` (keys strictly ascending)."""

    def __init__(self, path, name=None, page_size=0, map_size=0):
        self._w = C.c_void_p()
        _check(load_library().hny_lmdb_writer_open(os.fsencode(path), page_size, map_size,
                                                   None if name is None else name.encode(), C.byref(self._w)))

    @property
    def handle(self):
        return self._w

    def put(self, k, v):
        _check(load_library().hny_lmdb_writer_put(self._w, k, len(k), v, len(v)))

    def finish(self):
        w, self._w = self._w, None
        _check(load_library().hny_lmdb_writer_finish(w))

    def abort(self):
        if self._w is not None:
            load_library().hny_lmdb_writer_abort(self._w)
            self._w = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.finish()
        else:
            self.abort()
        return False


class LmdbEnv:
    """hny_lmdb_open / get / scan: read side of a `data.mdb` (RoTxn + Database::get / iter)."""

    def __init__(self, path, name=None):
        self._e = C.c_void_p()
        _check(load_library().hny_lmdb_open(os.fsencode(path), None if name is None else name.encode(),
                                            C.byref(self._e)))

    def stat(self):
        st = LmdbStat()
        _check(load_library().hny_lmdb_stat_get(self._e, C.byref(st)))
        return {f: getattr(st, f) for f, _ in LmdbStat._fields_}

    def get(self, k):
        val, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = load_library().hny_lmdb_get(self._e, k, len(k), C.byref(val), C.byref(n))
        if rc < 0:
            _check(rc)
        return C.string_at(val, n.value) if rc == 1 else None

    def items(self, lo=None, hi=None):
        """(key, value) pairs with lo <= key <= hi in key order; a full scan also verifies the
        page counts of the MDB_db record"""
        out = []

        def sink(_ctx, k, kl, v, vl):
            out.append((C.string_at(k, kl), C.string_at(v, vl)))
            return 0
        cb = KV_SINK(sink)
        _check(load_library().hny_lmdb_scan(self._e, lo, len(lo) if lo else 0, hi, len(hi) if hi else 0, cb, None))
        return out

    def verify(self):
        _check(load_library().hny_lmdb_scan(self._e, None, 0, None, 0, KV_SINK(), None))

    def close(self):
        if self._e is not None:
            load_library().hny_lmdb_close(self._e)
            self._e = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


def build(items, **kw):
    """hny_build: the drop-in for HnswBuilder::build (/root/reference/src/hnsw.rs:122-216)."""
    o = make_opts(items.metric, items.dim, **kw)
    it = items.struct()
    gp = C.POINTER(GraphStruct)()
    _check(load_library().hny_build(C.byref(o), C.byref(it), C.byref(gp)))
    return Graph(gp, o, items)


def _prev_struct(prev):
    keep = [np.ascontiguousarray(prev.rec_item, np.uint32), np.ascontiguousarray(prev.rec_layer, np.uint8),
            np.ascontiguousarray(prev.offsets, np.uint64),
            np.ascontiguousarray(prev.nbrs if len(prev.nbrs) else np.zeros(1), np.uint32),
            np.ascontiguousarray(prev.entry_points, np.uint32)]
    pg = PrevGraph(len(keep[0]), _p(keep[0]).value, _p(keep[1]).value, _p(keep[2]).value,
                   _p(keep[3]).value, _p(keep[4]).value, len(keep[4]), int(prev.max_level))
    return pg, keep


def build_incremental(items, prev, to_insert, to_delete, **kw):
    """hny_build_incremental: `items` = every item present after the update (levels, if any: one per
    to_insert id); `prev` = graph of the previous build (anything with rec_item/rec_layer/offsets/
    nbrs/entry_points/max_level)."""
    o = make_opts(items.metric, items.dim, **kw)
    it = items.struct()
    ins = np.ascontiguousarray(to_insert, np.uint32)
    dl = np.ascontiguousarray(to_delete, np.uint32)
    keep = [np.ascontiguousarray(prev.rec_item, np.uint32), np.ascontiguousarray(prev.rec_layer, np.uint8),
            np.ascontiguousarray(prev.offsets, np.uint64),
            np.ascontiguousarray(prev.nbrs if len(prev.nbrs) else np.zeros(1), np.uint32),
            np.ascontiguousarray(prev.entry_points, np.uint32)]
    pg = PrevGraph(len(keep[0]), _p(keep[0]).value, _p(keep[1]).value, _p(keep[2]).value,
                   _p(keep[3]).value, _p(keep[4]).value, len(keep[4]), int(prev.max_level))
    gp = C.POINTER(GraphStruct)()
    _check(load_library().hny_build_incremental(C.byref(o), C.byref(it), _p(ins), len(ins), _p(dl),
                                                len(dl), C.byref(pg), C.byref(gp)))
    return Graph(gp, o, items)


class MultiBuilder:
    """hny_multi_builder_*: one resident replica per GPU of this node in ONE process (one host thread
    per GPU, RCCL all-gathers on the builders' streams, hny_multi.cpp); every run() is a complete fresh
    build whose records equal the one-GPU build's."""

    def __init__(self, items, devices=None, n_gpus=0, **kw):
        self.items = items
        self.opts = make_opts(items.metric, items.dim, n_gpus=n_gpus, devices=devices, **kw)
        self._h = C.c_void_p()
        self._replicas = []
        it = items.struct()
        _check(load_library().hny_multi_builder_create(C.byref(self.opts), C.byref(it), C.byref(self._h)))

    @property
    def world(self):
        return int(load_library().hny_multi_builder_world(self._h))

    @property
    def n_collectives(self):
        return int(load_library().hny_multi_builder_collectives(self._h))

    def set_profiling(self, on=True):
        _check(load_library().hny_multi_builder_set_profiling(self._h, int(on)))

    def run(self):
        gp = C.POINTER(GraphStruct)()
        _check(load_library().hny_multi_builder_run(self._h, C.byref(gp)))
        return Graph(gp, self.opts, self.items)

    def replica(self, rank=0):
        """the resident builder of `rank` as a Builder (library-owned: close() on it is a no-op)"""
        h = load_library().hny_multi_builder_replica(self._h, rank)
        if not h:
            raise IndexError(rank)
        b = Builder.__new__(Builder)
        b.items, b.opts, b.incremental = self.items, self.opts, False
        b._h, b._borrowed = C.c_void_p(h), True
        b._mb = self  # the borrowed handle lives exactly as long as this multi-builder: keep it alive ...
        self._replicas.append(b)
        return b

    def close(self):
        if self._h:
            for b in self._replicas:  # ... and make a replica handle unusable once it is gone
                b._h = C.c_void_p()
            self._replicas = []
            load_library().hny_multi_builder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Builder:
    """Stepwise builder (hny_builder_*): vectors stay resident in HBM across reset()/rebuilds."""
    _borrowed = False

    def __init__(self, items, prev=None, to_insert=(), to_delete=(), load=False, **kw):
        """prev given -> incremental builder on top of a stored graph (hny_builder_create_incremental);
        with load=True the stored graph is only loaded for searching (hny_builder_load)"""
        self.items = items
        self.opts = make_opts(items.metric, items.dim, **kw)
        self._h = C.c_void_p()
        self.incremental = prev is not None
        it = items.struct()
        if prev is None:
            _check(load_library().hny_builder_create(C.byref(self.opts), C.byref(it), C.byref(self._h)))
        elif load:
            pg, _keep = _prev_struct(prev)
            _check(load_library().hny_builder_load(C.byref(self.opts), C.byref(it), C.byref(pg), C.byref(self._h)))
        else:
            ins = np.ascontiguousarray(to_insert, np.uint32)
            dl = np.ascontiguousarray(to_delete, np.uint32)
            pg, _keep = _prev_struct(prev)
            _check(load_library().hny_builder_create_incremental(
                C.byref(self.opts), C.byref(it), _p(ins), len(ins), _p(dl), len(dl), C.byref(pg),
                C.byref(self._h)))

    def close(self):
        if self._h and not self._borrowed:
            load_library().hny_builder_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(load_library().hny_builder_reset(self._h))

    def set_profiling(self, on=True):
        _check(load_library().hny_builder_set_profiling(self._h, int(on)))

    def next_batch(self):
        b = Batch()
        _check(load_library().hny_builder_next_batch(self._h, C.byref(b)))
        return b

    def search(self, lo, hi, sel_ptr=None):
        _check(load_library().hny_builder_search(self._h, lo, hi, sel_ptr))

    def apply(self, sel_ptr=None):
        _check(load_library().hny_builder_apply(self._h, sel_ptr))

    def apply_begin(self, sel_ptr=None):
        """emit / sort / append links; returns the number of targets whose list must be re-pruned"""
        n = C.c_uint32()
        _check(load_library().hny_builder_apply_begin(self._h, sel_ptr, C.byref(n)))
        return n.value

    def apply_deferred(self, rank=0, world=1, exch_ptr=None):
        _check(load_library().hny_builder_apply_deferred(self._h, rank, world, exch_ptr))

    def apply_merge(self, exch_ptr=None, rank=0, world=1):
        _check(load_library().hny_builder_apply_merge(self._h, exch_ptr, rank, world))

    @property
    def exch_stride_u64(self):
        return load_library().hny_builder_exch_stride_u64(self._h)

    def sync(self):
        _check(load_library().hny_builder_sync(self._h))

    @property
    def stream_ptr(self):
        """hipStream_t every step is enqueued on (for collectives ordered on the same stream)"""
        L = load_library()
        L.hny_builder_stream.restype = C.c_void_p
        L.hny_builder_stream.argtypes = [C.c_void_p]
        return L.hny_builder_stream(self._h) or 0

    def run(self):
        """All batches on this GPU (what hny_build loops over)."""
        n = 0
        while True:
            b = self.next_batch()
            if b.count == 0:
                break
            self.search(0, b.count)
            self.apply()
            n += 1
        if self.incremental:
            self.fill_gaps()
        return n

    def fill_gaps(self):
        _check(load_library().hny_builder_fill_gaps(self._h))

    def finish(self):
        gp = C.POINTER(GraphStruct)()
        _check(load_library().hny_builder_finish(self._h, C.byref(gp)))
        return Graph(gp, self.opts, self.items)

    def distances(self, slot_a, slot_b):
        a = np.ascontiguousarray(slot_a, np.uint32)
        b = np.ascontiguousarray(slot_b, np.uint32)
        out = np.zeros(len(a), np.float32)
        _check(load_library().hny_builder_distances(self._h, len(a), _p(a), _p(b), _p(out)))
        return out

    def search_knn(self, qcodes, qheaders, k=10, ef_search=100):
        """Reader::nns(k).by_vector (/root/reference/src/reader.rs:132-148) on the built graph."""
        qcodes = np.ascontiguousarray(qcodes, np.uint8)
        qheaders = np.ascontiguousarray(qheaders, np.uint8)
        nq = qcodes.shape[0]
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        _check(load_library().hny_builder_search_knn(self._h, nq, _p(qcodes), qcodes.shape[1],
                                                     _p(qheaders), k, ef_search, _p(ids),
                                                     _p(dists), _p(counts)))
        return ids, dists, counts

    def nns(self, qcodes=None, qheaders=None, k=10, ef_search=100, candidates=None, query_items=None,
            linear_below=1000, linear_below_ratio=1.0, cancel=None):
        """Reader::nns(k).ef_search(..).candidates(..).linear_below(..).by_vector / .by_item
        (/root/reference/src/reader.rs:60-262).  counts == NNS_NONE where by_item returns None.
        cancel: the closure of the *_with_cancellation variants; self.did_cancel tells whether it fired."""
        qo = QueryOpts()
        qo.k, qo.ef_search = k, ef_search
        flag = C.c_int32(0)
        if cancel is not None:
            fn = CANCEL_FN(lambda _ctx: 1 if cancel() else 0)
            qo.cancel = fn
            qo.did_cancel = C.pointer(flag)
        self._cancel_flag = flag
        cand = None
        if candidates is not None:
            cand = np.ascontiguousarray(candidates, np.uint32)
            qo.has_candidates, qo.candidates, qo.n_candidates = 1, cand.ctypes.data, len(cand)
        qo.linear_below, qo.linear_below_ratio = linear_below, linear_below_ratio
        if query_items is not None:
            query_items = np.ascontiguousarray(query_items, np.uint32)
            nq, qc, qs, qh, qi = len(query_items), None, 0, None, _p(query_items)
        else:
            qcodes = np.ascontiguousarray(qcodes, np.uint8)
            qheaders = np.ascontiguousarray(qheaders, np.uint8)
            nq, qc, qs, qh, qi = qcodes.shape[0], _p(qcodes), qcodes.shape[1], _p(qheaders), None
        ids = np.zeros((nq, k), np.uint32)
        dists = np.zeros((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        _check(load_library().hny_builder_nns(self._h, C.byref(qo), nq, qc, qs, qh, qi, _p(ids), _p(dists),
                                              _p(counts)))
        self.did_cancel = bool(flag.value)
        return ids, dists, counts
