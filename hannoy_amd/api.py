"""Reference-shaped host API on top of the C ABI: `Database` / `Writer` / `Reader` / `Metric` with the
names and argument meaning of hannoy's Python binding (/root/reference/hannoy.pyi,
src/python.rs) and the build logic of `Writer::build` (/root/reference/src/writer.rs:521-603).

The key/value store is an in-memory ordered map holding the byte-exact hannoy records (8-byte keys,
tagged values — key.rs:54-82, node.rs:130-174, metadata.rs:22-73, version.rs:33-60,
update_status.rs:8-33).  With a `path` it is persisted as an LMDB environment (`<path>/data.mdb`,
hny_lmdb_writer_* / hny_lmdb_*: the file format of LMDB restated, there is no liblmdb in this image):
loaded when the Database opens, rewritten when a transaction commits.  All graph work goes
through libhannoy_amd.so (hny_build / hny_build_incremental / hny_builder_search_knn): there is no
CPU fallback.
"""
import enum
import os
import struct

import numpy as np

from . import _capi as capi

MODE_METADATA, MODE_UPDATED, MODE_LINKS, MODE_ITEM = 0, 1, 2, 3  # node_id.rs:11-21
UPDATED, REMOVED = b"\x00", b"\x01"                                # update_status.rs:8-11


class Metric(enum.Enum):
    """hannoy.pyi Metric"""
    COSINE = capi.COSINE
    EUCLIDEAN = capi.EUCLIDEAN
    MANHATTAN = capi.MANHATTAN
    BQ_COSINE = capi.BQ_COSINE
    BQ_EUCLIDEAN = capi.BQ_EUCLIDEAN
    BQ_MANHATTAN = capi.BQ_MANHATTAN
    HAMMING = capi.HAMMING

    def __str__(self):
        return capi.METRIC_NAMES[self.value]


class InvalidVecDimension(ValueError):
    """Error::InvalidVecDimension (error.rs:19-26)"""


class MissingMetadata(KeyError):
    """Error::MissingMetadata (reader.rs:392)"""


class NeedBuild(RuntimeError):
    """Error::NeedBuild (reader.rs:408-417): items were added or removed since the last build"""


class UnmatchingDistance(ValueError):
    """Error::UnmatchingDistance (reader.rs:401-406)"""


def decode_vector(metric, code, dim):
    """UnalignedVectorCodec::to_vec truncated to `dim` (reader.rs:581-587): f32 as stored; `Binary`
    bits -> 0.0 / 1.0 (binary.rs:160-162); `BinaryQuantized` bits -> -1.0 / 1.0
    (binary_quantized.rs:156-158)."""
    code = np.ascontiguousarray(code, np.uint8)
    if metric in (capi.COSINE, capi.EUCLIDEAN, capi.MANHATTAN):
        return code.view(np.float32)[:dim].copy()
    bits = np.unpackbits(code, bitorder="little")[:dim].astype(np.float32)
    return bits if metric == capi.HAMMING else bits * 2.0 - 1.0


def key(index, mode, item=0, layer=0):
    """KeyCodec (key.rs:57-66): index u16 BE | mode u8 | item u32 BE | layer u8"""
    return struct.pack(">HBIB", index, mode, item, layer)


def roaring_deserialize(buf):
    """Portable RoaringFormatSpec without run containers (what roaring 0.10 writes)."""
    cookie, n = struct.unpack_from("<II", buf, 0)
    if cookie != 12346:
        raise ValueError(f"unsupported roaring cookie {cookie}")
    heads = [struct.unpack_from("<HH", buf, 8 + 4 * i) for i in range(n)]
    p = 8 + 4 * n + 4 * n  # descriptive header + offset header
    out = []
    for hi, card_m1 in heads:
        card = card_m1 + 1
        if card <= 4096:
            vals = np.frombuffer(buf, dtype="<u2", count=card, offset=p).astype(np.uint32)
            p += 2 * card
        else:
            bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8, count=8192, offset=p), bitorder="little")
            vals = np.nonzero(bits)[0].astype(np.uint32)
            p += 8192
        out.append(vals | np.uint32(hi << 16))
    return np.concatenate(out) if out else np.zeros(0, np.uint32)


class _StoredGraph:
    """the Links records + Metadata of one index, decoded (what FrozenReader::iter_links yields)"""

    def __init__(self, db, index):
        self.rec_item, self.rec_layer, offs, nb = [], [], [0], []
        lo, hi = key(index, MODE_LINKS), key(index, MODE_LINKS, 0xFFFFFFFF, 0xFF)
        for k in sorted(k for k in db.kv if lo <= k <= hi):
            _, _, item, layer = struct.unpack(">HBIB", k)
            v = db.kv[k]
            assert v[0] == 1  # LINKS_TAG, node.rs:21-22
            ids = roaring_deserialize(v[1:])
            self.rec_item.append(item)
            self.rec_layer.append(layer)
            nb.append(ids)
            offs.append(offs[-1] + len(ids))
        self.rec_item = np.array(self.rec_item, np.uint32)
        self.rec_layer = np.array(self.rec_layer, np.uint8)
        self.offsets = np.array(offs, np.uint64)
        self.nbrs = np.concatenate(nb) if nb else np.zeros(0, np.uint32)
        meta = db.metadata(index)
        self.entry_points = meta["entry_points"] if meta else np.zeros(0, np.uint32)
        self.max_level = meta["max_level"] if meta else 0


class Database:
    """hannoy.pyi Database (python.rs:60-100): `path` = the LMDB environment directory (None = in
    memory only), `name` = a named database inside it, `env_size` = the map size recorded in the
    meta page."""

    def __init__(self, path=None, distance=Metric.COSINE, name=None, env_size=None):
        capi.load_library()
        self.distance = distance
        self.kv = {}
        self.path, self.name, self.env_size = path, name, env_size
        if path is not None:
            os.makedirs(path, exist_ok=True)
            f = os.path.join(path, "data.mdb")
            if os.path.exists(f):
                with capi.LmdbEnv(f, name) as env:
                    self.kv = dict(env.items())

    def commit(self):
        """RwTxn::commit (python.rs:312): with a `path`, the records become `<path>/data.mdb`"""
        if self.path is None:
            return
        tmp = os.path.join(self.path, "data.mdb.tmp")
        with capi.LmdbWriter(tmp, self.name, 0, self.env_size or 0) as w:
            for k in sorted(self.kv):
                w.put(k, self.kv[k])
        os.replace(tmp, os.path.join(self.path, "data.mdb"))

    def writer(self, dimensions, index=0, m=16, ef=96):
        return Writer(self, dimensions, index, m, ef)

    def reader(self, index=0):
        return Reader(self, index)

    # -- record level helpers -------------------------------------------------------------------
    def dump(self, index=None):
        """records in LMDB key order"""
        return [(k, self.kv[k]) for k in sorted(self.kv)
                if index is None or struct.unpack(">H", k[:2])[0] == index]

    def metadata(self, index):
        v = self.kv.get(key(index, MODE_METADATA))
        if v is None:
            return None
        z = v.index(b"\0")  # MetadataCodec, metadata.rs:50-73
        dims, rsz = struct.unpack_from(">II", v, z + 1)
        p = z + 9
        items = roaring_deserialize(v[p:p + rsz])
        rest = v[p + rsz:]
        eps = np.frombuffer(rest[:-1], dtype=np.uint32).copy() if rest else np.zeros(0, np.uint32)
        return {"distance": v[:z].decode(), "dimensions": dims, "items": items, "entry_points": eps,
                "max_level": rest[-1] if rest else 0}

    def item_ids(self, index):
        lo, hi = key(index, MODE_ITEM), key(index, MODE_ITEM, 0xFFFFFFFF, 0xFF)
        return np.array(sorted(struct.unpack(">HBIB", k)[2] for k in self.kv if lo <= k <= hi), np.uint32)

    def item_set(self, index, ids, dim):
        """ItemSet (codec bytes + headers) for the given ascending ids"""
        metric = self.distance.value
        hb, vb = capi.header_bytes(metric), capi.vector_bytes(metric, dim)
        codes = np.zeros((len(ids), vb), np.uint8)
        hdrs = np.zeros((len(ids), hb), np.uint8)
        for r, i in enumerate(ids):
            v = self.kv[key(index, MODE_ITEM, int(i))]
            hdrs[r] = np.frombuffer(v, np.uint8, hb, 1)
            codes[r] = np.frombuffer(v, np.uint8, vb, 1 + hb)
        return capi.ItemSet(metric, dim, np.asarray(ids, np.uint32), codes, hdrs)


class Writer:
    """hannoy.pyi Writer / src/writer.rs Writer + HannoyBuilder.  As in python.rs:305-314 the build
    runs when the `with` block exits (M0 = 2*m, StdRng::seed_from_u64(42), python.rs:118-120,261)."""

    def __init__(self, db, dimensions, index=0, m=16, ef=96):
        self.db, self.dimensions, self.index, self.m, self.ef = db, int(dimensions), index, m, ef
        self.alpha, self.seed = 1.0, 42
        self.last_graph = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.build()
            self.db.commit()
        return False

    def add_item(self, item, vector):
        """writer.rs:462-480"""
        v = np.asarray(vector, np.float32)
        if v.ndim != 1 or len(v) != self.dimensions:
            raise InvalidVecDimension(f"expected {self.dimensions}, received {v.size}")
        self.add_items([item], v[None, :])

    def add_items(self, ids, matrix):
        matrix = np.ascontiguousarray(matrix, np.float32)
        if matrix.ndim != 2 or matrix.shape[1] != self.dimensions:
            raise InvalidVecDimension(f"expected {self.dimensions}, received {matrix.shape[-1]}")
        codes, hdrs = capi.encode_vectors(self.db.distance.value, matrix)
        for r, i in enumerate(ids):
            self.db.kv[key(self.index, MODE_ITEM, int(i))] = b"\x00" + hdrs[r].tobytes() + codes[r].tobytes()
            self.db.kv[key(self.index, MODE_UPDATED, int(i))] = UPDATED

    def del_item(self, item):
        """writer.rs:483-495"""
        if self.db.kv.pop(key(self.index, MODE_ITEM, int(item)), None) is None:
            return False
        self.db.kv[key(self.index, MODE_UPDATED, int(item))] = REMOVED
        return True

    def contains_item(self, item):
        return key(self.index, MODE_ITEM, int(item)) in self.db.kv

    def is_empty(self):
        """writer.rs:418-420"""
        return len(self.db.item_ids(self.index)) == 0

    def need_build(self):
        """writer.rs:423-436: an `updated` stone exists, or there is no metadata yet"""
        lo, hi = key(self.index, MODE_UPDATED), key(self.index, MODE_UPDATED, 0xFFFFFFFF, 0xFF)
        return any(lo <= k <= hi for k in self.db.kv) or self.db.metadata(self.index) is None

    def item_vector(self, item):
        """writer.rs:439-445"""
        v = self.db.kv.get(key(self.index, MODE_ITEM, int(item)))
        if v is None:
            return None
        hb = capi.header_bytes(self.db.distance.value)
        return decode_vector(self.db.distance.value, np.frombuffer(v, np.uint8, offset=1 + hb), self.dimensions)

    def iter(self):
        """writer.rs:457-459: (item id, vector) in id order"""
        for i in self.db.item_ids(self.index):
            yield int(i), self.item_vector(int(i))

    def clear(self):
        """writer.rs:498-511: every record of this index"""
        for k in [k for k in self.db.kv if struct.unpack(">H", k[:2])[0] == self.index]:
            del self.db.kv[k]

    def prepare_changing_distance(self, distance):
        """writer.rs:358-410: re-encode every item for the new distance, mark all of them updated;
        links and metadata go unless the new distance is the binary-quantized form of the old one."""
        old = self.db.distance
        if distance != old:
            new_name, old_name = str(distance), str(old)
            if not (new_name.startswith("binary quantized ") and new_name[len("binary quantized "):] == old_name):
                llo, lhi = key(self.index, MODE_LINKS), key(self.index, MODE_LINKS, 0xFFFFFFFF, 0xFF)
                for k in [k for k in self.db.kv if llo <= k <= lhi]:
                    del self.db.kv[k]
                self.db.kv.pop(key(self.index, MODE_METADATA), None)
            ids = self.db.item_ids(self.index)
            hb = capi.header_bytes(old.value)
            vecs = []
            for i in ids:  # vector.to_vec(): the whole decoded vector, padding included (:377)
                v = self.db.kv[key(self.index, MODE_ITEM, int(i))]
                code = np.frombuffer(v, np.uint8, offset=1 + hb)
                full = len(code) // 4 if old.value in (capi.COSINE, capi.EUCLIDEAN, capi.MANHATTAN) else len(code) * 8
                vecs.append(decode_vector(old.value, code, full))
            self.db.distance = distance
            if len(ids):
                codes, hdrs = capi.encode_vectors(distance.value, np.stack(vecs))
                for r, i in enumerate(ids):
                    self.db.kv[key(self.index, MODE_ITEM, int(i))] = b"\x00" + hdrs[r].tobytes() + codes[r].tobytes()
                    self.db.kv[key(self.index, MODE_UPDATED, int(i))] = UPDATED
        w = Writer(self.db, self.dimensions, self.index, self.m, self.ef)
        w.alpha, w.seed = self.alpha, self.seed
        return w

    def builder(self, rng=42):
        """writer.rs:514-517: HannoyBuilder with the options of writer.rs:34-58.  `rng`: a u64 seed
        (StdRng::seed_from_u64 per build, what python.rs:261 does) or a capi.StdRng that is carried
        across builds like the reference's `&mut rng`."""
        return HannoyBuilder(self, rng)

    def force_rebuild(self, levels=None, **opts):
        """writer.rs:246-259, 610-638: drop every Links record and re-link all indexed items (the
        previous entry points and max_level still seed the build, hnsw.rs:236-267)."""
        meta = self.db.metadata(self.index)
        if meta is None:
            raise MissingMetadata("The metadata must be there")
        keep = set(meta["items"].tolist())
        snapshot = dict(self.db.kv)  # the reference works inside one RwTxn that is aborted on error
        llo, lhi = key(self.index, MODE_LINKS), key(self.index, MODE_LINKS, 0xFFFFFFFF, 0xFF)
        for k in [k for k in self.db.kv if llo <= k <= lhi]:  # delete_links_from_db(&item_ids)
            if struct.unpack(">HBIB", k)[2] in keep:
                del self.db.kv[k]
        try:
            return self.build(levels=levels, relink_all_items=True, **opts)
        except BaseException:
            self.db.kv = snapshot
            raise

    def build(self, levels=None, relink_all_items=False, rng=None, **opts):
        """Writer::build (writer.rs:521-603).  `levels`: optional {item id: level} for the items that
        get (re)inserted, instead of drawing them from StdRng::seed_from_u64(self.seed)."""
        snapshot = dict(self.db.kv)  # writer.rs:521-603 runs in the caller's RwTxn: an error (cancel,
        try:                         # unsupported, device, OOM) aborts it and nothing is changed
            return self._build(levels, relink_all_items, rng, opts)
        except BaseException:
            self.db.kv = snapshot
            raise

    def _build(self, levels, relink_all_items, rng, opts):
        db, index = self.db, self.index
        meta = db.metadata(index)
        indexed = set(meta["items"].tolist()) if meta else set()
        if relink_all_items:  # writer.rs:539-540
            item_indices, to_delete, to_insert = indexed, [], sorted(indexed)
        else:
            # reset_and_retrieve_updated_items (writer.rs:645-688)
            lo, hi = key(index, MODE_UPDATED), key(index, MODE_UPDATED, 0xFFFFFFFF, 0xFF)
            upd = {struct.unpack(">HBIB", k)[2]: db.kv[k] for k in list(db.kv) if lo <= k <= hi}
            for i in upd:
                del db.kv[key(index, MODE_UPDATED, i)]
            all_updated = set(upd)
            deleted = {i for i, s in upd.items() if s == REMOVED}
            item_indices = ((all_updated - deleted) | indexed) - deleted  # writer.rs:548-553
            to_delete = sorted(all_updated - item_indices)
            to_insert = sorted(item_indices & all_updated)
        ids = np.array(sorted(item_indices), np.uint32)
        items = db.item_set(index, ids, self.dimensions)
        kw = dict(M=self.m, M0=2 * self.m, ef_construction=self.ef, alpha=self.alpha, seed=self.seed)
        kw.update(opts)
        if rng is not None and levels is None:  # get_random_level per to_insert id, ascending (hnsw.rs:142-149)
            levels = dict(zip(to_insert, rng.draw_levels(kw["M"], len(to_insert)).tolist()))
        prev = _StoredGraph(db, index)
        if len(prev.rec_item) == 0 and meta is None:
            if levels is not None:
                items.levels = np.array([levels[int(i)] for i in ids], np.uint8)
            g = capi.build(items, **kw)
        else:
            if levels is not None:
                items.levels = np.array([levels[int(i)] for i in to_insert], np.uint8)
            g = capi.build_incremental(items, prev, to_insert, to_delete, **kw)
        # write-back: every Links record of the new state (hnsw.rs:195-213) — stale ones and those
        # of deleted items go (writer.rs:580) —, then Metadata and Version (writer.rs:585-600)
        llo, lhi = key(index, MODE_LINKS), key(index, MODE_LINKS, 0xFFFFFFFF, 0xFF)
        for k in [k for k in db.kv if llo <= k <= lhi]:
            del db.kv[k]
        for k, v in g.encode_kv(index, with_items=False):
            db.kv[k] = v
        self.last_graph = g
        return g


class HannoyBuilder:
    """writer.rs:34-259 HannoyBuilder: `.ef_construction()`, `.alpha()`, `.cancel()`, `.progress()`
    then `.build(M, M0)` / `.force_rebuild(M, M0)` (const generics in the reference)."""

    def __init__(self, writer, rng=42):
        self.writer, self.rng = writer, rng
        self._ef, self._alpha, self._cancel, self._progress = 100, 1.0, None, None  # writer.rs:47-58

    def ef_construction(self, ef):
        self._ef = int(ef)
        return self

    def alpha(self, alpha):
        self._alpha = float(alpha)
        return self

    def cancel(self, fn):
        self._cancel = fn
        return self

    def progress(self, fn):
        self._progress = fn
        return self

    def _opts(self, M, M0, opts):
        kw = dict(M=M, M0=M0, ef_construction=self._ef, alpha=self._alpha,
                  cancel=self._cancel, progress=self._progress)
        if isinstance(self.rng, capi.StdRng):
            kw["rng"] = self.rng
        else:
            kw["seed"] = int(self.rng)
        kw.update(opts)
        return kw

    def build(self, M=16, M0=32, levels=None, **opts):
        return self.writer.build(levels=levels, **self._opts(M, M0, opts))

    def force_rebuild(self, M=16, M0=32, levels=None, **opts):
        return self.writer.force_rebuild(levels=levels, **self._opts(M, M0, opts))


class Searched:
    """reader.rs:34-57"""

    def __init__(self, nns, did_cancel=False):
        self.nns, self._did_cancel = nns, did_cancel

    def did_cancel(self):
        return self._did_cancel

    def into_nns(self):
        return self.nns


class QueryBuilder:
    """reader.rs:60-262.  The *_with_cancellation variants hand `cancel_fn` to hny_builder_nns, which
    polls it while the batch runs and stops taking queries once it fires (hny_query_opts.cancel): a
    cancelled search returns what was found so far."""

    def __init__(self, reader, count):
        self.reader, self.count = reader, int(count)
        self._ef, self._candidates = 100, None          # reader.rs:23, 614
        self._linear_below, self._ratio = 1000, 1.0     # reader.rs:28, 31

    def ef_search(self, ef):
        self._ef = int(ef)
        return self

    def candidates(self, ids):
        self._candidates = np.ascontiguousarray(list(ids) if not isinstance(ids, np.ndarray) else ids, np.uint32)
        return self

    def linear_below(self, threshold):
        self._linear_below = int(threshold)
        return self

    def linear_below_ratio(self, ratio):
        assert 0.0 <= ratio <= 1.0, "linear scan threshold ratio must be between 0.0 and 1.0"
        self._ratio = float(ratio)
        return self

    def _kw(self):
        return dict(k=self.count, ef_search=self._ef, candidates=self._candidates,
                    linear_below=self._linear_below, linear_below_ratio=self._ratio)

    def by_vectors(self, vectors, cancel=None):
        """batched by_vector: (ids [nq, count], distances, counts)"""
        r = self.reader
        q = np.ascontiguousarray(vectors, np.float32)
        if q.ndim != 2 or q.shape[1] != r.dimensions:
            raise InvalidVecDimension(f"expected {r.dimensions}, received {q.shape[-1]}")
        qc, qh = capi.encode_vectors(r.db.distance.value, q)
        return r._b.nns(qc, qh, cancel=cancel, **self._kw())

    def by_items(self, items, cancel=None):
        """batched by_item; counts == capi.NNS_NONE where the reference returns None"""
        return self.reader._b.nns(query_items=np.ascontiguousarray(items, np.uint32), cancel=cancel, **self._kw())

    def by_vector(self, vector):
        """reader.rs:132-148"""
        return self.by_vector_with_cancellation(vector, lambda: False)

    def by_vector_with_cancellation(self, vector, cancel_fn):
        """reader.rs:167-186"""
        q = np.asarray(vector, np.float32)
        if q.ndim != 1 or len(q) != self.reader.dimensions:
            raise InvalidVecDimension(f"expected {self.reader.dimensions}, received {q.size}")
        ids, dists, counts = self.by_vectors(q[None, :], cancel=cancel_fn)
        return Searched([(int(ids[0, j]), float(dists[0, j])) for j in range(counts[0])],
                        self.reader._b.did_cancel)

    def by_item(self, item):
        """reader.rs:81-90"""
        return self.by_item_with_cancellation(item, lambda: False)

    def by_item_with_cancellation(self, item, cancel_fn):
        """reader.rs:108-119"""
        ids, dists, counts = self.by_items([item], cancel=cancel_fn)
        if counts[0] == capi.NNS_NONE:
            return None
        return Searched([(int(ids[0, j]), float(dists[0, j])) for j in range(counts[0])],
                        self.reader._b.did_cancel)


class Reader:
    """hannoy.pyi Reader + src/reader.rs Reader: `open` checks (reader.rs:387-431), accessors
    (:546-608), `nns(count)` -> QueryBuilder (:611-619), `by_vec` = Reader::nns(n).by_vector on the
    GPU."""

    def __init__(self, db, index=0):
        self.db, self.index = db, index
        meta = db.metadata(index)
        if meta is None:
            raise MissingMetadata("MissingMetadata")  # Error::MissingMetadata
        if meta["distance"] != str(db.distance):
            raise UnmatchingDistance(f"expected {meta['distance']}, received {db.distance}")
        lo, hi = key(index, MODE_UPDATED), key(index, MODE_UPDATED, 0xFFFFFFFF, 0xFF)
        if any(lo <= k <= hi for k in db.kv):
            raise NeedBuild(index)
        self.meta = meta
        self._dimensions = meta["dimensions"]
        items = db.item_set(index, meta["items"], self._dimensions)
        prev = _StoredGraph(db, index)
        m0 = max([int(c) for c in np.diff(prev.offsets.astype(np.int64))[prev.rec_layer == 0]] + [1])
        mu = max([int(c) for c in np.diff(prev.offsets.astype(np.int64))[prev.rec_layer > 0]] + [1])
        self._graph = prev
        self._b = capi.Builder(items, prev=prev, load=True, M=mu, M0=max(m0, mu), ef_construction=1)

    @property
    def dimensions(self):
        return self._dimensions

    def n_entrypoints(self):
        return len(self.meta["entry_points"])

    def n_items(self):
        return len(self.meta["items"])

    def item_ids(self):
        return self.meta["items"]

    def version(self):
        v = self.db.kv.get(key(self.index, MODE_METADATA, 1))
        return struct.unpack(">III", v) if v else (0, 0, 0)

    def n_nodes(self):
        """reader.rs:576-578: every record of the database"""
        return len(self.db.kv) or None

    def item_vector(self, item):
        v = self.db.kv.get(key(self.index, MODE_ITEM, int(item)))
        if v is None:
            return None
        hb = capi.header_bytes(self.db.distance.value)
        return decode_vector(self.db.distance.value, np.frombuffer(v, np.uint8, offset=1 + hb), self._dimensions)

    def is_empty(self):
        return len(self.db.item_ids(self.index)) == 0

    def contains_item(self, item):
        return key(self.index, MODE_ITEM, int(item)) in self.db.kv

    def iter(self):
        for i in self.db.item_ids(self.index):
            yield int(i), self.item_vector(int(i))

    def nns(self, count):
        return QueryBuilder(self, count)

    def assert_validity(self):
        """reader.rs:905-948: every item is linked, links only name existing items, entry points exist"""
        g, items = self._graph, set(self.db.item_ids(self.index).tolist())
        assert items == set(self.meta["items"].tolist()), "Item records differ from the metadata"
        assert set(g.nbrs.tolist()) <= items, "links to items that are not in the database"
        assert items == set(g.rec_item.tolist()), "each item should have one or more Links records"
        assert set(np.asarray(self.meta["entry_points"]).tolist()) <= items

    def by_vec(self, query, n=10, ef_search=200):
        q = np.asarray(query, np.float32)
        if q.ndim != 1 or len(q) != self._dimensions:
            raise InvalidVecDimension(f"expected {self._dimensions}, received {q.size}")
        qc, qh = capi.encode_vectors(self.db.distance.value, q[None, :])
        ids, dists, counts = self._b.search_knn(qc, qh, k=n, ef_search=ef_search)
        return [(int(ids[0, j]), float(dists[0, j])) for j in range(counts[0])]

    def by_vecs(self, queries, n=10, ef_search=200):
        qc, qh = capi.encode_vectors(self.db.distance.value, np.asarray(queries, np.float32))
        return self._b.search_knn(qc, qh, k=n, ef_search=ef_search)

    def close(self):
        self._b.close()
