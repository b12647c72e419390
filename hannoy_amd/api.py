"""Reference-shaped host API on top of the C ABI: `Database` / `Writer` / `Reader` / `Metric` with the
names and argument meaning of hannoy's Python binding (/root/reference/hannoy.pyi,
src/python.rs) and the build logic of `Writer::build` (/root/reference/src/writer.rs:521-603).

The key/value store is an in-memory ordered map holding the byte-exact hannoy records (8-byte keys,
tagged values — key.rs:54-82, node.rs:130-174, metadata.rs:22-73, version.rs:33-60,
update_status.rs:8-33); LMDB itself is out of scope (no LMDB in this image).  All graph work goes
through libhannoy_amd.so (hny_build / hny_build_incremental / hny_builder_search_knn): there is no
CPU fallback.
"""
import enum
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
    """hannoy.pyi Database; `path`, `name`, `env_size` are accepted for signature parity only."""

    def __init__(self, path=None, distance=Metric.COSINE, name=None, env_size=None):
        capi.load_library()
        self.distance = distance
        self.kv = {}

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

    def build(self, levels=None, **opts):
        """Writer::build (writer.rs:521-603).  `levels`: optional {item id: level} for the items that
        get (re)inserted, instead of drawing them from StdRng::seed_from_u64(self.seed)."""
        db, index = self.db, self.index
        meta = db.metadata(index)
        indexed = set(meta["items"].tolist()) if meta else set()
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


class Reader:
    """hannoy.pyi Reader: `by_vec` = Reader::nns(n).by_vector (reader.rs:132-148) on the GPU."""

    def __init__(self, db, index=0):
        self.db, self.index = db, index
        meta = db.metadata(index)
        if meta is None:
            raise KeyError("MissingMetadata")  # Error::MissingMetadata
        self.meta = meta
        self.dimensions = meta["dimensions"]
        items = db.item_set(index, meta["items"], self.dimensions)
        prev = _StoredGraph(db, index)
        m0 = max([int(c) for c in np.diff(prev.offsets.astype(np.int64))[prev.rec_layer == 0]] + [1])
        mu = max([int(c) for c in np.diff(prev.offsets.astype(np.int64))[prev.rec_layer > 0]] + [1])
        self._b = capi.Builder(items, prev=prev, M=mu, M0=max(m0, mu), ef_construction=1)
        self._b.run()  # nothing to insert: loads the stored graph into HBM

    def by_vec(self, query, n=10, ef_search=200):
        q = np.asarray(query, np.float32)
        if q.ndim != 1 or len(q) != self.dimensions:
            raise InvalidVecDimension(f"expected {self.dimensions}, received {q.size}")
        qc, qh = capi.encode_vectors(self.db.distance.value, q[None, :])
        ids, dists, counts = self._b.search_knn(qc, qh, k=n, ef_search=ef_search)
        return [(int(ids[0, j]), float(dists[0, j])) for j in range(counts[0])]

    def by_vecs(self, queries, n=10, ef_search=200):
        qc, qh = capi.encode_vectors(self.db.distance.value, np.asarray(queries, np.float32))
        return self._b.search_knn(qc, qh, k=n, ef_search=ef_search)

    def close(self):
        self._b.close()
