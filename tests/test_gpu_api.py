"""The reference's own API-level tests, restated on the reference-shaped host API (hannoy_amd.api):
tests/test_basic.py, src/tests/writer.rs (write_one_vector, overwrite_one_item_incremental,
delete_one_item), src/tests/fuzz.rs invariants, src/tests/reader.rs."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hannoy_amd
    hannoy_amd.load_library()
    return hannoy_amd


def _dump_links(db, index=0):
    from hannoy_amd.api import MODE_LINKS, roaring_deserialize
    out = []
    for k, v in db.dump(index):
        idx, mode, item, layer = struct.unpack(">HBIB", k)
        if mode == MODE_LINKS:
            out.append([item, layer, roaring_deserialize(v[1:]).tolist()])
    return out


def test_python_binding_fixture(H, kat):
    """tests/test_basic.py:8-34 (KAT-8)"""
    k = kat["kat8"]
    db = H.Database(None, H.Metric.HAMMING)
    with db.writer(3, m=4, ef=10) as writer:
        for i, v in zip(k["ids"], k["vectors"]):
            writer.add_item(i, v)
    reader = db.reader(0)
    res = reader.by_vec(k["query"], n=k["k"])
    assert len(res) == k["n_hits"]
    assert list(res[0]) == k["first_hit"]
    with pytest.raises(H.InvalidVecDimension):
        db.writer(3).add_item(9, [1.0, 2.0])


def test_write_one_vector_snapshot(H, kat):
    """src/tests/writer.rs:109-128 — every record of the one-vector index, byte level"""
    c = kat["kat5"][0]
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(3, m=3)
    w.add_item(0, c["vector"])
    w.build(levels={0: c["level"]}, M0=3)
    recs = db.dump(0)
    meta = db.metadata(0)
    assert meta == {"distance": "euclidean", "dimensions": 3, "items": meta["items"],
                    "entry_points": meta["entry_points"], "max_level": 1}
    assert meta["items"].tolist() == [0] and meta["entry_points"].tolist() == [0]
    assert _dump_links(db) == c["links"]
    assert recs[1] == (bytes.fromhex("0000000000000100"), bytes.fromhex("000000000000000100000003"))
    assert recs[-1] == (bytes.fromhex("0000030000000000"),
                        b"\x00" + np.float32(0).tobytes() + np.array(c["vector"], np.float32).tobytes())


def test_overwrite_and_delete_snapshots(H, kat):
    """src/tests/writer.rs:376-437 and 589-677 through add_item / del_item / build"""
    k1, k2, k3, k4 = kat["kat1"], kat["kat2"], kat["kat3"], kat["kat4"]
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(2, m=3, ef=100)
    for i, v in zip(k1["ids"], k1["vectors"]):
        w.add_item(i, v)
    w.build(levels=dict(zip(k1["ids"], k1["levels"])), M0=3, batch_max=1)
    assert _dump_links(db) == k1["links"]
    assert db.metadata(0)["entry_points"].tolist() == k1["entry_points"]
    snapshot = dict(db.kv)
    # overwrite item 3 (KAT-2)
    w.add_item(3, k2["overwrite"]["vector"])
    w.build(levels={3: 0}, M0=3, batch_max=1)
    assert _dump_links(db) == k2["links"]
    assert db.metadata(0)["entry_points"].tolist() == k2["entry_points"]
    # back to the KAT-1 DB, delete item 3, then item 1 (KAT-3, KAT-4)
    db.kv = dict(snapshot)
    assert w.del_item(3) and not w.del_item(3)
    w.build(M0=3, batch_max=1)
    assert _dump_links(db) == k3["links"]
    m = db.metadata(0)
    assert m["entry_points"].tolist() == k3["entry_points"] and m["items"].tolist() == [0, 1, 2, 4, 5]
    w.del_item(1)
    w.build(M0=3, batch_max=1)
    assert _dump_links(db) == k4["links"]
    assert db.metadata(0)["entry_points"].tolist() == k4["entry_points"]


@pytest.mark.parametrize("M0", [32, 768])
def test_fuzz_invariants(H, M0):
    """src/tests/fuzz.rs:31-77, 83-143: random add/del batches on 32-d Cosine, M = 16, ef_construction 32,
    incremental build after every batch; after every build all items are findable (nns(1) of each stored
    vector returns itself) and no link points to a deleted item.  M0 = 768 is the reference's own pair
    (fuzz.rs:86-87): lists of several hundred links, fill_gaps_from_deleted on them (k_fill_gaps_wg)."""
    rng = np.random.default_rng(42)
    dim = 32
    db = H.Database(None, H.Metric.COSINE)
    w = db.writer(dim, m=16, ef=32)
    alive = {}
    for rnd in range(4):
        for _ in range(400 if rnd == 0 else 100):
            i = int(rng.integers(0, 1000))
            if i in alive and rng.random() < 0.4:
                assert w.del_item(i)
                del alive[i]
            else:
                v = rng.uniform(-1, 1, dim).astype(np.float32)
                w.add_item(i, v)
                alive[i] = v
        w.build(M0=M0)
        ids = sorted(alive)
        assert db.metadata(0)["items"].tolist() == ids
        for item, layer, nb in _dump_links(db):
            assert item in alive and set(nb) <= set(ids)
        assert sorted({i for i, l, _ in _dump_links(db) if l == 0}) == ids  # assert_validity
        r = db.reader(0)
        got, dists, cnt = r.by_vecs(np.stack([alive[i] for i in ids]), n=1, ef_search=100)
        r.close()
        assert np.all(cnt == 1)
        found = np.mean(got[:, 0] == np.array(ids, np.uint32))
        assert found >= 0.99, found
        assert np.all(dists[got[:, 0] == np.array(ids, np.uint32), 0] < 1e-6)


@pytest.mark.parametrize("n,dim", [(500, 64), (9999, 768)])
def test_all_items_are_reachable(H, n, dim):
    """src/tests/reader.rs:82-98: M = M0 = 6, nns(n).ef_search(n) from the zero vector finds all — at the
    reference's own largest case (proptest n in 1..10000, DIM 768): a result set of 9 999 entries lives in
    HBM (general walk kernel), every distance ties at 0.0"""
    rng = np.random.default_rng(7)
    db = H.Database(None, H.Metric.COSINE)
    w = db.writer(dim, m=6, ef=100)
    w.add_items(range(n), rng.uniform(-1, 1, (n, dim)).astype(np.float32))
    w.build(M0=6)
    r = db.reader(0)
    res = r.by_vec(np.zeros(dim, np.float32), n=n, ef_search=n)
    assert sorted(i for i, _ in res) == list(range(n))


def _rng42(H):
    """src/tests/mod.rs:145-147: StdRng::from_seed([42; 32]); writer tests use M = M0 = 3 (writer.rs:16-17)"""
    return H.StdRng.from_seed(bytes([42] * 32))


def _rand_index(H, metric, n, dim, m=16, seed=42, index=0, db=None):
    rng = np.random.default_rng(seed)
    db = db or H.Database(None, metric)
    w = db.writer(dim, index=index, m=m, ef=100)
    vecs = rng.uniform(0, 1, (n, dim)).astype(np.float32)
    w.add_items(range(n), vecs)
    w.build()
    return db, w, vecs, rng


def test_clear_small_database(H):
    """src/tests/writer.rs:21-43: clear() only touches its own index"""
    db = H.Database(None, H.Metric.COSINE)
    zero = db.writer(3, index=0)
    zero.add_item(0, [0.0, 1.0, 2.0])
    zero.clear()
    zero.builder().build()
    one = db.writer(3, index=1)
    one.add_item(0, [1.0, 2.0, 3.0])
    one.builder().build()
    db.writer(3, index=0).clear()
    assert db.dump(0) == []
    r = db.reader(1)
    assert r.item_vector(0).tolist() == [1.0, 2.0, 3.0]
    r.close()


def test_delete_all_but_one_item_and_build(H):
    """src/tests/writer.rs:47-65 (issue #52)"""
    db = H.Database(None, H.Metric.COSINE)
    w = db.writer(3)
    for i, v in ((1, [1, 2, 0]), (2, [2, 1, 0]), (3, [1, 0, 2]), (0, [0, 1, 2])):
        w.add_item(i, v)
    w.builder().build()
    w = db.writer(3)
    for i in (0, 2, 3):
        assert w.del_item(i)
    w.builder().build()
    assert db.metadata(0)["items"].tolist() == [1]
    r = db.reader(0)
    r.assert_validity()
    assert [i for i, _ in r.nns(10).by_vector([1, 2, 0]).into_nns()] == [1]
    r.close()


@pytest.mark.parametrize("item", [0xFFFFFFFF, 0xFFFFFFFE])
def test_use_u32_max_for_a_vec(H, item):
    """src/tests/writer.rs:67-107"""
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(3)
    w.add_item(item, [0.0, 1.0, 2.0])
    w.builder(_rng42(H)).build(3, 3)
    m = db.metadata(0)
    assert m["items"].tolist() == [item] and m["entry_points"].tolist() == [item] and m["max_level"] == 1
    assert _dump_links(db) == [[item, 0, []], [item, 1, []]]


def test_write_multiple_and_random_indexes(H):
    """src/tests/writer.rs:157-228: several indexes in one database do not interfere"""
    db = H.Database(None, H.Metric.EUCLIDEAN)
    for i in range(5):
        w = db.writer(3, index=i)
        w.add_item(0, [0.0, 1.0, 2.0])
        w.builder(_rng42(H)).build(3, 3)
    for i in range(5):
        m = db.metadata(i)
        assert (m["dimensions"], m["items"].tolist(), m["entry_points"].tolist(), m["max_level"]) == (3, [0], [0], 1)
        assert _dump_links(db, i) == [[0, 0, []], [0, 1, []]]
    rng = np.random.default_rng(42)
    db = H.Database(None, H.Metric.EUCLIDEAN)
    for index in rng.permutation(10):
        w = db.writer(10, index=int(index))
        for i in range(10):
            w.add_item(i, rng.random(10, dtype=np.float32))
        w.builder().build()
    for index in range(10):
        r = db.reader(index)
        r.assert_validity()
        assert r.n_items() == 10
        r.close()


@pytest.mark.parametrize("metric", ["EUCLIDEAN", "COSINE"])
def test_delete_one_item_in_a_one_item_db(H, metric):
    """src/tests/writer.rs:441-480, 550-586: the index becomes empty (metadata stays, no links)"""
    db = H.Database(None, getattr(H.Metric, metric))
    rng = _rng42(H)
    w = db.writer(2)
    w.add_item(0, [0.0, 0.0])
    w.builder(rng).build(3, 3)
    m = db.metadata(0)
    assert (m["items"].tolist(), m["entry_points"].tolist(), m["max_level"]) == ([0], [0], 1)
    assert _dump_links(db) == [[0, 0, []], [0, 1, []]]
    w = db.writer(2)
    assert w.del_item(0)
    w.builder(rng).build(3, 3)
    m = db.metadata(0)
    assert (m["items"].tolist(), m["entry_points"].tolist(), m["max_level"]) == ([], [], 0)
    assert len(db.dump(0)) == 2  # metadata + version
    r = db.reader(0)
    assert r.item_vector(0) is None
    r.close()


def test_delete_document_in_an_empty_index_74(H):
    """src/tests/writer.rs:482-548"""
    db = H.Database(None, H.Metric.EUCLIDEAN)
    rng = _rng42(H)
    w = db.writer(2)
    assert not w.del_item(0)
    w.add_item(0, [0.0, 0.0])
    w.builder(rng).build(3, 3)
    assert _dump_links(db) == [[0, 0, []], [0, 1, []]]
    w1, w2 = db.writer(2, index=0), db.writer(2, index=1)
    assert w1.del_item(0)
    assert not w2.del_item(0)
    w1.builder(rng).build(3, 3)
    w2.builder(rng).build(3, 3)
    for index in (0, 1):
        m = db.metadata(index)
        assert (m["items"].tolist(), m["entry_points"].tolist(), m["max_level"]) == ([], [], 0)
    r = db.reader(1)
    s = r.nns(10).by_vector([0.0, 0.0])
    assert s.into_nns() == [] and not s.did_cancel()
    r.close()


def test_delete_one_item_no_snapshots(H):
    """src/tests/writer.rs:680-730"""
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(2)
    for i in range(6):
        w.add_item(i, [float(i), 0.0])
    w.builder().build(3, 3)
    for gone in (3, 1):
        w = db.writer(2)
        assert w.del_item(gone)
        w.builder().build(3, 3)
    assert not db.writer(2).contains_item(3) and not db.writer(2).contains_item(1)
    assert not ({1, 3} & {i for i, _, _ in _dump_links(db)})
    r = db.reader(0)
    r.assert_validity()
    r.close()


def test_force_rebuild_and_search(H):
    """src/tests/writer.rs:749-775 + search_by_item tests (src/tests/reader.rs:113-143)"""
    db, w, vecs, rng = _rand_index(H, H.Metric.COSINE, 100, 768)
    before = _dump_links(db)
    db.writer(768).builder().force_rebuild()
    assert sorted({i for i, l, _ in _dump_links(db) if l == 0}) == list(range(100))
    assert not db.writer(768).need_build()
    r = db.reader(0)
    r.assert_validity()
    found = r.nns(10).by_item(0).into_nns()
    assert len(found) == 10 and 0 not in [i for i, _ in found]
    assert r.nns(10).by_item(101) is None
    # search_cancellation_works (src/tests/reader.rs:145-170)
    q = rng.random(768, dtype=np.float32)
    assert not r.nns(10).by_vector_with_cancellation(q, lambda: False).did_cancel()
    assert r.nns(10).by_vector_with_cancellation(q, lambda: True).did_cancel()
    assert not r.nns(10).by_item_with_cancellation(0, lambda: False).did_cancel()
    assert r.nns(10).by_item_with_cancellation(0, lambda: True).did_cancel()
    r.close()
    assert before  # the first build had links too


def test_search_on_candidates_has_right_num(H):
    """src/tests/reader.rs:41-78, several indexes in one database"""
    db = H.Database(None, H.Metric.COSINE)
    for index in range(1, 4):
        _rand_index(H, H.Metric.COSINE, 1000, 768, seed=index, index=index, db=db)
    rng = np.random.default_rng(0)
    for index in rng.permutation([1, 2, 3]):
        r = db.reader(int(index))
        q = rng.random(768, dtype=np.float32)
        cand = np.unique(rng.integers(0, 1000, 10)).astype(np.uint32)
        found = r.nns(10).candidates(cand).by_vector(q).into_nns()
        assert sorted(i for i, _ in found) == cand.tolist()
        one = rng.integers(0, 1000, 1).astype(np.uint32)
        found = r.nns(1).candidates(one).by_vector(q).into_nns()
        assert [i for i, _ in found] == one.tolist()
        r.close()


def test_quantized_iter_has_right_dimensions(H):
    """src/tests/reader.rs:17-38 (issue #78): a prime number of dimensions, binary quantized"""
    dim = 1063
    db = H.Database(None, H.Metric.BQ_COSINE)
    w = db.writer(dim)
    v = np.random.default_rng(42).random(dim, dtype=np.float32) - 0.5
    w.add_item(0, v)
    w.builder().build()
    r = db.reader(0)
    (_, new_vec), = list(r.iter())
    assert len(new_vec) == dim
    assert np.array_equal(new_vec, np.where(np.signbit(v), -1.0, 1.0).astype(np.float32))
    r.close()


def test_need_build_and_reader_open_errors(H):
    """writer.rs:423-436, reader.rs:387-417"""
    from hannoy_amd.api import MissingMetadata, NeedBuild, UnmatchingDistance
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(4)
    assert w.is_empty() and w.need_build()
    with pytest.raises(MissingMetadata):
        db.reader(0)
    w.add_items(range(20), np.random.default_rng(1).random((20, 4), dtype=np.float32))
    assert not w.is_empty() and w.need_build()
    w.builder().ef_construction(32).alpha(1.0).build()
    assert not w.need_build()
    w.add_item(99, [1, 2, 3, 4])
    assert w.need_build()
    with pytest.raises(NeedBuild):
        db.reader(0)
    w.builder().build()
    r = db.reader(0)
    assert r.n_items() == 21 and r.dimensions == 4 and r.version() == (0, 1, 3) and r.n_entrypoints() >= 1
    assert r.contains_item(99) and not r.contains_item(98) and not r.is_empty()
    assert r.item_vector(99).tolist() == [1, 2, 3, 4]
    r.close()
    db.distance = H.Metric.COSINE
    with pytest.raises(UnmatchingDistance):
        db.reader(0)


def test_prepare_changing_distance(H):
    """writer.rs:358-410: cosine -> binary quantized cosine keeps links and metadata, every item is
    re-encoded and marked updated; cosine -> euclidean drops links + metadata"""
    from hannoy_amd.api import MODE_ITEM, MODE_UPDATED, key
    db, w, vecs, rng = _rand_index(H, H.Metric.COSINE, 300, 70)
    vecs = vecs - 0.5
    w.add_items(range(300), vecs)
    w.builder().build()
    links_before = _dump_links(db)
    w2 = db.writer(70).prepare_changing_distance(H.Metric.BQ_COSINE)
    assert db.distance == H.Metric.BQ_COSINE and w2.need_build()
    assert _dump_links(db) == links_before and db.metadata(0) is not None
    assert all(key(0, MODE_UPDATED, i) in db.kv for i in range(300))
    v = db.kv[key(0, MODE_ITEM, 7)]
    assert len(v) == 1 + 4 + 16  # tag + norm header + 128 bits
    bits = np.unpackbits(np.frombuffer(v, np.uint8, offset=5), bitorder="little")[:70]
    assert np.array_equal(bits, (~np.signbit(vecs[7])).astype(np.uint8))
    w2.builder().build()
    r = db.reader(0)
    r.assert_validity()
    assert r.nns(1).by_item(7) is not None
    r.close()
    # a different family: links and metadata are dropped, the next build starts from scratch
    db, w, vecs, rng = _rand_index(H, H.Metric.COSINE, 200, 16)
    w3 = db.writer(16).prepare_changing_distance(H.Metric.EUCLIDEAN)
    assert _dump_links(db) == [] and db.metadata(0) is None
    w3.builder().build()
    r = db.reader(0)
    r.assert_validity()
    got = r.nns(1).by_vector(vecs[5]).into_nns()
    assert got[0][0] == 5 and got[0][1] == 0.0
    r.close()


def test_build_cancel_and_progress(H):
    """writer.rs:97-131: cancel -> BuildCancelled, progress reports inserted items"""
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(8)
    w.add_items(range(500), np.random.default_rng(3).random((500, 8), dtype=np.float32))
    before = db.dump() if hasattr(db, "dump") else sorted(db.kv.items())
    with pytest.raises(H.BuildCancelled):
        w.builder().cancel(lambda: True).build()
    # the reference's build runs in the caller's RwTxn, which the error aborts: nothing changed, the
    # `updated` stones are still there and the next build picks them up
    assert w.need_build()
    assert (db.dump() if hasattr(db, "dump") else sorted(db.kv.items())) == before
    w.builder().build()
    assert not w.need_build()
    r = db.reader(0)
    r.assert_validity()
    r.close()
    seen = []
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(8)
    w.add_items(range(500), np.random.default_rng(3).random((500, 8), dtype=np.float32))
    w.builder().progress(lambda done, total: seen.append((done, total))).build()
    assert seen and seen[-1] == (500, 500)


def test_multithreaded_reads(H, kat):
    """tests/test_basic.py:37-54: two threads, each with its own reader on the same database"""
    import threading
    k = kat["kat8"]
    db = H.Database(None, H.Metric.HAMMING)
    with db.writer(3, m=4, ef=10) as writer:
        for i, v in zip(k["ids"], k["vectors"]):
            writer.add_item(i, v)
    got, errs = {}, []

    def _read(q):
        try:
            reader = db.reader(0)
            got[tuple(q)] = reader.by_vec(q, 1)
            reader.close()
        except Exception as e:  # pragma: no cover
            errs.append(e)
    threads = [threading.Thread(target=_read, args=(q,)) for q in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs
    assert got[(1.0, 0.0, 0.0)] == [(0, 0.0)] and got[(0.0, 1.0, 0.0)] == [(1, 0.0)]


def test_database_path_persists_as_lmdb(H, tmp_path):
    """python.rs:60-100 + 305-314: a Database with a path is an LMDB environment; the index built
    and committed by one handle is what a second handle opens (Reader::open, reader.rs:387-431)"""
    from hannoy_amd import _capi as capi
    rng = np.random.default_rng(3)
    vecs = rng.normal(size=(3000, 48)).astype(np.float32)
    path = str(tmp_path / "env")
    db = H.Database(path, H.Metric.COSINE, name="vectors", env_size=1 << 30)
    with db.writer(48, m=8, ef=64) as w:
        w.add_items(range(3000), vecs)
    q = rng.normal(size=(20, 48)).astype(np.float32)
    want = db.reader(0).by_vecs(q, n=5)
    with capi.LmdbEnv(path + "/data.mdb", "vectors") as env:
        env.verify()
        st = env.stat()
        assert st["entries"] == len(db.kv) and st["map_size"] == 1 << 30
        assert env.items() == db.dump()
    db2 = H.Database(path, H.Metric.COSINE, name="vectors")
    assert db2.dump() == db.dump()
    r = db2.reader(0)
    assert r.n_items() == 3000
    assert all(np.array_equal(a, b) for a, b in zip(r.by_vecs(q, n=5), want))
    # incremental update through the reopened environment, committed again
    with db2.writer(48, m=8, ef=64) as w:
        w.del_item(17)
        w.add_item(5000, vecs[17])
    db3 = H.Database(path, H.Metric.COSINE, name="vectors")
    ids = db3.reader(0).item_ids().tolist()
    assert 17 not in ids and 5000 in ids and db3.dump() == db2.dump()
    db3.reader(0).assert_validity()


def test_graph_write_lmdb_equals_record_stream(H, tmp_path):
    """hny_encode_kv straight into hny_lmdb_writer_put: the environment holds exactly the records
    of the stream (Item values of 3 077 B sit on overflow pages, node.rs:136-140)"""
    from hannoy_amd import _capi as capi
    rng = np.random.default_rng(4)
    vecs = rng.normal(size=(4000, 768)).astype(np.float32)
    items = H.ItemSet.from_f32(H.COSINE, vecs)
    g = H.build(items, M=16, M0=32, ef_construction=48, seed=1)
    p = str(tmp_path / "data.mdb")
    g.write_lmdb(p, index=3, with_items=True)
    recs = g.encode_kv(3, with_items=True)
    with capi.LmdbEnv(p) as env:
        env.verify()
        big = [len(v) for k, v in recs if 8 + len(k) + len(v) > 2038]  # me_nodemax on 4 KiB pages
        assert len(big) == 4000 + 1  # every Item value + the Metadata record (8 KB item bitmap)
        assert env.stat()["overflow_pages"] == sum((15 + n) // 4096 + 1 for n in big)
        assert env.items() == recs
        k, v = recs[-1]
        assert env.get(k) == v and len(v) == 1 + 4 + 3072


def test_write_and_update_lot_of_random_points_with_snapshot(H, orc):
    """src/tests/writer.rs:130-155 with its two insta snapshots (tests/golden/kat9_100x30.json), through the
    Writer API: one StdRng::from_seed([42; 32]) feeds the 100 random vectors, the level draws of the first
    build, the 50 replacement vectors and the level draws of the second build; `build::<3, 3>` in strict mode
    with one insertion at a time (the reference's test pins one rayon thread) must leave exactly the Links
    records, entry points and max_level of the two dumps."""
    import json
    import os
    from tests.test_oracle_kat import kat9_inputs
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat9_100x30.json")) as f:
        k = json.load(f)
    v1, lv1, upd, v2, lv2 = kat9_inputs(orc, k)  # rng.gen::<f32>() restated in the oracle, checked against the dumps
    rng = _rng42(H)
    db = H.Database(None, H.Metric.EUCLIDEAN)
    w = db.writer(k["dim"])
    for i in range(k["n"]):
        w.add_item(i, v1[i])
    rng.drawn += k["n"] * k["dim"]  # what `std::array::from_fn(|_| rng.gen())` took from the generator
    w.builder(rng).build(3, 3, batch_max=1, x86_order=True)
    m = db.metadata(0)
    assert (m["entry_points"].tolist(), m["max_level"]) == (k["fresh"]["entry_points"], k["fresh"]["max_level"])
    assert _dump_links(db) == k["fresh"]["links"]
    w = db.writer(k["dim"])
    for j, i in enumerate(upd):
        w.add_item(int(i), v2[int(i)])
    rng.drawn += len(upd) * k["dim"]
    w.builder(rng).build(3, 3, batch_max=1, x86_order=True)
    m = db.metadata(0)
    assert (m["entry_points"].tolist(), m["max_level"]) == (k["updated"]["entry_points"], k["updated"]["max_level"])
    assert _dump_links(db) == k["updated"]["links"]
    r = db.reader(0)
    r.assert_validity()
    r.close()
