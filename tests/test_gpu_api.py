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


def test_fuzz_invariants(H):
    """src/tests/fuzz.rs:31-77: random add/del batches on 32-d Cosine; after every build all items
    are findable (nns(1) of each stored vector returns itself) and no link points to a deleted item."""
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
        w.build()
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


def test_all_items_are_reachable(H):
    """src/tests/reader.rs:82-98: M = M0 = 6, nns(n).ef_search(n) from the zero vector finds all"""
    rng = np.random.default_rng(7)
    n, dim = 500, 64
    db = H.Database(None, H.Metric.COSINE)
    w = db.writer(dim, m=6, ef=100)
    w.add_items(range(n), rng.uniform(-1, 1, (n, dim)).astype(np.float32))
    w.build(M0=6)
    r = db.reader(0)
    res = r.by_vec(np.zeros(dim, np.float32), n=n, ef_search=n)
    assert sorted(i for i, _ in res) == list(range(n))
