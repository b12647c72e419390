"""CPU: the oracle's restated Reader (QueryBuilder with candidates / linear scan / by_item,
/root/reference/src/reader.rs:60-262, 621-711, 809-896) against brute force and against the
reference's own property tests (src/tests/reader.rs:42-78, 114-143)."""
import numpy as np
import pytest

from conftest import draw_levels


@pytest.fixture(scope="module")
def small(orc):
    rng = np.random.default_rng(0)
    n, dim = 3000, 24
    x = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint32) * 3 + 1
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, x, draw_levels(n, 16, 7), ids)
    g = orc.build(ds, M=16, M0=32, ef=64)
    q = rng.standard_normal((40, dim)).astype(np.float32)
    qc = orc.encode_vectors(orc.EUCLIDEAN, q)
    qh = orc.make_headers(orc.EUCLIDEAN, dim, qc)
    return rng, x, ids, ds, g, q, qc, qh


def _exact(x, ids, q, mask, k):
    d = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    d = np.where(mask[None, :], d, np.inf)
    return [set(ids[np.argsort(d[r])[:k]].tolist()) for r in range(len(q))]


def test_candidates_filter(orc, small):
    rng, x, ids, ds, g, q, qc, qh = small
    cand = np.concatenate([ids[rng.choice(len(ids), 1500, replace=False)], [2, 5, 100000]]).astype(np.uint32)
    truth = _exact(x, ids, q, np.isin(ids, cand), 10)
    for lb in (0, 1000, 5000):  # HNSW walk with the filter / (1500 >= 1000) still HNSW / linear scan
        i, d, c = orc.search(ds, g, qc, qh, k=10, ef_search=200, candidates=cand, linear_below=lb)
        assert (c == 10).all()
        assert np.all(np.diff(d.view(np.uint32).astype(np.int64), axis=1) >= 0)
        rec = np.mean([len(truth[r] & set(i[r].tolist())) / 10 for r in range(len(q))])
        assert rec == 1.0 if lb == 5000 else rec >= 0.98
    # few candidates through the HNSW path: the exhaustive fallback finds every one of them
    i, d, c = orc.search(ds, g, qc, qh, k=10, ef_search=100, candidates=ids[:7], linear_below=0)
    assert (c == 7).all() and all(sorted(i[r, :7].tolist()) == ids[:7].tolist() for r in range(len(q)))
    i, d, c = orc.search(ds, g, qc, qh, k=1, ef_search=100, candidates=ids[5:6], linear_below=0)
    assert (c == 1).all() and (i[:, 0] == ids[5]).all()
    # nothing can match (reader.rs:652-654)
    assert not orc.search(ds, g, qc, qh, k=10, candidates=np.array([0, 2], np.uint32))[2].any()
    assert not orc.search(ds, g, qc, qh, k=10, candidates=np.zeros(0, np.uint32))[2].any()
    # the ratio gate (reader.rs:637): both conditions must hold for the linear scan
    a = orc.search(ds, g, qc, qh, k=5, ef_search=8, candidates=ids[:300], linear_below_ratio=0.01)
    b = orc.search(ds, g, qc, qh, k=5, ef_search=8, candidates=ids[:300], linear_below=0)
    assert np.array_equal(a[0], b[0])


def test_by_item(orc, small):
    rng, x, ids, ds, g, q, qc, qh = small
    qi = np.array([1, 4, 7, 2, 3001, 0], np.uint32)
    i, d, c = orc.search(ds, g, None, None, k=10, ef_search=100, query_items=qi)
    assert c.tolist() == [10, 10, 10, orc.NONE, 10, orc.NONE]  # unknown items: None (reader.rs:826)
    for r in (0, 1, 2, 4):
        assert int(qi[r]) not in i[r].tolist()                   # src/tests/reader.rs:114-127
        slot = (int(qi[r]) - 1) // 3
        mask = np.ones(len(ids), bool)
        mask[slot] = False
        truth = _exact(x, ids, x[slot:slot + 1], mask, 10)[0]
        assert len(truth & set(i[r].tolist())) >= 9
    # with candidates: linear scan keeps the item itself (reader.rs:831-833), the HNSW path drops it
    i, d, c = orc.search(ds, g, None, None, k=10, query_items=qi, candidates=ids[:50])
    assert i[0, 0] == 1 and d[0, 0] == 0.0
    j, e, c2 = orc.search(ds, g, None, None, k=10, query_items=qi, candidates=ids[:50], linear_below=0)
    assert np.array_equal(j[0, :9], i[0, 1:10]) and 1 not in j[0].tolist()
    # disjoint candidates: None for every query (reader.rs:822-824)
    c3 = orc.search(ds, g, None, None, k=3, query_items=qi, candidates=np.array([0], np.uint32))[2]
    assert (c3 == orc.NONE).all()


def test_unfiltered_path_unchanged(orc, small):
    """orc_search (the form pinned by KAT-8 and the recall checks) == orc_search_ex without options"""
    rng, x, ids, ds, g, q, qc, qh = small
    a = orc.search(ds, g, qc, qh, k=10, ef_search=50)
    truth = _exact(x, ids, q, np.ones(len(ids), bool), 10)
    assert np.mean([len(truth[r] & set(a[0][r].tolist())) / 10 for r in range(len(q))]) >= 0.95
