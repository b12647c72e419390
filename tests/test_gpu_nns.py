"""GPU parity: hny_builder_nns (QueryBuilder with candidates / linear scan / by_item,
/root/reference/src/reader.rs:60-262, 621-711, 809-896) against the restated Reader of the oracle on
the same graph — same ids, same distance bits — plus the reference's own property tests
(src/tests/reader.rs:42-78, 114-143)."""
import numpy as np
import pytest

from conftest import draw_levels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hny():
    import hannoy_amd
    hannoy_amd.load_library()
    return hannoy_amd


def _index(orc, hny, metric, n, dim, M, M0, ef, seed, ids=None):
    rng = np.random.default_rng(seed)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(metric, vecs, draw_levels(n, M, seed=seed), ids)
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    b = hny.Builder(items, M=M, M0=M0, ef_construction=ef, batch_frac=0.1, batch_max=256)
    b.run()
    g = b.finish()
    return rng, vecs, ds, b, g


def _queries(orc, metric, rng, nq, dim):
    qs = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    qc = orc.encode_vectors(metric, qs)
    return qs, qc, orc.make_headers(metric, dim, qc)


def _same(got, want):
    ids, dists, counts = got
    oids, odists, ocounts = want
    assert np.array_equal(counts, ocounts)
    for r in range(len(counts)):
        c = 0 if counts[r] == 0xFFFFFFFF else int(counts[r])
        assert np.array_equal(ids[r, :c], oids[r, :c]), r
        assert np.array_equal(dists[r, :c].view(np.uint32), odists[r, :c].view(np.uint32)), r


@pytest.mark.parametrize("metric,n,dim,M,M0", [(0, 3000, 96, 8, 16), (1, 2500, 40, 6, 12), (3, 3000, 256, 8, 16),
                                               (4, 2000, 192, 8, 16)])
def test_filtered_search_equals_oracle(orc, hny, metric, n, dim, M, M0):
    ids = (np.arange(n, dtype=np.uint32) * 3 + 1)
    rng, vecs, ds, b, g = _index(orc, hny, metric, n, dim, M, M0, 48, 11 + metric, ids)
    qs, qc, qh = _queries(orc, metric, rng, 300, dim)
    with b:
        for frac_kept, lb in ((0.5, 0), (0.05, 0), (0.004, 0), (0.1, 1000), (0.5, 5000)):
            cand = ids[rng.random(n) < frac_kept]
            cand = np.concatenate([cand, [0, 2, 10 ** 7]]).astype(np.uint32)  # unknown ids are ignored
            rng.shuffle(cand)
            for k, ef in ((10, 50), (3, 2), (1, 100)):
                got = b.nns(qc, qh, k=k, ef_search=ef, candidates=cand, linear_below=lb)
                want = orc.search(ds, g, qc, qh, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                                  candidates=cand, linear_below=lb)
                _same(got, want)
                cs = set(cand.tolist())
                assert all(int(v) in cs for r in range(len(qc)) for v in got[0][r, :got[2][r]])
        # nothing can match: empty Vec (reader.rs:652-654)
        got = b.nns(qc, qh, k=5, candidates=np.array([0, 2], np.uint32))
        assert not got[2].any()
        got = b.nns(qc, qh, k=5, candidates=np.zeros(0, np.uint32))
        assert not got[2].any()
        # ratio gate (reader.rs:637): 100 candidates of n is above ratio 0.01 -> HNSW path, below 1.0 -> linear
        cand = ids[:100]
        for ratio in (0.01, 1.0):
            got = b.nns(qc, qh, k=10, ef_search=30, candidates=cand, linear_below_ratio=ratio)
            want = orc.search(ds, g, qc, qh, k=10, ef_search=30, order=orc.ORDER_WAVE, threads=8,
                              candidates=cand, linear_below_ratio=ratio)
            _same(got, want)


@pytest.mark.parametrize("metric,n,dim,M,M0", [(0, 6000, 64, 8, 16), (3, 7000, 64, 8, 16), (1, 5000, 24, 4, 8)])
def test_filtered_and_by_item_search_with_result_sets_beyond_the_lds(orc, hny, metric, n, dim, M, M0):
    """max(ef_search, k) >= 4 096 with a candidates filter / by_item: `res` no longer fits k_nns_filtered's LDS, the
    same Visitor::visit runs with the search queue AND `res` as heaps in HBM (k_nns_heap) — ids, distances and
    counts of the restated Reader; sparse filters reach the exhaustive fallback (reader.rs:771-795 / 864-890) there,
    64-bit Hamming codes are all ties; the linear scan keeps its LDS ranking (min(k, candidates) hits)."""
    ids = (np.arange(n, dtype=np.uint32) * 3 + 1)
    rng, vecs, ds, b, g = _index(orc, hny, metric, n, dim, M, M0, 48, 31 + metric, ids)
    qs, qc, qh = _queries(orc, metric, rng, 40, dim)
    qi = np.concatenate([ids[rng.integers(0, n, 40)], [0, 2, 10 ** 7]]).astype(np.uint32)
    with b:
        for frac_kept in (0.9, 0.3, 0.01):
            cand = ids[rng.random(n) < frac_kept]
            cand = np.concatenate([cand, [0, 2, 10 ** 7]]).astype(np.uint32)
            for k, ef in ((10, 4500), (4200, 100), (5, 7000)):
                got = b.nns(qc, qh, k=k, ef_search=ef, candidates=cand, linear_below=0)
                want = orc.search(ds, g, qc, qh, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                                  candidates=cand, linear_below=0)
                _same(got, want)
                got = b.nns(k=k, ef_search=ef, query_items=qi, candidates=cand, linear_below=0)
                want = orc.search(ds, g, None, None, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                                  query_items=qi, candidates=cand, linear_below=0)
                _same(got, want)
        for k, ef in ((10, 4500), (4100, 10)):  # by_item without a filter
            got = b.nns(k=k, ef_search=ef, query_items=qi)
            want = orc.search(ds, g, None, None, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8, query_items=qi)
            _same(got, want)
        cand = ids[rng.random(n) < 0.1]  # linear scan: below the default threshold of 1 000 candidates
        got = b.nns(qc, qh, k=5000, ef_search=10, candidates=cand, linear_below=1000)
        want = orc.search(ds, g, qc, qh, k=5000, ef_search=10, order=orc.ORDER_WAVE, threads=8, candidates=cand,
                          linear_below=1000)
        _same(got, want)


@pytest.mark.parametrize("metric,n,dim,M,M0", [(0, 2000, 768, 16, 32), (1, 3000, 24, 4, 8), (3, 2500, 128, 8, 16)])
def test_by_item_equals_oracle(orc, hny, metric, n, dim, M, M0):
    ids = (np.arange(n, dtype=np.uint32) * 2 + 5)
    rng, vecs, ds, b, g = _index(orc, hny, metric, n, dim, M, M0, 64, 23 + metric, ids)
    qi = np.concatenate([ids[rng.integers(0, n, 400)], [0, 4, 6, 10 ** 6]]).astype(np.uint32)
    with b:
        for k, ef in ((10, 100), (5, 3), (1, 1)):
            got = b.nns(k=k, ef_search=ef, query_items=qi)
            want = orc.search(ds, g, None, None, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                              query_items=qi)
            _same(got, want)
            # search_by_item_returns_none_if_not_exists (src/tests/reader.rs:130-143)
            assert (got[2][-4:] == hny.NNS_NONE).all()
            # search_by_item_does_not_contain_item (src/tests/reader.rs:114-127)
            for r in range(400):
                assert got[2][r] == k and int(qi[r]) not in got[0][r, :k].tolist()
        for lb in (0, 1000):  # with candidates: HNSW path / linear scan (the item itself stays in, :831-833)
            cand = ids[rng.random(n) < 0.2]
            got = b.nns(k=10, ef_search=40, query_items=qi, candidates=cand, linear_below=lb)
            want = orc.search(ds, g, None, None, k=10, ef_search=40, order=orc.ORDER_WAVE, threads=8,
                              query_items=qi, candidates=cand, linear_below=lb)
            _same(got, want)
        got = b.nns(k=3, query_items=qi, candidates=np.array([1], np.uint32))  # disjoint: None (:822-824)
        assert (got[2] == hny.NNS_NONE).all()


def test_search_on_candidates_has_right_num(orc, hny):
    """src/tests/reader.rs:42-78: 1000 x 768 cosine; 10 (then 1) random candidates -> exactly those."""
    rng, vecs, ds, b, g = _index(orc, hny, 0, 1000, 768, 16, 32, 100, 3)
    qs, qc, qh = _queries(orc, 0, rng, 20, 768)
    with b:
        for r in range(20):
            cand = np.unique(rng.integers(0, 1000, 10).astype(np.uint32))
            ids, _, cnt = b.nns(qc[r:r + 1], qh[r:r + 1], k=10, candidates=cand)
            assert sorted(ids[0, :cnt[0]].tolist()) == cand.tolist()
            ids, _, cnt = b.nns(qc[r:r + 1], qh[r:r + 1], k=10, candidates=cand, linear_below=0)
            assert sorted(ids[0, :cnt[0]].tolist()) == cand.tolist()
            one = rng.integers(0, 1000, 1).astype(np.uint32)
            for lb in (1000, 0):
                ids, _, cnt = b.nns(qc[r:r + 1], qh[r:r + 1], k=1, candidates=one, linear_below=lb)
                assert cnt[0] == 1 and ids[0, 0] == one[0]


def test_queue_overflow_retries_with_full_heap(orc, hny):
    """a very selective filter on a larger index: the search queue outgrows the per-wave heap and the
    query is run again with room for every item — same answer as the oracle."""
    n, dim = 40000, 16
    rng, vecs, ds, b, g = _index(orc, hny, 1, n, dim, 8, 16, 32, 5)
    qs, qc, qh = _queries(orc, 1, rng, 64, dim)
    cand = rng.choice(n, 12, replace=False).astype(np.uint32)
    with b:
        got = b.nns(qc, qh, k=10, ef_search=100, candidates=cand, linear_below=0)
    want = orc.search(ds, g, qc, qh, k=10, ef_search=100, order=orc.ORDER_WAVE, threads=8, candidates=cand,
                      linear_below=0)
    _same(got, want)
    assert (got[2] == 10).all()


def test_loaded_graph_search_equals_oracle(orc, hny):
    """Reader::open -> hny_builder_load: searching the STORED records gives what the restated Reader
    gives on the same records (nothing may be re-linked while loading), with and without a filter."""
    from hannoy_amd.api import _StoredGraph
    rng = np.random.default_rng(9)
    n, dim = 3000, 48
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    db = hny.Database(None, hny.Metric.EUCLIDEAN)
    w = db.writer(dim, m=8, ef=48)
    w.add_items(range(n), vecs)
    w.build()
    for i in range(0, 300, 3):  # an incremental update on top, so disk + rebuilt lists mix
        w.del_item(i)
    w.add_items(range(n, n + 200), rng.uniform(-1, 1, (200, dim)).astype(np.float32))
    w.build()
    stored = _StoredGraph(db, 0)
    ids = db.metadata(0)["items"]
    items = db.item_set(0, ids, dim)
    ds = orc.Dataset(1, dim, ids, items.codes, items.headers, np.zeros(len(ids), np.uint8))
    qs, qc, qh = _queries(orc, 1, rng, 2500, dim)
    r = db.reader(0)
    r.assert_validity()
    got = r.nns(10).ef_search(40).by_vectors(qs)
    want = orc.search(ds, stored, qc, qh, k=10, ef_search=40, order=orc.ORDER_WAVE, threads=8)
    _same(got, want)
    got = r.by_vecs(qs, n=10, ef_search=40)
    _same(got, want)
    cand = ids[rng.random(len(ids)) < 0.3]
    got = r.nns(10).ef_search(40).candidates(cand).linear_below(0).by_vectors(qs)
    want = orc.search(ds, stored, qc, qh, k=10, ef_search=40, order=orc.ORDER_WAVE, threads=8, candidates=cand,
                      linear_below=0)
    _same(got, want)
    qi = ids[rng.integers(0, len(ids), 500)]
    got = r.nns(10).ef_search(40).by_items(qi)
    want = orc.search(ds, stored, None, None, k=10, ef_search=40, order=orc.ORDER_WAVE, threads=8, query_items=qi)
    _same(got, want)
    r.close()


def test_search_cancellation_through_the_c_abi(orc, hny):
    """by_vector_with_cancellation / by_item_with_cancellation (reader.rs:108-119, 167-186; probe at
    :333; src/tests/reader.rs:145-170): hny_query_opts.cancel is polled while the batch runs.  A
    closure that never fires changes nothing; one that fires at once starts no query (0 hits each,
    unknown items stay None, did_cancel set); one that fires in the middle of a large batch leaves
    the finished queries' results exactly as an uncancelled search returns them."""
    rng = np.random.default_rng(12)
    n, dim = 20000, 64
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    qs = rng.uniform(-1, 1, (30000, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, vecs, draw_levels(n, 16, seed=3))
    items = hny.ItemSet(hny.EUCLIDEAN, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    qc = orc.encode_vectors(orc.EUCLIDEAN, qs)
    qh = orc.make_headers(orc.EUCLIDEAN, dim, qc)
    with hny.Builder(items, M=16, M0=32, ef_construction=64) as b:
        b.run()
        b.finish()
        ref = b.nns(qc, qh, k=10, ef_search=64)
        assert not b.did_cancel
        same = b.nns(qc, qh, k=10, ef_search=64, cancel=lambda: False)
        assert not b.did_cancel
        for x, y in zip(ref, same):
            assert np.array_equal(x, y)
        ids, dists, counts = b.nns(qc, qh, k=10, ef_search=64, cancel=lambda: True)
        assert b.did_cancel and not counts.any()
        ids, dists, counts = b.nns(query_items=np.array([5, 2 ** 31, 7], np.uint32), k=5, cancel=lambda: True)
        assert b.did_cancel and counts.tolist() == [0, hny.NNS_NONE, 0]
        calls = []

        def later():  # fires on the third probe: the batch is under way by then
            calls.append(1)
            return len(calls) >= 3
        ids, dists, counts = b.nns(qc, qh, k=10, ef_search=64, cancel=later)
        done = counts > 0
        if b.did_cancel:  # (a very fast batch may finish before the third probe)
            assert np.array_equal(ids[done], ref[0][done]) and np.array_equal(counts[done], ref[2][done])
            assert np.array_equal(dists[done].view(np.uint32), ref[1][done].view(np.uint32))
        else:
            assert done.all()


@pytest.mark.parametrize("metric,n,dim", [(1, 6000, 16), (3, 4200, 128)])
def test_filtered_and_by_item_search_from_more_entry_points_than_the_lds_set_holds(orc, hny, metric, n, dim):
    """An all-level-0 index of more than 4 095 items: every item is an entry point (hnsw.rs:278-285) and every one
    of them is pushed to `res` before the first pop (reader.rs:755-761), whatever ef_search says — k_nns_filtered's LDS
    result set (4 096 entries) cannot hold them.  Round 4 accepted such builds (up to 8 192 entry points) but sent a
    filtered / by_item search with a small ef to the LDS kernel; it now takes the heaps in HBM (k_nns_heap).  Same
    ids, distances and counts as the restated Reader."""
    rng = np.random.default_rng(n)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(metric, vecs, np.zeros(n, np.uint8))
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    qs, qc, qh = _queries(orc, metric, rng, 24, dim)
    qi = np.concatenate([ds.ids[rng.integers(0, n, 24)], [10 ** 7]]).astype(np.uint32)
    with hny.Builder(items, M=8, M0=16, ef_construction=24, batch_frac=0.5, batch_max=512) as b:
        b.run()
        g = b.finish()
        assert len(g.entry_points) == n and g.max_level == 0
        for frac_kept in (0.5, 0.02):
            cand = ds.ids[rng.random(n) < frac_kept]
            for k, ef in ((10, 50), (3, 2)):
                got = b.nns(qc, qh, k=k, ef_search=ef, candidates=cand, linear_below=0)
                want = orc.search(ds, g, qc, qh, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                                  candidates=cand, linear_below=0)
                _same(got, want)
                got = b.nns(k=k, ef_search=ef, query_items=qi, candidates=cand, linear_below=0)
                want = orc.search(ds, g, None, None, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8,
                                  query_items=qi, candidates=cand, linear_below=0)
                _same(got, want)
        got = b.nns(k=5, ef_search=20, query_items=qi)
        want = orc.search(ds, g, None, None, k=5, ef_search=20, order=orc.ORDER_WAVE, threads=8, query_items=qi)
        _same(got, want)
