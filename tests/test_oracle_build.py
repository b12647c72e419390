"""Host-side / oracle behaviour beyond the golden vectors: invariants the reference's tests assert
(Reader::assert_validity reader.rs:905-948, all_items_are_reachable tests/reader.rs:82-98) and
consistency between the oracle's build modes."""
import numpy as np
import pytest

from conftest import draw_levels


def _validity(g, ds):
    """assert_validity (reader.rs:905-948): every link target exists; every item has >= 1 Links
    key; entry points exist; + each item has one record per layer 0..=level."""
    ids = set(ds.ids.tolist())
    d = g.as_dict()
    assert set(i for i, _ in d) == ids
    for (i, l), nb in d.items():
        assert set(nb) <= ids
        assert nb == sorted(set(nb))
    for e in g.entry_points:
        assert int(e) in ids
    for s, i in enumerate(ds.ids.tolist()):
        for l in range(int(ds.levels[s]) + 1):
            assert (i, l) in d
    assert g.max_level == int(ds.levels.max())


@pytest.mark.parametrize("metric,dim", [(0, 24), (1, 40), (3, 128)])
def test_modes_agree_and_are_valid(orc, metric, dim):
    rng = np.random.default_rng(dim)
    n = 600
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(metric, vecs, draw_levels(n, 8, 3))
    seq = orc.build(ds, M=8, M0=16, ef=32)
    _validity(seq, ds)
    b1 = orc.build(ds, M=8, M0=16, ef=32, batch_frac=0.5, batch_max=1)
    assert seq.as_dict() == b1.as_dict()  # batch size 1 == sequential insertion
    bt = orc.build(ds, M=8, M0=16, ef=32, batch_frac=0.1, batch_max=64)
    _validity(bt, ds)
    bt2 = orc.build(ds, M=8, M0=16, ef=32, batch_frac=0.1, batch_max=64, threads=4)
    assert bt.as_dict() == bt2.as_dict()  # the batched schedule is deterministic under threads
    th = orc.build(ds, M=8, M0=16, ef=32, threads=4)  # rayon-like: valid, not deterministic
    _validity(th, ds)


def test_all_items_are_reachable(orc):
    """tests/reader.rs:82-98: M = M0 = 6, nns(n).ef_search(n) from the zero vector finds all."""
    rng = np.random.default_rng(5)
    n, dim = 400, 32
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(orc.COSINE, vecs, draw_levels(n, 6, 9))
    for kw in ({}, {"batch_frac": 0.05, "batch_max": 32}):
        g = orc.build(ds, M=6, M0=6, ef=100, **kw)
        q = np.zeros((1, dim), np.float32)
        qc = orc.encode_vectors(orc.COSINE, q)
        qh = orc.make_headers(orc.COSINE, dim, qc)
        ids, dists, counts = orc.search(ds, g, qc, qh, k=n, ef_search=n)
        assert counts[0] == n and sorted(ids[0].tolist()) == list(range(n))


def test_recall_sequential_vs_batched(orc):
    """The batch-synchronous schedule (GPU semantics) keeps recall@10 within 0.5 % of the
    sequential reference semantics (north-star tolerance), small case."""
    rng = np.random.default_rng(11)
    n, dim, nq = 4000, 48, 200
    centres = rng.uniform(-1, 1, (32, dim))
    vecs = (centres[rng.integers(0, 32, n)] + 0.2 * rng.normal(size=(n, dim))).astype(np.float32)
    qs = (centres[rng.integers(0, 32, nq)] + 0.2 * rng.normal(size=(nq, dim))).astype(np.float32)
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, vecs, draw_levels(n, 16, 2))
    qc = orc.encode_vectors(orc.EUCLIDEAN, qs)
    qh = orc.make_headers(orc.EUCLIDEAN, dim, qc)
    d2 = ((qs[:, None, :].astype(np.float64) - vecs[None, :, :]) ** 2).sum(-1)
    truth = np.argsort(d2, axis=1)[:, :10]

    def recall(g):
        ids, _, cnt = orc.search(ds, g, qc, qh, k=10, ef_search=100)
        return sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(nq)) / (10 * nq)
    r_seq = recall(orc.build(ds, M=16, M0=32, ef=100))
    r_bat = recall(orc.build(ds, M=16, M0=32, ef=100, batch_frac=0.02, batch_max=16384))
    assert r_seq > 0.95
    assert abs(r_seq - r_bat) <= 0.005


def test_wave_and_x86_orders_build_equivalent_graphs(orc):
    """Different f32 summation orders (reference x86 vs GPU wave order) change distances by a few
    ulp; on integer-valued data (exact in f32) the graphs must be identical."""
    rng = np.random.default_rng(3)
    n, dim = 500, 64
    vecs = rng.integers(-8, 9, (n, dim)).astype(np.float32)
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, vecs, draw_levels(n, 8, 4))
    a = orc.build(ds, M=8, M0=16, ef=40, order=orc.ORDER_X86)
    b = orc.build(ds, M=8, M0=16, ef=40, order=orc.ORDER_WAVE)
    assert a.as_dict() == b.as_dict()


def test_empty_and_single(orc):
    ds = orc.Dataset(orc.COSINE, 4, np.zeros(0, np.uint32), np.zeros((0, 16), np.uint8),
                     np.zeros((0, 4), np.uint8), np.zeros(0, np.uint8))
    g = orc.build(ds)
    assert len(g.rec_item) == 0 and len(g.entry_points) == 0


def test_c1_readme_config_on_the_oracle(orc):
    """BASELINE config C1 (10k x 3-d Cosine, M=16/M0=32, efC=100 — the README example's shape,
    README.md:43-51, generator src/tests/mod.rs:133-136) through the CPU oracle: sequential reference
    semantics vs the GPU's batch schedule.  dim < 16 takes the reference's scalar summation path
    (simple.rs:19-47), where x86 and wave order coincide for 3 elements up to fma contraction; both
    builds are valid and reach the same recall@10 (+-0.5 %)."""
    rng = np.random.default_rng(42)
    n, nq = 10_000, 200
    v = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    v[:2] = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]  # README.md:45-46
    ds = orc.Dataset.from_f32(orc.COSINE, v, draw_levels(n, 16, 42))
    seq = orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_X86, threads=1)
    bat = orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=65536, threads=8)
    _validity(seq, ds)
    _validity(bat, ds)
    assert seq.n_evals_walk > 0 and bat.n_evals_walk > 0
    qs = rng.uniform(-1, 1, (nq, 3)).astype(np.float32)
    qc = orc.encode_vectors(orc.COSINE, qs)
    qh = orc.make_headers(orc.COSINE, 3, qc)
    vn = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
    qn = qs / np.linalg.norm(qs, axis=1, keepdims=True)
    sim = qn.astype(np.float64) @ vn.astype(np.float64).T
    kth = np.sort(sim, axis=1)[:, -10]

    def recall(g):  # ties are ubiquitous in 3-d: count a hit when the similarity reaches the 10th best
        ids, _, cnt = orc.search(ds, g, qc, qh, k=10, ef_search=100)
        hit = sum(int(np.sum(sim[i, ids[i, :cnt[i]]] >= kth[i] - 1e-9)) for i in range(nq))
        return hit / (10 * nq)
    r_seq, r_bat = recall(seq), recall(bat)
    assert r_seq > 0.95 and abs(r_seq - r_bat) <= 0.005
    # README.md:56-57: nns(1).ef_search(10) by [0, 1, 0] returns item 1 at distance 0
    q1 = np.array([[0.0, 1.0, 0.0]], np.float32)
    c1 = orc.encode_vectors(orc.COSINE, q1)
    ids, dists, cnt = orc.search(ds, seq, c1, orc.make_headers(orc.COSINE, 3, c1), k=1, ef_search=10)
    assert cnt[0] == 1 and dists[0, 0] == 0.0


@pytest.mark.parametrize("metric,dim,order", [(0, 24, "x86"), (3, 256, "wave"), (1, 128, "wave")])
def test_threaded_link_phase_of_a_batch_is_the_sequential_one(orc, metric, dim, order):
    """Batches of 256 members and more replay their add_link calls on several threads (round 5): every thread
    walks the whole sequence in batch order and performs the calls whose TARGET it owns, so per target the order is
    the sequential one.  The raw lists — insertion order, distances, duplicates and all — and the evaluation and
    link counters must be those of the one-thread build, for any thread count, with lists that overflow and
    re-prune (M0 = 12 on 5 000 clustered points) and with the AVX2 form of the wave-order reduction."""
    rng = np.random.default_rng(dim)
    n = 5000
    cent = rng.uniform(-1, 1, (12, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 12, n)] + 0.25 * rng.standard_normal((n, dim))).astype(np.float32)
    ds = orc.Dataset.from_f32(metric, vecs, draw_levels(n, 6, 3))
    kw = dict(M=6, M0=12, ef=32, order=orc.ORDER_X86 if order == "x86" else orc.ORDER_WAVE, batch_frac=1.0, batch_max=2048)
    one = orc.build(ds, threads=1, **kw)
    for t in (2, 5, 8):
        g = orc.build(ds, threads=t, **kw)
        assert np.array_equal(g.raw_offsets, one.raw_offsets) and np.array_equal(g.raw_nbrs, one.raw_nbrs)
        assert np.array_equal(g.raw_dists.view(np.uint32), one.raw_dists.view(np.uint32))
        assert (g.n_links_added, g.n_distance_evals, g.n_evals_walk) == (one.n_links_added, one.n_distance_evals, one.n_evals_walk)
