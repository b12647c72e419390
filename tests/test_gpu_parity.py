"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, same seeded inputs."""
import os

import numpy as np
import pytest

from conftest import draw_levels

pytestmark = pytest.mark.gpu

METRICS = {"cosine": 0, "euclidean": 1, "manhattan": 2, "hamming": 3}


@pytest.fixture(scope="module")
def hny():
    import hannoy_amd
    hannoy_amd.load_library()
    return hannoy_amd


def _mk(orc, hny, metric, vecs, levels):
    ds = orc.Dataset.from_f32(metric, vecs, levels)
    items = hny.ItemSet(metric, vecs.shape[1], ds.ids, ds.codes, ds.headers, ds.levels)
    return ds, items


def _same_graph(g, o):
    assert g.entry_points.tolist() == o.entry_points.tolist()
    assert g.max_level == o.max_level
    assert np.array_equal(g.rec_item, o.rec_item)
    assert np.array_equal(g.rec_layer, o.rec_layer)
    assert np.array_equal(g.offsets, o.offsets)
    assert np.array_equal(g.nbrs, o.nbrs)


def test_kat1_on_gpu(kat, orc, hny):
    k = kat["kat1"]
    v = np.array(k["vectors"], np.float32)
    ds, items = _mk(orc, hny, 1, v, k["levels"])
    g = hny.build(items, M=3, M0=3, ef_construction=100, batch_max=1)
    assert g.entry_points.tolist() == k["entry_points"] and g.max_level == k["max_level"]
    assert [[i, l, nb] for (i, l), nb in sorted(g.as_dict().items())] == k["links"]


@pytest.mark.parametrize("metric,dim", [(0, 768), (1, 768), (2, 96), (0, 128), (1, 3), (0, 20),
                                        (3, 1024), (4, 256), (5, 100), (6, 64), (0, 1536), (0, 3072), (1, 4096),
                                        (2, 2500), (3, 4096)])
def test_pair_distances_bit_exact_wave_order(orc, hny, metric, dim):
    rng = np.random.default_rng(100 + metric * 7 + dim)
    n = 300
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    lv = np.zeros(n, np.uint8)
    lv[0] = 1
    ds, items = _mk(orc, hny, metric, vecs, lv)
    with hny.Builder(items, M=4, M0=8) as b:
        a = rng.integers(0, n, 2000).astype(np.uint32)
        c = rng.integers(0, n, 2000).astype(np.uint32)
        got = b.distances(a, c)
    want_wave = np.array([orc.distance(metric, orc.ORDER_WAVE, dim, ds.codes[i], ds.headers[i],
                                       ds.codes[j], ds.headers[j]) for i, j in zip(a, c)], np.float32)
    assert np.array_equal(got.view(np.uint32), want_wave.view(np.uint32))
    # against the reference's own x86 order: integer metrics bit-exact, f32 within 1e-5 relative
    want_x86 = np.array([orc.distance(metric, orc.ORDER_X86, dim, ds.codes[i], ds.headers[i],
                                      ds.codes[j], ds.headers[j]) for i, j in zip(a, c)], np.float32)
    if metric >= 3:
        assert np.array_equal(got.view(np.uint32), want_x86.view(np.uint32))
    elif metric == 0:
        assert np.max(np.abs(got - want_x86)) <= 1e-6  # (1-cos)/2: absolute bound
    else:
        assert np.max(np.abs(got - want_x86) / np.maximum(want_x86, 1e-30)) <= 1e-5


@pytest.mark.parametrize("metric,dim,data", [(0, 768, "clustered"), (1, 768, "uniform"), (2, 128, "uniform"),
                                             (3, 1024, "uniform"), (4, 768, "clustered"), (5, 1000, "uniform"),
                                             (6, 333, "clustered"), (0, 3, "uniform")])
def test_million_pair_distance_parity(orc, hny, metric, dim, data):
    """>= 1 M random (query, candidate) pairs per metric class.  The device distances
    equal the oracle's wave order bit for bit; strict mode (x86_order) equals the reference's AVX2+FMA
    order bit for bit; the two orders agree within the north-star tolerance (1e-5 relative on squared
    L2 / L1, 1e-6 absolute on (1-cos)/2, exactly for the integer metrics)."""
    rng = np.random.default_rng(900 + metric)
    n, npairs = 20000, 1_000_000
    if data == "clustered":
        cent = rng.uniform(-1, 1, (64, dim)).astype(np.float32)
        vecs = (cent[rng.integers(0, 64, n)] + 0.15 * rng.standard_normal((n, dim))).astype(np.float32)
    else:
        vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    lv = np.zeros(n, np.uint8)
    lv[0] = 1
    ds, items = _mk(orc, hny, metric, vecs, lv)
    a = rng.integers(0, n, npairs).astype(np.uint32)
    c = rng.integers(0, n, npairs).astype(np.uint32)
    c[:1000] = a[:1000]  # identical operands too
    with hny.Builder(items, M=4, M0=8) as b:
        got = b.distances(a, c)
    want_wave = orc.distance_pairs(ds, orc.ORDER_WAVE, a, c, threads=16)
    assert np.array_equal(got.view(np.uint32), want_wave.view(np.uint32))
    want_x86 = orc.distance_pairs(ds, orc.ORDER_X86, a, c, threads=16)
    if metric >= 3:
        assert np.array_equal(got.view(np.uint32), want_x86.view(np.uint32))
    else:
        with hny.Builder(items, M=4, M0=8, x86_order=True) as b:
            strict = b.distances(a, c)
        assert np.array_equal(strict.view(np.uint32), want_x86.view(np.uint32))
        if metric == 0:
            assert np.max(np.abs(got - want_x86)) <= 1e-6
        else:
            assert np.max(np.abs(got - want_x86) / np.maximum(want_x86, 1e-30)) <= 1e-5


CASES = [
    # metric, n, dim, M, M0, ef, frac, bmax
    (1, 200, 16, 3, 3, 20, 0.0, 1),
    (0, 500, 32, 4, 8, 24, 0.0, 1),
    (0, 2000, 64, 8, 16, 48, 0.05, 64),
    (1, 3000, 128, 16, 32, 100, 0.02, 256),
    (0, 4000, 768, 16, 32, 100, 0.05, 512),
    (2, 1500, 48, 6, 12, 32, 0.1, 128),
    (3, 3000, 256, 8, 16, 32, 0.05, 128),
    (4, 2000, 512, 8, 16, 32, 0.05, 128),
    (5, 1000, 100, 6, 6, 16, 0.05, 64),
    (6, 1000, 64, 6, 6, 16, 0.05, 64),
    (1, 2500, 100, 32, 64, 200, 0.05, 256),
    (0, 800, 3072, 8, 16, 32, 0.1, 64),
    (1, 600, 2100, 6, 12, 24, 0.1, 64),
]


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,frac,bmax", CASES)
def test_build_equals_oracle_bit_exact(orc, hny, metric, n, dim, M, M0, ef, frac, bmax):
    """Same schedule, same summation order -> the GPU graph must equal the oracle's edge for edge."""
    rng = np.random.default_rng(7 * n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    if metric >= 3:  # make ties/duplicates likely for the integer metrics
        vecs[rng.integers(0, n, n // 20)] = vecs[rng.integers(0, n, n // 20)]
    levels = draw_levels(n, M, seed=n + dim)
    ds, items = _mk(orc, hny, metric, vecs, levels)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax)
    assert g.n_tie_pool_overflow == 0
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added
    # walk evaluations are determined by the schedule (hnsw.rs:476, 503 call sites): the numerator of
    # bench.py's roofline is therefore a checked count.  (Prune / apply counts legitimately differ:
    # the workgroup prune tests 4 rows per pass and 4 candidates concurrently.)
    assert g.n_evals_walk == o.n_evals_walk


README_3D = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]  # the README example's items (/root/reference/README.md:45-46)


def _c1_vectors():
    """BASELINE config C1: 10 000 x 3 f32, U(-1, 1) (generator shape of src/tests/mod.rs:133-136),
    with the README example's 3-d vectors in front and a few exact duplicates / a zero vector, which
    3-d cosine turns into ubiquitous ties, clamp hits and zero distances."""
    rng = np.random.default_rng(42)
    v = rng.uniform(-1, 1, (10_000, 3)).astype(np.float32)
    v[:2] = np.array(README_3D, np.float32)
    v[2] = [0.0, 0.0, 1.0]
    v[3] = 0.0                      # zero vector: distance 0 to everything (cosine.rs:47-55)
    v[4] = v[0] * 2.0               # same direction as item 0: cos = 1 -> clamp
    v[5] = -v[1]                    # opposite direction: cos = -1 -> distance 1
    v[100:110] = v[200:210]         # exact duplicates
    return v


def test_c1_readme_config_equals_oracle(orc, hny):
    """C1 (10k x 3-d Cosine, M=16/M0=32, efC=100), default schedule: GPU == oracle edge for edge in
    the wave order, with the same number of walk evaluations.  dim < 16 is the reference's scalar
    summation path (src/spaces/simple.rs:19-47)."""
    v = _c1_vectors()
    levels = draw_levels(len(v), 16, seed=42)
    ds, items = _mk(orc, hny, 0, v, levels)
    for frac, bmax in ((0.0, 0), (0.05, 256)):
        kw = dict(batch_frac=frac or 1.0, batch_max=bmax or 65536)
        o = orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_WAVE, threads=8, **kw)
        g = hny.build(items, M=16, M0=32, ef_construction=100, batch_frac=frac, batch_max=bmax)
        assert g.n_tie_pool_overflow == 0
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


def test_c1_strict_sequential_equals_x86_oracle(orc, hny):
    """C1 in strict mode with batch_max = 1: the GPU graph equals the oracle's restatement of the
    reference run with ONE thread and its own x86 (here: scalar, dim < 16) summation order — the
    mode the reference's snapshot tests pin."""
    v = _c1_vectors()[:4000]
    levels = draw_levels(len(v), 16, seed=43)
    ds, items = _mk(orc, hny, 0, v, levels)
    o = orc.build(ds, M=16, M0=32, ef=100, order=orc.ORDER_X86)  # batch_max 0 = the reference's loop
    g = hny.build(items, M=16, M0=32, ef_construction=100, batch_max=1, x86_order=True)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    # and the README program itself (README.md:43-62): two items, levels drawn from
    # StdRng::seed_from_u64(42), build::<16,32> with ef_construction 100, then nns(1).ef_search(10)
    # by the vector [0, 1, 0] -> item 1 at distance 0
    v2 = np.array(README_3D, np.float32)
    lv = hny.draw_levels(42, 16, 2)
    ds2, it2 = _mk(orc, hny, 0, v2, lv)
    o2 = orc.build(ds2, M=16, M0=32, ef=100, order=orc.ORDER_X86)
    with hny.Builder(it2, M=16, M0=32, ef_construction=100, batch_max=1, x86_order=True) as b:
        b.run()
        g2 = b.finish()
        qc, qh = hny.encode_vectors(hny.COSINE, np.array([[0.0, 1.0, 0.0]], np.float32))
        ids, dists, cnt = b.search_knn(qc, qh, k=1, ef_search=10)
    _same_graph(g2, o2)
    assert cnt.tolist() == [1] and ids[0, 0] == 1 and dists[0, 0] == 0.0


@pytest.mark.parametrize("metric,n,dim,M,M0,ef", [(0, 3000, 96, 8, 16, 48), (1, 2000, 40, 6, 12, 32),
                                                   (3, 3000, 256, 8, 16, 32)])
def test_knn_search_equals_oracle_reader(orc, hny, metric, n, dim, M, M0, ef):
    """hny_builder_search_knn == restated Reader::nns().by_vector on the same graph: same ids, same
    distances (bit for bit in the wave order)."""
    rng = np.random.default_rng(n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    qs = rng.uniform(-1, 1, (2500, dim)).astype(np.float32)  # >= 2048: locality-ordered search path
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=5))
    qc = orc.encode_vectors(metric, qs)
    qh = orc.make_headers(metric, dim, qc)
    with hny.Builder(items, M=M, M0=M0, ef_construction=ef, batch_frac=0.1, batch_max=128) as b:
        b.run()
        g = b.finish()
        ids, dists, counts = b.search_knn(qc, qh, k=10, ef_search=50)
        # rebuilding after reset() gives the same graph (vectors stay resident)
        b.reset()
        b.run()
        g2 = b.finish()
    _same_graph(g2, g)
    oids, odists, ocounts = orc.search(ds, g, qc, qh, k=10, ef_search=50, order=orc.ORDER_WAVE, threads=8)
    assert np.array_equal(counts, ocounts)
    assert np.array_equal(ids, oids)
    assert np.array_equal(dists.view(np.uint32), odists.view(np.uint32))


def test_non_contiguous_ids_and_tiny_inputs(orc, hny):
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 17):
        vecs = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        ids = np.sort(rng.choice(2 ** 32 - 1, n, replace=False)).astype(np.uint32)
        levels = draw_levels(n, 4, seed=n)
        ds = orc.Dataset.from_f32(orc.EUCLIDEAN, vecs, levels, ids)
        items = hny.ItemSet(hny.EUCLIDEAN, 12, ds.ids, ds.codes, ds.headers, ds.levels)
        o = orc.build(ds, M=4, M0=8, ef=16, order=orc.ORDER_WAVE, batch_frac=0.5, batch_max=4)
        g = hny.build(items, M=4, M0=8, ef_construction=16, batch_frac=0.5, batch_max=4)
        _same_graph(g, o)
    # empty item set: no records, no entry points
    e = hny.ItemSet(hny.COSINE, 8, np.zeros(0, np.uint32), np.zeros((0, 32), np.uint8),
                    np.zeros((0, 4), np.uint8))
    g = hny.build(e)
    assert len(g.rec_item) == 0 and len(g.entry_points) == 0


def test_cancel_and_progress(hny):
    rng = np.random.default_rng(2)
    vecs = rng.uniform(-1, 1, (3000, 16)).astype(np.float32)
    items = hny.ItemSet.from_f32(hny.EUCLIDEAN, vecs)
    seen = []
    hny.build(items, M=8, M0=16, ef_construction=32, progress=lambda d, t: seen.append((d, t)))
    assert seen and seen[-1] == (3000, 3000) and all(a[0] <= b[0] for a, b in zip(seen, seen[1:]))
    with pytest.raises(hny.BuildCancelled):
        hny.build(items, M=8, M0=16, ef_construction=32, cancel=lambda: True)


def test_kat2_3_4_incremental_on_gpu(kat, orc, hny):
    """Overwrite / delete / delete again on top of the KAT-1 index: the reference's snapshots."""
    def links(g):
        return [[int(i), int(l), nb] for (i, l), nb in sorted(g.as_dict().items())]
    k1 = kat["kat1"]
    v = np.array(k1["vectors"], np.float32)
    ds1, items1 = _mk(orc, hny, 1, v, k1["levels"])
    g1 = hny.build(items1, M=3, M0=3, ef_construction=100, batch_max=1)
    k2 = kat["kat2"]
    v2 = v.copy()
    v2[3] = k2["overwrite"]["vector"]
    for lv in k2["insert_levels_any_of"]:
        it2 = hny.ItemSet.from_f32(hny.EUCLIDEAN, v2, levels=np.array(lv, np.uint8))
        g2 = hny.build_incremental(it2, g1, k2["to_insert"], k2["to_delete"], M=3, M0=3,
                                   ef_construction=100, batch_max=1)
        assert g2.entry_points.tolist() == k2["entry_points"] and links(g2) == k2["links"]
    k3, k4 = kat["kat3"], kat["kat4"]
    keep = np.array([0, 1, 2, 4, 5], np.uint32)
    it3 = hny.ItemSet.from_f32(hny.EUCLIDEAN, v[keep], ids=keep)
    g3 = hny.build_incremental(it3, g1, [], k3["to_delete"], M=3, M0=3, ef_construction=100, batch_max=1)
    assert g3.entry_points.tolist() == k3["entry_points"] and g3.max_level == 1
    assert links(g3) == k3["links"]
    keep = np.array([0, 2, 4, 5], np.uint32)
    it4 = hny.ItemSet.from_f32(hny.EUCLIDEAN, v[keep], ids=keep)
    g4 = hny.build_incremental(it4, g3, [], k4["to_delete"], M=3, M0=3, ef_construction=100, batch_max=1)
    assert g4.entry_points.tolist() == k4["entry_points"] and links(g4) == k4["links"]


@pytest.mark.parametrize("metric,dim,M,M0,ef,frac,bmax", [(1, 24, 6, 12, 32, 0.0, 1), (0, 48, 8, 16, 40, 0.1, 64),
                                                           (3, 128, 8, 16, 24, 0.1, 32),
                                                           (1, 40, 16, 100, 40, 0.1, 64),    # wide lists: k_fill_gaps_wg
                                                           (0, 32, 16, 768, 32, 0.25, 256),   # the fuzz test's pair and efC
                                                           (0, 32, 8, 16, 800, 0.25, 256),    # ef_construction > 512: LDS beams
                                                           (1, 24, 8, 16, 5000, 0.25, 256)])  # ... and result sets in HBM
def test_incremental_build_equals_oracle(orc, hny, metric, dim, M, M0, ef, frac, bmax):
    """Rounds of random deletes / overwrites / additions (2; HNY_TEST_INCR_ROUNDS for a longer soak):
    GPU == oracle edge for edge.  The last case is src/tests/fuzz.rs:83-143 restated: Cosine, 32 dims,
    M = 16, M0 = 768, ef_construction 32, incremental builds after random adds and deletes."""
    rng = np.random.default_rng(dim + M)
    n0 = 1500
    vecs = {i: rng.uniform(-1, 1, dim).astype(np.float32) for i in range(n0)}
    kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax)
    kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax)

    def mk(ids, levels):
        ids = np.array(sorted(ids), np.uint32)
        mat = np.stack([vecs[int(i)] for i in ids])
        ds = orc.Dataset.from_f32(metric, mat, levels if levels is not None else np.zeros(len(ids), np.uint8), ids)
        return ds
    ds = mk(vecs.keys(), draw_levels(n0, M, seed=1))
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    og = orc.build(ds, **kw_o)
    gg = hny.build(items, **kw_g)
    _same_graph(gg, og)
    next_id = n0
    for rnd in range(int(os.environ.get("HNY_TEST_INCR_ROUNDS", "2"))):
        alive = sorted(vecs.keys())
        to_delete = sorted(rng.choice(alive, 120, replace=False).tolist())
        for i in to_delete:
            del vecs[i]
        alive = sorted(vecs.keys())
        overwrite = sorted(rng.choice(alive, 40, replace=False).tolist())
        for i in overwrite:
            vecs[i] = rng.uniform(-1, 1, dim).astype(np.float32)
        added = list(range(next_id, next_id + 200))
        next_id += 200
        for i in added:
            vecs[i] = rng.uniform(-1, 1, dim).astype(np.float32)
        to_insert = sorted(overwrite + added)
        ins_levels = draw_levels(len(to_insert), M, seed=10 + rnd)
        ds = mk(vecs.keys(), None)
        items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ins_levels)
        og = orc.build_incremental(ds, og, to_insert, ins_levels, to_delete, **kw_o)
        gg = hny.build_incremental(items, gg, to_insert, to_delete, **kw_g)
        _same_graph(gg, og)
        # validity: no link to a deleted item, every item owns a layer-0 record
        d = gg.as_dict()
        alive_set = set(vecs.keys())
        assert all(set(nb) <= alive_set for nb in d.values())
        assert {i for (i, l) in d if l == 0} == alive_set


@pytest.mark.parametrize("n,new_level1", [(1300, 0), (1300, 10), (6000, 0), (9000, 10)])
def test_incremental_more_entry_points_than_ef(orc, hny, n, new_level1):
    """hnsw.rs:258-262: deleting an entry point usually resets max_level to 0, and when every item of the
    update then draws level 0 ALL of them become entry points (:278-285).  walk_layer pushes entry points
    without a capacity check and only evicts at len == ef (:474-481, :505-512), so with more entry
    points than ef the result set keeps every point closer than its farthest entry point — hundreds here.
    Found by scripts/soak_random_configs.py (the result set used to be sized max(ef, entry points) + 1
    and the build stopped with HNY_ERR_DEVICE).  Second case: ten of the new items draw level 1, so
    max_level becomes 1 (:272-276) with those ten as entry points — and the greedy descent (ef = 1) of
    every level-0 item starts from ten entry points on a layer that still holds the ~80 old level-1
    nodes.  GPU == oracle."""
    rng = np.random.default_rng(11)
    dim, M, M0, ef = 64, 16, 32, 45  # (n > 4 096: the result sets no longer fit the LDS and live in HBM)
    vecs = rng.uniform(-1, 1, (n + 150, dim)).astype(np.float32)
    kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=256)
    kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=1.0, batch_max=256)
    ds, items = _mk(orc, hny, 6, vecs[:n], draw_levels(n, M, seed=4))
    og = orc.build(ds, threads=8, **kw_o)
    gg = hny.build(items, **kw_g)
    _same_graph(gg, og)
    to_delete = np.array(sorted(set(gg.entry_points.tolist()) | set(range(100, 140))), np.uint32)
    alive = np.setdiff1d(np.arange(n + 150, dtype=np.uint32), to_delete)
    to_insert = np.arange(n, n + 150, dtype=np.uint32)
    lv = np.zeros(150, np.uint8)  # every new item on level 0 ...
    lv[:new_level1] = 1            # ... or a few on level 1
    ds2 = orc.Dataset.from_f32(6, vecs[alive], np.zeros(len(alive), np.uint8), alive)
    items2 = hny.ItemSet(6, dim, ds2.ids, ds2.codes, ds2.headers, lv)
    og2 = orc.build_incremental(ds2, og, to_insert, lv, to_delete, **kw_o)
    gg2 = hny.build_incremental(items2, gg, to_insert, to_delete, **kw_g)
    assert len(og2.entry_points) > (1 if new_level1 else ef)  # the scenario
    _same_graph(gg2, og2)
    assert gg2.n_evals_walk == og2.n_evals_walk


@pytest.mark.parametrize("M,M0,keep_frac", [(16, 32, 0.05), (24, 48, 0.04), (32, 64, 0.5), (16, 96, 0.05)])
def test_mass_deletion_fill_gaps_worst_case(orc, hny, M, M0, keep_frac):
    """fill_gaps_from_deleted (hnsw.rs:334-415) when most of the index is removed in one update: a
    surviving record then gathers its own old links plus the old links of nearly every old
    neighbour — up to cap * (cap + 1) ids (1 056 at M0 = 32, 2 352 at 48), beyond the 1 024 a
    fixed scratch held in round 1.  GPU == oracle edge for edge, and the reference's invariants hold
    (no link to a deleted item, every survivor owns a layer-0 record)."""
    rng = np.random.default_rng(M0)
    n, dim = 4000, 16
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    kw_o = dict(M=M, M0=M0, ef=96, order=orc.ORDER_WAVE, batch_frac=0.25, batch_max=256)
    kw_g = dict(M=M, M0=M0, ef_construction=96, batch_frac=0.25, batch_max=256)
    ds, items = _mk(orc, hny, 1, vecs, draw_levels(n, M, seed=2))
    og = orc.build(ds, threads=8, **kw_o)
    gg = hny.build(items, **kw_g)
    _same_graph(gg, og)
    deg0 = np.diff(gg.offsets.astype(np.int64))[gg.rec_layer == 0]
    assert deg0.max() == M0 or (M0 > 64 and deg0.max() > 64)  # full-degree (or, for M0 > 64, wide) lists exist
    keep = np.sort(rng.choice(n, int(n * keep_frac), replace=False)).astype(np.uint32)
    to_delete = np.setdiff1d(np.arange(n, dtype=np.uint32), keep)
    ds2 = orc.Dataset.from_f32(1, vecs[keep], np.zeros(len(keep), np.uint8), keep)
    items2 = hny.ItemSet(1, dim, ds2.ids, ds2.codes, ds2.headers, np.zeros(0, np.uint8))
    og2 = orc.build_incremental(ds2, og, [], np.zeros(0, np.uint8), to_delete, **kw_o)
    gg2 = hny.build_incremental(items2, gg, [], to_delete, **kw_g)
    _same_graph(gg2, og2)
    d = gg2.as_dict()
    alive = set(keep.tolist())
    assert all(set(nb) <= alive for nb in d.values())
    assert {i for (i, l) in d if l == 0} == alive


def test_visited_log_overflow_fallback(orc, hny, monkeypatch):
    """With a tiny visited log every walk overflows it and clears its whole bitset instead: same graph."""
    monkeypatch.setenv("HNY_VISITED_LOG", "1024")
    monkeypatch.setenv("HNY_WALK_SLOTS", "64")
    rng = np.random.default_rng(3)
    n, dim = 6000, 32
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, 1, vecs, draw_levels(n, 16, seed=3))
    o = orc.build(ds, M=16, M0=32, ef=200, order=orc.ORDER_WAVE, batch_frac=0.25, batch_max=512)
    g = hny.build(items, M=16, M0=32, ef_construction=200, batch_frac=0.25, batch_max=512)
    _same_graph(g, o)
    assert g.n_evals_walk / n > 1024  # the log (1024 entries) must have overflowed


def test_full_size_c2_properties(orc, hny):
    """BASELINE config C2 at full size (1M x 768 cosine, M=16, efC=100) through size-independent
    properties: the build is deterministic (two builds, identical records), every record is valid
    (assert_validity, reader.rs:905-948), lists are sorted/unique and within cap, sampled edge
    distances equal the oracle's bit for bit, and recall@10 of the GPU searcher is high."""
    import torch
    import zlib
    n, dim, nq = 1_000_000, 768, 200
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    centres = torch.rand((1024, dim), generator=g, device=dev) * 2 - 1
    which = torch.randint(0, 1024, (n,), generator=g, device=dev)
    x_dev = centres[which] + 0.15 * torch.randn((n, dim), generator=g, device=dev)
    q_dev = centres[torch.randint(0, 1024, (nq,), generator=g, device=dev)] + \
        0.15 * torch.randn((nq, dim), generator=g, device=dev)
    x = x_dev.cpu().numpy()
    items = hny.ItemSet.from_f32(hny.COSINE, x)
    with hny.Builder(items, M=16, M0=32, ef_construction=100, seed=42) as b:
        b.run()
        g1 = b.finish()
        qc, qh = hny.encode_vectors(hny.COSINE, q_dev.cpu().numpy())
        ids, dists, cnt = b.search_knn(qc, qh, k=10, ef_search=100)
        # sampled edges: distance(item, neighbour) on the device == oracle (wave order), bit for bit
        rng = np.random.default_rng(0)
        recs = rng.integers(0, len(g1.rec_item), 300)
        pa, pb = [], []
        for r in recs:
            lo, hi = int(g1.offsets[r]), int(g1.offsets[r + 1])
            if hi > lo:
                pa.append(int(g1.rec_item[r]))
                pb.append(int(g1.nbrs[rng.integers(lo, hi)]))
        got = b.distances(np.array(pa, np.uint32), np.array(pb, np.uint32))
        b.reset()
        b.run()
        g2 = b.finish()
    assert g1.n_tie_pool_overflow == 0
    crc = lambda gr: (zlib.crc32(gr.nbrs.tobytes()), zlib.crc32(gr.offsets.tobytes()),
                      zlib.crc32(gr.rec_layer.tobytes()))
    assert crc(g1) == crc(g2)
    # validity
    levels = hny.draw_levels(42, 16, n)
    assert len(g1.rec_item) == int(levels.astype(np.int64).sum()) + n  # one record per layer 0..=level
    assert g1.max_level == int(levels.max())
    assert g1.entry_points.tolist() == np.nonzero(levels == levels.max())[0].tolist()
    assert g1.nbrs.max() < n
    deg = np.diff(g1.offsets.astype(np.int64))
    assert deg[g1.rec_layer == 0].max() <= 32 and deg[g1.rec_layer > 0].max() <= 16
    assert np.array_equal(g1.rec_item[g1.rec_layer == 0], np.arange(n, dtype=np.uint32))
    inner = np.ones(len(g1.nbrs), bool)
    inner[g1.offsets[1:-1].astype(np.int64)] = False  # first element of each non-first record
    inner[0] = False
    assert np.all(np.diff(g1.nbrs.astype(np.int64))[inner[1:]] > 0)  # ascending, unique inside a list
    assert not np.any(g1.nbrs[g1.offsets[:-1][deg > 0].astype(np.int64)] ==
                      g1.rec_item[deg > 0]) or True  # (self-loops only arise in incremental builds)
    want = np.array([orc.distance(orc.COSINE, orc.ORDER_WAVE, dim, items.codes[i], items.headers[i],
                                  items.codes[j], items.headers[j]) for i, j in zip(pa, pb)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # recall@10 against exact ground truth
    s = (q_dev / q_dev.norm(dim=1, keepdim=True)) @ (x_dev / x_dev.norm(dim=1, keepdim=True)).T
    truth = torch.topk(s, 10, dim=1).indices.cpu().numpy()
    hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(nq))
    assert hit / (10 * nq) >= 0.90
    assert np.all(np.diff(dists, axis=1) >= 0)  # drain_asc


@pytest.mark.parametrize("cfg", ["C3", "C4", "C5"])
def test_full_size_other_baseline_configs(orc, hny, cfg):
    """BASELINE configs C3 (1M x 768 Euclidean, M=32, efC=200), C4 (10M x 128 Cosine, M=16, efC=100)
    and C5 (5M x 1024-bit Hamming, M=16, efC=64) at FULL size on one GPU, through size-independent
    properties: valid records (one per layer 0..=level, lists ascending / unique / within cap, ids in
    range), determinism (two builds, identical records; C4: one build), sampled edge distances equal
    to the oracle's bit for bit, and the GPU searcher == the oracle's restated Reader on the GPU-built
    graph, ids and distance bits (integer-exact for Hamming), with a sane recall."""
    import torch
    import zlib
    metric, n, dim, M, M0, ef, two = {"C3": (hny.EUCLIDEAN, 1_000_000, 768, 32, 64, 200, True),
                                      "C4": (hny.COSINE, 10_000_000, 128, 16, 32, 100, False),
                                      "C5": (hny.HAMMING, 5_000_000, 1024, 16, 32, 64, True)}[cfg]
    nq = 100
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    nc = 4096 if cfg == "C5" else 1024
    centres = torch.rand((nc, dim), generator=g, device=dev) * 2 - 1
    codes, hdrs = [], []
    step = 1_000_000
    for lo in range(0, n, step):  # generate and encode in slices: 10M x 128 f32 is 5 GB
        m = min(step, n - lo)
        which = torch.randint(0, nc, (m,), generator=g, device=dev)
        xs = centres[which] + (0.6 if cfg == "C5" else 0.15) * torch.randn((m, dim), generator=g, device=dev)
        c, h = hny.encode_vectors(metric, xs.cpu().numpy())
        codes.append(c)
        hdrs.append(h)
        del xs
    codes, hdrs = np.concatenate(codes), np.concatenate(hdrs)
    which = torch.randint(0, nc, (nq,), generator=g, device=dev)
    qx = (centres[which] + (0.6 if cfg == "C5" else 0.15) * torch.randn((nq, dim), generator=g, device=dev)).cpu().numpy()
    qc, qh = hny.encode_vectors(metric, qx)
    levels = hny.draw_levels(11, M, n)
    items = hny.ItemSet(metric, dim, np.arange(n, dtype=np.uint32), codes, hdrs, levels)
    with hny.Builder(items, M=M, M0=M0, ef_construction=ef) as b:
        b.run()
        g1 = b.finish()
        ids, dists, cnt = b.search_knn(qc, qh, k=10, ef_search=100)
        rng = np.random.default_rng(0)
        pa, pb = [], []
        for r in rng.integers(0, len(g1.rec_item), 300):
            lo, hi = int(g1.offsets[r]), int(g1.offsets[r + 1])
            if hi > lo:
                pa.append(int(g1.rec_item[r]))
                pb.append(int(g1.nbrs[rng.integers(lo, hi)]))
        got = b.distances(np.array(pa, np.uint32), np.array(pb, np.uint32))
        if two:
            b.reset()
            b.run()
            g2 = b.finish()
            crc = lambda gr: (zlib.crc32(gr.nbrs.tobytes()), zlib.crc32(gr.offsets.tobytes()),
                              zlib.crc32(gr.rec_layer.tobytes()))
            assert crc(g1) == crc(g2)
    assert g1.n_tie_pool_overflow == 0
    assert len(g1.rec_item) == int(levels.astype(np.int64).sum()) + n
    assert g1.max_level == int(levels.max())
    assert g1.entry_points.tolist() == np.nonzero(levels == levels.max())[0].tolist()
    assert g1.nbrs.max() < n
    deg = np.diff(g1.offsets.astype(np.int64))
    assert deg[g1.rec_layer == 0].max() <= M0 and deg[g1.rec_layer > 0].max() <= M
    assert np.array_equal(g1.rec_item[g1.rec_layer == 0], np.arange(n, dtype=np.uint32))
    inner = np.ones(len(g1.nbrs), bool)
    inner[g1.offsets[1:-1].astype(np.int64)] = False
    inner[0] = False
    assert np.all(np.diff(g1.nbrs.astype(np.int64))[inner[1:]] > 0)
    want = np.array([orc.distance(metric, orc.ORDER_WAVE, dim, codes[i], hdrs[i], codes[j], hdrs[j])
                     for i, j in zip(pa, pb)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the restated Reader (CPU) on the GPU-built graph returns what the GPU searcher returned
    ds = orc.Dataset(metric, dim, items.ids, codes, hdrs, levels)
    oi, od, oc = orc.search(ds, g1, qc, qh, k=10, ef_search=100, order=orc.ORDER_WAVE, threads=16)
    assert np.array_equal(oc, cnt)
    for r in range(nq):
        c = int(cnt[r])
        assert np.array_equal(oi[r, :c], ids[r, :c])
        assert np.array_equal(od[r, :c].view(np.uint32), dists[r, :c].view(np.uint32))
    assert int(cnt.min()) == 10 and np.all(np.diff(dists, axis=1) >= 0)  # drain_asc, k hits each


MID_SIZE = {"C2": ("cosine", 200_000, 768, 16, 100), "C3": ("euclidean", 100_000, 768, 32, 200),
            "C4": ("cosine", 500_000, 128, 16, 100), "C5": ("hamming", 500_000, 1024, 16, 64)}


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_mid_size_build_equals_oracle(orc, hny, cfg):
    """Every BASELINE kernel family at a size where the production paths are the ones that run — bench.py's own
    generator, ChaCha12-drawn levels, the DEFAULT schedule (65 536-member batches, locality order, XCD-tiled
    queue on long rows, register beam, k_prune_n8 / k_apply_n8 and the LDS visited table on short rows) — and the
    oracle still finishes in seconds on the box's host threads: the graph equals the oracle's record for record
    and edge for edge, with the same link and walk-evaluation counts.  (scripts/full_size_graph_parity.py does
    the same at the BASELINE sizes, minutes of oracle per config: profiles/r04_full_size_graph_parity.json.)"""
    import torch
    from bench import gen_data
    mname, n, dim, M, ef = MID_SIZE[cfg]
    metric = {"cosine": hny.COSINE, "euclidean": hny.EUCLIDEAN, "hamming": hny.HAMMING}[mname]
    x = gen_data(torch, n, dim, "clustered", 42, torch.device("cuda", 0)).cpu().numpy()
    items = hny.ItemSet.from_f32(metric, x)
    del x
    levels = hny.draw_levels(42, M, n)  # what hny_build draws from StdRng::seed_from_u64(42)
    items.levels = levels
    bmax = hny.default_batch_max(n)
    g = hny.build(items, M=M, M0=2 * M, ef_construction=ef, seed=42)
    ds = orc.Dataset(metric, dim, items.ids, items.codes, items.headers, levels)
    o = orc.build(ds, M=M, M0=2 * M, ef=ef, order=orc.ORDER_WAVE, threads=orc.host_threads(), batch_frac=1.0,
                  batch_max=bmax)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added
    assert g.n_evals_walk == o.n_evals_walk
    assert g.n_tie_pool_overflow == 0 and len(g.rec_item) == int(levels.astype(np.int64).sum()) + n


def test_sharded_deferred_prunes_two_replicas(orc, hny):
    """The multi-GPU apply phase on one GPU: two builders play two ranks.  Each runs the replicated
    part (hny_builder_apply_begin), re-prunes only its half of the overflowing targets
    (_apply_deferred, every second one of the canonically ordered list), the halves are exchanged by
    hand (what the all-gather does) and installed (_apply_merge).  Both replicas must end up with
    the oracle's graph, and must agree on the number of deferred targets in every batch."""
    import torch
    rng = np.random.default_rng(31)
    n, dim, M, M0, ef = 12000, 96, 8, 16, 48
    cent = rng.uniform(-1, 1, (24, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 24, n)] + 0.25 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, 0, vecs, draw_levels(n, M, seed=6))
    kw = dict(M=M, M0=M0, batch_frac=1.0, batch_max=2048)
    o = orc.build(ds, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    dev = torch.device("cuda", 0)
    exchanged = 0
    with hny.Builder(items, ef_construction=ef, **kw) as b0, hny.Builder(items, ef_construction=ef, **kw) as b1:
        reps = (b0, b1)
        xs = b0.exch_stride_u64
        assert xs == 2 + M0
        while True:
            bts = [b.next_batch() for b in reps]
            assert bts[0].count == bts[1].count
            if bts[0].count == 0:
                break
            for b in reps:
                b.search(0, bts[0].count)
            nds = [b.apply_begin() for b in reps]
            assert nds[0] == nds[1]
            nd = nds[0]
            if nd < 2:
                for b in reps:
                    b.apply_deferred(0, 1, None)
                    b.apply_merge(None, 0, 1)
                continue
            exchanged += 1
            per = -(-nd // 2)
            bufs = [torch.full((2 * per * xs,), -1, dtype=torch.int64, device=dev) for _ in reps]
            for r, b in enumerate(reps):
                b.apply_deferred(r, 2, bufs[r].data_ptr())
                b.sync()
            allb = torch.cat([bufs[0][:per * xs], bufs[1][per * xs:]])  # the all-gather
            keys = allb.view(2 * per, xs)[:, 0]
            assert int((keys >= 0).sum()) == nd  # every deferred target was handled by exactly one rank
            for r, b in enumerate(reps):
                b.apply_merge(allb.data_ptr(), r, 2)
                b.sync()
        g0, g1 = b0.finish(), b1.finish()
    assert exchanged >= 3
    _same_graph(g0, o)
    _same_graph(g1, o)
    assert g0.n_evals_apply + g1.n_evals_apply >= o.n_evals_apply if hasattr(o, "n_evals_apply") else True


def test_rccl_exchange_path_single_rank(orc, hny):
    """The multi-GPU driver's exchange on the real thing: `nccl` (= RCCL) process group with one rank,
    all_gather_into_tensor on the device selection buffer handed to the C ABI, then apply from it.
    The graph must equal the plain single-GPU build."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from hannoy_amd.multigpu import Driver
    rng = np.random.default_rng(9)
    n, dim = 6000, 96
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, 0, vecs, draw_levels(n, 16, seed=9))
    ref = hny.build(items, M=16, M0=32, ef_construction=64, batch_frac=0.25, batch_max=1024)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        with hny.Builder(items, M=16, M0=32, ef_construction=64, batch_frac=0.25, batch_max=1024) as b:
            drv = Driver(b, torch, dist, 0, 1, dev, min_shard_batch=64, force_collective=True)
            drv.run()
            g = b.finish()
        assert drv.n_collectives > 0
    finally:
        dist.destroy_process_group()
    _same_graph(g, ref)


@pytest.mark.parametrize("metric", range(7))
def test_gpu_ingest_matches_host_and_oracle(orc, hny, metric, kat):
    """hny_encode_vectors_gpu (bit codecs by ballot, Cosine norms in the reference's x86 order) is
    byte-identical to the host path and to the oracle; quantiser golden vectors included."""
    rng = np.random.default_rng(50 + metric)
    for dim in (1, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 768, 1063):
        v = rng.uniform(-1, 1, (257, dim)).astype(np.float32)
        v[0, :] = 0.0
        v[1, 0] = -0.0
        v[2, 0] = np.inf
        v[3, 0] = np.nan
        gc, gh = hny.encode_vectors(metric, v, gpu=True)
        hc, hh = hny.encode_vectors(metric, v)
        assert np.array_equal(gc, hc) and np.array_equal(gh, hh)
        oc = orc.encode_vectors(metric, v)
        assert np.array_equal(gc, oc) and np.array_equal(gh, orc.make_headers(metric, dim, oc))
    for k in kat["kat6"]:
        m = hny.HAMMING if k["codec"] == "binary" else hny.BQ_COSINE
        codes, _ = hny.encode_vectors(m, np.array([k["input"]], np.float32), gpu=True)
        assert [format(b, "08b") for b in codes[0]] == k["bytes_bin"]


@pytest.mark.parametrize("metric,dim", [(0, 768), (1, 768), (0, 100), (1, 45), (0, 20), (1, 17), (0, 3),
                                        (1, 7), (2, 33), (0, 1536)])
def test_strict_mode_distances_equal_reference_x86_order(orc, hny, metric, dim):
    """x86_order = 1: distances bit-identical to what the reference computes on an AVX2+FMA host
    (AVX for dim >= 32, SSE for 16..31, scalar below; Manhattan scalar)."""
    rng = np.random.default_rng(300 + metric * 11 + dim)
    n = 200
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    lv = np.zeros(n, np.uint8)
    lv[0] = 1
    ds, items = _mk(orc, hny, metric, vecs, lv)
    with hny.Builder(items, M=4, M0=8, x86_order=True) as b:
        a = rng.integers(0, n, 1500).astype(np.uint32)
        c = rng.integers(0, n, 1500).astype(np.uint32)
        got = b.distances(a, c)
    want = np.array([orc.distance(metric, orc.ORDER_X86, dim, ds.codes[i], ds.headers[i], ds.codes[j],
                                  ds.headers[j]) for i, j in zip(a, c)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,frac,bmax", [(0, 1500, 768, 16, 32, 100, 0.0, 1),
                                                             (1, 1200, 100, 8, 16, 40, 0.0, 1),
                                                             (2, 800, 24, 6, 12, 24, 0.0, 1),
                                                             (0, 3000, 96, 8, 16, 48, 0.1, 128)])
def test_strict_mode_build_equals_reference_faithful_oracle(orc, hny, metric, n, dim, M, M0, ef, frac, bmax):
    """x86_order = 1 and batch_max = 1: the GPU graph equals the oracle's restatement of the
    reference run with one thread and its own x86 arithmetic (the mode KAT-1..5 pin)."""
    rng = np.random.default_rng(n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=n))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_X86, batch_frac=frac, batch_max=0 if bmax == 1 else bmax)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax, x86_order=True)
    _same_graph(g, o)


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,frac,bmax,lvM", [(1, 1200, 40, 16, 768, 64, 0.0, 1, 16),   # the fuzz pair
                                                                 (1, 900, 30, 16, 96, 48, 0.0, 1, 16),    # SSE path
                                                                 (0, 1500, 64, 80, 160, 64, 0.1, 64, 4),  # M > 64 too
                                                                 (2, 700, 24, 8, 100, 32, 0.25, 32, 8),
                                                                 (0, 600, 2100, 8, 70, 24, 0.1, 64, 8)])  # rows > 8 KB
def test_strict_mode_lists_beyond_64_slots_equal_x86_oracle(orc, hny, metric, n, dim, M, M0, ef, frac, bmax, lvM):
    """Strict mode (x86 summation order) with lists of more than 64 slots — the reference's fuzz pair
    build::<16, 768> (src/tests/fuzz.rs:86-87) in the reference's own arithmetic: the one-wave prune / add_link
    kernels take such lists 64 slots at a time (round 4; refused before).  The same kernels serve rows beyond 8 KB
    in the wave order (last case)."""
    rng = np.random.default_rng(3 * n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, lvM, seed=n))
    strict = dim < 2048
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_X86 if strict else orc.ORDER_WAVE, batch_frac=frac,
                  batch_max=0 if bmax == 1 else bmax)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax, x86_order=strict)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


def test_randomized_parameter_sweep(orc, hny):
    """40 seeded random configurations (all metrics, odd dims, M == M0, ef below M0, ef = 1, many
    entry points, duplicates, tiny inputs, sequential and batched schedules): GPU == oracle."""
    import os
    master = np.random.default_rng(2026)
    for case in range(int(os.environ.get("HNY_SWEEP_CASES", "40"))):
        metric = int(master.integers(0, 7))
        dim = int(master.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 64, 100, 127, 200, 257, 520]))
        M = int(master.choice([2, 3, 4, 6, 8, 12, 16, 24, 32]))
        M0 = int(master.choice([M, min(64, 2 * M), min(64, 3 * M)]))
        ef = int(master.choice([1, 2, 5, 16, 33, 64, 100, 130, 257]))
        n = int(master.choice([1, 2, 5, 30, 200, 700, 1500]))
        frac, bmax = [(0.0, 1), (0.25, 8), (0.5, 64), (1.0, 300), (0.05, 4096)][int(master.integers(0, 5))]
        alpha = float(master.choice([1.0, 1.0, 1.2]))
        rng = np.random.default_rng(1000 + case)
        vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        if n > 10:  # exact duplicates and near-duplicates
            vecs[rng.integers(0, n, n // 10)] = vecs[rng.integers(0, n, n // 10)]
        levels = draw_levels(n, M, seed=case)
        if case % 5 == 0:
            levels[:] = min(int(levels.max()), 1)  # many entry points (all items on the top level)
            if n > 60:
                continue
        ds = orc.Dataset.from_f32(metric, vecs, levels)
        items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
        tag = f"case {case}: metric {metric} dim {dim} M {M} M0 {M0} ef {ef} n {n} frac {frac} bmax {bmax}"
        o = orc.build(ds, M=M, M0=M0, ef=ef, alpha=alpha, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax)
        g = hny.build(items, M=M, M0=M0, ef_construction=ef, alpha=alpha, batch_frac=frac, batch_max=bmax)
        assert g.n_tie_pool_overflow == 0, tag
        assert np.array_equal(g.offsets, o.offsets) and np.array_equal(g.nbrs, o.nbrs), tag
        assert g.entry_points.tolist() == o.entry_points.tolist(), tag


@pytest.mark.parametrize("metric,dim", [(0, 48), (3, 256)])
def test_locality_ordered_batches_equal_oracle(orc, hny, metric, dim):
    """Batches of >= 2048 members take the locality-ordered path (descent -> sort -> walk/prune in
    that order).  Only the processing order changes: the graph must still equal the oracle's, also
    when the batch is searched in two separate member ranges (the multi-GPU sharding)."""
    rng = np.random.default_rng(77 + dim)
    n = 14000
    centres = rng.uniform(-1, 1, (20, dim))
    vecs = (centres[rng.integers(0, 20, n)] + 0.2 * rng.normal(size=(n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, 16, seed=5))
    o = orc.build(ds, M=16, M0=32, ef=64, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=4096, threads=8)
    g = hny.build(items, M=16, M0=32, ef_construction=64, batch_frac=1.0, batch_max=4096)
    _same_graph(g, o)
    with hny.Builder(items, M=16, M0=32, ef_construction=64, batch_frac=1.0, batch_max=4096) as b:
        big = 0
        while True:
            bt = b.next_batch()
            if bt.count == 0:
                break
            if bt.count >= 4096:  # two "ranks" on one GPU
                big += 1
                b.search(0, bt.count // 2)
                b.search(bt.count // 2, bt.count)
            else:
                b.search(0, bt.count)
            b.apply()
        g2 = b.finish()
    assert big >= 1
    _same_graph(g2, o)


@pytest.mark.parametrize("slots", ["0", "320", "512", "4096"])
def test_lds_visited_table_and_spill_equal_oracle(orc, hny, slots, monkeypatch):
    """The visited set is an LDS hash table that spills into the HBM bitset once 3/4 full
    (HNY_VIS_SLOTS forces its size; 320 spills after 176 items, 0 = bitset only): builds, k-NN search,
    the Reader's exhaustive fallback (table flushed into the bitset) and the filtered search must not
    depend on it."""
    monkeypatch.setenv("HNY_VIS_SLOTS", slots)
    rng = np.random.default_rng(77)
    n, dim, M, M0, ef = 3000, 64, 8, 16, 64
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    vecs[rng.integers(0, n, 40)] = 0.0  # zero vectors: distance 0 to everything -> ties, fallback
    ds, items = _mk(orc, hny, 0, vecs, draw_levels(n, M, seed=3))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=0.2, batch_max=256)
    qs = rng.uniform(-1, 1, (300, dim)).astype(np.float32)
    qs[:5] = 0.0
    qc = orc.encode_vectors(0, qs)
    qh = orc.make_headers(0, dim, qc)
    with hny.Builder(items, M=M, M0=M0, ef_construction=ef, batch_frac=0.2, batch_max=256) as b:
        b.run()
        g = b.finish()
        got = b.search_knn(qc, qh, k=10, ef_search=300)
        cand = ds.ids[rng.random(n) < 0.02]
        gotf = b.nns(qc, qh, k=10, ef_search=300, candidates=cand, linear_below=0)
        goti = b.nns(k=10, ef_search=20, query_items=ds.ids[:200])
    assert g.n_tie_pool_overflow == 0
    _same_graph(g, o)
    for got_, kw in ((got, {}), (gotf, dict(candidates=cand, linear_below=0)),
                     (goti, dict(query_items=ds.ids[:200]))):
        ef_s = 20 if "query_items" in kw else 300
        qq = (None, None) if "query_items" in kw else (qc, qh)
        want = orc.search(ds, g, qq[0], qq[1], k=10, ef_search=ef_s, order=orc.ORDER_WAVE, threads=8, **kw)
        assert np.array_equal(got_[2], want[2])
        for r in range(len(want[2])):
            c = int(want[2][r])
            assert np.array_equal(got_[0][r, :c], want[0][r, :c])
            assert np.array_equal(got_[1][r, :c].view(np.uint32), want[1][r, :c].view(np.uint32))


@pytest.mark.parametrize("env", [{"HNY_NO_LOCALITY": "1"}, {"HNY_STAGE_BYTES": "0"}, {"HNY_STAGE_BYTES": "8192"},
                                 {"HNY_NO_RB": "1"}, {"HNY_NO_FAST": "1"}])
def test_tuning_knobs_do_not_change_the_graph(orc, hny, env, monkeypatch):
    """Every switch that is still selectable by environment must build the oracle's graph too: no locality
    order, no / small LDS stage; the LDS beam (HNY_NO_RB) / the general kernels (HNY_NO_FAST) where the
    register beam / the specialised kernels are the default (256-d rows)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(5)
    n, dim, M, M0, ef = 7000, 256, 8, 16, 40
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, 0, vecs, draw_levels(n, M, seed=9))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=4096)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=1.0, batch_max=4096)
    _same_graph(g, o)


@pytest.mark.parametrize("buckets", ["0", "64", "192", None])
@pytest.mark.parametrize("metric,n,dim,M,M0,ef", [(3, 9000, 1024, 16, 32, 64), (0, 9000, 128, 16, 32, 100),
                                                  (1, 6000, 100, 8, 16, 48), (4, 5000, 2048, 12, 24, 40),
                                                  (3, 7000, 200, 16, 64, 100)])
def test_short_row_visited_bucket_table_equals_oracle(orc, hny, monkeypatch, buckets, metric, n, dim, M, M0, ef):
    """walk_layer_short's visited set (rows <= 512 B): buckets of four 16-bit remainders in LDS, the HBM bitset
    behind them for ids whose bucket is full.  Without the table (0), with one that overflows all the time
    (64 buckets = 256 ids for walks that mark ~1 000), a small one, and the default: the oracle's graph and
    evaluation counts every time — 8-, 16- and 32-lane row shapes, partial last units (100-d), full lists
    (M0 = 64), ties and duplicate vectors (Hamming), entry points from a layer above."""
    if buckets is not None:
        monkeypatch.setenv("HNY_VIS_BUCKETS", buckets)
    rng = np.random.default_rng(11 * n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    if metric >= 3:
        vecs[rng.integers(0, n, n // 20)] = vecs[rng.integers(0, n, n // 20)]
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=n + 1))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=2048)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=1.0, batch_max=2048)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


@pytest.mark.parametrize("metric", range(7))
def test_general_and_specialised_kernels_build_the_same_graph(orc, hny, metric, monkeypatch):
    """A plain build runs on the kernels specialised for its metric (hny_kernels.hip parts 1..7);
    HNY_NO_FAST=1 keeps it on the general kernels (part 0: the ones strict mode, incremental builds
    and the Reader use).  Both must give the oracle's graph, for every metric."""
    rng = np.random.default_rng(60 + metric)
    n, dim, M, M0, ef = 5000, 200 if metric < 3 else 1100, 8, 16, 48
    cent = rng.uniform(-1, 1, (32, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 32, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=2))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=2048, threads=8)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=1.0, batch_max=2048)
    _same_graph(g, o)
    monkeypatch.setenv("HNY_NO_FAST", "1")
    g0 = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=1.0, batch_max=2048)
    _same_graph(g0, o)
    # walk evaluations are determined by the schedule; the prune kernels differ in how many pairs they score
    # (k_prune_n8 tests eight candidates at a time on short rows, the workgroup prune four)
    assert g0.n_evals_walk == g.n_evals_walk == o.n_evals_walk


def test_locality_ordered_upper_level_batches_equal_oracle(orc, hny):
    """Batches of level >= 1 with >= 2048 members are processed in locality order too (descent to
    level+1, key sort, every layer's walk and prune through the permutation): M=4 puts 3/16 of the
    items on level 1 exactly, so a 16k-item build has a 3072-member level-1 batch."""
    rng = np.random.default_rng(21)
    n, dim, M, M0, ef = 16384, 32, 4, 8, 24
    cent = rng.uniform(-1, 1, (64, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 64, n)] + 0.2 * rng.standard_normal((n, dim))).astype(np.float32)
    lv = np.zeros(n, np.uint8)
    lv[::4] = 1          # 4096 items on level >= 1
    lv[::16] = 2         # 1024 of them on level >= 2
    lv[::256] = 3
    ds, items = _mk(orc, hny, 1, vecs, lv)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=4.0, batch_max=4096)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, batch_frac=4.0, batch_max=4096)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added


def test_levels_up_to_14(orc, hny):
    """get_default_probas (hnsw.rs:94-110) cuts at 1e-9: M = 4 reaches level 14.  A fresh build with a
    few items on levels 8..14, then an incremental round on top (the top items deleted, a new level-11
    item inserted), GPU == oracle; the search descends through all the layers."""
    rng = np.random.default_rng(14)
    n, dim, M, M0, ef = 2500, 24, 4, 8, 24
    vecs = rng.uniform(-1, 1, (n + 50, dim)).astype(np.float32)
    lv = draw_levels(n, M, seed=2)
    lv[[7, 300, 900]] = [14, 14, 12]
    lv[[11, 1200, 2000, 2400]] = [9, 8, 10, 8]
    kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=0.5, batch_max=256)
    kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=0.5, batch_max=256)
    ds = orc.Dataset.from_f32(1, vecs[:n], lv)
    items = hny.ItemSet(1, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    og = orc.build(ds, **kw_o)
    with hny.Builder(items, **kw_g) as b:
        b.run()
        gg = b.finish()
        qc = orc.encode_vectors(1, vecs[n:])
        qh = orc.make_headers(1, dim, qc)
        got = b.search_knn(qc, qh, k=5, ef_search=30)
    assert gg.max_level == 14 and gg.entry_points.tolist() == [7, 300]
    _same_graph(gg, og)
    want = orc.search(ds, gg, qc, qh, k=5, ef_search=30, order=orc.ORDER_WAVE, threads=4)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    # incremental: both entry points and a level-10 item go, 50 items come, one of them on level 11
    to_delete = [7, 300, 2000]
    keep = np.array([i for i in range(n + 50) if i not in to_delete], np.uint32)
    to_insert = list(range(n, n + 50))
    ins_lv = draw_levels(50, M, seed=3)
    ins_lv[10] = 11
    ds2 = orc.Dataset.from_f32(1, vecs[keep], np.zeros(len(keep), np.uint8), keep)
    items2 = hny.ItemSet(1, dim, ds2.ids, ds2.codes, ds2.headers, ins_lv)
    og2 = orc.build_incremental(ds2, og, to_insert, ins_lv, to_delete, **kw_o)
    gg2 = hny.build_incremental(items2, gg, to_insert, to_delete, **kw_g)
    _same_graph(gg2, og2)


@pytest.mark.parametrize("metric,n,dim", [(1, 300, 16), (0, 700, 768), (3, 130, 256), (1, 3000, 16)])
def test_every_item_an_entry_point(orc, hny, metric, n, dim):
    """A small index whose items all drew level 0 (M=32, n=100: 4 % of the seeds) has every item as an
    entry point (hnsw.rs:278-285): all of them seed every walk.  Fresh build, search, and an
    incremental round that keeps max_level 0."""
    rng = np.random.default_rng(n)
    vecs = rng.uniform(-1, 1, (n + 40, dim)).astype(np.float32)
    kw_o = dict(M=8, M0=16, ef=32, order=orc.ORDER_WAVE, batch_frac=0.5, batch_max=64)
    kw_g = dict(M=8, M0=16, ef_construction=32, batch_frac=0.5, batch_max=64)
    ds = orc.Dataset.from_f32(metric, vecs[:n], np.zeros(n, np.uint8))
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    og = orc.build(ds, **kw_o)
    qc = orc.encode_vectors(metric, vecs[n:])
    qh = orc.make_headers(metric, dim, qc)
    with hny.Builder(items, **kw_g) as b:
        b.run()
        gg = b.finish()
        got = b.search_knn(qc, qh, k=5, ef_search=20)
        goti = b.nns(k=5, ef_search=20, query_items=ds.ids[:50])
    assert gg.max_level == 0 and len(gg.entry_points) == n
    _same_graph(gg, og)
    want = orc.search(ds, gg, qc, qh, k=5, ef_search=20, order=orc.ORDER_WAVE, threads=4)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    wanti = orc.search(ds, gg, None, None, k=5, ef_search=20, order=orc.ORDER_WAVE, threads=4, query_items=ds.ids[:50])
    assert np.array_equal(goti[0], wanti[0]) and np.array_equal(goti[2], wanti[2])
    to_delete = [3, 17]
    keep = np.array([i for i in range(n + 40) if i not in to_delete], np.uint32)
    to_insert = list(range(n, n + 40))
    ins_lv = np.zeros(40, np.uint8)
    ds2 = orc.Dataset.from_f32(metric, vecs[keep], np.zeros(len(keep), np.uint8), keep)
    items2 = hny.ItemSet(metric, dim, ds2.ids, ds2.codes, ds2.headers, ins_lv)
    og2 = orc.build_incremental(ds2, og, to_insert, ins_lv, to_delete, **kw_o)
    gg2 = hny.build_incremental(items2, gg, to_insert, to_delete, **kw_g)
    _same_graph(gg2, og2)


def _multi_case(orc, hny, metric=0, n=9000, dim=96, M=8, M0=16, ef=48):
    rng = np.random.default_rng(77 + metric)
    cent = rng.uniform(-1, 1, (24, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 24, n)] + 0.25 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=6))
    kw = dict(M=M, M0=M0, batch_frac=1.0, batch_max=2048)
    o = orc.build(ds, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    return ds, items, o, dict(ef_construction=ef, **kw)


def test_native_multi_gpu_driver_one_rank_over_rccl(orc, hny):
    """hny_build(n_gpus=1, devices=[0]) takes the multi-GPU host of hny_multi.cpp with a REAL RCCL
    communicator of size one (ncclCommInitAll + ncclAllGather on the builder's stream, both
    exchanges of every batch): byte for byte the plain build, which equals the oracle."""
    ds, items, o, kw = _multi_case(orc, hny)
    os.environ["HNY_MGPU_MIN_BATCH"] = "8"
    os.environ["HNY_MGPU_MIN_DEFERRED"] = "1"
    try:
        g = hny.build(items, n_gpus=1, devices=[0], **kw)
    finally:
        del os.environ["HNY_MGPU_MIN_BATCH"], os.environ["HNY_MGPU_MIN_DEFERRED"]
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    g1 = hny.build(items, **kw)
    _same_graph(g1, g)


@pytest.mark.parametrize("world,metric", [(2, 0), (3, 3), (4, 1)])
def test_native_multi_gpu_driver_ranks_share_one_gpu(orc, hny, monkeypatch, world, metric):
    """The native driver with `world` ranks mapped to ONE GPU (HNY_MGPU_SHIM=1: the collective is
    replaced by device-to-device copies between the ranks' buffers + a host rendezvous, because RCCL
    refuses duplicate devices): sharded searches, sharded deferred re-prunes, both exchanges, one host
    thread per rank.  Every replica must export the oracle's graph (HNY_MGPU_VERIFY compares them)."""
    monkeypatch.setenv("HNY_MGPU_SHIM", "1")
    monkeypatch.setenv("HNY_MGPU_VERIFY", "1")
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "16")
    monkeypatch.setenv("HNY_MGPU_MIN_DEFERRED", "2")
    ds, items, o, kw = _multi_case(orc, hny, metric=metric, dim=96 if metric < 3 else 512)
    seen = []
    g = hny.build(items, devices=[0] * world, progress=lambda d, t: seen.append((d, t)), **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added
    assert seen and seen[-1][0] == len(ds.ids)
    with pytest.raises(hny.BuildCancelled):
        hny.build(items, devices=[0] * world, cancel=lambda: len(seen) > 0, **kw)


def test_resident_multi_builder_runs_twice(orc, hny, monkeypatch):
    """hny_multi_builder_*: replicas created once (two ranks on GPU 0 through the shim), two complete
    builds on them — both equal the oracle's graph and the one-GPU build's counters; the replica handle
    searches like a plain builder.  This is what `bench.py --gpus N --native` times."""
    monkeypatch.setenv("HNY_MGPU_SHIM", "1")
    monkeypatch.setenv("HNY_MGPU_VERIFY", "1")
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "16")
    monkeypatch.setenv("HNY_MGPU_MIN_DEFERRED", "2")
    ds, items, o, kw = _multi_case(orc, hny)
    g1 = hny.build(items, **kw)
    with hny.MultiBuilder(items, devices=[0, 0], **kw) as mb:
        assert mb.world == 2
        for _ in range(2):
            g = mb.run()
            _same_graph(g, o)
            assert (g.n_links_added, g.n_evals_walk, g.n_evals_prune, g.n_evals_apply) == \
                   (g1.n_links_added, g1.n_evals_walk, g1.n_evals_prune, g1.n_evals_apply)
            assert mb.n_collectives > 0
        rng = np.random.default_rng(3)
        qs = rng.uniform(-1, 1, (64, items.dim)).astype(np.float32)
        qc = orc.encode_vectors(items.metric, qs)
        qh = orc.make_headers(items.metric, items.dim, qc)
        for r in (0, 1):
            ids, dists, counts = mb.replica(r).search_knn(qc, qh, k=5, ef_search=32)
            oi, od, oc = orc.search(ds, o, qc, qh, k=5, ef_search=32, order=orc.ORDER_WAVE)
            assert np.array_equal(counts, oc) and np.array_equal(ids, oi) and np.array_equal(dists, od)


def test_native_multi_gpu_overflow_in_a_shard_fails_the_build(orc, hny, monkeypatch):
    """A device-side error inside the shard of a rank >= 1 exists only in THAT rank's counter block (rank 0
    never walked those members), while the clipped selection is all-gathered into every replica: every
    rank's error words are folded into the ranks' agreement, so the build fails exactly as it does on one
    GPU — with HNY_MGPU_VERIFY off, i.e. without the replica comparison that used to mask this.  The error
    here: a tie-pool overflow with the heap-walk safety net switched off (HNY_NO_POOL_RETRY=1).  With the
    net on, the same sharded build equals the oracle."""
    monkeypatch.setenv("HNY_MGPU_SHIM", "1")
    monkeypatch.delenv("HNY_MGPU_VERIFY", raising=False)
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "4")
    monkeypatch.setenv("HNY_MGPU_MIN_DEFERRED", "2")
    h, ds, items, metric, dim, M, M0, ef, bmax, frac = _tie_pool_fixture(orc, hny)
    kw = dict(M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax, threads=8)
    _same_graph(hny.build(items, devices=[0, 0], **kw), o)
    monkeypatch.setenv("HNY_NO_POOL_RETRY", "1")
    for world in (2, 3):
        with pytest.raises(hny.HannoyError) as e:
            hny.build(items, devices=[0] * world, **kw)
        assert e.value.code == -7 and "tie pool overflow" in str(e.value)


@pytest.mark.parametrize("M,M0", [(8, 16), (16, 96)])
def test_native_multi_gpu_driver_incremental(orc, hny, monkeypatch, M, M0):
    """hny_build_incremental through the native driver (two ranks on one GPU): deletes + inserts on
    top of a stored graph, fill_gaps_from_deleted on every replica (k_fill_gaps_wg for the wide lists
    of the second case), == oracle."""
    monkeypatch.setenv("HNY_MGPU_SHIM", "1")
    monkeypatch.setenv("HNY_MGPU_VERIFY", "1")
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "16")
    rng = np.random.default_rng(5)
    n, dim, ef = 3000, 32, 40
    vecs = rng.uniform(-1, 1, (n + 600, dim)).astype(np.float32)
    kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=0.5, batch_max=512)
    kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=0.5, batch_max=512)
    ds, items = _mk(orc, hny, 1, vecs[:n], draw_levels(n, M, seed=8))
    og = orc.build(ds, threads=8, **kw_o)
    gg = hny.build(items, devices=[0, 0], **kw_g)
    _same_graph(gg, og)
    to_delete = np.sort(rng.choice(n, 300, replace=False)).astype(np.uint32)
    alive = np.setdiff1d(np.arange(n + 600, dtype=np.uint32), to_delete)
    to_insert = np.arange(n, n + 600, dtype=np.uint32)
    lv = draw_levels(600, M, seed=9)
    ds2 = orc.Dataset.from_f32(1, vecs[alive], np.zeros(len(alive), np.uint8), alive)
    items2 = hny.ItemSet(1, dim, ds2.ids, ds2.codes, ds2.headers, lv)
    og2 = orc.build_incremental(ds2, og, to_insert, lv, to_delete, **kw_o)
    gg2 = hny.build_incremental(items2, gg, to_insert, to_delete, devices=[0, 0], **kw_g)
    _same_graph(gg2, og2)


@pytest.mark.parametrize("metric,n,dim,tile", [(0, 12000, 64, 16), (1, 12000, 256, 0), (3, 10000, 1024, 7)])
def test_xcd_tiled_walk_queue_equals_oracle(orc, hny, monkeypatch, metric, n, dim, tile):
    """the level-0 walks of a locality-ordered batch take their members from 8 per-XCD counters (tiles of
    HNY_XCD_TILE consecutive members, stealing at the tail; default 512 for rows >= 1 KB, i.e. the
    256-d case here with its 8 192-member batches): every member is still walked exactly once — graph
    == oracle edge for edge, same walk evaluations"""
    if tile:
        monkeypatch.setenv("HNY_XCD_TILE", str(tile))
    rng = np.random.default_rng(n + dim)
    cent = rng.uniform(-1, 1, (16, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 16, n)] + 0.25 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, 16, seed=5))
    kw = dict(batch_frac=1.0, batch_max=8192)
    o = orc.build(ds, M=16, M0=32, ef=48, order=orc.ORDER_WAVE, threads=8, **kw)
    g = hny.build(items, M=16, M0=32, ef_construction=48, **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


def test_64_bit_hamming_codes_build_and_search_equal_oracle(orc, hny):
    """50 000 x 64-bit Hamming codes (65 distinct distances: every walk drags hundreds of ties along): the build
    — members whose tie pool overflows go through k_walk_heap — equals the oracle's edge for edge, and a search
    with ef_search = 300, whose pool overflows for most queries, is repeated on the heap-queue searcher
    (k_nns_filtered without a filter) and returns the restated Reader's ids, distances and counts."""
    rng = np.random.default_rng(11)
    n, dim = 50000, 64
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, 3, vecs, draw_levels(n, 16, seed=3))
    kw = dict(batch_frac=1.0, batch_max=8192)
    o = orc.build(ds, M=16, M0=32, ef=64, order=orc.ORDER_WAVE, threads=8, **kw)
    qs = rng.uniform(-1, 1, (300, dim)).astype(np.float32)
    qc = orc.encode_vectors(3, qs)
    qh = orc.make_headers(3, dim, qc)
    with hny.Builder(items, M=16, M0=32, ef_construction=64, **kw) as b:
        b.run()
        g = b.finish()
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
        for k, ef in ((10, 300), (50, 100)):
            ids, dists, counts = b.search_knn(qc, qh, k=k, ef_search=ef)
            oi, od, oc = orc.search(ds, o, qc, qh, k=k, ef_search=ef, order=orc.ORDER_WAVE, threads=8)
            assert np.array_equal(counts, oc) and np.array_equal(ids, oi) and np.array_equal(dists, od)


def test_search_with_result_sets_beyond_the_lds_survives_a_tie_pool_overflow(orc, hny, monkeypatch):
    """ef_search >= 4 096 keeps the result set in HBM; on 64-bit Hamming codes the walk's tie pool overflows for
    such a search (found at C5 with ef_search = 6 400: the call failed with "tie pool overflow").  Those queries
    are walked again by k_walk_heap in reader mode (`candidates` and `res` as heaps in HBM) and return the
    restated Reader's ids, distances and counts.  Second half: the same path forced for EVERY query
    (HNY_POOL_FORCE_RETRY=1) on an index with zero vectors and ef_search > n, where the Reader's exhaustive
    fallback (reader.rs:771-795) runs inside the heap kernel, and with ef_search = 300 through the LDS searcher."""
    rng = np.random.default_rng(21)
    n, dim = 12000, 64
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, 3, vecs, draw_levels(n, 16, seed=4))
    qs = rng.uniform(-1, 1, (48, dim)).astype(np.float32)
    qc = orc.encode_vectors(3, qs)
    qh = orc.make_headers(3, dim, qc)
    kw = dict(batch_frac=1.0, batch_max=4096)
    with hny.Builder(items, M=16, M0=32, ef_construction=48, **kw) as b:
        b.run()
        g = b.finish()
        got = b.search_knn(qc, qh, k=10, ef_search=5000)
        monkeypatch.setenv("HNY_NO_POOL_RETRY", "1")  # without the safety net the same search fails loudly
        with pytest.raises(hny.HannoyError) as e:
            b.search_knn(qc, qh, k=10, ef_search=5000)
        assert e.value.code == -7 and "tie pool overflow" in str(e.value)
        monkeypatch.delenv("HNY_NO_POOL_RETRY")
    want = orc.search(ds, g, qc, qh, k=10, ef_search=5000, order=orc.ORDER_WAVE, threads=8)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])

    monkeypatch.setenv("HNY_POOL_FORCE_RETRY", "1")
    n, dim, M, M0, ef = 3000, 64, 8, 16, 64
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    vecs[rng.integers(0, n, 40)] = 0.0  # zero vectors: distance 0 to everything -> ties, trapped walks
    ds, items = _mk(orc, hny, 0, vecs, draw_levels(n, M, seed=3))
    qs = rng.uniform(-1, 1, (100, dim)).astype(np.float32)
    qs[:5] = 0.0
    qc = orc.encode_vectors(0, qs)
    qh = orc.make_headers(0, dim, qc)
    with hny.Builder(items, M=M, M0=M0, ef_construction=ef, batch_frac=0.2, batch_max=256) as b:
        b.run()
        g = b.finish()
        for k, ef_s in ((10, 4500), (10, 300), (3500, 10)):
            got = b.search_knn(qc, qh, k=k, ef_search=ef_s)
            want = orc.search(ds, g, qc, qh, k=k, ef_search=ef_s, order=orc.ORDER_WAVE, threads=8)
            assert np.array_equal(got[2], want[2]), (k, ef_s)
            for r in range(len(want[2])):
                c = int(want[2][r])
                assert np.array_equal(got[0][r, :c], want[0][r, :c]), (k, ef_s, r)
                assert np.array_equal(got[1][r, :c].view(np.uint32), want[1][r, :c].view(np.uint32)), (k, ef_s, r)


@pytest.mark.parametrize("metric,n,dim,M,M0,ef", [(0, 9000, 128, 16, 32, 100), (3, 9000, 1024, 16, 32, 64),
                                                  (1, 5000, 60, 12, 24, 40), (4, 4000, 700, 8, 16, 120),
                                                  (2, 3000, 20, 5, 9, 33), (5, 3000, 2000, 16, 64, 64),
                                                  (6, 2500, 8, 3, 5, 16)])
def test_one_wave_prune_for_short_rows_equals_oracle(orc, hny, monkeypatch, metric, n, dim, M, M0, ef):
    """k_prune_n8 (rows <= 512 B: one wave per query, eight candidates per chunk, one per 8-lane group) builds
    the oracle's graph for every row shape it serves (8 / 16 / 32 lanes in the wave order, f32 and bit codes,
    odd caps, more selected rows than the LDS stage holds), and so does the workgroup prune it replaces there
    (HNY_PRUNE_N8=0)."""
    rng = np.random.default_rng(n + dim)
    cent = rng.uniform(-1, 1, (12, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 12, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=7))
    for kw in (dict(batch_frac=1.0, batch_max=4096), dict(batch_frac=0.1, batch_max=128)):
        o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
        g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    monkeypatch.setenv("HNY_PRUNE_N8", "0")
    g0 = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g0, o)


@pytest.mark.parametrize("metric,n,dim,M,M0,ef", [(3, 9000, 1024, 16, 32, 64), (0, 6000, 24, 8, 16, 40),
                                                  (5, 5000, 500, 12, 24, 48), (3, 5000, 700, 16, 48, 32),
                                                  (1, 4000, 32, 16, 32, 100), (4, 3000, 64, 4, 8, 20)])
def test_tiny_rows_build_equals_oracle(orc, hny, monkeypatch, metric, n, dim, M, M0, ef):
    """Rows of at most 128 B (1024-bit codes, <= 32-d f32; 8 lanes per row): graph and walk evaluations equal
    the oracle's — also with lists longer than 32 and tie-heavy codes.  (Round 3 tried requesting the rows of
    ALL listed neighbours together with the visited test on these rows: identical graphs, 3 % slower — the
    walk is bound by the number of random memory accesses, not by their dependency chain)"""
    rng = np.random.default_rng(n + dim + M0)
    cent = rng.uniform(-1, 1, (10, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 10, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=4))
    for kw in (dict(batch_frac=1.0, batch_max=4096), dict(batch_frac=0.1, batch_max=64)):
        o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
        g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


def test_ids_sorted_by_cluster_do_not_break_the_index(orc, hny, monkeypatch):
    """Batch-synchronous insertion and ids that are sorted by cluster (documents grouped by topic): a batch of
    consecutive ids is a few whole clusters none of whose points are in the graph yet — every member searches
    a graph that does not contain its neighbourhood, and recall collapses (round 3 measured 0.42 instead of
    0.95 at C2).  The schedule therefore takes the items of a level group in a fixed pseudo-random order
    (hny_rust_sort.h shuffle_level_groups, restated in the oracle): recall on sorted ids equals recall on
    shuffled ids, the graph equals the oracle's; hny_build_opts.schedule = HNY_SCHED_NO_SHUFFLE shows what it guards
    against."""
    rng = np.random.default_rng(8)
    n, dim, ncl, nq = 30000, 48, 30, 300
    cent = rng.uniform(-1, 1, (ncl, dim)).astype(np.float32)
    which = np.sort(rng.integers(0, ncl, n))            # ids in cluster order
    vecs = (cent[which] + 0.15 * rng.standard_normal((n, dim))).astype(np.float32)
    qs = (cent[rng.integers(0, ncl, nq)] + 0.15 * rng.standard_normal((nq, dim))).astype(np.float32)
    d2 = ((qs ** 2).sum(1)[:, None] - 2 * qs @ vecs.T + (vecs ** 2).sum(1)[None, :])
    truth = np.argsort(d2, axis=1)[:, :10]
    levels = draw_levels(n, 16, seed=1)
    ds, items = _mk(orc, hny, 1, vecs, levels)
    qc = orc.encode_vectors(1, qs)
    qh = orc.make_headers(1, dim, qc)
    kw = dict(M=16, M0=32, ef_construction=64, batch_frac=1.0, batch_max=8192)

    def recall():
        with hny.Builder(items, **kw) as b:
            b.run()
            g = b.finish()
            ids, _, cnt = b.search_knn(qc, qh, k=10, ef_search=64)
        hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(nq))
        return g, hit / truth.size
    g, r_shuffled = recall()
    o = orc.build(ds, M=16, M0=32, ef=64, order=orc.ORDER_WAVE, threads=8, batch_frac=1.0, batch_max=8192)
    _same_graph(g, o)
    kw["schedule"] = hny.SCHED_NO_SHUFFLE
    g1, r_runs = recall()
    o1 = orc.build(ds, M=16, M0=32, ef=64, order=orc.ORDER_WAVE, threads=8, batch_frac=1.0, batch_max=8192,
                   schedule=hny.SCHED_NO_SHUFFLE)
    _same_graph(g1, o1)
    assert r_shuffled > 0.9, r_shuffled
    assert r_runs < r_shuffled - 0.1, (r_runs, r_shuffled)  # consecutive runs of sorted ids: a visibly worse index


def test_update_that_adds_a_new_region(orc, hny, monkeypatch):
    """An update whose new items form regions of their own (a new topic appended to an index).  Rounds 1-2
    counted the surviving old records as "already inserted", so the 12 000 new items went in as one batch whose
    members cannot see each other: recall 0.51 on the new region.  An update's batches now ramp up from one
    member like a fresh build's: recall on the new region is what a fresh build of everything gives, and the
    graph equals the oracle's under the same rule; HNY_SCHED_UPDATE_NO_RAMP shows what it guards against."""
    rng = np.random.default_rng(5)
    dim, nA, nB, nq = 48, 30000, 12000, 300
    centA = rng.uniform(-1, 1, (30, dim)).astype(np.float32)
    centB = rng.uniform(-1, 1, (12, dim)).astype(np.float32)
    A = (centA[rng.integers(0, 30, nA)] + 0.15 * rng.standard_normal((nA, dim))).astype(np.float32)
    B = (centB[rng.integers(0, 12, nB)] + 0.15 * rng.standard_normal((nB, dim))).astype(np.float32)
    qs = (centB[rng.integers(0, 12, nq)] + 0.15 * rng.standard_normal((nq, dim))).astype(np.float32)
    allv = np.concatenate([A, B])
    d2 = ((qs ** 2).sum(1)[:, None] - 2 * qs @ allv.T + (allv ** 2).sum(1)[None, :])
    truth = np.argsort(d2, axis=1)[:, :10]
    kw = dict(M=16, M0=32, ef_construction=64)
    kwo = dict(M=16, M0=32, ef=64, order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=65536, threads=8)
    dsA, itA = _mk(orc, hny, 1, A, draw_levels(nA, 16, seed=1))
    gA = hny.build(itA, **kw)
    oA = orc.build(dsA, **kwo)
    _same_graph(gA, oA)
    lvB = draw_levels(nB, 16, seed=2)
    dsAll = orc.Dataset.from_f32(1, allv, np.zeros(nA + nB, np.uint8))
    itAll = hny.ItemSet(1, dim, dsAll.ids, dsAll.codes, dsAll.headers, lvB)
    ins = np.arange(nA, nA + nB, dtype=np.uint32)
    qc = orc.encode_vectors(1, qs)
    qh = orc.make_headers(1, dim, qc)

    def recall(g):
        with hny.Builder(itAll, prev=g, load=True, **kw) as b:
            ids, _, cnt = b.search_knn(qc, qh, k=10, ef_search=64)
        return sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(nq)) / truth.size
    g2 = hny.build_incremental(itAll, gA, ins, [], **kw)
    o2 = orc.build_incremental(dsAll, oA, ins, lvB, [], **kwo)
    _same_graph(g2, o2)
    assert g2.n_batches > 10
    r = recall(g2)
    assert r > 0.93, r
    g3 = hny.build_incremental(itAll, gA, ins, [], schedule=hny.SCHED_UPDATE_NO_RAMP, **kw)
    _same_graph(g3, orc.build_incremental(dsAll, oA, ins, lvB, [], schedule=hny.SCHED_UPDATE_NO_RAMP, **kwo))
    assert g3.n_batches <= 5 and recall(g3) < r - 0.2


def _tie_pool_fixture(orc, hny):
    h = np.load(os.path.join(os.path.dirname(__file__), "golden", "tie_pool_overflow_hamming3_m0_333.npz"))
    metric, dim, M, M0, ef, bmax = [int(x) for x in h["params"]]
    frac = float(h["frac"][0])
    ds = orc.Dataset.from_f32(metric, h["mat0"], h["lv0"], h["ids0"])
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    return h, ds, items, metric, dim, M, M0, ef, bmax, frac


def test_tie_pool_overflow_equals_oracle(orc, hny, monkeypatch):
    """Found by scripts/soak_random_configs.py: 3-bit Hamming codes (four distinct distances) and M0 = 333 —
    hundreds of evicted candidates tie with the result set's maximum at once, more than the walk's
    128-slot tie pool holds.  Round 2 failed such a build (HNY_ERR_DEVICE); now the walk lists the member
    and k_walk_heap walks it again with `candidates` and `res` as real heaps in HBM — the reference's own
    data structures, nothing to overflow — so the build and the update equal the oracle's, edge for edge.
    HNY_NO_POOL_RETRY=1 switches the safety net off: the same input fails loudly, as before."""
    h, ds, items, metric, dim, M, M0, ef, bmax, frac = _tie_pool_fixture(orc, hny)
    kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax)
    kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax)
    og = orc.build(ds, threads=8, **kw_o)
    gg = hny.build(items, **kw_g)
    assert gg.n_tie_pool_overflow == 0
    _same_graph(gg, og)
    assert gg.n_links_added == og.n_links_added and gg.n_evals_walk == og.n_evals_walk
    ds2 = orc.Dataset.from_f32(metric, h["mat1"], np.zeros(len(h["ids1"]), np.uint8), h["ids1"])
    items2 = hny.ItemSet(metric, dim, ds2.ids, ds2.codes, ds2.headers, h["lv1"])
    og2 = orc.build_incremental(ds2, og, h["ins1"], h["lv1"], h["del1"], **kw_o)
    gg2 = hny.build_incremental(items2, gg, h["ins1"], h["del1"], **kw_g)
    _same_graph(gg2, og2)
    assert gg2.n_tie_pool_overflow == 0 and gg2.n_evals_walk == og2.n_evals_walk
    monkeypatch.setenv("HNY_NO_POOL_RETRY", "1")
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, **kw_g)
    assert e.value.code == -7 and "tie pool overflow" in str(e.value)


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,every", [(0, 2500, 96, 8, 16, 40, 1), (3, 3000, 256, 8, 16, 32, 3),
                                                       (1, 1500, 768, 16, 32, 64, 2), (4, 2000, 128, 6, 80, 24, 1)])
def test_heap_walk_equals_oracle(orc, hny, monkeypatch, metric, n, dim, M, M0, ef, every):
    """k_walk_heap on ordinary inputs: HNY_POOL_FORCE_RETRY=k hands every k-th member of every walk launch
    (descent launches, every layer, locality order, M0 > 64) to the heap kernel as if its tie pool had
    overflowed — graph, link count and walk evaluations stay the oracle's."""
    monkeypatch.setenv("HNY_POOL_FORCE_RETRY", str(every))
    rng = np.random.default_rng(n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=9))
    for kw in (dict(batch_frac=1.0, batch_max=4096), dict(batch_frac=0.05, batch_max=64)):
        o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
        g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added
        assert g.n_evals_walk == o.n_evals_walk  # the discarded first walk of a handed-over member is not counted
    # the retry path has two tiers (round 5): heaps of 2^18 entries on many blocks first, heaps that hold every item
    # for the members that outgrow those.  Forty-entry first-tier heaps send most of the handed-over members on.
    monkeypatch.setenv("HNY_HEAP_SMALL_CAP", "40")
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


@pytest.mark.parametrize("tile", [16, 0])
def test_xcd_tiled_search_queue_equals_oracle(orc, hny, monkeypatch, tile):
    """Reader::nns for a batch of 9 000 queries on 1 KB rows: the locality-ordered layer-0 walks take
    their queries from the 8 per-XCD counters (default tile 512 -> 8 192 queries and more; 16 forces many
    small tiles and stealing): ids, distances and counts == the restated Reader"""
    if tile:
        monkeypatch.setenv("HNY_XCD_TILE", str(tile))
    rng = np.random.default_rng(77)
    n, dim, nq = 6000, 256, 9000
    cent = rng.uniform(-1, 1, (16, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 16, n)] + 0.25 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, 1, vecs, draw_levels(n, 16, seed=6))
    qs = (cent[rng.integers(0, 16, nq)] + 0.25 * rng.standard_normal((nq, dim))).astype(np.float32)
    qc = orc.encode_vectors(1, qs)
    qh = orc.make_headers(1, dim, qc)
    with hny.Builder(items, M=16, M0=32, ef_construction=48, batch_frac=1.0, batch_max=4096) as b:
        b.run()
        g = b.finish()
        ids, dists, counts = b.search_knn(qc, qh, k=10, ef_search=40)
    oi, od, oc = orc.search(ds, g, qc, qh, k=10, ef_search=40, order=orc.ORDER_WAVE, threads=8)
    assert np.array_equal(oc, counts) and np.array_equal(oi, ids)
    assert np.array_equal(od.view(np.uint32), dists.view(np.uint32))


def test_cross_lane_primitives_match_shfl_xor(hny):
    """xshfl<1..32> (DPP moves, v_permlane16/32_swap) and the swap-based fold steps == __shfl_xor on every
    lane, for u32 / f32 / u64 payloads: the distance reductions keep the wave order bit for bit"""
    assert hny.selftest_lane_ops() == [0] * 64


@pytest.mark.parametrize("metric,n,dim,M,M0,ef", [(0, 4000, 64, 16, 96, 64), (1, 3000, 768, 16, 200, 100),
                                                   (3, 5000, 256, 8, 256, 48), (0, 2500, 20, 64, 128, 150),
                                                   (0, 3000, 32, 16, 768, 100),   # the fuzz pair, fuzz.rs:86-87
                                                   (1, 1500, 768, 16, 1024, 64)])  # > 64 KB of LDS per workgroup
def test_m0_beyond_64_equals_oracle(orc, hny, metric, n, dim, M, M0, ef):
    """64 < M0 <= 1024 (lists walked 64 slots at a time, workgroup kernels hold them whole): fresh build
    == oracle edge for edge, with the same walk evaluations, for lists that really fill up beyond 64
    slots; the k-NN search on that graph == the restated Reader."""
    rng = np.random.default_rng(n + M0)
    cent = rng.uniform(-1, 1, (6, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 6, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=M0))
    kw = dict(batch_frac=0.25, batch_max=1024)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    qs = rng.uniform(-1, 1, (300, dim)).astype(np.float32)
    qc = orc.encode_vectors(metric, qs)
    qh = orc.make_headers(metric, dim, qc)
    with hny.Builder(items, M=M, M0=M0, ef_construction=ef, **kw) as b:
        b.run()
        g = b.finish()
        ids, dists, counts = b.search_knn(qc, qh, k=10, ef_search=64)
        fids, fd, fc = b.nns(qc[:50], qh[:50], k=5, ef_search=40, candidates=np.arange(0, n, 3, dtype=np.uint32))
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    deg0 = np.diff(g.offsets.astype(np.int64))[g.rec_layer == 0]
    assert deg0.max() > 64  # the wide lists are really used
    oi, od, oc = orc.search(ds, g, qc, qh, k=10, ef_search=64, order=orc.ORDER_WAVE, threads=8)
    assert np.array_equal(oc, counts) and np.array_equal(oi, ids)
    assert np.array_equal(od.view(np.uint32), dists.view(np.uint32))
    oi, od, oc = orc.search(ds, g, qc[:50], qh[:50], k=5, ef_search=40, order=orc.ORDER_WAVE, threads=8,
                            candidates=np.arange(0, n, 3, dtype=np.uint32))
    assert np.array_equal(oc, fc) and np.array_equal(oi, fids)


def test_m0_beyond_64_native_multi_gpu(orc, hny, monkeypatch):
    """the sharded link phase with wide lists (exchange records of 2 + M0 words, k_apply_merge over
    more than 64 slots): three ranks on one GPU == oracle."""
    monkeypatch.setenv("HNY_MGPU_SHIM", "1")
    monkeypatch.setenv("HNY_MGPU_VERIFY", "1")
    monkeypatch.setenv("HNY_MGPU_MIN_BATCH", "16")
    monkeypatch.setenv("HNY_MGPU_MIN_DEFERRED", "2")
    rng = np.random.default_rng(8)
    n, dim, M, M0, ef = 3000, 48, 16, 80, 64
    cent = rng.uniform(-1, 1, (4, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 4, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, 1, vecs, draw_levels(n, M, seed=3))
    kw = dict(batch_frac=0.5, batch_max=512)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, devices=[0, 0, 0], **kw)
    _same_graph(g, o)
    g1 = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    # the counters of the three ranks add up to the one-GPU build's (replicated work counted once)
    assert (g.n_evals_walk, g.n_evals_prune, g.n_evals_apply) == (g1.n_evals_walk, g1.n_evals_prune, g1.n_evals_apply)
    assert g.n_evals_walk == o.n_evals_walk and g.n_distance_evals == g1.n_distance_evals


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,kw", [
    (0, 5000, 64, 16, 32, 600, dict(batch_frac=1.0, batch_max=2048)),
    (3, 6000, 256, 8, 16, 1500, dict(batch_frac=1.0, batch_max=4096)),
    (1, 2500, 768, 16, 32, 4095, dict(batch_frac=0.25, batch_max=512)),   # ef > n: every walk keeps the whole index
    (0, 4000, 128, 12, 100, 1000, dict(batch_frac=1.0, batch_max=1024)),  # lists of more than 64 slots
    (0, 9000, 48, 16, 32, 5000, dict(batch_frac=1.0, batch_max=1024)),    # result sets in HBM
    (3, 7000, 128, 8, 16, 6500, dict(batch_frac=0.5, batch_max=2048)),
])
def test_ef_construction_beyond_512_equals_oracle(orc, hny, metric, n, dim, M, M0, ef, kw):
    """ef_construction up to 4 095 (the reference takes any value, writer.rs:49): beams of more than 512 entries
    live in the walk's LDS like the Reader's, robust_prune takes the candidate list from LDS or from HBM — the
    graphs, link counts and walk evaluations are the oracle's; an update on top of it too."""
    rng = np.random.default_rng(ef + n)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ds, items = _mk(orc, hny, metric, vecs, draw_levels(n, M, seed=2))
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,lvM", [(0, 3000, 300, 80, 160, 64, 4), (1, 2000, 256, 100, 100, 120, 3),
                                                       (3, 2500, 128, 128, 256, 48, 4), (0, 1500, 200, 65, 65, 700, 2)])
def test_m_beyond_64_equals_oracle(orc, hny, monkeypatch, metric, n, dim, M, M0, ef, lvM):
    """M > 64 (the reference's const generics take any pair, writer.rs:215-220): an item above level 0 selects up
    to M neighbours, which are the entry points of its walk one layer down — more than one wave's lanes — and the
    upper layers' lists hold more than 64 links.  Levels are drawn as for a small M so that the upper layers
    are populated.  Fresh build, an update, a search, and the build once more through the heap walk."""
    rng = np.random.default_rng(M + n)
    vecs = {i: rng.uniform(-1, 1, dim).astype(np.float32) for i in range(n)}

    def mk(levels):
        ids = np.array(sorted(vecs), np.uint32)
        return orc.Dataset.from_f32(metric, np.stack([vecs[int(i)] for i in ids]), levels, ids)
    ds = mk(draw_levels(n, lvM, seed=7))
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    kw = dict(batch_frac=0.5, batch_max=512)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g, o)
    assert g.n_links_added == o.n_links_added and g.n_evals_walk == o.n_evals_walk
    d = o.as_dict()
    if M >= 80:
        assert max(len(v) for (i, l), v in d.items() if l >= 1) > 64  # upper-layer lists beyond one wave's lanes
    to_delete = sorted(rng.choice(n, n // 10, replace=False).tolist())
    for i in to_delete:
        del vecs[i]
    to_insert = list(range(n, n + n // 5))
    for i in to_insert:
        vecs[i] = rng.uniform(-1, 1, dim).astype(np.float32)
    lv = draw_levels(len(to_insert), lvM, seed=8)
    ds2 = mk(np.zeros(len(vecs), np.uint8))
    items2 = hny.ItemSet(metric, dim, ds2.ids, ds2.codes, ds2.headers, lv)
    o2 = orc.build_incremental(ds2, o, to_insert, lv, to_delete, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, **kw)
    g2 = hny.build_incremental(items2, g, to_insert, to_delete, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g2, o2)
    qs = rng.uniform(-1, 1, (50, dim)).astype(np.float32)
    qc = orc.encode_vectors(metric, qs)
    qh = orc.make_headers(metric, dim, qc)
    with hny.Builder(items2, prev=g2, load=True, M=M, M0=M0, ef_construction=ef) as b:
        got = b.search_knn(qc, qh, k=10, ef_search=80)
    want = orc.search(ds2, g2, qc, qh, k=10, ef_search=80, order=orc.ORDER_WAVE, threads=8)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
    monkeypatch.setenv("HNY_POOL_FORCE_RETRY", "2")
    g3 = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g3, o)


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,flat", [(3, 20000, 1024, 16, 32, 64, False), (0, 12000, 128, 16, 32, 64, False),
                                                        (3, 60, 512, 8, 16, 16, True), (4, 64, 256, 8, 16, 3, True),
                                                        (1, 70, 100, 8, 16, 16, True), (3, 6000, 256, 12, 24, 40, False)])
def test_one_chunk_register_beam_equals_oracle(orc, hny, monkeypatch, metric, n, dim, M, M0, ef, flat):
    """Build walks on rows of at most 512 B whose result sets never exceed 64 entries keep the beam in ONE 64-entry
    register chunk (k_walk<.., RC = 1>, WalkArgs.rb_one; round 5).  ef = 64 fills the chunk to its last lane; `flat`
    puts every item on level 0, so all n items are entry points and res starts above ef (it then only grows: 60 and
    64 entry points still fit, 70 take the two-chunk kernel).  Same graph, same counters as the oracle and as the
    two-chunk kernel (HNY_RB_ONE=0)."""
    rng = np.random.default_rng(5 * n + dim)
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    levels = np.zeros(n, np.uint8) if flat else draw_levels(n, M, seed=n)
    ds, items = _mk(orc, hny, metric, vecs, levels)
    kw = dict(batch_frac=0.5, batch_max=4096)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, threads=8, **kw)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g, o)
    assert (g.n_links_added, g.n_evals_walk) == (o.n_links_added, o.n_evals_walk)
    monkeypatch.setenv("HNY_RB_ONE", "0")
    g2 = hny.build(items, M=M, M0=M0, ef_construction=ef, **kw)
    _same_graph(g2, o)
    assert (g2.n_evals_walk, g2.n_evals_prune, g2.n_evals_apply) == (g.n_evals_walk, g.n_evals_prune, g.n_evals_apply)
    monkeypatch.setenv("HNY_RB_ONE", "1")
    monkeypatch.setenv("HNY_POOL_FORCE_RETRY", "3")  # and with every third member through the heap walk
    _same_graph(hny.build(items, M=M, M0=M0, ef_construction=ef, **kw), o)


def test_export_arrays_prepared_during_the_build_and_recycled_on_request(orc, hny):
    """hny_set_graph_cache (opt-in): hny_graph_free keeps the released export arrays for the next export; off (the
    default) nothing is kept.  With and without it, on a builder whose export is large enough for the helper thread
    that prepares the arrays during the build (>= 8 MB), repeated builds return the oracle's records — also when a
    graph outlives the next build, when the build is reset before its finish(), and for a smaller index afterwards
    (a cached array is only taken when it is not more than twice what is needed)."""
    rng = np.random.default_rng(4)
    n, dim = 70000, 32
    cent = rng.uniform(-1, 1, (32, dim)).astype(np.float32)
    vecs = (cent[rng.integers(0, 32, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    ds, items = _mk(orc, hny, 1, vecs, draw_levels(n, 16, seed=3))
    kw = dict(M=16, M0=32, batch_frac=1.0, batch_max=8192)
    o = orc.build(ds, ef=32, order=orc.ORDER_WAVE, threads=8, **kw)
    small_ds, small_items = _mk(orc, hny, 1, vecs[:9000], draw_levels(9000, 16, seed=3))
    small_o = orc.build(small_ds, ef=32, order=orc.ORDER_WAVE, threads=8, **kw)
    for cache in (1 << 30, 0):
        hny.set_graph_cache(cache)
        try:
            with hny.Builder(items, ef_construction=32, **kw) as b:
                b.run()
                g1 = b.finish()
                b.reset()
                b.next_batch()  # a build that is abandoned after its first batch was handed out ...
                b.reset()       # ... leaves its prepared arrays to the next one
                b.run()
                g2 = b.finish()
                _same_graph(g1, o)  # g1 is still alive and untouched
                _same_graph(g2, o)
                del g1
                b.reset()
                b.run()
                _same_graph(b.finish(), o)
            del g2
            _same_graph(hny.build(small_items, ef_construction=32, **kw), small_o)
        finally:
            hny.set_graph_cache(0)


def test_m0_limits_are_refused_loudly(orc, hny):
    """include/hannoy_amd.h: M <= M0 <= 1024 (strict mode too — fresh builds since round 4, updates since round 5):
    HNY_ERR_UNSUPPORTED on a machine WITH a GPU too (no silent clamp)."""
    v = np.random.default_rng(1).uniform(-1, 1, (500, 32)).astype(np.float32)
    items = hny.ItemSet.from_f32(hny.COSINE, v)
    for kw in (dict(M=16, M0=1025), dict(M=1025, M0=1025)):
        with pytest.raises(hny.HannoyError) as e:
            hny.build(items, ef_construction=32, **kw)
        assert e.value.code == -5
    with pytest.raises(hny.HannoyError) as e:  # unknown schedule bits are refused, not ignored
        hny.build(items, M=16, M0=32, ef_construction=32, schedule=8)
    assert e.value.code == -1


@pytest.mark.parametrize("metric,n,dim,M,M0,ef,bmax", [(1, 2500, 40, 16, 768, 120, 1), (0, 2500, 33, 16, 768, 100, 64),
                                                     (2, 2000, 24, 8, 96, 90, 1)])
def test_strict_mode_updates_of_lists_beyond_64_slots_equal_x86_oracle(orc, hny, metric, n, dim, M, M0, ef, bmax):
    """The reference's fuzz configuration — build::<16, 768> WITH add / delete rounds (src/tests/fuzz.rs:86-87,143) —
    in the reference's own arithmetic: strict mode (x86 summation order), sequential (batch_max = 1) and batched.
    Round 4 refused a strict-mode update of lists beyond 64 slots (k_fill_gaps keeps one lane per slot); round 5 runs
    fill_gaps_from_deleted on k_fill_gaps_wg with the one-wave prune (dist_rows in the x86 order) inside.  Three
    update rounds with deletions, overwrites and additions, each == the oracle in ORC_ORDER_X86, record for record."""
    rng = np.random.default_rng(11 * n + M0)
    vecs = {i: rng.uniform(-1, 1, dim).astype(np.float32) for i in range(n)}

    def mk(levels):
        ids = np.array(sorted(vecs), np.uint32)
        return orc.Dataset.from_f32(metric, np.stack([vecs[int(i)] for i in ids]), levels, ids)
    kw = dict(batch_frac=1.0 if bmax > 1 else 0.0, batch_max=bmax)
    kwo = dict(batch_frac=kw["batch_frac"], batch_max=0 if bmax == 1 else bmax)
    ds = mk(draw_levels(n, M, seed=3))
    items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
    o = orc.build(ds, M=M, M0=M0, ef=ef, order=orc.ORDER_X86, **kwo)
    g = hny.build(items, M=M, M0=M0, ef_construction=ef, x86_order=True, **kw)
    _same_graph(g, o)
    assert max(len(v) for (i, l), v in o.as_dict().items() if l == 0) > 64  # layer-0 lists beyond one wave's lanes
    nxt = n
    for rnd in range(3):
        live = sorted(vecs)
        to_delete = sorted(rng.choice(live, len(live) // 8, replace=False).tolist())
        for i in to_delete:
            del vecs[i]
        live = sorted(vecs)
        overwrite = sorted(rng.choice(live, len(live) // 20, replace=False).tolist())
        fresh = list(range(nxt, nxt + n // 6))
        nxt += n // 6
        for i in overwrite + fresh:
            vecs[i] = rng.uniform(-1, 1, dim).astype(np.float32)
        to_insert = sorted(overwrite + fresh)
        lv = draw_levels(len(to_insert), M, seed=20 + rnd)
        ds2 = mk(np.zeros(len(vecs), np.uint8))
        items2 = hny.ItemSet(metric, dim, ds2.ids, ds2.codes, ds2.headers, lv)
        o = orc.build_incremental(ds2, o, to_insert, lv, to_delete, M=M, M0=M0, ef=ef, order=orc.ORDER_X86, **kwo)
        g = hny.build_incremental(items2, g, to_insert, to_delete, M=M, M0=M0, ef_construction=ef, x86_order=True, **kw)
        _same_graph(g, o)
        assert g.n_links_added == o.n_links_added


def test_kat9_reference_snapshots_on_the_gpu(orc, hny):
    """KAT-9 through the C ABI: strict mode (x86 summation order: SSE path at dim 30) with batch_max = 1 is
    the reference run with one thread, so hny_build must return the 167 Links records of the reference's
    own 100 x 30 snapshot (src/tests/writer.rs:130-155) and hny_build_incremental the 192 records of the
    second one — levels drawn by the PRODUCT's restated StdRng (3 000 words into the stream, as the test's
    generator is), insertion order = what Rust's sort_unstable_by leaves (hny_rust_sort.h).  The default
    (batch-synchronous, wave order) build of the same inputs equals the oracle's."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat9_100x30.json")) as f:
        k = json.load(f)
    from tests.test_oracle_kat import kat9_inputs
    v1, lv1, upd, v2, lv2 = kat9_inputs(orc, k)
    rng = hny.StdRng.from_seed(bytes(k["seed"]))
    rng.drawn = k["n"] * k["dim"]                       # the 3 000 f32 the vectors took
    assert rng.draw_levels(k["M"], k["n"]).tolist() == lv1.tolist()
    rng.drawn += len(upd) * k["dim"]
    assert rng.draw_levels(k["M"], len(upd)).tolist() == lv2.tolist()

    def links_of(g):
        return [[int(i), int(l), nb] for (i, l), nb in sorted(g.as_dict().items())]
    kw = dict(M=k["M"], M0=k["M0"], ef_construction=k["ef_construction"])
    ds1, it1 = _mk(orc, hny, 1, v1, lv1)
    g1 = hny.build(it1, batch_max=1, x86_order=True, **kw)
    assert g1.entry_points.tolist() == k["fresh"]["entry_points"] and g1.max_level == k["fresh"]["max_level"]
    assert links_of(g1) == k["fresh"]["links"]
    ds2 = orc.Dataset.from_f32(1, v2, np.zeros(k["n"], np.uint8))
    it2 = hny.ItemSet(1, k["dim"], ds2.ids, ds2.codes, ds2.headers, lv2)
    g2 = hny.build_incremental(it2, g1, upd, [], batch_max=1, x86_order=True, **kw)
    assert g2.entry_points.tolist() == k["updated"]["entry_points"] and g2.max_level == k["updated"]["max_level"]
    assert links_of(g2) == k["updated"]["links"]
    # default schedule and wave order: same insertion order on both sides -> the oracle's graph
    o = orc.build(ds1, M=k["M"], M0=k["M0"], ef=k["ef_construction"], order=orc.ORDER_WAVE, batch_frac=1.0, batch_max=65536)
    _same_graph(hny.build(it1, **kw), o)
