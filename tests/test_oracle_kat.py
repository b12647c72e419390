"""Pins the CPU oracle against the reference's own golden vectors."""
import numpy as np
import pytest

METRICS = {"cosine": 0, "euclidean": 1, "manhattan": 2, "hamming": 3}


def _links_of(g):
    return [[int(i), int(l), nb] for (i, l), nb in sorted(g.as_dict().items())]


def test_kat1_fresh_build(kat, orc):
    k = kat["kat1"]
    ds = orc.Dataset.from_f32(METRICS[k["metric"]], np.array(k["vectors"], np.float32),
                              k["levels"], np.array(k["ids"], np.uint32))
    for order in (orc.ORDER_X86, orc.ORDER_WAVE):
        g = orc.build(ds, M=k["M"], M0=k["M0"], ef=k["ef_construction"], alpha=k["alpha"],
                      order=order)
        assert g.entry_points.tolist() == k["entry_points"]
        assert g.max_level == k["max_level"]
        assert _links_of(g) == k["links"]
    # batch-synchronous schedule with batch size 1 == sequential
    g = orc.build(ds, M=3, M0=3, ef=100, batch_frac=0.0, batch_max=1)
    assert _links_of(g) == k["links"]
    # the in-memory list of node 0 holds duplicates before RoaringBitmap dedup (hnsw.rs:521 TODO)
    raw = g.raw_dict()
    assert sorted(raw[(0, 0)][0]) != sorted(set(raw[(0, 0)][0])) or len(raw[(0, 0)][0]) >= 2


def test_kat5_single_item(kat, orc):
    for k in kat["kat5"]:
        ds = orc.Dataset.from_f32(1, np.array([k["vector"]], np.float32), [k["level"]],
                                  np.array([k["id"]], np.uint32))
        g = orc.build(ds, M=k["M"], M0=k["M0"])
        assert g.entry_points.tolist() == k["entry_points"]
        assert g.max_level == k["max_level"]
        assert _links_of(g) == k["links"]
        assert ds.headers.view(np.float32)[0, 0] == k["header_f32"]
    # Cosine zero vector: header norm 0.0 (writer.rs:562-570)
    ds = orc.Dataset.from_f32(0, np.zeros((1, 3), np.float32), [1])
    assert ds.headers.view(np.float32)[0, 0] == 0.0


def test_kat6_quantisers(kat, orc):
    for k in kat["kat6"]:
        metric = orc.HAMMING if k["codec"] == "binary" else orc.BQ_COSINE
        v = np.array([k["input"]], np.float32)
        code = orc.encode_vectors(metric, v)[0]
        assert [format(b, "08b") for b in code] == k["bytes_bin"], k["source"]
    # +0.0 / -0.0 / NaN / inf handling (binary.rs:87-89, binary_quantized.rs:86)
    special = np.array([[0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1e-45, -1e-45]], np.float32)
    special[0, 5] = np.frombuffer(np.uint32(0xFFC00000).tobytes(), np.float32)[0]
    assert format(orc.encode_vectors(orc.HAMMING, special)[0][0], "08b") == "01010100"
    assert format(orc.encode_vectors(orc.BQ_EUCLIDEAN, special)[0][0], "08b") == "01010101"


def test_kat7_simd_equals_scalar(kat, orc):
    v1, v2 = np.array(kat["kat7"]["v1"], np.float32), np.array(kat["kat7"]["v2"], np.float32)
    assert len(v1) == 70
    scalar_dot = np.float32(0)
    scalar_l2 = np.float32(0)
    for a, b in zip(v1, v2):
        scalar_dot = np.float32(scalar_dot + np.float32(a * b))
        scalar_l2 = np.float32(scalar_l2 + np.float32((a - b) * (a - b)))
    assert orc.dot(orc.ORDER_X86, v1, v2) == scalar_dot
    assert orc.sqeuclid(orc.ORDER_X86, v1, v2) == scalar_l2
    assert orc.dot_emulated(v1, v2) == scalar_dot
    assert orc.dot(orc.ORDER_WAVE, v1, v2) == scalar_dot  # exact in f32 → order-insensitive
    assert orc.sqeuclid(orc.ORDER_WAVE, v1, v2) == scalar_l2


def test_avx_intrinsics_equal_emulation(orc):
    rng = np.random.default_rng(7)
    for dim in (1, 3, 15, 16, 17, 31, 32, 33, 63, 64, 100, 128, 767, 768, 769, 1536):
        for _ in range(20):
            a = rng.uniform(-1, 1, dim).astype(np.float32)
            b = rng.uniform(-1, 1, dim).astype(np.float32)
            assert orc.dot(orc.ORDER_X86, a, b).tobytes() == orc.dot_emulated(a, b).tobytes()
            assert (orc.sqeuclid(orc.ORDER_X86, a, b).tobytes()
                    == orc.sqeuclid_emulated(a, b).tobytes())


def test_wave_order_within_tolerance(orc):
    """north_star: f32 distances within 1e-5 relative. Checked on pq / squared-L2 themselves."""
    rng = np.random.default_rng(8)
    for dim in (3, 100, 128, 768, 1024, 1536):
        for _ in range(50):
            a = rng.uniform(-1, 1, dim).astype(np.float32)
            b = rng.uniform(-1, 1, dim).astype(np.float32)
            ref = np.dot(a.astype(np.float64), b.astype(np.float64))
            scale = np.dot(np.abs(a).astype(np.float64), np.abs(b).astype(np.float64))
            for order in (orc.ORDER_X86, orc.ORDER_WAVE):
                assert abs(float(orc.dot(order, a, b)) - ref) <= 1e-5 * scale
            l2 = float(np.sum((a.astype(np.float64) - b.astype(np.float64)) ** 2))
            for order in (orc.ORDER_X86, orc.ORDER_WAVE):
                assert abs(float(orc.sqeuclid(order, a, b)) - l2) <= 1e-5 * l2


def test_kat8_python_hamming(kat, orc):
    k = kat["kat8"]
    ds = orc.Dataset.from_f32(orc.HAMMING, np.array(k["vectors"], np.float32), [0, 0, 0])
    assert ds.codes.shape[1] == 8 and ds.headers.shape[1] == 8
    g = orc.build(ds, M=k["M"], M0=k["M0"], ef=k["ef_construction"])
    q = orc.encode_vectors(orc.HAMMING, np.array([k["query"]], np.float32))
    qh = orc.make_headers(orc.HAMMING, 3, q)
    ids, dists, counts = orc.search(ds, g, q, qh, k=k["k"])
    assert counts[0] == k["n_hits"]
    assert [int(ids[0, 0]), float(dists[0, 0])] == k["first_hit"]
    # distance = popcount / padded dims (64), hamming.rs:44-47
    assert dists[0, 1] == np.float32(2 / 64)


def test_key_codec(kat, orc):
    for index, mode, item, layer, hx in kat["keys"]["cases"]:
        assert orc.encode_key(index, mode, item, layer).hex() == hx
    # ordering: Metadata < Updated < Links < Item (node_id.rs:131-136)
    keys = [orc.encode_key(0, m, 0, 0) for m in (0, 1, 2, 3)]
    assert keys == sorted(keys)


def test_level_probas(orc):
    # hnsw.rs:94-110: P(l) = M^-l (1 - 1/M), truncated below 1e-9
    p = orc.level_probas(16)
    assert len(p) == 8
    assert abs(p[0] - 15 / 16) < 1e-6 and abs(p[1] - 15 / 256) < 1e-6
    assert len(orc.level_probas(32)) == 6


def test_kat2_3_4_incremental_builds(kat, orc):
    """Incremental path (prepare_levels_and_entry_points deletion branch, on-disk links in
    get_neighbours, fill_gaps_from_deleted) against the reference's snapshots."""
    k1 = kat["kat1"]
    v = np.array(k1["vectors"], np.float32)
    ds1 = orc.Dataset.from_f32(orc.EUCLIDEAN, v, k1["levels"])
    g1 = orc.build(ds1, M=3, M0=3, ef=100)
    # KAT-2: overwrite item 3 with [6, 0]
    k2 = kat["kat2"]
    v2 = v.copy()
    v2[k2["overwrite"]["id"]] = k2["overwrite"]["vector"]
    ds2 = orc.Dataset.from_f32(orc.EUCLIDEAN, v2, np.zeros(6, np.uint8))
    for lv in k2["insert_levels_any_of"]:
        g2 = orc.build_incremental(ds2, g1, k2["to_insert"], lv, k2["to_delete"], M=3, M0=3, ef=100)
        assert g2.entry_points.tolist() == k2["entry_points"] and g2.max_level == k2["max_level"]
        assert _links_of(g2) == k2["links"]
    # KAT-3: delete item 3 from the KAT-1 DB
    k3 = kat["kat3"]
    keep = [0, 1, 2, 4, 5]
    ds3 = orc.Dataset.from_f32(orc.EUCLIDEAN, v[keep], np.zeros(5, np.uint8), np.array(keep, np.uint32))
    g3 = orc.build_incremental(ds3, g1, [], [], k3["to_delete"], M=3, M0=3, ef=100)
    assert g3.entry_points.tolist() == k3["entry_points"] and g3.max_level == k3["max_level"]
    assert _links_of(g3) == k3["links"]  # includes the reference's self-loops
    # KAT-4: then delete item 1
    k4 = kat["kat4"]
    keep = [0, 2, 4, 5]
    ds4 = orc.Dataset.from_f32(orc.EUCLIDEAN, v[keep], np.zeros(4, np.uint8), np.array(keep, np.uint32))
    g4 = orc.build_incremental(ds4, g3, [], [], k4["to_delete"], M=3, M0=3, ef=100)
    assert g4.entry_points.tolist() == k4["entry_points"] and g4.max_level == k4["max_level"]
    assert _links_of(g4) == k4["links"]


def test_level_rng_reproduces_reference_draws(kat, orc):
    """[3P] rand 0.8.5 StdRng (ChaCha12) + WeightedIndex<f32>, restated: the first draws from
    StdRng::from_seed([42; 32]) (tests/mod.rs:145-147) with M = 3 must be the levels the KAT-1
    snapshot implies, and the first draw the level KAT-5 implies."""
    lv = orc.draw_levels(3, 7, seed32=[42] * 32)
    assert lv[:6].tolist() == kat["kat1"]["levels"]
    assert lv[0] == kat["kat5"][0]["level"]
    # distribution sanity for the python binding's seed_from_u64(42) (python.rs:261), M = 16
    lv = orc.draw_levels(16, 200000, seed_u64=42)
    frac0 = np.mean(lv == 0)
    assert abs(frac0 - 15 / 16) < 0.003 and lv.max() <= 7


@pytest.fixture(scope="module")
def kat9():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat9_100x30.json")) as f:
        return json.load(f)


def kat9_inputs(orc, k):
    """The inputs of src/tests/writer.rs:130-155, regenerated: 100 x 30 `rng.gen::<f32>()` from
    StdRng::from_seed([42; 32]), the 100 level draws of the first build (the same generator goes on), the
    50 x 30 replacement vectors, the 50 level draws of the second build — every printed component and every
    level the snapshots imply is checked."""
    seed, n, dim = bytes(k["seed"]), k["n"], k["dim"]
    v1 = orc.gen_f32(seed, 0, n * dim).reshape(n, dim)
    lv1 = orc.draw_levels_skip(seed, n * dim, k["M"], n)
    upd = np.array(k["updated_ids"], np.uint32)
    v2 = v1.copy()
    v2[upd] = orc.gen_f32(seed, n * dim + n, len(upd) * dim).reshape(len(upd), dim)
    lv2 = orc.draw_levels_skip(seed, n * dim + n + len(upd) * dim, k["M"], len(upd))
    for vecs, snap in ((v1, k["fresh"]), (v2, k["updated"])):
        for i in range(n):
            printed = snap["printed"][str(i)]
            assert ["%.4f" % x for x in vecs[i, :len(printed)]] == printed, i
    return v1, lv1, upd, v2, lv2


def test_kat9_rng_reproduces_the_snapshot_inputs(orc, kat9):
    """[3P] rand 0.8.5 `Standard` f32 (24 bits x 2^-24) on the restated ChaCha12: the 1 000 + 1 000 vector
    components the two snapshots print, and the level of all 100 items of the first build (an item's level
    = its number of Links records - 1) — a 100-draw pin of WeightedIndex<f32> at stream position 3 000."""
    v1, lv1, upd, v2, lv2 = kat9_inputs(orc, kat9)
    n_rec = np.zeros(kat9["n"], np.int64)
    for item, layer, _ in kat9["fresh"]["links"]:
        n_rec[item] = max(n_rec[item], layer + 1)
    assert (lv1.astype(np.int64) + 1 == n_rec).all()
    assert lv1.max() == kat9["fresh"]["max_level"] == 6
    # after the update an item owns max(old, new) + 1 records: new levels never exceed what the second dump shows
    n_rec2 = np.zeros(kat9["n"], np.int64)
    for item, layer, _ in kat9["updated"]["links"]:
        n_rec2[item] = max(n_rec2[item], layer + 1)
    assert (np.maximum(lv1[upd], lv2).astype(np.int64) + 1 == n_rec2[upd]).all()


def test_kat9_100x30_snapshots(orc, kat9):
    """KAT-9, the reference's largest live graph fixture: 100 x 30-d Euclidean (the SSE summation path,
    16 <= dim < 32, src/spaces/simple.rs:19-47), M = M0 = 3, seven levels, one thread — the oracle in the
    x86 order reproduces the fresh build's 167 Links records and, after the 50-item overwrite, all 192
    records of the second snapshot, given the order Rust's sort_unstable_by (hnsw.rs:268, ipnsort) leaves
    equal levels in.  With ids ascending inside a level (a stable sort) 93 of the 167 records differ: the
    order is part of the reference's result."""
    k = kat9
    v1, lv1, upd, v2, lv2 = kat9_inputs(orc, k)
    kw = dict(M=k["M"], M0=k["M0"], ef=k["ef_construction"], order=orc.ORDER_X86)
    ds1 = orc.Dataset.from_f32(orc.EUCLIDEAN, v1, lv1)
    g1 = orc.build(ds1, level_sort=orc.LEVEL_SORT_RUST, **kw)
    assert g1.entry_points.tolist() == k["fresh"]["entry_points"] and g1.max_level == k["fresh"]["max_level"]
    assert _links_of(g1) == k["fresh"]["links"]
    ds2 = orc.Dataset.from_f32(orc.EUCLIDEAN, v2, np.zeros(k["n"], np.uint8))
    g2 = orc.build_incremental(ds2, g1, upd, lv2, [], level_sort=orc.LEVEL_SORT_RUST, **kw)
    assert g2.entry_points.tolist() == k["updated"]["entry_points"] and g2.max_level == k["updated"]["max_level"]
    assert _links_of(g2) == k["updated"]["links"]
    # the control: same inputs, ties by ascending id
    g1s = orc.build(ds1, level_sort=orc.LEVEL_SORT_BY_ID, **kw)
    assert sum(a != b for a, b in zip(_links_of(g1s), k["fresh"]["links"])) > 50
    # and the first pairs of the order itself: the entry point first, then levels descending
    ids, lv = orc.rust_sort_levels(np.arange(k["n"]), lv1)
    assert ids[0] == 65 and (np.diff(lv.astype(np.int64)) <= 0).all() and sorted(ids.tolist()) == list(range(k["n"]))
    assert ids.tolist() != sorted(ids.tolist(), key=lambda i: (-int(lv1[i]), i))


def test_wave_order_avx_form_equals_its_scalar_statement(orc):
    """The oracle's WAVE-order reduction has an AVX2 form (8 lanes of the group per register, round 5: the 10M-item
    parity run spends its time there) next to the scalar statement of the order.  Both must give the same bits for
    every op, over dims that exercise every lanes-per-row / chunks-per-lane shape and the zero padding, including
    signed zeros, denormals and large magnitudes."""
    import ctypes as C
    L = orc.lib()
    L.orc_wave_reduce_both.argtypes = [C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(12)
    dims = list(range(1, 70)) + [96, 100, 127, 128, 129, 200, 255, 256, 257, 333, 511, 512, 513, 767, 768, 769, 1000,
                                 1024, 1025, 1536, 2047, 2048, 3000, 4095, 4096]
    out = np.zeros(2, np.float32)
    for dim in dims:
        for scale in (1.0, 1e-20, 1e18):
            a = (rng.standard_normal(dim) * scale).astype(np.float32)
            b = (rng.standard_normal(dim) * scale).astype(np.float32)
            if dim > 3:
                a[rng.integers(0, dim)] = -0.0
                b[rng.integers(0, dim)] = 0.0
            for op in (0, 1, 2):
                L.orc_wave_reduce_both(op, dim, a.ctypes.data, b.ctypes.data, out.ctypes.data)
                assert out[:1].view(np.uint32)[0] == out[1:].view(np.uint32)[0], (dim, scale, op, out)
