"""CPU-side checks of the product library: it loads, exports every symbol the header declares,
refuses to compute without a GPU, and its host-side codecs / record encoders agree byte for byte
with the oracle and with hand-derived golden bytes.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hny():
    import hannoy_amd
    hannoy_amd.load_library()
    return hannoy_amd


def test_exports_every_declared_symbol(hny):
    hdr = open(os.path.join(ROOT, "include", "hannoy_amd.h")).read()
    declared = set(re.findall(r"\b(hny_[a-z_0-9]+)\s*\(", hdr)) - {"hny_kv_sink"}
    lib = hny.load_library()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hannoy_amd.h but not exported"
    from hannoy_amd import _capi
    assert declared == set(_capi.EXPORTED)


def test_abi_struct_sizes_library_vs_bindings(hny, tmp_path):
    """hny_abi_sizes: sizeof of every public struct as the library was compiled == (1) the ctypes
    declarations of hannoy_amd/_capi.py, (2) the header compiled as C99 by gcc, (3) the #[repr(C)]
    structs INTEGRATION.md §2 tells a Rust maintainer to write (parsed from the document, laid out by
    the C rules: a field missing there hands the library garbage — round 2's defect)."""
    import subprocess
    from hannoy_amd import _capi
    lib_sizes = hny.abi_sizes()
    assert list(lib_sizes) == [n for n, _ in _capi.ABI_STRUCTS]
    for name, cls in _capi.ABI_STRUCTS:
        assert C.sizeof(cls) == lib_sizes[name], name
    # (2) gcc's view of include/hannoy_amd.h
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "hannoy_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(hny_build_opts),sizeof(hny_items),sizeof(hny_graph),sizeof(hny_prev_graph),'
                   'sizeof(hny_batch),sizeof(hny_query_opts),sizeof(hny_lmdb_stat));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert got == list(lib_sizes.values())
    # (3) the Rust structs of INTEGRATION.md
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = {"HnyBuildOpts": "hny_build_opts", "HnyItems": "hny_items", "HnyGraph": "hny_graph",
            "HnyPrevGraph": "hny_prev_graph"}

    def size_align(ty):
        ty = ty.strip()
        if ty in ("i32", "u32", "f32", "c_int"):
            return 4, 4
        if ty in ("u8", "i8"):
            return 1, 1
        if ty in ("u16", "i16"):
            return 2, 2
        if ty in ("u64", "i64", "f64", "usize", "isize") or ty.startswith("*") or ty.startswith("Option<extern"):
            return 8, 8
        raise AssertionError(f"unknown Rust type in INTEGRATION.md: {ty!r}")

    for rname, cname in rust.items():
        m = re.search(r"#\[repr\(C\)\] pub struct %s \{(.*?)\}\n" % rname, doc, re.S)
        assert m, f"INTEGRATION.md lacks #[repr(C)] struct {rname}"
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        # split on the commas between fields (none of the field types nests a comma outside parentheses)
        fields, depth, cur = [], 0, ""
        for ch in body:
            depth += ch in "(<"
            depth -= ch in ")>" and depth > 0
            if ch == "," and depth == 0:
                fields.append(cur)
                cur = ""
            else:
                cur += ch
        fields.append(cur)
        off, maxal = 0, 1
        for f in fields:
            f = f.strip()
            if not f:
                continue
            assert f.startswith("pub ") and ":" in f, f
            sz, al = size_align(f.split(":", 1)[1].replace("->", "").strip() if "Option<extern" not in f
                                else "Option<extern")
            off = (off + al - 1) // al * al + sz
            maxal = max(maxal, al)
        total = (off + maxal - 1) // maxal * maxal
        assert total == lib_sizes[cname], f"INTEGRATION.md {rname}: {total} bytes, library {cname}: {lib_sizes[cname]}"


def test_no_cpu_fallback(hny):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    v = np.random.default_rng(0).uniform(-1, 1, (16, 8)).astype(np.float32)
    items = hny.ItemSet.from_f32(hny.COSINE, v)
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items)
    assert e.value.code == -6  # HNY_ERR_NO_DEVICE
    # the multi-GPU host (hny_multi.cpp) fails just as loudly, before it touches RCCL
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, n_gpus=2)
    assert e.value.code == -6
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, n_gpus=100)
    assert e.value.code == -1
    with pytest.raises(hny.HannoyError) as e:  # the resident form (hny_multi_builder_create)
        hny.MultiBuilder(items, n_gpus=2)
    assert e.value.code == -6


def test_argument_validation(hny):
    v = np.random.default_rng(0).uniform(-1, 1, (4, 8)).astype(np.float32)
    items = hny.ItemSet.from_f32(hny.EUCLIDEAN, v, ids=np.array([3, 2, 5, 9], np.uint32))
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items)
    assert e.value.code == -1 and "ascending" in str(e.value)
    items = hny.ItemSet.from_f32(hny.EUCLIDEAN, v)
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, M=16, M0=8)
    assert e.value.code == -1
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, M=16, M0=2048)  # beyond HNY_BIG_CAP = 1024
    assert e.value.code == -5
    with pytest.raises(hny.HannoyError) as e:
        hny.build(items, M=16, M0=128, x86_order=True)  # accepted since round 4 (strict mode, lists beyond 64 slots):
    assert e.value.code == -6                            # passes validation and stops at "no HIP device" here
    bad = hny.ItemSet(hny.EUCLIDEAN, 16, items.ids, items.codes, items.headers)  # stride < 16*4
    with pytest.raises(hny.HannoyError) as e:
        hny.build(bad)
    assert e.value.code == -4  # Error::InvalidVecDimension


@pytest.mark.parametrize("metric", range(7))
def test_codecs_match_oracle(hny, orc, metric):
    rng = np.random.default_rng(metric)
    for dim in (3, 20, 64, 100, 768):
        v = rng.uniform(-1, 1, (50, dim)).astype(np.float32)
        v[0, :] = 0.0
        v[1, 0] = -0.0
        v[2, 0] = np.inf
        v[3, 0] = np.nan
        codes, hdrs = hny.encode_vectors(metric, v)
        ocodes = orc.encode_vectors(metric, v)
        ohdrs = orc.make_headers(metric, dim, ocodes)
        assert codes.shape == ocodes.shape and np.array_equal(codes, ocodes)
        assert np.array_equal(hdrs, ohdrs)
        assert hny.vector_bytes(metric, dim) == orc.vector_bytes(metric, dim)
        assert hny.header_bytes(metric) == orc.header_bytes(metric)


def test_quantiser_goldens(hny, kat):
    for k in kat["kat6"]:
        metric = hny.HAMMING if k["codec"] == "binary" else hny.BQ_COSINE
        codes, _ = hny.encode_vectors(metric, np.array([k["input"]], np.float32))
        assert [format(b, "08b") for b in codes[0]] == k["bytes_bin"], k["source"]


def test_schedule_matches_oracle(hny, orc):
    lib = hny.load_library()
    for frac, bmax in ((0.02, 16384), (0.05, 64), (0.0, 1), (1.0, 7), (0.3, 0)):
        for n_done in (0, 1, 2, 10, 49, 50, 51, 999, 12345, 10 ** 6, 10 ** 9):
            assert lib.hny_batch_size(frac, bmax, n_done) == orc.batch_size(frac, bmax, n_done)


def _graph_struct_from_oracle(hny, og):
    from hannoy_amd import _capi
    g = _capi.GraphStruct()
    keep = [np.ascontiguousarray(og.rec_item, np.uint32), np.ascontiguousarray(og.rec_layer, np.uint8),
            np.ascontiguousarray(og.offsets, np.uint64),
            np.ascontiguousarray(og.nbrs if len(og.nbrs) else np.zeros(1), np.uint32),
            np.ascontiguousarray(og.entry_points, np.uint32)]
    g.n_records = len(og.rec_item)
    g.rec_item = keep[0].ctypes.data_as(C.POINTER(C.c_uint32))
    g.rec_layer = keep[1].ctypes.data_as(C.POINTER(C.c_uint8))
    g.rec_offset = keep[2].ctypes.data_as(C.POINTER(C.c_uint64))
    g.neighbours = keep[3].ctypes.data_as(C.POINTER(C.c_uint32))
    g.entry_points = keep[4].ctypes.data_as(C.POINTER(C.c_uint32))
    g.n_entry_points = len(og.entry_points)
    g.max_level = og.max_level
    return g, keep


def _encode_kv(hny, g, opts, items, with_items):
    from hannoy_amd import _capi
    out = []

    def sink(_c, k, kl, v, vl):
        out.append((bytes(k[:kl]), bytes(v[:vl])))
        return 0
    it = items.struct()
    rc = hny.load_library().hny_encode_kv(C.byref(g), C.byref(opts), C.byref(it), 0, with_items,
                                          _capi.KV_SINK(sink), None)
    assert rc == 0
    return out


def test_kv_records_kat1_bytes(hny, orc, kat):
    """Byte-exact Key/Links/Metadata/Version/Item records for the KAT-1 index."""
    k = kat["kat1"]
    v = np.array(k["vectors"], np.float32)
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, v, k["levels"])
    og = orc.build(ds, M=3, M0=3, ef=100)
    items = hny.ItemSet(hny.EUCLIDEAN, 2, ds.ids, ds.codes, ds.headers, ds.levels)
    opts = hny.make_opts(hny.EUCLIDEAN, 2, M=3, M0=3)
    g, _keep = _graph_struct_from_oracle(hny, og)
    recs = _encode_kv(hny, g, opts, items, 1)
    assert recs == orc.encode_kv(ds, og, 0, True)
    keys = [r[0] for r in recs]
    assert keys == sorted(keys)  # LMDB order: metadata, version, links by (item, layer), items
    # metadata.rs:28-48, hand-assembled
    roaring_items = bytes.fromhex("3a300000" "01000000" "0000" "0500" "10000000"
                                  "0000" "0100" "0200" "0300" "0400" "0500")
    meta = (b"euclidean\0" + (2).to_bytes(4, "big") + len(roaring_items).to_bytes(4, "big")
            + roaring_items + b"".join(int(e).to_bytes(4, "little") for e in (0, 2, 3)) + b"\x01")
    assert recs[0] == (bytes.fromhex("0000000000000000"), meta)
    assert recs[1] == (bytes.fromhex("0000000000000100"), bytes.fromhex("000000000000000100000003"))
    # Links(0, layer 0) = {1, 2}: node.rs:141-144 tag 1 + roaring
    assert recs[2] == (bytes.fromhex("0000020000000000"),
                       bytes.fromhex("01" "3a300000" "01000000" "0000" "0100" "10000000" "0100" "0200"))
    # Item 5 = [5.0, 0.0], header bias 0.0: node.rs:136-140
    assert recs[-1] == (bytes.fromhex("0000030000000500"),
                        b"\x00" + np.float32(0).tobytes() + np.array([5, 0], np.float32).tobytes())
    assert len(recs) == 2 + 9 + 6


def test_default_batch_cap_scales_with_the_index(hny):
    """batch_max = 0: the largest power of two <= n / 12, at least 65 536 (include/hannoy_amd.h)."""
    f = hny.default_batch_max
    assert [f(n) for n in (0, 1, 10_000, 1_000_000, 1_572_863, 1_572_864, 5_000_000, 10_000_000)] == \
        [65536, 65536, 65536, 65536, 65536, 131072, 262144, 524288]
    assert f(10 ** 12) == 1 << 21  # bounded: 2^21 members x 64 slots x 2 link ops < 2^29 sequence numbers


def test_roaring_serialisation_shapes(hny, orc):
    """RoaringFormatSpec facts: empty bitmap = 8 bytes; array container up to 4096 values, bitmap
    container (8 KiB) above; several 64Ki-containers; u32::MAX."""
    assert orc.roaring_serialize([]) == bytes.fromhex("3a30000000000000")
    ids = np.arange(0, 4096, dtype=np.uint32)
    assert len(orc.roaring_serialize(ids)) == 8 + 8 + 2 * 4096
    ids = np.arange(0, 4097, dtype=np.uint32)
    assert len(orc.roaring_serialize(ids)) == 8 + 8 + 8192
    ids = np.array([5, 70000, 4294967295], np.uint32)
    b = orc.roaring_serialize(ids)
    assert b == bytes.fromhex("3a300000" "03000000" "0000" "0000" "0100" "0000" "ffff" "0000"
                              "20000000" "22000000" "24000000" "0500" "7011" "ffff")
    # the product's encoder agrees on a large id set (metadata of a 100k-item index)
    big = np.unique(np.random.default_rng(1).integers(0, 300000, 100000).astype(np.uint32))
    ds_ids = big
    v = np.zeros((len(big), 4), np.float32)
    lv = np.zeros(len(big), np.uint8)
    lv[0] = 1
    items = hny.ItemSet.from_f32(hny.EUCLIDEAN, v, ids=ds_ids, levels=lv)

    class G:
        rec_item = np.zeros(0, np.uint32)
        rec_layer = np.zeros(0, np.uint8)
        offsets = np.zeros(1, np.uint64)
        nbrs = np.zeros(0, np.uint32)
        entry_points = big[:1]
        max_level = 1
    g, _keep = _graph_struct_from_oracle(hny, G)
    recs = _encode_kv(hny, g, hny.make_opts(hny.EUCLIDEAN, 4), items, 0)
    meta = recs[0][1]
    rsz = int.from_bytes(meta[14:18], "big")
    assert meta[18:18 + rsz] == orc.roaring_serialize(big)


def test_level_draws_match_oracle_rng(hny, orc):
    """hny_draw_levels == the oracle's restated StdRng::seed_from_u64 + WeightedIndex draws."""
    for seed, M in ((42, 16), (0, 3), (2 ** 63 + 5, 32), (7, 4)):
        assert np.array_equal(hny.draw_levels(seed, M, 50000), orc.draw_levels(M, 50000, seed_u64=seed))


def test_short_row_visited_table_arithmetic():
    """The arithmetic behind VisB (hny_kernels.hip), restated: slot id -> bijection of [0, 2^k) -> (bucket, remainder)
    by a multiply-shift division that must be exact, and back (the Reader's flush reconstructs ids from the table).
    Constants read from the sources, so a change there is checked here."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "hannoy_amd", "csrc", "hny_kernels.hip")).read()
    mul = int(re.search(r"#define HNY_VISB_MUL (0x[0-9A-Fa-f]+)u", src).group(1), 16)
    inv = int(re.search(r"#define HNY_VISB_INV (0x[0-9A-Fa-f]+)u", src).group(1), 16)
    assert mul & 1 and (mul * inv) % (1 << 32) == 1
    rng = np.random.default_rng(0)
    for n, nb in ((9000, 64), (5_000_000, 448), (10_000_000, 896), (1 << 24, 448), (1 << 25, 896), (3, 64), (1 << 28, 4096)):
        k = 1
        while k < 28 and (1 << k) < max(n, 2):
            k += 1
        smask = (1 << k) - 1
        lg = 0
        while (2 << lg) <= nb:
            lg += 1
        shift = 31 + lg
        magic = (1 << shift) // nb + 1
        assert magic < (1 << 32)
        ids = np.unique(np.concatenate([rng.integers(0, n, 200_000), [0, n - 1]])).astype(np.uint64)
        sid = (ids * mul) & smask
        q = (sid * magic) >> shift                     # what the kernel computes (64-bit product)
        assert np.array_equal(q, sid // nb)            # == sid div nb, exactly
        bk = sid - q * nb
        assert bk.max() < nb
        back = (((q * nb + bk) & 0xFFFFFFFF) * inv) & smask
        assert np.array_equal(back, ids)               # bucket and remainder give the id back
        assert len(np.unique(sid)) == len(ids)         # a bijection on the ids
        if (smask // nb) + 1 < 65535:
            assert (q + 1).max() <= 65535              # remainder + 1 fits 16 bits (0 = empty)
