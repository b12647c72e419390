"""LMDB writeback: hny_lmdb_writer_* / hny_lmdb_* on the CPU.

No LMDB exists in this image, so the writer is pinned three ways: (1) against bytes assembled by hand
from the mdb.c layout, (2) against `parse_lmdb` below — an independent pure-Python reader of that
layout that shares no code with the C++ —, (3) through the C++ reader (mdb_get / cursor walk
restated).  "Parity unpinned" against a real liblmdb.
"""
import os
import struct

import numpy as np
import pytest

import hannoy_amd as H
from hannoy_amd import _capi as capi

P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
INVALID = 0xFFFFFFFFFFFFFFFF


def parse_db(b):
    pad, flags, depth, branch, leaf, ovf, entries, root = struct.unpack("<IHHQQQQQ", b)
    return dict(pad=pad, flags=flags, depth=depth, branch=branch, leaf=leaf, overflow=ovf,
                entries=entries, root=root)


def parse_lmdb(path, name=None):
    """Independent reader: returns (records in file order of an in-order tree walk, stats)."""
    buf = open(path, "rb").read()
    metas = []
    psize = struct.unpack_from("<I", buf, 16 + 24)[0]
    for m in range(2):
        p = m * psize
        pgno, _pad, flags = struct.unpack_from("<QHH", buf, p)
        assert pgno == m and flags == P_META
        magic, ver, addr, mapsize = struct.unpack_from("<IIQQ", buf, p + 16)
        assert magic == 0xBEEFC0DE and ver == 1 and addr == 0
        free, main = parse_db(buf[p + 40:p + 88]), parse_db(buf[p + 88:p + 136])
        last, txnid = struct.unpack_from("<QQ", buf, p + 136)
        assert free["pad"] == psize and free["flags"] == 8 and free["root"] == INVALID and free["entries"] == 0
        metas.append(dict(main=main, last=last, txnid=txnid, mapsize=mapsize))
    assert metas[0]["txnid"] == 0 and metas[0]["last"] == 1 and metas[0]["main"]["root"] == INVALID
    meta = max(metas, key=lambda m: m["txnid"])
    assert meta["txnid"] == 1 and len(buf) == (meta["last"] + 1) * psize and meta["mapsize"] >= len(buf)
    seen = {"branch": 0, "leaf": 0, "overflow": 0, "pages": set()}

    def walk(pg, depth, out, depths):
        base = pg * psize
        pgno, _pad, flags, lower, upper = struct.unpack_from("<QHHHH", buf, base)
        assert pgno == pg and pg not in seen["pages"]
        seen["pages"].add(pg)
        n = (lower - 16) // 2
        slots = struct.unpack_from(f"<{n}H", buf, base + 16)
        assert 16 <= lower <= upper <= psize and all(upper <= s < psize and s % 2 == 0 for s in slots)
        if flags == P_BRANCH:
            seen["branch"] += 1
            assert n >= 2
            for i, s in enumerate(slots):
                lo, hi, fl, ks = struct.unpack_from("<HHHH", buf, base + s)
                assert (ks == 0) == (i == 0)
                child = lo | hi << 16 | fl << 32
                sep = buf[base + s + 8:base + s + 8 + ks]
                first = len(out)
                walk(child, depth + 1, out, depths)
                if i:  # separator = lower bound of its subtree, above everything to its left
                    assert out[first][0] >= sep and out[first - 1][0] < sep
                    assert out[first][0] == sep  # a bulk load uses the first key itself
        else:
            assert flags == P_LEAF and n >= 1
            seen["leaf"] += 1
            depths.add(depth)
            for s in slots:
                lo, hi, fl, ks = struct.unpack_from("<HHHH", buf, base + s)
                k = buf[base + s + 8:base + s + 8 + ks]
                size = lo | hi << 16
                if fl & 1:
                    ov = struct.unpack_from("<Q", buf, base + s + 8 + ks)[0]
                    opg, _p, ofl, npages = struct.unpack_from("<QHHI", buf, ov * psize)
                    assert opg == ov and ofl == P_OVERFLOW and npages == (15 + size) // psize + 1
                    for q in range(ov, ov + npages):
                        assert q not in seen["pages"]
                        seen["pages"].add(q)
                    seen["overflow"] += npages
                    v = buf[ov * psize + 16:ov * psize + 16 + size]
                    assert 8 + ks + size > ((psize - 16) // 2 & ~1) - 2
                else:
                    v = buf[base + s + 8 + ks:base + s + 8 + ks + size]
                    assert 8 + ks + size <= ((psize - 16) // 2 & ~1) - 2
                out.append((k, v, fl))

    def tree(db):
        out, depths = [], set()
        before = dict(seen, pages=None)
        if db["root"] != INVALID:
            walk(db["root"], 1, out, depths)
        assert depths <= {db["depth"]} and (db["depth"] == 0) == (db["root"] == INVALID)
        assert seen["branch"] - before["branch"] == db["branch"] and seen["leaf"] - before["leaf"] == db["leaf"]
        assert seen["overflow"] - before["overflow"] == db["overflow"] and len(out) == db["entries"]
        assert all(a[0] < b[0] for a, b in zip(out, out[1:]))
        return out

    main = tree(meta["main"])
    if name is not None:
        assert [(k, fl) for k, _, fl in main] == [(name.encode(), 2)]
        sub = parse_db(main[0][1])
        recs = tree(sub)
        st = sub
    else:
        recs, st = main, meta["main"]
    assert seen["pages"] == set(range(2, meta["last"] + 1))  # every page reachable exactly once
    return [(k, v) for k, v, _ in recs], dict(st, page_size=psize, last=meta["last"])


def write(path, recs, **kw):
    with capi.LmdbWriter(path, **kw) as w:
        for k, v in recs:
            w.put(k, v)


def test_golden_bytes_two_records(tmp_path):
    """the whole file of a two-record environment, assembled by hand from the mdb.c layout"""
    p = str(tmp_path / "data.mdb")
    k0, k1 = bytes(8), bytes(7) + b"\x01"
    write(p, [(k0, b"abc"), (k1, b"")])
    ps = 4096
    db_free = struct.pack("<IHHQQQQQ", ps, 8, 0, 0, 0, 0, 0, INVALID)
    db_empty = struct.pack("<IHHQQQQQ", 0, 0, 0, 0, 0, 0, 0, INVALID)
    db_main = struct.pack("<IHHQQQQQ", 0, 0, 1, 0, 1, 0, 2, 2)

    def meta(pgno, main, last, txnid):
        m = struct.pack("<QHHHH", pgno, 0, P_META, 0, 0)
        m += struct.pack("<IIQQ", 0xBEEFC0DE, 1, 0, 3 * ps) + db_free + main + struct.pack("<QQ", last, txnid)
        return m.ljust(ps, b"\0")
    leaf = bytearray(ps)
    leaf[0:16] = struct.pack("<QHHHH", 2, 0, P_LEAF, 16 + 4, 4060)
    leaf[16:20] = struct.pack("<HH", 4076, 4060)
    leaf[4076:4096] = struct.pack("<HHHH", 3, 0, 0, 8) + k0 + b"abc\0"
    leaf[4060:4076] = struct.pack("<HHHH", 0, 0, 0, 8) + k1
    expected = meta(0, db_empty, 1, 0) + meta(1, db_main, 2, 1) + bytes(leaf)
    assert open(p, "rb").read() == expected


def test_golden_overflow_and_named_db(tmp_path):
    """a 3 077-byte Item value (C2's size) goes to a one-page overflow run; named DB = F_SUBDATA node"""
    p = str(tmp_path / "data.mdb")
    k = H.api.key(0, 3, 7)
    v = bytes(range(256)) * 12 + b"\x05" * 5
    assert len(v) == 3077
    write(p, [(k, v)], name="vectors")
    buf = open(p, "rb").read()
    assert len(buf) == 5 * 4096  # metas, overflow page 2, leaf 3, main-DB leaf 4
    assert buf[2 * 4096:2 * 4096 + 16] == struct.pack("<QHHI", 2, 0, P_OVERFLOW, 1)
    assert buf[2 * 4096 + 16:2 * 4096 + 16 + 3077] == v and not any(buf[2 * 4096 + 16 + 3077:3 * 4096])
    node = struct.pack("<HHHH", 3077, 0, 1, 8) + k + struct.pack("<Q", 2)
    assert buf[4 * 4096 - len(node):4 * 4096] == node
    sub = struct.pack("<IHHQQQQQ", 0, 0, 1, 0, 1, 1, 1, 3)
    node = struct.pack("<HHHH", 48, 0, 2, 7) + b"vectors" + sub + b"\0"
    assert buf[5 * 4096 - len(node):] == node
    main = struct.pack("<IHHQQQQQ", 0, 0, 1, 0, 1, 0, 1, 4)
    assert buf[4096 + 88:4096 + 136] == main and struct.unpack_from("<QQ", buf, 4096 + 136) == (4, 1)
    recs, st = parse_lmdb(p, "vectors")
    assert recs == [(k, v)] and st["overflow"] == 1


def random_records(rng, n, max_val, klen=8):
    keys = sorted({bytes(rng.integers(0, 256, klen if klen else int(rng.integers(1, 40)), dtype=np.uint8))
                   for _ in range(n)})
    sizes = rng.integers(0, max_val, len(keys))
    return [(k, bytes(rng.integers(0, 256, int(s), dtype=np.uint8))) for k, s in zip(keys, sizes)]


@pytest.mark.parametrize("page_size,n,max_val,name,klen", [
    (4096, 1, 10, None, 8), (4096, 300, 5000, None, 8), (4096, 40000, 60, None, 8),
    (4096, 5000, 300, "hannoy", 0), (512, 3000, 700, None, 8), (16384, 2000, 20000, "x", 8),
    (4096, 3000, 2100, None, 8)])
def test_round_trip_matches_independent_parser(tmp_path, page_size, n, max_val, name, klen):
    rng = np.random.default_rng(n + page_size)
    recs = random_records(rng, n, max_val, klen)
    p = str(tmp_path / "data.mdb")
    write(p, recs, name=name, page_size=page_size)
    got, st = parse_lmdb(p, name)
    assert got == recs and st["page_size"] == page_size
    with capi.LmdbEnv(p, name) as env:
        s = env.stat()
        assert (s["entries"], s["depth"], s["branch_pages"], s["leaf_pages"], s["overflow_pages"]) == \
               (st["entries"], st["depth"], st["branch"], st["leaf"], st["overflow"])
        assert s["txnid"] == 1 and s["last_pgno"] == st["last"]
        assert env.items() == recs
        for i in rng.integers(0, len(recs), 200):
            assert env.get(recs[i][0]) == recs[i][1]
        for _ in range(100):  # absent keys (mdb_get -> MDB_NOTFOUND)
            k = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
            if k not in dict(recs):
                assert env.get(k) is None
        if len(recs) > 10:  # MDB_SET_RANGE .. <= hi
            a, b = sorted(rng.integers(0, len(recs), 2))
            assert env.items(recs[a][0], recs[b][0]) == recs[a:b + 1]
            lo = recs[a][0][:-1] + bytes([max(recs[a][0][-1] - 1, 0)])
            assert env.items(lo, None) == [r for r in recs if r[0] >= lo]


def test_three_level_tree_and_branch_minimum(tmp_path):
    """enough 500-byte keys for three levels on 4 KiB pages; every branch page keeps >= 2 keys, also when
    the greedy fill would leave one child for the last page"""
    for n in (7 * 7 * 5 + 1, 7 * 7 * 7 + 1, 2000):
        recs = [(struct.pack(">I", i) + b"k" * 496, b"") for i in range(n)]
        p = str(tmp_path / f"d{n}.mdb")
        write(p, recs)
        got, st = parse_lmdb(p)  # asserts n >= 2 on every branch page
        assert got == recs and st["depth"] >= 3
        with capi.LmdbEnv(p) as env:
            env.verify()
            assert env.get(recs[n - 1][0]) == b"" and env.get(recs[0][0]) == b""


def test_empty_environment(tmp_path):
    p = str(tmp_path / "data.mdb")
    write(p, [])
    assert os.path.getsize(p) == 2 * 4096
    recs, st = parse_lmdb(p)
    assert recs == [] and st["depth"] == 0 and st["root"] == INVALID
    with capi.LmdbEnv(p) as env:
        assert env.items() == [] and env.get(b"k") is None
    write(p, [], name="empty")
    assert parse_lmdb(p, "empty")[0] == []
    with capi.LmdbEnv(p, "empty") as env:
        assert env.items() == []
    with pytest.raises(H.HannoyError) as e:
        capi.LmdbEnv(p, "other")
    assert e.value.code == capi.ERR_MISSING_KEY


def test_writer_rejects_what_lmdb_rejects(tmp_path):
    p = str(tmp_path / "data.mdb")
    w = capi.LmdbWriter(p)
    w.put(b"b", b"1")
    for k in (b"a", b"b", b"", b"k" * 512):  # MDB_APPEND order, MDB_BAD_VALSIZE
        with pytest.raises(H.HannoyError) as e:
            w.put(k, b"")
        assert e.value.code == capi.ERR_INVALID_ARG
    w.put(b"ba", b"2")  # a longer key with the same prefix sorts after
    w.finish()
    assert parse_lmdb(p)[0] == [(b"b", b"1"), (b"ba", b"2")]
    with pytest.raises(H.HannoyError):
        capi.LmdbWriter(p, page_size=3000)
    with pytest.raises(H.HannoyError) as e:
        capi.LmdbWriter(str(tmp_path / "no" / "such" / "dir.mdb"))
    assert e.value.code == capi.ERR_IO


def test_reader_reports_corruption_instead_of_crashing(tmp_path):
    rng = np.random.default_rng(5)
    recs = random_records(rng, 3000, 3000)
    p = str(tmp_path / "data.mdb")
    write(p, recs)
    good = open(p, "rb").read()
    with pytest.raises(H.HannoyError):  # not an LMDB file
        open(p, "wb").write(b"\0" * 8192)
        capi.LmdbEnv(p)
    open(p, "wb").write(good[:len(good) // 2])  # truncated
    with pytest.raises(H.HannoyError):
        capi.LmdbEnv(p)
    bad = 0
    for trial in range(40):  # random page-header damage must surface as an error or as intact data
        b = bytearray(good)
        pg = int(rng.integers(2, len(good) // 4096))
        off = pg * 4096 + int(rng.integers(0, 24))
        b[off] ^= 1 << int(rng.integers(0, 8))
        open(p, "wb").write(bytes(b))
        try:
            with capi.LmdbEnv(p) as env:
                got = env.items()
            assert len(got) <= len(recs) + 1
        except H.HannoyError as e:
            assert e.code == capi.ERR_IO
            bad += 1
    assert bad > 0


def test_hannoy_records_of_kat1_round_trip(tmp_path):
    """the byte-exact records of the KAT-1 index (oracle encoders) through data.mdb and back"""
    from oracle import orc
    vecs = np.array([[i, 0] for i in range(6)], np.float32)
    ds = orc.Dataset.from_f32(orc.EUCLIDEAN, vecs, np.array([1, 0, 1, 1, 0, 0], np.uint8))
    g = orc.build(ds, M=3, M0=3, ef=100, order=orc.ORDER_X86, batch_max=1)
    recs = orc.encode_kv(ds, g, index=0, with_items=True)
    assert recs == sorted(recs)
    p = str(tmp_path / "data.mdb")
    write(p, recs)
    assert parse_lmdb(p)[0] == recs
    with capi.LmdbEnv(p) as env:
        assert env.items() == recs
        assert env.get(H.api.key(0, 0)) == recs[0][1]  # Metadata
        links = env.items(H.api.key(0, 2), H.api.key(0, 2, 0xFFFFFFFF, 0xFF))
        assert len(links) == 6 + 3  # layer 0 for all, layer 1 for items 0, 2, 3
