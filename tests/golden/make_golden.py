"""Writes tests/golden/kat.json.

The reference cannot be built or imported in this image (no cargo/rustc, no
LMDB), so these known-answer vectors are DATA transcribed from the reference's
own test expectations (inputs + expected outputs), not outputs of running it:

  KAT-1  /root/reference/src/tests/writer.rs:376-408 (== 589-622)   fresh build snapshot
  KAT-2  /root/reference/src/tests/writer.rs:410-437                overwrite one item, rebuild
  KAT-3/4 /root/reference/src/tests/writer.rs:624-677               delete item 3, then item 1
  KAT-5  /root/reference/src/tests/writer.rs:67-128, 562-570        single-item builds
  KAT-6  /root/reference/src/unaligned_vector/binary_quantized_test.rs:11-27,100-167
         /root/reference/src/unaligned_vector/binary_test.rs:11-64   quantiser bit patterns
  KAT-7  /root/reference/src/spaces/simple_avx.rs:120-144           SIMD == scalar inputs
  KAT-8  /root/reference/tests/test_basic.py:8-34                   python Hamming fixture
  codec  /root/reference/src/key.rs:129-162, node_id.rs:111-138      key layout facts

Levels in KAT-1/5 are the ones the snapshot implies for StdRng::from_seed([42;32])
(tests/mod.rs:145-147) with M = M0 = 3; the RNG itself is not restated.
"""
import json
import os

kat = {}

kat["kat1"] = {
    "source": "src/tests/writer.rs:376-408",
    "metric": "euclidean", "dim": 2, "M": 3, "M0": 3, "ef_construction": 100, "alpha": 1.0,
    "vectors": [[float(i), 0.0] for i in range(6)],
    "ids": list(range(6)),
    "levels": [1, 0, 1, 1, 0, 0],
    "entry_points": [0, 2, 3], "max_level": 1,
    "links": [  # [item, layer, [neighbours]] in LMDB key order (item, layer)
        [0, 0, [1, 2]], [0, 1, [2]], [1, 0, [0, 2]], [2, 0, [0, 1, 3]], [2, 1, [0, 3]],
        [3, 0, [2, 4]], [3, 1, [2]], [4, 0, [3, 5]], [5, 0, [4]]],
}

# incremental builds on top of KAT-1's DB (same M = M0 = 3, efC = 100)
kat["kat2"] = {
    "source": "src/tests/writer.rs:410-437 (overwrite_one_item_incremental, second snapshot)",
    "overwrite": {"id": 3, "vector": [6.0, 0.0]}, "to_insert": [3], "to_delete": [],
    "insert_levels_any_of": [[0], [1]],
    "entry_points": [0, 2, 3], "max_level": 1,
    "links": [[0, 0, [1, 2]], [0, 1, [2]], [1, 0, [0, 2]], [2, 0, [1, 4]], [2, 1, [0, 3]],
              [3, 0, [5]], [3, 1, [2]], [4, 0, [2, 3, 5]], [5, 0, [3, 4]]],
}
kat["kat3"] = {
    "source": "src/tests/writer.rs:624-650 (delete_one_item, second snapshot)",
    "to_insert": [], "to_delete": [3], "entry_points": [0, 1, 2], "max_level": 1,
    "links": [[0, 0, [1]], [0, 1, [1, 2]], [1, 0, [0, 2]], [1, 1, [0, 2]], [2, 0, [1, 2, 4]],
              [2, 1, [1, 2]], [4, 0, [2, 4, 5]], [5, 0, [4]]],
}
kat["kat4"] = {
    "source": "src/tests/writer.rs:652-677 (delete_one_item, third snapshot; applied after kat3)",
    "to_insert": [], "to_delete": [1], "entry_points": [0, 2, 4], "max_level": 1,
    "links": [[0, 0, [0, 2]], [0, 1, [0, 2]], [2, 0, [0, 2, 4]], [2, 1, [0, 2, 4]],
              [4, 0, [2, 4, 5]], [4, 1, [2]], [5, 0, [4]]],
}

kat["kat5"] = [
    {"source": "src/tests/writer.rs:109-128", "metric": "euclidean", "dim": 3, "M": 3, "M0": 3,
     "id": 0, "vector": [0.0, 1.0, 2.0], "level": 1, "entry_points": [0], "max_level": 1,
     "links": [[0, 0, []], [0, 1, []]], "header_f32": 0.0},
    {"source": "src/tests/writer.rs:67-86", "metric": "euclidean", "dim": 3, "M": 3, "M0": 3,
     "id": 4294967294, "vector": [0.0, 1.0, 2.0], "level": 1, "entry_points": [4294967294],
     "max_level": 1, "links": [[4294967294, 0, []], [4294967294, 1, []]], "header_f32": 0.0},
    {"source": "src/tests/writer.rs:88-107", "metric": "euclidean", "dim": 3, "M": 3, "M0": 3,
     "id": 4294967295, "vector": [0.0, 1.0, 2.0], "level": 1, "entry_points": [4294967295],
     "max_level": 1, "links": [[4294967295, 0, []], [4294967295, 1, []]], "header_f32": 0.0},
]

large = [(-1.0 if (n % 3 == 0 or n % 5 == 0) else 1.0) for n in range(100)]
kat["kat6"] = [
    {"source": "binary_quantized_test.rs:11-27", "codec": "binary_quantized",
     "input": [0.1, 0.2, -0.3, 0.4, -0.5, 0.6, -0.7, 0.8, -0.9],
     "bytes_bin": ["10101011"] + ["00000000"] * 7},
    {"source": "binary_quantized_test.rs:100-117", "codec": "binary_quantized",
     "input": [-1.0, 2.0, -3.0, 4.0, 5.0], "bytes_bin": ["00011010"] + ["00000000"] * 7},
    {"source": "binary_quantized_test.rs:132-160", "codec": "binary_quantized", "input": large,
     "bytes_bin": ["10010110", "01101001", "11001011", "10110100", "01100101", "11011010",
                   "00110010", "01101101", "10011001", "10110110", "01001100", "01011011",
                   "00000110", "00000000", "00000000", "00000000"]},
    {"source": "binary_test.rs:11-27", "codec": "binary",
     "input": [0.1, 0.2, -0.3, 0.4, -0.5, 0.6, -0.7, 0.8, -0.9],
     "bytes_bin": ["10101011"] + ["00000000"] * 7},
    # binary_test.rs:29-64: Binary maps 0.0 -> 0 (iter gives [1,0,0,1,0,1,0,1,0])
    {"source": "binary_test.rs:29-64", "codec": "binary",
     "input": [0.1, 0.0, -0.3, 0.4, -0.5, 0.6, -0.7, 0.8, -0.9],
     "bytes_bin": ["10101001"] + ["00000000"] * 7},
]

v1 = [float(x) for x in list(range(10, 26)) * 4 + list(range(26, 32))]
v2 = [float(x) for x in list(range(40, 56)) + list(range(10, 26)) * 3 + list(range(56, 62))]
kat["kat7"] = {"source": "src/spaces/simple_avx.rs:120-144", "v1": v1, "v2": v2}

kat["kat8"] = {
    "source": "tests/test_basic.py:8-34", "metric": "hamming", "dim": 3, "M": 4, "M0": 8,
    "ef_construction": 10,
    "vectors": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], "ids": [0, 1, 2],
    "query": [0.0, 1.0, 0.0], "k": 2, "first_hit": [1, 0.0], "n_hits": 2,
}

kat["keys"] = {
    "source": "src/key.rs:54-66, src/node_id.rs:11-21",
    "cases": [  # [index, mode, item, layer, hex]
        [0, 0, 0, 0, "0000000000000000"],            # metadata
        [0, 0, 1, 0, "0000000000000100"],            # version
        [0, 2, 5, 1, "0000020000000501"],            # links(5, layer 1)
        [258, 3, 4294967294, 0, "010203fffffffe00"],  # item u32::MAX-1, index 0x0102
    ],
}

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
with open(out, "w") as f:
    json.dump(kat, f, indent=1)
print("wrote", out)
