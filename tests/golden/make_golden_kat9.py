"""Writes tests/golden/kat9_100x30.json from the reference's own snapshot files (run in the build container,
where /root/reference exists; the tests read only the JSON).

KAT-9 = src/tests/writer.rs:130-155 `write_and_update_lot_of_random_points_with_snapshot`: 100 items of
30 dimensions (`rng.gen::<f32>()` from StdRng::from_seed([42; 32])), Euclidean, build::<3, 3>, then the 50
even ids overwritten with fresh random vectors and a second build — with the two `insta` snapshots
src/tests/snapshots/hannoy__tests__writer__write_and_update_lot_of_random_points_with_snapshot{,-2}.snap
(Version { 0, 1, 3 }: current).  The fixture is DATA transcribed from those files: per snapshot the
entry points, max_level, every Links record as [item, layer, [neighbours]] (a record's layer = its position
among the item's Links keys, LMDB key order (item, layer)) and the vector components the dump prints (the
first ten of each item, four decimals).  The inputs themselves are not in the files; the tests regenerate
them with the restated rand 0.8.5 StdRng and check them against the printed components."""
import json
import os
import re

SNAP = "/root/reference/src/tests/snapshots/hannoy__tests__writer__write_and_update_lot_of_random_points_with_snapshot%s.snap"


def parse(path):
    out = {"links": [], "printed": {}}
    seen = {}
    for line in open(path):
        line = line.strip()
        m = re.match(r'Root: Metadata \{ dimensions: (\d+), items: RoaringBitmap<(\d+) values between (\d+) and (\d+)>, '
                     r'distance: "(\w+)", entry_points: \[(.*)\], max_level: (\d+) \}', line)
        if m:
            out.update(dim=int(m.group(1)), n_items=int(m.group(2)), distance=m.group(5),
                       entry_points=[int(x) for x in m.group(6).split(",") if x.strip()], max_level=int(m.group(7)))
        m = re.match(r"Version: Version \{ major: (\d+), minor: (\d+), patch: (\d+) \}", line)
        if m:
            out["version"] = [int(m.group(i)) for i in (1, 2, 3)]
        m = re.match(r"Links (\d+): Links\(Links \{ links: RoaringBitmap<\[(.*)\]> \}\)", line)
        if m:
            item = int(m.group(1))
            layer = seen.get(item, 0)
            seen[item] = layer + 1
            out["links"].append([item, layer, [int(x) for x in m.group(2).split(",") if x.strip()]])
        m = re.match(r'Item (\d+): Item\(Item \{ header: NodeHeaderEuclidean \{ bias: "(.*)" \}, vector: \[(.*)\] \}\)', line)
        if m:
            comps = [x.strip() for x in m.group(3).split(",")]
            assert comps[-1] == '"other ..."' and m.group(2) == "0.0000"
            out["printed"][m.group(1)] = comps[:-1]
    return out


if __name__ == "__main__":
    kat = {"source": "src/tests/writer.rs:130-155 + src/tests/snapshots/*_with_snapshot{,-2}.snap",
           "seed": [42] * 32, "M": 3, "M0": 3, "ef_construction": 100, "alpha": 1.0, "n": 100, "dim": 30,
           "metric": "euclidean", "updated_ids": list(range(0, 100, 2)),
           "fresh": parse(SNAP % ""), "updated": parse(SNAP % "-2")}
    for k in ("fresh", "updated"):
        s = kat[k]
        assert s["dim"] == 30 and s["n_items"] == 100 and s["version"] == [0, 1, 3] and len(s["printed"]) == 100
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "kat9_100x30.json"), "w") as f:
        json.dump(kat, f, separators=(",", ":"))
    print("fresh:", len(kat["fresh"]["links"]), "records; updated:", len(kat["updated"]["links"]), "records")
