"""Random-configuration soak: fresh build + incremental rounds (deletes — now and then of the entry points —
overwrites, additions) of random shapes — metric, dim, M, M0 up to several hundred, ef, schedule — GPU ==
oracle edge for edge; after the last round the stored graph is loaded and searched (Reader::nns by vector,
random k / ef_search, with and without a candidates filter) == the restated Reader.
  python scripts/soak_random_configs.py [n_configs] [seed] [rounds]     (on the MI355X box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hannoy_amd as hny
from oracle import orc
from conftest import draw_levels

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# SOAK_RANKS=k: through the native multi-GPU driver with k ranks sharing GPU 0 (HNY_MGPU_SHIM=1 HNY_MGPU_VERIFY=1);
# SOAK_NMAX: largest index (default 2500; larger ones reach the locality-ordered batches and the XCD-tiled queue)
ranks = int(os.environ.get("SOAK_RANKS", "0"))
nmax = int(os.environ.get("SOAK_NMAX", "2500"))
# SOAK_BIG_EF=1: half of the configurations with ef_construction 513 .. 7 000, half of the searches with ef_search up to 7 000
big_ef = os.environ.get("SOAK_BIG_EF", "0") != "0"


def same(g, o):
    return (np.array_equal(g.rec_item, o.rec_item) and np.array_equal(g.rec_layer, o.rec_layer)
            and np.array_equal(g.offsets, o.offsets) and np.array_equal(g.nbrs, o.nbrs)
            and g.entry_points.tolist() == o.entry_points.tolist() and g.max_level == o.max_level
            and g.n_evals_walk == o.n_evals_walk)


t0 = time.time()
refused = 0
for ci in range(n_cfg):
  try:
      metric = int(rng.integers(0, 7))
      dim = int(rng.choice([3, 20, 48, 100, 128, 300, 768, 1024]))
      M = int(rng.choice([4, 8, 12, 16, 24, 32]))
      M0 = int(rng.choice([M, 2 * M, 2 * M + 1, 70, 100, 200, 333, 768])) if rng.random() < 0.7 else 2 * M
      lvM = M
      if big_ef and rng.random() < 0.25:  # M beyond one wave's lanes; levels drawn as for M = 4 so that upper layers fill
          M = int(rng.choice([65, 80, 128]))
          lvM = 4
      M0 = max(M0, M)
      n0 = int(rng.integers(400, nmax))
      ef = int(rng.integers(16, 90))
      if big_ef and rng.random() < 0.5:
          ef = int(rng.choice([513, 700, 1500, 4095, 4096, 7000]))
      frac = float(rng.choice([0.05, 0.25, 1.0]))
      bmax = int(rng.choice([16, 256, 4096] if nmax <= 2500 else [256, 4096, 16384]))
      clustered = rng.random() < 0.5
      def vec(k):
          if clustered:
              return (cent[rng.integers(0, len(cent), k)] + 0.3 * rng.standard_normal((k, dim))).astype(np.float32)
          return rng.uniform(-1, 1, (k, dim)).astype(np.float32)
      cent = rng.uniform(-1, 1, (8, dim)).astype(np.float32)
      vecs = {i: v for i, v in enumerate(vec(n0))}
      kw_o = dict(M=M, M0=M0, ef=ef, order=orc.ORDER_WAVE, batch_frac=frac, batch_max=bmax, threads=8)
      kw_g = dict(M=M, M0=M0, ef_construction=ef, batch_frac=frac, batch_max=bmax)
      if ranks:
          kw_g["devices"] = [0] * ranks
      tag = f"#{ci} metric {metric} dim {dim} M {M} M0 {M0} n {n0} ef {ef} frac {frac} bmax {bmax} {'clustered' if clustered else 'uniform'}"
      print("run ", tag, flush=True)
      def mk(levels):
          ids = np.array(sorted(vecs), np.uint32)
          mat = np.stack([vecs[int(i)] for i in ids])
          return orc.Dataset.from_f32(metric, mat, levels if levels is not None else np.zeros(len(ids), np.uint8), ids)
      ds = mk(draw_levels(n0, lvM, seed=ci))
      hist = dict(params=np.array([metric, dim, M, M0, ef, bmax], np.int64), frac=np.array([frac]), ids0=ds.ids.copy(), mat0=np.stack([vecs[int(i)] for i in ds.ids]), lv0=np.asarray(ds.levels).copy())
      items = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, ds.levels)
      og = orc.build(ds, **kw_o)
      gg = hny.build(items, **kw_g)
      ok1 = same(gg, og)
      if not ok1:
          dg, do = gg.as_dict(), og.as_dict()
          bad = [k2 for k2 in sorted(do) if list(dg.get(k2, [])) != list(do[k2])]
          print("   fresh build differs in", len(bad), "records; evals", gg.n_evals_walk, og.n_evals_walk,
                [(k2, len(dg.get(k2, [])), len(do[k2]), sorted(set(dg.get(k2, [])) ^ set(do[k2]))[:8]) for k2 in bad[:4]], flush=True)
      # incremental rounds
      ok2 = True
      next_id = n0
      for rnd in range(rounds):
          alive = sorted(vecs)
          to_delete = set(rng.choice(alive, max(1, len(alive) // 10), replace=False).tolist())
          if rng.random() < 0.4:  # hnsw.rs:236-263: deleted entry points get replaced, max_level may reset
              to_delete |= set(int(x) for x in gg.entry_points.tolist())
          to_delete = sorted(to_delete & set(alive))
          if len(to_delete) >= len(alive) - 2:
              break
          for i in to_delete:
              del vecs[i]
          alive = sorted(vecs)
          overwrite = sorted(rng.choice(alive, max(1, len(alive) // 30), replace=False).tolist())
          for i, v in zip(overwrite, vec(len(overwrite))):
              vecs[i] = v
          added = list(range(next_id, next_id + int(rng.integers(1, max(2, n0 // 6)))))
          next_id = added[-1] + 1
          for i, v in zip(added, vec(len(added))):
              vecs[i] = v
          to_insert = sorted(overwrite + added)
          lv = draw_levels(len(to_insert), lvM, seed=100 * (rnd + 1) + ci)
          if rng.random() < 0.3:
              lv = np.zeros(len(to_insert), np.uint8)  # every new item on level 0 (hnsw.rs:278-285)
          ds2 = mk(None)
          items2 = hny.ItemSet(metric, dim, ds2.ids, ds2.codes, ds2.headers, lv)
          hist[f"ids{rnd + 1}"] = ds2.ids.copy(); hist[f"mat{rnd + 1}"] = np.stack([vecs[int(i)] for i in ds2.ids])
          hist[f"ins{rnd + 1}"] = np.array(to_insert, np.uint32); hist[f"lv{rnd + 1}"] = np.asarray(lv, np.uint8); hist[f"del{rnd + 1}"] = np.array(to_delete, np.uint32)
          og = orc.build_incremental(ds2, og, to_insert, lv, to_delete, **{k: v for k, v in kw_o.items() if k != "threads"})
          gg = hny.build_incremental(items2, gg, to_insert, to_delete, **kw_g)
          okr = same(gg, og)
          if not okr:
              dg, do = gg.as_dict(), og.as_dict()
              bad = [k2 for k2 in sorted(do) if list(dg.get(k2, [])) != list(do[k2])]
              print("   round", rnd, "differs in", len(bad), "records of", len(do), "/", len(dg), "; evals", gg.n_evals_walk, og.n_evals_walk,
                    "eps", gg.entry_points.tolist()[:6], og.entry_points.tolist()[:6], "deleted eps" , len(set(to_delete)),
                    [(k2, len(dg.get(k2, [])), len(do[k2])) for k2 in bad[:4]], flush=True)
          ok2 = ok2 and okr
          ds = ds2
      # search on the stored graph
      nq = 64
      qs = vec(nq)
      qc = orc.encode_vectors(metric, qs)
      qh = orc.make_headers(metric, dim, qc)
      k = int(rng.integers(1, 20))
      efs = int(rng.integers(1, 120))
      if big_ef and rng.random() < 0.5:
          efs = int(rng.choice([513, 2000, 4095, 4096, 7000]))
      efs_f = efs  # (the filtered search takes any ef_search too: k_nns_heap beyond 4 095)
      items_s = hny.ItemSet(metric, dim, ds.ids, ds.codes, ds.headers, np.zeros(0, np.uint8))
      cand = np.sort(rng.choice(ds.ids, max(1, len(ds.ids) // int(rng.integers(2, 40))), replace=False)).astype(np.uint32)
      with hny.Builder(items_s, prev=gg, load=True, M=M, M0=M0, ef_construction=ef) as b:
          gi, gd, gc = b.search_knn(qc, qh, k=k, ef_search=efs)
          fi, fd, fc = b.nns(qc, qh, k=k, ef_search=efs_f, candidates=cand)
      oi, od, oc = orc.search(ds, gg, qc, qh, k=k, ef_search=efs, order=orc.ORDER_WAVE, threads=8)
      ok3 = np.array_equal(oc, gc) and np.array_equal(oi, gi) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
      oi, od, oc = orc.search(ds, gg, qc, qh, k=k, ef_search=efs_f, order=orc.ORDER_WAVE, threads=8, candidates=cand)
      ok3 = ok3 and np.array_equal(oc, fc) and np.array_equal(oi, fi)
      if not ok3:
          print("   search differs: k", k, "ef", efs, "counts equal", np.array_equal(oc, fc), flush=True)
      ok2 = ok2 and ok3
      print(("ok  " if ok1 and ok2 else "FAIL"), tag, "fresh", ok1, "incremental", ok2, f"[{time.time() - t0:.0f} s]", flush=True)
      if not (ok1 and ok2):
          os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
          np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"soak_fail_{ci}.npz"), **hist)
      assert ok1 and ok2, tag
  except hny.HannoyError as e:
    # inside the contract: degenerate ties overflow the walk's tie pool / a result set beyond 4 096 entries — refused loudly
    if (e.code == -7 and ("tie pool overflow" in str(e) or "kernel overflow: res" in str(e))) or (e.code == -5 and "entry points >" in str(e)):
        refused += 1
        print("refused", tag, "--", str(e)[:90], flush=True)
    else:
        raise
print("soak ok:", n_cfg, "configurations,", refused, "refused loudly")
