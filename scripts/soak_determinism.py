"""Determinism soak: several full builds per configuration on one builder, identical records and
evaluation counts every time (python scripts/soak_determinism.py on the MI355X box)."""
import sys, zlib, numpy as np, torch
sys.path.insert(0, ".")
import hannoy_amd as H
dev = torch.device("cuda", 0)
def data(n, dim, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    c = torch.rand((1024, dim), generator=g, device=dev) * 2 - 1
    w = torch.randint(0, 1024, (n,), generator=g, device=dev)
    return (c[w] + 0.15 * torch.randn((n, dim), generator=g, device=dev)).cpu().numpy()
for metric, n, dim, M, ef in ((H.COSINE, 1_000_000, 768, 16, 100), (H.HAMMING, 2_000_000, 1024, 16, 64),
                              (H.COSINE, 2_000_000, 128, 16, 100), (H.EUCLIDEAN, 500_000, 768, 32, 200),
                              (H.BQ_COSINE, 1_000_000, 768, 16, 100)):
    items = H.ItemSet.from_f32(metric, data(n, dim, 5))
    crcs = set()
    with H.Builder(items, M=M, M0=2 * M, ef_construction=ef, seed=3) as b:
        for rep in range(6):
            b.reset(); b.run(); g = b.finish()
            crcs.add((zlib.crc32(g.nbrs.tobytes()), zlib.crc32(g.offsets.tobytes()), g.n_distance_evals, g.n_tie_pool_overflow))
    print(metric, n, dim, "distinct results:", len(crcs), crcs, flush=True)
    assert len(crcs) == 1
print("soak ok")
