"""hannoy_amd — MI355X-native HNSW index builder for hannoy (drop-in for Writer::builder().build()).

The compute path is the HIP extension hannoy_amd/libhannoy_amd.so (C ABI: include/hannoy_amd.h).
"""
from ._capi import (BQ_COSINE, BQ_EUCLIDEAN, BQ_MANHATTAN, COSINE, EUCLIDEAN, HAMMING, MANHATTAN,  # noqa: F401
                    METRIC_NAMES, BuildCancelled, Builder, MultiBuilder, StdRng, abi_sizes, Graph, HannoyError, ItemSet, build, build_incremental, draw_levels, default_batch_max, set_graph_cache, selftest_lane_ops,
                    NNS_NONE, SCHED_NO_SHUFFLE, SCHED_LEVEL_ORDER_ID, SCHED_UPDATE_NO_RAMP, encode_vectors, header_bytes, load_library, make_opts, vector_bytes)

from .api import Database, InvalidVecDimension, Metric, Reader, Writer  # noqa: E402,F401

__all__ = ["Database", "Writer", "Reader", "Metric", "Builder", "MultiBuilder", "Graph", "ItemSet", "build", "encode_vectors", "load_library", "HannoyError",
           "BuildCancelled"]
