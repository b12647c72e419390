/*
 * hannoy_amd.h — C ABI of the MI355X-native HNSW index builder for hannoy.
 *
 * Drop-in boundary: the reference has no FFI for the build; the seam is inside
 * HnswBuilder::build (/root/reference/src/hnsw.rs:122-216) between "inputs read through
 * FrozenReader" (hnsw.rs:138, src/parallel.rs:33-45) and the single-threaded LMDB write loop
 * (hnsw.rs:191-213).  A maintainer replaces the body of HnswBuilder::build with: export items →
 * hny_build() → for each record db.put(Key::links(..), Links) (INTEGRATION.md shows the Rust
 * binding).  All pointers are caller-owned unless stated; the library never frees caller memory.
 * Plain C types only.  One host thread drives one builder (same contract as Writer::build, which
 * takes &mut RwTxn, /root/reference/src/writer.rs:521).
 *
 * The library REQUIRES a gfx950 GPU: there is no CPU fallback; every entry point that computes
 * returns HNY_ERR_NO_DEVICE when no device is usable.
 */
#ifndef HANNOY_AMD_H
#define HANNOY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* replaces: trait Distance implementors, src/distance/mod.rs:3-10 (names: cosine.rs:32-34 etc.) */
typedef enum {
  HNY_COSINE = 0,
  HNY_EUCLIDEAN = 1,
  HNY_MANHATTAN = 2,
  HNY_HAMMING = 3,
  HNY_BQ_COSINE = 4,
  HNY_BQ_EUCLIDEAN = 5,
  HNY_BQ_MANHATTAN = 6
} hny_metric;

/* error codes; mirror of the variants the build can surface (src/error.rs:10-87) */
enum {
  HNY_OK = 0,
  HNY_ERR_INVALID_ARG = -1,      /* null pointer, M0 < M, unsorted ids, ... */
  HNY_ERR_CANCELLED = -2,        /* Error::BuildCancelled, error.rs:58-59 */
  HNY_ERR_MISSING_KEY = -3,      /* Error::MissingKey, error.rs:61-72 (entry point not in items) */
  HNY_ERR_INVALID_DIM = -4,      /* Error::InvalidVecDimension, error.rs:19-26 (stride/dim mismatch) */
  HNY_ERR_UNSUPPORTED = -5,      /* dim / M0 / ef beyond what the kernels are built for */
  HNY_ERR_NO_DEVICE = -6,        /* no usable gfx950 device / HIP failure */
  HNY_ERR_DEVICE = -7,           /* a kernel reported an internal overflow (see hny_last_error): the result
                                  * set of a walk that never evicts outgrew 65 536 entries (4 096 in a
                                  * filtered search) — an input the reference still handles, slowly.  (More
                                  * than 128 candidates tying with a result set's maximum — short codes, wide
                                  * lists — are NOT an error any more: those walks are repeated on heaps in
                                  * HBM, k_walk_heap / the heap-queue searcher.)  Nothing has been written and
                                  * no graph is returned; there is no CPU path in this library to repeat the
                                  * call on */
  HNY_ERR_OOM = -8
};

/* replaces: BuildOption + const generics M, M0 (src/writer.rs:34-58, 215-220) + rng (hnsw.rs:129) */
typedef struct {
  int32_t metric;            /* hny_metric */
  uint32_t dim;              /* user dimensions (binary codecs pad to 64, binary.rs:80-94) */
  uint32_t M, M0;            /* defaults 16, 32 (README.md:51, python.rs:120).  1 <= M <= M0 <= 1024
                              * in the wave order — fresh and incremental builds,
                              * loading / searching a stored graph; strict mode
                              * (x86_order) too, except for an incremental build over lists of more than 64 slots.
                              * That covers every pair the reference's Python API offers ((4,8) ..
                              * (32,64), python.rs:280) and the pair of the reference's fuzz test,
                              * M = 16, M0 = 768 with incremental builds (src/tests/fuzz.rs:86-87,143).
                              * Anything wider: HNY_ERR_UNSUPPORTED, never a silently different
                              * graph.  (Incremental builds on lists of more than 64 slots read a
                              * per-record bitmap over all slots back: meant for small indexes.) */
  uint32_t ef_construction;  /* default 100 (writer.rs:49); 1 .. 65 535 (x86_order: result sets of at most 4 096 entries) */
  float alpha;               /* default 1.0 (writer.rs:51) */
  uint64_t seed;             /* levels when items.levels == NULL: drawn exactly as the reference
                              * draws them from StdRng::seed_from_u64(seed) (python.rs:261) */
  int (*cancel)(void *);     /* polled before every batch, i.e. every <= batch_max items (default
                              * 65 536, ~30 ms of build; the reference probes every 10 000 items,
                              * lib.rs:140 — batch_max <= 10 000 gives exactly that bound) */
  void *cancel_ctx;
  void (*progress)(void *, uint64_t done, uint64_t total); /* progress.rs:3-16 */
  void *progress_ctx;
  /* batch-synchronous insertion schedule: batch = clamp(floor(batch_frac * n_inserted), 1,
   * batch_max).  batch_max = 1 reproduces strictly sequential insertion in the reference's own order
   * (hnsw.rs:268 incl. the order Rust's sort_unstable_by leaves equal levels in — the reference with
   * one rayon thread, src/tests/mod.rs:105).  With batch_max != 1 the items of a level group are taken
   * in a fixed pseudo-random order, so that a batch is a sample of the group whatever the id order
   * (DESIGN.md §1).  0 / 0.0 select the defaults: batch_frac 1.0, batch_max hny_default_batch_max(n). */
  double batch_frac;
  uint32_t batch_max;
  int32_t device;            /* HIP device ordinal; -1 = current */
  /* 1 = strict mode: f32 distances in the reference's own x86 summation order (AVX2+FMA for
   * dim >= 32, SSE for 16..31, scalar below; src/spaces/simple*.rs), bit for bit.  Slower; with
   * batch_max = 1 the build then equals the reference run with one thread.  0 = wave order
   * (within 1e-5 relative of it). */
  int32_t x86_order;
  /* replaces: the rayon pool of the insert loop (hnsw.rs:172-185).  n_gpus > 1, or n_gpus == 1 with
   * `devices` given: hny_build / hny_build_incremental run one replica per listed GPU of this node
   * (devices == NULL: ordinals 0 .. n_gpus-1; `device` is ignored), shard every batch's searches
   * across them and exchange the results with RCCL all-gathers over xGMI (hny_multi.cpp).  The
   * result is byte for byte the n_gpus == 1 result.  0 / 1 without `devices`: one GPU, no RCCL. */
  int32_t n_gpus;
  const int32_t *devices;
  /* HNY_SCHED_* bits: the insertion schedules of rounds 1-2, kept for A/B comparisons.  They
   * change which graph is built, so they are part of the options, not of the process environment.  0 = default. */
  uint32_t schedule;
  uint32_t reserved_;        /* 0 */
} hny_build_opts;

enum {
  HNY_SCHED_NO_SHUFFLE = 1u,     /* batches are consecutive runs of the reference's order instead of a fixed
                                  * pseudo-random sample of the level group */
  HNY_SCHED_LEVEL_ORDER_ID = 2u, /* equal levels in ascending id order instead of the order Rust's
                                  * sort_unstable_by leaves them in (hnsw.rs:268) */
  HNY_SCHED_UPDATE_NO_RAMP = 4u  /* an update's first batch counts the surviving records as already inserted
                                  * instead of ramping up from one member */
};

/* replaces: what FrozenReader hands to the builder (src/parallel.rs:33-45) */
typedef struct {
  uint64_t n;
  const uint32_t *ids;       /* strictly ascending (RoaringBitmap order, hnsw.rs:142-144) */
  const void *vectors;       /* codec bytes exactly as stored after the header (node.rs:136-140) */
  size_t stride;             /* bytes between vectors; >= codec size */
  const void *headers;       /* D::Header bytes: f32 norm | f32 bias | u64 idx (hamming.rs:21-25) */
  size_t header_size;        /* 4 or 8 */
  const uint8_t *levels;     /* optional: inject levels instead of drawing them from `seed` */
} hny_items;

/* replaces: what the write loop consumes (hnsw.rs:195-213) + entry_points/max_level read back by
 * Writer::build (writer.rs:591-592) + BuildStats (stats.rs:10-19).  Library-owned. */
typedef struct {
  uint64_t n_records;        /* one per (item, layer) key, sorted by (item id, layer) */
  const uint32_t *rec_item;  /* item id */
  const uint8_t *rec_layer;
  const uint64_t *rec_offset; /* n_records + 1 offsets into neighbours */
  const uint32_t *neighbours; /* item ids, ascending, deduplicated (RoaringBitmap::from_iter) */
  const uint32_t *entry_points;
  uint32_t n_entry_points;
  uint32_t max_level;
  uint64_t n_links_added;     /* BuildStats.n_links_added */
  uint64_t n_distance_evals;  /* distance evaluations performed on the device */
  uint64_t n_evals_walk, n_evals_prune, n_evals_apply; /* n_evals_walk is the reference's count (hnsw.rs:476,503 call
                              * sites; every parity test compares it with the oracle's).  The other two count what the
                              * kernels computed: robust_prune tests several candidates at once (the one-wave kernel
                              * for rows <= 512 B still counts a candidate only up to its first violating row), and
                              * add_link skips the re-prunes of lists already known to prune to themselves */
  uint64_t n_batches;
  double t_upload_s, t_build_s, t_export_s; /* host wall clock of the three phases */
  uint64_t n_tie_pool_overflow; /* always 0 in a returned graph: a walk whose tie pool overflows is repeated
                                 * on heaps in HBM (k_walk_heap) and is not counted here */
  /* device time per kernel family, from HIP events on the build stream; filled only after
   * hny_builder_set_profiling(b, 1) (else 0) */
  double t_walk_kernels_s, t_prune_kernels_s, t_sort_kernels_s, t_apply_kernels_s;
  uint64_t n_walk_launches;
} hny_graph;

typedef struct hny_builder hny_builder;

typedef struct {
  uint64_t first;  /* index of the batch's first member in insertion order */
  uint32_t count;  /* 0 = build finished */
  uint32_t level;  /* level shared by every member (batches never straddle level groups) */
  uint32_t n_layers; /* level + 1 */
  uint32_t sel_stride_u64; /* u64 words per member in a selection buffer */
} hny_batch;

/* ---- one-call build: replaces HnswBuilder::build (hnsw.rs:122-216) ---- */
int hny_build(const hny_build_opts *opts, const hny_items *items, hny_graph **out);
void hny_graph_free(hny_graph *g);
/* Opt-in recycling of the four large arrays of an exported graph, 1.4 GB at C4: with max_bytes > 0, hny_graph_free keeps
 * them (at most max_bytes in total, process-wide) for the next export instead of returning them to the system — a
 * caller that rebuilds in a loop saves the munmap of the old arrays and the first touch of the new ones (~100 ms
 * per C4 build).  0 (the default) turns it off and releases what is held: nothing stays resident behind the
 * caller's back.  Independent of this, a builder allocates and touches its export arrays on a helper thread while
 * the device builds, so a one-off build does not pay their first touch either; the neighbour array keeps the
 * capacity of full lists until hny_graph_free. */
void hny_set_graph_cache(size_t max_bytes);

/* ---- incremental build: HnswBuilder::build on a non-empty index (hnsw.rs:122-216 with
 * prepare_levels_and_entry_points' deletion branch :236-289, on-disk links in get_neighbours
 * :438-441, fill_gaps_from_deleted :334-415) + delete_links_from_db (writer.rs:692-718).
 * `items`: every item that exists AFTER the update (item_indices, writer.rs:548-553; deleted items
 * gone, overwritten items with their new vector); items->levels, if given, holds one level per
 * to_insert id.  `prev`: what FrozenReader::links / iter_links and the Metadata record yield.
 * The result is the complete set of Links records after the build. ---- */
typedef struct {
  uint64_t n_records;
  const uint32_t *rec_item;
  const uint8_t *rec_layer;
  const uint64_t *rec_offset;
  const uint32_t *neighbours;
  const uint32_t *entry_points;
  uint32_t n_entry_points;
  uint32_t max_level;
} hny_prev_graph;
int hny_build_incremental(const hny_build_opts *opts, const hny_items *items, const uint32_t *to_insert,
                          uint64_t n_insert, const uint32_t *to_delete, uint64_t n_delete,
                          const hny_prev_graph *prev, hny_graph **out);

/* ---- stepwise build (what hny_build loops over; used by the multi-GPU driver) ---- */
int hny_builder_create(const hny_build_opts *opts, const hny_items *items, hny_builder **out);
/* stepwise form of hny_build_incremental: after the batches call hny_builder_fill_gaps once, then
 * hny_builder_finish. */
int hny_builder_create_incremental(const hny_build_opts *opts, const hny_items *items,
                                   const uint32_t *to_insert, uint64_t n_insert,
                                   const uint32_t *to_delete, uint64_t n_delete,
                                   const hny_prev_graph *prev, hny_builder **out);
/* Reader::open (src/reader.rs:387-431): a stored graph (every Links record + Metadata.entry_points,
 * max_level) and its items loaded into HBM as they are, for hny_builder_search_knn / hny_builder_nns.
 * Nothing is inserted (an empty hny_builder_create_incremental would re-link the old entry points,
 * like an empty Writer::build does, hnsw.rs:267). */
int hny_builder_load(const hny_build_opts *opts, const hny_items *items, const hny_prev_graph *prev,
                     hny_builder **out);
int hny_builder_fill_gaps(hny_builder *b); /* fill_gaps_from_deleted, hnsw.rs:187, 334-415 */
int hny_builder_reset(hny_builder *b); /* empty graph again, vectors stay resident in HBM */
int hny_builder_next_batch(hny_builder *b, hny_batch *out);
/* search + prune (walk_layer hnsw.rs:460-518, robust_prune :565-597) for members [lo, hi) of the
 * current batch against the frozen graph; results go to sel_dev (device memory, count *
 * sel_stride_u64 u64 words for the WHOLE batch; NULL = internal buffer) */
int hny_builder_search(hny_builder *b, uint32_t lo, uint32_t hi, void *sel_dev);
/* add_link for every selected pair, both directions (hnsw.rs:316-324, 523-560), in batch order */
int hny_builder_apply(hny_builder *b, const void *sel_dev);
/* hny_builder_apply in three steps, for the multi-GPU driver: most targets just append their new
 * links (cheap, done by every replica in _begin); a target whose list overflows re-runs
 * robust_prune on it (hnsw.rs:547-552) — these "deferred" targets, n_deferred of them in an order
 * that is the same on every replica, are split across the ranks by _deferred (rank r takes every
 * world-th one and writes the finished lists to its part of exch_dev: world * ceil(n_deferred /
 * world) records of hny_builder_exch_stride_u64 words), the parts are all-gathered by the caller,
 * and _merge installs the other ranks' lists and closes the batch.  world == 1: exch may be NULL. */
int hny_builder_apply_begin(hny_builder *b, const void *sel_dev, uint32_t *n_deferred);
int hny_builder_apply_deferred(hny_builder *b, uint32_t rank, uint32_t world, void *exch_dev);
int hny_builder_apply_merge(hny_builder *b, const void *exch_all_dev, uint32_t rank, uint32_t world);
uint32_t hny_builder_exch_stride_u64(const hny_builder *b);
int hny_builder_sync(hny_builder *b);
/* the HIP stream (hipStream_t) every step above is enqueued on: a multi-GPU driver puts its
 * collectives on it, so that search -> all-gather -> apply need no host synchronisation */
void *hny_builder_stream(hny_builder *b);
/* record a HIP event pair around every kernel family launch (bench roofline accounting) */
int hny_builder_set_profiling(hny_builder *b, int on);
int hny_builder_finish(hny_builder *b, hny_graph **out);
void hny_builder_destroy(hny_builder *b);
/* get_random_level (hnsw.rs:113-119) for n items in ascending id order, as the reference draws
 * them from StdRng::seed_from_u64(seed); what hny_build uses when items.levels == NULL */
int hny_draw_levels(uint64_t seed, uint32_t M, uint64_t n, uint8_t *out);
/* the same from StdRng::from_seed(seed) (the reference's test rng, src/tests/mod.rs:145-147) after
 * `skip` earlier draws of the same generator (each level costs one u32), for a caller that keeps
 * one rng across several builds like `writer.builder(&mut rng)` does */
int hny_draw_levels_from_seed(const uint8_t seed[32], uint64_t skip, uint32_t M, uint64_t n, uint8_t *out);
/* the schedule: batch size when n_done items are already inserted */
uint32_t hny_batch_size(double batch_frac, uint32_t batch_max, uint64_t n_done);
/* what batch_max = 0 selects for an index of n_items: the largest power of two <= n_items / 12, at
 * least 65 536 (a batch stays the same small fraction of the index as it grows) */
uint32_t hny_default_batch_max(uint64_t n_items);

/* ---- resident multi-GPU build: replaces the rayon pool of the insert loop (hnsw.rs:172-185 over
 * src/parallel.rs:11-45) for a caller that builds more than once on the same vectors — the stepwise
 * counterpart of hny_build(n_gpus > 1), like hny_builder_* is of hny_build.  _create uploads a full
 * replica of the items to every GPU of opts->devices (NULL: 0 .. n_gpus-1; the uploads run in parallel)
 * and sets up the RCCL communicator; every _run is one complete fresh build (graph reset -> every batch
 * sharded across the GPUs, two all-gathers per batch on the builders' streams -> records exported by rank
 * 0) and returns the bytes hny_build(n_gpus = 1) returns.  Callbacks of `opts` fire on the calling
 * thread.  Fresh builds only (an incremental build is one-shot: hny_build_incremental). ---- */
typedef struct hny_multi_builder hny_multi_builder;
int hny_multi_builder_create(const hny_build_opts *opts, const hny_items *items, hny_multi_builder **out);
int hny_multi_builder_run(hny_multi_builder *mb, hny_graph **out);
int hny_multi_builder_set_profiling(hny_multi_builder *mb, int on); /* per-kernel times of rank 0's shard */
uint32_t hny_multi_builder_world(const hny_multi_builder *mb);       /* replicas = GPUs in use */
uint64_t hny_multi_builder_collectives(const hny_multi_builder *mb); /* all-gathers of the last run */
/* the replica of `rank` (library-owned), e.g. for hny_builder_search_knn after a run */
hny_builder *hny_multi_builder_replica(hny_multi_builder *mb, uint32_t rank);
void hny_multi_builder_destroy(hny_multi_builder *mb);

/* ---- distances (trait Distance::distance, src/distance/mod.rs:41): pairs of stored items ---- */
int hny_builder_distances(hny_builder *b, uint64_t n_pairs, const uint32_t *slot_a,
                          const uint32_t *slot_b, float *out);

/* ---- search: Reader::nns().by_vector (src/reader.rs:132-148, 642-665, 722-800) on the graph
 * held by the builder (after the build, before destroy).  Queries are codec bytes + headers. */
/* ef_search (and k) up to 65 535: result sets of up to 4 096 entries live in the walk's LDS, larger ones in
 * HBM — the reference's own tests search with ef_search = n up to 9 999 (src/tests/reader.rs:82-98).  A query
 * whose walk overflows its tie pool (short codes: a handful of distinct distances) is searched again on heaps
 * in HBM and returns the same hits: no input fails for it. */
int hny_builder_search_knn(hny_builder *b, uint64_t n_queries, const void *qvectors, size_t qstride,
                           const void *qheaders, uint32_t k, uint32_t ef_search, uint32_t *out_ids,
                           float *out_dists, uint32_t *out_counts);

/* ---- search with the rest of QueryBuilder (src/reader.rs:60-262): `.candidates(&bitmap)`
 * (:200-203), `.linear_below()` / `.linear_below_ratio()` (:234-262) and `by_item` (:81-90).
 * With candidates fewer than linear_below (and within the ratio) the candidates are ranked by
 * brute force (should_linear_scan :621-640, brute_force_search :667-711); otherwise the HNSW walk
 * runs with the filter applied to the result heap only (Visitor::visit :301-369), then the
 * exhaustive fallback (:771-795 / :864-890).  query_items != NULL = by_item: one item id per query
 * instead of qvectors/qheaders (which may then be NULL); out_counts[i] = HNY_NNS_NONE where the
 * reference returns None (unknown item, or nothing can ever match, :822-826). */
#define HNY_NNS_NONE 0xFFFFFFFFu
typedef struct {
  uint32_t k;                 /* Reader::nns(count) */
  uint32_t ef_search;         /* default 100 (reader.rs:23); max(ef_search, k) <= 65 535 (a linear scan returns at most 4 095 hits) */
  int32_t has_candidates;     /* .candidates() given (it may be empty) */
  const uint32_t *candidates; /* item ids, any order, duplicates and unknown ids allowed */
  uint64_t n_candidates;
  uint32_t linear_below;      /* default 1000 (reader.rs:28) */
  float linear_below_ratio;   /* default 1.0 (reader.rs:31); must be in [0, 1] (:253-256) */
  /* by_vector_with_cancellation / by_item_with_cancellation (reader.rs:108-119, 167-186; the probe
   * sits in Visitor::visit, :333): `cancel` is polled by the calling thread while the batch runs
   * (every ~0.2 ms); once it returns non-zero the kernels stop taking queries from the work queue.
   * Queries already finished keep their results, the others report 0 hits (Searched { nns: what was
   * found so far, did_cancel: true }), and *did_cancel (if given) is set to 1.  NULL = no probe. */
  int (*cancel)(void *);
  void *cancel_ctx;
  int32_t *did_cancel;
} hny_query_opts;
int hny_builder_nns(hny_builder *b, const hny_query_opts *opts, uint64_t n_queries, const void *qvectors,
                    size_t qstride, const void *qheaders, const uint32_t *query_items, uint32_t *out_ids,
                    float *out_dists, uint32_t *out_counts);

/* ---- codecs: UnalignedVectorCodec::from_slice (src/unaligned_vector/{f32,binary,
 * binary_quantized}.rs) + Distance::new_header (cosine.rs:36-38 ...) ---- */
size_t hny_vector_bytes(int32_t metric, uint32_t dim);
size_t hny_header_bytes(int32_t metric);
int hny_encode_vectors(int32_t metric, uint32_t dim, uint64_t n, const float *vectors,
                       void *out_codes, void *out_headers);
/* the same on the GPU (bulk ingest): bit codecs by ballot, Cosine norms in the reference's x86
 * summation order (simple_avx.rs / simple_sse.rs / scalar) — byte-identical to the host path */
int hny_encode_vectors_gpu(int32_t metric, uint32_t dim, uint64_t n, const float *vectors,
                           void *out_codes, void *out_headers, int32_t device);

/* Diagnostic: runs the distance kernels' cross-lane primitives (DPP moves, v_permlane16/32_swap) next
 * to the generic __shfl_xor on one wave of `device` (-1 = current).  HNY_OK when every lane agrees;
 * HNY_ERR_DEVICE otherwise, with bit (5 * log2(offset) + check) of mismatch64[lane] set (may be
 * NULL).  No reference counterpart: the reference's reductions are the AVX/SSE horizontal sums of
 * src/spaces/simple_avx.rs:69-110. */
int hny_selftest_lane_ops(int32_t device, uint32_t *mismatch64);

/* ---- on-disk records (key.rs:54-82, node.rs:130-174, metadata.rs:22-73, version.rs:33-60) ---- */
typedef int (*hny_kv_sink)(void *ctx, const uint8_t *key, size_t key_len, const uint8_t *val,
                           size_t val_len);
/* emits, in LMDB key order, exactly the records Writer::build leaves behind for a fresh index:
 * Metadata, Version, every Links record, and (with_items) every Item record */
int hny_encode_kv(const hny_graph *g, const hny_build_opts *opts, const hny_items *items,
                  uint16_t index, int with_items, hny_kv_sink sink, void *ctx);

/* ---- LMDB writeback: the KV stream above as an LMDB environment file
 * (`<dir>/data.mdb`) that heed/LMDB can open — what `db.put` produces in the reference's write
 * loop (hnsw.rs:195-213, writer.rs:462-480, 585-600) through heed 0.22 / LMDB 0.9 (third party,
 * not under /root/reference: the file format of mdb.c is restated).  The writer is
 * a bulk loader: keys must arrive in strictly ascending LMDB order (mdb_cmp_memn: bytewise, then
 * length), which is the order hny_encode_kv emits.  The result is the state after ONE committed
 * write transaction on a fresh environment: meta page 1 holds txnid 1, the free DB is empty, the
 * records live in the main DB (name == NULL, `env.create_database(wtxn, None)`,
 * src/tests/mod.rs:111) or in a named sub-DB (python.rs:75).  HNY_ERR_IO on a failed write. */
#define HNY_ERR_IO (-9)
typedef struct hny_lmdb_writer hny_lmdb_writer;
int hny_lmdb_writer_open(const char *data_mdb_path, uint32_t page_size /* 0 = 4096 */,
                         uint64_t map_size /* 0 = file size */, const char *db_name /* or NULL */,
                         hny_lmdb_writer **out);
/* same signature as hny_kv_sink with ctx = the writer: pass it straight to hny_encode_kv */
int hny_lmdb_writer_put(void *writer, const uint8_t *key, size_t key_len, const uint8_t *val,
                        size_t val_len);
/* writes the branch pages and both meta pages, closes the file, frees the writer (also on error) */
int hny_lmdb_writer_finish(hny_lmdb_writer *w);
void hny_lmdb_writer_abort(hny_lmdb_writer *w);

/* read side (what heed's Database::get / iter do through mdb_get / mdb_cursor_get on a RoTxn):
 * used to load an index written by the writer above — or by the reference — back into HBM */
typedef struct hny_lmdb_env hny_lmdb_env;
typedef struct {
  uint32_t page_size, depth;
  uint64_t branch_pages, leaf_pages, overflow_pages, entries, last_pgno, txnid, map_size;
} hny_lmdb_stat;
int hny_lmdb_open(const char *data_mdb_path, const char *db_name /* or NULL */, hny_lmdb_env **out);
int hny_lmdb_stat_get(const hny_lmdb_env *e, hny_lmdb_stat *out);
/* mdb_get: *val points into the mapping (valid until hny_lmdb_close); returns 1 if found, 0 if
 * not (MDB_NOTFOUND), < 0 on a corrupt file */
int hny_lmdb_get(const hny_lmdb_env *e, const uint8_t *key, size_t key_len, const uint8_t **val,
                 size_t *val_len);
/* cursor walk MDB_SET_RANGE(lo) .. MDB_NEXT while key <= hi (NULL bounds = open), in key order;
 * also checks the page invariants it passes (flags, bounds, key order, counts) */
int hny_lmdb_scan(const hny_lmdb_env *e, const uint8_t *lo, size_t lo_len, const uint8_t *hi,
                  size_t hi_len, hny_kv_sink sink, void *ctx);
void hny_lmdb_close(hny_lmdb_env *e);

const char *hny_last_error(void);
const char *hny_version(void);

/* sizeof of every public struct as THIS library was compiled, in the order below — a binding that
 * declares the structs itself (Rust #[repr(C)], ctypes, cgo) compares them with its own at start-up,
 * so that a field added here cannot go unnoticed there.  Writes min(n, HNY_ABI_N_STRUCTS) entries,
 * returns HNY_ABI_N_STRUCTS. */
enum {
  HNY_ABI_BUILD_OPTS = 0, /* hny_build_opts */
  HNY_ABI_ITEMS = 1,      /* hny_items */
  HNY_ABI_GRAPH = 2,      /* hny_graph */
  HNY_ABI_PREV_GRAPH = 3, /* hny_prev_graph */
  HNY_ABI_BATCH = 4,      /* hny_batch */
  HNY_ABI_QUERY_OPTS = 5, /* hny_query_opts */
  HNY_ABI_LMDB_STAT = 6,  /* hny_lmdb_stat */
  HNY_ABI_N_STRUCTS = 7
};
uint32_t hny_abi_sizes(uint32_t *out_sizes, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
