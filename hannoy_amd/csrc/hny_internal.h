// hny_internal.h — shared between the host driver (hny_host.cpp) and the gfx950 kernels
// (hny_kernels.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define HNY_SENT 0xFFFFFFFFu      // empty neighbour slot
#define HNY_MAX_CAP 64            // max(M, M0) of the one-lane-per-slot kernels (incremental builds, strict mode)
#define HNY_BIG_CAP 1024          // M0 of fresh builds / loaded graphs: lists are walked 64 slots at a time
#define HNY_MAX_EPS 8192          // max entry points (every item of a small all-level-0 index is one): 32 KB of LDS ids
#define HNY_POOL_CAP 128          // tie pool (DESIGN.md "candidate heap")
#define HNY_MAX_EF 65535         // ef_construction: result sets of ef + 1 entries, in the walk's LDS up to 4 096, in HBM beyond
#define HNY_OP_INVALID 0xFFFFFFFFFFFFFFFFull
// link-op sort key: layer:4 | target:31 | sequence:29 (levels 0..14: M = 4 draws up to level 14 before
// its probability drops under the 1e-9 cut of get_default_probas, hnsw.rs:94-110)
#define HNY_SEQ_BITS 29
#define HNY_MAX_LEVEL 14

// metric classes of the inner loop
enum { MC_DOT = 0, MC_L2 = 1, MC_L1 = 2, MC_BIN = 3 };

// device-side error / statistics words
enum {
  ST_EVALS_WALK = 0,
  ST_EVALS_PRUNE = 1,
  ST_EVALS_APPLY = 2,
  ST_LINKS = 3,
  ST_POOL_OVERFLOW = 4,
  ST_LOG_OVERFLOW = 5,
  ST_ERR_RES_OVERFLOW = 6,
  ST_ERR_ITER = 7,
  ST_ERR_GAPS_OVERFLOW = 8,
#ifdef HNY_PHASE_CLOCKS // diagnostic build (HNY_CFLAGS=-DHNY_PHASE_CLOCKS): wave cycles per walk phase
  ST_PH_POP = 16, ST_PH_LIST = 17, ST_PH_DIST = 18, ST_PH_INSERT = 19, ST_PH_EXPANSIONS = 20, ST_PH_REST = 21,
  // walk_layer_short only: the visited round trip apart from the list fetch; lanes that asked the visited set,
  // accepted keys, expansions that accepted any, tie-pool scans, expansions with nothing new
  ST_PH_VIS = 22, ST_PH_NASK = 23, ST_PH_NACC = 24, ST_PH_NMERGE = 25, ST_PH_NPOOL = 26, ST_PH_NONEW = 27,
  ST_COUNT = 32
#else
  ST_COUNT = 16
#endif
};

struct GraphDev {
  // items (resident in HBM for the whole build)
  u32 n;
  int metric;       // hny_metric
  int mclass;       // MC_*
  u32 n16;          // 16-byte units per row that carry data
  u32 row_stride;   // bytes, multiple of 16
  u32 bin_bits;     // binary codecs: padded dims = vector bytes * 8
  const unsigned char *rows;
  const float *norms; // header norm (cosine / bq cosine), else unused
  const unsigned char *level;
  const int *upper_idx; // slot -> index among items with level >= 1, or -1
  // graph
  u32 M, M0, max_level, n_upper;
  float alpha;
  u32 *l0_ids;   // [n][M0], HNY_SENT beyond the count
  float *l0_dist; // [n][M0]
  u32 *l0_cnt;   // [n] low 16 bits = count, bit 31 = frozen (full and self-pruned to full)
  u32 *up_ids;   // [n_upper][up_layers][M]
  float *up_dist;
  u32 *up_cnt;   // [n_upper][up_layers]
  u64 *stats;    // [ST_COUNT]
  u32 dim;       // user dimensions (f32 metrics)
  int x86_order; // 1: f32 distances in the reference's x86 summation order (strict mode)
  u32 up_layers; // upper layers that have storage (== max(max_level,1) for a fresh build)
  // incremental builds: the previous graph as stored in LMDB (read-only) + which items still exist
  int incremental;
  const u32 *d0_ids;            // [n][M0] old layer-0 Links, ascending, HNY_SENT padded
  const u32 *du_ids;            // [n_upper][up_layers][M]
  const unsigned char *has_vec; // [n] 0 = deleted item (no Item record any more)
  float bin_inv;                // 1 / bin_bits when bin_bits is a power of two (exact), else 0
};

struct WalkArgs {
  // queries: build mode -> stored items q_slots[member]; knn mode -> external rows
  const u32 *q_slots;
  const unsigned char *q_rows; // knn mode (else null)
  const float *q_norms;
  u32 q_stride;
  u32 lo, hi;       // member range
  u32 layer;        // layer of the ef walk
  u32 ef;
  int first;        // 1: start from the entry points and descend greedily to `layer`
  int reader_mode;  // Reader::hnsw_search visited-set semantics (reader.rs:731-743)
  const u32 *entry_points;
  u32 n_entry_points;
  // eps source when !first: selection of layer+1
  const u64 *sel;
  u32 sel_stride, cap_sel, batch_level;
  // output: sorted (dist bits << 32 | slot) lists
  u64 *cand;
  u32 *cand_n;
  u32 rcap;         // capacity of res in LDS and of cand rows
  // per-block workspace
  u32 *bits;        // [grid][bits_words]
  u32 bits_words;
  u32 *vlog;        // [grid][log_cap]
  u32 log_cap;
  u32 *queue;       // work counter, zeroed before the launch
  // locality-ordered processing (DESIGN.md "processing order"): results are stored per member, so
  // the ORDER in which members are processed is free.  descend_only: run the greedy descent down to
  // layer+1, write the closest node (eps_out) and a hierarchical locality key (key_out), no ef walk.
  // eps_in: start the ef walk at `layer` from eps_in[m].  perm: processing order (queue index ->
  // member).
  int descend_only;
  u32 *eps_out;
  u64 *key_out;
  const u32 *eps_in;
  const u64 *perm;
  u32 knn_k;        // reader mode: wanted hits (exhaustive fallback below that, reader.rs:771-795)
  u32 knn_ef;       // reader mode: opt.ef of the query builder
  u32 vis_slots;    // LDS visited table entries per wave (0: HBM bitset only)
  u32 eps_cap;      // LDS entries of the eps array: >= max(64, n_entry_points)
  u32 key_base;     // key_out is indexed by member - key_base (== lo, except in a retry launch)
  const u32 *cancel; // reader mode: pinned host word, non-zero = take no further query (reader.rs:333)
  u64 *res_global;   // general kernels: result sets of more than 4 096 entries live in HBM, [grid][rcap] (else null: LDS)
  u32 xcd_tile;      // != 0: `queue` is 8 counters, one per XCD; tile k of xcd_tile members belongs to XCD k % 8
  // Tie-pool overflow (build walks): k_walk lists the member here instead of failing the build;
  // k_walk_heap then redoes exactly those members with walk_layer's own data structures — `candidates` and
  // `res` as real heaps in HBM (heap_c / heap_r, [grid][cap] each), no pool, nothing to overflow but memory.
  u32 *pool_retry;   // members whose walk overflowed the 128-slot tie pool (null: count it as an error)
  u32 *n_pool_retry;
  // k_walk_heap, first tier (many blocks, heaps of heap_c_cap < n entries): a member that outgrows its heap is
  // listed here and walked once more by the second tier (few blocks, heaps that hold every item); null = the last
  // tier: an overflow is an error
  u32 *pool_retry2;
  u32 *n_pool_retry2;
  u64 *heap_c, *heap_r;
  u32 heap_c_cap, heap_r_cap;
  u32 force_pool;    // test hook (HNY_POOL_FORCE_RETRY=n): treat every member with m % n == 0 as overflowed
  u32 pool_flag;     // reader mode: a query whose tie pool overflowed reports cand_n = 0xFFFFFFFE (the host
                     // repeats it on the heap-queue searcher) instead of counting an error
  // short-row build walks (walk_layer_short): LDS visited table of vis_buckets x 4 x 16-bit remainders, 0 = none
  // (VisB in hny_kernels.hip); vis_magic = floor(2^vis_shift / vis_buckets) + 1 with vis_shift = 31 + floor(log2 buckets)
  // (the magic then fits 32 bits and x div buckets == (x * magic) >> vis_shift for x < 2^30); vis_smask = 2^k - 1 >= n - 1
  u32 vis_buckets, vis_magic, vis_shift, vis_smask;
  u32 rb_one;        // build walks on rows <= 512 B: no result set of this builder exceeds 64 entries -> one-chunk register beam
};

// Reader::nns with a candidates filter and/or by_item (reader.rs:301-369 with `candidates`, 642-711,
// 809-896).  The search queue and `res` decouple here (the queue takes points the filter rejects), so
// the queue is a real heap: 64-ary, in HBM, one per resident wave.
struct NnsArgs {
  const u32 *q_slots;          // by_item: slot of each query item
  const unsigned char *q_rows; // by_vector: query rows / header norms
  const float *q_norms;
  u32 q_stride;
  const u32 *members;          // queue index -> query index (retry pass), or null = identity
  u32 n_members;
  const u32 *filter;           // candidates as a bitset over slots, or null
  int by_item;
  u32 k, ef_main, ef_opt;      // count, max(ef, count), opt.ef
  const u32 *entry_points;
  u32 n_entry_points;
  u64 *cand;                   // [n_queries][rcap] dist bits << 32 | slot, ascending
  u32 *cand_n;
  u32 rcap;
  u32 *bits;
  u32 bits_words;
  u32 *vlog;
  u32 log_cap;
  u64 *heap;                   // [grid][heap_cap]
  u32 heap_cap;
  u64 *heap_r;                 // k_nns_heap: `res` as a heap too, [grid][heap_r_cap] (result sets beyond the LDS)
  u32 heap_r_cap;
  u32 *queue;
  u32 vis_slots;               // LDS visited table entries per wave
  u32 eps_cap;                 // LDS entries of the eps array
  u32 *status;                 // [n_queries] 1 = the heap overflowed: run again with a larger one
  // brute_force_search (reader.rs:667-711): the existing candidates, ascending
  const u32 *cand_slots;
  u32 n_cand_slots;
  // cancellation (reader.rs:333): a host word (pinned, mapped) the work-queue loop polls; non-zero =
  // take no further query.  status[] == 2 marks the queries that were never started.
  const u32 *cancel;
};

struct PruneArgs {
  const u32 *q_slots;
  u32 lo, hi;
  u32 layer;
  u32 cap;          // cap chosen from the item's top level (hnsw.rs:317 quirk)
  const u64 *cand;
  const u32 *cand_n;
  u32 rcap;
  u64 *sel;
  u32 sel_stride, cap_sel, batch_level;
  const u64 *perm;  // processing order (index -> member), or null
  u32 list_global;  // general kernels: candidate lists too long for LDS are pruned straight from `cand`
  u32 xcd_tile;     // k_prune_wg, != 0: tiles of xcd_tile consecutive members (locality order) stay on one XCD
};

struct EmitArgs {
  const u32 *q_slots;
  u32 count;
  const u64 *sel;
  u32 sel_stride, cap_sel, batch_level;
  u64 *keys;  // [count * n_layers * cap_sel * 2]
  u64 *vals;
};

struct ApplyArgs {
  const u64 *keys; // sorted
  const u64 *vals;
  u32 n_ops;
  const u32 *seg_start;
  const u32 *n_seg;
  u32 *deferred;   // first op of every segment whose list may overflow (handled by k_apply_wg);
                   // null = k_apply does everything
  u32 *n_deferred;
  // multi-GPU: k_apply_wg takes deferred[i] with i % shard_world == shard_rank (the list is sorted,
  // so every rank sees the same order) and also writes the finished list to exch[(i / shard_world)]:
  // exch_stride u64 words = key (layer << 31 | target), count word, cap x (dist bits << 32 | slot)
  u32 shard_rank, shard_world;
  u64 *exch;
  u32 exch_stride;
};

struct LaunchShape {
  int lpr; // lanes per row: 8,16,32,64
  int nch; // 16-byte chunks per lane: 1,2,3,4,6,8
};

// kernels' host launchers (hny_kernels.hip)
hipError_t hnyk_walk(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid, hipStream_t st);
// the members of a.pool_retry[0 .. *a.n_pool_retry) again, on heaps in HBM (a.queue: its own work counter)
hipError_t hnyk_walk_heap(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid, hipStream_t st);
hipError_t hnyk_prune(const GraphDev &g, const PruneArgs &a, LaunchShape s, int grid, hipStream_t st);
hipError_t hnyk_nns_filtered(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st);
hipError_t hnyk_nns_linear(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st);
hipError_t hnyk_nns_heap(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st);
hipError_t hnyk_emit(const GraphDev &g, const EmitArgs &a, hipStream_t st);
hipError_t hnyk_segments(const u64 *keys, u32 n_ops, u32 *seg_start, u32 *n_seg, hipStream_t st);
hipError_t hnyk_apply(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int grid, hipStream_t st);
// the segments that cannot overflow, one thread each (a.deferred must be set); the rest -> a.deferred
hipError_t hnyk_apply_append(const GraphDev &g, const ApplyArgs &a, hipStream_t st);
hipError_t hnyk_prune_wg(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL, int nw, int grid,
                         hipStream_t st);
// robust_prune for rows <= 512 B: one wave per query, 8 candidates at a time (SL = selected rows staged in LDS)
bool hnyk_prune_n8_ok(const GraphDev &g, const PruneArgs &a, LaunchShape s);
hipError_t hnyk_prune_n8(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL, int grid, hipStream_t st);
// the deferred add_link targets (k_apply_wg's job) for rows <= 512 B: one wave per target
bool hnyk_apply_n8_ok(const GraphDev &g, LaunchShape s);
hipError_t hnyk_apply_n8(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL, int grid, hipStream_t st);
hipError_t hnyk_apply_merge(const GraphDev &g, const u64 *exch, u32 n_def, u32 world, u32 rank, u32 per,
                            u32 stride, hipStream_t st);
hipError_t hnyk_sort_u32(void *temp, size_t &temp_bytes, u32 *in, u32 *out, u32 n, hipStream_t st);
hipError_t hnyk_apply_wg(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL, int grid,
                         hipStream_t st);
hipError_t hnyk_sort_pairs(void *temp, size_t &temp_bytes, u64 *keys_in, u64 *keys_out, u64 *vals_in,
                           u64 *vals_out, u32 n, u32 begin_bit, u32 end_bit, hipStream_t st);
hipError_t hnyk_sort_pairs48(void *temp, size_t &temp_bytes, u64 *keys_in, u64 *keys_out, u64 *vals_in,
                             u64 *vals_out, u32 n, hipStream_t st);
hipError_t hnyk_iota_u64(u64 *p, u32 base, u32 n, hipStream_t st);
hipError_t hnyk_take_topk(const u64 *cand, const u32 *cand_n, u32 rcap, u32 k, u32 n, u64 *out,
                          hipStream_t st);
hipError_t hnyk_lane_selftest(u32 *out64, hipStream_t st);
hipError_t hnyk_pair_distances(const GraphDev &g, const u32 *a, const u32 *b, u32 n, float *out,
                               LaunchShape s, hipStream_t st);
hipError_t hnyk_fill_u32(u32 *p, u32 v, size_t n, hipStream_t st);
// fill_gaps_from_deleted (hnsw.rs:334-415) for the old records rec[i] = layer << 31 | slot
hipError_t hnyk_fill_gaps_wg(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                             u32 *bitmaps, u32 words, u32 *bm, u32 maxb, u64 *keys, u64 *sorted, int SL, int grid,
                             LaunchShape s, hipStream_t st);
hipError_t hnyk_fill_gaps(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                          LaunchShape s, hipStream_t st);
hipError_t hnyk_finalize_lists(u32 *ids, u32 *cnt_out, u32 n_lists, u32 cap, hipStream_t st);
size_t hnyk_walk_lds_bytes(u32 rcap, u32 eps_cap);
// the build kernels specialised for metric N-1 (hny_kernels.hip compiled with -DHNY_PART=N)
#define HNY_DECL_SP(N)                                                                                      \
  hipError_t hnyk_walk_sp##N(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid, hipStream_t st); \
  hipError_t hnyk_prune_wg_sp##N(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL, int nw,      \
                                 int grid, hipStream_t st);                                                 \
  hipError_t hnyk_prune_n8_sp##N(const GraphDev &g, const PruneArgs &a, int lpro, int SL, int grid,         \
                                 hipStream_t st);                                                           \
  hipError_t hnyk_apply_n8_sp##N(const GraphDev &g, const ApplyArgs &a, int lpro, int SL, int grid,         \
                                 hipStream_t st);                                                           \
  hipError_t hnyk_apply_sp##N(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int grid,               \
                              hipStream_t st);                                                              \
  hipError_t hnyk_apply_wg_sp##N(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL, int grid,    \
                                 hipStream_t st);
HNY_DECL_SP(1) HNY_DECL_SP(2) HNY_DECL_SP(3) HNY_DECL_SP(4) HNY_DECL_SP(5) HNY_DECL_SP(6) HNY_DECL_SP(7)
#undef HNY_DECL_SP
hipError_t hnyk_norms_x86(const float *v, u32 dim, u64 n, float *out, hipStream_t st);
hipError_t hnyk_quantize(const float *v, u32 dim, u64 n, int binary_codec, u64 *out, hipStream_t st);
