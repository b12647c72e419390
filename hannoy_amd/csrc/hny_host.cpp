// hny_host.cpp — host side of the C ABI in include/hannoy_amd.h: validation, level assignment,
// HBM residency, the batch-synchronous build driver, export, codecs and on-disk record encoders.
// Mirrors HnswBuilder::build (/root/reference/src/hnsw.rs:122-216) around the gfx950 kernels in
// hny_kernels.hip.  No CPU fallback: without a device every computing entry point fails.
#include "../../include/hannoy_amd.h"
#include "hny_internal.h"
#include "hny_rust_sort.h"

#include <malloc.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(x)                                                                           \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess)                                                                    \
      return fail(e_ == hipErrorOutOfMemory ? HNY_ERR_OOM : HNY_ERR_NO_DEVICE, "%s: %s", #x, \
                  hipGetErrorString(e_));                                                    \
  } while (0)

} // namespace
// error reporting for the other host translation units (hny_lmdb.cpp)
int hny_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }
// hny_build_opts.schedule (HNY_SCHED_*): LEVEL_ORDER_ID = items of one level inserted in ascending id order
// (rounds 1-2) instead of the order Rust's sort_unstable_by leaves them in (hny_rust_sort.h); batches of more than
// one member take the items of a level group in a fixed pseudo-random order (hny_rust_sort.h) unless NO_SHUFFLE
// asks for consecutive runs of the reference's order (rounds 1-2)
static bool shuffle_groups(const hny_build_opts &o, uint32_t batch_max) {
  return batch_max != 1u && !(o.schedule & HNY_SCHED_NO_SHUFFLE);
}
static bool level_order_by_id(const hny_build_opts &o) { return (o.schedule & HNY_SCHED_LEVEL_ORDER_ID) != 0; }
namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool is_binary(int metric) { return metric >= HNY_HAMMING; }
size_t vec_bytes(int metric, uint32_t dim) {
  return is_binary(metric) ? (size_t)((dim + 63) / 64) * 8 : (size_t)dim * 4;
}
size_t hdr_bytes(int metric) { return metric == HNY_HAMMING ? 8 : 4; }
uint32_t pow2ceil(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// ---- get_default_probas (hnsw.rs:94-110) ----
std::vector<float> level_probas(uint32_t M) {
  std::vector<float> p;
  float level_factor = 1.0f / logf((float)M + 1.1920929e-07f);
  for (uint32_t level = 0;; level++) {
    float proba = expf((float)level * (-1.0f / level_factor)) * (1.0f - expf(-1.0f / level_factor));
    if (proba < 1e-09f) break;
    p.push_back(proba);
  }
  return p;
}
// rand 0.8.5 StdRng = ChaCha12 block generator (key = 32-byte seed, 64-bit counter, stream 0)
class StdRngChaCha12 {
 public:
  explicit StdRngChaCha12(uint64_t seed_u64) { // SeedableRng::seed_from_u64 (PCG32 expansion)
    uint64_t st = seed_u64;
    for (int c = 0; c < 8; c++) {
      st = st * 6364136223846793005ull + 11634580027462260723ull;
      uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
      key_[c] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
  }
  explicit StdRngChaCha12(const uint8_t seed[32]) { // SeedableRng::from_seed: little-endian words
    for (int c = 0; c < 8; c++)
      key_[c] = (uint32_t)seed[4 * c] | ((uint32_t)seed[4 * c + 1] << 8) | ((uint32_t)seed[4 * c + 2] << 16) |
                ((uint32_t)seed[4 * c + 3] << 24);
  }
  uint32_t next_u32() {
    if (pos_ == 16) block();
    return out_[pos_++];
  }

 private:
  static uint32_t rl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
  void block() {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key_[0], key_[1], key_[2],
                       key_[3], key_[4], key_[5], key_[6], key_[7], (uint32_t)ctr_,
                       (uint32_t)(ctr_ >> 32), 0u, 0u};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = in[i];
    auto q = [&](int a, int b2, int c, int d) {
      x[a] += x[b2]; x[d] = rl(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b2] = rl(x[b2] ^ x[c], 12);
      x[a] += x[b2]; x[d] = rl(x[d] ^ x[a], 8);
      x[c] += x[d]; x[b2] = rl(x[b2] ^ x[c], 7);
    };
    for (int r = 0; r < 12; r += 2) {
      q(0, 4, 8, 12); q(1, 5, 9, 13); q(2, 6, 10, 14); q(3, 7, 11, 15);
      q(0, 5, 10, 15); q(1, 6, 11, 12); q(2, 7, 8, 13); q(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out_[i] = x[i] + in[i];
    ctr_++;
    pos_ = 0;
  }
  uint32_t key_[8], out_[16];
  uint64_t ctr_ = 0;
  int pos_ = 16;
};

// get_random_level (hnsw.rs:113-119): WeightedIndex<f32>::new(probas).sample(rng), one draw per
// item in ascending id order (hnsw.rs:142-149).  Reproduces what the reference draws from
// StdRng::seed_from_u64(seed) (the Python binding's rng, python.rs:261).
static void draw_levels_rng(StdRngChaCha12 &rng, uint64_t skip, uint32_t M, uint32_t n, uint8_t *out) {
  std::vector<float> p = level_probas(M);
  std::vector<float> cum; // running totals, last weight excluded
  float total = p[0];
  for (size_t i = 1; i < p.size(); i++) {
    cum.push_back(total);
    total = total + p[i];
  }
  // UniformFloat<f32>::new(0, total): shrink the scale until the largest sample stays below total
  uint32_t mb = (0xFFFFFFFFu >> 9) | 0x3F800000u;
  float max_rand;
  memcpy(&max_rand, &mb, 4);
  max_rand -= 1.0f;
  float scale = total;
  while (scale * max_rand + 0.0f >= total) {
    uint32_t sb;
    memcpy(&sb, &scale, 4);
    sb -= 1;
    memcpy(&scale, &sb, 4);
  }
  for (uint64_t i = 0; i < skip; i++) (void)rng.next_u32(); // one u32 per earlier draw
  for (uint32_t s = 0; s < n; s++) {
    uint32_t u = (rng.next_u32() >> 9) | 0x3F800000u;
    float v;
    memcpy(&v, &u, 4);
    float x = (v - 1.0f) * scale + 0.0f;
    size_t l = 0;
    while (l < cum.size() && cum[l] <= x) l++; // partition_point(|w| w <= x)
    out[s] = (uint8_t)l;
  }
}
void draw_levels(uint64_t seed, uint32_t M, uint32_t n, uint8_t *out) {
  StdRngChaCha12 rng(seed);
  draw_levels_rng(rng, 0, M, n, out);
}

// ---- f32 dot in the reference's x86 order, for Distance::new_header (cosine.rs:36-38,58-60):
// 32 fma partials + hsum tree (simple_avx.rs:8-13,69-110), 16 unfused partials for 16 <= n < 32
// (simple_sse.rs:64-110), scalar below (simple.rs:81-83) ----
float hsum8(const float *x) {
  float a0 = x[4] + x[0], a1 = x[5] + x[1], a2 = x[6] + x[2], a3 = x[7] + x[3];
  float b0 = a0 + a2, b1 = a1 + a3;
  return b0 + b1;
}
float hsum4(const float *x) {
  float b0 = x[0] + x[2], b1 = x[1] + x[3];
  return b0 + b1;
}
float dot_x86_order(const float *a, const float *b, size_t n) {
  if (n >= 32) {
    size_t m = n - n % 32;
    float acc[32] = {0};
    for (size_t i = 0; i < m; i += 32)
      for (int j = 0; j < 32; j++) acc[j] = fmaf(a[i + j], b[i + j], acc[j]);
    float r = hsum8(acc) + hsum8(acc + 8) + hsum8(acc + 16) + hsum8(acc + 24);
    for (size_t i = m; i < n; i++) {
      float p = a[i] * b[i];
      r += p;
    }
    return r;
  }
  if (n >= 16) {
    size_t m = n - n % 16;
    float acc[16] = {0};
    for (size_t i = 0; i < m; i += 16)
      for (int j = 0; j < 16; j++) {
        float p = a[i + j] * b[i + j];
        acc[j] = p + acc[j];
      }
    float r = hsum4(acc) + hsum4(acc + 4) + hsum4(acc + 8) + hsum4(acc + 12);
    for (size_t i = m; i < n; i++) {
      float p = a[i] * b[i];
      r += p;
    }
    return r;
  }
  float s = 0.f;
  for (size_t i = 0; i < n; i++) {
    float p = a[i] * b[i];
    s = s + p;
  }
  return s;
}

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (!count) return hipSuccess;
    return hipMalloc((void **)&p, count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

} // namespace

struct hny_builder {
  hny_build_opts o{};
  uint32_t n = 0;
  std::vector<uint32_t> ids;
  std::vector<uint8_t> level;
  std::vector<uint32_t> order;        // insertion order (slots), level desc, id asc inside a level
  std::vector<uint8_t> order_level;   // level of each insertion (an item can be re-inserted)
  std::vector<int8_t> ins_level;      // highest level a slot is inserted at in this build, -1 = none
  std::vector<uint16_t> old_mask;     // incremental: bit l = an old Links record (slot, l) exists
  std::vector<uint8_t> deleted;       // incremental: slot is in to_delete
  std::vector<u64> old_recs;          // incremental: surviving old records, layer << 31 | slot
  bool incremental = false;
  uint64_t n_done0 = 0;
  uint32_t up_layers = 1;
  std::vector<uint32_t> entry_points; // slots ascending
  // export: which (item, layer) records exist is fixed for the builder's life (finish() reuses it)
  std::vector<uint64_t> rec_first;
  std::vector<uint32_t> rec_item_t;
  std::vector<uint8_t> rec_layer_t;
  std::vector<int32_t> upper_idx;
  uint32_t max_level = 0, n_upper = 0;
  LaunchShape shape{64, 1};
  double frac = 0.0;
  uint32_t bmax = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> sync_evs;       // cross-stream dependencies (no timing)
  size_t sync_used = 0;
  // schedule state
  size_t pos = 0;
  uint64_t n_done = 0, n_batches = 0;
  hny_batch cur{};
  bool in_batch = false, finalized = false;
  // device memory
  GraphDev g{};
  DevBuf<unsigned char> d_rows, d_level;
  DevBuf<float> d_norms, d_l0_dist, d_up_dist;
  DevBuf<int> d_upper_idx;
  DevBuf<u32> d_l0_ids, d_l0_cnt, d_up_ids, d_up_cnt, d_order, d_eps, d_bits, d_vlog, d_cand_n,
      d_seg_start, d_nseg, d_deferred, d_deferred_b, d_fin_cnt0, d_fin_cntu, d_d0_ids, d_du_ids;
  DevBuf<unsigned char> d_has_vec, d_deleted;
  DevBuf<u64> d_old_recs, d_lkey_a, d_lkey_b, d_perm_a, d_perm_b;
  // fill_gaps_from_deleted on lists of more than 64 slots (k_fill_gaps_wg): per-block scratch in HBM
  DevBuf<u32> d_gap_bitmap, d_gap_bm;
  DevBuf<u64> d_gap_keys, d_gap_sorted;
  int gap_grid = 0;
  DevBuf<u32> d_eps0;
  // k_walk_heap (walk_layer on heaps in HBM) for the members whose walk overflowed its tie pool: the list
  // of those members, a (count, work counter) pair per walk launch of a search call, the heaps
  DevBuf<u32> d_pool_retry, d_pool_retry2, d_pool_ctr;
  DevBuf<u64> d_heap_c, d_heap_r;
  uint32_t heap_grid = 0, heap_c_cap = 0, heap_r_cap = 0, pool_ctr_used = 0;
  uint32_t heap_grid1 = 0, heap_c_cap1 = 0; // first tier of the build's retry path: many blocks, small heaps (0: one tier)
  u64 *heap_c_ptr = nullptr; // d_ops (borrowed while a batch is searched) or d_heap_c
  // the four large arrays of the hny_graph this build will export, allocated and touched page by page on a helper
  // thread WHILE the device builds (the host is idle then): finish() would otherwise pay the first touch of up to
  // 1.3 GB of fresh pages (C4: 45 ms, C5: 25 ms of the step).  Owned by the builder until finish() hands them to
  // the graph; hny_graph_free frees them plainly (rounds 3-4 kept them in a process-wide cache instead, which
  // only helped the second build of a loop and held ~1.3 GB for the life of the process).
  struct ExportBufs {
    void *p[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    std::thread th;
    void join() {
      if (th.joinable()) th.join();
    }
    void drop() {
      join();
      for (int i = 0; i < 4; i++) {
        free(p[i]);
        p[i] = nullptr;
        cap[i] = 0;
      }
    }
    ~ExportBufs() { drop(); }
  } xbuf;
  bool will_export = true;       // hny_multi.cpp: only rank 0 exports
  uint64_t nrec_bound = 0, nbr_bound = 0;
  bool locality = true;
  u32 *h_l0 = nullptr, *h_up = nullptr, *h_cnt0 = nullptr, *h_cntu = nullptr; // pinned staging
  DevBuf<u64> d_stats, d_stats_scratch, d_sel, d_cand, d_res_global;
  // the link-op arrays of phase 2 (emit -> sort -> segments -> apply): four views of ONE allocation, because
  // phase 1's safety net (k_walk_heap's `candidates` heaps) borrows the whole of it — the two phases of a batch
  // never overlap on the builder's stream, and the heaps are the only large thing the retry path needs
  DevBuf<u64> d_ops;
  struct { u64 *p = nullptr; size_t n = 0; } d_keys_a, d_keys_b, d_vals_a, d_vals_b;
  DevBuf<unsigned char> d_sort_tmp;
  size_t sort_tmp_bytes = 0;
  uint32_t walk_slots = 0, bits_words = 0, log_cap = 0, rcap = 0, max_batch = 0;
  uint64_t top_layer_nodes = 0; // see res_capacity
  int stage_rows = 0;      // selected rows staged in LDS by the workgroup prune kernels
  u32 cur_n_ops = 0, cur_n_def = 0; // of the batch being applied
  bool apply_open = false;          // between hny_builder_apply_begin and _merge
  int vis_slots_env = -1;  // HNY_VIS_SLOTS: LDS visited table entries per walk wave, -1 = auto
  bool wave_prune_only = false;
  size_t max_ops = 0, sel_words = 0;
  double t_upload = 0, t_build0 = 0, t_build = 0;
  // optional per-kernel-family timing (HIP events on `stream`)
  struct Ev {
    hipEvent_t a, b;
    int kind;
  };
  std::vector<Ev> evs;
  size_t ev_used = 0;
  bool profiling = false;
  uint64_t n_walk_dispatch = 0; // k_walk launches since the last reset
  ~hny_builder() {
    for (auto &e : sync_evs) (void)hipEventDestroy(e);
    if (h_l0) (void)hipHostFree(h_l0);
    if (h_up) (void)hipHostFree(h_up);
    if (h_cnt0) (void)hipHostFree(h_cnt0);
    if (h_cntu) (void)hipHostFree(h_cntu);
    for (auto &e : evs) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  }
};
enum { EV_WALK = 0, EV_PRUNE = 1, EV_SORT = 2, EV_APPLY = 3, EV_KINDS = 4 };
// LDS visited table of a walk wave: what is left of a 10 KB share (16 waves per CU in 160 KB) after
// the beam, in whole 64-entry rows
static uint32_t eps_cap_of(const hny_builder *b) {
  // entry points, or what robust_prune selected on the layer above: up to M for an item above level 0
  return (uint32_t)std::max<size_t>(64, (std::max<size_t>(b->entry_points.size(), b->o.M) + 63) / 64 * 64);
}
// Entries of a walk's result set (beam).  walk_layer pushes every entry point without a capacity check
// (hnsw.rs:474-481) and only evicts when res.len() == ef (:505-512): a walk that starts from MORE entry
// points than ef never evicts and keeps every point closer than its farthest entry point — up to all
// items.  That happens when an index (or the batch of an incremental build that reset max_level to 0,
// hnsw.rs:258-262) has drawn level 0 only.  Such a walk gets room for every item while the index is
// small, 4x the entry points otherwise (at most 4 096 entries; beyond it the kernels report the
// overflow, never a clipped result).
// The same holds for the greedy descent (ef = 1) from several entry points: on the top layer it can
// keep every node of that layer.  A fresh index has nothing but its entry points there; after an
// update that lowered max_level (hnsw.rs:258-276) the layer also holds old nodes: `top_layer_nodes`.
// `cap`: 4 096 entries fit the walk's LDS (the Reader's searches stop there); the build takes up to
// HNY_RES_GLOBAL_MAX with the result sets in HBM (WalkArgs.res_global, the general kernels).
#define HNY_RES_LDS_MAX 4096u
#define HNY_RES_GLOBAL_MAX 65536u
static uint32_t res_capacity(uint32_t ef, uint32_t n_eps, uint64_t n_slots, uint64_t top_layer_nodes,
                             uint32_t cap = HNY_RES_LDS_MAX) {
  uint64_t need = (uint64_t)std::max(ef, n_eps) + 1;
  if (n_eps > 1) need = std::max<uint64_t>(need, std::max<uint64_t>(top_layer_nodes, n_eps) + 1);
  if (n_eps >= ef) need = n_slots + 1 <= cap ? n_slots + 1 : std::max<uint64_t>(4 * need, 1024);
  uint32_t rcap = 64;
  while (rcap < need && rcap < cap) rcap *= 2;
  return rcap;
}

static uint32_t vis_slots_for(const hny_builder *b, uint32_t rcap) {
  // measured: +5 % on 3 KB rows (C2/C3), -3 % on 512-B rows (the table clear per greedy layer and
  // the longer probes outweigh the saved L2 atomics when a row costs little; round 2, 5M x 1024 bits:
  // 0.73-0.91 s against 0.66 s; round 4, same table in walk_layer_short: 0.58-0.72 s against 0.59 s): rows > 1 KB
  // only — rows <= 512 B have their own table of 16-bit remainders (vis_buckets_for)
  if ((size_t)b->g.n16 * 16 <= 1024) return 0;
  if (b->vis_slots_env >= 0) return (uint32_t)std::min(8192, b->vis_slots_env);
  const size_t fixed = hnyk_walk_lds_bytes(rcap, eps_cap_of(b));
  if (fixed + 512 * 4 > 10240) return 512;
  return (uint32_t)((10240 - fixed) / 4 / 64 * 64);
}
// Short rows (<= 512 B, walk_layer_short): the LDS visited table of 16-bit remainders.  As many buckets as keep
// the walk's occupancy: 10 KB of LDS per wave at 4 waves per SIMD (f32: 896 buckets; 640: +2 %, 1 024: +5 % walk
// time at 4M x 128), 6.25 KB at 6 (binary codes: 448 buckets and 6 144 resident waves; at 5 waves per SIMD 512
// buckets were best, 640 / 768 3 % / 1.5 % slower) — and only while a remainder fits 16 bits (2^k / buckets
// < 65 535, n < 2^28).
// HNY_VIS_BUCKETS overrides (0 = bitset only).
static void vis_buckets_for(const hny_builder *b, WalkArgs &w) {
  w.vis_buckets = w.vis_magic = w.vis_shift = w.vis_smask = 0;
  if ((size_t)b->g.n16 * 16 > 512 || w.vis_slots || w.res_global || w.rcap > 128 || w.eps_cap > 64) return;
  const size_t fixed = hnyk_walk_lds_bytes(w.rcap, w.eps_cap);
  const size_t budget = (b->shape.nch == 1 && b->o.metric >= HNY_HAMMING) ? 6400 : 10240;
  int nb = fixed + 1024 <= budget ? (int)((budget - fixed) / 8 / 64 * 64) : 0;
  const char *e = getenv("HNY_VIS_BUCKETS");
  if (e && *e) nb = std::max(0, std::min(4096, atoi(e) / 2 * 2));
  if (nb < 64) return;
  uint32_t k = 1;
  while (k < 28 && (1ull << k) < (uint64_t)std::max<uint32_t>(b->g.n, 2)) k++;
  if ((1ull << k) < (uint64_t)b->g.n) return;
  if (!(e && *e)) // a larger index: more buckets (fewer resident waves) before giving the table up
    while (((1ull << k) - 1) / (uint64_t)nb + 1 >= 65535 && fixed + (size_t)(nb + 64) * 8 <= 12288) nb += 64;
  if (((1ull << k) - 1) / (uint64_t)nb + 1 >= 65535) return;
  w.vis_buckets = (u32)nb;
  uint32_t lg = 0;
  while ((2u << lg) <= (uint32_t)nb) lg++;
  w.vis_shift = 31 + lg;
  w.vis_magic = (u32)((1ull << w.vis_shift) / (uint64_t)nb + 1);
  w.vis_smask = (u32)((1ull << k) - 1);
}
static void prof_begin(hny_builder *b, int kind, hipStream_t st = nullptr) {
  if (!b->profiling) return;
  if (!st) st = b->stream;
  if (b->ev_used == b->evs.size()) {
    hny_builder::Ev e{};
    e.kind = kind;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
    b->evs.push_back(e);
  }
  b->evs[b->ev_used].kind = kind;
  (void)hipEventRecord(b->evs[b->ev_used].a, st);
}
static void prof_end(hny_builder *b, hipStream_t st = nullptr) {
  if (!b->profiling || b->ev_used >= b->evs.size()) return;
  if (!st) st = b->stream;
  (void)hipEventRecord(b->evs[b->ev_used].b, st);
  b->ev_used++;
}

namespace {

int pick_shape(int metric, uint32_t dim, LaunchShape &s, uint32_t &n16) {
  n16 = is_binary(metric) ? (uint32_t)((vec_bytes(metric, dim) + 15) / 16) : (dim + 3) / 4;
  uint32_t l = pow2ceil(n16);
  if (l < 8) l = 8;
  if (l > 64) l = 64;
  uint32_t c = (n16 + l - 1) / l;
  static const uint32_t set[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (uint32_t v : set)
    if (c <= v) {
      s.lpr = (int)l;
      s.nch = (int)v;
      return HNY_OK;
    }
  return fail(HNY_ERR_UNSUPPORTED, "dim %u needs more than 16 chunks per lane (max f32 dim 4096)", dim);
}

int mclass_of(int metric) {
  switch (metric) {
    case HNY_COSINE: return MC_DOT;
    case HNY_EUCLIDEAN: return MC_L2;
    case HNY_MANHATTAN: return MC_L1;
    default: return MC_BIN;
  }
}

uint32_t cap_of(const hny_builder *b, uint32_t layer_or_level) {
  return layer_or_level == 0 ? b->o.M0 : b->o.M; // hnsw.rs:540, 572
}

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// codec bytes -> zero padded device rows (UnalignedVector is byte-packed and unaligned inside
// LMDB pages, f32.rs:9-55; the device wants 16-byte aligned rows)
int upload_rows(const void *vectors, size_t stride, size_t vbytes, uint64_t n, uint32_t row_stride,
                unsigned char *dst, hipStream_t st) {
  if (stride == row_stride && vbytes == row_stride) {
    HIP_TRY(hipMemcpyAsync(dst, vectors, (size_t)n * row_stride, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HNY_OK;
  }
  const size_t chunk_rows = std::max<size_t>(1, (64u << 20) / row_stride);
  std::vector<unsigned char> stage(chunk_rows * row_stride);
  for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
    size_t cnt = (size_t)std::min<uint64_t>(chunk_rows, n - r0);
    std::fill(stage.begin(), stage.begin() + cnt * row_stride, 0);
    for (size_t r = 0; r < cnt; r++)
      memcpy(&stage[r * row_stride], (const unsigned char *)vectors + (r0 + r) * stride, vbytes);
    HIP_TRY(hipMemcpyAsync(dst + r0 * row_stride, stage.data(), cnt * row_stride,
                           hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return HNY_OK;
}

int reset_graph(hny_builder *b) {
  hipStream_t st = b->stream;
  HIP_TRY(hnyk_fill_u32(b->d_l0_ids.p, HNY_SENT, b->d_l0_ids.n, st));
  HIP_TRY(hipMemsetAsync(b->d_l0_cnt.p, 0, b->d_l0_cnt.n * 4, st));
  HIP_TRY(hipMemsetAsync(b->d_l0_dist.p, 0, b->d_l0_dist.n * 4, st));
  if (b->d_up_ids.n) {
    HIP_TRY(hnyk_fill_u32(b->d_up_ids.p, HNY_SENT, b->d_up_ids.n, st));
    HIP_TRY(hipMemsetAsync(b->d_up_cnt.p, 0, b->d_up_cnt.n * 4, st));
    HIP_TRY(hipMemsetAsync(b->d_up_dist.p, 0, b->d_up_dist.n * 4, st));
  }
  HIP_TRY(hipMemsetAsync(b->d_stats.p, 0, ST_COUNT * 8, st));
  HIP_TRY(hipMemsetAsync(b->d_bits.p, 0, b->d_bits.n * 4, st));
  b->pos = 0;
  b->n_done = b->n_done0;
  b->n_batches = 0;
  b->in_batch = false;
  b->finalized = false;
  b->ev_used = 0;
  b->n_walk_dispatch = 0;
  b->sync_used = 0;
  b->t_build0 = now_s();
  return HNY_OK;
}

size_t group_end(const hny_builder *b, size_t pos) {
  size_t e = pos;
  while (e < b->order.size() && b->order_level[e] == b->order_level[pos]) e++;
  return e;
}

} // namespace

extern "C" {

const char *hny_last_error(void) { return g_err.c_str(); }
const char *hny_version(void) { return "hannoy_amd 0.1.0 (gfx950)"; }
uint32_t hny_abi_sizes(uint32_t *out, uint32_t n) {
  const uint32_t sizes[HNY_ABI_N_STRUCTS] = {
      (uint32_t)sizeof(hny_build_opts), (uint32_t)sizeof(hny_items),      (uint32_t)sizeof(hny_graph),
      (uint32_t)sizeof(hny_prev_graph), (uint32_t)sizeof(hny_batch),      (uint32_t)sizeof(hny_query_opts),
      (uint32_t)sizeof(hny_lmdb_stat)};
  for (uint32_t i = 0; out && i < n && i < HNY_ABI_N_STRUCTS; i++) out[i] = sizes[i];
  return HNY_ABI_N_STRUCTS;
}

size_t hny_vector_bytes(int32_t metric, uint32_t dim) { return vec_bytes(metric, dim); }
size_t hny_header_bytes(int32_t metric) { return hdr_bytes(metric); }

int hny_draw_levels_from_seed(const uint8_t seed[32], uint64_t skip, uint32_t M, uint64_t n, uint8_t *out) {
  if (!seed || !out || M == 0 || n >= (1ull << 31))
    return fail(HNY_ERR_INVALID_ARG, "hny_draw_levels_from_seed: bad argument");
  StdRngChaCha12 rng(seed);
  draw_levels_rng(rng, skip, M, (uint32_t)n, out);
  return HNY_OK;
}

int hny_draw_levels(uint64_t seed, uint32_t M, uint64_t n, uint8_t *out) {
  if (!out || M == 0 || n >= (1ull << 31)) return fail(HNY_ERR_INVALID_ARG, "hny_draw_levels: bad argument");
  draw_levels(seed, M, (uint32_t)n, out);
  return HNY_OK;
}

// the default cap of the batch schedule: the largest power of two <= n / 12, at least 65 536.  At 1M
// items that is the 65 536 whose recall was validated against CPU-built indexes; the
// cap grows with the index so that a batch stays the same small fraction of it (5M: 262 144, 10M:
// 524 288) — fewer, larger launches, same relative staleness (C5: recall@10 0.635 vs 0.631).
uint32_t hny_default_batch_max(uint64_t n_items) {
  uint32_t b = 65536u;
  while ((uint64_t)b * 2 * 12 <= n_items && b < (1u << 21)) b *= 2; // 2^21 x 64 x 2 link ops < 2^29
  return b;
}

uint32_t hny_batch_size(double frac, uint32_t bmax, uint64_t n_done) {
  if (bmax == 0) return 1;
  double v = std::floor(frac * (double)n_done);
  if (v < 1.0) v = 1.0;
  if (v > (double)bmax) v = (double)bmax;
  return (uint32_t)v;
}

// UnalignedVectorCodec::from_slice + Distance::new_header (host; ingest is not on the timed path)
int hny_encode_vectors(int32_t metric, uint32_t dim, uint64_t n, const float *vectors, void *out_codes,
                       void *out_headers) {
  if (!vectors || !out_codes || !out_headers || metric < 0 || metric > HNY_BQ_MANHATTAN || dim == 0)
    return fail(HNY_ERR_INVALID_ARG, "hny_encode_vectors: bad argument");
  const size_t vb = vec_bytes(metric, dim), hb = hdr_bytes(metric);
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 4096) nt = 1;
  auto work = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      const float *v = vectors + i * dim;
      unsigned char *code = (unsigned char *)out_codes + i * vb;
      unsigned char *hdr = (unsigned char *)out_headers + i * hb;
      if (!is_binary(metric)) {
        memcpy(code, v, vb); // f32.rs:9-55 raw native-endian bytes
        float h = 0.0f;      // bias = 0.0 (euclidean.rs:38-40, manhattan.rs:37-39)
        if (metric == HNY_COSINE) h = sqrtf(dot_x86_order(v, v, dim)); // cosine.rs:36-38,58-60
        memcpy(hdr, &h, 4);
        continue;
      }
      uint32_t ones = 0;
      for (uint32_t base = 0; base < dim; base += 64) {
        uint64_t word = 0;
        uint32_t cnt = std::min<uint32_t>(64, dim - base);
        for (uint32_t k = 0; k < cnt; k++) {
          uint32_t bits;
          memcpy(&bits, &v[base + k], 4);
          // binary.rs:87-89: 0 < bits < 0x8000_0000 ; binary_quantized.rs:86: is_sign_positive
          bool one = metric == HNY_HAMMING ? (bits < 0x80000000u && bits > 0u) : (bits >> 31) == 0;
          if (one) word |= 1ull << k; // dim i -> bit (i mod 64), LSB first
        }
        memcpy(code + (base / 64) * 8, &word, 8);
        ones += (uint32_t)__builtin_popcountll(word);
      }
      if (metric == HNY_HAMMING) {
        uint64_t z = 0; // hamming.rs:40-42 idx = 0usize
        memcpy(hdr, &z, 8);
      } else {
        float h = 0.0f;
        // binary_quantized_cosine.rs:40-42,61-63: sqrt(dot_bq(v,v)) = sqrt(padded dims)
        if (metric == HNY_BQ_COSINE) h = sqrtf((float)(int32_t)(vb * 8));
        memcpy(hdr, &h, 4);
      }
      (void)ones;
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
  for (auto &t : th) t.join();
  return HNY_OK;
}

// the same on the device: codes through k_quantize, Cosine norms through k_norms_x86 (bit-identical
// to the host path above), streamed in chunks so that any n fits
int hny_selftest_lane_ops(int32_t device, uint32_t *mismatch64) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(HNY_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  DevBuf<u32> d;
  HIP_TRY(d.alloc(64));
  HIP_TRY(hnyk_lane_selftest(d.p, nullptr));
  uint32_t h[64];
  HIP_TRY(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
  uint32_t any = 0;
  for (int i = 0; i < 64; i++) {
    any |= h[i];
    if (mismatch64) mismatch64[i] = h[i];
  }
  if (any) return fail(HNY_ERR_DEVICE, "cross-lane primitives disagree with __shfl_xor: mask 0x%08x", any);
  return HNY_OK;
}

int hny_encode_vectors_gpu(int32_t metric, uint32_t dim, uint64_t n, const float *vectors,
                           void *out_codes, void *out_headers, int32_t device) {
  if (!vectors || !out_codes || !out_headers || metric < 0 || metric > HNY_BQ_MANHATTAN || dim == 0)
    return fail(HNY_ERR_INVALID_ARG, "hny_encode_vectors_gpu: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(HNY_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  const size_t vb = vec_bytes(metric, dim), hb = hdr_bytes(metric);
  const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / ((uint64_t)dim * 4));
  DevBuf<float> dv, dn;
  DevBuf<u64> dc;
  HIP_TRY(dv.alloc(std::min(chunk, std::max<uint64_t>(n, 1)) * dim));
  if (metric == HNY_COSINE) HIP_TRY(dn.alloc(std::min(chunk, std::max<uint64_t>(n, 1))));
  if (is_binary(metric)) HIP_TRY(dc.alloc(std::min(chunk, std::max<uint64_t>(n, 1)) * (vb / 8)));
  std::vector<float> norms;
  for (uint64_t i0 = 0; i0 < n; i0 += chunk) {
    const uint64_t cnt = std::min(chunk, n - i0);
    unsigned char *codes = (unsigned char *)out_codes + i0 * vb;
    unsigned char *hdrs = (unsigned char *)out_headers + i0 * hb;
    memset(hdrs, 0, cnt * hb); // bias 0.0 (euclidean.rs:38-40 ...), idx 0 (hamming.rs:40-42)
    if (metric == HNY_COSINE || is_binary(metric))
      HIP_TRY(hipMemcpy(dv.p, vectors + i0 * dim, cnt * dim * 4, hipMemcpyHostToDevice));
    if (!is_binary(metric)) {
      memcpy(codes, vectors + i0 * dim, cnt * vb); // f32.rs:9-55
      if (metric == HNY_COSINE) {
        HIP_TRY(hnyk_norms_x86(dv.p, dim, cnt, dn.p, nullptr));
        norms.resize(cnt);
        HIP_TRY(hipMemcpy(norms.data(), dn.p, cnt * 4, hipMemcpyDeviceToHost));
        memcpy(hdrs, norms.data(), cnt * 4);
      }
    } else {
      HIP_TRY(hnyk_quantize(dv.p, dim, cnt, metric == HNY_HAMMING, dc.p, nullptr));
      HIP_TRY(hipMemcpy(codes, dc.p, cnt * vb, hipMemcpyDeviceToHost));
      if (metric == HNY_BQ_COSINE) { // sqrt(dot_bq(v,v)) = sqrt(padded dims)
        float h = sqrtf((float)(int32_t)(vb * 8));
        for (uint64_t i = 0; i < cnt; i++) memcpy(hdrs + i * 4, &h, 4);
      }
    }
  }
  return HNY_OK;
}

void hny_builder_destroy(hny_builder *b) {
  if (!b) return;
  if (b->stream) {
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    (void)hipStreamDestroy(b->stream);
  }
  delete b;
}

// incremental-build inputs (hny_build_incremental); null for a fresh index
struct IncrementalSpec {
  const uint32_t *to_insert;
  uint64_t n_insert;
  const uint32_t *to_delete;
  uint64_t n_delete;
  const hny_prev_graph *prev;
  bool load_only = false; // Reader::open: the stored graph as it is, nothing gets (re)inserted
};

static int create_impl(const hny_build_opts *opts, const hny_items *items, const IncrementalSpec *inc,
                       hny_builder **out) {
  if (!opts || !items || !out) return fail(HNY_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  const hny_build_opts &o = *opts;
  if (o.metric < 0 || o.metric > HNY_BQ_MANHATTAN) return fail(HNY_ERR_INVALID_ARG, "bad metric");
  if (o.dim == 0) return fail(HNY_ERR_INVALID_DIM, "dim must be > 0");
  if (o.M == 0 || o.M0 < o.M) return fail(HNY_ERR_INVALID_ARG, "need 1 <= M <= M0");
  if ((o.schedule & ~(uint32_t)(HNY_SCHED_NO_SHUFFLE | HNY_SCHED_LEVEL_ORDER_ID | HNY_SCHED_UPDATE_NO_RAMP)) || o.reserved_)
    return fail(HNY_ERR_INVALID_ARG, "unknown hny_build_opts.schedule bits 0x%x", o.schedule);
  if (o.M0 > HNY_BIG_CAP) return fail(HNY_ERR_UNSUPPORTED, "M0 %u > %d", o.M0, HNY_BIG_CAP);
  // 64 < M0 <= HNY_BIG_CAP: lists are walked 64 slots at a time; the workgroup kernels hold them whole (incremental
  // builds: k_fill_gaps_wg), strict mode's one-wave kernels (k_prune, k_apply) take them 64 slots at a time —
  // for fresh builds and loaded graphs; a strict-mode UPDATE of such lists (k_fill_gaps: one lane per slot) is refused
  const bool bigcap = o.M0 > HNY_MAX_CAP;
  if (o.ef_construction == 0 || o.ef_construction > HNY_MAX_EF)
    return fail(HNY_ERR_UNSUPPORTED, "ef_construction %u outside [1, %d]", o.ef_construction,
                HNY_MAX_EF);
  if (items->n >= (1ull << 31)) return fail(HNY_ERR_UNSUPPORTED, "n >= 2^31");
  const size_t vb = vec_bytes(o.metric, o.dim), hb = hdr_bytes(o.metric);
  if (items->n && (!items->ids || !items->vectors || !items->headers))
    return fail(HNY_ERR_INVALID_ARG, "null item arrays");
  if (items->n && items->stride < vb)
    return fail(HNY_ERR_INVALID_DIM, "stride %zu < %zu codec bytes for dim %u", items->stride, vb,
                o.dim); // Error::InvalidVecDimension
  if (items->n && items->header_size != hb)
    return fail(HNY_ERR_INVALID_ARG, "header_size %zu, expected %zu", items->header_size, hb);
  for (uint64_t i = 1; i < items->n; i++)
    if (items->ids[i] <= items->ids[i - 1]) return fail(HNY_ERR_INVALID_ARG, "ids not ascending");

  auto b = std::unique_ptr<hny_builder, void (*)(hny_builder *)>(new hny_builder(), hny_builder_destroy);
  b->o = o;
  b->frac = o.batch_frac > 0.0 ? o.batch_frac : 1.0;
  b->bmax = o.batch_max ? o.batch_max : hny_default_batch_max(items->n);
  b->incremental = inc != nullptr;
  uint32_t n16;
  int rc = pick_shape(o.metric, o.dim, b->shape, n16);
  if (rc) return rc;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(HNY_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
  if (o.device >= 0) {
    b->device = o.device;
    HIP_TRY(hipSetDevice(b->device));
  } else {
    HIP_TRY(hipGetDevice(&b->device));
  }
  HIP_TRY(hipStreamCreate(&b->stream));
  hipStream_t st = b->stream;
  double t0 = now_s();

  // ---- slot universe: fresh = the items; incremental = items U deleted U everything the old graph
  // mentions (ids ascending => slot order == id order) ----
  std::vector<uint32_t> &U = b->ids;
  U.assign(items->ids, items->ids + items->n);
  if (inc) {
    const hny_prev_graph *pg = inc->prev;
    if (!pg || (inc->n_insert && !inc->to_insert) || (inc->n_delete && !inc->to_delete))
      return fail(HNY_ERR_INVALID_ARG, "incremental: null argument");
    U.insert(U.end(), inc->to_delete, inc->to_delete + inc->n_delete);
    U.insert(U.end(), pg->rec_item, pg->rec_item + pg->n_records);
    if (pg->n_records) U.insert(U.end(), pg->neighbours, pg->neighbours + pg->rec_offset[pg->n_records]);
    U.insert(U.end(), pg->entry_points, pg->entry_points + pg->n_entry_points);
    std::sort(U.begin(), U.end());
    U.erase(std::unique(U.begin(), U.end()), U.end());
    if (U.size() >= (1ull << 31)) return fail(HNY_ERR_UNSUPPORTED, "n >= 2^31");
  }
  b->n = (uint32_t)U.size();
  const uint32_t n = b->n;
  auto slot_of = [&](uint32_t id) { return (uint32_t)(std::lower_bound(U.begin(), U.end(), id) - U.begin()); };

  std::vector<uint8_t> has_vec(n, inc ? 0 : 1);
  std::vector<uint32_t> item_slot(items->n);
  for (uint64_t i = 0; i < items->n; i++) {
    item_slot[i] = inc ? slot_of(items->ids[i]) : (uint32_t)i;
    has_vec[item_slot[i]] = 1;
  }
  b->ins_level.assign(n, -1);
  b->old_mask.assign(n, 0);
  b->deleted.assign(n, 0);
  std::vector<std::pair<uint32_t, uint8_t>> levels; // (slot, level) in the reference's order

  if (!inc) {
    // ---- levels (hnsw.rs:141-149) + prepare_levels_and_entry_points, fresh DB (:222-289) ----
    std::vector<uint8_t> lv(n);
    if (items->levels)
      memcpy(lv.data(), items->levels, n);
    else
      draw_levels(o.seed, o.M, n, lv.data());
    for (uint32_t s = 0; s < n; s++) levels.push_back({s, lv[s]});
    hny_rust_sort::sort_levels(levels, level_order_by_id(o)); // hnsw.rs:268, ties as the reference leaves them
    if (shuffle_groups(o, b->bmax)) hny_rust_sort::shuffle_level_groups(levels);
    b->max_level = n ? levels[0].second : 0;
    for (uint32_t s = 0; s < n; s++)
      if (lv[s] == b->max_level) b->entry_points.push_back(s);
  } else {
    // ---- incremental: writer.rs:539-554 set algebra is the caller's; here hnsw.rs:141-149 and the
    // deletion branch of prepare_levels_and_entry_points (:236-289) ----
    const hny_prev_graph *pg = inc->prev;
    for (uint64_t i = 0; i < inc->n_delete; i++) b->deleted[slot_of(inc->to_delete[i])] = 1;
    for (uint64_t r = 0; r < pg->n_records; r++) {
      if (pg->rec_layer[r] > HNY_MAX_LEVEL) return fail(HNY_ERR_INVALID_ARG, "old record on layer > %d", HNY_MAX_LEVEL);
      b->old_mask[slot_of(pg->rec_item[r])] |= (uint16_t)(1u << pg->rec_layer[r]);
    }
    std::vector<uint8_t> lv(inc->n_insert);
    if (items->levels)
      memcpy(lv.data(), items->levels, inc->n_insert); // one per to_insert id, ascending
    else
      draw_levels(o.seed, o.M, (uint32_t)inc->n_insert, lv.data());
    uint32_t cur_max = 0;
    for (uint64_t i = 0; i < inc->n_insert; i++) {
      if (i && inc->to_insert[i] <= inc->to_insert[i - 1]) return fail(HNY_ERR_INVALID_ARG, "to_insert not ascending");
      uint32_t s = slot_of(inc->to_insert[i]);
      if (s >= n || U[s] != inc->to_insert[i] || !has_vec[s])
        return fail(HNY_ERR_MISSING_KEY, "to_insert id %u has no item", inc->to_insert[i]); // Error::MissingKey
      levels.push_back({s, lv[i]});
      cur_max = std::max<uint32_t>(cur_max, lv[i]);
    }
    uint32_t max_level = pg->max_level;
    std::vector<uint8_t> in_new(n, 0), in_old(n, 0);
    uint32_t n_old = 0, n_new = 0;
    std::vector<uint32_t> del_eps;
    for (uint32_t i = 0; i < pg->n_entry_points; i++) {
      uint32_t s = slot_of(pg->entry_points[i]);
      if (!in_old[s]) { in_old[s] = 1; n_old++; }
      if (b->deleted[s]) del_eps.push_back(s);
      else if (!in_new[s]) { in_new[s] = 1; n_new++; }
    }
    uint32_t l = max_level;
    for (size_t k = 0; k < del_eps.size(); k++) { // :243-257 replace deleted entry points
      for (;;) {
        if (l <= HNY_MAX_LEVEL)
          for (uint32_t s = 0; s < n; s++) // iter_layer_links(l): ascending item
            if (((b->old_mask[s] >> l) & 1) && !b->deleted[s] && !in_new[s]) {
              in_new[s] = 1;
              n_new++;
              break;
            }
        if (l == 0) break;
        l -= 1;
      }
    }
    if (!del_eps.empty() && n_new != n_old) max_level = 0; // :261-263
    for (uint32_t s = 0; s < n; s++)
      if (in_new[s] && !inc->load_only) levels.push_back({s, (uint8_t)max_level}); // :267
    hny_rust_sort::sort_levels(levels, level_order_by_id(o)); // :268
    if (shuffle_groups(o, b->bmax)) hny_rust_sort::shuffle_level_groups(levels);
    if (cur_max > max_level) { // :272-276
      std::fill(in_new.begin(), in_new.end(), 0);
      max_level = cur_max;
    }
    for (auto &pr : levels) { // :278-285
      if (pr.second != max_level) break;
      in_new[pr.first] = 1;
    }
    b->max_level = max_level;
    for (uint32_t s = 0; s < n; s++)
      if (in_new[s]) {
        b->entry_points.push_back(s);
        if (!has_vec[s]) return fail(HNY_ERR_MISSING_KEY, "entry point %u has no item", U[s]);
      }
    // The batch schedule of an update ramps up from one member like a fresh build's (n_done0 stays 0): rounds
    // 1-2 counted the surviving old records as "already inserted", so an update went in as one or two batches —
    // and new items that form a region of their own (a new topic appended to an index) searched a graph that
    // held none of them: measured recall 0.51 on such a region against 0.99 with the ramp
    // (scripts/r3_new_region_update.py).  HNY_SCHED_UPDATE_NO_RAMP: the old rule.
    if (o.schedule & HNY_SCHED_UPDATE_NO_RAMP)
      for (uint32_t s = 0; s < n; s++)
        if ((b->old_mask[s] & 1) && !b->deleted[s]) b->n_done0++;
    for (uint32_t s = 0; s < n; s++)
      for (uint32_t ll = 0; ll <= HNY_MAX_LEVEL; ll++)
        if (((b->old_mask[s] >> ll) & 1) && !b->deleted[s]) b->old_recs.push_back(((u64)ll << 31) | s);
  }
  if (b->max_level > HNY_MAX_LEVEL) return fail(HNY_ERR_INVALID_ARG, "level > %d", HNY_MAX_LEVEL);
  for (auto &pr : levels) {
    b->order.push_back(pr.first);
    b->order_level.push_back(pr.second);
    b->ins_level[pr.first] = std::max<int8_t>(b->ins_level[pr.first], (int8_t)pr.second);
  }
  for (uint32_t s : b->entry_points) // pre-registered in every layer (:278-285)
    b->ins_level[s] = std::max<int8_t>(b->ins_level[s], (int8_t)b->max_level);
  // storage level of a slot = highest layer it can own a list on
  b->level.assign(n, 0);
  b->upper_idx.assign(n, -1);
  b->up_layers = std::max<uint32_t>(b->max_level, 1);
  for (uint32_t s = 0; s < n; s++) {
    int top = b->ins_level[s] > 0 ? b->ins_level[s] : 0;
    for (int ll = HNY_MAX_LEVEL; ll > top; ll--)
      if ((b->old_mask[s] >> ll) & 1) {
        top = ll;
        break;
      }
    b->level[s] = (uint8_t)top;
    if (top >= 1) b->upper_idx[s] = (int32_t)b->n_upper++;
    b->up_layers = std::max<uint32_t>(b->up_layers, (uint32_t)top);
  }
  if (b->entry_points.size() > HNY_MAX_EPS)
    return fail(HNY_ERR_UNSUPPORTED, "%zu entry points > %d", b->entry_points.size(), HNY_MAX_EPS);
  for (uint32_t s = 0; s < n; s++) { // what finish() will export at most (its rec_mask)
    if (b->deleted[s]) continue;
    uint32_t m = b->old_mask[s];
    if (b->ins_level[s] >= 0) m |= (2u << b->ins_level[s]) - 1u;
    const uint32_t c = (uint32_t)__builtin_popcount(m);
    b->nrec_bound += c;
    b->nbr_bound += (m & 1u ? o.M0 : 0u) + (uint64_t)(c - (m & 1u)) * o.M;
  }

  // ---- sizes ----
  b->top_layer_nodes = 0; // nodes a walk can meet on layer max_level: old records there + what gets inserted there
  for (uint32_t s = 0; s < n; s++)
    if ((((b->old_mask.empty() ? 0u : b->old_mask[s]) >> b->max_level) & 1u && !b->deleted[s]) ||
        b->ins_level[s] >= (int8_t)b->max_level)
      b->top_layer_nodes++;
  b->rcap = res_capacity(o.ef_construction, (uint32_t)b->entry_points.size(), n, b->top_layer_nodes, HNY_RES_GLOBAL_MAX);
  b->max_batch = 1;
  b->max_ops = 2;
  b->sel_words = 2;
  for (size_t pos = 0; pos < b->order.size();) {
    size_t e = group_end(b.get(), pos);
    uint32_t L = b->order_level[pos];
    uint64_t bs = std::min<uint64_t>(b->bmax, e - pos);
    uint64_t cs = cap_of(b.get(), L);
    b->max_batch = std::max<uint32_t>(b->max_batch, (uint32_t)bs);
    b->max_ops = std::max<size_t>(b->max_ops, (size_t)(bs * (L + 1) * cs * 2));
    b->sel_words = std::max<size_t>(b->sel_words, (size_t)(bs * (L + 1) * (cs + 1)));
    pos = e;
  }
  if (b->max_ops >= (1ull << HNY_SEQ_BITS))
    return fail(HNY_ERR_UNSUPPORTED,
                "batch_max %u too large for M0 %u: %zu link ops per batch >= 2^%d (pass batch_max <= %u; the schedule "
                "is part of the result, so it is never shrunk silently)",
                b->bmax, o.M0, b->max_ops, HNY_SEQ_BITS, (uint32_t)((1ull << (HNY_SEQ_BITS - 1)) / std::max(o.M0, 2 * o.M) / 2));
  // resident walk waves: 256 CUs x 4 SIMDs x waves per SIMD (6 for binary codes <= 512 B on the register beam —
  // the instances k_walk's launch bound gives six waves, see there; 4 otherwise)
  const bool six_waves = b->shape.nch == 1 && b->shape.lpr <= 32 && o.metric >= HNY_HAMMING && b->rcap <= 128 &&
                         env_int("HNY_NO_RB", 0) == 0;
  b->walk_slots = (uint32_t)std::min<int64_t>(std::max(1, env_int("HNY_WALK_SLOTS", six_waves ? 6144 : 4096)), 65536);
  b->bits_words = (n + 31) / 32 + 1;
  b->log_cap = (uint32_t)std::max(1024, env_int("HNY_VISITED_LOG", 16384));
  b->vis_slots_env = env_int("HNY_VIS_SLOTS", -1);
  {
    // LDS staging budget of the workgroup prune, whole load groups
    int rpg = 64 / b->shape.lpr;
    int sl = (int)((u32)std::max(0, env_int("HNY_STAGE_BYTES", 24576)) / (n16 * 16u));
    if (sl > HNY_MAX_CAP) sl = HNY_MAX_CAP;
    b->stage_rows = sl / rpg * rpg;
    // the workgroup prune kernels carry their own wave-order arithmetic: strict mode and very long
    // rows use the single-wave kernels, which all go through dist_rows
    b->wave_prune_only = b->shape.nch > 8 || o.x86_order; // the one-wave prune: strict mode, rows beyond 8 KB
    // (strict mode updates of lists beyond 64 slots: k_fill_gaps_wg with the one-wave prune inside, round 5)
    if (bigcap && b->shape.nch > 8 && inc && !inc->load_only)
      return fail(HNY_ERR_UNSUPPORTED, "M0 %u > %d: an incremental build needs the workgroup kernels (rows <= 8 KB)",
                  o.M0, HNY_MAX_CAP);
  }

  // ---- device memory ----
  GraphDev &g = b->g;
  g.n = n;
  g.metric = o.metric;
  g.mclass = mclass_of(o.metric);
  g.n16 = n16;
  g.row_stride = n16 * 16;
  g.bin_bits = (u32)(vb * 8);
  g.bin_inv = (g.bin_bits && (g.bin_bits & (g.bin_bits - 1)) == 0) ? 1.0f / (float)g.bin_bits : 0.0f;
  g.M = o.M;
  g.M0 = o.M0;
  g.max_level = b->max_level;
  g.up_layers = b->up_layers;
  g.n_upper = b->n_upper;
  g.alpha = o.alpha;
  g.incremental = inc ? 1 : 0;
  g.dim = o.dim;
  g.x86_order = o.x86_order ? 1 : 0;
  const size_t nn = std::max<uint32_t>(n, 1);
  const size_t no = std::max<size_t>(b->order.size(), 1);
  HIP_TRY(b->d_rows.alloc(nn * g.row_stride));
  HIP_TRY(b->d_level.alloc(nn));
  HIP_TRY(b->d_upper_idx.alloc(nn));
  const bool has_norm = o.metric == HNY_COSINE || o.metric == HNY_BQ_COSINE;
  if (has_norm) HIP_TRY(b->d_norms.alloc(nn));
  HIP_TRY(b->d_l0_ids.alloc(nn * o.M0));
  HIP_TRY(b->d_l0_dist.alloc(nn * o.M0));
  HIP_TRY(b->d_l0_cnt.alloc(nn));
  const size_t nup = (size_t)b->n_upper * b->up_layers;
  HIP_TRY(b->d_up_ids.alloc(nup * o.M));
  HIP_TRY(b->d_up_dist.alloc(nup * o.M));
  HIP_TRY(b->d_up_cnt.alloc(nup));
  HIP_TRY(b->d_order.alloc(no));
  HIP_TRY(b->d_fin_cnt0.alloc(nn));
  HIP_TRY(b->d_fin_cntu.alloc(std::max<size_t>(nup, 1)));
  HIP_TRY(hipHostMalloc((void **)&b->h_l0, nn * o.M0 * 4));
  HIP_TRY(hipHostMalloc((void **)&b->h_cnt0, nn * 4));
  HIP_TRY(hipHostMalloc((void **)&b->h_up, std::max<size_t>(nup * o.M, 1) * 4));
  HIP_TRY(hipHostMalloc((void **)&b->h_cntu, std::max<size_t>(nup, 1) * 4));
  HIP_TRY(b->d_eps.alloc(std::max<size_t>(64, b->entry_points.size())));
  HIP_TRY(b->d_stats.alloc(ST_COUNT));
  const uint32_t slots = std::min<uint32_t>(b->walk_slots, std::max<uint32_t>(b->max_batch, 256));
  b->walk_slots = slots;
  HIP_TRY(b->d_bits.alloc((size_t)slots * b->bits_words));
  HIP_TRY(b->d_vlog.alloc((size_t)slots * b->log_cap));
  HIP_TRY(b->d_sel.alloc(b->sel_words));
  const size_t cand_rows = std::max<uint32_t>(b->max_batch, 256);
  if (b->rcap > HNY_RES_LDS_MAX) { // walks that never evict (res_capacity): result sets and candidate lists in HBM
    if (b->wave_prune_only)
      return fail(HNY_ERR_UNSUPPORTED, "a result set of %u entries needs the workgroup prune kernels (no x86_order, rows <= 8 KB)", b->rcap);
    if ((cand_rows + slots) * (size_t)b->rcap * 8 > ((size_t)24 << 30))
      return fail(HNY_ERR_UNSUPPORTED, "result sets of %u entries for %zu batch members exceed the 24 GB set aside for them; "
                  "pass a smaller batch_max", b->rcap, cand_rows);
    HIP_TRY(b->d_res_global.alloc((size_t)slots * b->rcap));
  }
  HIP_TRY(b->d_cand.alloc(cand_rows * b->rcap));
  HIP_TRY(b->d_cand_n.alloc(cand_rows));
  HIP_TRY(b->d_ops.alloc(4 * b->max_ops));
  b->d_keys_a.p = b->d_ops.p;
  b->d_keys_b.p = b->d_ops.p + b->max_ops;
  b->d_vals_a.p = b->d_ops.p + 2 * b->max_ops;
  b->d_vals_b.p = b->d_ops.p + 3 * b->max_ops;
  b->d_keys_a.n = b->d_keys_b.n = b->d_vals_a.n = b->d_vals_b.n = b->max_ops;
  HIP_TRY(b->d_seg_start.alloc(b->max_ops));
  HIP_TRY(b->d_nseg.alloc(4 + 16 + 8 * 16)); // + 8 per-XCD counters for each of the 16 work queues
  b->locality = env_int("HNY_NO_LOCALITY", 0) == 0;
  HIP_TRY(b->d_lkey_a.alloc(cand_rows));
  HIP_TRY(b->d_lkey_b.alloc(cand_rows));
  HIP_TRY(b->d_perm_a.alloc(cand_rows));
  HIP_TRY(b->d_perm_b.alloc(cand_rows));
  HIP_TRY(b->d_eps0.alloc(cand_rows));
  {
    // walk_layer on heaps (k_walk_heap): `candidates` never holds more than the items visited (+ the entry
    // points), `res` no more than a candidate row.  The retry path normally sees no member, so the `candidates`
    // heaps own no memory: they live in the link-op arrays (d_ops), idle while a batch is searched — rounds 3-4
    // set ~1 GB per builder aside for them (x 8 replicas on a node) and clipped a heap at 2^22 entries, which a
    // walk over > 4 M equidistant items would have overflowed.  Only a builder whose op arrays cannot hold ONE
    // full heap (tiny batch_max on a large index) gets heaps of its own, of at most 2^25 entries (256 MB).
    const uint64_t c_full = (uint64_t)n + 1 + eps_cap_of(b.get());
    b->heap_r_cap = b->rcap + 1;
    if (b->d_ops.n >= c_full) {
      b->heap_c_cap = (uint32_t)std::min<uint64_t>(c_full, 0xFFFFFFFFull);
      b->heap_grid = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(slots, 512), b->d_ops.n / c_full);
      b->heap_c_ptr = b->d_ops.p;
    } else {
      b->heap_c_cap = (uint32_t)std::min<uint64_t>(c_full, (uint64_t)1 << 25);
      b->heap_grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(slots, 512), ((uint64_t)1 << 25) / b->heap_c_cap));
      HIP_TRY(b->d_heap_c.alloc((size_t)b->heap_grid * b->heap_c_cap));
      b->heap_c_ptr = b->d_heap_c.p;
    }
    // Two tiers for the build's retry path: a `candidates` heap holds what a walk ACCEPTED and has not popped yet —
    // a few times ef in practice — so the members first run on heaps of 2^18 entries, as many blocks as the area
    // holds (up to 512), and only a member that outgrows such a heap is walked once more by the few blocks whose
    // heaps hold every item (C4: 13).  One tier when the full heaps are no larger than that.
    const uint64_t area = b->heap_c_ptr == b->d_ops.p ? b->d_ops.n : b->d_heap_c.n;
    const uint64_t small_cap = (uint64_t)std::max(16, env_int("HNY_HEAP_SMALL_CAP", 1 << 18));
    if ((uint64_t)b->heap_c_cap > small_cap) {
      b->heap_c_cap1 = (uint32_t)small_cap;
      b->heap_grid1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(slots, 512), area / small_cap));
      HIP_TRY(b->d_pool_retry2.alloc(cand_rows));
    }
    HIP_TRY(b->d_heap_r.alloc((size_t)std::max(b->heap_grid, b->heap_grid1) * b->heap_r_cap));
    HIP_TRY(b->d_pool_retry.alloc(cand_rows));
    HIP_TRY(b->d_pool_ctr.alloc(4 * 64)); // per walk launch: listed members, work counter — for each of the two tiers
  }
  HIP_TRY(b->d_deferred.alloc(b->max_ops));
  HIP_TRY(b->d_deferred_b.alloc(b->max_ops));
  {
    // rocPRIM does not check the size of the scratch it is given, and what it needs depends on the
    // bit range (it picks a different algorithm): take the largest of every variant that is called
    u32 tbits = 1;
    while ((1ull << tbits) < (u64)b->n + 1ull) tbits++;
    const u32 ranges[3][2] = {{0, 64}, {HNY_SEQ_BITS, 64}, {HNY_SEQ_BITS, (u32)HNY_SEQ_BITS + tbits}};
    b->sort_tmp_bytes = 0;
    // ... and on the number of elements (merge sort below a tuned limit, radix above): probe sizes too
    for (size_t sz = b->max_ops; sz >= 1; sz = sz > 1 ? sz / 2 : 0) {
      for (auto &r : ranges) {
        size_t need = 0;
        HIP_TRY(hnyk_sort_pairs(nullptr, need, b->d_keys_a.p, b->d_keys_b.p, b->d_vals_a.p, b->d_vals_b.p,
                                (u32)sz, r[0], r[1], st));
        b->sort_tmp_bytes = std::max(b->sort_tmp_bytes, need);
      }
      size_t need = 0;
      HIP_TRY(hnyk_sort_pairs48(nullptr, need, b->d_keys_a.p, b->d_keys_b.p, b->d_vals_a.p, b->d_vals_b.p,
                                (u32)sz, st));
      b->sort_tmp_bytes = std::max(b->sort_tmp_bytes, need);
      need = 0;
      HIP_TRY(hnyk_sort_u32(nullptr, need, b->d_deferred.p, b->d_deferred_b.p, (u32)sz, st));
      b->sort_tmp_bytes = std::max(b->sort_tmp_bytes, need);
      if (sz == 1) break;
    }
  }
  HIP_TRY(b->d_sort_tmp.alloc(b->sort_tmp_bytes + 16));

  g.rows = b->d_rows.p;
  g.norms = has_norm ? b->d_norms.p : nullptr;
  g.level = b->d_level.p;
  g.upper_idx = b->d_upper_idx.p;
  g.l0_ids = b->d_l0_ids.p;
  g.l0_dist = b->d_l0_dist.p;
  g.l0_cnt = b->d_l0_cnt.p;
  g.up_ids = b->d_up_ids.p;
  g.up_dist = b->d_up_dist.p;
  g.up_cnt = b->d_up_cnt.p;
  g.stats = b->d_stats.p;

  // ---- upload (the "export to HBM" that replaces FrozenReader, parallel.rs:11-45) ----
  if (n) {
    if (!inc) {
      rc = upload_rows(items->vectors, items->stride, vb, n, g.row_stride, b->d_rows.p, st);
      if (rc) return rc;
    } else {
      HIP_TRY(hipMemsetAsync(b->d_rows.p, 0, (size_t)n * g.row_stride, st));
      const size_t chunk = std::max<size_t>(1, (64u << 20) / g.row_stride);
      std::vector<unsigned char> stage(chunk * g.row_stride);
      uint64_t i = 0;
      for (uint32_t s0 = 0; s0 < n; s0 += (uint32_t)chunk) { // items are a subset of the universe
        uint32_t cnt = (uint32_t)std::min<size_t>(chunk, n - s0);
        std::fill(stage.begin(), stage.begin() + (size_t)cnt * g.row_stride, 0);
        for (; i < items->n && item_slot[i] < s0 + cnt; i++)
          memcpy(&stage[(size_t)(item_slot[i] - s0) * g.row_stride],
                 (const unsigned char *)items->vectors + i * items->stride, vb);
        HIP_TRY(hipMemcpyAsync(b->d_rows.p + (size_t)s0 * g.row_stride, stage.data(),
                               (size_t)cnt * g.row_stride, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
      }
    }
    HIP_TRY(hipMemcpyAsync(b->d_level.p, b->level.data(), n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_upper_idx.p, b->upper_idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (!b->order.empty())
      HIP_TRY(hipMemcpyAsync(b->d_order.p, b->order.data(), b->order.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_eps.p, b->entry_points.data(), b->entry_points.size() * 4,
                           hipMemcpyHostToDevice, st));
    std::vector<float> norms;
    if (has_norm) {
      norms.assign(n, 0.f);
      for (uint64_t i = 0; i < items->n; i++)
        memcpy(&norms[item_slot[i]], (const unsigned char *)items->headers + (size_t)i * hb, 4);
      HIP_TRY(hipMemcpyAsync(b->d_norms.p, norms.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (inc && n) {
    // the previous graph, as Links records: ascending ids, HNY_SENT padded
    const hny_prev_graph *pg = inc->prev;
    std::vector<u32> d0((size_t)n * o.M0, HNY_SENT), du(std::max<size_t>(nup * o.M, 1), HNY_SENT);
    for (uint64_t r = 0; r < pg->n_records; r++) {
      uint32_t s = slot_of(pg->rec_item[r]), ll = pg->rec_layer[r];
      uint64_t c = pg->rec_offset[r + 1] - pg->rec_offset[r];
      uint32_t cap = ll == 0 ? o.M0 : o.M;
      if (c > cap) return fail(HNY_ERR_UNSUPPORTED, "old record (%u, %u) has %llu > %u links", U[s], ll,
                               (unsigned long long)c, cap);
      u32 *dst = ll == 0 ? &d0[(size_t)s * o.M0]
                         : &du[((size_t)b->upper_idx[s] * b->up_layers + (ll - 1)) * o.M];
      for (uint64_t k = 0; k < c; k++) dst[k] = slot_of(pg->neighbours[pg->rec_offset[r] + k]);
    }
    HIP_TRY(b->d_d0_ids.alloc(d0.size()));
    HIP_TRY(b->d_du_ids.alloc(du.size()));
    HIP_TRY(b->d_has_vec.alloc(n));
    HIP_TRY(b->d_deleted.alloc(n));
    HIP_TRY(b->d_old_recs.alloc(std::max<size_t>(b->old_recs.size(), 1)));
    HIP_TRY(hipMemcpyAsync(b->d_d0_ids.p, d0.data(), d0.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_du_ids.p, du.data(), du.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_has_vec.p, has_vec.data(), n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_deleted.p, b->deleted.data(), n, hipMemcpyHostToDevice, st));
    if (!b->old_recs.empty())
      HIP_TRY(hipMemcpyAsync(b->d_old_recs.p, b->old_recs.data(), b->old_recs.size() * 8,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    g.d0_ids = b->d_d0_ids.p;
    g.du_ids = b->d_du_ids.p;
    g.has_vec = b->d_has_vec.p;
  }
  rc = reset_graph(b.get());
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  b->t_upload = now_s() - t0;
  b->t_build0 = now_s();
  *out = b.release();
  return HNY_OK;
}

int hny_builder_create(const hny_build_opts *opts, const hny_items *items, hny_builder **out) {
  return create_impl(opts, items, nullptr, out);
}

int hny_builder_reset(hny_builder *b) {
  if (!b) return fail(HNY_ERR_INVALID_ARG, "null builder");
  HIP_TRY(hipSetDevice(b->device));
  return reset_graph(b);
}

static void start_export_prefault(hny_builder *b);
int hny_builder_next_batch(hny_builder *b, hny_batch *out) {
  if (!b || !out) return fail(HNY_ERR_INVALID_ARG, "null argument");
  if (b->in_batch) return fail(HNY_ERR_INVALID_ARG, "previous batch not applied");
  if (b->finalized && b->pos < b->order.size()) return fail(HNY_ERR_INVALID_ARG, "graph already finalised");
  memset(out, 0, sizeof *out);
  if (b->pos >= b->order.size()) return HNY_OK;
  if (b->pos == 0) start_export_prefault(b);
  size_t gend = group_end(b, b->pos);
  uint32_t L = b->order_level[b->pos];
  uint64_t bs = hny_batch_size(b->frac, b->bmax, b->n_done);
  bs = std::min<uint64_t>(bs, gend - b->pos);
  out->first = b->pos;
  out->count = (uint32_t)bs;
  out->level = L;
  out->n_layers = L + 1;
  out->sel_stride_u64 = (L + 1) * (cap_of(b, L) + 1);
  b->cur = *out;
  b->in_batch = true;
  return HNY_OK;
}

static hipError_t next_sync_event(hny_builder *b, hipEvent_t *ev) {
  if (b->sync_used == b->sync_evs.size()) {
    hipEvent_t e;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return rc;
    b->sync_evs.push_back(e);
  }
  *ev = b->sync_evs[b->sync_used++];
  return hipSuccess;
}

int hny_builder_search(hny_builder *b, uint32_t lo, uint32_t hi, void *sel_dev) {
  if (!b || !b->in_batch) return fail(HNY_ERR_INVALID_ARG, "no current batch");
  if (lo > hi || hi > b->cur.count) return fail(HNY_ERR_INVALID_ARG, "bad member range");
  if (lo == hi) return HNY_OK;
  HIP_TRY(hipSetDevice(b->device));
  const uint32_t L = b->cur.level, cs = cap_of(b, L);
  u64 *sel = sel_dev ? (u64 *)sel_dev : b->d_sel.p;
  auto walk_args = [&](int32_t l, uint32_t clo, uint32_t chi, u32 *queue) {
    WalkArgs w{};
    w.q_slots = b->d_order.p + b->cur.first;
    w.lo = clo;
    w.hi = chi;
    w.layer = (u32)l;
    w.ef = b->o.ef_construction;
    w.first = (l == (int32_t)L);
    w.reader_mode = 0;
    w.entry_points = b->d_eps.p;
    w.n_entry_points = (u32)b->entry_points.size();
    w.sel = sel;
    w.sel_stride = b->cur.sel_stride_u64;
    w.cap_sel = cs;
    w.batch_level = L;
    w.cand = b->d_cand.p;
    w.cand_n = b->d_cand_n.p;
    w.rcap = b->rcap;
    w.bits = b->d_bits.p;
    w.bits_words = b->bits_words;
    w.vlog = b->d_vlog.p;
    w.log_cap = b->log_cap;
    w.vis_slots = vis_slots_for(b, b->d_res_global.p ? 0u : w.rcap);
    w.eps_cap = eps_cap_of(b);
    w.queue = queue;
    w.res_global = b->d_res_global.p; // null unless the result sets outgrow the LDS (res_capacity)
    vis_buckets_for(b, w);
    {
      // one-chunk register beam (k_walk<.., RC = 1>, rows <= 512 B): no result set of this builder's walks exceeds 64
      // entries — ef, the entry points (all pushed without a capacity check), a top layer that only grows
      const uint64_t neps = b->entry_points.size();
      const uint64_t most = std::max<uint64_t>(std::max<uint64_t>(b->o.ef_construction, neps),
                                               neps > 1 ? std::max<uint64_t>(b->top_layer_nodes, neps) : 0);
      w.rb_one = most <= 64 && b->o.M <= 64 && env_int("HNY_RB_ONE", 1) != 0;
    }
    return w;
  };
  auto prune_args = [&](int32_t l, uint32_t clo, uint32_t chi) {
    PruneArgs p{};
    p.q_slots = b->d_order.p + b->cur.first;
    p.lo = clo;
    p.hi = chi;
    p.layer = (u32)l;
    p.cap = cs; // NB: from the item's top level, hnsw.rs:317
    p.cand = b->d_cand.p;
    p.cand_n = b->d_cand_n.p;
    p.rcap = b->rcap;
    p.sel = sel;
    p.sel_stride = b->cur.sel_stride_u64;
    p.cap_sel = cs;
    p.batch_level = L;
    p.list_global = b->rcap > HNY_RES_LDS_MAX ? 1u : 0u;
    // neighbouring members of the long-row prune on one XCD, like the walk's tiles: k_prune_wg's L2 hit rate 0.12 ->
    // 0.19, its fabric reads 368 -> 338 GB at C2 — and not a microsecond of its 66 ms (profiles/r05_prune_xcd_tile_fetch.txt:
    // the workgroup prune is bound by its barriers and dependent chains, not by bytes)
    p.xcd_tile = (u32)std::max(0, env_int("HNY_PRUNE_XCD_TILE", 512));
    return p;
  };
  auto launch_prune = [&](const PruneArgs &p, hipStream_t st) -> hipError_t {
    if (!b->wave_prune_only && hnyk_prune_n8_ok(b->g, p, b->shape)) {
      // short rows: one wave per query; as many selected rows staged as 6 KB per wave hold (20 waves per CU)
      const int sl = (int)std::min<uint32_t>(HNY_MAX_CAP, std::max<uint32_t>(1, 6144u / b->g.row_stride));
      return hnyk_prune_n8(b->g, p, b->shape, sl, (int)std::min<uint32_t>(p.hi - p.lo, 5120u), st);
    }
    if (b->wave_prune_only)
      return hnyk_prune(b->g, p, b->shape, (int)std::min<uint32_t>(p.hi - p.lo, b->walk_slots), st);
    return hnyk_prune_wg(b->g, p, b->shape, b->stage_rows, 4, (int)std::min<uint32_t>(p.hi - p.lo, 2048), st);
  };
  // XCD-tiled work queue of the level-0 walks (WalkArgs.xcd_tile): rows of 1 KB and more, where the walk
  // is HBM-bound and neighbouring queries on one L2 save fabric traffic (C2 walk 0.312 -> 0.289 s, C3
  // 0.601 -> 0.573 s; tiles of 256..1024 members alike); neutral within the noise on 136 / 516-byte
  // rows, so off there.  HNY_XCD_TILE overrides (0 = one counter).
  const u32 xcd_tile = (u32)std::max(0, env_int("HNY_XCD_TILE", b->g.row_stride >= 1024u ? 512 : 0));
  u32 *queues = b->d_nseg.p + 4; // 16 work counters: the descent + one per layer of the batch
  u32 *xqueues = queues + 16;    // the same 16, as 8 per-XCD counters each (WalkArgs.xcd_tile)
  HIP_TRY(hipMemsetAsync(queues, 0, (16 + 8 * 16) * 4, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_pool_ctr.p, 0, b->d_pool_ctr.n * 4, b->stream));
  b->pool_ctr_used = 0;
  // one walk launch: rows <= 512 B go to the four-queries-per-wave kernel first, and the one-wave
  // kernel then takes the members it gave up on (none, normally) from the retry list
  auto launch_walk_fast = [&](const WalkArgs &w, hipStream_t st) -> hipError_t {
    return hnyk_walk(b->g, w, b->shape, (int)std::min<uint32_t>(w.hi - w.lo, b->walk_slots), st);
  };
  // one walk launch + its safety net: the members whose tie pool overflowed (none, normally) are listed on
  // the device and walked again by k_walk_heap, which reads the count itself — no host round trip
  auto launch_walk = [&](WalkArgs w, hipStream_t st) -> hipError_t {
    w.key_base = w.lo;
    if (b->pool_ctr_used + 4 > b->d_pool_ctr.n) { // (more walk launches in one search call than counter sets)
      hipError_t e = hipMemsetAsync(b->d_pool_ctr.p, 0, b->d_pool_ctr.n * 4, st);
      if (e != hipSuccess) return e;
      b->pool_ctr_used = 0;
    }
    u32 *pc = b->d_pool_ctr.p + b->pool_ctr_used;
    b->pool_ctr_used += 4;
    const bool no_retry = env_int("HNY_NO_POOL_RETRY", 0) != 0; // tests: an overflow is an error again
    w.pool_retry = no_retry ? nullptr : b->d_pool_retry.p;
    w.n_pool_retry = pc;
    w.force_pool = (u32)std::max(0, env_int("HNY_POOL_FORCE_RETRY", 0));
    hipError_t e = launch_walk_fast(w, st);
    if (e != hipSuccess || no_retry) return e;
    WalkArgs h = w;
    h.queue = pc + 1;
    h.xcd_tile = 0;
    h.heap_c = b->heap_c_ptr;
    h.heap_r = b->d_heap_r.p;
    h.heap_c_cap = b->heap_c_cap;
    h.heap_r_cap = b->heap_r_cap;
    if (b->heap_grid1) { // first tier: small heaps, many blocks; what outgrows them is listed for the second
      WalkArgs h1 = h;
      h1.heap_c_cap = b->heap_c_cap1;
      h1.pool_retry2 = b->d_pool_retry2.p;
      h1.n_pool_retry2 = pc + 2;
      e = hnyk_walk_heap(b->g, h1, b->shape, (int)std::min<uint32_t>(w.hi - w.lo, b->heap_grid1), st);
      if (e != hipSuccess) return e;
      h.pool_retry = b->d_pool_retry2.p;
      h.n_pool_retry = pc + 2;
      h.queue = pc + 3;
    }
    return hnyk_walk_heap(b->g, h, b->shape, (int)std::min<uint32_t>(w.hi - w.lo, b->heap_grid), st);
  };

  const uint32_t cnt = hi - lo;
  if (b->locality && b->max_level > L && cnt >= 2048) {
    // batch in LOCALITY ORDER: (1) greedy descent for every member, recording the closest node of
    // the last greedy layers as a coarse-to-fine key; (2) sort the members by that key; (3) the beam
    // searches and prunes of every layer take the members in that order, so that the waves running
    // at the same time work in the same region of the graph and share candidate rows in L2 /
    // Infinity Cache.  Results are stored per member: the build is unchanged, only its memory traffic.
    WalkArgs d = walk_args(L, lo, hi, queues + 0);
    d.descend_only = 1;
    d.eps_out = b->d_eps0.p;
    d.key_out = b->d_lkey_a.p;
    prof_begin(b, EV_WALK);
    b->n_walk_dispatch++;
    HIP_TRY(launch_walk(d, b->stream));
    HIP_TRY(hnyk_iota_u64(b->d_perm_a.p, lo, cnt, b->stream));
    size_t tmp = b->sort_tmp_bytes;
    HIP_TRY(hnyk_sort_pairs48(b->d_sort_tmp.p, tmp, b->d_lkey_a.p, b->d_lkey_b.p, b->d_perm_a.p,
                              b->d_perm_b.p, cnt, b->stream));
    for (int32_t l = (int32_t)L; l >= 0; l--) { // hnsw.rs:312-325
      WalkArgs w = walk_args(l, lo, hi, queues + (l + 1));
      if (l == (int32_t)L) {
        w.first = 0;
        w.eps_in = b->d_eps0.p;
      } else {
        prof_begin(b, EV_WALK);
      }
      w.perm = b->d_perm_b.p;
      if (xcd_tile && cnt >= 16u * xcd_tile) {
        w.xcd_tile = xcd_tile;
        w.queue = xqueues + 8 * (l + 1);
      }
      b->n_walk_dispatch++;
      HIP_TRY(launch_walk(w, b->stream));
      prof_end(b);
      PruneArgs p = prune_args(l, lo, hi);
      p.perm = b->d_perm_b.p;
      prof_begin(b, EV_PRUNE);
      HIP_TRY(launch_prune(p, b->stream));
      prof_end(b);
    }
    return HNY_OK;
  }
  for (int32_t l = (int32_t)L; l >= 0; l--) { // hnsw.rs:312-325
    WalkArgs w = walk_args(l, lo, hi, queues + l);
    prof_begin(b, EV_WALK);
    b->n_walk_dispatch++;
    HIP_TRY(launch_walk(w, b->stream));
    prof_end(b);
    prof_begin(b, EV_PRUNE);
    HIP_TRY(launch_prune(prune_args(l, lo, hi), b->stream));
    prof_end(b);
  }
  return HNY_OK;
}

// phase 2 up to and including k_apply: link ops emitted, sorted, segmented; every target whose list
// cannot overflow is done, the others are listed in d_deferred (first op of their segment)
static int apply_front(hny_builder *b, const void *sel_dev, ApplyArgs &a) {
  const uint32_t L = b->cur.level, cs = cap_of(b, L);
  const u64 *sel = sel_dev ? (const u64 *)sel_dev : b->d_sel.p;
  const u32 n_ops = b->cur.count * (L + 1) * cs * 2;
  EmitArgs e{};
  e.q_slots = b->d_order.p + b->cur.first;
  e.count = b->cur.count;
  e.sel = sel;
  e.sel_stride = b->cur.sel_stride_u64;
  e.cap_sel = cs;
  e.batch_level = L;
  e.keys = b->d_keys_a.p;
  e.vals = b->d_vals_a.p;
  prof_begin(b, EV_SORT);
  HIP_TRY(hnyk_emit(b->g, e, b->stream));
  size_t tmp = b->sort_tmp_bytes;
  // (sorting only the (layer, target) bits and relying on the stability of the radix sort for the
  // sequence order — 3 passes instead of 8 — crashed non-deterministically inside the test suite:
  // rocPRIM switches algorithms with size and bit range; the full 64-bit key is sorted)
  HIP_TRY(hnyk_sort_pairs(b->d_sort_tmp.p, tmp, b->d_keys_a.p, b->d_keys_b.p, b->d_vals_a.p,
                          b->d_vals_b.p, n_ops, 0, 64, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_nseg.p, 0, 8, b->stream));
  HIP_TRY(hnyk_segments(b->d_keys_b.p, n_ops, b->d_seg_start.p, b->d_nseg.p, b->stream));
  prof_end(b);
  a = ApplyArgs{};
  a.keys = b->d_keys_b.p;
  a.vals = b->d_vals_b.p;
  a.n_ops = n_ops;
  a.seg_start = b->d_seg_start.p;
  a.n_seg = b->d_nseg.p;
  a.deferred = b->wave_prune_only ? nullptr : b->d_deferred.p;
  a.n_deferred = b->d_nseg.p + 1;
  const int grid = (int)std::min<u32>(std::max<u32>(n_ops / 2, 1), 8192);
  prof_begin(b, EV_APPLY);
  if (a.deferred)
    HIP_TRY(hnyk_apply_append(b->g, a, b->stream)); // appends: one thread per target; overflowing lists deferred
  else // strict mode / rows beyond 8 KB: one wave per target (one lane per list slot), prunes included
    HIP_TRY(hnyk_apply(b->g, a, b->shape, grid, b->stream));
  prof_end(b);
  b->cur_n_ops = n_ops;
  return HNY_OK;
}

// the targets whose list overflows: one wave each on short rows (k_apply_n8), a 256-thread workgroup otherwise
static hipError_t launch_apply_deferred(hny_builder *b, const ApplyArgs &a, u32 work) {
  if (hnyk_apply_n8_ok(b->g, b->shape)) {
    const int sl = (int)std::min<uint32_t>(HNY_MAX_CAP, std::max<uint32_t>(1, 6144u / b->g.row_stride));
    return hnyk_apply_n8(b->g, a, b->shape, sl, (int)std::min<u32>(work, 5120u), b->stream);
  }
  return hnyk_apply_wg(b->g, a, b->shape, b->stage_rows, (int)std::min<u32>(work, 2048u), b->stream);
}

static void apply_back(hny_builder *b) {
  b->pos += b->cur.count;
  b->n_done += b->cur.count;
  b->n_batches++;
  b->in_batch = false;
  b->apply_open = false;
}

int hny_builder_apply(hny_builder *b, const void *sel_dev) {
  if (!b || !b->in_batch || b->apply_open) return fail(HNY_ERR_INVALID_ARG, "no current batch");
  HIP_TRY(hipSetDevice(b->device));
  ApplyArgs a;
  if (int rc = apply_front(b, sel_dev, a)) return rc;
  if (!b->wave_prune_only) {
    prof_begin(b, EV_APPLY);
    HIP_TRY(launch_apply_deferred(b, a, std::max<u32>(b->cur_n_ops / 8, 1)));
    prof_end(b);
  }
  apply_back(b);
  return HNY_OK;
}

uint32_t hny_builder_exch_stride_u64(const hny_builder *b) {
  return b ? 2u + std::max(b->o.M, b->o.M0) : 0u;
}

int hny_builder_apply_begin(hny_builder *b, const void *sel_dev, uint32_t *n_deferred) {
  if (!b || !b->in_batch || b->apply_open || !n_deferred) return fail(HNY_ERR_INVALID_ARG, "no current batch");
  HIP_TRY(hipSetDevice(b->device));
  ApplyArgs a;
  if (int rc = apply_front(b, sel_dev, a)) return rc;
  u32 nd = 0;
  if (!b->wave_prune_only) {
    HIP_TRY(hipMemcpyAsync(&nd, b->d_nseg.p + 1, 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (nd > 1) { // canonical order: k_apply appended in atomic order, which differs between replicas
      size_t tmp = b->sort_tmp_bytes;
      HIP_TRY(hnyk_sort_u32(b->d_sort_tmp.p, tmp, b->d_deferred.p, b->d_deferred_b.p, nd, b->stream));
      HIP_TRY(hipMemcpyAsync(b->d_deferred.p, b->d_deferred_b.p, (size_t)nd * 4, hipMemcpyDeviceToDevice, b->stream));
    }
  }
  b->cur_n_def = nd;
  b->apply_open = true;
  *n_deferred = nd;
  return HNY_OK;
}

int hny_builder_apply_deferred(hny_builder *b, uint32_t rank, uint32_t world, void *exch_dev) {
  if (!b || !b->apply_open || world == 0 || rank >= world || (world > 1 && !exch_dev))
    return fail(HNY_ERR_INVALID_ARG, "hny_builder_apply_deferred: bad argument or no open apply");
  HIP_TRY(hipSetDevice(b->device));
  if (b->wave_prune_only || b->cur_n_def == 0) return HNY_OK;
  ApplyArgs a{};
  a.keys = b->d_keys_b.p;
  a.vals = b->d_vals_b.p;
  a.n_ops = b->cur_n_ops;
  a.seg_start = b->d_seg_start.p;
  a.n_seg = b->d_nseg.p;
  a.deferred = b->d_deferred.p;
  a.n_deferred = b->d_nseg.p + 1;
  a.shard_rank = rank;
  a.shard_world = world;
  a.exch_stride = hny_builder_exch_stride_u64(b);
  const u32 per = (b->cur_n_def + world - 1) / world;
  a.exch = world > 1 ? (u64 *)exch_dev + (size_t)rank * per * a.exch_stride : nullptr;
  prof_begin(b, EV_APPLY);
  HIP_TRY(launch_apply_deferred(b, a, std::max<u32>(per, 1)));
  prof_end(b);
  return HNY_OK;
}

int hny_builder_apply_merge(hny_builder *b, const void *exch_all_dev, uint32_t rank, uint32_t world) {
  if (!b || !b->apply_open || world == 0 || rank >= world)
    return fail(HNY_ERR_INVALID_ARG, "hny_builder_apply_merge: bad argument or no open apply");
  HIP_TRY(hipSetDevice(b->device));
  if (world > 1 && b->cur_n_def && !b->wave_prune_only) {
    if (!exch_all_dev) return fail(HNY_ERR_INVALID_ARG, "hny_builder_apply_merge: null exchange buffer");
    const u32 per = (b->cur_n_def + world - 1) / world;
    prof_begin(b, EV_APPLY);
    HIP_TRY(hnyk_apply_merge(b->g, (const u64 *)exch_all_dev, b->cur_n_def, world, rank, per,
                             hny_builder_exch_stride_u64(b), b->stream));
    prof_end(b);
  }
  apply_back(b);
  return HNY_OK;
}

int hny_builder_set_profiling(hny_builder *b, int on) {
  if (!b) return fail(HNY_ERR_INVALID_ARG, "null builder");
  b->profiling = on != 0;
  return HNY_OK;
}

void *hny_builder_stream(hny_builder *b) { return b ? (void *)b->stream : nullptr; }

int hny_builder_sync(hny_builder *b) {
  if (!b) return fail(HNY_ERR_INVALID_ARG, "null builder");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return HNY_OK;
}

// the device error words are reported once, then cleared (they sit behind the counters in d_stats)
static hipError_t clear_error_counters(hny_builder *b) {
  static_assert(ST_ERR_GAPS_OVERFLOW == ST_ERR_RES_OVERFLOW + 2 && ST_ERR_ITER == ST_ERR_RES_OVERFLOW + 1,
                "error words are contiguous");
  hipError_t e = hipMemsetAsync(b->d_stats.p + ST_ERR_RES_OVERFLOW, 0, 3 * sizeof(u64), b->stream);
  if (e != hipSuccess) return e;
  return hipMemsetAsync(b->d_stats.p + ST_POOL_OVERFLOW, 0, sizeof(u64), b->stream);
}
// `candidates` entries that tie with the result set's maximum after being evicted from it stay poppable
// in a 128-slot pool; more than that at once only happens when the distance takes a handful of values
// (3-bit Hamming codes with lists of hundreds of links).  The dropped ones would make the graph differ
// from the reference's without a trace, so the call fails instead.
static int pool_overflow_error(hny_builder *b, unsigned long long n) {
  (void)clear_error_counters(b);
  return fail(HNY_ERR_DEVICE,
              "tie pool overflow: %llu candidates that tie with the result set's maximum were dropped (more than %d "
              "at once) — the graph would differ from the reference's", n, HNY_POOL_CAP);
}

// what the kernels report through the error words of a counter block: reported once (HNY_ERR_DEVICE), then
// cleared, so that a later reset / search starts clean
static int device_error_words(hny_builder *b, const u64 *stats) {
  if (stats[ST_ERR_RES_OVERFLOW] || stats[ST_ERR_ITER] || stats[ST_ERR_GAPS_OVERFLOW]) {
    (void)clear_error_counters(b);
    return fail(HNY_ERR_DEVICE,
                "kernel overflow: res=%llu iter=%llu gaps=%llu (res: a walk's result set outgrew its %u entries — a walk "
                "that starts from more entry points than its ef keeps every closer point, see res_capacity / DESIGN.md limits)",
                stats[ST_ERR_RES_OVERFLOW], stats[ST_ERR_ITER], stats[ST_ERR_GAPS_OVERFLOW], b->rcap);
  }
  if (stats[ST_POOL_OVERFLOW]) return pool_overflow_error(b, stats[ST_POOL_OVERFLOW]);
  return HNY_OK;
}

// Export arrays released by hny_graph_free, kept for the next export — OFF unless the caller asks for it
// (hny_set_graph_cache): a service that rebuilds in a loop saves the munmap of the old arrays and the first touch of
// the new ones (C4: 1.4 GB each way, ~100 ms per build); anybody else gets plain malloc / free and holds nothing.
namespace {
struct GraphBufCache {
  std::mutex mu;
  size_t limit = 0; // bytes the cache may hold; 0 = off
  void *p[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  size_t held() const { return cap[0] + cap[1] + cap[2] + cap[3]; }
  void drop_all() {
    for (int i = 0; i < 4; i++) {
      free(p[i]);
      p[i] = nullptr;
      cap[i] = 0;
    }
  }
  ~GraphBufCache() { drop_all(); }
} g_gcache;
// a cached array of at least `bytes` (and not more than twice that), or null
void *gcache_take(int slot, size_t bytes, size_t *cap_out) {
  std::lock_guard<std::mutex> lk(g_gcache.mu);
  if (g_gcache.p[slot] && g_gcache.cap[slot] >= bytes && g_gcache.cap[slot] / 2 <= bytes + ((size_t)1 << 20)) {
    void *q = g_gcache.p[slot];
    *cap_out = g_gcache.cap[slot];
    g_gcache.p[slot] = nullptr;
    g_gcache.cap[slot] = 0;
    return q;
  }
  return nullptr;
}
void gcache_release(int slot, void *q) {
  if (!q) return;
  void *drop = q;
  {
    std::lock_guard<std::mutex> lk(g_gcache.mu);
    const size_t cap = g_gcache.limit ? malloc_usable_size(q) : 0;
    if (cap >= ((size_t)1 << 20) && cap > g_gcache.cap[slot] && g_gcache.held() - g_gcache.cap[slot] + cap <= g_gcache.limit) {
      drop = g_gcache.p[slot];
      g_gcache.p[slot] = q;
      g_gcache.cap[slot] = cap;
    }
  }
  free(drop);
}
} // namespace
void hny_set_graph_cache(size_t max_bytes) {
  std::lock_guard<std::mutex> lk(g_gcache.mu);
  g_gcache.limit = max_bytes;
  if (g_gcache.held() > max_bytes) g_gcache.drop_all();
}

// Prepare the export arrays of the build that is starting (hny_builder.xbuf): sizes are upper bounds known when
// the builder is created (which records exist never changes; a list holds at most its cap).  The neighbour array
// keeps its prepared capacity (handing the unused tail back costs a munmap of touched pages, as much as the first
// touch saved); hny_graph_free returns all of it.
static void start_export_prefault(hny_builder *b) {
  if (!b->will_export) return;
  b->xbuf.join();
  const size_t want[4] = {(size_t)std::max<uint64_t>(b->nrec_bound, 1) * 4, (size_t)std::max<uint64_t>(b->nrec_bound, 1),
                          (size_t)(b->nrec_bound + 1) * 8, (size_t)std::max<uint64_t>(b->nbr_bound, 1) * 4};
  if (want[0] + want[1] + want[2] + want[3] < ((size_t)8 << 20)) return; // small: finish() allocates
  bool have = true;
  for (int i = 0; i < 4; i++) {
    if (!(b->xbuf.p[i] && b->xbuf.cap[i] >= want[i])) { // (else: a build reset before its finish() left it here)
      size_t cap = 0;
      if (void *q = gcache_take(i, want[i], &cap)) { // released by an earlier graph: allocated and touched already
        free(b->xbuf.p[i]);
        b->xbuf.p[i] = q;
        b->xbuf.cap[i] = cap;
      }
    }
    have = have && b->xbuf.p[i] && b->xbuf.cap[i] >= want[i];
  }
  if (have) return;
  auto *x = &b->xbuf;
  x->th = std::thread([x, want]() {
    for (int i = 0; i < 4; i++) {
      if (x->p[i] && x->cap[i] >= want[i]) continue;
      free(x->p[i]);
      x->p[i] = malloc(want[i]);
      x->cap[i] = x->p[i] ? want[i] : 0;
      volatile unsigned char *q = (volatile unsigned char *)x->p[i];
      for (size_t o = 0; q && o < want[i]; o += 4096) q[o] = 0;
    }
  });
}
// one export array: the prepared one when it is large enough, else a fresh allocation
static void *take_export_buf(hny_builder *b, int slot, size_t bytes) {
  b->xbuf.join();
  if (b->xbuf.p[slot] && b->xbuf.cap[slot] >= bytes) {
    void *q = b->xbuf.p[slot];
    b->xbuf.p[slot] = nullptr;
    b->xbuf.cap[slot] = 0;
    return q;
  }
  size_t cap = 0;
  if (void *q = gcache_take(slot, bytes, &cap)) return q;
  return malloc(bytes);
}

void hny_graph_free(hny_graph *g) {
  if (!g) return;
  gcache_release(0, (void *)g->rec_item);
  gcache_release(1, (void *)g->rec_layer);
  gcache_release(2, (void *)g->rec_offset);
  gcache_release(3, (void *)g->neighbours);
  free((void *)g->entry_points);
  free(g);
}

// the write loop's input (hnsw.rs:191-213): one record per (item, layer), ids deduplicated
int hny_builder_finish(hny_builder *b, hny_graph **out) {
  if (!b || !out) return fail(HNY_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (b->pos < b->order.size() || b->in_batch) return fail(HNY_ERR_INVALID_ARG, "build not finished");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->t_build = now_s() - b->t_build0;
  double t0 = now_s();
  const uint32_t n = b->n, M = b->o.M, M0 = b->o.M0, ml = b->max_level;
  u64 stats[ST_COUNT] = {0};
  HIP_TRY(hipMemcpy(stats, b->d_stats.p, sizeof stats, hipMemcpyDeviceToHost));
  if (int rc = device_error_words(b, stats)) return rc;
  // finalise every list on the device (sort + dedup), then copy through pinned staging
  const uint32_t upl = b->up_layers;
  const size_t nup = (size_t)b->n_upper * upl;
  if (!b->finalized) {
    HIP_TRY(hnyk_finalize_lists(b->d_l0_ids.p, b->d_fin_cnt0.p, n, M0, b->stream));
    HIP_TRY(hnyk_finalize_lists(b->d_up_ids.p, b->d_fin_cntu.p, (u32)nup, M, b->stream));
    b->finalized = true;
  }
  // the counts first (small), then the lists (128 MB at C2): the record offsets are computed on the
  // host while the lists are still in flight
  if (n) HIP_TRY(hipMemcpyAsync(b->h_cnt0, b->d_fin_cnt0.p, (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
  if (nup) HIP_TRY(hipMemcpyAsync(b->h_cntu, b->d_fin_cntu.p, nup * 4, hipMemcpyDeviceToHost, b->stream));
  hipEvent_t ev_counts;
  HIP_TRY(next_sync_event(b, &ev_counts));
  HIP_TRY(hipEventRecord(ev_counts, b->stream));
  // The layer-0 lists travel in as many pieces as the host has compaction threads (the same slot ranges), an
  // event behind each: thread t compacts its range as soon as ITS piece has landed, while the later pieces are
  // still on the bus — the export costs the transfer plus one piece's compaction instead of their sum (C4: 1.28 GB
  // down at ~30 GB/s, 0.84 GB compacted: 54 -> 44 ms; C5 29 -> 23 ms).  The small upper-layer lists go first.
  unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency() / 2));
  if (n < 10000) nt = 1;
  if (nup) HIP_TRY(hipMemcpyAsync(b->h_up, b->d_up_ids.p, nup * M * 4, hipMemcpyDeviceToHost, b->stream));
  hipEvent_t up_landed;
  HIP_TRY(next_sync_event(b, &up_landed));
  HIP_TRY(hipEventRecord(up_landed, b->stream));
  // (pieces of at least 16 MB: C2's 128 MB go in 8, not in 64 — a piece costs a copy command and an event.  The
  // download itself runs at ~30 GB/s on these boxes whether one stream or two alternate on the pieces — measured.)
  std::vector<hipEvent_t> l0_landed(nt, nullptr);
  {
    const unsigned pieces = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt, (uint64_t)n * M0 * 4 / ((uint64_t)16 << 20)));
    for (unsigned c = 0; c < pieces && n; c++) {
      const unsigned t0 = (unsigned)((uint64_t)c * nt / pieces), t1 = (unsigned)((uint64_t)(c + 1) * nt / pieces); // threads of the piece
      const size_t lo = (size_t)((uint64_t)n * t0 / nt), hi = (size_t)((uint64_t)n * t1 / nt);
      if (hi > lo)
        HIP_TRY(hipMemcpyAsync(b->h_l0 + lo * M0, b->d_l0_ids.p + lo * M0, (hi - lo) * M0 * 4, hipMemcpyDeviceToHost, b->stream));
      hipEvent_t ev;
      HIP_TRY(next_sync_event(b, &ev));
      HIP_TRY(hipEventRecord(ev, b->stream));
      for (unsigned t = t0; t < t1; t++) l0_landed[t] = ev;
    }
  }
  const u32 *l0 = b->h_l0, *up = b->h_up;

  // every inserted item owns a (possibly empty) record on layers 0..=level (add_in_layers_below,
  // hnsw.rs:419-424); after an incremental build every surviving old record is rewritten too
  // (fill_gaps_from_deleted puts it in memory, :398/:410) and deleted items own nothing
  // (delete_links_from_db, writer.rs:692-718).  Which records exist never changes during a builder's
  // life: the record table (first record of every slot, item id and layer of every record) is
  // computed once and reused by every finish().
  auto rec_mask = [&](uint32_t s) -> uint32_t {
    if (b->deleted[s]) return 0u;
    uint32_t m = b->old_mask[s];
    if (b->ins_level[s] >= 0) m |= (2u << b->ins_level[s]) - 1u;
    return m;
  };
  auto parallel = [&](auto &&fn) { // fn(thread, lo, hi) over the slots
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++)
      th.emplace_back(fn, t, (uint32_t)((uint64_t)n * t / nt), (uint32_t)((uint64_t)n * (t + 1) / nt));
    fn(0u, 0u, (uint32_t)((uint64_t)n / nt));
    for (auto &t : th) t.join();
  };
  if (b->rec_first.size() != (size_t)n + 1) {
    b->rec_first.assign((size_t)n + 1, 0);
    for (uint32_t s = 0; s < n; s++) b->rec_first[s + 1] = b->rec_first[s] + (uint32_t)__builtin_popcount(rec_mask(s));
    const uint64_t nr = b->rec_first[n];
    b->rec_item_t.resize(nr);
    b->rec_layer_t.resize(nr);
    parallel([&](unsigned, uint32_t lo, uint32_t hi) {
      for (uint32_t s = lo; s < hi; s++) {
        uint64_t r = b->rec_first[s];
        for (uint32_t m = rec_mask(s), l = 0; m; m >>= 1, l++)
          if (m & 1) {
            b->rec_item_t[r] = b->ids[s];
            b->rec_layer_t[r] = (uint8_t)l;
            r++;
          }
      }
    });
  }
  const std::vector<uint64_t> &rec_first = b->rec_first;
  const uint64_t nrec = rec_first[n];
  // the graph owns its arrays from the start, so that every error return below frees them
  std::unique_ptr<hny_graph, void (*)(hny_graph *)> gh((hny_graph *)calloc(1, sizeof(hny_graph)), hny_graph_free);
  hny_graph *g = gh.get();
  uint32_t *rec_item = (uint32_t *)take_export_buf(b, 0, std::max<uint64_t>(nrec, 1) * 4);
  uint8_t *rec_layer = (uint8_t *)take_export_buf(b, 1, std::max<uint64_t>(nrec, 1));
  uint64_t *rec_off = (uint64_t *)take_export_buf(b, 2, (nrec + 1) * 8);
  if (!g || !rec_item || !rec_layer || !rec_off) {
    free(rec_item);
    free(rec_layer);
    free(rec_off);
    return fail(HNY_ERR_OOM, "out of host memory for %llu records", (unsigned long long)nrec);
  }
  g->rec_item = rec_item;
  g->rec_layer = rec_layer;
  g->rec_offset = rec_off;
  rec_off[0] = 0;
  HIP_TRY(hipEventSynchronize(ev_counts));
  // offsets = prefix sum over the device-computed counts, in record order: per-thread sums of a slot
  // range, a scan of those, then every thread fills its range (the record table is copied alongside)
  auto count_of = [&](uint32_t s, uint32_t l) -> uint32_t {
    return l == 0 ? b->h_cnt0[s] : b->h_cntu[(size_t)b->upper_idx[s] * upl + (l - 1)];
  };
  std::vector<uint64_t> part(nt + 1, 0);
  parallel([&](unsigned t, uint32_t lo, uint32_t hi) {
    uint64_t sum = 0;
    for (uint32_t s = lo; s < hi; s++)
      for (uint32_t m = rec_mask(s), l = 0; m; m >>= 1, l++)
        if (m & 1) sum += count_of(s, l);
    part[t + 1] = sum;
    if (hi > lo) {
      memcpy(rec_item + rec_first[lo], b->rec_item_t.data() + rec_first[lo], (rec_first[hi] - rec_first[lo]) * 4);
      memcpy(rec_layer + rec_first[lo], b->rec_layer_t.data() + rec_first[lo], rec_first[hi] - rec_first[lo]);
    }
  });
  for (unsigned t = 0; t < nt; t++) part[t + 1] += part[t];
  parallel([&](unsigned t, uint32_t lo, uint32_t hi) {
    uint64_t off = part[t];
    for (uint32_t s = lo; s < hi; s++) {
      uint64_t r = rec_first[s];
      for (uint32_t m = rec_mask(s), l = 0; m; m >>= 1, l++)
        if (m & 1) {
          off += count_of(s, l);
          rec_off[++r] = off;
        }
    }
  });
  uint32_t *nbrs = (uint32_t *)take_export_buf(b, 3, std::max<uint64_t>(rec_off[nrec], 1) * 4);
  if (!nbrs) return fail(HNY_ERR_OOM, "out of host memory for %llu links", (unsigned long long)rec_off[nrec]);
  g->neighbours = nbrs;
  const bool identity = !b->incremental && n && b->ids[n - 1] == n - 1; // ids 0..n-1: slot == item id
  std::atomic<int> landed_err{0};
  parallel([&](unsigned t, uint32_t lo, uint32_t hi) {
    // the upper-layer lists and this thread's piece of the layer-0 lists have arrived
    if (l0_landed[t] && (hipSetDevice(b->device) != hipSuccess || hipEventSynchronize(up_landed) != hipSuccess ||
                         hipEventSynchronize(l0_landed[t]) != hipSuccess))
      landed_err.store(1);
    for (uint32_t s = lo; s < hi; s++) {
      uint64_t r = rec_first[s];
      for (uint32_t m = rec_mask(s), l = 0; m; m >>= 1, l++) {
        if (!(m & 1)) continue;
        const u32 *src = l == 0 ? &l0[(size_t)s * M0]
                                : &up[((size_t)b->upper_idx[s] * upl + (l - 1)) * M];
        const uint32_t c = (uint32_t)(rec_off[r + 1] - rec_off[r]);
        uint32_t *dst = nbrs + rec_off[r];
        if (identity)
          memcpy(dst, src, (size_t)c * 4);
        else
          for (uint32_t k = 0; k < c; k++) dst[k] = b->ids[src[k]]; // slot -> item id (order kept)
        r++;
      }
    }
  });
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (landed_err.load()) return fail(HNY_ERR_NO_DEVICE, "export: waiting for the lists failed");
  uint32_t *eps = (uint32_t *)malloc(std::max<size_t>(b->entry_points.size(), 1) * 4);
  for (size_t i = 0; i < b->entry_points.size(); i++) eps[i] = b->ids[b->entry_points[i]];
  g->n_records = nrec;
  g->entry_points = eps;
  g->n_entry_points = (uint32_t)b->entry_points.size();
  g->max_level = ml;
#ifdef HNY_PHASE_CLOCKS
  fprintf(stderr, "[hny] walk wave cycles: pop %llu list+visited %llu distances %llu insert %llu | expansions %llu | whole kernel %llu\n",
          stats[ST_PH_POP], stats[ST_PH_LIST], stats[ST_PH_DIST], stats[ST_PH_INSERT], stats[ST_PH_EXPANSIONS],
          stats[ST_PH_REST]);
  fprintf(stderr, "[hny] short walk: visited wait %llu | lanes asked %llu accepted %llu expansions that accepted %llu pool scans %llu "
          "expansions with nothing new %llu\n", stats[ST_PH_VIS], stats[ST_PH_NASK], stats[ST_PH_NACC], stats[ST_PH_NMERGE],
          stats[ST_PH_NPOOL], stats[ST_PH_NONEW]);
#endif
  if (getenv("HNY_DEBUG_COUNTS"))
    fprintf(stderr, "[hny] expansions %llu accepted %llu notfull %llu evals_walk %llu\n", stats[9], stats[10], stats[11],
            stats[ST_EVALS_WALK]);
  g->n_links_added = stats[ST_LINKS];
  g->n_evals_walk = stats[ST_EVALS_WALK];
  g->n_evals_prune = stats[ST_EVALS_PRUNE];
  g->n_evals_apply = stats[ST_EVALS_APPLY];
  g->n_distance_evals = stats[ST_EVALS_WALK] + stats[ST_EVALS_PRUNE] + stats[ST_EVALS_APPLY];
  g->n_batches = b->n_batches;
  g->n_tie_pool_overflow = stats[ST_POOL_OVERFLOW];
  g->t_upload_s = b->t_upload;
  g->t_build_s = b->t_build;
  g->t_export_s = now_s() - t0;
  if (b->profiling) {
    double acc[EV_KINDS] = {0};
    uint64_t cnt[EV_KINDS] = {0};
    for (size_t i = 0; i < b->ev_used; i++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, b->evs[i].a, b->evs[i].b) == hipSuccess) {
        acc[b->evs[i].kind] += ms * 1e-3;
        cnt[b->evs[i].kind]++;
      }
    }
    g->t_walk_kernels_s = acc[EV_WALK];
    g->t_prune_kernels_s = acc[EV_PRUNE];
    g->t_sort_kernels_s = acc[EV_SORT];
    g->t_apply_kernels_s = acc[EV_APPLY];
    (void)cnt;
    g->n_walk_launches = b->n_walk_dispatch; // k_walk dispatches (rocprofv3 counts the same)
  }
  *out = gh.release();
  return HNY_OK;
}

extern "C" int hny_internal_build_multi(const hny_build_opts *opts, const hny_items *items, const uint32_t *to_insert,
                                        uint64_t n_insert, const uint32_t *to_delete, uint64_t n_delete,
                                        const hny_prev_graph *prev, hny_graph **out); // hny_multi.cpp
static bool wants_multi(const hny_build_opts *o) { return o && (o->n_gpus > 1 || (o->n_gpus == 1 && o->devices)); }

int hny_build(const hny_build_opts *opts, const hny_items *items, hny_graph **out) {
  if (!out) return fail(HNY_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (wants_multi(opts)) {
    if (!items) return fail(HNY_ERR_INVALID_ARG, "null argument");
    return hny_internal_build_multi(opts, items, nullptr, 0, nullptr, 0, nullptr, out);
  }
  hny_builder *b = nullptr;
  int rc = hny_builder_create(opts, items, &b);
  if (rc) return rc;
  for (;;) {
    // cancel: probed before every batch, i.e. every <= batch_max items (the reference probes every
    // CANCELLATION_PROBING = 10 000 items, lib.rs:140, hnsw.rs:174-177)
    if (opts->cancel && opts->cancel(opts->cancel_ctx)) {
      hny_builder_destroy(b);
      return fail(HNY_ERR_CANCELLED, "build cancelled");
    }
    hny_batch bt;
    rc = hny_builder_next_batch(b, &bt);
    if (rc || bt.count == 0) break;
    rc = hny_builder_search(b, 0, bt.count, nullptr);
    if (rc) break;
    rc = hny_builder_apply(b, nullptr);
    if (rc) break;
    if (opts->progress) opts->progress(opts->progress_ctx, b->pos, b->order.size());
  }
  if (!rc) rc = hny_builder_finish(b, out);
  hny_builder_destroy(b);
  return rc;
}

// fill_gaps_from_deleted (hnsw.rs:187, 334-415): merge old and new links of every surviving old
// record and bridge the holes deleted items leave
static int run_fill_gaps(hny_builder *b) {
  if (!b->incremental || b->old_recs.empty()) return HNY_OK;
  const u32 cap = std::max(b->g.M0, b->g.M);
  if (cap > HNY_MAX_CAP) {
    // wide lists: the workgroup kernel with its gathered set, scored list and bitmap in HBM
    const u32 n_recs = (u32)b->old_recs.size();
    const u32 capmax = (cap + 63u) / 64u * 64u;
    const u32 words = (b->g.n + 31u) / 32u + 1u;
    const u32 maxb = (u32)std::min<uint64_t>((uint64_t)b->g.n, (uint64_t)cap * (cap + 1u));
    const size_t per_block = (size_t)words * 4 + (size_t)maxb * 4 + 2 * ((size_t)maxb + capmax) * 8;
    if (!b->gap_grid) {
      const size_t budget = (size_t)2 << 30;
      int grid = (int)std::min<size_t>(std::max<size_t>(budget / per_block, 1), 1024);
      b->gap_grid = grid;
      HIP_TRY(b->d_gap_bitmap.alloc((size_t)grid * words));
      HIP_TRY(b->d_gap_bm.alloc((size_t)grid * std::max<u32>(maxb, 1)));
      HIP_TRY(b->d_gap_keys.alloc((size_t)grid * ((size_t)maxb + capmax)));
      HIP_TRY(b->d_gap_sorted.alloc((size_t)grid * ((size_t)maxb + capmax)));
      HIP_TRY(hipMemsetAsync(b->d_gap_bitmap.p, 0, (size_t)grid * words * 4, b->stream)); // the kernel leaves it zero
    }
    prof_begin(b, EV_APPLY);
    HIP_TRY(hnyk_fill_gaps_wg(b->g, b->d_old_recs.p, n_recs, b->d_deleted.p, b->d_gap_bitmap.p, words, b->d_gap_bm.p,
                              maxb, b->d_gap_keys.p, b->d_gap_sorted.p, b->stage_rows,
                              (int)std::min<u32>(n_recs, (u32)b->gap_grid), b->shape, b->stream));
    prof_end(b);
    return HNY_OK;
  }
  prof_begin(b, EV_APPLY);
  HIP_TRY(hnyk_fill_gaps(b->g, b->d_old_recs.p, (u32)b->old_recs.size(), b->d_deleted.p, b->shape,
                         b->stream));
  prof_end(b);
  return HNY_OK;
}

int hny_builder_create_incremental(const hny_build_opts *opts, const hny_items *items,
                                   const uint32_t *to_insert, uint64_t n_insert,
                                   const uint32_t *to_delete, uint64_t n_delete,
                                   const hny_prev_graph *prev, hny_builder **out) {
  IncrementalSpec inc{to_insert, n_insert, to_delete, n_delete, prev};
  return create_impl(opts, items, &inc, out);
}

int hny_builder_load(const hny_build_opts *opts, const hny_items *items, const hny_prev_graph *prev,
                     hny_builder **out) {
  if (!prev) return fail(HNY_ERR_INVALID_ARG, "null graph");
  IncrementalSpec inc{nullptr, 0, nullptr, 0, prev};
  inc.load_only = true;
  return create_impl(opts, items, &inc, out);
}

// hny_multi.cpp: the distance-evaluation counters of a replica.  Work that every rank repeats (ramp-up
// batches, small deferred sets, fill_gaps) is counted on rank 0 only: the other ranks point their kernels'
// counter block at a scratch copy meanwhile (the error words in it are the same on every replica).
extern "C" void hny_internal_builder_set_export(hny_builder *b, int on) { b->will_export = on != 0; }
extern "C" int hny_internal_builder_count_evals(hny_builder *b, int on) {
  if (!b) return fail(HNY_ERR_INVALID_ARG, "null builder");
  HIP_TRY(hipSetDevice(b->device));
  if (!on && !b->d_stats_scratch.p) {
    HIP_TRY(b->d_stats_scratch.alloc(ST_COUNT));
    HIP_TRY(hipMemsetAsync(b->d_stats_scratch.p, 0, ST_COUNT * 8, b->stream));
  }
  b->g.stats = on ? b->d_stats.p : b->d_stats_scratch.p;
  return HNY_OK;
}
extern "C" int hny_internal_builder_read_evals(hny_builder *b, uint64_t out3[3]) {
  if (!b || !out3) return fail(HNY_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  u64 stats[ST_COUNT] = {0};
  HIP_TRY(hipMemcpy(stats, b->d_stats.p, sizeof stats, hipMemcpyDeviceToHost));
  out3[0] = stats[ST_EVALS_WALK];
  out3[1] = stats[ST_EVALS_PRUNE];
  out3[2] = stats[ST_EVALS_APPLY];
  // the error words of THIS replica: an overflow inside the shard a rank >= 1 searched shows up nowhere
  // else (hny_multi.cpp folds the result into the ranks' agreement, so the whole build fails with it)
  return device_error_words(b, stats);
}

int hny_builder_fill_gaps(hny_builder *b) {
  if (!b) return fail(HNY_ERR_INVALID_ARG, "null builder");
  if (b->pos < b->order.size() || b->in_batch) return fail(HNY_ERR_INVALID_ARG, "build not finished");
  HIP_TRY(hipSetDevice(b->device));
  return run_fill_gaps(b);
}

int hny_build_incremental(const hny_build_opts *opts, const hny_items *items, const uint32_t *to_insert,
                          uint64_t n_insert, const uint32_t *to_delete, uint64_t n_delete,
                          const hny_prev_graph *prev, hny_graph **out) {
  if (!out) return fail(HNY_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (wants_multi(opts)) {
    if (!items || !prev) return fail(HNY_ERR_INVALID_ARG, "null argument");
    return hny_internal_build_multi(opts, items, to_insert, n_insert, to_delete, n_delete, prev, out);
  }
  IncrementalSpec inc{to_insert, n_insert, to_delete, n_delete, prev};
  hny_builder *b = nullptr;
  int rc = create_impl(opts, items, &inc, &b);
  if (rc) return rc;
  for (;;) {
    if (opts->cancel && opts->cancel(opts->cancel_ctx)) {
      hny_builder_destroy(b);
      return fail(HNY_ERR_CANCELLED, "build cancelled");
    }
    hny_batch bt;
    rc = hny_builder_next_batch(b, &bt);
    if (rc || bt.count == 0) break;
    rc = hny_builder_search(b, 0, bt.count, nullptr);
    if (rc) break;
    rc = hny_builder_apply(b, nullptr);
    if (rc) break;
    if (opts->progress) opts->progress(opts->progress_ctx, b->pos, b->order.size());
  }
  if (!rc) rc = run_fill_gaps(b);
  if (!rc) rc = hny_builder_finish(b, out);
  hny_builder_destroy(b);
  return rc;
}

int hny_builder_distances(hny_builder *b, uint64_t n_pairs, const uint32_t *slot_a,
                          const uint32_t *slot_b, float *out) {
  if (!b || !slot_a || !slot_b || !out) return fail(HNY_ERR_INVALID_ARG, "null argument");
  if (!n_pairs) return HNY_OK;
  for (uint64_t i = 0; i < n_pairs; i++)
    if (slot_a[i] >= b->n || slot_b[i] >= b->n) return fail(HNY_ERR_MISSING_KEY, "slot out of range");
  HIP_TRY(hipSetDevice(b->device));
  DevBuf<u32> da, db;
  DevBuf<float> dout;
  HIP_TRY(da.alloc(n_pairs));
  HIP_TRY(db.alloc(n_pairs));
  HIP_TRY(dout.alloc(n_pairs));
  HIP_TRY(hipMemcpyAsync(da.p, slot_a, n_pairs * 4, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(db.p, slot_b, n_pairs * 4, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hnyk_pair_distances(b->g, da.p, db.p, (u32)n_pairs, dout.p, b->shape, b->stream));
  HIP_TRY(hipMemcpyAsync(out, dout.p, n_pairs * 4, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return HNY_OK;
}

// The cancel closure of the *_with_cancellation searches (reader.rs:108-119, 167-186; probe :333): a
// pinned, device-visible word the kernels' work-queue loops poll, raised by the calling thread, which
// probes the closure while it waits for the stream.
struct SearchCancel {
  int (*fn)(void *) = nullptr;
  void *ctx = nullptr;
  u32 *h = nullptr, *d = nullptr;
  hipEvent_t ev = nullptr; // its own event: searches must not grow the builder's event pool
  bool cancelled = false;
  ~SearchCancel() {
    if (h) (void)hipHostFree(h);
    if (ev) (void)hipEventDestroy(ev);
  }
  hipError_t init(const hny_query_opts *qo) {
    if (!qo || !qo->cancel) return hipSuccess;
    fn = qo->cancel;
    ctx = qo->cancel_ctx;
    hipError_t e = hipHostMalloc((void **)&h, 64, hipHostMallocMapped);
    if (e != hipSuccess) return e;
    *h = 0u;
    e = hipHostGetDevicePointer((void **)&d, h, 0);
    if (e != hipSuccess) return e;
    return hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  }
  bool probe() { // before a chunk is started
    if (fn && !cancelled && fn(ctx)) {
      cancelled = true;
      __atomic_store_n(h, 1u, __ATOMIC_RELEASE);
    }
    return cancelled;
  }
  hipError_t wait(hny_builder *b) {
    if (!fn) return hipStreamSynchronize(b->stream);
    hipError_t e = hipEventRecord(ev, b->stream);
    if (e != hipSuccess) return e;
    for (;;) {
      e = hipEventQuery(ev);
      if (e != hipErrorNotReady) return e;
      (void)probe();
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
};

static int search_knn_impl(hny_builder *b, uint64_t nq, const void *qvectors, size_t qstride,
                           const void *qheaders, uint32_t k, uint32_t ef_search, uint32_t *out_ids,
                           float *out_dists, uint32_t *out_counts, const hny_query_opts *qo);
// the QueryBuilder searcher with its search queue as a real heap in HBM (k_nns_filtered); force_heap: also
// for queries without a candidates filter — where search_knn_impl sends the queries whose tie pool overflowed
static int nns_impl(hny_builder *b, const hny_query_opts *qo, uint64_t nq, const void *qvectors, size_t qstride,
                    const void *qheaders, const uint32_t *query_items, uint32_t *out_ids, float *out_dists,
                    uint32_t *out_counts, bool force_heap);

int hny_builder_search_knn(hny_builder *b, uint64_t nq, const void *qvectors, size_t qstride,
                           const void *qheaders, uint32_t k, uint32_t ef_search, uint32_t *out_ids,
                           float *out_dists, uint32_t *out_counts) {
  return search_knn_impl(b, nq, qvectors, qstride, qheaders, k, ef_search, out_ids, out_dists, out_counts, nullptr);
}

static int search_knn_impl(hny_builder *b, uint64_t nq, const void *qvectors, size_t qstride,
                           const void *qheaders, uint32_t k, uint32_t ef_search, uint32_t *out_ids,
                           float *out_dists, uint32_t *out_counts, const hny_query_opts *qo) {
  if (!b || !qvectors || !qheaders || !out_ids || !out_dists || !out_counts || k == 0)
    return fail(HNY_ERR_INVALID_ARG, "bad argument");
  SearchCancel sc;
  if (qo && qo->did_cancel) *qo->did_cancel = 0;
  HIP_TRY(sc.init(qo));
  if (b->pos < b->order.size()) return fail(HNY_ERR_INVALID_ARG, "build not finished");
  const uint32_t ef = std::max(ef_search, k); // reader.rs:746
  // result sets of up to 4 096 entries live in the walk's LDS, larger ones in HBM (WalkArgs.res_global, the
  // general kernel): the reference's own tests search with ef_search = n up to 9 999 (src/tests/reader.rs:82-98)
  if ((uint64_t)ef + 1 > HNY_RES_GLOBAL_MAX)
    return fail(HNY_ERR_UNSUPPORTED, "ef_search %u: result sets hold at most %u entries", ef, HNY_RES_GLOBAL_MAX - 1);
  HIP_TRY(hipSetDevice(b->device));
  if (b->n == 0) {
    for (uint64_t i = 0; i < nq; i++) out_counts[i] = 0; // reader.rs:652-654
    return HNY_OK;
  }
  if (!b->finalized) { // Reader::visit iterates Links bitmaps: ascending, deduplicated
    HIP_TRY(hnyk_finalize_lists(b->d_l0_ids.p, b->d_fin_cnt0.p, b->n, b->o.M0, b->stream));
    HIP_TRY(hnyk_finalize_lists(b->d_up_ids.p, b->d_fin_cntu.p,
                                (u32)((size_t)b->n_upper * b->up_layers), b->o.M, b->stream));
    b->finalized = true;
  }
  const uint32_t rcap = res_capacity(ef, (uint32_t)b->entry_points.size(), b->n, b->top_layer_nodes, HNY_RES_GLOBAL_MAX);
  const size_t vb = vec_bytes(b->o.metric, b->o.dim), hb = hdr_bytes(b->o.metric);
  if (qstride < vb) return fail(HNY_ERR_INVALID_DIM, "query stride too small");
  uint32_t chunk = std::max<uint32_t>(b->max_batch, 256);
  chunk = (uint32_t)std::min<uint64_t>(chunk, std::max<uint64_t>(nq, 1)); // buffers are sized chunk x rcap / k
  // at most ~2 GB of candidate lists per chunk
  chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(chunk, ((uint64_t)2 << 30) / ((uint64_t)std::max(rcap, k) * 8)));
  DevBuf<unsigned char> dq;
  DevBuf<float> dqn;
  DevBuf<u64> dcand, dres; // dres: result sets beyond the LDS (a search from more entry points than ef, res_capacity)
  DevBuf<u32> dcn;
  if (rcap > HNY_RES_LDS_MAX) {
    chunk = std::min<uint32_t>(chunk, 4096);
    HIP_TRY(dres.alloc((size_t)std::min<uint32_t>(chunk, b->walk_slots) * rcap));
  }
  HIP_TRY(dq.alloc((size_t)chunk * b->g.row_stride));
  HIP_TRY(dqn.alloc(chunk));
  HIP_TRY(dcand.alloc((size_t)chunk * rcap));
  HIP_TRY(dcn.alloc(chunk));
  const bool has_norm = b->g.norms != nullptr;
  std::vector<float> qn(chunk);
  DevBuf<u64> dtop;
  HIP_TRY(dtop.alloc((size_t)chunk * k));
  std::vector<u64> hc((size_t)chunk * k);
  std::vector<u32> hn(chunk);
  u32 *queues = b->d_nseg.p + 4;
  for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
    uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
    if (sc.probe()) { // cancelled: nothing of this chunk is started
      for (uint32_t i = 0; i < cnt; i++) out_counts[q0 + i] = 0u;
      continue;
    }
    int rc = upload_rows((const unsigned char *)qvectors + q0 * qstride, qstride, vb, cnt,
                         b->g.row_stride, dq.p, b->stream);
    if (rc) return rc;
    if (has_norm) {
      for (uint32_t i = 0; i < cnt; i++) memcpy(&qn[i], (const unsigned char *)qheaders + (q0 + i) * hb, 4);
      HIP_TRY(hipMemcpyAsync(dqn.p, qn.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, b->stream));
    }
    WalkArgs w{};
    w.q_rows = dq.p;
    w.q_norms = has_norm ? dqn.p : nullptr;
    w.q_stride = b->g.row_stride;
    w.lo = 0;
    w.hi = cnt;
    w.layer = 0;
    w.ef = ef;
    w.first = 1;
    w.reader_mode = 1;
    w.knn_k = k;
    w.knn_ef = ef_search;
    w.entry_points = b->d_eps.p;
    w.n_entry_points = (u32)b->entry_points.size();
    w.cand = dcand.p;
    w.cand_n = dcn.p;
    w.rcap = rcap;
    w.bits = b->d_bits.p;
    w.bits_words = b->bits_words;
    w.vlog = b->d_vlog.p;
    w.log_cap = b->log_cap;
    w.res_global = dres.p;
    w.vis_slots = vis_slots_for(b, dres.p ? 0u : w.rcap);
    w.eps_cap = eps_cap_of(b);
    w.queue = queues;
    w.cancel = sc.d;
    vis_buckets_for(b, w);
    w.pool_flag = env_int("HNY_NO_POOL_RETRY", 0) == 0 ? 1u : 0u;
    w.force_pool = (u32)std::max(0, env_int("HNY_POOL_FORCE_RETRY", 0)); // tests: every k-th query takes the retry path
    if (sc.d) HIP_TRY(hnyk_fill_u32(dcn.p, 0xFFFFFFFFu, cnt, b->stream)); // = never finished
    HIP_TRY(hipMemsetAsync(queues, 0, 8 * 4, b->stream));
    const int grid = (int)std::min<uint32_t>(cnt, b->walk_slots);
    if (b->locality && b->max_level >= 1 && cnt >= 2048) {
      // same locality ordering as the build: descent -> sort the queries by their region -> layer 0
      WalkArgs d = w;
      d.descend_only = 1;
      d.eps_out = b->d_eps0.p;
      d.key_out = b->d_lkey_a.p;
      b->n_walk_dispatch++;
    HIP_TRY(hnyk_walk(b->g, d, b->shape, grid, b->stream));
      HIP_TRY(hnyk_iota_u64(b->d_perm_a.p, 0, cnt, b->stream));
      size_t tmp = b->sort_tmp_bytes;
      HIP_TRY(hnyk_sort_pairs48(b->d_sort_tmp.p, tmp, b->d_lkey_a.p, b->d_lkey_b.p, b->d_perm_a.p,
                                b->d_perm_b.p, cnt, b->stream));
      w.first = 0;
      w.eps_in = b->d_eps0.p;
      w.perm = b->d_perm_b.p;
      w.queue = queues + 1;
      // the same XCD-tiled work queue as the build's level-0 walks (rows >= 1 KB, see run_batch)
      const u32 xcd_tile = (u32)std::max(0, env_int("HNY_XCD_TILE", b->g.row_stride >= 1024u ? 512 : 0));
      if (xcd_tile && cnt >= 16u * xcd_tile) {
        w.xcd_tile = xcd_tile;
        w.queue = queues + 16; // 8 counters, zeroed below
        HIP_TRY(hipMemsetAsync(queues + 16, 0, 8 * 4, b->stream));
      }
    }
    b->n_walk_dispatch++;
    HIP_TRY(hnyk_walk(b->g, w, b->shape, grid, b->stream));
    HIP_TRY(hnyk_take_topk(dcand.p, dcn.p, rcap, k, cnt, dtop.p, b->stream));
    HIP_TRY(hipMemcpyAsync(hc.data(), dtop.p, (size_t)cnt * k * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(hn.data(), dcn.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(sc.wait(b));
    std::vector<uint32_t> again; // queries whose tie pool overflowed (short codes, large ef_search: ties everywhere)
    for (uint32_t i = 0; i < cnt; i++) { // drain_asc().take(k), reader.rs:797-798
      if (sc.d && hn[i] == 0xFFFFFFFFu) { // the batch was cancelled before this query finished
        out_counts[q0 + i] = 0u;
        continue;
      }
      if (hn[i] == 0xFFFFFFFEu) {
        again.push_back(i);
        continue;
      }
      uint32_t c = std::min<uint32_t>(k, hn[i]);
      for (uint32_t j = 0; j < c; j++) {
        u64 e = hc[(size_t)i * k + j];
        out_ids[(q0 + i) * k + j] = b->ids[(uint32_t)(e & 0xFFFFFFFFull)];
        uint32_t db = (uint32_t)(e >> 32);
        memcpy(&out_dists[(q0 + i) * k + j], &db, 4);
      }
      out_counts[q0 + i] = c;
    }
    if (!again.empty() && !sc.cancelled && ef + 1 > HNY_RES_LDS_MAX) {
      // result sets beyond the LDS: the same queries once more with `candidates` AND `res` as heaps in HBM
      // (k_walk_heap in reader mode: the build's own safety net), on the chunk's buffers
      const u32 na = (u32)again.size();
      DevBuf<u32> dlist; // [0] = count, [1] = work counter, then the members
      DevBuf<u64> dheap_r;
      HIP_TRY(dlist.alloc((size_t)na + 2));
      std::vector<u32> hl((size_t)na + 2);
      hl[0] = na;
      hl[1] = 0u;
      std::copy(again.begin(), again.end(), hl.begin() + 2);
      HIP_TRY(hipMemcpyAsync(dlist.p, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, b->stream));
      const u32 hgrid = (u32)std::max<uint64_t>(
          1, std::min<uint64_t>(std::min<u32>(na, b->heap_grid), ((uint64_t)1 << 30) / (((uint64_t)rcap + 1) * 8)));
      HIP_TRY(dheap_r.alloc((size_t)hgrid * ((size_t)rcap + 1)));
      WalkArgs h = w;
      h.first = 1;
      h.eps_in = nullptr;
      h.perm = nullptr;
      h.xcd_tile = 0;
      h.descend_only = 0;
      h.pool_flag = 0;
      h.pool_retry = dlist.p + 2;
      h.n_pool_retry = dlist.p;
      h.queue = dlist.p + 1;
      h.heap_c = b->heap_c_ptr;
      h.heap_c_cap = b->heap_c_cap;
      h.heap_r = dheap_r.p;
      h.heap_r_cap = rcap + 1;
      if (sc.d) { // "never finished" for the members a cancellation leaves out
        for (u32 i : again) hn[i] = 0xFFFFFFFFu;
        HIP_TRY(hipMemcpyAsync(dcn.p, hn.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, b->stream));
      }
      b->n_walk_dispatch++;
      HIP_TRY(hnyk_walk_heap(b->g, h, b->shape, (int)hgrid, b->stream));
      HIP_TRY(hnyk_take_topk(dcand.p, dcn.p, rcap, k, cnt, dtop.p, b->stream));
      HIP_TRY(hipMemcpyAsync(hc.data(), dtop.p, (size_t)cnt * k * 8, hipMemcpyDeviceToHost, b->stream));
      HIP_TRY(hipMemcpyAsync(hn.data(), dcn.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, b->stream));
      HIP_TRY(sc.wait(b));
      for (uint32_t i : again) {
        if (hn[i] >= 0xFFFFFFFEu) { // cancelled before it ran (or failed: the error words say so below)
          out_counts[q0 + i] = 0u;
          continue;
        }
        const uint32_t c = std::min<uint32_t>(k, hn[i]);
        for (uint32_t j = 0; j < c; j++) {
          const u64 e = hc[(size_t)i * k + j];
          out_ids[(q0 + i) * k + j] = b->ids[(uint32_t)(e & 0xFFFFFFFFull)];
          const uint32_t db = (uint32_t)(e >> 32);
          memcpy(&out_dists[(q0 + i) * k + j], &db, 4);
        }
        out_counts[q0 + i] = c;
      }
    } else if (!again.empty() && !sc.cancelled) {
      // the same queries on the searcher whose queue is a real heap in HBM: nothing to overflow, same results
      const size_t na = again.size();
      std::vector<unsigned char> av(na * vb), ah(na * hb);
      std::vector<uint32_t> ai(na * k), ac(na);
      std::vector<float> ad(na * k);
      for (size_t j = 0; j < na; j++) {
        memcpy(&av[j * vb], (const unsigned char *)qvectors + (q0 + again[j]) * qstride, vb);
        memcpy(&ah[j * hb], (const unsigned char *)qheaders + (q0 + again[j]) * hb, hb);
      }
      hny_query_opts o2{};
      o2.k = k;
      o2.ef_search = ef_search;
      o2.linear_below = 1000;
      o2.linear_below_ratio = 1.0f;
      if (qo) {
        o2.cancel = qo->cancel;
        o2.cancel_ctx = qo->cancel_ctx;
      }
      int rc2 = nns_impl(b, &o2, na, av.data(), vb, ah.data(), nullptr, ai.data(), ad.data(), ac.data(), true);
      if (rc2) return rc2;
      for (size_t j = 0; j < na; j++) {
        const uint64_t qi = q0 + again[j];
        out_counts[qi] = ac[j];
        memcpy(&out_ids[qi * k], &ai[j * k], (size_t)k * 4);
        memcpy(&out_dists[qi * k], &ad[j * k], (size_t)k * 4);
      }
    } else {
      for (uint32_t i : again) out_counts[q0 + i] = 0u;
    }
  }
  if (sc.cancelled && qo && qo->did_cancel) *qo->did_cancel = 1;
  u64 stats[ST_COUNT] = {0};
  HIP_TRY(hipMemcpy(stats, b->d_stats.p, sizeof stats, hipMemcpyDeviceToHost));
  if (stats[ST_ERR_RES_OVERFLOW] || stats[ST_ERR_ITER]) {
    (void)clear_error_counters(b); // not sticky: the next search on this builder starts clean
    return fail(HNY_ERR_DEVICE, "kernel overflow: res=%llu iter=%llu", stats[ST_ERR_RES_OVERFLOW],
                stats[ST_ERR_ITER]);
  }
  if (stats[ST_POOL_OVERFLOW]) return pool_overflow_error(b, stats[ST_POOL_OVERFLOW]);
  return HNY_OK;
}

// QueryBuilder with .candidates() and/or by_item (reader.rs:60-262, 621-711, 809-896)
int hny_builder_nns(hny_builder *b, const hny_query_opts *qo, uint64_t nq, const void *qvectors,
                    size_t qstride, const void *qheaders, const uint32_t *query_items, uint32_t *out_ids,
                    float *out_dists, uint32_t *out_counts) {
  return nns_impl(b, qo, nq, qvectors, qstride, qheaders, query_items, out_ids, out_dists, out_counts, false);
}

static int nns_impl(hny_builder *b, const hny_query_opts *qo, uint64_t nq, const void *qvectors, size_t qstride,
                    const void *qheaders, const uint32_t *query_items, uint32_t *out_ids, float *out_dists,
                    uint32_t *out_counts, bool force_heap) {
  if (!b || !qo || !out_ids || !out_dists || !out_counts || qo->k == 0)
    return fail(HNY_ERR_INVALID_ARG, "bad argument");
  const bool by_item = query_items != nullptr;
  if (!by_item && (!qvectors || !qheaders)) return fail(HNY_ERR_INVALID_ARG, "no queries");
  if (qo->has_candidates && qo->n_candidates && !qo->candidates)
    return fail(HNY_ERR_INVALID_ARG, "candidates missing");
  if (!(qo->linear_below_ratio >= 0.f && qo->linear_below_ratio <= 1.f)) // reader.rs:253-256
    return fail(HNY_ERR_INVALID_ARG, "linear scan threshold ratio must be between 0.0 and 1.0");
  if (!qo->has_candidates && !by_item && !force_heap)
    return search_knn_impl(b, nq, qvectors, qstride, qheaders, qo->k, qo->ef_search, out_ids, out_dists,
                           out_counts, qo);
  if (b->pos < b->order.size()) return fail(HNY_ERR_INVALID_ARG, "build not finished");
  const uint32_t k = qo->k, ef = std::max(qo->ef_search, k); // reader.rs:746, 837
  if ((uint64_t)ef + 1 > HNY_RES_GLOBAL_MAX)
    return fail(HNY_ERR_UNSUPPORTED, "ef_search %u: result sets hold at most %u entries", ef, HNY_RES_GLOBAL_MAX - 1);
  // k_nns_filtered keeps its result set in LDS (up to 4 096 entries); beyond that the same search runs with
  // `res` as a heap in HBM next to the search queue's (k_nns_heap)
  // ... and so does a search that starts from more entry points than the LDS set holds (every entry point is pushed
  // to `res` without a capacity check, reader.rs:755-761: an all-level-0 index of 4 096 - 8 192 items)
  const uint64_t eps_need = std::max<uint64_t>(b->entry_points.size(), b->entry_points.size() > 1 ? b->top_layer_nodes : 0) + 1;
  const bool big = ef + 1 > HNY_RES_LDS_MAX || eps_need > HNY_RES_LDS_MAX;
  HIP_TRY(hipSetDevice(b->device));
  const uint32_t n = b->n;
  auto exists = [&](uint32_t s) { return !b->incremental || !b->deleted[s]; };
  auto slot_of = [&](uint32_t id) -> int64_t {
    auto it = std::lower_bound(b->ids.begin(), b->ids.end(), id);
    if (it == b->ids.end() || *it != id) return -1;
    uint32_t s = (uint32_t)(it - b->ids.begin());
    return exists(s) ? (int64_t)s : -1;
  };
  uint64_t n_items = 0;
  for (uint32_t s = 0; s < n; s++) n_items += exists(s) ? 1 : 0;
  // candidates ∩ item_ids as a mask over slots
  std::vector<u32> mask;
  std::vector<u32> cand_slots;
  if (qo->has_candidates) {
    mask.assign(((size_t)n + 31) / 32 + 1, 0u);
    for (uint64_t i = 0; i < qo->n_candidates; i++) {
      int64_t sl = slot_of(qo->candidates[i]);
      if (sl >= 0) mask[(size_t)sl >> 5] |= 1u << (sl & 31);
    }
    for (uint32_t s = 0; s < n; s++)
      if ((mask[s >> 5] >> (s & 31)) & 1u) cand_slots.push_back(s);
  }
  const uint32_t NONE = HNY_NNS_NONE;
  // reader.rs:652-654 / 822-824
  if (n_items == 0 || (qo->has_candidates && cand_slots.empty())) {
    for (uint64_t i = 0; i < nq; i++) out_counts[i] = by_item ? NONE : 0u;
    return HNY_OK;
  }
  // should_linear_scan, reader.rs:621-640
  bool linear = false;
  if (qo->has_candidates) {
    const uint64_t cl = cand_slots.size();
    linear = cl < (uint64_t)qo->linear_below && (float)cl / (float)n_items <= qo->linear_below_ratio;
  }
  if (!b->finalized) { // Reader::visit iterates Links bitmaps: ascending, deduplicated
    HIP_TRY(hnyk_finalize_lists(b->d_l0_ids.p, b->d_fin_cnt0.p, b->n, b->o.M0, b->stream));
    HIP_TRY(hnyk_finalize_lists(b->d_up_ids.p, b->d_fin_cntu.p, (u32)((size_t)b->n_upper * b->up_layers),
                                b->o.M, b->stream));
    b->finalized = true;
  }
  uint32_t rcap = res_capacity(ef, (uint32_t)b->entry_points.size(), b->n, b->top_layer_nodes,
                               big ? HNY_RES_GLOBAL_MAX : HNY_RES_LDS_MAX);
  if (big && linear) { // brute_force_search ranks in LDS: it returns min(k, candidates) hits
    const uint64_t need = std::min<uint64_t>(k, cand_slots.size()) + 1;
    if (need > HNY_RES_LDS_MAX)
      return fail(HNY_ERR_UNSUPPORTED, "linear scan for %u hits among %zu candidates: at most %u (lower linear_below)", k,
                  cand_slots.size(), HNY_RES_LDS_MAX - 1);
    rcap = 64;
    while (rcap < need) rcap *= 2;
  }
  const size_t vb = vec_bytes(b->o.metric, b->o.dim), hb = hdr_bytes(b->o.metric);
  if (!by_item && qstride < vb) return fail(HNY_ERR_INVALID_DIM, "query stride too small");
  const uint32_t chunk = (uint32_t)std::max<uint64_t>(
      1, std::min<uint64_t>(std::min<uint64_t>(std::max<uint32_t>(b->max_batch, 256), std::max<uint64_t>(nq, 1)),
                            ((uint64_t)2 << 30) / ((uint64_t)std::max(rcap, k) * 8))); // <= ~2 GB of candidate lists
  const bool has_norm = b->g.norms != nullptr;
  DevBuf<unsigned char> dq;
  DevBuf<float> dqn;
  DevBuf<u64> dcand, dtop, dheap;
  DevBuf<u32> dcn, dstatus, dqslots, dmembers, dfilter, dcslots;
  if (!by_item) {
    HIP_TRY(dq.alloc((size_t)chunk * b->g.row_stride));
    HIP_TRY(dqn.alloc(chunk));
  } else {
    HIP_TRY(dqslots.alloc(chunk));
  }
  HIP_TRY(dcand.alloc((size_t)chunk * rcap));
  HIP_TRY(dtop.alloc((size_t)chunk * k));
  HIP_TRY(dcn.alloc(chunk));
  HIP_TRY(dstatus.alloc(chunk));
  HIP_TRY(dmembers.alloc(chunk));
  if (qo->has_candidates) {
    HIP_TRY(dfilter.alloc(mask.size()));
    HIP_TRY(hipMemcpyAsync(dfilter.p, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, b->stream));
    if (linear) {
      HIP_TRY(dcslots.alloc(cand_slots.size()));
      HIP_TRY(hipMemcpyAsync(dcslots.p, cand_slots.data(), cand_slots.size() * 4, hipMemcpyHostToDevice,
                             b->stream));
    }
  }
  // search queue heaps: a modest one per resident wave first; queries that outgrow it run again
  // with room for every item (the queue never holds more than the visited set)
  const uint32_t heap_small = (uint32_t)std::min<uint64_t>((uint64_t)n + 1, 16384);
  const uint32_t heap_full = n + 1;
  uint32_t grid_small = std::min<uint32_t>(chunk, b->walk_slots);
  // big: queues with room for every item + the entry points and result heaps of rcap + 1, as many as 2 GB hold
  const uint32_t heap_big_c = (uint32_t)std::min<uint64_t>((uint64_t)n + 1 + eps_cap_of(b), 0xFFFFFFFFull);
  DevBuf<u64> dheap_r;
  uint32_t grid_big = 0;
  if (big && !linear) {
    grid_big = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>(grid_small, ((uint64_t)2 << 30) / (((uint64_t)heap_big_c + rcap + 1) * 8)));
    HIP_TRY(dheap.alloc((size_t)grid_big * heap_big_c));
    HIP_TRY(dheap_r.alloc((size_t)grid_big * ((size_t)rcap + 1)));
  } else if (!linear)
    HIP_TRY(dheap.alloc((size_t)grid_small * heap_small));
  DevBuf<u64> dheap_full;
  uint32_t grid_full = 0;
  std::vector<float> qn(chunk);
  std::vector<u32> qs(chunk), members(chunk), hn(chunk), hst(chunk);
  std::vector<u64> hc((size_t)chunk * k);
  u32 *queues = b->d_nseg.p + 4;
  SearchCancel sc; // hny_query_opts.cancel
  if (qo->did_cancel) *qo->did_cancel = 0;
  HIP_TRY(sc.init(qo));
  for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
    if (sc.probe()) { // nothing of this chunk is started: 0 hits each (unknown items stay None)
      for (uint32_t i = 0; i < cnt; i++)
        out_counts[q0 + i] = by_item && slot_of(query_items[q0 + i]) < 0 ? NONE : 0u;
      continue;
    }
    uint32_t n_mem = 0;
    if (by_item) {
      for (uint32_t i = 0; i < cnt; i++) {
        int64_t sl = slot_of(query_items[q0 + i]); // item_vector(..)? else Ok(None), reader.rs:826
        qs[i] = sl >= 0 ? (uint32_t)sl : 0u;
        hn[i] = 0;
        if (sl >= 0) members[n_mem++] = i;
        else out_counts[q0 + i] = NONE;
      }
      HIP_TRY(hipMemcpyAsync(dqslots.p, qs.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, b->stream));
    } else {
      int rc = upload_rows((const unsigned char *)qvectors + q0 * qstride, qstride, vb, cnt, b->g.row_stride,
                           dq.p, b->stream);
      if (rc) return rc;
      if (has_norm) {
        for (uint32_t i = 0; i < cnt; i++)
          memcpy(&qn[i], (const unsigned char *)qheaders + (q0 + i) * hb, 4);
        HIP_TRY(hipMemcpyAsync(dqn.p, qn.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, b->stream));
      }
      for (uint32_t i = 0; i < cnt; i++) members[n_mem++] = i;
    }
    if (n_mem == 0) continue;
    HIP_TRY(hipMemcpyAsync(dmembers.p, members.data(), (size_t)n_mem * 4, hipMemcpyHostToDevice, b->stream));
    NnsArgs a{};
    a.q_slots = dqslots.p;
    a.q_rows = dq.p;
    a.q_norms = has_norm && !by_item ? dqn.p : nullptr;
    a.q_stride = b->g.row_stride;
    a.members = dmembers.p;
    a.n_members = n_mem;
    a.filter = qo->has_candidates ? dfilter.p : nullptr;
    a.by_item = by_item ? 1 : 0;
    a.k = k;
    a.ef_main = ef;
    a.ef_opt = qo->ef_search;
    a.entry_points = b->d_eps.p;
    a.n_entry_points = (u32)b->entry_points.size();
    a.cand = dcand.p;
    a.cand_n = dcn.p;
    a.rcap = rcap;
    a.bits = b->d_bits.p;
    a.bits_words = b->bits_words;
    a.vlog = b->d_vlog.p;
    a.log_cap = b->log_cap;
    a.vis_slots = vis_slots_for(b, a.rcap);
    a.eps_cap = eps_cap_of(b);
    a.queue = queues;
    a.status = dstatus.p;
    a.cand_slots = dcslots.p;
    a.n_cand_slots = (u32)cand_slots.size();
    a.cancel = sc.d;
    if (sc.d) HIP_TRY(hnyk_fill_u32(dstatus.p, 2u, cnt, b->stream)); // 2 = never started
    HIP_TRY(hipMemsetAsync(queues, 0, 8 * 4, b->stream));
    if (linear) {
      HIP_TRY(hnyk_nns_linear(b->g, a, b->shape, (int)std::min<uint32_t>(n_mem, b->walk_slots), b->stream));
    } else if (big) {
      a.heap = dheap.p;
      a.heap_cap = heap_big_c;
      a.heap_r = dheap_r.p;
      a.heap_r_cap = rcap + 1;
      HIP_TRY(hnyk_nns_heap(b->g, a, b->shape, (int)std::min<uint32_t>(n_mem, grid_big), b->stream));
    } else {
      a.heap = dheap.p;
      a.heap_cap = heap_small;
      HIP_TRY(hnyk_nns_filtered(b->g, a, b->shape, (int)std::min<uint32_t>(n_mem, grid_small), b->stream));
      HIP_TRY(hipMemcpyAsync(hst.data(), dstatus.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, b->stream));
      HIP_TRY(sc.wait(b));
      uint32_t n_retry = 0;
      for (uint32_t j = 0; j < n_mem; j++)
        if (hst[members[j]] == 1u) members[n_retry++] = members[j];
      if (n_retry && heap_small < heap_full) {
        if (!grid_full) {
          grid_full = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b->walk_slots, (2ull << 30) / ((uint64_t)heap_full * 8)));
          HIP_TRY(dheap_full.alloc((size_t)grid_full * heap_full));
        }
        HIP_TRY(hipMemcpyAsync(dmembers.p, members.data(), (size_t)n_retry * 4, hipMemcpyHostToDevice, b->stream));
        a.n_members = n_retry;
        a.heap = dheap_full.p;
        a.heap_cap = heap_full;
        a.queue = queues + 1;
        HIP_TRY(hnyk_nns_filtered(b->g, a, b->shape, (int)std::min<uint32_t>(n_retry, grid_full), b->stream));
      } else if (n_retry) {
        return fail(HNY_ERR_DEVICE, "search queue overflow");
      }
    }
    HIP_TRY(hnyk_take_topk(dcand.p, dcn.p, rcap, k, cnt, dtop.p, b->stream));
    HIP_TRY(hipMemcpyAsync(hc.data(), dtop.p, (size_t)cnt * k * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(hn.data(), dcn.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(hst.data(), dstatus.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(sc.wait(b));
    for (uint32_t i = 0; i < cnt; i++) {
      if (by_item && out_counts[q0 + i] == NONE && slot_of(query_items[q0 + i]) < 0) continue;
      if (hst[i] == 2u || (sc.cancelled && hst[i] == 1u)) { // cancelled before this query (re)started
        out_counts[q0 + i] = 0u;
        continue;
      }
      if (hst[i]) return fail(HNY_ERR_DEVICE, "search queue overflow");
      uint32_t c = std::min<uint32_t>(k, hn[i]);
      for (uint32_t j = 0; j < c; j++) {
        u64 e = hc[(size_t)i * k + j];
        out_ids[(q0 + i) * k + j] = b->ids[(uint32_t)(e & 0xFFFFFFFFull)];
        uint32_t db = (uint32_t)(e >> 32);
        memcpy(&out_dists[(q0 + i) * k + j], &db, 4);
      }
      out_counts[q0 + i] = c;
    }
  }
  if (sc.cancelled && qo->did_cancel) *qo->did_cancel = 1;
  u64 stats[ST_COUNT] = {0};
  HIP_TRY(hipMemcpy(stats, b->d_stats.p, sizeof stats, hipMemcpyDeviceToHost));
  if (stats[ST_ERR_RES_OVERFLOW] || stats[ST_ERR_ITER]) {
    (void)clear_error_counters(b); // not sticky: the next search on this builder starts clean
    return fail(HNY_ERR_DEVICE, "kernel overflow: res=%llu iter=%llu", stats[ST_ERR_RES_OVERFLOW],
                stats[ST_ERR_ITER]);
  }
  if (stats[ST_POOL_OVERFLOW]) return pool_overflow_error(b, stats[ST_POOL_OVERFLOW]);
  return HNY_OK;
}

// ---------------------------------------------------------------------------------------------
// on-disk records
// ---------------------------------------------------------------------------------------------
static void put_key(uint16_t index, uint8_t mode, uint32_t item, uint8_t layer, uint8_t k[8]) {
  // key.rs:57-66: index u16 BE | mode u8 | item u32 BE | layer u8
  k[0] = (uint8_t)(index >> 8);
  k[1] = (uint8_t)index;
  k[2] = mode;
  k[3] = (uint8_t)(item >> 24);
  k[4] = (uint8_t)(item >> 16);
  k[5] = (uint8_t)(item >> 8);
  k[6] = (uint8_t)item;
  k[7] = layer;
}
// [3P] roaring 0.10.9 RoaringBitmap::serialize_into: portable format, cookie 12346 (no run
// containers), u16 key + u16 (card-1) per container, u32 offsets, array (<= 4096) or 8 KiB bitmap
static void roaring_append(std::vector<uint8_t> &out, const uint32_t *ids, uint64_t n) {
  struct Ct {
    uint16_t key;
    uint64_t b, e;
  };
  std::vector<Ct> cs;
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i;
    while (j < n && (ids[j] >> 16) == (ids[i] >> 16)) j++;
    cs.push_back({(uint16_t)(ids[i] >> 16), i, j});
    i = j;
  }
  auto w16 = [&](uint32_t v) {
    out.push_back((uint8_t)v);
    out.push_back((uint8_t)(v >> 8));
  };
  auto w32 = [&](uint32_t v) {
    for (int k = 0; k < 4; k++) out.push_back((uint8_t)(v >> (8 * k)));
  };
  w32(12346u);
  w32((uint32_t)cs.size());
  for (auto &c : cs) {
    w16(c.key);
    w16((uint32_t)(c.e - c.b - 1));
  }
  uint32_t off = (uint32_t)(8 + 8 * cs.size());
  for (auto &c : cs) {
    w32(off);
    off += (c.e - c.b) <= 4096 ? (uint32_t)(2 * (c.e - c.b)) : 8192u;
  }
  for (auto &c : cs) {
    if (c.e - c.b <= 4096) {
      for (uint64_t i = c.b; i < c.e; i++) w16(ids[i] & 0xFFFFu);
    } else {
      size_t at = out.size();
      out.resize(at + 8192, 0);
      for (uint64_t i = c.b; i < c.e; i++) {
        uint32_t lo = ids[i] & 0xFFFFu;
        out[at + (lo >> 3)] |= (uint8_t)(1u << (lo & 7));
      }
    }
  }
}
static const char *metric_name(int m) {
  static const char *names[] = {"cosine", "euclidean", "manhattan", "hamming",
                                "binary quantized cosine", "binary quantized euclidean",
                                "binary quantized manhattan"}; // cosine.rs:32-34 ...
  return names[m];
}

int hny_encode_kv(const hny_graph *g, const hny_build_opts *opts, const hny_items *items,
                  uint16_t index, int with_items, hny_kv_sink sink, void *ctx) {
  if (!g || !opts || !items || !sink) return fail(HNY_ERR_INVALID_ARG, "null argument");
  if (opts->metric < 0 || opts->metric > HNY_BQ_MANHATTAN) return fail(HNY_ERR_INVALID_ARG, "bad metric");
  uint8_t key[8];
  std::vector<uint8_t> val;
  auto emit = [&]() { return sink(ctx, key, 8, val.data(), val.size()); };
  // Metadata (metadata.rs:28-48)
  const char *nm = metric_name(opts->metric);
  val.assign(nm, nm + strlen(nm));
  val.push_back(0);
  for (int k = 3; k >= 0; k--) val.push_back((uint8_t)(opts->dim >> (8 * k)));
  std::vector<uint8_t> rb;
  roaring_append(rb, items->ids, items->n);
  for (int k = 3; k >= 0; k--) val.push_back((uint8_t)((uint32_t)rb.size() >> (8 * k)));
  val.insert(val.end(), rb.begin(), rb.end());
  for (uint32_t i = 0; i < g->n_entry_points; i++) { // ItemIds::raw_bytes: native-endian u32
    uint8_t e[4];
    memcpy(e, &g->entry_points[i], 4);
    val.insert(val.end(), e, e + 4);
  }
  val.push_back((uint8_t)g->max_level);
  put_key(index, 0, 0, 0, key);
  if (int rc = emit()) return rc;
  // Version (version.rs:36-48): crate version 0.1.3
  val.clear();
  for (uint32_t x : {0u, 1u, 3u})
    for (int k = 3; k >= 0; k--) val.push_back((uint8_t)(x >> (8 * k)));
  put_key(index, 0, 1, 0, key);
  if (int rc = emit()) return rc;
  // Links (node.rs:141-144)
  for (uint64_t r = 0; r < g->n_records; r++) {
    val.assign(1, 1);
    roaring_append(val, g->neighbours + g->rec_offset[r], g->rec_offset[r + 1] - g->rec_offset[r]);
    put_key(index, 2, g->rec_item[r], g->rec_layer[r], key);
    if (int rc = emit()) return rc;
  }
  // Items (node.rs:136-140)
  if (with_items) {
    const size_t vb = vec_bytes(opts->metric, opts->dim), hb = items->header_size;
    for (uint64_t s = 0; s < items->n; s++) {
      val.assign(1 + hb + vb, 0);
      memcpy(val.data() + 1, (const uint8_t *)items->headers + s * hb, hb);
      memcpy(val.data() + 1 + hb, (const uint8_t *)items->vectors + s * items->stride, vb);
      put_key(index, 3, items->ids[s], 0, key);
      if (int rc = emit()) return rc;
    }
  }
  return HNY_OK;
}

} // extern "C"
