// hny_multi.cpp — the multi-GPU host of hny_build: what replaces the rayon loop of
// /root/reference/src/hnsw.rs:172-185 (items of a level group inserted by a thread pool,
// src/parallel.rs:11-45 handing every thread the vectors) when the build owns several MI355X of one
// node (BASELINE north_star).
//
// One process, one host thread and one hny_builder per GPU; every GPU holds a full replica of the
// vectors and of the graph.  Per batch, rank r runs walk_layer + robust_prune for a contiguous
// slice of the members (hny_builder_search), the fixed-size selection records are exchanged with
// ncclAllGather (RCCL over xGMI) ENQUEUED ON THE BUILDER'S OWN STREAM — no host synchronisation
// between search, gather and the link phase — and every replica replays the same link ops in the
// same order (hny_builder_apply_begin), which keeps the replicas bit-identical.  The targets whose
// list overflows re-run robust_prune (hnsw.rs:547-552); they are split across the ranks too and
// their finished lists travel in a second, small all-gather (hny_builder_apply_deferred / _merge).
// The one host round trip per batch is the number of such targets (a 4-byte read).
//
// Written on top of the public stepwise C ABI only (include/hannoy_amd.h): a caller that wants its
// own collectives can do exactly this.  RCCL is loaded with dlopen at the first multi-GPU build (the
// library is 570 MB; a process that already holds an RCCL — PyTorch — shares that copy: same
// soname), so single-GPU users never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hannoy_amd.h"

int hny_internal_fail(int code, const char *msg); // hny_host.cpp: sets hny_last_error()

namespace {

int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return hny_internal_fail(code, buf);
}

// ---- RCCL entry points, resolved at run time
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr; // required: the only way out of a collective a failed peer never joins
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return HNY_OK;
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return failf(HNY_ERR_NO_DEVICE, "multi-GPU build: cannot load librccl (%s)", dlerror());
  Rccl r;
  r.handle = h;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GetErrorString || !r.CommAbort)
    return failf(HNY_ERR_NO_DEVICE, "multi-GPU build: librccl lacks an expected symbol");
  g_rccl = r;
  return HNY_OK;
}

// ---- rendezvous of the per-GPU host threads
class Barrier {
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t gen = gen_;
    if (++arrived_ == n_) {
      arrived_ = 0;
      gen_++;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen_ != gen; });
    }
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, arrived_ = 0;
  uint64_t gen_ = 0;
};

struct Shared {
  int world = 1;
  Barrier bar;
  std::vector<int> rc;             // one slot per rank, combined by agree()
  std::vector<std::string> msg;
  std::vector<void *> buf;         // shim exchange: every rank's buffer of the current collective
  std::vector<uint64_t> evals;     // [rank][walk, prune, apply]: summed into the graph rank 0 returns
  explicit Shared(int w) : world(w), bar(w), rc(w, 0), msg(w), buf(w, nullptr), evals((size_t)w * 3, 0) {}
  // every rank posts its status; all of them get the first error (or 0).  Called before each
  // collective so that no rank waits inside one for a rank that has already failed.
  int agree(int rank, int my_rc) {
    rc[rank] = my_rc;
    if (my_rc) msg[rank] = hny_last_error();
    bar.wait();
    int first = 0;
    for (int r = 0; r < world && !first; r++) first = rc[r];
    bar.wait();
    return first;
  }
};

// in-place all-gather of buf[rank * bytes .. (rank + 1) * bytes) on `st`
struct Exchange {
  virtual ~Exchange() {}
  virtual int all_gather(int rank, void *buf, size_t bytes, hipStream_t st) = 0;
  virtual void abort(int rank) { (void)rank; } // a peer failed between enqueue and completion
};

struct RcclExchange : Exchange {
  std::vector<ncclComm_t> comms;
  ~RcclExchange() override {
    for (ncclComm_t c : comms)
      if (c) (void)g_rccl.CommDestroy(c);
  }
  int init(const std::vector<int> &devices) {
    if (int rc = load_rccl()) return rc;
    comms.assign(devices.size(), nullptr);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devices.size(), devices.data());
    if (r != ncclSuccess) {
      comms.clear();
      return failf(HNY_ERR_NO_DEVICE, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
    }
    return HNY_OK;
  }
  void abort(int rank) override {
    if (g_rccl.CommAbort && comms[rank]) {
      (void)g_rccl.CommAbort(comms[rank]); // frees the communicator too
      comms[rank] = nullptr;
    }
  }
  int all_gather(int rank, void *buf, size_t bytes, hipStream_t st) override {
    if (!bytes) return HNY_OK;
    if (!comms[rank]) return failf(HNY_ERR_DEVICE, "ncclAllGather: communicator aborted");
    ncclResult_t r = g_rccl.AllGather((const char *)buf + (size_t)rank * bytes, buf, bytes, ncclUint8, comms[rank], st);
    if (r != ncclSuccess) return failf(HNY_ERR_DEVICE, "ncclAllGather: %s", g_rccl.GetErrorString(r));
    return HNY_OK;
  }
};

// Test shim (HNY_MGPU_SHIM=1): the same driver with the collective replaced by device-to-device
// copies between the builders' buffers and a host rendezvous, so that several "ranks" can share one
// GPU (RCCL refuses duplicate devices).  Exercises everything but RCCL itself.
struct ShimExchange : Exchange {
  Shared *sh;
  explicit ShimExchange(Shared *s) : sh(s) {}
  int all_gather(int rank, void *buf, size_t bytes, hipStream_t st) override {
    hipError_t e = hipStreamSynchronize(st); // my slice is complete
    sh->buf[rank] = buf;
    sh->bar.wait();
    for (int r = 0; r < sh->world && e == hipSuccess; r++)
      if (r != rank && bytes)
        e = hipMemcpyAsync((char *)buf + (size_t)r * bytes, (const char *)sh->buf[r] + (size_t)r * bytes, bytes,
                           hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    sh->bar.wait(); // nobody reuses a buffer a peer is still reading
    if (e != hipSuccess) return failf(HNY_ERR_DEVICE, "shim all-gather: %s", hipGetErrorString(e));
    return HNY_OK;
  }
};

struct DevMem {
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return HNY_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    need += need / 4;
    if (hipMalloc(&p, need) != hipSuccess) return failf(HNY_ERR_OOM, "multi-GPU build: exchange buffer of %zu bytes", need);
    bytes = need;
    return HNY_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  ~DevMem() { release(); }
};

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

struct Job {
  const hny_build_opts *opts;
  const hny_items *items;
  const uint32_t *to_insert, *to_delete;
  uint64_t n_insert, n_delete;
  const hny_prev_graph *prev; // null: fresh build
};

extern "C" int hny_internal_builder_count_evals(hny_builder *b, int on);
extern "C" void hny_internal_builder_set_export(hny_builder *b, int on); // only the exporting rank prepares arrays
extern "C" int hny_internal_builder_read_evals(hny_builder *b, uint64_t out3[3]);

} // namespace

// A resident multi-GPU build: one hny_builder (full replica of the vectors in HBM) per GPU, one
// communicator, the exchange buffers — created once, run any number of times (hny_multi_builder_run =
// graph reset + every batch + export, like hny_builder_reset / next_batch.. / finish on one GPU).
struct hny_multi_builder {
  int world = 1;
  std::vector<int> devices;
  hny_build_opts opts{};           // the caller's options (callbacks included), devices -> this->devices
  Shared sh;
  std::unique_ptr<Exchange> ex;
  std::vector<hny_builder *> b;    // one per rank
  std::vector<DevMem> sel, exch;   // per rank, allocated on the rank's device
  bool incremental = false, ran = false, profiling = false;
  uint64_t n_items = 0;
  uint64_t n_collectives = 0;      // of the last run (rank 0's count; every rank issues the same)
  explicit hny_multi_builder(int w) : world(w), sh(w), b(w, nullptr), sel(w), exch(w) {}
  ~hny_multi_builder() {
    for (int r = 0; r < world; r++) {
      (void)hipSetDevice(devices.empty() ? 0 : devices[r]);
      sel[r].release();
      exch[r].release();
      if (b[r]) hny_builder_destroy(b[r]);
    }
  }
};

namespace {

// rank r's part of hny_multi_builder_create: its replica
int create_rank(hny_multi_builder &mb, int rank, const Job &job) {
  hny_build_opts o = mb.opts;
  o.device = mb.devices[rank];
  o.n_gpus = 0;
  o.devices = nullptr;
  if (rank != 0) { // callbacks fire on rank 0 only
    o.cancel = nullptr;
    o.progress = nullptr;
  }
  hny_builder *b = nullptr;
  int rc = job.prev ? hny_builder_create_incremental(&o, job.items, job.to_insert, job.n_insert, job.to_delete,
                                                     job.n_delete, job.prev, &b)
                    : hny_builder_create(&o, job.items, &b);
  mb.b[rank] = b;
  return mb.sh.agree(rank, rc);
}

// the batch loop of one rank (hny_build's loop with the two exchanges in it)
int run_rank(hny_multi_builder &mb, int rank, hny_graph **out, uint64_t *checksum) {
  Shared &sh = mb.sh;
  Exchange &ex = *mb.ex;
  const int world = sh.world;
  const int device = mb.devices[rank];
  hny_builder *b = mb.b[rank];
  const hny_build_opts &opts = mb.opts;
  int rc = hipSetDevice(device) == hipSuccess ? HNY_OK : failf(HNY_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
  if (!rc && mb.ran) rc = hny_builder_reset(b); // a second run: empty graph again, vectors stay in HBM
  if (!rc) rc = hny_internal_builder_count_evals(b, 1);
  if (!rc) hny_internal_builder_set_export(b, rank == 0 || env_int("HNY_MGPU_VERIFY", 0) != 0);
  if (!rc) rc = hny_builder_set_profiling(b, mb.profiling ? 1 : 0);
  rc = sh.agree(rank, rc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)hny_builder_stream(b);
  DevMem &sel = mb.sel[rank], &exch = mb.exch[rank];
  const uint32_t min_shard = (uint32_t)std::max(1, env_int("HNY_MGPU_MIN_BATCH", 64 * world));
  const uint32_t min_def = (uint32_t)std::max(1, env_int("HNY_MGPU_MIN_DEFERRED", 32 * world));
  const uint32_t xs = hny_builder_exch_stride_u64(b);
  uint64_t done = 0, total = 0, n_coll = 0;
  // evaluation counters: what every rank repeats counts once (on rank 0), the shards add up
  auto replicated = [&](bool yes) -> int { return rank ? hny_internal_builder_count_evals(b, yes ? 0 : 1) : 0; };
  // a collective is enqueued by every rank or by none: the ranks agree before it, and once more after the
  // enqueue — a rank whose ncclAllGather failed must not leave its peers waiting inside theirs
  auto gather = [&](void *buf, size_t bytes) -> int {
    int r = ex.all_gather(rank, buf, bytes, st);
    n_coll++;
    r = sh.agree(rank, r);
    if (r) ex.abort(rank); // releases a collective this rank has already enqueued
    return r;
  };
  for (;;) {
    int cancelled = 0;
    if (opts.cancel) { // polled by rank 0 before every batch, obeyed by all (lib.rs:140)
      if (rank == 0) cancelled = opts.cancel(opts.cancel_ctx) ? HNY_ERR_CANCELLED : 0;
      if (rank == 0 && cancelled) (void)failf(HNY_ERR_CANCELLED, "build cancelled");
      cancelled = sh.agree(rank, cancelled);
      if (cancelled) return cancelled;
    }
    hny_batch bt;
    rc = hny_builder_next_batch(b, &bt);
    if (rc) return sh.agree(rank, rc);
    if (bt.count == 0) break;
    if (bt.count < min_shard) { // ramp-up: the exchange would cost more than the search
      rc = replicated(true);
      if (!rc) rc = hny_builder_search(b, 0, bt.count, nullptr);
      if (!rc) rc = hny_builder_apply(b, nullptr);
      if (rc) return sh.agree(rank, rc);
    } else {
      rc = replicated(false);
      if (rc) return sh.agree(rank, rc);
      const uint32_t per = (bt.count + (uint32_t)world - 1) / (uint32_t)world;
      const uint32_t lo = std::min<uint32_t>((uint32_t)rank * per, bt.count), hi = std::min<uint32_t>(lo + per, bt.count);
      const size_t words = (size_t)world * per * bt.sel_stride_u64;
      rc = sel.ensure(words * 8);
      if (!rc) rc = hny_builder_search(b, lo, hi, sel.p);
      rc = sh.agree(rank, rc);
      if (rc) return rc;
      rc = gather(sel.p, (size_t)per * bt.sel_stride_u64 * 8);
      if (rc) return rc;
      uint32_t nd = 0;
      rc = hny_builder_apply_begin(b, sel.p, &nd); // the same number on every rank
      if (!rc && nd < min_def) {
        rc = replicated(true);
        if (!rc) rc = hny_builder_apply_deferred(b, 0, 1, nullptr);
        if (!rc) rc = hny_builder_apply_merge(b, nullptr, 0, 1);
        if (rc) return sh.agree(rank, rc);
      } else {
        const uint32_t per2 = (nd + (uint32_t)world - 1) / (uint32_t)world;
        if (!rc) rc = exch.ensure((size_t)world * per2 * xs * 8);
        if (!rc) rc = hny_builder_apply_deferred(b, (uint32_t)rank, (uint32_t)world, exch.p);
        rc = sh.agree(rank, rc);
        if (rc) return rc;
        rc = gather(exch.p, (size_t)per2 * xs * 8);
        if (rc) return rc;
        rc = hny_builder_apply_merge(b, exch.p, (uint32_t)rank, (uint32_t)world);
        if (rc) return sh.agree(rank, rc);
      }
    }
    done += bt.count;
    if (rank == 0 && opts.progress) {
      if (!total) total = mb.incremental ? done : mb.n_items;
      opts.progress(opts.progress_ctx, done, std::max(total, done));
    }
  }
  if (mb.incremental) { // fill_gaps_from_deleted (hnsw.rs:187): deterministic, every replica does it
    rc = replicated(true);
    if (!rc) rc = hny_builder_fill_gaps(b);
    if (rc) return sh.agree(rank, rc);
  }
  // Every rank reads its own counters AND its own device error words here: an overflow of the tie pool or
  // of a result set inside the shard of a rank >= 1 exists only in that rank's counter block (rank 0 never
  // walked those members), and the clipped selection has already been all-gathered into every replica —
  // so it must fail the whole build, as it does on one GPU (hny_builder_finish).
  rc = replicated(false);
  if (!rc) rc = hny_internal_builder_read_evals(b, &sh.evals[(size_t)rank * 3]);
  rc = sh.agree(rank, rc);
  if (rc) return rc;
  if (rank == 0) mb.n_collectives = n_coll;
  const bool verify = env_int("HNY_MGPU_VERIFY", 0) != 0;
  if (rank == 0 || verify) {
    hny_graph *g = nullptr;
    rc = hny_builder_finish(b, &g);
    if (!rc && checksum) { // FNV-1a over the exported lists: replicas must agree
      uint64_t h = 1469598103934665603ull;
      const uint64_t nl = g->rec_offset[g->n_records];
      for (uint64_t i = 0; i < nl; i++) h = (h ^ g->neighbours[i]) * 1099511628211ull;
      for (uint64_t i = 0; i <= g->n_records; i++) h = (h ^ g->rec_offset[i]) * 1099511628211ull;
      *checksum = h;
    }
    if (rank == 0 && !rc) {
      for (int r = 1; r < world; r++) { // the other ranks' shards of the sharded work
        g->n_evals_walk += sh.evals[(size_t)r * 3 + 0];
        g->n_evals_prune += sh.evals[(size_t)r * 3 + 1];
        g->n_evals_apply += sh.evals[(size_t)r * 3 + 2];
        g->n_distance_evals += sh.evals[(size_t)r * 3 + 0] + sh.evals[(size_t)r * 3 + 1] + sh.evals[(size_t)r * 3 + 2];
      }
      *out = g;
    } else {
      hny_graph_free(g);
    }
  } else {
    rc = hny_builder_sync(b);
  }
  return rc;
}

// runs fn(rank) on one host thread per rank (rank 0 = the caller's thread: callbacks fire there) and
// reports the first failure with the message of the rank that failed first
template <class F>
int on_all_ranks(hny_multi_builder &mb, F &&fn) {
  const int world = mb.world;
  std::vector<int> rcs(world, 0);
  std::vector<std::string> msgs(world);
  for (auto &m : mb.sh.msg) m.clear();
  std::fill(mb.sh.rc.begin(), mb.sh.rc.end(), 0);
  std::vector<std::thread> th;
  for (int r = 1; r < world; r++)
    th.emplace_back([&, r]() {
      rcs[r] = fn(r);
      if (rcs[r]) msgs[r] = hny_last_error();
    });
  rcs[0] = fn(0);
  if (rcs[0]) msgs[0] = hny_last_error();
  for (auto &t : th) t.join();
  for (int r = 0; r < world; r++)
    if (rcs[r]) {
      // the rank that failed first holds the message (the others report "a peer failed")
      for (int q = 0; q < world; q++)
        if (!mb.sh.msg[q].empty()) return hny_internal_fail(mb.sh.rc[q] ? mb.sh.rc[q] : rcs[r], mb.sh.msg[q].c_str());
      return hny_internal_fail(rcs[r], msgs[r].empty() ? "multi-GPU build failed" : msgs[r].c_str());
    }
  return HNY_OK;
}

int create_multi(const Job &job, hny_multi_builder **out) {
  *out = nullptr;
  const hny_build_opts *opts = job.opts;
  const int world = opts->n_gpus;
  if (world < 1 || world > 64) return failf(HNY_ERR_INVALID_ARG, "n_gpus %d outside [1, 64]", world);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return failf(HNY_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
  std::unique_ptr<hny_multi_builder> mb(new hny_multi_builder(world));
  mb->devices.resize(world);
  for (int r = 0; r < world; r++) {
    mb->devices[r] = opts->devices ? opts->devices[r] : r;
    if (mb->devices[r] < 0 || mb->devices[r] >= ndev)
      return failf(HNY_ERR_NO_DEVICE, "device %d of rank %d: %d devices visible", mb->devices[r], r, ndev);
  }
  const bool shim = env_int("HNY_MGPU_SHIM", 0) != 0;
  if (!shim)
    for (int r = 0; r < world; r++)
      for (int q = 0; q < r; q++)
        if (mb->devices[q] == mb->devices[r]) return failf(HNY_ERR_INVALID_ARG, "device %d listed twice", mb->devices[r]);
  if (shim) {
    mb->ex.reset(new ShimExchange(&mb->sh));
  } else {
    std::unique_ptr<RcclExchange> rx(new RcclExchange());
    if (int rc = rx->init(mb->devices)) return rc;
    mb->ex = std::move(rx);
  }
  mb->opts = *opts;
  mb->opts.devices = mb->devices.data();
  mb->incremental = job.prev != nullptr;
  mb->n_items = job.items->n;
  hny_multi_builder &m = *mb;
  if (int rc = on_all_ranks(m, [&](int r) { return create_rank(m, r, job); })) return rc; // uploads run in parallel
  *out = mb.release();
  return HNY_OK;
}

int run_multi(hny_multi_builder *mb, hny_graph **out) {
  *out = nullptr;
  std::vector<uint64_t> sums(mb->world, 0);
  int rc = on_all_ranks(*mb, [&](int r) { return run_rank(*mb, r, out, &sums[r]); });
  mb->ran = true;
  if (rc) {
    if (*out) hny_graph_free(*out);
    *out = nullptr;
    return rc;
  }
  if (env_int("HNY_MGPU_VERIFY", 0))
    for (int r = 1; r < mb->world; r++)
      if (sums[r] != sums[0]) {
        hny_graph_free(*out);
        *out = nullptr;
        return failf(HNY_ERR_DEVICE, "replicas diverged: rank %d exports a different graph than rank 0", r);
      }
  return HNY_OK;
}

} // namespace

extern "C" int hny_internal_build_multi(const hny_build_opts *opts, const hny_items *items, const uint32_t *to_insert,
                                        uint64_t n_insert, const uint32_t *to_delete, uint64_t n_delete,
                                        const hny_prev_graph *prev, hny_graph **out) {
  *out = nullptr;
  Job job{opts, items, to_insert, to_delete, n_insert, n_delete, prev};
  hny_multi_builder *mb = nullptr;
  if (int rc = create_multi(job, &mb)) return rc;
  std::unique_ptr<hny_multi_builder> guard(mb);
  return run_multi(mb, out);
}

// ---- the resident form behind the C ABI (include/hannoy_amd.h)
extern "C" int hny_multi_builder_create(const hny_build_opts *opts, const hny_items *items, hny_multi_builder **out) {
  if (!out) return failf(HNY_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  if (!opts || !items) return failf(HNY_ERR_INVALID_ARG, "null argument");
  hny_build_opts o = *opts;
  if (o.n_gpus < 1) o.n_gpus = 1;
  Job job{&o, items, nullptr, nullptr, 0, 0, nullptr};
  return create_multi(job, out);
}
extern "C" int hny_multi_builder_run(hny_multi_builder *mb, hny_graph **out) {
  if (!mb || !out) return failf(HNY_ERR_INVALID_ARG, "null argument");
  return run_multi(mb, out);
}
extern "C" int hny_multi_builder_set_profiling(hny_multi_builder *mb, int on) {
  if (!mb) return failf(HNY_ERR_INVALID_ARG, "null builder");
  mb->profiling = on != 0;
  return HNY_OK;
}
extern "C" uint32_t hny_multi_builder_world(const hny_multi_builder *mb) { return mb ? (uint32_t)mb->world : 0u; }
extern "C" uint64_t hny_multi_builder_collectives(const hny_multi_builder *mb) { return mb ? mb->n_collectives : 0u; }
extern "C" hny_builder *hny_multi_builder_replica(hny_multi_builder *mb, uint32_t rank) {
  return mb && rank < (uint32_t)mb->world ? mb->b[rank] : nullptr;
}
extern "C" void hny_multi_builder_destroy(hny_multi_builder *mb) { delete mb; }
