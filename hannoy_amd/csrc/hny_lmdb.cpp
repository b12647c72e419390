// hny_lmdb.cpp — LMDB writeback of the index: a bulk-load writer and a reader
// for the LMDB environment file (`data.mdb`) that the reference fills through heed 0.22 / LMDB
// 0.9 `db.put` (/root/reference/src/hnsw.rs:195-213, src/writer.rs:462-480, 585-600) and opens
// in Reader::open (src/reader.rs:387-431).  LMDB itself is a third-party dependency that is not
// under /root/reference and not in this image, so the on-disk format of its mdb.c is restated
// here from the published layout (64-bit little-endian host, MDB_DATA_VERSION 1):
//
//   page header (16 B): pgno u64 | pad u16 | flags u16 | lower u16, upper u16  (overflow: pages u32)
//   flags: P_BRANCH 1, P_LEAF 2, P_OVERFLOW 4, P_META 8
//   node (8 B + key + data): lo u16 | hi u16 | flags u16 | ksize u16 | key | data
//     leaf:   lo|hi<<16 = data size; F_BIGDATA (1): data = u64 pgno of an overflow run holding it;
//             F_SUBDATA (2): data = an MDB_db (a named database inside the main DB)
//     branch: lo | hi<<16 | flags<<32 = child pgno; node 0 of a page carries no key
//   slots: u16 offsets from `lower` upwards, nodes from `upper` downwards, every node 2-aligned
//   meta (pages 0 and 1, after the header): magic 0xBEEFC0DE u32 | version u32 | address u64 |
//     mapsize u64 | MDB_db free | MDB_db main | last_pgno u64 | txnid u64; the valid meta with
//     the larger txnid wins; free.pad holds the page size, free.flags MDB_INTEGERKEY (8)
//   MDB_db (48 B): pad u32 | flags u16 | depth u16 | branch_pages | leaf_pages | overflow_pages |
//     entries | root  (u64 each; root = ~0 when empty)
//   a value goes to an overflow run when 8 + ksize + dsize > nodemax = (((psize-16)/2) & ~1) - 2
//
// Parity unpinned: no LMDB or hannoy binary exists here to open the result with; the
// tests pin the writer against an independent parser of the layout above and hand-assembled
// bytes.  Pure host code: no GPU work in this file.
#include "../../include/hannoy_amd.h"

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

int hny_internal_fail(int code, const char *msg);

namespace {

enum { P_BRANCH = 1, P_LEAF = 2, P_OVERFLOW = 4, P_META = 8 };
enum { F_BIGDATA = 1, F_SUBDATA = 2, F_DUPDATA = 4 };
constexpr uint32_t MAGIC = 0xBEEFC0DEu, DATA_VERSION = 1;
constexpr size_t HDR = 16, NODE = 8, DBSZ = 48, MAXKEY = 511;
constexpr uint64_t P_INVALID = ~0ull;

struct Db {
  uint32_t pad = 0;
  uint16_t flags = 0, depth = 0;
  uint64_t branch = 0, leaf = 0, overflow = 0, entries = 0, root = P_INVALID;
};
void put16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
void put32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
void put64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
uint16_t get16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
uint32_t get32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t get64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
void db_store(uint8_t *p, const Db &d) {
  put32(p, d.pad), put16(p + 4, d.flags), put16(p + 6, d.depth), put64(p + 8, d.branch);
  put64(p + 16, d.leaf), put64(p + 24, d.overflow), put64(p + 32, d.entries), put64(p + 40, d.root);
}
Db db_load(const uint8_t *p) {
  Db d;
  d.pad = get32(p), d.flags = get16(p + 4), d.depth = get16(p + 6), d.branch = get64(p + 8);
  d.leaf = get64(p + 16), d.overflow = get64(p + 24), d.entries = get64(p + 32), d.root = get64(p + 40);
  return d;
}
size_t even(size_t x) { return (x + 1) & ~(size_t)1; }
// mdb_cmp_memn: bytewise, the shorter key first on a common prefix
int keycmp(const uint8_t *a, size_t al, const uint8_t *b, size_t bl) {
  int c = memcmp(a, b, std::min(al, bl));
  return c ? c : (al < bl ? -1 : al > bl ? 1 : 0);
}
int failf(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return hny_internal_fail(code, buf);
}

// one page under construction: slots grow up from `lower`, nodes grow down from `upper`
struct PageBuf {
  std::vector<uint8_t> b;
  uint16_t lower, upper;
  void start(uint32_t psize, uint16_t flags) {
    b.assign(psize, 0);
    put16(b.data() + 10, flags);
    lower = HDR, upper = (uint16_t)psize;  // psize <= 32768
  }
  size_t room() const { return (size_t)upper - lower; }
  unsigned nkeys() const { return (lower - HDR) / 2; }
  uint8_t *add(size_t node_size) {  // mdb_node_add: carve the node, append its slot
    upper = (uint16_t)(upper - node_size);
    put16(b.data() + lower, upper);
    lower += 2;
    return b.data() + upper;
  }
  void seal() { put16(b.data() + 12, lower), put16(b.data() + 14, upper); }
};

} // namespace

struct hny_lmdb_writer {
  int fd = -1;
  uint32_t psize = 4096;
  uint64_t map_size = 0;
  bool named = false;
  std::string name;
  size_t nodemax = 0;
  uint64_t next_pgno = 2;
  std::vector<uint8_t> out;  // pages not yet written, strictly sequential in the file
  bool io_error = false;
  PageBuf leaf;
  bool leaf_open = false;
  std::string leaf_first, last_key;
  bool have_last = false;
  std::vector<std::pair<std::string, uint64_t>> level;  // (first key, pgno) of the finished pages
  Db db;

  void flush() {
    size_t off = 0;
    while (off < out.size()) {
      ssize_t w = write(fd, out.data() + off, out.size() - off);
      if (w < 0) {
        if (errno == EINTR) continue;
        io_error = true;
        break;
      }
      off += (size_t)w;
    }
    out.clear();
  }
  // appends `npages` pages (first one starts with a page header) and returns their first pgno
  uint64_t append(const uint8_t *bytes, size_t len, size_t npages) {
    uint64_t pg = next_pgno;
    next_pgno += npages;
    size_t at = out.size();
    out.resize(at + npages * (size_t)psize, 0);
    memcpy(out.data() + at, bytes, len);
    put64(out.data() + at, pg);
    if (out.size() >= (8u << 20)) flush();
    return pg;
  }
  void finish_leaf() {
    if (!leaf_open) return;
    leaf.seal();
    uint64_t pg = append(leaf.b.data(), psize, 1);
    level.emplace_back(leaf_first, pg);
    db.leaf++;
    leaf_open = false;
  }
};

extern "C" {

int hny_lmdb_writer_open(const char *path, uint32_t page_size, uint64_t map_size, const char *db_name,
                         hny_lmdb_writer **out) {
  if (!path || !out) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_writer_open: null argument");
  if (page_size == 0) page_size = 4096;
  if (page_size < 512 || page_size > 32768 || (page_size & (page_size - 1)))
    return failf(HNY_ERR_INVALID_ARG, "page size %u: LMDB pages are powers of two in 512..32768", page_size);
  if (db_name && (!*db_name || strlen(db_name) > MAXKEY))
    return failf(HNY_ERR_INVALID_ARG, "database name must be 1..511 bytes");
  int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0664);
  if (fd < 0) return failf(HNY_ERR_IO, "open %s: %s", path, strerror(errno));
  if (lseek(fd, 2 * (off_t)page_size, SEEK_SET) < 0) {
    close(fd);
    return failf(HNY_ERR_IO, "lseek %s: %s", path, strerror(errno));
  }
  auto *w = new hny_lmdb_writer;
  w->fd = fd, w->psize = page_size, w->map_size = map_size;
  w->named = db_name != nullptr;
  if (db_name) w->name = db_name;
  w->nodemax = (((page_size - HDR) / 2) & ~(size_t)1) - 2;  // me_nodemax, MDB_MINKEYS = 2
  *out = w;
  return HNY_OK;
}

int hny_lmdb_writer_put(void *ctx, const uint8_t *key, size_t klen, const uint8_t *val, size_t vlen) {
  auto *w = (hny_lmdb_writer *)ctx;
  if (!w || !key || (!val && vlen)) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_writer_put: null argument");
  // me_maxkey: 511, less on small pages (a key must fit a node next to an MDB_db)
  const size_t maxkey = std::min(MAXKEY, w->nodemax - (NODE + DBSZ));
  if (klen == 0 || klen > maxkey) return failf(HNY_ERR_INVALID_ARG, "key of %zu bytes (MDB_BAD_VALSIZE: 1..%zu)", klen, maxkey);
  if (vlen > 0xFFFFFFFFull) return failf(HNY_ERR_INVALID_ARG, "value of %zu bytes (MDB_BAD_VALSIZE)", vlen);
  if (w->have_last && keycmp((const uint8_t *)w->last_key.data(), w->last_key.size(), key, klen) >= 0)
    return failf(HNY_ERR_INVALID_ARG, "keys must arrive in strictly ascending order (MDB_KEYEXIST with MDB_APPEND)");
  w->last_key.assign((const char *)key, klen), w->have_last = true;
  const bool big = NODE + klen + vlen > w->nodemax;
  const size_t nsz = even(NODE + klen + (big ? 8 : vlen));
  if (w->leaf_open && w->leaf.room() < nsz + 2) w->finish_leaf();
  uint64_t ovpg = 0;
  if (big) {  // OVPAGES(size, psize) pages: header with the page count, then the bare value
    size_t np = (HDR - 1 + vlen) / w->psize + 1;
    uint8_t hdr[HDR] = {0};
    put16(hdr + 10, P_OVERFLOW), put32(hdr + 12, (uint32_t)np);
    uint64_t pg = w->next_pgno;
    w->next_pgno += np;
    size_t at = w->out.size();
    w->out.resize(at + np * (size_t)w->psize, 0);
    memcpy(w->out.data() + at, hdr, HDR);
    put64(w->out.data() + at, pg);
    memcpy(w->out.data() + at + HDR, val, vlen);
    if (w->out.size() >= (8u << 20)) w->flush();
    w->db.overflow += np;
    ovpg = pg;
  }
  if (!w->leaf_open) {
    w->leaf.start(w->psize, P_LEAF);
    w->leaf_first.assign((const char *)key, klen);
    w->leaf_open = true;
  }
  uint8_t *n = w->leaf.add(nsz);
  put16(n, (uint16_t)(vlen & 0xFFFF)), put16(n + 2, (uint16_t)(vlen >> 16));
  put16(n + 4, big ? F_BIGDATA : 0), put16(n + 6, (uint16_t)klen);
  memcpy(n + NODE, key, klen);
  if (big) put64(n + NODE + klen, ovpg);
  else if (vlen) memcpy(n + NODE + klen, val, vlen);
  w->db.entries++;
  return w->io_error ? failf(HNY_ERR_IO, "write failed: %s", strerror(errno)) : HNY_OK;
}

void hny_lmdb_writer_abort(hny_lmdb_writer *w) {
  if (!w) return;
  if (w->fd >= 0) close(w->fd);
  delete w;
}

int hny_lmdb_writer_finish(hny_lmdb_writer *w) {
  if (!w) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_writer_finish: null writer");
  w->finish_leaf();
  // branch levels, bottom up, until one page is left (the root)
  uint16_t depth = w->level.empty() ? 0 : 1;
  while (w->level.size() > 1) {
    auto &ents = w->level;
    // greedy split into pages; node 0 of a page has no key
    std::vector<size_t> starts{0};
    size_t room = w->psize - HDR;
    for (size_t i = 0; i < ents.size(); i++) {
      bool first = (i == starts.back());
      size_t need = even(NODE + (first ? 0 : ents[i].first.size())) + 2;
      if (need > room) {
        starts.push_back(i);
        room = w->psize - HDR;
        need = even(NODE) + 2;
      }
      room -= need;
    }
    // a branch page of a user DB must hold at least 2 keys (mdb_page_search_root asserts it)
    if (starts.size() > 1 && ents.size() - starts.back() < 2) starts.back()--;
    std::vector<std::pair<std::string, uint64_t>> up;
    for (size_t p = 0; p < starts.size(); p++) {
      size_t lo = starts[p], hi = p + 1 < starts.size() ? starts[p + 1] : ents.size();
      PageBuf pb;
      pb.start(w->psize, P_BRANCH);
      for (size_t i = lo; i < hi; i++) {
        size_t kl = i == lo ? 0 : ents[i].first.size();
        uint8_t *n = pb.add(even(NODE + kl));
        uint64_t pg = ents[i].second;
        put16(n, (uint16_t)(pg & 0xFFFF)), put16(n + 2, (uint16_t)((pg >> 16) & 0xFFFF));
        put16(n + 4, (uint16_t)(pg >> 32)), put16(n + 6, (uint16_t)kl);
        memcpy(n + NODE, ents[i].first.data(), kl);
      }
      pb.seal();
      up.emplace_back(ents[lo].first, w->append(pb.b.data(), w->psize, 1));
      w->db.branch++;
    }
    w->level.swap(up);
    depth++;
  }
  w->db.depth = depth;
  w->db.root = w->level.empty() ? P_INVALID : w->level[0].second;
  Db main_db = w->db;
  if (w->named) {  // the records live in a sub-DB: the main DB holds one F_SUBDATA node for it
    PageBuf pb;
    pb.start(w->psize, P_LEAF);
    uint8_t *n = pb.add(even(NODE + w->name.size() + DBSZ));
    put16(n, DBSZ), put16(n + 2, 0), put16(n + 4, F_SUBDATA), put16(n + 6, (uint16_t)w->name.size());
    memcpy(n + NODE, w->name.data(), w->name.size());
    db_store(n + NODE + w->name.size(), w->db);
    pb.seal();
    main_db = Db();
    main_db.root = w->append(pb.b.data(), w->psize, 1);
    main_db.depth = 1, main_db.leaf = 1, main_db.entries = 1;
  }
  w->flush();
  // meta pages: 0 = the state mdb_env_init_meta leaves (txnid 0), 1 = after this commit (txnid 1)
  const uint64_t last = w->next_pgno - 1;
  uint64_t mapsize = std::max<uint64_t>(w->map_size, (last + 1) * (uint64_t)w->psize);
  std::vector<uint8_t> metas(2 * (size_t)w->psize, 0);
  for (int m = 0; m < 2; m++) {
    uint8_t *p = metas.data() + (size_t)m * w->psize;
    put64(p, (uint64_t)m), put16(p + 10, P_META);
    uint8_t *q = p + HDR;
    put32(q, MAGIC), put32(q + 4, DATA_VERSION), put64(q + 8, 0), put64(q + 16, mapsize);
    Db freedb;
    freedb.pad = w->psize, freedb.flags = 0x08;  // mm_psize, MDB_INTEGERKEY
    db_store(q + 24, freedb);
    db_store(q + 24 + DBSZ, m ? main_db : Db());
    put64(q + 24 + 2 * DBSZ, m ? last : 1), put64(q + 32 + 2 * DBSZ, (uint64_t)m);
  }
  bool ok = !w->io_error && pwrite(w->fd, metas.data(), metas.size(), 0) == (ssize_t)metas.size();
  if (ok && last == 1) ok = ftruncate(w->fd, 2 * (off_t)w->psize) == 0;
  if (ok) ok = fsync(w->fd) == 0;
  int e = errno;
  close(w->fd);
  delete w;
  return ok ? HNY_OK : failf(HNY_ERR_IO, "writing data.mdb failed: %s", strerror(e));
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// read side: mdb_env_open (pick the newer valid meta) + mdb_get / cursor walk on the mapped file
// ------------------------------------------------------------------------------------------------
struct hny_lmdb_env {
  int fd = -1;
  const uint8_t *map = nullptr;
  size_t len = 0;
  uint32_t psize = 0;
  uint64_t last_pgno = 0, txnid = 0, map_size = 0;
  Db db;
};

namespace {

struct PageView {
  const uint8_t *p;
  uint16_t flags, lower, upper;
  unsigned nkeys;
};
int page_at(const hny_lmdb_env *e, uint64_t pg, PageView *v) {
  if (pg < 2 || pg > e->last_pgno || (pg + 1) * (uint64_t)e->psize > e->len)
    return failf(HNY_ERR_IO, "corrupt data.mdb: page %llu outside the file (last %llu)",
                 (unsigned long long)pg, (unsigned long long)e->last_pgno);
  v->p = e->map + pg * (uint64_t)e->psize;
  if (get64(v->p) != pg) return failf(HNY_ERR_IO, "corrupt data.mdb: page %llu carries pgno %llu",
                                     (unsigned long long)pg, (unsigned long long)get64(v->p));
  v->flags = get16(v->p + 10);
  v->lower = get16(v->p + 12), v->upper = get16(v->p + 14);
  if (v->flags & P_OVERFLOW) { v->nkeys = 0; return HNY_OK; }
  if (v->lower < HDR || v->lower > v->upper || v->upper > e->psize || (v->lower & 1))
    return failf(HNY_ERR_IO, "corrupt data.mdb: page %llu bounds %u..%u", (unsigned long long)pg, v->lower, v->upper);
  v->nkeys = (v->lower - HDR) / 2;
  return HNY_OK;
}
struct NodeView {
  const uint8_t *key;
  size_t klen;
  uint16_t flags;
  uint64_t word;  // leaf: data size; branch: child pgno
  const uint8_t *data;
};
int node_at(const hny_lmdb_env *e, const PageView &pv, unsigned i, NodeView *n) {
  uint16_t off = get16(pv.p + HDR + 2 * i);
  if (off < pv.upper || off + NODE > e->psize || (off & 1))
    return failf(HNY_ERR_IO, "corrupt data.mdb: node offset %u", off);
  const uint8_t *q = pv.p + off;
  n->flags = get16(q + 4), n->klen = get16(q + 6), n->key = q + NODE;
  if (off + NODE + n->klen > e->psize) return failf(HNY_ERR_IO, "corrupt data.mdb: key past the page end");
  if (pv.flags & P_BRANCH) {
    n->word = (uint64_t)get16(q) | ((uint64_t)get16(q + 2) << 16) | ((uint64_t)n->flags << 32);
    n->data = nullptr;
  } else {
    n->word = (uint64_t)get16(q) | ((uint64_t)get16(q + 2) << 16);
    n->data = q + NODE + n->klen;
    size_t inl = (n->flags & F_BIGDATA) ? 8 : n->word;
    if (off + NODE + n->klen + inl > e->psize) return failf(HNY_ERR_IO, "corrupt data.mdb: data past the page end");
  }
  return HNY_OK;
}
// value of a leaf node: inline, or the overflow run it points at
int node_value(const hny_lmdb_env *e, const NodeView &n, const uint8_t **val, size_t *vlen, uint64_t *ovpages) {
  *vlen = n.word;
  if (ovpages) *ovpages = 0;
  if (!(n.flags & F_BIGDATA)) { *val = n.data; return HNY_OK; }
  uint64_t pg = get64(n.data);
  PageView ov;
  if (int rc = page_at(e, pg, &ov)) return rc;
  uint32_t np = get32(ov.p + 12);
  if (!(ov.flags & P_OVERFLOW) || np != (HDR - 1 + n.word) / e->psize + 1 ||
      (pg + np) * (uint64_t)e->psize > e->len)
    return failf(HNY_ERR_IO, "corrupt data.mdb: overflow run at page %llu", (unsigned long long)pg);
  *val = ov.p + HDR;
  if (ovpages) *ovpages = np;
  return HNY_OK;
}
// mdb_node_search: binary search; branch pages skip the key-less node 0.  Returns the index of the
// smallest key >= `key` (nkeys if none) and whether it matched.
int node_search(const hny_lmdb_env *e, const PageView &pv, const uint8_t *key, size_t klen, unsigned *idx, bool *exact) {
  int low = (pv.flags & P_BRANCH) ? 1 : 0, high = (int)pv.nkeys - 1, rc = 0, i = 0;
  NodeView n;
  while (low <= high) {
    i = (low + high) >> 1;
    if (int r = node_at(e, pv, (unsigned)i, &n)) return r;
    rc = keycmp(key, klen, n.key, n.klen);
    if (rc == 0) break;
    if (rc > 0) low = i + 1;
    else high = i - 1;
  }
  if (rc > 0) i++;
  *idx = (unsigned)i, *exact = (rc == 0 && pv.nkeys > 0);
  return HNY_OK;
}
int tree_get(const hny_lmdb_env *e, const Db &db, const uint8_t *key, size_t klen, NodeView *out, bool *found) {
  *found = false;
  if (db.root == P_INVALID) return HNY_OK;
  uint64_t pg = db.root;
  for (unsigned d = 0; d < 64; d++) {
    PageView pv;
    if (int rc = page_at(e, pg, &pv)) return rc;
    unsigned i;
    bool exact;
    if (pv.flags & P_BRANCH) {  // mdb_page_search_root
      if (pv.nkeys < 1) return failf(HNY_ERR_IO, "corrupt data.mdb: empty branch page");
      if (int rc = node_search(e, pv, key, klen, &i, &exact)) return rc;
      if (i >= pv.nkeys) i = pv.nkeys - 1;
      else if (!exact) i--;
      NodeView n;
      if (int rc = node_at(e, pv, i, &n)) return rc;
      pg = n.word;
      continue;
    }
    if (!(pv.flags & P_LEAF)) return failf(HNY_ERR_IO, "corrupt data.mdb: page %llu is neither branch nor leaf", (unsigned long long)pg);
    if (int rc = node_search(e, pv, key, klen, &i, &exact)) return rc;
    if (i < pv.nkeys && exact) {
      *found = true;
      return node_at(e, pv, i, out);
    }
    return HNY_OK;
  }
  return failf(HNY_ERR_IO, "corrupt data.mdb: tree deeper than 64");
}

struct Scan {
  const hny_lmdb_env *e;
  const uint8_t *lo, *hi;
  size_t lo_len, hi_len;
  hny_kv_sink sink;
  void *ctx;
  std::string prev;
  bool have_prev = false;
  uint64_t branch = 0, leaf = 0, overflow = 0, entries = 0;
  unsigned leaf_depth = 0;
  bool full;
  int sink_rc = 0;
};
enum { SCAN_STOP = 1, SCAN_SINK = 2 };  // positive returns of scan_page: range end / sink said stop
// [kmin, kmax): the key interval the parent's separators promise for this subtree (null = open)
int scan_page(Scan &s, uint64_t pg, unsigned depth, const std::string *kmin, const std::string *kmax) {
  if (depth > 64) return failf(HNY_ERR_IO, "corrupt data.mdb: tree deeper than 64");
  PageView pv;
  if (int rc = page_at(s.e, pg, &pv)) return rc;
  NodeView n;
  if (pv.flags & P_BRANCH) {
    if (pv.nkeys < 2) return failf(HNY_ERR_IO, "corrupt data.mdb: branch page %llu with %u keys", (unsigned long long)pg, pv.nkeys);
    s.branch++;
    std::vector<std::string> seps(pv.nkeys);
    std::vector<uint64_t> kids(pv.nkeys);
    for (unsigned i = 0; i < pv.nkeys; i++) {
      if (int rc = node_at(s.e, pv, i, &n)) return rc;
      kids[i] = n.word;
      if (i) seps[i].assign((const char *)n.key, n.klen);
      if (i > 1 && keycmp((const uint8_t *)seps[i - 1].data(), seps[i - 1].size(), n.key, n.klen) >= 0)
        return failf(HNY_ERR_IO, "corrupt data.mdb: separators out of order on page %llu", (unsigned long long)pg);
    }
    for (unsigned i = 0; i < pv.nkeys; i++) {
      const std::string *cmin = i ? &seps[i] : kmin, *cmax = i + 1 < pv.nkeys ? &seps[i + 1] : kmax;
      if (!s.full) {  // prune subtrees wholly outside [lo, hi]
        if (s.lo && cmax && keycmp((const uint8_t *)cmax->data(), cmax->size(), s.lo, s.lo_len) <= 0) continue;
        if (s.hi && cmin && keycmp((const uint8_t *)cmin->data(), cmin->size(), s.hi, s.hi_len) > 0) break;
      }
      if (int rc = scan_page(s, kids[i], depth + 1, cmin, cmax)) return rc;
    }
    return HNY_OK;
  }
  if (!(pv.flags & P_LEAF)) return failf(HNY_ERR_IO, "corrupt data.mdb: page %llu has flags %#x", (unsigned long long)pg, pv.flags);
  if (pv.nkeys == 0) return failf(HNY_ERR_IO, "corrupt data.mdb: empty leaf page %llu", (unsigned long long)pg);
  s.leaf++;
  if (s.leaf_depth == 0) s.leaf_depth = depth;
  else if (s.leaf_depth != depth) return failf(HNY_ERR_IO, "corrupt data.mdb: leaves at depths %u and %u", s.leaf_depth, depth);
  for (unsigned i = 0; i < pv.nkeys; i++) {
    if (int rc = node_at(s.e, pv, i, &n)) return rc;
    if (s.have_prev && keycmp((const uint8_t *)s.prev.data(), s.prev.size(), n.key, n.klen) >= 0)
      return failf(HNY_ERR_IO, "corrupt data.mdb: keys out of order on leaf page %llu", (unsigned long long)pg);
    if ((kmin && keycmp(n.key, n.klen, (const uint8_t *)kmin->data(), kmin->size()) < 0) ||
        (kmax && keycmp(n.key, n.klen, (const uint8_t *)kmax->data(), kmax->size()) >= 0))
      return failf(HNY_ERR_IO, "corrupt data.mdb: key outside its separators on leaf page %llu", (unsigned long long)pg);
    s.prev.assign((const char *)n.key, n.klen), s.have_prev = true;
    const uint8_t *val;
    size_t vlen;
    uint64_t ovp;
    if (n.flags & (F_SUBDATA | F_DUPDATA)) return failf(HNY_ERR_UNSUPPORTED, "sub-database / dupsort node inside the index database");
    if (int rc = node_value(s.e, n, &val, &vlen, &ovp)) return rc;
    s.overflow += ovp, s.entries++;
    if (s.lo && keycmp(n.key, n.klen, s.lo, s.lo_len) < 0) continue;
    if (s.hi && keycmp(n.key, n.klen, s.hi, s.hi_len) > 0) return SCAN_STOP;
    if (s.sink && (s.sink_rc = s.sink(s.ctx, n.key, n.klen, val, vlen)) != 0) return SCAN_SINK;
  }
  return HNY_OK;
}

} // namespace

extern "C" {

int hny_lmdb_open(const char *path, const char *db_name, hny_lmdb_env **out) {
  if (!path || !out) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_open: null argument");
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return failf(HNY_ERR_IO, "open %s: %s", path, strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 2 * 512) {
    close(fd);
    return failf(HNY_ERR_IO, "%s: not an LMDB data file (too short)", path);
  }
  void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
  if (m == MAP_FAILED) {
    close(fd);
    return failf(HNY_ERR_IO, "mmap %s: %s", path, strerror(errno));
  }
  auto *e = new hny_lmdb_env;
  e->fd = fd, e->map = (const uint8_t *)m, e->len = (size_t)st.st_size;
  // mdb_env_read_header: meta 0 sits at offset 0 and tells the page size; meta 1 one page later
  auto meta_ok = [&](const uint8_t *p) {
    return (get16(p + 10) & P_META) && get32(p + HDR) == MAGIC && get32(p + HDR + 4) == DATA_VERSION;
  };
  int rc = HNY_OK;
  const uint8_t *m0 = e->map, *best = nullptr;
  if (!meta_ok(m0)) rc = failf(HNY_ERR_IO, "%s: bad magic / version in meta page 0 (MDB_INVALID / MDB_VERSION_MISMATCH)", path);
  if (!rc) {
    e->psize = get32(m0 + HDR + 24);  // mm_psize = free DB's md_pad
    if (e->psize < 512 || e->psize > 32768 || (e->psize & (e->psize - 1)) || 2 * (size_t)e->psize > e->len)
      rc = failf(HNY_ERR_IO, "%s: page size %u", path, e->psize);
  }
  if (!rc) {
    const uint8_t *m1 = e->map + e->psize;
    best = m0;
    if (meta_ok(m1) && get64(m1 + HDR + 32 + 2 * DBSZ) > get64(m0 + HDR + 32 + 2 * DBSZ)) best = m1;
    const uint8_t *q = best + HDR;
    e->map_size = get64(q + 16);
    e->last_pgno = get64(q + 24 + 2 * DBSZ), e->txnid = get64(q + 32 + 2 * DBSZ);
    e->db = db_load(q + 24 + DBSZ);
    if ((e->last_pgno + 1) * (uint64_t)e->psize > e->len)
      rc = failf(HNY_ERR_IO, "%s: truncated (last page %llu)", path, (unsigned long long)e->last_pgno);
  }
  if (!rc && db_name) {  // mdb_dbi_open: the name is a key of the main DB, its data an MDB_db
    NodeView n;
    bool found;
    rc = tree_get(e, e->db, (const uint8_t *)db_name, strlen(db_name), &n, &found);
    if (!rc && !found) rc = failf(HNY_ERR_MISSING_KEY, "no database named '%s' (MDB_NOTFOUND)", db_name);
    if (!rc && (!(n.flags & F_SUBDATA) || n.word != DBSZ)) rc = failf(HNY_ERR_IO, "'%s' is not a database record (MDB_INCOMPATIBLE)", db_name);
    if (!rc) e->db = db_load(n.data);
  }
  if (rc) {
    hny_lmdb_close(e);
    return rc;
  }
  *out = e;
  return HNY_OK;
}

void hny_lmdb_close(hny_lmdb_env *e) {
  if (!e) return;
  if (e->map) munmap((void *)e->map, e->len);
  if (e->fd >= 0) close(e->fd);
  delete e;
}

int hny_lmdb_stat_get(const hny_lmdb_env *e, hny_lmdb_stat *out) {
  if (!e || !out) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_stat_get: null argument");
  out->page_size = e->psize, out->depth = e->db.depth;
  out->branch_pages = e->db.branch, out->leaf_pages = e->db.leaf, out->overflow_pages = e->db.overflow;
  out->entries = e->db.entries, out->last_pgno = e->last_pgno, out->txnid = e->txnid, out->map_size = e->map_size;
  return HNY_OK;
}

int hny_lmdb_get(const hny_lmdb_env *e, const uint8_t *key, size_t klen, const uint8_t **val, size_t *vlen) {
  if (!e || !key || !val || !vlen) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_get: null argument");
  NodeView n;
  bool found;
  if (int rc = tree_get(e, e->db, key, klen, &n, &found)) return rc;
  if (!found) return 0;
  if (n.flags & (F_SUBDATA | F_DUPDATA)) return failf(HNY_ERR_UNSUPPORTED, "sub-database / dupsort node");
  if (int rc = node_value(e, n, val, vlen, nullptr)) return rc;
  return 1;
}

int hny_lmdb_scan(const hny_lmdb_env *e, const uint8_t *lo, size_t lo_len, const uint8_t *hi, size_t hi_len,
                  hny_kv_sink sink, void *ctx) {
  if (!e) return failf(HNY_ERR_INVALID_ARG, "hny_lmdb_scan: null environment");
  Scan s;
  s.e = e, s.lo = lo, s.hi = hi, s.lo_len = lo_len, s.hi_len = hi_len, s.sink = sink, s.ctx = ctx;
  s.full = !lo && !hi;
  if (e->db.root == P_INVALID) {
    if (e->db.entries || e->db.depth) return failf(HNY_ERR_IO, "corrupt data.mdb: empty root with %llu entries", (unsigned long long)e->db.entries);
    return HNY_OK;
  }
  int rc = scan_page(s, e->db.root, 1, nullptr, nullptr);
  if (rc == SCAN_STOP) return HNY_OK;
  if (rc == SCAN_SINK) return s.sink_rc;  // the sink's own code
  if (rc) return rc;
  if (s.full && (s.branch != e->db.branch || s.leaf != e->db.leaf || s.overflow != e->db.overflow ||
                 s.entries != e->db.entries || s.leaf_depth != e->db.depth))
    return failf(HNY_ERR_IO, "corrupt data.mdb: MDB_db says depth %u, %llu/%llu/%llu pages, %llu entries; the tree has depth %u, "
                 "%llu/%llu/%llu pages, %llu entries", e->db.depth, (unsigned long long)e->db.branch,
                 (unsigned long long)e->db.leaf, (unsigned long long)e->db.overflow, (unsigned long long)e->db.entries,
                 s.leaf_depth, (unsigned long long)s.branch, (unsigned long long)s.leaf,
                 (unsigned long long)s.overflow, (unsigned long long)s.entries);
  return HNY_OK;
}

} // extern "C"
