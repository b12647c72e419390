// hny_rust_sort.h — the insertion order of a build: `levels.sort_unstable_by(|(_, a), (_, b)| b.cmp(a))`
// (/root/reference/src/hnsw.rs:268) exactly as Rust's standard library sorts `(u32, usize)` pairs since
// 1.81 (core::slice::sort::unstable, "ipnsort"; third party, restated from the published algorithm).
//
// Why the product needs it: the sort is UNSTABLE, so the order in which items of one level are inserted —
// and with it the graph — is whatever this algorithm leaves behind.  With it, strict mode + batch_max = 1
// rebuilds the reference's own 100-point snapshots record for record (tests: KAT-9); a stable sort (ids
// ascending inside a level, what rounds 1-2 used and what Rust itself does for <= 20 pairs) gets 93 of their
// 167 records wrong.  hny_build_opts.schedule & HNY_SCHED_LEVEL_ORDER_ID selects that older order.
//
// Only the steps that can reorder equal elements are spelled out: the existing-run check, the pivot choice
// (median of 3, recursive from 64 elements), the branchless cyclic Lomuto partition, the quicksort driver
// with its equal-to-ancestor partition, and the heapsort fallback.  Slices of <= 32 elements go to the
// library's small-sort for 16-byte elements (sorting networks on 8, insertion, bidirectional merge): every
// step of it is stable, so std::stable_sort stands in; <= 20 elements: insertion sort, stable as well.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace hny_rust_sort {

template <class P> // P: pair-like, .second = level
struct Sorter {
  // is_less(a, b) of the comparator |a, b| b.level.cmp(&a.level): a sorts before b iff its level is HIGHER
  static bool lt(const P &a, const P &b) { return b.second < a.second; }
  static void small_sort(P *v, size_t n) {
    std::stable_sort(v, v + n, [](const P &a, const P &b) { return lt(a, b); });
  }
  static size_t median3(const P *v, size_t a, size_t b, size_t c) {
    const bool x = lt(v[a], v[b]), y = lt(v[a], v[c]);
    if (x != y) return a;
    const bool z = lt(v[b], v[c]);
    return (z != x) ? c : b;
  }
  static size_t median3_rec(const P *v, size_t a, size_t b, size_t c, size_t n) {
    if (n * 8 >= 64) {
      const size_t n8 = n / 8;
      a = median3_rec(v, a, a + n8 * 4, a + n8 * 7, n8);
      b = median3_rec(v, b, b + n8 * 4, b + n8 * 7, n8);
      c = median3_rec(v, c, c + n8 * 4, c + n8 * 7, n8);
    }
    return median3(v, a, b, c);
  }
  static size_t choose_pivot(const P *v, size_t len) {
    const size_t d = len / 8;
    return len < 64 ? median3(v, 0, d * 4, d * 7) : median3_rec(v, 0, d * 4, d * 7, d);
  }
  // pivot to the front, cyclic Lomuto over the rest (the element taken out first — the "gap" — is
  // classified last), pivot swapped into its place; le = the `!is_less(pivot, x)` form
  static size_t partition(P *v, size_t len, size_t pivot_pos, bool le) {
    std::swap(v[0], v[pivot_pos]);
    const P pivot = v[0];
    P *w = v + 1;
    const size_t n = len - 1;
    size_t num = 0;
    auto goes_left = [&](const P &x) { return le ? !lt(pivot, x) : lt(x, pivot); };
    if (n) {
      const P gap_value = w[0];
      size_t gap = 0;
      for (size_t right = 1; right < n; right++) {
        const bool l = goes_left(w[right]);
        w[gap] = w[num];
        w[num] = w[right];
        gap = right;
        num += l ? 1 : 0;
      }
      const bool l = goes_left(gap_value);
      w[gap] = w[num];
      w[num] = gap_value;
      num += l ? 1 : 0;
    }
    std::swap(v[0], v[num]);
    return num;
  }
  static void sift_down(P *v, size_t len, size_t node) {
    for (;;) {
      size_t child = 2 * node + 1;
      if (child >= len) break;
      if (child + 1 < len && lt(v[child], v[child + 1])) child++;
      if (!lt(v[node], v[child])) break;
      std::swap(v[node], v[child]);
      node = child;
    }
  }
  static void heapsort(P *v, size_t len) { // after 2 * ilog2(n) unbalanced partitions; not reached by level data
    for (size_t i = len + len / 2; i-- > 0;) {
      size_t sift;
      if (i >= len) {
        sift = i - len;
      } else {
        std::swap(v[0], v[i]);
        sift = 0;
      }
      sift_down(v, std::min(i, len), sift);
    }
  }
  static void quicksort(P *v, size_t len, const P *ancestor, uint32_t limit) {
    for (;;) {
      if (len <= 32) return small_sort(v, len);
      if (limit == 0) return heapsort(v, len);
      limit--;
      const size_t pp = choose_pivot(v, len);
      if (ancestor && !lt(*ancestor, v[pp])) { // the pivot equals its predecessor: strip everything equal to it
        const size_t num_le = partition(v, len, pp, true);
        v += num_le + 1;
        len -= num_le + 1;
        ancestor = nullptr;
        continue;
      }
      const size_t num_lt = partition(v, len, pp, false);
      quicksort(v, num_lt, ancestor, limit);
      ancestor = v + num_lt;
      v += num_lt + 1;
      len -= num_lt + 1;
    }
  }
  static void sort(std::vector<P> &vec) {
    P *v = vec.data();
    const size_t len = vec.size();
    if (len < 2) return;
    if (len <= 20) return small_sort(v, len);
    size_t run = 2; // an input that is already sorted (or strictly descending) is returned as is (reversed)
    const bool desc = lt(v[1], v[0]);
    if (desc)
      while (run < len && lt(v[run], v[run - 1])) run++;
    else
      while (run < len && !lt(v[run], v[run - 1])) run++;
    if (run == len) {
      if (desc) std::reverse(vec.begin(), vec.end());
      return;
    }
    uint32_t lg = 0;
    for (size_t x = len | 1; x > 1; x >>= 1) lg++;
    quicksort(v, len, nullptr, 2 * lg);
  }
};

// The batch-synchronous schedule takes the items of a level group in a FIXED PSEUDO-RANDOM order (a batch then
// is a sample of the whole group, whatever the id order is).  Why: every member of a batch searches the graph
// as it stood before the batch; with ids sorted by cluster — documents grouped by topic — a batch of 65 536
// consecutive ids IS a few clusters, none of whose points are in the graph yet, and the index comes out broken
// (measured, bench.py --sort-by-cluster: recall@10 0.42 instead of 0.95 at C2, 0.24 instead of 0.63 at C5).
// The reference never meets this: rayon has a few hundred items in flight and its par_iter hands every thread
// a different stretch of the group, i.e. it interleaves distant parts of it as well.  Any order inside a level
// group is a legitimate parallel execution of hnsw.rs:172-185; batch_max = 1 (the reference with one thread)
// keeps the reference's own order.  Fisher-Yates on splitmix64, seeded by the group's level and size; the
// oracle restates it.
inline uint64_t splitmix64(uint64_t &x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <class P>
void shuffle_level_groups(std::vector<P> &v) {
  size_t b = 0;
  while (b < v.size()) {
    size_t e = b;
    while (e < v.size() && v[e].second == v[b].second) e++;
    uint64_t st = 0x68616E6E6F79ull ^ ((uint64_t)v[b].second << 48) ^ (uint64_t)(e - b);
    for (size_t i = e - b; i > 1; i--) {
      const size_t j = (size_t)(splitmix64(st) % i);
      std::swap(v[b + i - 1], v[b + j]);
    }
    b = e;
  }
}

// levels descending; ties as the reference leaves them (by_id: ascending id / slot instead)
template <class P>
void sort_levels(std::vector<P> &v, bool by_id) {
  if (by_id)
    std::stable_sort(v.begin(), v.end(), [](const P &a, const P &b) { return a.second > b.second; });
  else
    Sorter<P>::sort(v);
}

} // namespace hny_rust_sort
