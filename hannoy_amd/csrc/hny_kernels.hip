// hny_kernels.hip — gfx950 (CDNA4, wave64) kernels of the HNSW build hot path.
//
//   k_walk   : walk_layer          (/root/reference/src/hnsw.rs:460-518) + greedy descent (:303-307)
//              and Reader visit     (/root/reference/src/reader.rs:301-369, 722-767)
//   k_prune  : robust_prune        (hnsw.rs:565-597)
//   k_emit / k_segments / k_apply : add_link, both directions (hnsw.rs:316-324, 523-560)
//   distances: src/distance/*.rs over src/spaces/simple*.rs (K1..K13 of SURVEY.md §2.1)
//
// Execution model: one wave64 == one workgroup == one query / prune job / link target.  All
// bookkeeping values (lengths, positions, candidate ids) are wave-uniform; candidate rows are
// streamed from HBM with 16-byte-per-lane coalesced loads, LPR lanes per row, several rows in
// flight per wave; the beam (res), the tie pool and the neighbour frontier live in LDS; the visited
// set is a per-wave bitset in HBM updated with atomicOr (replaces the RoaringBitmap, hnsw.rs:471).
// HBM-bound integer/f32 streaming work: no MFMA (see DESIGN.md for why).
#include "hny_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
// This file is compiled once per HNY_PART (hannoy_amd/buildlib.py, in parallel):
//   part 0      every kernel with the metric / strict / incremental / reader switches at run time
//               (the general path) + the launch entry points the host calls;
//   part 1..7   the four build kernels specialised for metric HNY_PART-1 (`SP` below): the metric
//               switch, strict mode, the incremental-build branches and the Reader fallback are
//               compile-time dead there.  The general k_walk<64,3> spills 60 VGPRs at the 128 of 4
//               waves per SIMD and issues ~270 instructions per distance evaluation; specialised
//               it fits without scratch (C2: walk 0.353 -> 0.328 s, prune 0.109 -> 0.093 s).
#ifndef HNY_PART
#define HNY_PART 0
#endif
#if HNY_PART == 0
#include <rocprim/device/device_radix_sort.hpp>
#endif

// block == one wave (every user of WSYNC is a 64-thread kernel): lanes exchange data through LDS, and
// the LDS executes one wave's operations in issue order, so all that is needed is that the compiler
// keeps the order and that pending LDS returns are waited for.  __syncthreads() did that too, but it
// also drains vmcnt(0) — every row / neighbour-list load in flight — at each of the ~20 syncs of an
// expansion.
#ifdef HNY_WSYNC_BARRIER
#define WSYNC() __syncthreads()
#else
#define WSYNC()                                               \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");    \
  } while (0)
#endif

#ifndef HNY_RB_MERGE
#define HNY_RB_MERGE 1
#endif
#ifndef HNY_RB_MAX_NCH
#define HNY_RB_MAX_NCH 3
#endif
#ifndef HNY_RB_MERGE_MIN
#define HNY_RB_MERGE_MIN 2
#endif

// Lane of the calling wave.  The wave-level routines below (distances, beam, visited set, walk_one_layer)
// run in 64-thread blocks — one wave is the workgroup, and with __launch_bounds__(64) the mask folds away —
// and in the waves of larger workgroups (the mask then selects the wave-relative lane).
#define HNY_LANE ((int)(threadIdx.x & 63u))

namespace {

__device__ __forceinline__ u32 fbits(float f) { return __float_as_uint(f); }
// the wave's vote on a predicate as a 64-bit mask.  (HIP's __ballot takes an int: a bool that the compiler
// holds as a lane mask is first turned into 0 / 1 per lane and compared with 0 again — two vector
// instructions per vote, ~20 votes per expansion of a walk)
__device__ __forceinline__ u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// SP = 0: general kernel.  SP = metric + 1: the kernel's private copy of GraphDev carries the
// metric as a constant and has strict mode and the incremental branches switched off, so that
// every `g.metric` / `g.mclass` / `g.x86_order` / `g.incremental` test below folds at compile time.
template <int SP>
__device__ __forceinline__ void specialize(GraphDev &g) {
  if constexpr (SP != 0) {
    g.metric = SP - 1;
    g.mclass = SP - 1 < 3 ? SP - 1 : (int)MC_BIN; // cosine/euclidean/manhattan = MC_DOT/MC_L2/MC_L1
    g.x86_order = 0;
    g.incremental = 0;
  }
}
// wave-uniform values that were read from LDS or produced by a cross-lane reduction sit in VGPRs;
// readfirstlane moves them to SGPRs (lower VGPR pressure, scalar branches)
__device__ __forceinline__ u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 uni64(u64 x) { return ((u64)uni((u32)(x >> 32)) << 32) | (u64)uni((u32)x); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ u64 uni(u64 x) {
  return ((u64)uni((u32)(x >> 32)) << 32) | (u64)uni((u32)(x & 0xFFFFFFFFull));
}

// ---------------------------------------------------------------------------------------------
// distance inner loops.  Lane t of an LPR-lane group owns 16-byte unit #(c*LPR + t), c < NCH.
// f32 ("wave order", restated in oracle/hannoy_oracle.cpp wave_reduce): one fma chain per lane
// over its 4*NCH elements in index order, then an xor butterfly LPR/2 .. 1.
// ---------------------------------------------------------------------------------------------
template <int LPR, int NCH>
__device__ __forceinline__ void load_row(const unsigned char *p, int t, u32 n16, float4 (&r)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    u32 f = (u32)(c * LPR + t);
    r[c] = f < n16 ? *reinterpret_cast<const float4 *>(p + (size_t)f * 16)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// same, from an LDS image of the row.  The pointer is cast to address space 3 explicitly: a generic
// pointer here makes hipcc emit flat_load, whose waits (vmcnt(0) + lgkmcnt(0)) serialise against
// every HBM load in flight.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4_t lds_cf4;
template <int LPR, int NCH>
__device__ __forceinline__ void load_row_lds(const unsigned char *p, int t, u32 n16, float4 (&r)[NCH]) {
  const unsigned lp = (unsigned)(size_t)p; // low 32 bits of a generic LDS address = LDS offset
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    u32 f = (u32)(c * LPR + t);
    if (f < n16) {
      f32x4_t v = *reinterpret_cast<lds_cf4 *>((size_t)(lp + f * 16u));
      r[c] = make_float4(v.x, v.y, v.z, v.w);
    } else {
      r[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int NCH>
__device__ __forceinline__ float partial_f32(int mclass, const float4 (&q)[NCH],
                                             const float4 (&r)[NCH]) {
  float acc = 0.f;
  if (mclass == MC_DOT) { // K1 dot_product (simple_avx.rs:69-110 computes the same sum)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      acc = __builtin_fmaf(q[c].x, r[c].x, acc);
      acc = __builtin_fmaf(q[c].y, r[c].y, acc);
      acc = __builtin_fmaf(q[c].z, r[c].z, acc);
      acc = __builtin_fmaf(q[c].w, r[c].w, acc);
    }
  } else if (mclass == MC_L2) { // K2 squared euclidean (simple_avx.rs:17-65)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float d0 = q[c].x - r[c].x, d1 = q[c].y - r[c].y, d2 = q[c].z - r[c].z, d3 = q[c].w - r[c].w;
      acc = __builtin_fmaf(d0, d0, acc);
      acc = __builtin_fmaf(d1, d1, acc);
      acc = __builtin_fmaf(d2, d2, acc);
      acc = __builtin_fmaf(d3, d3, acc);
    }
  } else { // K10 manhattan (manhattan.rs:41-43)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      acc = acc + __builtin_fabsf(q[c].x - r[c].x);
      acc = acc + __builtin_fabsf(q[c].y - r[c].y);
      acc = acc + __builtin_fabsf(q[c].z - r[c].z);
      acc = acc + __builtin_fabsf(q[c].w - r[c].w);
    }
  }
  return acc;
}

// K11..K13: popcount(u ^ v) (hamming.rs:55-85, simple.rs:119-131, binary_quantized_*.rs)
template <int NCH>
__device__ __forceinline__ u32 partial_bin(const float4 (&q)[NCH], const float4 (&r)[NCH]) {
  u32 pc = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    pc += __popc(__float_as_uint(q[c].x) ^ __float_as_uint(r[c].x));
    pc += __popc(__float_as_uint(q[c].y) ^ __float_as_uint(r[c].y));
    pc += __popc(__float_as_uint(q[c].z) ^ __float_as_uint(r[c].z));
    pc += __popc(__float_as_uint(q[c].w) ^ __float_as_uint(r[c].w));
  }
  return pc;
}

// value of lane (lane ^ OFF), OFF a power of two: the data movement of __shfl_xor(v, OFF, 64) without
// its per-call index arithmetic (xor, width clamp, select, shift = 4 VALU instructions in front of
// every ds_bpermute) and without the trip through the LDS pipe, which queues behind the row reads:
// 1, 2, 8: one DPP move (quad_perm / row_ror:8; folds into the consuming add); 4: row_half_mirror then
// quad_perm:[3,2,1,0] (7 - i, then reversed inside its quad = i ^ 4); 16, 32: gfx950's
// v_permlane16_swap / v_permlane32_swap of the value with itself, then a select.
template <int OFF>
__device__ __forceinline__ u32 xshfl_u32(u32 v) {
  static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "xor lane offset");
  if constexpr (OFF == 1) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xF, 0xF, false);
  } else if constexpr (OFF == 2) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /*quad_perm:[2,3,0,1]*/, 0xF, 0xF, false);
  } else if constexpr (OFF == 4) {
    const int m = __builtin_amdgcn_update_dpp(0, (int)v, 0x141 /*row_half_mirror*/, 0xF, 0xF, false);
    return (u32)__builtin_amdgcn_update_dpp(0, m, 0x1B /*quad_perm:[3,2,1,0]*/, 0xF, 0xF, false);
  } else if constexpr (OFF == 8) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /*row_ror:8*/, 0xF, 0xF, false);
  } else if constexpr (OFF == 16) {
    // rows of 16 lanes [v0 v1 v2 v3] -> {[v0 v0 v2 v2], [v1 v1 v3 v3]}
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16u) ? r[0] : r[1];
  } else {
    // halves [lo hi] -> {[lo lo], [hi hi]}
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32u) ? r[0] : r[1];
  }
}
template <int OFF>
__device__ __forceinline__ float xshfl(float v) {
  return __uint_as_float(xshfl_u32<OFF>(__float_as_uint(v)));
}
template <int OFF>
__device__ __forceinline__ u32 xshfl(u32 v) {
  return xshfl_u32<OFF>(v);
}
template <int OFF>
__device__ __forceinline__ int xshfl(int v) {
  return (int)xshfl_u32<OFF>((u32)v);
}
template <int OFF>
__device__ __forceinline__ u64 xshfl(u64 v) {
  return ((u64)xshfl_u32<OFF>((u32)(v >> 32)) << 32) | (u64)xshfl_u32<OFF>((u32)v);
}
// v[t] + v[t ^ OFF] for OFF = FROM, FROM/2, ... 1 (the xor butterfly, in that order)
template <int FROM, typename T>
__device__ __forceinline__ T xor_sum_from(T v) {
  if constexpr (FROM >= 1) {
    v = v + xshfl<FROM>(v);
    return xor_sum_from<FROM / 2, T>(v);
  } else {
    return v;
  }
}

template <int LPR>
__device__ __forceinline__ float butterfly_f32(float v) {
  return xor_sum_from<LPR / 2, float>(v);
}
template <int LPR>
__device__ __forceinline__ u32 butterfly_u32(u32 v) {
  return xor_sum_from<LPR / 2, u32>(v);
}

// Folded butterflies.  The xor butterfly computes, at every step, v[t] + v[t ^ off] on all lanes, so
// lanes t and t ^ off hold the same value.  When R rows are reduced together the two halves can work
// on DIFFERENT rows instead of duplicating each other: the pair sums are the very same f32
// additions (bit-identical result), but 4 rows cost 2+1+(log2(LPR)-2) shuffles instead of
// 4*log2(LPR).  Row j of a fold4 ends up in the lanes with bit(LPR/2) == (j & 1) and
// bit(LPR/4) == (j >> 1); fold2: bit(LPR/2) == j.
template <int OFF, typename T>
__device__ __forceinline__ T fold_step(T a, T b) {
  static_assert(sizeof(T) == 4, "32-bit lanes");
  if constexpr (OFF == 32 || OFF == 16) {
    // the swap IS the exchange: {[a.lo b.lo], [a.hi b.hi]} (halves; rows of 16 likewise), and the sum
    // of the two is keep + recv on every lane (IEEE addition commutes, so the bits are the same)
    u32 ua, ub;
    __builtin_memcpy(&ua, &a, 4);
    __builtin_memcpy(&ub, &b, 4);
    const auto r = OFF == 32 ? __builtin_amdgcn_permlane32_swap(ua, ub, false, false)
                             : __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
    T x, y;
    const u32 r0 = r[0], r1 = r[1];
    __builtin_memcpy(&x, &r0, 4);
    __builtin_memcpy(&y, &r1, 4);
    return x + y;
  } else {
    const bool hi = (threadIdx.x & OFF) != 0;
    T keep = hi ? b : a, send = hi ? a : b;
    return keep + xshfl<OFF>(send);
  }
}
template <int LPR, typename T>
__device__ __forceinline__ T fold4(T p0, T p1, T p2, T p3) {
  T x01 = fold_step<LPR / 2, T>(p0, p1), x23 = fold_step<LPR / 2, T>(p2, p3);
  T y = fold_step<LPR / 4, T>(x01, x23);
  return xor_sum_from<LPR / 8, T>(y);
}
template <int LPR, typename T>
__device__ __forceinline__ T fold2(T p0, T p1) {
  T y = fold_step<LPR / 2, T>(p0, p1);
  return xor_sum_from<LPR / 4, T>(y);
}
template <int LPR>
__device__ __forceinline__ int fold4_row() { // which of the 4 rows this lane's fold4 result belongs to
  return (int)((threadIdx.x / (LPR / 2)) & 1) | (int)(((threadIdx.x / (LPR / 4)) & 1) << 1);
}
template <int LPR>
__device__ __forceinline__ int fold2_row() {
  return (int)((threadIdx.x / (LPR / 2)) & 1);
}

__device__ __forceinline__ float finalize_f32(const GraphDev &g, float acc, float qn, float rn) {
  if (g.metric == 0 /*HNY_COSINE, cosine.rs:40-56*/) {
    float pnqn = qn * rn;
    if (pnqn > 1.1920929e-07f) {
      float c = acc / pnqn;
      if (c < -1.0f) c = -1.0f;
      if (c > 1.0f) c = 1.0f;
      return (1.0f - c) / 2.0f;
    }
    return 0.0f;
  }
  return acc; // squared L2 (euclidean.rs:42-44) / L1
}

__device__ __forceinline__ float finalize_bin(const GraphDev &g, u32 pop, float qn, float rn) {
  switch (g.metric) {
    case 3: // hamming.rs:44-47: popcount / padded dims.  A power-of-two divisor (bin_inv != 0: its exact
      // reciprocal) makes the IEEE division an exact scaling: one multiply, the same bits
      return g.bin_inv != 0.f ? (float)pop * g.bin_inv : (float)pop / (float)g.bin_bits;
    case 4: { // binary_quantized_cosine.rs:44-59
      float pq = (float)((int)g.bin_bits - 2 * (int)pop);
      float pnqn = qn * rn;
      if (pnqn != 0.0f) {
        float c = pq / pnqn;
        return (1.0f - c) / 2.0f;
      }
      return 0.0f;
    }
    case 5: // binary_quantized_euclidean.rs:76-83
      return (float)(pop * 4u);
    default: // binary_quantized_manhattan.rs:72-79
      return (float)(pop * 2u);
  }
}

template <int NCH>
struct RowsInFlight {
  // 16-byte loads in flight per lane ~ 12..16
  static constexpr int U = NCH >= 12 ? 1 : (NCH >= 6 ? 2 : (NCH >= 3 ? 4 : (NCH == 2 ? 6 : 8)));
};
// dist_rows: load groups per pass.  A pass covers (64 / LPR) * U rows and the build never has more
// than max(M, M0) <= 64 new rows at once — typically ~11 — while every group of a pass costs
// instructions even when it is skipped: short rows take 4 groups (LPR 8: 32 rows per pass).
template <int LPR, int NCH>
struct DistGroups {
  static constexpr int U = (NCH == 1 && LPR <= 16) ? 4 : RowsInFlight<NCH>::U;
};

// Strict mode: the f32 distances exactly as the reference computes them on an x86_64 host with
// AVX+FMA (dispatch simple.rs:19-47,53-79): dim >= 32 -> 32 fma partials, element i -> partial
// i % 32, hsum256 per 8 partials then ((h1+h2)+h3)+h4, unfused scalar tail (simple_avx.rs);
// 16 <= dim < 32 -> 16 unfused partials (simple_sse.rs); below -> scalar left to right
// (simple.rs:49-51,81-83); Manhattan is always the scalar iterator sum (manhattan.rs:41-43).
// One half-wave per row (lane j = partial j), dword loads: slower than the wave order, meant for
// parity runs (x86_order = 1).
__device__ __forceinline__ float x86_row_distance(const GraphDev &g, const float *a, const float *b,
                                                  int j, int base) {
  const u32 dim = g.dim;
  float r;
  if (g.mclass == MC_L1) {
    r = 0.f;
    for (u32 i = 0; i < dim; i++) r = r + __builtin_fabsf(a[i] - b[i]);
    return r;
  }
  const bool dot = g.mclass == MC_DOT;
  if (dim >= 32) {
    const u32 m = dim - dim % 32;
    float acc = 0.f;
    for (u32 i = 0; i < m; i += 32) {
      const float x = a[i + j], y = b[i + j];
      if (dot) {
        acc = __builtin_fmaf(x, y, acc);
      } else {
        const float d = x - y;
        acc = __builtin_fmaf(d, d, acc);
      }
    }
    acc = acc + __shfl_xor(acc, 4, 64);
    acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 1, 64);
    const float h1 = __shfl(acc, base, 64), h2 = __shfl(acc, base + 8, 64),
                h3 = __shfl(acc, base + 16, 64), h4 = __shfl(acc, base + 24, 64);
    r = ((h1 + h2) + h3) + h4;
    for (u32 i = m; i < dim; i++) {
      float p;
      if (dot) {
        p = a[i] * b[i];
      } else {
        const float d = a[i] - b[i];
        p = d * d;
      }
      r = r + p;
    }
  } else if (dim >= 16) {
    const u32 m = dim - dim % 16;
    float acc = 0.f;
    if (j < 16)
      for (u32 i = 0; i < m; i += 16) {
        const float x = a[i + j], y = b[i + j];
        float p;
        if (dot) {
          p = x * y;
        } else {
          const float d = x - y;
          p = d * d;
        }
        acc = p + acc;
      }
    acc = acc + __shfl_xor(acc, 2, 64);
    acc = acc + __shfl_xor(acc, 1, 64);
    const float h1 = __shfl(acc, base, 64), h2 = __shfl(acc, base + 4, 64),
                h3 = __shfl(acc, base + 8, 64), h4 = __shfl(acc, base + 12, 64);
    r = ((h1 + h2) + h3) + h4;
    for (u32 i = m; i < dim; i++) {
      float p;
      if (dot) {
        p = a[i] * b[i];
      } else {
        const float d = a[i] - b[i];
        p = d * d;
      }
      r = r + p;
    }
  } else {
    r = 0.f;
    for (u32 i = 0; i < dim; i++) {
      float p;
      if (dot) {
        p = a[i] * b[i];
      } else {
        const float d = a[i] - b[i];
        p = d * d;
      }
      r = r + p;
    }
  }
  return r;
}

__device__ void dist_rows_x86(const GraphDev &g, const unsigned char *qrow, float qn, const u32 *ids,
                              int n, float *out) {
  const int ln = threadIdx.x & 63, j = ln & 31, half = ln >> 5;
  for (int k0 = 0; k0 < n; k0 += 2) {
    int ri = k0 + half;
    const bool on = ri < n;
    if (!on) ri = n - 1;
    const u32 rid = ids[ri];
    const float *b = reinterpret_cast<const float *>(g.rows + (size_t)rid * g.row_stride);
    const float acc = x86_row_distance(g, reinterpret_cast<const float *>(qrow), b, j, half * 32);
    const float d = finalize_f32(g, acc, qn, g.norms ? g.norms[rid] : 0.f);
    if (j == 0 && on) out[ri] = d;
  }
}

// distances from the query (registers) to rows ids[0..n) (LDS) -> out[0..n) (LDS).
// D::distance at hnsw.rs:476,503,584.
template <int LPR, int NCH>
__device__ __forceinline__ void dist_rows(const GraphDev &g, const float4 (&q)[NCH], float qn,
                                          const u32 *ids, int n, float *out,
                                          const unsigned char *qrow) {
  if (g.x86_order && g.mclass != MC_BIN) { // strict mode (wave-uniform)
    dist_rows_x86(g, qrow, qn, ids, n, out);
    return;
  }
  constexpr int RPG = 64 / LPR;               // rows per wave-wide load instruction
  constexpr int U = DistGroups<LPR, NCH>::U;  // load groups in flight
  constexpr int RPI = RPG * U;
  const int ln = HNY_LANE, t = ln % LPR, sub = ln / LPR;
  for (int k0 = 0; k0 < n; k0 += RPI) {
    float4 r[U][NCH];
    float rn[U];
    u32 rids[U]; // every group's row id first: the LDS reads go out back to back, not one per branch
#pragma unroll
    for (int u = 0; u < U; u++) {
      int ri = k0 + u * RPG + sub;
      if (ri > n - 1) ri = n - 1;
      rids[u] = ids[ri];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      rn[u] = 0.f;
      if (k0 + u * RPG < n) { // wave-uniform
        u32 rid = rids[u];
        if (LPR == 64) rid = __builtin_amdgcn_readfirstlane(rid);
        const unsigned char *p = g.rows + (size_t)rid * g.row_stride;
        load_row<LPR, NCH>(p, t, g.n16, r[u]);
        if (g.norms) rn[u] = g.norms[rid];
      } else {
#pragma unroll
        for (int c = 0; c < NCH; c++) r[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // reduce the U load groups 4 (or 2) at a time with folded butterflies
    constexpr int F = (U % 4 == 0) ? 4 : (U % 2 == 0 ? 2 : 1);
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += F) {
      if (k0 + u0 * RPG < n) { // wave-uniform
        int j;
        float d;
        if (g.mclass == MC_BIN) {
          u32 pc;
          if constexpr (F == 4) {
            pc = fold4<LPR, u32>(partial_bin<NCH>(q, r[u0]), partial_bin<NCH>(q, r[u0 + 1]),
                                 partial_bin<NCH>(q, r[u0 + 2]), partial_bin<NCH>(q, r[u0 + 3]));
            j = fold4_row<LPR>();
          } else if constexpr (F == 1) {
            pc = butterfly_u32<LPR>(partial_bin<NCH>(q, r[u0]));
            j = 0;
          } else {
            pc = fold2<LPR, u32>(partial_bin<NCH>(q, r[u0]), partial_bin<NCH>(q, r[u0 + 1]));
            j = fold2_row<LPR>();
          }
          float rnj = rn[u0];
#pragma unroll
          for (int jj = 1; jj < F; jj++) rnj = (j == jj) ? rn[u0 + jj] : rnj;
          d = finalize_bin(g, pc, qn, rnj);
        } else {
          float pa;
          if constexpr (F == 4) {
            pa = fold4<LPR, float>(partial_f32<NCH>(g.mclass, q, r[u0]),
                                   partial_f32<NCH>(g.mclass, q, r[u0 + 1]),
                                   partial_f32<NCH>(g.mclass, q, r[u0 + 2]),
                                   partial_f32<NCH>(g.mclass, q, r[u0 + 3]));
            j = fold4_row<LPR>();
          } else if constexpr (F == 1) {
            pa = butterfly_f32<LPR>(partial_f32<NCH>(g.mclass, q, r[u0]));
            j = 0;
          } else {
            pa = fold2<LPR, float>(partial_f32<NCH>(g.mclass, q, r[u0]),
                                   partial_f32<NCH>(g.mclass, q, r[u0 + 1]));
            j = fold2_row<LPR>();
          }
          float rnj = rn[u0];
#pragma unroll
          for (int jj = 1; jj < F; jj++) rnj = (j == jj) ? rn[u0 + jj] : rnj;
          d = finalize_f32(g, pa, qn, rnj);
        }
        int ri = k0 + (u0 + j) * RPG + sub;
        if ((t & (F == 1 ? LPR - 1 : LPR / F - 1)) == 0 && ri < n) out[ri] = d;
      }
    }
  }
}

// The wave order of a 16- or 32-lane row group, computed by EIGHT lanes per row (k_walk, rows of
// 9..32 sixteen-byte units: 128-d f32, 129..512-B codes).  One wave-wide load then covers 8 rows
// instead of 4 or 2, and the per-group instructions (row id, address, exec mask, shuffles,
// finaliser) are paid once per 8 rows: ~90 instead of ~220 VALU instructions for the ~11 new rows of
// an expansion at 128-d.  Bit-identical by construction: lane t holds units t, t + 8 (, t + 16,
// t + 24), runs the same 4-element chain per unit and adds the per-unit partials in the order of the
// xor butterfly's first steps — off = 16: p[t] + p[t ^ 16], off = 8: that + its partner — before
// the butterfly continues across the 8 lanes with off = 4, 2, 1.
template <int LPRO>
__device__ __forceinline__ void dist_rows_narrow(const GraphDev &g, const float4 (&q)[LPRO / 8], float qn,
                                                 const u32 *ids, int n, float *out) {
  static_assert(LPRO == 16 || LPRO == 32, "8 lanes stand in for 16 or 32");
  constexpr int NQ = LPRO / 8, U = 2;
  const int ln = HNY_LANE, t = ln & 7, sub = ln >> 3;
  const int j2 = fold2_row<8>();
  for (int k0 = 0; k0 < n; k0 += 8 * U) {
    float4 r[U][NQ];
    float rn[U];
    u32 rids[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int ri = k0 + u * 8 + sub;
      if (ri > n - 1) ri = n - 1;
      rids[u] = ids[ri];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      rn[u] = 0.f;
      if (k0 + u * 8 < n) { // wave-uniform
        const unsigned char *p = g.rows + (size_t)rids[u] * g.row_stride;
#pragma unroll
        for (int c = 0; c < NQ; c++) {
          const u32 f = (u32)(c * 8 + t);
          r[u][c] = f < g.n16 ? *reinterpret_cast<const float4 *>(p + (size_t)f * 16)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (g.norms) rn[u] = g.norms[rids[u]];
      } else {
#pragma unroll
        for (int c = 0; c < NQ; c++) r[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const bool two = k0 + 8 < n; // wave-uniform: the second group holds rows
    float d;
    int j;
    if (g.mclass == MC_BIN) {
      u32 pc[U];
#pragma unroll
      for (int u = 0; u < U; u++) pc[u] = partial_bin<NQ>(q, r[u]);
      u32 y;
      if (two) {
        y = fold2<8, u32>(pc[0], pc[1]);
        j = j2;
      } else {
        y = butterfly_u32<8>(pc[0]);
        j = 0;
      }
      d = finalize_bin(g, y, qn, j ? rn[1] : rn[0]);
    } else {
      float pa[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        float pu[NQ];
#pragma unroll
        for (int c = 0; c < NQ; c++) {
          const float4 qc[1] = {q[c]}, rc[1] = {r[u][c]};
          pu[c] = partial_f32<1>(g.mclass, qc, rc);
        }
        if constexpr (NQ == 4)
          pa[u] = (pu[0] + pu[2]) + (pu[1] + pu[3]); // off = 16, then off = 8
        else
          pa[u] = pu[0] + pu[1];                     // off = 8
      }
      float y;
      if (two) {
        y = fold2<8, float>(pa[0], pa[1]);
        j = j2;
      } else {
        y = butterfly_f32<8>(pa[0]);
        j = 0;
      }
      d = finalize_f32(g, y, qn, j ? rn[1] : rn[0]);
    }
    const int ri = k0 + j * 8 + sub;
    if ((two ? (t & 3) == 0 : t == 0) && ri < n) out[ri] = d;
  }
}

// neighbour list of (layer, node): ids in insertion order, HNY_SENT beyond the count
// (get_neighbours, hnsw.rs:428-456, fresh DB: in-memory lists only)
__device__ __forceinline__ const u32 *nbr_ids(const GraphDev &g, u32 layer, u32 node, u32 &cap) {
  if (layer == 0) {
    cap = g.M0;
    return g.l0_ids + (size_t)node * g.M0;
  }
  int ui = g.upper_idx[node];
  cap = g.M;
  return g.up_ids + ((size_t)ui * g.up_layers + (layer - 1)) * g.M;
}
// on-disk Links of the previous build (incremental only); null when the node has no such layer
__device__ __forceinline__ const u32 *disk_ids(const GraphDev &g, u32 layer, u32 node, u32 &cap) {
  if (layer == 0) {
    cap = g.M0;
    return g.d0_ids + (size_t)node * g.M0;
  }
  cap = g.M;
  int ui = g.upper_idx[node];
  if (ui < 0 || layer > g.up_layers) return nullptr;
  return g.du_ids + ((size_t)ui * g.up_layers + (layer - 1)) * g.M;
}

// ---------------------------------------------------------------------------------------------
// beam state in LDS.  res: sorted ascending array of keys (dist bits << 32 | slot << 1 | expanded)
// == MinMaxHeap `res` (hnsw.rs:470) and, through the `expanded` bit, the part of the `candidates`
// BinaryHeap (hnsw.rs:469) that can still be popped.  pool: candidates evicted from res whose
// distance ties res.max (they are popped before in-res entries of equal distance because the
// BinaryHeap pops the larger id first).  Entries evicted with a larger distance can never be
// popped before the walk breaks and are dropped.
// ---------------------------------------------------------------------------------------------
struct Beam {
  u64 *res;
  u64 *pool;
  int res_len, pool_len, rcap;
  int n_weird;    // pool entries whose distance bits are not an ordinary non-negative float
  u32 tie_bits;   // distance bits shared by the ordinary pool entries (== bits of res.max)
  int dropped;    // some evicted ordinary candidate with bits > res.max is still in `candidates`
  u32 pool_over, err;
#ifdef HNY_PHASE_CLOCKS
  u64 ph[12], ph_t; // cycles in pop / list+visited / distances / insert, expansions (5..10: walk_layer_short's extras); last stamp
#endif
};
#ifdef HNY_PHASE_CLOCKS
#define PH_STAMP(s, i)                                   \
  do {                                                   \
    const u64 now_ = __builtin_readcyclecounter();       \
    (s).ph[i] += now_ - (s).ph_t;                        \
    (s).ph_t = now_;                                     \
  } while (0)
#else
#define PH_STAMP(s, i) do { } while (0)
#endif

// OrderedFloat orders by bit pattern (ordered_float.rs:25-29) while the two raw compares in
// walk_layer (hnsw.rs:485, 505) use float order.  They disagree only for sign-bit-set or NaN
// distances ("weird": BQ-cosine has no clamp and gives -6e-8 for identical codes when
// fl(sqrt(D))^2 < D).  Such entries sort last in every heap but never trigger the `f > f_max`
// break, so evicted ones must stay poppable.
__device__ __forceinline__ bool weird_bits(u32 b) { return b > 0x7F800000u; }

// minimum of a 64-bit value over the wave, on every lane: xor butterfly on DPP / permlane swaps (no LDS)
template <int OFF>
__device__ __forceinline__ u64 wave_min_step(u64 v) {
  if constexpr (OFF >= 1) {
    const u64 o = xshfl<OFF>(v);
    return wave_min_step<OFF / 2>(o < v ? o : v);
  } else {
    return v;
  }
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) { return wave_min_step<32>(v); }

// Every field below is wave-uniform by construction, but the compiler cannot always prove it: a lane-conditional
// store (`if (ln == 0) pool[i] = key`) that ends a uniform region makes the join behind it a "divergent" phi for
// whatever else is merged there, and from then on every `if (s.pool_len > 0)`, every loop exit, is compiled as
// exec-mask arithmetic (s_and_saveexec / s_andn2 / s_or / s_cbranch_execz) instead of one scalar branch — ~400
// of the ~430 scalar instructions of a C5 expansion were that.  settle() pins the state back into SGPRs.
__device__ __forceinline__ void settle(Beam &s) {
  s.res_len = uni(s.res_len);
  s.pool_len = uni(s.pool_len);
  s.n_weird = uni(s.n_weird);
  s.tie_bits = uni(s.tie_bits);
  s.dropped = uni(s.dropped);
  s.pool_over = uni(s.pool_over);
  s.err = uni(s.err);
}
__device__ __forceinline__ void pool_push(Beam &s, u64 key) {
  if (s.pool_len < HNY_POOL_CAP) {
    if (HNY_LANE == 0) s.pool[s.pool_len] = key & ~1ull;
    s.pool_len++;
    if (weird_bits((u32)(key >> 32))) s.n_weird++;
    WSYNC();
  } else {
    s.pool_over++;
  }
  settle(s);
}

// res.max moved below the ordinary pool entries: they can never be popped before the break
__device__ __forceinline__ void pool_drop_ties(Beam &s) {
  if (s.pool_len - s.n_weird <= 0) return;
  s.dropped = true;
  if (s.n_weird == 0) {
    s.pool_len = 0;
    return;
  }
  const int ln = HNY_LANE;
  u64 a0 = ln < s.pool_len ? s.pool[ln] : 0ull;
  u64 a1 = ln + 64 < s.pool_len ? s.pool[ln + 64] : 0ull;
  bool k0 = ln < s.pool_len && weird_bits((u32)(a0 >> 32));
  bool k1 = ln + 64 < s.pool_len && weird_bits((u32)(a1 >> 32));
  u64 m0 = ballot(k0), m1 = ballot(k1);
  u64 lt = (1ull << ln) - 1ull;
  WSYNC();
  if (k0) s.pool[__popcll(m0 & lt)] = a0;
  if (k1) s.pool[__popcll(m0) + __popcll(m1 & lt)] = a1;
  s.pool_len = __popcll(m0) + __popcll(m1);
  WSYNC();
  settle(s);
}

// an entry leaves res (push_pop_max); nd = distance bits of the res.max that remains
__device__ __forceinline__ void beam_evicted(Beam &s, u64 x, u32 nd) {
  if (x & 1ull) return; // already popped from `candidates`
  const u32 xb = (u32)(x >> 32);
  if (weird_bits(xb)) {
    pool_push(s, x);
  } else if (xb == nd) {
    s.tie_bits = nd;
    pool_push(s, x);
  } else {
    s.dropped = true;
  }
}

// res.push (len != ef) or res.push_pop_max (len == ef), hnsw.rs:508-512
__device__ __forceinline__ void beam_insert(Beam &s, u64 key, int ef) {
  const int ln = HNY_LANE;
  const int len = s.res_len;
  int pos = 0;
  for (int base = 0; base < len; base += 64) {
    int e = base + ln;
    bool lt = e < len && ((s.res[e] & ~1ull) < key);
    pos += __popcll(ballot(lt));
  }
  const bool evict = (len == ef);
  const u64 oldmax = len ? uni(s.res[len - 1]) : 0ull;
  if (evict && pos == len) { // the new entry is the max: pushed and popped at once
    beam_evicted(s, key, (u32)(oldmax >> 32));
    return;
  }
  if (!evict && len >= s.rcap) {
    s.err = 1;
    return;
  }
  const int hi = evict ? len - 1 : len; // entries [pos, hi) move up by one
  if (hi > pos) {
    for (int base = (hi - 1) & ~63; base >= (pos & ~63); base -= 64) {
      int e = base + ln;
      bool mv = e >= pos && e < hi;
      u64 v = mv ? s.res[e] : 0ull;
      WSYNC();
      if (mv) s.res[e + 1] = v;
      WSYNC();
    }
  }
  if (ln == 0) s.res[pos] = key;
  WSYNC();
  if (!evict) {
    s.res_len = len + 1;
    return;
  }
  const u32 nd = uni((u32)(s.res[len - 1] >> 32));
  if (s.pool_len - s.n_weird > 0 && s.tie_bits != nd) pool_drop_ties(s);
  beam_evicted(s, oldmax, nd);
}

// ---- the same beam held in REGISTERS (RB kernels: rows <= 2 KB, ef <= 127, <= 64 entry points).
// On short rows the walk is bound by instruction issue, not by bytes: two thirds of its
// instructions are beam / visited bookkeeping, and every res access above is an LDS round trip plus
// address arithmetic.  Entry e of the sorted array lives in lane e % 64 of r0 (e < 64) or r1; an
// insertion is a ballot rank and a one-lane DPP shift, a pop is a ballot on the expanded bits.
template <int RC>
struct BeamR { // RC x 64 entries: 2 for ef <= 127, 4 for ef <= 255
  u64 r[RC];
};
template <int RC>
__device__ __forceinline__ u64 rb_get(const BeamR<RC> &r, int e) { // e: wave-uniform index
  u64 v = r.r[0];
#pragma unroll
  for (int c = 1; c < RC; c++) v = (e >> 6) == c ? r.r[c] : v;
  const int l = e & 63;
  const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
  const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
// lane l <- lane l - 1 (v_mov_b32_dpp wave_shr:1), lane 0 <- `lane0`
__device__ __forceinline__ u64 rb_shift_up(u64 v, u64 lane0) {
  const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)lane0, (int)(u32)v, 0x138, 0xf, 0xf, false);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(lane0 >> 32), (int)(u32)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((u64)hi << 32) | lo;
}
// beam_insert on the register beam: identical outcome (same res, pool, tie_bits, dropped)
template <int RC>
__device__ __forceinline__ void beam_insert_rb(Beam &s, BeamR<RC> &r, u64 key, int ef) {
  const int ln = HNY_LANE;
  settle(s);
  const int len = s.res_len;
  int pos = 0;
#pragma unroll
  for (int c = 0; c < RC; c++) pos += __popcll(ballot(ln + 64 * c < len && (r.r[c] & ~1ull) < key));
  const bool evict = (len == ef);
  const u64 oldmax = len ? rb_get<RC>(r, len - 1) : 0ull;
  if (evict && pos == len) { // the new entry is the max: pushed and popped at once
    beam_evicted(s, key, (u32)(oldmax >> 32));
    return;
  }
  if (!evict && len >= s.rcap) {
    s.err = 1;
    return;
  }
  const int hi = evict ? len - 1 : len; // indices (pos, hi] take the entry below them
  u64 up[RC];
#pragma unroll
  for (int c = 0; c < RC; c++) up[c] = rb_shift_up(r.r[c], c ? rb_get<RC>(r, 64 * c - 1) : 0ull);
#pragma unroll
  for (int c = 0; c < RC; c++) {
    const int e = ln + 64 * c;
    r.r[c] = e == pos ? key : ((e > pos && e <= hi) ? up[c] : r.r[c]);
  }
  if (!evict) {
    s.res_len = len + 1;
    return;
  }
  const u32 nd = (u32)(rb_get<RC>(r, len - 1) >> 32);
  if (s.pool_len - s.n_weird > 0 && s.tie_bits != nd) pool_drop_ties(s);
  beam_evicted(s, oldmax, nd);
}

// All accepted neighbours of one expansion merged into the register beam at once.  The outcome of
// the reference's sequence of push / push_pop_max calls (hnsw.rs:508-512) does not depend on their
// order: res ends up as the new_len smallest keys of (res U accepted); an unexpanded entry that
// falls out stays poppable only while its distance ties the final res.max (tie pool) or is "weird",
// and otherwise can never be popped before the walk breaks (`dropped`) — exactly what the
// one-at-a-time beam_insert leaves behind (pool_drop_ties when the max moves on, beam_evicted per
// evicted entry).  One rank loop over the accepted keys (a ballot per key), one scatter through the
// LDS array that the LDS beam would occupy, one classification of what fell out: ~25 instructions
// per accepted key + ~110, against ~170 per key for sequential inserts (2.8 keys per expansion).
template <int RC>
__device__ __forceinline__ void beam_merge_rb(Beam &s, BeamR<RC> &r, bool acc, u64 key, int ef) {
  const int ln = HNY_LANE;
  settle(s);
  const int len = s.res_len;
  u64 m = ballot(acc);
  const int A = __popcll(m);
  u64 kc[RC];
  bool inr[RC];
  int sh[RC]; // accepted keys below my res entry of chunk c
#pragma unroll
  for (int c = 0; c < RC; c++) {
    kc[c] = r.r[c] & ~1ull;
    inr[c] = ln + 64 * c < len;
    sh[c] = 0;
  }
  int mypos = 0; // final index of my key
  while (m) {
    const int i = __ffsll((long long)m) - 1;
    m &= m - 1ull;
    const u64 ki = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(key >> 32), i) << 32) |
                   (u64)(u32)__builtin_amdgcn_readlane((int)(u32)key, i);
    int below = 0;
#pragma unroll
    for (int c = 0; c < RC; c++) {
      const bool l = inr[c] && kc[c] < ki; // keys are distinct (one slot, one key)
      below += __popcll(ballot(l));
      sh[c] += (inr[c] && !l) ? 1 : 0;
    }
    mypos += (acc && ki < key) ? 1 : 0;
    if (ln == i) mypos += below;
  }
  const int total = len + A;
  // push while len != ef, push_pop_max at len == ef; a res that starts above ef (entry points are
  // pushed without a capacity check, :474-481) only grows
  const int new_len = len > ef ? total : (total < ef ? total : ef);
  if (new_len > s.rcap) {
    s.err = 1;
    return;
  }
  u64 *st = s.res;
  int idx[RC];
  u64 old[RC];
#pragma unroll
  for (int c = 0; c < RC; c++) {
    idx[c] = ln + 64 * c + sh[c];
    old[c] = r.r[c];
    if (inr[c] && idx[c] < new_len) st[idx[c]] = r.r[c];
  }
  if (acc && mypos < new_len) st[mypos] = key;
  WSYNC();
#pragma unroll
  for (int c = 0; c < RC; c++) r.r[c] = ln + 64 * c < new_len ? st[ln + 64 * c] : 0ull;
  WSYNC();
  s.res_len = new_len;
  if (total == new_len) return; // nothing fell out
  const u32 nd = (u32)(rb_get<RC>(r, new_len - 1) >> 32);
  if (s.pool_len - s.n_weird > 0 && s.tie_bits != nd) pool_drop_ties(s);
  const u64 lt = (1ull << ln) - 1ull;
#pragma unroll
  for (int round = 0; round <= RC; round++) {
    const bool ev = round < RC ? (inr[round < RC ? round : 0] && idx[round < RC ? round : 0] >= new_len)
                               : (acc && mypos >= new_len);
    const u64 x = round < RC ? old[round < RC ? round : 0] : key;
    const u32 xb = (u32)(x >> 32);
    const bool un = ev && !(x & 1ull); // already popped from `candidates`: nothing to keep
    const bool w = weird_bits(xb);
    const bool keep = un && (w || xb == nd);
    if (ballot(un && !w && xb != nd)) s.dropped = true;
    const u64 pm = ballot(keep);
    if (pm) {
      if (ballot(keep && !w)) s.tie_bits = nd;
      int room = HNY_POOL_CAP - s.pool_len;
      if (room < 0) room = 0;
      const int rank = __popcll(pm & lt);
      const bool put = keep && rank < room;
      if (put) s.pool[s.pool_len + rank] = x & ~1ull;
      const int np = __popcll(pm), nput = np < room ? np : room;
      s.n_weird += __popcll(ballot(put && w));
      s.pool_len += nput;
      s.pool_over += (u32)(np - nput);
      WSYNC();
      settle(s);
    }
  }
}

// The same batched merge for a beam that LIVES in LDS (ef 128..255, C3's efC = 200): its 64-entry
// chunks are pulled into registers for the duration of the merge only — beam_merge_rb scatters the
// merged array back into s.res, which is where this beam lives anyway.  (A register beam kept across
// the whole expansion was measured slower for 4 chunks: its 8 registers are live while the row loads
// of the distance pass need the file.)  One rank pass per accepted key instead of a rank loop plus a
// shift loop with two LDS round trips per 64 entries.
template <int RC>
__device__ __forceinline__ void beam_merge_lds(Beam &s, bool acc, u64 key, int ef) {
  BeamR<RC> r;
#pragma unroll
  for (int c = 0; c < RC; c++) r.r[c] = HNY_LANE + 64 * c < s.res_len ? s.res[HNY_LANE + 64 * c] : 0ull;
  beam_merge_rb<RC>(s, r, acc, key, ef);
}

// visited set of one query (RoaringBitmap `visited`, hnsw.rs:471 / `path`, reader.rs:726).  First
// level: an open-addressing hash table in LDS (a query marks ~1e3 of the N items, so a per-wave
// N-bit set in HBM costs one scattered L2 atomic per neighbour looked at and a dirty 128-B line per
// visited item — ≈10 % of k_walk's traffic at C2 — while the table costs an LDS compare-and-swap).
// Second level: the per-wave HBM bitset + log, used only once the table is 3/4 full (`spill`) and by
// the Reader's exhaustive fallback, which scans for unvisited items (visited_flush).
struct Visited {
  u32 *bits;
  u32 *vlog;
  u32 bits_words, log_cap, log_len;
  int log_over;
  u32 *tab;       // LDS, `slots` entries, HNY_SENT = empty; null: bitset only
  u32 slots, count, limit;
  int spill;      // the table is closed for insertion: new ids go to the bitset
  int in_bits;    // the last visited_insert went to the bitset (its new ids must be logged)
};

__device__ __forceinline__ void visited_init(Visited &v, u32 *bits, u32 bits_words, u32 *vlog, u32 log_cap,
                                             u32 *tab, u32 slots) {
  v.bits = bits;
  v.vlog = vlog;
  v.bits_words = bits_words;
  v.log_cap = log_cap;
  v.log_len = 0;
  v.log_over = false;
  v.tab = slots ? tab : nullptr;
  v.slots = slots;
  v.count = 0;
  v.limit = slots > 256u ? slots / 4u * 3u - 64u : 0u;
  v.spill = slots <= 256u;
  v.in_bits = v.tab == nullptr;
  for (u32 i = HNY_LANE; i < slots; i += 64) tab[i] = HNY_SENT;
  WSYNC();
}

__device__ __forceinline__ void visited_clear(Visited &v) {
  const int ln = HNY_LANE;
  if (v.log_over) {
    for (u32 i = ln; i < v.bits_words; i += 64) v.bits[i] = 0u;
  } else {
    // eight log reads in flight per lane, then their eight stores: `bits` and `vlog` may alias as far as the
    // compiler knows, so the plain loop ran load -> store -> load ..., one memory round trip per 64 entries
    // (~20 of them per walk on short rows).  Measured neutral within the +-3 % run-to-run spread of the
    // short-row walks (C5 0.617 vs 0.631 s over three alternating runs each); kept: never slower.
    for (u32 i0 = 0; i0 < v.log_len; i0 += 512u) {
      u32 w[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u32 i = i0 + (u32)k * 64u + (u32)ln;
        w[k] = i < v.log_len ? v.vlog[i] >> 5 : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (w[k] != 0xFFFFFFFFu) v.bits[w[k]] = 0u;
    }
  }
  v.log_len = 0;
  v.log_over = false;
  if (v.tab) {
    for (u32 i = ln; i < v.slots; i += 64) v.tab[i] = HNY_SENT;
    v.count = 0;
    v.spill = v.slots <= 256u;
  }
  WSYNC();
}

// mark ids (one per lane, `valid` lanes) visited; returns whether this lane's id was new.  Two lanes
// holding the same id: exactly one of them is told "new" (CAS / atomicOr decide which).
__device__ __forceinline__ bool visited_insert(Visited &v, u32 id, bool valid) {
  bool isnew = false, found = false;
  if (v.tab) {
    const bool ins = !v.spill; // wave-uniform
    if (valid) {
      u32 h = (u32)(((u64)(id * 0x9E3779B1u) * (u64)v.slots) >> 32);
      for (;;) {
        const u32 cur = ins ? atomicCAS(&v.tab[h], HNY_SENT, id) : v.tab[h];
        if (cur == HNY_SENT) {
          isnew = ins;
          break;
        }
        if (cur == id) {
          found = true;
          break;
        }
        h = h + 1u == v.slots ? 0u : h + 1u;
      }
    }
    if (ins) {
      v.count += (u32)__popcll(ballot(isnew));
      if (v.count > v.limit) v.spill = true;
      v.in_bits = false;
      return isnew;
    }
  }
  v.in_bits = true;
  if (valid && !found) {
    u32 b = 1u << (id & 31);
    u32 old = atomicOr(&v.bits[id >> 5], b);
    isnew = !(old & b);
  }
  return isnew;
}

__device__ __forceinline__ void visited_log(Visited &v, u32 id, bool isnew, u64 nmask, int rank) {
  if (!v.in_bits) return; // table entries are cleared with the table
  int n_new = __popcll(nmask);
  if (v.log_len + n_new <= v.log_cap) {
    if (isnew) v.vlog[v.log_len + rank] = id;
  } else {
    v.log_over = true;
  }
  v.log_len += n_new;
}

// copy the table into the bitset so that the bitset alone answers "visited?" (the Reader's
// exhaustive fallback scans it word by word); later ids go to the bitset
__device__ __forceinline__ void visited_flush(Visited &v) {
  if (!v.tab) return;
  for (u32 i = HNY_LANE; i < v.slots; i += 64) {
    const u32 id = v.tab[i];
    if (id != HNY_SENT) atomicOr(&v.bits[id >> 5], 1u << (id & 31));
  }
  if (v.count) v.log_over = true; // not logged: clear the whole bitset afterwards
  v.spill = true;
  __threadfence_block();
  WSYNC();
}

__device__ __forceinline__ void settle(Visited &v) {
  v.log_len = uni(v.log_len);
  v.log_over = uni(v.log_over);
  v.count = uni(v.count);
  v.spill = uni(v.spill);
  v.in_bits = uni(v.in_bits);
}

// One walk_layer call (hnsw.rs:460-518).  eps[0..n_eps) and all scratch in LDS.
// QN != NCH: the query is held 8 lanes per row (dist_rows_narrow)
// LMERGE: the LDS beam (RC == 0) takes the accepted keys of an expansion in one batched merge
// PAGED: lists of more than 64 slots (64 < M0 <= 256) are taken 64 at a time — the general kernels;
// the specialised ones serve M0 <= 64 and keep the single pass (the loop's live values cost them 0.6 %)
template <int LPR, int NCH, bool BIG_EPS, int RC = 0, int QN = NCH, bool LMERGE = false, bool PAGED = false> // RC: 64-entry chunks of a register beam, 0 = LDS beam
__device__ __forceinline__ void walk_one_layer(const GraphDev &g, const float4 (&q)[QN], float qn, u32 layer,
                               int ef, const u32 *eps, int n_eps, Beam &s, Visited &vis,
                               u32 *nb_ids, float *nb_d, u64 &evals, u32 &err_iter,
                               const unsigned char *qrow, BeamR<(RC ? RC : 1)> &rb) {
  constexpr bool RB = RC != 0;
  constexpr int RCN = RC ? RC : 1;
  static_assert(!(RB && BIG_EPS), "the register beam holds at most 64 * RC entries");
  const int ln = HNY_LANE;
  s.res_len = 0;
  s.pool_len = 0;
  s.n_weird = 0;
  s.tie_bits = 0;
  s.dropped = false;
  // :474-481 every entry point goes to candidates and res (no capacity check) and is visited
  // (64 at a time: a small index whose items all drew level 0 has every item as an entry point)
  // (BIG_EPS: more than 64 of them, 64 at a time — a small index whose items all drew level 0 has
  // every item as an entry point; a separate instantiation so that the common kernel keeps its
  // register allocation)
  for (int e0 = 0; e0 < (BIG_EPS ? n_eps : 1); e0 += 64) {
    const int ne = BIG_EPS ? (n_eps - e0 < 64 ? n_eps - e0 : 64) : n_eps;
    u32 id = ln < ne ? eps[e0 + ln] : 0u;
    bool isnew = visited_insert(vis, id, ln < ne);
    u64 nmask = ballot(isnew);
    visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
    if (BIG_EPS) WSYNC();
    if (ln < ne) nb_ids[ln] = id;
    WSYNC();
    if constexpr (QN != NCH) dist_rows_narrow<LPR>(g, q, qn, nb_ids, ne, nb_d);
    else dist_rows<LPR, NCH>(g, q, qn, nb_ids, ne, nb_d, qrow);
    evals += (u64)ne;
    WSYNC();
    for (int r = 0; r < ne; r++) {
      u64 key = ((u64)uni(fbits(nb_d[r])) << 32) | ((u64)uni(nb_ids[r]) << 1);
      if constexpr (RB) beam_insert_rb<RCN>(s, rb, key, 0x7FFFFFFF);
      else beam_insert(s, key, 0x7FFFFFFF);
    }
  }
  for (u32 iter = 0;; iter++) {
    settle(s);
    settle(vis);
    evals = uni(evals);
    if (iter > 200000u || s.err) {
      if (iter > 200000u) err_iter = 1;
      break;
    }
#ifdef HNY_PHASE_CLOCKS
    s.ph_t = __builtin_readcyclecounter();
    s.ph[4]++;
#endif
    // ---- candidates.peek()/pop(): smallest distance bits, larger id first among equals
    // (BinaryHeap<(Reverse<OrderedFloat>, ItemId)>, :469, :483-488)
    int first_un = -1, last = -1;
    u32 dmax;
    u64 ta = ~0ull;
    if constexpr (RB) {
      const int len = s.res_len;
      bool un[RCN];
#pragma unroll
      for (int c = RCN - 1; c >= 0; c--) {
        un[c] = ln + 64 * c < len && !(rb.r[c] & 1ull);
        const u64 mk = ballot(un[c]);
        if (mk) first_un = 64 * c + __ffsll((long long)mk) - 1; // the lowest chunk wins (descending loop)
      }
      dmax = (u32)(rb_get<RCN>(rb, len - 1) >> 32);
      last = first_un;
      if (first_un >= 0) { // pop-order key: distance bits ascending, then id DESCENDING
        const u32 d0 = (u32)(rb_get<RCN>(rb, first_un) >> 32);
#pragma unroll
        for (int c = 0; c < RCN; c++) {
          const u64 tk = ballot(un[c] && (u32)(rb.r[c] >> 32) == d0);
          if (tk) last = 64 * c + 63 - __clzll((long long)tk); // the highest chunk wins
        }
        ta = ((u64)d0 << 32) | (u64)(~(u32)(rb_get<RCN>(rb, last) & 0xFFFFFFFEull));
      }
    } else {
    for (int base = 0; base < s.res_len; base += 64) {
      int e = base + ln;
      bool un = e < s.res_len && !(s.res[e] & 1ull);
      u64 mk = ballot(un);
      if (mk) {
        first_un = base + __ffsll((long long)mk) - 1;
        break;
      }
    }
    dmax = uni((u32)(s.res[s.res_len - 1] >> 32));
    // pop-order key: distance bits ascending, then id DESCENDING
    last = first_un;
    if (first_un >= 0) {
      const u32 d0 = uni((u32)(s.res[first_un] >> 32));
      for (int base = first_un & ~63; base < s.res_len; base += 64) {
        int e = base + ln;
        bool ok = e >= first_un && e < s.res_len && (u32)(s.res[e] >> 32) == d0 && !(s.res[e] & 1ull);
        u64 mk = ballot(ok);
        if (mk) last = base + 63 - __clzll((long long)mk);
        int ce = base + 63 < s.res_len - 1 ? base + 63 : s.res_len - 1;
        if (uni((u32)(s.res[ce] >> 32)) != d0) break;
      }
      ta = ((u64)d0 << 32) | (u64)(~uni((u32)(s.res[last] & 0xFFFFFFFEull)));
    }
    }
    u64 tp = ~0ull;
    int pi = -1;
    if (s.pool_len > 0) {
      // the pool's minimum in pop order (distance bits ascending, then id DESCENDING): two entries per lane
      // at most, one DPP butterfly, the owner found by ballot (keys are unique: one slot, one key).  (The
      // entry-by-entry scan with a shuffle argmin that stood here was 250 of the 1 350 instructions of a
      // C5 expansion: Hamming distances tie all the time, the pool is non-empty in 2 of 3 expansions.)
      static_assert(HNY_POOL_CAP == 128, "two pool entries per lane");
      const bool h0 = ln < s.pool_len, h1 = ln + 64 < s.pool_len;
      const u64 k0 = h0 ? s.pool[ln] : 0ull, k1 = h1 ? s.pool[ln + 64] : 0ull;
      const u64 t0 = h0 ? ((k0 & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k0 & 0xFFFFFFFFull))) : ~0ull;
      const u64 t1 = h1 ? ((k1 & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k1 & 0xFFFFFFFFull))) : ~0ull;
      const bool second = h1 && t1 < t0;
      const u64 mine = second ? t1 : t0;
      tp = uni(wave_min_u64(mine));
      const int wl = __ffsll((long long)ballot(h0 && mine == tp)) - 1; // h0: the lane holds an entry at all
      pi = wl + (((ballot(second) >> wl) & 1ull) ? 64 : 0);
    }
    const bool have_a = first_un >= 0, have_p = pi >= 0;
    if (!have_a && !have_p) break; // candidates exhausted (or only dropped entries: they break)
    const bool use_pool = have_p && (!have_a || tp < ta);
    const u32 fb = (u32)((use_pool ? tp : ta) >> 32);
    // a dropped ordinary candidate precedes every weird one in pop order and breaks the walk
    if (use_pool && weird_bits(fb) && s.dropped) break;
    if (__uint_as_float(fb) > __uint_as_float(dmax)) break; // raw f32 compare, :485
    u32 cslot;
    if (use_pool) {
      cslot = (~(u32)(tp & 0xFFFFFFFFull)) >> 1;
      u64 lastk = uni(s.pool[s.pool_len - 1]);
      WSYNC();
      if (ln == 0) s.pool[pi] = lastk;
      s.pool_len--;
      if (weird_bits(fb)) s.n_weird--;
      WSYNC();
    } else if constexpr (RB) {
      cslot = (u32)(rb_get<RCN>(rb, last) >> 1) & 0x7FFFFFFFu;
#pragma unroll
      for (int c = 0; c < RCN; c++) rb.r[c] |= (u64)(ln + 64 * c == last);
    } else {
      cslot = uni((u32)(s.res[last] >> 1) & 0x7FFFFFFFu);
      WSYNC();
      if (ln == 0) s.res[last] |= 1ull;
      WSYNC();
    }
    const float fmax = __uint_as_float(dmax); // f_max captured once per pop (:484)
#ifdef HNY_DEBUG_COUNTS
    if (ln == 0) atomicAdd(&g.stats[9], 1ull);
#endif
    PH_STAMP(s, 0);

    // ---- neighbours of c (:491-495): on-disk Links first (incremental builds, :438-441), then the
    // in-memory list
    for (int pass = g.incremental ? 0 : 1; pass < 2; pass++) {
      u32 cap;
      const u32 *nl = pass == 0 ? disk_ids(g, layer, cslot, cap) : nbr_ids(g, layer, cslot, cap);
      if (!nl) continue;
      // a list of more than 64 slots (64 < M0 <= 256) is taken 64 at a time, in list order: f_max stays
      // the one captured at the pop, the visited set carries over, so the outcome is the sequential one
      for (u32 c0 = 0; c0 < (PAGED ? cap : 1u); c0 += 64u) {
      const u32 pcap = PAGED ? (cap - c0 < 64u ? cap - c0 : 64u) : cap;
      u32 id = (u32)ln < pcap ? nl[(PAGED ? c0 : 0u) + ln] : HNY_SENT;
      bool valid = id != HNY_SENT;
      bool isnew = visited_insert(vis, id, valid);
      u64 nmask = ballot(isnew);
      if (!nmask) {
        PH_STAMP(s, 1);
        continue;
      }
      visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
      if (g.incremental) { // MissingKey => the item was deleted: visited, but never scored (:498-502)
        isnew = isnew && g.has_vec[id] != 0;
        nmask = ballot(isnew);
        if (!nmask) continue;
      }
      if (s.res_len < ef) {
        // duplicates inside one list (add_link never dedups, hnsw.rs:521): while res is not full
        // the order of acceptance matters, so the FIRST occurrence must be the one that counts
        nb_ids[ln] = valid ? id : HNY_SENT;
        WSYNC();
        int firstj = ln;
        bool anynew = isnew;
        for (int j = 0; j < (int)pcap; j++) {
          u32 oj = nb_ids[j];
          if (valid && oj == id) {
            if (j < firstj) firstj = j;
            if ((nmask >> j) & 1ull) anynew = true;
          }
        }
        isnew = valid && anynew && firstj == ln;
        WSYNC();
        nmask = ballot(isnew);
      }
      const int n_new = __popcll(nmask);
      const int rank = __popcll(nmask & ((1ull << ln) - 1ull));
      if (isnew) nb_ids[rank] = id;
      WSYNC();
      PH_STAMP(s, 1);
      if constexpr (QN != NCH) dist_rows_narrow<LPR>(g, q, qn, nb_ids, n_new, nb_d); // :503
      else dist_rows<LPR, NCH>(g, q, qn, nb_ids, n_new, nb_d, qrow);
      evals += (u64)n_new;
      WSYNC();
      PH_STAMP(s, 2);
      const float myd = ln < n_new ? nb_d[ln] : 0.f;
      const u32 myid = ln < n_new ? nb_ids[ln] : 0u;
      int room = ef - s.res_len;
      if (room < 0) room = 0;
      // :505 `res.len() < ef || dist < f_max` — the first `room` new points are taken regardless
      bool acc = ln < n_new && (ln < room || myd < fmax);
      u64 amask = ballot(acc);
      WSYNC();
      if constexpr (RB) {
        if (__popcll(amask) >= HNY_RB_MERGE_MIN && HNY_RB_MERGE) { // one key: the plain insert is cheaper
          beam_merge_rb<RCN>(s, rb, acc, ((u64)fbits(myd) << 32) | ((u64)myid << 1), ef);
          amask = 0ull;
        }
      } else if constexpr (LMERGE) {
        if (__popcll(amask) >= 2 && s.rcap <= 256) { // one key: the plain insert is cheaper (0.610 vs 0.602 s at C3)
          beam_merge_lds<4>(s, acc, ((u64)fbits(myd) << 32) | ((u64)myid << 1), ef);
          amask = 0ull;
        }
      }
#ifdef HNY_DEBUG_COUNTS
      if (ln == 0) { atomicAdd(&g.stats[10], (u64)__popcll(amask)); atomicAdd(&g.stats[11], (u64)(s.res_len < ef ? 1 : 0)); }
#endif
      while (amask) {
        int r = __ffsll((long long)amask) - 1;
        amask &= amask - 1ull;
        u32 db = (u32)__builtin_amdgcn_readlane((int)fbits(myd), r);
        u32 idr = (u32)__builtin_amdgcn_readlane((int)myid, r);
        if constexpr (RB) beam_insert_rb<RCN>(s, rb, ((u64)db << 32) | ((u64)idr << 1), ef);
        else beam_insert(s, ((u64)db << 32) | ((u64)idr << 1), ef);
      }
      PH_STAMP(s, 3);
      } // pages of the list
    }
  }
}

// ---------------------------------------------------------------------------------------------
// walk_layer for SHORT rows (at most 512 B: eight lanes per row), the beam in registers, the visited set
// the per-wave bitset, a fresh build's lists (M0 <= 64) — the specialised kernels' case for 128-d f32 and
// binary codes (BASELINE C4 / C5).  The algorithm is walk_one_layer's, statement for statement; what differs
// is what the instruction stream of one expansion costs.  Round 3 left that kernel at ~480 vector + ~430
// scalar instructions per expansion of ~10 rows of 128 B, i.e. bound by instruction issue, and round 4's
// reading of the compiler's own uniformity analysis (scripts/isa_report.py) showed why: the beam / pool /
// visited counters came out of the expansion loop as "divergent" values, so every `if (pool_len > 0)`, every
// loop exit, was exec-mask arithmetic.  Here
//   * all loop-carried state is pinned wave-uniform (settle()), branches on it are scalar;
//   * the tie pool is only scanned when it can win the pop: its ordinary entries all sit at distance
//     `tie_bits` == res.max, so while the first unexpanded entry of res is closer the pool is not looked at
//     (C5: the pool is non-empty in 2 of 3 expansions and wins almost never);
//   * the distances are consumed where the reduction leaves them — lane 8*sub + 4*j of row 8*j + sub —
//     and the accept test `dist < f_max` (hnsw.rs:505) is applied there: no round trip through LDS, no
//     second compaction; the (<= 4) chunks of 16 rows are interleaved by a DPP row shift so that ONE merge
//     serves the expansion;
//   * lane-conditional LDS traffic is predicated by address (a dump slot) instead of by exec mask.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Visited set of a SHORT-row walk: a bucketized table of 16-bit remainders in LDS.
// What bounds the short-row walk is not bytes and not instruction issue but the number of RETURNING
// GLOBAL ATOMICS: one atomicOr on the wave's HBM bitset per neighbour looked at — 10.7 G of them in a C5
// build (21 per expansion), and one MI355X sustains ~18 G random returning atomics per second on bitsets
// of this size whatever the number of waves (scripts/micro/random_access_roof.hip; plain random loads:
// ~55 G/s): 10.7 / 18 = the 0.59 s that the walk takes.  An open-addressing table of full 32-bit ids
// large enough for the ~1 150 ids a walk marks costs the LDS that occupancy needs (round 2: slower at
// every size).  So: the dense slot id goes through a bijection of [0, 2^k) (multiplication by an odd
// constant mod 2^k >= n), the result splits into bucket = sid mod nb and remainder = sid div nb < 65 535,
// and the bucket — ONE 8-byte LDS word — holds up to four remainders + 1 (0 = empty).  Bucket and remainder
// identify the id exactly; a lookup is one ds_read_b64, an insertion one 64-bit compare-and-swap on the
// bucket (a lane that loses the swap to a neighbour of the same bucket looks again).  640 buckets = 5 KB
// per wave keep 20 waves per CU resident and hold 2 560 ids; an id whose bucket is full (a few per cent of
// the insertions at ~1 150 ids) lives in the HBM bitset as before — membership is decided by "bucket full",
// which never reverts during a walk, so the two levels cannot disagree.
// ---------------------------------------------------------------------------------------------
struct VisB {
  u64 *tb;    // LDS: nb buckets x 4 x u16; null = bitset only
  u32 nb;     // buckets
  u32 magic;  // floor(2^shift / nb) + 1, shift = 31 + floor(log2 nb): sid div nb == (sid * magic) >> shift for sid < 2^30
  u32 shift;
  u32 smask;  // 2^k - 1
  int off;    // the table was flushed into the bitset (Reader's exhaustive fallback): the bitset alone answers
};
#define HNY_VISB_MUL 0x9E3779B1u
#define HNY_VISB_INV 0x0E8B2F51u // HNY_VISB_MUL * HNY_VISB_INV == 1 mod 2^32 (hence mod every 2^k)

__device__ __forceinline__ void visb_clear(VisB &v) {
  if (!v.tb) return;
  float4 *t4 = reinterpret_cast<float4 *>(v.tb);
  for (u32 i = HNY_LANE; i < v.nb / 2u; i += 64) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  v.off = 0;
  WSYNC();
}

// copy the table into the bitset so that the bitset alone answers "visited?" (the Reader's exhaustive fallback
// scans it word by word, reader.rs:771-795); later ids go to the bitset.  bucket and remainder give the id back.
__device__ __forceinline__ void visb_flush(VisB &v, Visited &vis) {
  if (!v.tb || v.off) return;
  for (u32 b = HNY_LANE; b < v.nb; b += 64) {
    const u64 cur = v.tb[b];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u32 r = (u32)(cur >> (16 * j)) & 0xFFFFu;
      if (r) {
        const u32 id = (((r - 1u) * v.nb + b) * HNY_VISB_INV) & v.smask;
        atomicOr(&vis.bits[id >> 5], 1u << (id & 31));
      }
    }
  }
  vis.log_over = 1; // not logged: clear the whole bitset afterwards
  v.off = 1;
  __threadfence_block();
  WSYNC();
}

// mark ids (one per lane, `valid` lanes) visited; returns whether this lane's id was new (of two lanes with
// the same id exactly one is told so).  `vis` = the bitset level (overflow), logged for its clear.
__device__ __forceinline__ bool visb_insert(const VisB &v, Visited &vis, u32 id, bool valid) {
  const u64 lt = (1ull << HNY_LANE) - 1ull;
  bool isnew = false, over = valid;
  if (v.tb && !v.off) { // wave-uniform
    const u32 sid = (id * HNY_VISB_MUL) & v.smask;
    const u32 q = (u32)(((u64)sid * (u64)v.magic) >> v.shift);
    const u32 bk = sid - q * v.nb;
    const u32 r = q + 1u, r2 = r | (r << 16); // the remainder (+ 1) in both halves of a word
    uint2 cur = reinterpret_cast<const uint2 *>(v.tb)[bk]; // (lanes without an id read some bucket too: no exec juggling)
    bool pending = valid;
    over = false;
    while (ballot(pending)) {
      // a 16-bit half of `cur` equal to r: the zero-half test of (cur ^ r2)
      const u32 x0 = cur.x ^ r2, x1 = cur.y ^ r2;
      const bool found = ((((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1)) & 0x80008000u) != 0u;
      // halves fill up in order and are never emptied: the first empty one
      const bool e0 = cur.x == 0u, e1 = (cur.x >> 16) == 0u, e2 = cur.y == 0u, e3 = (cur.y >> 16) == 0u;
      if (pending && !found && e3) {
        const u32 add = (e0 || (e2 && !e1)) ? r : (r << 16);
        uint2 want;
        want.x = cur.x | (e1 ? add : 0u);
        want.y = cur.y | (e1 ? 0u : add);
        const u64 c64 = ((u64)cur.y << 32) | cur.x, w64 = ((u64)want.y << 32) | want.x;
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&v.tb[bk]), (unsigned long long)c64,
                                  (unsigned long long)w64);
        if (old == c64) {
          isnew = true;
          pending = false;
        } else { // a neighbour of the same bucket got there first: look again
          cur.x = (u32)old;
          cur.y = (u32)(old >> 32);
        }
      } else if (pending) {
        over = !found; // full and not in it: the bitset decides
        pending = false;
      }
    }
  }
  const u64 om = ballot(over);
  if (om) { // wave-uniform; without a table: every valid lane
    bool onew = false;
    if (over) {
      const u32 b = 1u << (id & 31);
      onew = !(atomicOr(&vis.bits[id >> 5], b) & b);
    }
    const u64 nm = ballot(onew);
    const int n_log = __popcll(nm);
    if (n_log) {
      if (vis.log_len + (u32)n_log <= vis.log_cap) {
        if (onew) vis.vlog[vis.log_len + (u32)__popcll(nm & lt)] = id;
      } else {
        vis.log_over = true;
      }
      vis.log_len += (u32)n_log;
    }
    isnew = isnew || onew;
  }
  return isnew;
}

// distances of rows ids[k0 .. k0 + 16) (as far as they are < n) to the query: the lane with (t & 3) == 0 of
// lane group `sub` returns row ri = k0 + 8 * (t >> 2) + sub (wave order of an LPRO-lane row group, computed
// by 8 lanes exactly as dist_rows_narrow does; LPRO == 8 is the plain 8-lane butterfly)
template <int LPRO>
__device__ __forceinline__ void dist16(const GraphDev &g, const float4 (&q)[LPRO / 8], float qn, const u32 *ids,
                                       int n, int k0, float &d, u32 &rid, int &ri) {
  constexpr int NQ = LPRO / 8;
  const int ln = HNY_LANE, t = ln & 7, sub = ln >> 3;
  const int j = (t >> 2) & 1; // fold2_row<8>()
  float4 r[2][NQ];
  float rn[2];
  u32 rids[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    int x = k0 + u * 8 + sub;
    x = x < n - 1 ? x : n - 1;
    rids[u] = ids[x];
  }
  const bool two = k0 + 8 < n; // wave-uniform: the second load group holds rows
#pragma unroll
  for (int u = 0; u < 2; u++) {
    rn[u] = 0.f;
    if (u == 0 || two) {
      const unsigned char *p = g.rows + (size_t)rids[u] * g.row_stride;
#pragma unroll
      for (int c = 0; c < NQ; c++) {
        const u32 f = (u32)(c * 8 + t);
        r[u][c] = f < g.n16 ? *reinterpret_cast<const float4 *>(p + (size_t)f * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (g.norms) rn[u] = g.norms[rids[u]];
    } else {
#pragma unroll
      for (int c = 0; c < NQ; c++) r[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // fold2 on both groups always: for the rows of group 0 its pair sums are the butterfly's (a second group
  // of zeros only feeds the lanes of rows that do not exist)
  if (g.mclass == MC_BIN) {
    const u32 y = fold2<8, u32>(partial_bin<NQ>(q, r[0]), partial_bin<NQ>(q, r[1]));
    d = finalize_bin(g, y, qn, j ? rn[1] : rn[0]);
  } else {
    float pa[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      float pu[NQ];
#pragma unroll
      for (int c = 0; c < NQ; c++) {
        const float4 qc[1] = {q[c]}, rc[1] = {r[u][c]};
        pu[c] = partial_f32<1>(g.mclass, qc, rc);
      }
      if constexpr (NQ == 4) pa[u] = (pu[0] + pu[2]) + (pu[1] + pu[3]); // off = 16, then off = 8
      else if constexpr (NQ == 2) pa[u] = pu[0] + pu[1];                // off = 8
      else pa[u] = pu[0];
    }
    d = finalize_f32(g, fold2<8, float>(pa[0], pa[1]), qn, j ? rn[1] : rn[0]);
  }
  rid = j ? rids[1] : rids[0];
  ri = k0 + j * 8 + sub;
}

// value of lane - K within its row of 16 lanes (v_mov_b32_dpp row_shr:K); lanes without a source get 0
template <int K>
__device__ __forceinline__ u32 row_shr(u32 v) {
  if constexpr (K == 0) return v;
  else return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xF, 0xF, false);
}

template <int LPRO, int RC>
__device__ __forceinline__ void walk_layer_short(const GraphDev &g, const float4 (&q)[LPRO / 8], float qn, u32 layer,
                                                 int ef, const u32 *eps, int n_eps, Beam &s, Visited &vis,
                                                 const VisB &vb, u32 *nb_ids, float *nb_d, u64 &evals, u32 &err_iter,
                                                 BeamR<RC> &rb) {
  const int ln = HNY_LANE, t = ln & 7;
  const u64 lt = (1ull << ln) - 1ull;
  u32 *dump = reinterpret_cast<u32 *>(nb_d); // 64 words behind nb_ids: where the lanes without a new id write
  s.res_len = 0;
  s.pool_len = 0;
  s.n_weird = 0;
  s.tie_bits = 0;
  s.dropped = false;
  // this layer's lists (nbr_ids, resolved once per walk instead of once per expansion: the expansion loop then
  // carries base / stride / cap / index table instead of l0_ids, M0, up_ids, up_layers, M, upper_idx and layer)
  const u32 *const lst_base = layer == 0 ? g.l0_ids : g.up_ids + (size_t)(layer - 1u) * g.M;
  const u32 lst_stride = layer == 0 ? g.M0 : g.up_layers * g.M;
  const u32 lst_cap = layer == 0 ? g.M0 : g.M;
  const int *const lst_idx = layer == 0 ? nullptr : g.upper_idx;
  // :474-481 every entry point goes to candidates and res (no capacity check) and is visited
  {
    const int ne = n_eps; // <= 64 here
    u32 id = eps[ln < ne ? ln : 0];
    (void)visb_insert(vb, vis, id, ln < ne);
    nb_ids[ln] = id;
    WSYNC();
    for (int k0 = 0; k0 < ne; k0 += 16) {
      float d;
      u32 rid;
      int ri;
      dist16<LPRO>(g, q, qn, nb_ids, ne, k0, d, rid, ri);
      if ((t & 3) == 0 && ri < ne) dump[ri] = fbits(d);
    }
    evals += (u64)ne;
    WSYNC();
    for (int r = 0; r < ne; r++) {
      const u64 key = ((u64)uni(dump[r]) << 32) | ((u64)uni(nb_ids[r]) << 1);
      beam_insert_rb<RC>(s, rb, key, 0x7FFFFFFF);
    }
    WSYNC();
  }
  for (u32 iter = 0;; iter++) {
    settle(s);
    settle(vis);
    evals = uni(evals);
    if (iter > 200000u || s.err) {
      if (iter > 200000u) err_iter = 1;
      break;
    }
#ifdef HNY_PHASE_CLOCKS
    s.ph_t = __builtin_readcyclecounter();
    s.ph[4]++;
#endif
    // ---- candidates.peek()/pop(): smallest distance bits, larger id first among equals
    // (BinaryHeap<(Reverse<OrderedFloat>, ItemId)>, :469, :483-488)
    const int len = s.res_len;
    // (RC == 1: a beam of at most 64 entries, ef <= 64 — every "second chunk" term below folds away)
    const bool un0 = ln < len && !(rb.r[0] & 1ull), un1 = RC > 1 && ln + 64 < len && !(rb.r[RC - 1] & 1ull);
    const u64 m0 = ballot(un0), m1 = RC > 1 ? ballot(un1) : 0ull;
    const u32 dmax = (u32)(rb_get<RC>(rb, len - 1) >> 32);
    const bool have_a = (m0 | m1) != 0ull;
    int last = -1;
    u32 d0 = 0u;
    u64 ta = ~0ull;
    if (have_a) { // pop-order key: distance bits ascending, then id DESCENDING
      const int first_un = m0 ? __ffsll((long long)m0) - 1 : 64 + __ffsll((long long)m1) - 1;
      d0 = (u32)(rb_get<RC>(rb, first_un) >> 32);
      const u64 t0 = ballot(un0 && (u32)(rb.r[0] >> 32) == d0),
                t1 = RC > 1 ? ballot(un1 && (u32)(rb.r[RC - 1] >> 32) == d0) : 0ull;
      last = t1 ? 64 + 63 - __clzll((long long)t1) : 63 - __clzll((long long)t0);
      ta = ((u64)d0 << 32) | (u64)(~(u32)(rb_get<RC>(rb, last) & 0xFFFFFFFEull));
    }
    u64 tp = ~0ull;
    int pi = -1;
    // The ordinary pool entries all carry the distance bits `tie_bits` == those of res.max (beam_evicted,
    // pool_drop_ties), so they precede the first unexpanded entry of res in pop order only if that one is
    // not closer; weird entries (sign bit / NaN) sort behind every ordinary distance.
    if (s.pool_len > 0 && (!have_a || s.n_weird > 0 || d0 >= s.tie_bits)) {
      static_assert(HNY_POOL_CAP == 128, "two pool entries per lane");
#ifdef HNY_PHASE_CLOCKS
      s.ph[9]++;
#endif
      const bool h0 = ln < s.pool_len, h1 = ln + 64 < s.pool_len;
      const u64 k0 = s.pool[ln], k1 = s.pool[ln + 64];
      const u64 p0 = h0 ? ((k0 & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k0 & 0xFFFFFFFFull))) : ~0ull;
      const u64 p1 = h1 ? ((k1 & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k1 & 0xFFFFFFFFull))) : ~0ull;
      const bool second = h1 && p1 < p0;
      const u64 mine = second ? p1 : p0;
      tp = uni(wave_min_u64(mine));
      const int wl = __ffsll((long long)ballot(h0 && mine == tp)) - 1; // h0: the lane holds an entry at all
      pi = wl + (((ballot(second) >> wl) & 1ull) ? 64 : 0);
    }
    const bool have_p = pi >= 0;
    if (!have_a && !have_p) break; // candidates exhausted (or only dropped entries: they break)
    const bool use_pool = have_p && (!have_a || tp < ta);
    const u32 fb = (u32)((use_pool ? tp : ta) >> 32);
    // a dropped ordinary candidate precedes every weird one in pop order and breaks the walk
    if (use_pool && weird_bits(fb) && s.dropped) break;
    if (__uint_as_float(fb) > __uint_as_float(dmax)) break; // raw f32 compare, :485
    u32 cslot;
    if (use_pool) {
      cslot = (~(u32)(tp & 0xFFFFFFFFull)) >> 1;
      const u64 lastk = uni(s.pool[s.pool_len - 1]);
      WSYNC();
      if (ln == 0) s.pool[pi] = lastk;
      s.pool_len--;
      if (weird_bits(fb)) s.n_weird--;
      WSYNC();
      settle(s);
    } else {
      cslot = (u32)(rb_get<RC>(rb, last) >> 1) & 0x7FFFFFFFu;
      rb.r[0] |= (u64)(ln == last);
      if constexpr (RC > 1) rb.r[1] |= (u64)(ln + 64 == last);
    }
    const float fmax = __uint_as_float(dmax); // f_max captured once per pop (:484)
    PH_STAMP(s, 0);

    // ---- neighbours of c (:491-495): the in-memory list (fresh build), one lane per slot
    const u32 cap = lst_cap;
    const u32 node = lst_idx ? (u32)lst_idx[cslot] : cslot;
    const u32 *nl = lst_base + (size_t)node * lst_stride;
    u32 id = nl[(u32)ln < cap ? (u32)ln : cap - 1u];
    const bool valid = (u32)ln < cap && id != HNY_SENT;
#ifdef HNY_PHASE_CLOCKS
    s.ph[6] += (u64)__popcll(ballot(valid)); // (the ballot needs the ids: the list fetch ends here)
    PH_STAMP(s, 1);
#endif
    bool isnew = visb_insert(vb, vis, id, valid);
    u64 nmask = ballot(isnew);
    if (!nmask) {
#ifdef HNY_PHASE_CLOCKS
      s.ph[10]++;
#endif
      PH_STAMP(s, 5);
      continue;
    }
    if (len < ef) {
      // duplicates inside one list (add_link never dedups, hnsw.rs:521): while res is not full
      // the order of acceptance matters, so the FIRST occurrence must be the one that counts
      nb_ids[ln] = valid ? id : HNY_SENT;
      WSYNC();
      int firstj = ln;
      bool anynew = isnew;
      for (int j = 0; j < (int)cap; j++) {
        const u32 oj = nb_ids[j];
        if (valid && oj == id) {
          if (j < firstj) firstj = j;
          if ((nmask >> j) & 1ull) anynew = true;
        }
      }
      isnew = valid && anynew && firstj == ln;
      WSYNC();
      nmask = uni64(ballot(isnew));
    }
    const int n_new = __popcll(nmask);
    nb_ids[isnew ? __popcll(nmask & lt) : 64 + ln] = id; // (64 + ln: the dump words)
    WSYNC();
    PH_STAMP(s, 5);
    // ---- :503-512 score the new points, keep those that enter res
    int room = ef - len;
    if (room < 0) room = 0;
    u32 acc = 0u, khi = 0u, klo = 0u;
#define HNY_SHORT_CHUNK(K)                                                                   \
    if (16 * K < n_new) {                                                                    \
      float d;                                                                               \
      u32 rid;                                                                               \
      int ri;                                                                                \
      dist16<LPRO>(g, q, qn, nb_ids, n_new, 16 * K, d, rid, ri);                             \
      /* :505 `res.len() < ef || dist < f_max` — the first `room` new points regardless */   \
      const bool a = (t & 3) == 0 && ri < n_new && (ri < room || d < fmax);                  \
      const u32 ak = row_shr<K>(a ? 1u : 0u), hk = row_shr<K>(fbits(d)), lk = row_shr<K>(rid << 1); \
      if ((t & 3) == K) {                                                                    \
        acc = ak;                                                                            \
        khi = hk;                                                                            \
        klo = lk;                                                                            \
      }                                                                                      \
    }
    HNY_SHORT_CHUNK(0)
    HNY_SHORT_CHUNK(1)
    HNY_SHORT_CHUNK(2)
    HNY_SHORT_CHUNK(3)
#undef HNY_SHORT_CHUNK
    evals += (u64)n_new;
    PH_STAMP(s, 2);
    u64 amask = ballot(acc != 0u);
    if (!amask) continue;
#ifdef HNY_PHASE_CLOCKS
    s.ph[7] += (u64)__popcll(amask);
    s.ph[8]++;
#endif
    if (__popcll(amask) >= HNY_RB_MERGE_MIN && HNY_RB_MERGE) { // one key: the plain insert is cheaper
      beam_merge_rb<RC>(s, rb, acc != 0u, ((u64)khi << 32) | (u64)klo, ef);
      amask = 0ull;
    }
    while (amask) {
      const int r = __ffsll((long long)amask) - 1;
      amask &= amask - 1ull;
      const u32 db = (u32)__builtin_amdgcn_readlane((int)khi, r);
      const u32 idr = (u32)__builtin_amdgcn_readlane((int)klo, r);
      beam_insert_rb<RC>(s, rb, ((u64)db << 32) | (u64)idr, ef);
    }
    PH_STAMP(s, 3);
  }
}


#ifndef HNY_WALK_WPE
#define HNY_WALK_WPE 4
#endif
// binary codes of at most 1 KB (NCH == 1): the specialised BUILD kernels run 6 waves per SIMD in 79 VGPRs (two
// spilled) — the walk on 128-B codes is bound by instruction issue and latency, and with the visited set in LDS
// the sixth wave pays (C5 walk 0.401 -> 0.377 s with 6 144 resident waves and 448 buckets; 7 waves: 0.382, 9
// spilled VGPRs and too little LDS left for the table).  Only walk_layer_short was shown to fit (codes of at
// most 512 B, register beam): codes of 513 - 1 024 B and the LDS-beam variant (ef >= 128) run walk_one_layer,
// measured at ~96 VGPRs, and keep 4 waves.  The Reader's variant carries the exhaustive fallback
// and spills at 96: 4 waves; the f32 kernels of that row size need 128 VGPRs; the general kernels spill at 4.
#ifndef HNY_WALK_WPE_SMALL
#define HNY_WALK_WPE_SMALL 6
#endif
template <int LPR, int NCH, bool BIG_EPS, int SP, bool RM = false, int RC = 0>
__global__ __launch_bounds__(64, (NCH == 1 && SP >= 4 && !RM && (RC == 1 || RC == 2) && LPR <= 32 ? HNY_WALK_WPE_SMALL : HNY_WALK_WPE)) void k_walk(GraphDev g_in, WalkArgs a_in) {
  constexpr bool RB = RC != 0;
  constexpr int RCN = RC ? RC : 1;
  static_assert(!(RB && (BIG_EPS || SP == 0)), "register beam: specialised kernels only");
  // rows of 9..32 units in the specialised kernels: 8 lanes per row (dist_rows_narrow)
  constexpr bool NARROW = SP != 0 && NCH == 1 && (LPR == 16 || LPR == 32);
  constexpr int QN = NARROW ? LPR / 8 : NCH;
  constexpr bool LMERGE = SP != 0 && RC == 0 && !BIG_EPS; // specialised kernels with the beam in LDS
  constexpr bool PAGED = SP == 0;                          // general kernels: M0 up to HNY_BIG_CAP
  // rows of at most 32 units with the beam in registers: walk_layer_short (HNY_NO_SHORT_WALK: walk_one_layer)
#ifdef HNY_NO_SHORT_WALK
  constexpr bool SHORT = false;
#else
  constexpr bool SHORT = SP != 0 && (RC == 1 || RC == 2) && NCH == 1 && LPR <= 32;
#endif
  GraphDev g = g_in;
  specialize<SP>(g);
  // The ~70 scalars of WalkArgs are read where they are used, from the kernarg segment, through a pointer the
  // compiler cannot see through (KA_FRESH: an empty asm that redefines it).  As a by-value copy they were all live
  // across the expansion loop — which uses a handful of them — and the register allocator parked 106 SGPRs in the
  // lanes of two VGPRs (v_writelane / v_readlane: ~20 reloads per expansion on the vector port, and two VGPRs of
  // an 80-VGPR budget).  After a KA_FRESH a field (ak->field) is an s_load from the scalar cache at its next use.
  typedef const WalkArgs __attribute__((address_space(4))) *KArgs;
  KArgs ak = (KArgs)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() +
                     ((sizeof(GraphDev) + alignof(WalkArgs) - 1) / alignof(WalkArgs)) * alignof(WalkArgs));
  (void)a_in;
#define KA_FRESH() asm volatile("" : "+s"(ak))
  const int reader_mode = SP != 0 ? (RM ? 1 : 0) : ak->reader_mode; // RM: the Reader's search (hny_builder_search_knn)
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *res = reinterpret_cast<u64 *>(smem);
  u64 *pool = res + ak->rcap;
  if constexpr (SP == 0) {
    // a result set of more than 4 096 entries (a walk that never evicts, res_capacity in hny_host.cpp)
    // lives in HBM: the beam code only sees a pointer, WSYNC fences every address space
    if (ak->res_global) {
      pool = res;
      res = ak->res_global + (size_t)blockIdx.x * ak->rcap;
    }
  }
  u32 *nb_ids = reinterpret_cast<u32 *>(pool + HNY_POOL_CAP);
  float *nb_d = reinterpret_cast<float *>(nb_ids + 64);
  u32 *eps = reinterpret_cast<u32 *>(nb_d + 64);
  const int ln = threadIdx.x, t = ln % LPR;

  Beam s;
#ifdef HNY_PHASE_CLOCKS
  for (int i = 0; i < 12; i++) s.ph[i] = 0;
  s.ph_t = 0;
  const u64 ph_kernel_t0 = __builtin_readcyclecounter();
#endif
  s.res = res;
  s.pool = pool;
  s.rcap = (int)ak->rcap;
  s.pool_over = 0;
  s.err = 0;
  s.res_len = 0;
  s.pool_len = 0;
  s.n_weird = 0;
  s.tie_bits = 0;
  s.dropped = false;
  BeamR<RCN> rb;
#pragma unroll
  for (int c = 0; c < RCN; c++) rb.r[c] = 0ull;
  if constexpr (RB) s.rcap = s.rcap < 64 * RCN ? s.rcap : 64 * RCN; // what the register beam can hold
  Visited vis;
  // (for short rows, which have no LDS table, the same table in GLOBAL memory — L2 / Infinity-Cache
  // resident, 16-32 KB per wave — in front of the bitset was measured too: C5 walk 0.71 s against
  // 0.69-0.75, C4-like 0.88 against 0.88: no gain, DESIGN.md §5 "Short rows")
  // rows <= 1 KB never get an LDS table (the host passes vis_slots = 0 for them): say so at compile
  // time in the specialised kernels, so that the table's code and its six wave-uniform fields go
  constexpr bool NO_TAB = SP != 0 && NCH == 1;
  visited_init(vis, ak->bits + (size_t)blockIdx.x * ak->bits_words, ak->bits_words,
               ak->vlog + (size_t)blockIdx.x * ak->log_cap, ak->log_cap, eps + (BIG_EPS ? ak->eps_cap : 64u),
               NO_TAB ? 0u : ak->vis_slots);
  VisB vb;
  vb.tb = nullptr;
  vb.nb = 0;
  vb.magic = 0;
  vb.shift = 0;
  vb.smask = 0;
  vb.off = 0;
  if constexpr (SHORT) {
    if (ak->vis_buckets) { // (behind eps; NO_TAB kernels have no other table there)
      vb.tb = reinterpret_cast<u64 *>(eps + 64);
      vb.nb = ak->vis_buckets;
      vb.magic = ak->vis_magic;
      vb.shift = ak->vis_shift;
      vb.smask = ak->vis_smask;
      visb_clear(vb);
    }
  }
  u64 evals = 0;
  u32 err_iter = 0, log_over_cnt = 0;

  // dynamic work queue: queries differ a lot in length, a static stride leaves a long launch tail
  u32 xq_dead = 0; // XCD-tiled queue: how many of the 8 counters this wave has found exhausted
  for (;;) {
    KA_FRESH();
    u32 m = 0;
    if (ln == 0) {
      if (SP != 0 && ak->xcd_tile) {
        // Tiles of xcd_tile consecutive members (locality order) go round-robin to the 8 XCDs, each
        // with its own counter: the waves of one XCD (= one L2) work on neighbouring queries while all
        // eight stay inside the same window of the batch (= one Infinity-Cache footprint).  A wave
        // whose XCD has run out helps the next one.
        const u32 T = ak->xcd_tile, cnt_all = ak->hi - ak->lo;
        m = 0xFFFFFFFFu;
        while (xq_dead < 8u) {
          const u32 x = (blockIdx.x + xq_dead) & 7u;
          const u32 c = atomicAdd(ak->queue + x, 1u);
          const u32 idx = ((c / T) * 8u + x) * T + (c % T);
          if (idx < cnt_all) {
            m = ak->lo + idx;
            break;
          }
          xq_dead++;
        }
      } else {
        m = ak->lo + atomicAdd(ak->queue, 1u);
      }
      if constexpr (RM || SP == 0) // the Visitor's cancel probe (reader.rs:333), between queries
        if (ak->cancel && __hip_atomic_load(ak->cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) m = 0xFFFFFFFFu;
    }
    m = uni(m);
    if (m >= ak->hi) break;
    if (ak->perm) m = uni((u32)ak->perm[m - ak->lo]); // locality order; results stay indexed by member
    const u64 evals_before = evals;
    const unsigned char *qrow;
    float qn = 0.f;
    if (ak->q_rows) {
      qrow = ak->q_rows + (size_t)m * ak->q_stride;
      if (ak->q_norms) qn = ak->q_norms[m];
    } else {
      u32 qslot = ak->q_slots[m];
      qrow = g.rows + (size_t)qslot * g.row_stride;
      if (g.norms) qn = g.norms[qslot];
    }
    float4 q[QN];
    if constexpr (NARROW) {
#pragma unroll
      for (int c = 0; c < QN; c++) {
        const u32 f = (u32)(c * 8 + (ln & 7));
        q[c] = f < g.n16 ? *reinterpret_cast<const float4 *>(qrow + (size_t)f * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      load_row<LPR, NCH>(qrow, t, g.n16, q);
    }

    int n_eps;
    u32 start_layer;
    if (ak->first) { // :298 eps = all entry points
      n_eps = (int)ak->n_entry_points;
      if (BIG_EPS) {
        for (int i = ln; i < n_eps; i += 64) eps[i] = ak->entry_points[i];
      } else if (ln < n_eps) {
        eps[ln] = ak->entry_points[ln];
      }
      start_layer = g.max_level;
    } else if (ak->eps_in) { // resume after a descend_only launch
      n_eps = 1;
      if (ln == 0) eps[0] = ak->eps_in[m];
      start_layer = ak->layer;
    } else { // :316-321 eps = what was selected on the layer above
      const u64 *sl = ak->sel + (size_t)m * ak->sel_stride +
                      (size_t)(ak->batch_level - (ak->layer + 1)) * (ak->cap_sel + 1);
      n_eps = (int)sl[0];
      for (int i = ln; i < n_eps; i += 64) eps[i] = (u32)(sl[1 + i] & 0xFFFFFFFFull); // (more than 64: M > 64)
      start_layer = ak->layer;
    }
    // the lane-conditional stores above sit on the paths that define these two: without the readfirstlane the
    // compiler's uniformity analysis calls them divergent, and with them the layer loop's exit, `ef`, and
    // every piece of beam / pool / visited state that the expansion loop carries
    n_eps = uni(n_eps);
    start_layer = uni(start_layer);
    WSYNC();
    u64 lkey = 0;
    for (u32 layer = start_layer;; layer--) {
      const bool last = (layer == ak->layer);
      if (last && ak->descend_only) break;
      if constexpr (SHORT)
        walk_layer_short<LPR, RCN>(g, q, qn, layer, last ? (int)ak->ef : 1, eps, n_eps, s, vis, vb, nb_ids, nb_d, evals, err_iter, rb);
      else
        walk_one_layer<LPR, NCH, BIG_EPS, RC, QN, LMERGE, PAGED>(g, q, qn, layer, last ? (int)ak->ef : 1, eps, n_eps, s, vis, nb_ids,
                                     nb_d, evals, err_iter, qrow, rb);
      KA_FRESH();
      if (last) break;
      // :305-306 eps = [closest]
      const u32 closest = RB ? (u32)(rb_get<RCN>(rb, 0) >> 1) & 0x7FFFFFFFu : uni((u32)(s.res[0] >> 1) & 0x7FFFFFFFu);
      if (ln == 0) eps[0] = closest;
      n_eps = 1;
      // locality key: the closest node of the last three greedy layers, coarse to fine
      lkey = (lkey << 16) | (u64)((u32)g.upper_idx[closest] & 0xFFFFu);
      // walk_layer owns a fresh visited set; Reader::hnsw_search shares `path` across the greedy
      // layers and clears it once before layer 0 (reader.rs:731-743)
      if (!reader_mode || layer == ak->layer + 1) {
        if (vis.log_over) log_over_cnt++;
        visited_clear(vis);
        visb_clear(vb);
      }
      WSYNC();
    }
    // the tie pool overflowed somewhere in this member's walks: what was computed is not the reference's
    // result.  Build walks hand the member to k_walk_heap (same launch arguments, heaps in HBM); without a
    // retry list (searches) the overflow is counted and fails the call.
    if constexpr (!RM) {
      if (ak->force_pool && m % ak->force_pool == 0u) s.pool_over = 1u;
      if (s.pool_over && ak->pool_retry) {
        if (ln == 0) ak->pool_retry[atomicAdd(ak->n_pool_retry, 1u)] = m;
        s.pool_over = 0;
        evals = evals_before; // the heap kernel counts this member's evaluations (the reference's number)
      }
    }
    if (ak->descend_only) { // (a batch whose level equals max_level has no greedy layer: eps stay)
      if (ln == 0) {
        ak->eps_out[m] = eps[0];
        ak->key_out[m - ak->key_base] = lkey & 0xFFFFFFFFFFFFull;
      }
      WSYNC();
      continue;
    }
    // result, ascending (res.into_vec() is re-sorted by robust_prune anyway, :573)
    if constexpr (RB) {
#pragma unroll
      for (int c = 0; c < RCN; c++)
        if (ln + 64 * c < s.res_len)
          ak->cand[(size_t)m * ak->rcap + 64 * c + ln] = (rb.r[c] & 0xFFFFFFFF00000000ull) | ((rb.r[c] >> 1) & 0x7FFFFFFFull);
    } else {
      for (int e = ln; e < s.res_len; e += 64) {
        u64 k = s.res[e];
        ak->cand[(size_t)m * ak->rcap + e] = (k & 0xFFFFFFFF00000000ull) | ((k >> 1) & 0x7FFFFFFFull);
      }
    }
    int total = s.res_len;
    if (reader_mode && total < (int)ak->knn_k) {
      // Reader::hnsw_search exhaustive fallback (reader.rs:771-795): the walk got trapped in a
      // sub-graph with fewer than k items; restart from every item not seen yet (ascending id),
      // sharing the visited set, until opt.ef hits are collected.  Rare; written for clarity.
      const u32 nwords = (g.n + 31) >> 5;
      u32 pos = 0;
      visited_flush(vis);
      visb_flush(vb, vis);
      while (pos < g.n) {
        const u32 wbase = pos >> 5;
        const u32 widx = wbase + (u32)ln;
        u32 unv = 0u;
        if (widx < nwords) {
          unv = ~__hip_atomic_load(&vis.bits[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ln == 0 && (pos & 31u)) unv &= ~((1u << (pos & 31u)) - 1u);
          if (widx == nwords - 1 && (g.n & 31u)) unv &= (1u << (g.n & 31u)) - 1u;
        }
        const u64 mk = ballot(unv != 0u);
        if (!mk) {
          pos = (wbase + 64u) << 5;
          continue;
        }
        const int l0 = __ffsll((long long)mk) - 1;
        const u32 w0 = (u32)__builtin_amdgcn_readlane((int)unv, l0);
        const u32 slot = ((wbase + (u32)l0) << 5) + (u32)__builtin_ctz(w0);
        pos = slot + 1;
        if (g.incremental && !g.has_vec[slot]) continue; // prefix_iter over Item keys: existing items
        const int ef2 = (int)ak->knn_ef > total ? (int)ak->knn_ef - total : 0; // saturating_sub :786
        if (ln == 0) eps[0] = slot;
        WSYNC();
        if constexpr (SHORT)
          walk_layer_short<LPR, RCN>(g, q, qn, 0u, ef2, eps, 1, s, vis, vb, nb_ids, nb_d, evals, err_iter, rb);
        else
          walk_one_layer<LPR, NCH, BIG_EPS, RC, QN, LMERGE, PAGED>(g, q, qn, 0u, ef2, eps, 1, s, vis, nb_ids, nb_d, evals, err_iter, qrow, rb);
        KA_FRESH();
        if (total + s.res_len > (int)ak->rcap) {
          s.err = 1;
          break;
        }
        if constexpr (RB) { // neighbours.extend(more_nns)
#pragma unroll
          for (int c = 0; c < RCN; c++)
            if (ln + 64 * c < s.res_len)
              ak->cand[(size_t)m * ak->rcap + total + 64 * c + ln] =
                  (rb.r[c] & 0xFFFFFFFF00000000ull) | ((rb.r[c] >> 1) & 0x7FFFFFFFull);
        } else {
          for (int e = ln; e < s.res_len; e += 64) {
            u64 k = s.res[e];
            ak->cand[(size_t)m * ak->rcap + total + e] = (k & 0xFFFFFFFF00000000ull) | ((k >> 1) & 0x7FFFFFFFull);
          }
        }
        total += s.res_len;
        if (total >= (int)ak->knn_ef) break; // :792-794
      }
      // drain_asc(): sort everything that was collected
      __threadfence_block();
      WSYNC();
      for (int e = ln; e < total; e += 64) s.res[e] = ak->cand[(size_t)m * ak->rcap + e];
      WSYNC();
      for (int e = ln; e < total; e += 64) {
        const u64 mine = s.res[e];
        int rk = 0;
        for (int k2 = 0; k2 < total; k2++) rk += s.res[k2] < mine ? 1 : 0;
        ak->cand[(size_t)m * ak->rcap + rk] = mine;
      }
      WSYNC();
    }
    if constexpr (RM || SP == 0) {
      // a search whose tie pool overflowed: flagged instead of counted, the host repeats the query on the
      // heap-queue searcher (k_nns_filtered without a filter), which has no pool
      // (force_pool: tests send every k-th query that way)
      if (reader_mode && ak->pool_flag && (s.pool_over || (ak->force_pool && m % ak->force_pool == 0u))) {
        total = (int)0xFFFFFFFEu;
        s.pool_over = 0;
      }
    }
    if (ln == 0) ak->cand_n[m] = (u32)total;
    if (vis.log_over) log_over_cnt++;
    visited_clear(vis);
    visb_clear(vb);
  }
#ifdef HNY_PHASE_CLOCKS
  if (ln == 0) {
    atomicAdd(&g.stats[ST_PH_POP], s.ph[0]);
    atomicAdd(&g.stats[ST_PH_LIST], s.ph[1]);
    atomicAdd(&g.stats[ST_PH_DIST], s.ph[2]);
    atomicAdd(&g.stats[ST_PH_INSERT], s.ph[3]);
    atomicAdd(&g.stats[ST_PH_EXPANSIONS], s.ph[4]);
    atomicAdd(&g.stats[ST_PH_REST], __builtin_readcyclecounter() - ph_kernel_t0);
    atomicAdd(&g.stats[ST_PH_VIS], s.ph[5]);
    atomicAdd(&g.stats[ST_PH_NASK], s.ph[6]);
    atomicAdd(&g.stats[ST_PH_NACC], s.ph[7]);
    atomicAdd(&g.stats[ST_PH_NMERGE], s.ph[8]);
    atomicAdd(&g.stats[ST_PH_NPOOL], s.ph[9]);
    atomicAdd(&g.stats[ST_PH_NONEW], s.ph[10]);
  }
#endif
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (s.pool_over) atomicAdd(&g.stats[ST_POOL_OVERFLOW], (u64)s.pool_over);
    if (log_over_cnt) atomicAdd(&g.stats[ST_LOG_OVERFLOW], (u64)log_over_cnt);
    if (s.err) atomicAdd(&g.stats[ST_ERR_RES_OVERFLOW], 1ull);
    if (err_iter) atomicAdd(&g.stats[ST_ERR_ITER], 1ull);
  }
}
#undef KA_FRESH

// ---------------------------------------------------------------------------------------------
// Reader::nns with a candidates filter and/or by_item (reader.rs:301-369, 642-711, 809-896).
// With a filter the `search_queue` takes every accepted point while `res` only takes the ones the
// filter lets through (:356-360), so the two decouple: `res` stays the sorted LDS array (no
// expanded bit, no tie pool), the queue becomes a real 64-ary min-heap in HBM (one coalesced
// 512-B load per level on the way down), keyed dist bits << 32 | ~slot: smallest distance bits
// first, larger id first among equals — BinaryHeap<(Reverse<OrderedFloat>, ItemId)> (:310).
// ---------------------------------------------------------------------------------------------
struct QHeap {
  volatile u64 *h;
  u32 size, cap;
  u64 top;
};

__device__ __forceinline__ bool qheap_push(QHeap &Q, u64 key) {
  if (Q.size >= Q.cap) return false;
  const int ln = threadIdx.x;
  u32 j = Q.size++;
  while (j > 0) {
    const u32 p = (j - 1u) >> 6;
    const u64 pk = uni(Q.h[p]);
    if (pk <= key) break;
    if (ln == 0) Q.h[j] = pk;
    j = p;
  }
  if (ln == 0) Q.h[j] = key;
  if (j == 0) Q.top = key;
  __threadfence_block();
  WSYNC();
  return true;
}

__device__ __forceinline__ void qheap_pop(QHeap &Q) { // Q.size > 0
  const int ln = threadIdx.x;
  Q.size--;
  if (Q.size == 0) return;
  const u64 key = uni(Q.h[Q.size]);
  u32 i = 0;
  for (;;) {
    const u32 base = (i << 6) + 1u;
    if (base >= Q.size) break;
    const u32 c = base + (u32)ln;
    const u64 ck = c < Q.size ? Q.h[c] : ~0ull;
    const u64 mn = wave_min_u64(ck);
    if (mn >= key) break;
    const int wl = __ffsll((long long)ballot(ck == mn)) - 1; // keys are unique
    if (ln == 0) Q.h[i] = mn;
    if (i == 0) Q.top = mn;
    i = base + (u32)wl;
  }
  if (ln == 0) Q.h[i] = key;
  if (i == 0) Q.top = key;
  __threadfence_block();
  WSYNC();
}

// res.push (len != ef) / res.push_pop_max (len == ef) on the sorted array, reader.rs:361-365
__device__ __forceinline__ void sorted_insert(u64 *res, int &len, u64 key, int ef, int rcap, u32 &err) {
  const int ln = threadIdx.x;
  int pos = 0;
  for (int base = 0; base < len; base += 64) {
    int e = base + ln;
    bool lt = e < len && res[e] < key;
    pos += __popcll(ballot(lt));
  }
  const bool evict = (len == ef);
  if (evict && pos == len) return; // the new entry is the max: pushed and popped at once
  if (!evict && len >= rcap) {
    err = 1;
    return;
  }
  const int hi = evict ? len - 1 : len;
  if (hi > pos) {
    for (int base = (hi - 1) & ~63; base >= (pos & ~63); base -= 64) {
      int e = base + ln;
      bool mv = e >= pos && e < hi;
      u64 v = mv ? res[e] : 0ull;
      WSYNC();
      if (mv) res[e + 1] = v;
      WSYNC();
    }
  }
  if (ln == 0) res[pos] = key;
  WSYNC();
  if (!evict) len++;
}

__device__ __forceinline__ bool in_filter(const u32 *filter, u32 excl, u32 id) {
  if (id == excl) return false; // by_item: candidates.remove(item), reader.rs:840
  return !filter || ((filter[id >> 5] >> (id & 31u)) & 1u);
}

// Visitor::visit at level 0 with `candidates` (reader.rs:301-369).  Returns 0, or 1 when the heap
// is too small (the caller reports it and the host runs the query again with a larger one).
template <int LPR, int NCH>
__device__ int visit_filtered(const GraphDev &g, const float4 (&q)[NCH], float qn, int ef, const u32 *eps,
                              int n_eps, u64 *res, int &res_len, int rcap, u32 &res_err, Visited &vis,
                              u32 *nb_ids, float *nb_d, QHeap &Q, const u32 *filter, u32 excl,
                              u64 &evals, u32 &err_iter, const unsigned char *qrow) {
  const int ln = threadIdx.x;
  res_len = 0;
  Q.size = 0;
  Q.top = ~0ull;
  // :316-325 every entry point is queued and visited; res takes it only if the filter does
  for (int e0 = 0; e0 < n_eps; e0 += 64) {
    const int ne = n_eps - e0 < 64 ? n_eps - e0 : 64;
    u32 id = ln < ne ? eps[e0 + ln] : 0u;
    bool isnew = visited_insert(vis, id, ln < ne);
    u64 nmask = ballot(isnew);
    visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
    WSYNC();
    if (ln < ne) nb_ids[ln] = id;
    WSYNC();
    dist_rows<LPR, NCH>(g, q, qn, nb_ids, ne, nb_d, qrow);
    evals += (u64)ne;
    WSYNC();
    for (int r = 0; r < ne; r++) {
      const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
      if (!qheap_push(Q, ((u64)db << 32) | (u64)(~idr))) return 1;
      if (in_filter(filter, excl, idr)) sorted_insert(res, res_len, ((u64)db << 32) | idr, 0x7FFFFFFF, rcap, res_err);
    }
  }
  for (u32 iter = 0;; iter++) {
    if (iter > 4000000u || res_err) {
      if (iter > 4000000u) err_iter = 1;
      break;
    }
    if (Q.size == 0) break;
    const u64 top = Q.top;
    const float f = __uint_as_float((u32)(top >> 32));
    const float fmax = res_len ? __uint_as_float(uni((u32)(res[res_len - 1] >> 32))) : 3.4028235e38f; // :337
    if (f > fmax) break; // raw f32 compare, :338
    qheap_pop(Q);
    const u32 cslot = ~(u32)(top & 0xFFFFFFFFull);
    for (int pass = g.incremental ? 0 : 1; pass < 2; pass++) {
      u32 cap;
      const u32 *nl = pass == 0 ? disk_ids(g, 0u, cslot, cap) : nbr_ids(g, 0u, cslot, cap);
      if (!nl) continue;
      for (u32 c0 = 0; c0 < cap; c0 += 64u) { // lists of more than 64 slots: 64 at a time, in order
      u32 id = c0 + (u32)ln < cap ? nl[c0 + ln] : HNY_SENT;
      bool valid = id != HNY_SENT;
      bool isnew = visited_insert(vis, id, valid); // path.insert(point), :347
      u64 nmask = ballot(isnew);
      if (!nmask) continue;
      visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
      if (g.incremental) {
        isnew = isnew && g.has_vec[id] != 0;
        nmask = ballot(isnew);
        if (!nmask) continue;
      }
      const int n_new = __popcll(nmask);
      const int rank = __popcll(nmask & ((1ull << ln) - 1ull));
      WSYNC();
      if (isnew) nb_ids[rank] = id;
      WSYNC();
      dist_rows<LPR, NCH>(g, q, qn, nb_ids, n_new, nb_d, qrow); // :350-353
      evals += (u64)n_new;
      WSYNC();
      for (int r = 0; r < n_new; r++) { // ascending ids, like links.iter()
        const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
        if (res_len < ef || __uint_as_float(db) < fmax) { // :357
          if (!qheap_push(Q, ((u64)db << 32) | (u64)(~idr))) return 1;
          if (in_filter(filter, excl, idr)) sorted_insert(res, res_len, ((u64)db << 32) | idr, ef, rcap, res_err);
        }
      }
      } // pages of the list
    }
  }
  return 0;
}

template <int LPR, int NCH>
__global__ __launch_bounds__(64, 4) void k_nns_filtered(GraphDev g, NnsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *res = reinterpret_cast<u64 *>(smem);
  u64 *pool = res + a.rcap;
  u32 *nb_ids = reinterpret_cast<u32 *>(pool + HNY_POOL_CAP);
  float *nb_d = reinterpret_cast<float *>(nb_ids + 64);
  u32 *eps = reinterpret_cast<u32 *>(nb_d + 64);
  const int ln = threadIdx.x, t = ln % LPR;

  Beam s; // greedy descent through the upper layers: the ordinary (unfiltered) walk
#ifdef HNY_PHASE_CLOCKS
  for (int i = 0; i < 12; i++) s.ph[i] = 0;
  s.ph_t = 0;
#endif
  s.res = res;
  s.pool = pool;
  s.rcap = (int)a.rcap;
  s.pool_over = 0;
  s.err = 0;
  s.res_len = 0;
  s.pool_len = 0;
  s.n_weird = 0;
  s.tie_bits = 0;
  s.dropped = false;
  BeamR<1> rb_unused{{0ull}};
  Visited vis;
  visited_init(vis, a.bits + (size_t)blockIdx.x * a.bits_words, a.bits_words,
               a.vlog + (size_t)blockIdx.x * a.log_cap, a.log_cap, eps + a.eps_cap, a.vis_slots);
  QHeap Q;
  Q.h = a.heap + (size_t)blockIdx.x * a.heap_cap;
  Q.cap = a.heap_cap;
  Q.size = 0;
  Q.top = ~0ull;
  u64 evals = 0;
  u32 err_iter = 0, log_over_cnt = 0, res_err = 0;

  for (;;) {
    u32 mi = 0;
    if (ln == 0) {
      mi = atomicAdd(a.queue, 1u);
      // the Visitor's cancel probe (reader.rs:333), between queries: a cancelled batch starts no more
      if (a.cancel && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) mi = 0xFFFFFFFFu;
    }
    mi = uni(mi);
    if (mi >= a.n_members) break;
    const u32 m = a.members ? uni(a.members[mi]) : mi;
    const unsigned char *qrow;
    float qn = 0.f;
    u32 excl = HNY_SENT;
    if (a.by_item) { // :826-828 the stored vector is the query
      excl = uni(a.q_slots[m]);
      qrow = g.rows + (size_t)excl * g.row_stride;
      if (g.norms) qn = g.norms[excl];
    } else {
      qrow = a.q_rows + (size_t)m * a.q_stride;
      if (a.q_norms) qn = a.q_norms[m];
    }
    float4 q[NCH];
    load_row<LPR, NCH>(qrow, t, g.n16, q);
    int n_eps;
    if (a.by_item) { // Visitor::new(vec![item], 0, ef, Some(&candidates)), :842
      n_eps = 1;
      if (ln == 0) eps[0] = excl;
      WSYNC();
    } else { // :728-743 greedy descent, no filter, `path` shared and cleared before level 0
      n_eps = (int)a.n_entry_points;
      for (int i = ln; i < n_eps; i += 64) eps[i] = a.entry_points[i];
      WSYNC();
      for (u32 layer = g.max_level; layer >= 1u; layer--) {
        walk_one_layer<LPR, NCH, true>(g, q, qn, layer, 1, eps, n_eps, s, vis, nb_ids, nb_d, evals, err_iter, qrow, rb_unused);
        const u32 closest = uni((u32)(s.res[0] >> 1) & 0x7FFFFFFFu);
        WSYNC();
        if (ln == 0) eps[0] = closest;
        n_eps = 1;
        if (layer == 1u) {
          if (vis.log_over) log_over_cnt++;
          visited_clear(vis);
        }
        WSYNC();
      }
    }
    int res_len = 0;
    int st = visit_filtered<LPR, NCH>(g, q, qn, (int)a.ef_main, eps, n_eps, res, res_len, (int)a.rcap, res_err,
                                      vis, nb_ids, nb_d, Q, a.filter, excl, evals, err_iter, qrow);
    int total = 0;
    if (st == 0) {
      for (int e = ln; e < res_len; e += 64) a.cand[(size_t)m * a.rcap + e] = res[e];
      total = res_len;
      if (total < (int)a.k) {
        // exhaustive fallback (:771-795 / :864-890): restart from every item not on `path` yet
        const u32 nwords = (g.n + 31) >> 5;
        const int stop = a.by_item ? (int)a.k : (int)a.ef_opt;
        u32 pos = 0;
        visited_flush(vis);
        while (pos < g.n) {
          const u32 wbase = pos >> 5;
          const u32 widx = wbase + (u32)ln;
          u32 unv = 0u;
          if (widx < nwords) {
            unv = ~__hip_atomic_load(&vis.bits[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ln == 0 && (pos & 31u)) unv &= ~((1u << (pos & 31u)) - 1u);
            if (widx == nwords - 1 && (g.n & 31u)) unv &= (1u << (g.n & 31u)) - 1u;
          }
          const u64 mk = ballot(unv != 0u);
          if (!mk) {
            pos = (wbase + 64u) << 5;
            continue;
          }
          const int l0 = __ffsll((long long)mk) - 1;
          const u32 w0 = (u32)__builtin_amdgcn_readlane((int)unv, l0);
          const u32 slot = ((wbase + (u32)l0) << 5) + (u32)__builtin_ctz(w0);
          pos = slot + 1;
          if (g.incremental && !g.has_vec[slot]) continue;
          int ef2;
          if (a.by_item) ef2 = (int)a.k - total;                                   // :878
          else ef2 = (int)a.ef_opt > total ? (int)a.ef_opt - total : 0;           // :783
          WSYNC();
          if (ln == 0) eps[0] = slot;
          WSYNC();
          st = visit_filtered<LPR, NCH>(g, q, qn, ef2, eps, 1, res, res_len, (int)a.rcap, res_err, vis, nb_ids,
                                        nb_d, Q, a.filter, excl, evals, err_iter, qrow);
          if (st) break;
          if (total + res_len > (int)a.rcap) {
            res_err = 1;
            break;
          }
          for (int e = ln; e < res_len; e += 64) a.cand[(size_t)m * a.rcap + total + e] = res[e];
          total += res_len;
          if (total >= stop) break;
        }
        if (st == 0 && !res_err) { // drain_asc()
          __threadfence_block();
          WSYNC();
          for (int e = ln; e < total; e += 64) res[e] = a.cand[(size_t)m * a.rcap + e];
          WSYNC();
          for (int e = ln; e < total; e += 64) {
            const u64 mine = res[e];
            int rk = 0;
            for (int k2 = 0; k2 < total; k2++) rk += res[k2] < mine ? 1 : 0;
            a.cand[(size_t)m * a.rcap + rk] = mine;
          }
          WSYNC();
        }
      }
    }
    if (ln == 0) {
      a.cand_n[m] = st ? 0u : (u32)total;
      a.status[m] = st ? 1u : 0u;
    }
    if (vis.log_over) log_over_cnt++;
    visited_clear(vis);
  }
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (log_over_cnt) atomicAdd(&g.stats[ST_LOG_OVERFLOW], (u64)log_over_cnt);
    if (s.err || res_err) atomicAdd(&g.stats[ST_ERR_RES_OVERFLOW], 1ull);
    if (err_iter) atomicAdd(&g.stats[ST_ERR_ITER], 1ull);
  }
}

// ---------------------------------------------------------------------------------------------
// walk_layer with the reference's own data structures (hnsw.rs:460-518): `candidates` a real min-heap
// (64-ary, HBM, keyed dist bits << 32 | ~slot: smallest distance first, larger id first among equals —
// BinaryHeap<(Reverse<OrderedFloat>, ItemId)>), `res` a real max-heap (the same 64-ary heap on inverted
// keys ~(dist bits << 32 | slot): MinMaxHeap::peek_max / push_pop_max), every accepted point pushed to
// both, in list order, as the reference's loop does.  No beam array, no expanded bits, no tie pool: nothing
// can overflow but the heaps' memory.  Slow (every heap step is a trip to HBM) and only used for the
// members whose fast walk overflowed its 128-slot tie pool (k_walk_heap below) — inputs with a handful of
// distinct distances and lists of hundreds of links.  Returns 0, or 1 when a heap is full.
// rmin: the smallest key res holds (peek_min for the greedy descent; only the max ever leaves res).
// ---------------------------------------------------------------------------------------------
template <int LPR, int NCH>
__device__ int walk_layer_heap(const GraphDev &g, const float4 (&q)[NCH], float qn, u32 layer, int ef, const u32 *eps,
                               int n_eps, QHeap &C, QHeap &R, u64 &rmin, Visited &vis, u32 *nb_ids, float *nb_d,
                               u64 &evals, const unsigned char *qrow) {
  const int ln = threadIdx.x;
  C.size = 0;
  C.top = ~0ull;
  R.size = 0;
  R.top = ~0ull;
  rmin = ~0ull;
  for (int e0 = 0; e0 < n_eps; e0 += 64) { // :474-481 every entry point, no capacity check
    const int ne = n_eps - e0 < 64 ? n_eps - e0 : 64;
    u32 id = ln < ne ? eps[e0 + ln] : 0u;
    bool isnew = visited_insert(vis, id, ln < ne);
    u64 nmask = ballot(isnew);
    visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
    WSYNC();
    if (ln < ne) nb_ids[ln] = id;
    WSYNC();
    dist_rows<LPR, NCH>(g, q, qn, nb_ids, ne, nb_d, qrow);
    evals += (u64)ne;
    WSYNC();
    for (int r = 0; r < ne; r++) {
      const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
      const u64 key = ((u64)db << 32) | (u64)idr;
      if (!qheap_push(C, ((u64)db << 32) | (u64)(~idr))) return 1;
      if (!qheap_push(R, ~key)) return 1;
      rmin = key < rmin ? key : rmin;
    }
  }
  for (;;) {
    if (C.size == 0) break; // :483 candidates.peek()
    const u64 top = C.top;
    const u32 dmax = (u32)((~R.top) >> 32); // res.peek_max(), captured once per pop (:484)
    const float fmax = __uint_as_float(dmax);
    if (__uint_as_float((u32)(top >> 32)) > fmax) break; // raw f32 compare, :485
    qheap_pop(C);
    const u32 cslot = ~(u32)(top & 0xFFFFFFFFull);
    for (int pass = g.incremental ? 0 : 1; pass < 2; pass++) { // on-disk Links first (:438-441)
      u32 cap;
      const u32 *nl = pass == 0 ? disk_ids(g, layer, cslot, cap) : nbr_ids(g, layer, cslot, cap);
      if (!nl) continue;
      for (u32 c0 = 0; c0 < cap; c0 += 64u) { // lists of more than 64 slots: 64 at a time, in list order
        const u32 pcap = cap - c0 < 64u ? cap - c0 : 64u;
        u32 id = (u32)ln < pcap ? nl[c0 + ln] : HNY_SENT;
        const bool valid = id != HNY_SENT;
        bool isnew = visited_insert(vis, id, valid); // :493
        u64 nmask = ballot(isnew);
        if (!nmask) continue;
        visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
        // the same id twice in one list (add_link never dedups, :521): the FIRST occurrence is the one the
        // reference scores; which of the two lanes the atomic told "new" is arbitrary
        WSYNC();
        nb_ids[ln] = valid ? id : HNY_SENT;
        WSYNC();
        int firstj = ln;
        bool anynew = isnew;
        for (int j = 0; j < (int)pcap; j++) {
          const u32 oj = nb_ids[j];
          if (valid && oj == id) {
            if (j < firstj) firstj = j;
            if ((nmask >> j) & 1ull) anynew = true;
          }
        }
        isnew = valid && anynew && firstj == ln;
        if (g.incremental) isnew = isnew && g.has_vec[id] != 0; // MissingKey: visited, never scored (:498-502)
        WSYNC();
        nmask = ballot(isnew);
        if (!nmask) continue;
        const int n_new = __popcll(nmask);
        const int rank = __popcll(nmask & ((1ull << ln) - 1ull));
        if (isnew) nb_ids[rank] = id;
        WSYNC();
        dist_rows<LPR, NCH>(g, q, qn, nb_ids, n_new, nb_d, qrow); // :503
        evals += (u64)n_new;
        WSYNC();
        for (int r = 0; r < n_new; r++) { // list order, one at a time like the reference's loop
          const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
          if ((int)R.size < ef || __uint_as_float(db) < fmax) { // :505
            const u64 key = ((u64)db << 32) | (u64)idr;
            if (!qheap_push(C, ((u64)db << 32) | (u64)(~idr))) return 1;
            if ((int)R.size == ef) { // push_pop_max: the new key, unless it is the greatest itself
              if (key < ~R.top) {
                qheap_pop(R);
                if (!qheap_push(R, ~key)) return 1;
                rmin = key < rmin ? key : rmin;
              }
            } else {
              if (!qheap_push(R, ~key)) return 1;
              rmin = key < rmin ? key : rmin;
            }
          }
        }
      }
    }
  }
  return 0;
}

// k_walk's member loop for the members of a.pool_retry, every walk_layer call on heaps: build walks whose
// tie pool overflowed, and (a.reader_mode) the Reader's searches in the same situation whose result set is
// too long for k_nns_filtered's LDS (ef_search >= 4 096: hny_builder_search_knn)
template <int LPR, int NCH>
__global__ __launch_bounds__(64, 4) void k_walk_heap(GraphDev g, WalkArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  u32 *nb_ids = reinterpret_cast<u32 *>(smem);
  float *nb_d = reinterpret_cast<float *>(nb_ids + 64);
  u32 *eps = reinterpret_cast<u32 *>(nb_d + 64);
  const int ln = threadIdx.x, t = ln % LPR;
  const u32 n_mem = *a.n_pool_retry;
  if (n_mem == 0u) return;
  Visited vis;
  visited_init(vis, a.bits + (size_t)blockIdx.x * a.bits_words, a.bits_words, a.vlog + (size_t)blockIdx.x * a.log_cap,
               a.log_cap, nullptr, 0u);
  QHeap C, R;
  C.h = a.heap_c + (size_t)blockIdx.x * a.heap_c_cap;
  C.cap = a.heap_c_cap;
  R.h = a.heap_r + (size_t)blockIdx.x * a.heap_r_cap;
  R.cap = a.heap_r_cap;
  u64 evals = 0;
  u32 err = 0, log_over_cnt = 0;
  for (;;) {
    u32 mi = 0;
    if (ln == 0) {
      mi = atomicAdd(a.queue, 1u);
      // the Visitor's cancel probe (reader.rs:333), between queries (searches only: builds pass no flag)
      if (a.cancel && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) mi = 0xFFFFFFFFu;
    }
    mi = uni(mi);
    if (mi >= n_mem) break;
    const u32 m = uni(a.pool_retry[mi]);
    const u64 evals_before = evals;
    const unsigned char *qrow;
    float qn = 0.f;
    if (a.q_rows) { // a search: the query is not an item
      qrow = a.q_rows + (size_t)m * a.q_stride;
      if (a.q_norms) qn = a.q_norms[m];
    } else {
      const u32 qslot = a.q_slots[m];
      qrow = g.rows + (size_t)qslot * g.row_stride;
      if (g.norms) qn = g.norms[qslot];
    }
    float4 q[NCH];
    load_row<LPR, NCH>(qrow, t, g.n16, q);
    int n_eps;
    u32 start_layer;
    if (a.first) { // hnsw.rs:298 eps = all entry points
      n_eps = (int)a.n_entry_points;
      for (int i = ln; i < n_eps; i += 64) eps[i] = a.entry_points[i];
      start_layer = g.max_level;
    } else if (a.eps_in) { // resume after a descend_only launch
      n_eps = 1;
      if (ln == 0) eps[0] = a.eps_in[m];
      start_layer = a.layer;
    } else { // :316-321 eps = what was selected on the layer above
      const u64 *sl = a.sel + (size_t)m * a.sel_stride + (size_t)(a.batch_level - (a.layer + 1)) * (a.cap_sel + 1);
      n_eps = (int)sl[0];
      for (int i = ln; i < n_eps; i += 64) eps[i] = (u32)(sl[1 + i] & 0xFFFFFFFFull); // (more than 64: M > 64)
      start_layer = a.layer;
    }
    WSYNC();
    u64 lkey = 0, rmin = ~0ull;
    int st = 0;
    for (u32 layer = start_layer;; layer--) {
      const bool last = (layer == a.layer);
      if (last && a.descend_only) break;
      st = walk_layer_heap<LPR, NCH>(g, q, qn, layer, last ? (int)a.ef : 1, eps, n_eps, C, R, rmin, vis, nb_ids, nb_d,
                                     evals, qrow);
      if (st || last) break;
      const u32 closest = (u32)(rmin & 0xFFFFFFFFull); // :305-306 eps = [peek_min]
      WSYNC();
      if (ln == 0) eps[0] = closest;
      n_eps = 1;
      lkey = (lkey << 16) | (u64)((u32)g.upper_idx[closest] & 0xFFFFu);
      // walk_layer owns a fresh visited set; Reader::hnsw_search shares `path` across the greedy layers and
      // clears it once before layer 0 (reader.rs:731-743)
      if (!a.reader_mode || layer == a.layer + 1) {
        if (vis.log_over) log_over_cnt++;
        visited_clear(vis);
      }
      WSYNC();
    }
    if (!st && a.descend_only) {
      if (ln == 0) {
        a.eps_out[m] = eps[0];
        a.key_out[m - a.key_base] = lkey & 0xFFFFFFFFFFFFull;
      }
    } else if (!st) {
      // res.into_vec(), ascending: pop the maxima into the row from its end
      u32 total = R.size;
      u64 *row = a.cand + (size_t)m * a.rcap;
      auto drain = [&](u32 at) { // R, ascending, into row[at ..)
        for (u32 i = R.size; i-- > 0u;) {
          const u64 key = ~R.top;
          if (ln == 0) row[at + i] = key;
          qheap_pop(R);
        }
      };
      if (total > a.rcap) {
        st = 1;
      } else {
        drain(0u);
        if (a.reader_mode && total < a.knn_k) {
          // Reader::hnsw_search's exhaustive fallback (reader.rs:771-795), as in k_walk: restart from every
          // item not seen yet, ascending, sharing the visited set, until opt.ef hits are collected
          const u32 nwords = (g.n + 31) >> 5;
          u32 pos = 0;
          visited_flush(vis);
          while (pos < g.n && !st) {
            const u32 wbase = pos >> 5;
            const u32 widx = wbase + (u32)ln;
            u32 unv = 0u;
            if (widx < nwords) {
              unv = ~__hip_atomic_load(&vis.bits[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (ln == 0 && (pos & 31u)) unv &= ~((1u << (pos & 31u)) - 1u);
              if (widx == nwords - 1 && (g.n & 31u)) unv &= (1u << (g.n & 31u)) - 1u;
            }
            const u64 mk = ballot(unv != 0u);
            if (!mk) {
              pos = (wbase + 64u) << 5;
              continue;
            }
            const int l0 = __ffsll((long long)mk) - 1;
            const u32 w0 = (u32)__builtin_amdgcn_readlane((int)unv, l0);
            const u32 slot = ((wbase + (u32)l0) << 5) + (u32)__builtin_ctz(w0);
            pos = slot + 1;
            if (g.incremental && !g.has_vec[slot]) continue;
            const int ef2 = a.knn_ef > total ? (int)(a.knn_ef - total) : 0; // saturating_sub :786
            WSYNC();
            if (ln == 0) eps[0] = slot;
            WSYNC();
            st = walk_layer_heap<LPR, NCH>(g, q, qn, 0u, ef2, eps, 1, C, R, rmin, vis, nb_ids, nb_d, evals, qrow);
            if (st) break;
            if (total + R.size > a.rcap) {
              st = 1;
              break;
            }
            const u32 got = R.size;
            drain(total); // neighbours.extend(more_nns)
            total += got;
            if (total >= a.knn_ef) break; // :792-794
          }
          if (!st) { // drain_asc(): everything that was collected, sorted — through the heap once more
            __threadfence_block();
            WSYNC();
            R.size = 0;
            R.top = ~0ull;
            for (u32 i = 0; i < total && !st; i++) {
              const u64 key = uni64(row[i]);
              if (!qheap_push(R, ~key)) st = 1;
            }
            if (!st) drain(0u);
          }
        }
        if (!st && ln == 0) a.cand_n[m] = total;
      }
    }
    if (st) {
      if (a.pool_retry2) { // first tier: this member outgrew its heap — the second tier has room for every item
        if (ln == 0) a.pool_retry2[atomicAdd(a.n_pool_retry2, 1u)] = m;
        evals = evals_before; // (counted by the walk that completes)
      } else {
        err = 1;
      }
    }
    if (vis.log_over) log_over_cnt++;
    visited_clear(vis);
    WSYNC();
  }
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (log_over_cnt) atomicAdd(&g.stats[ST_LOG_OVERFLOW], (u64)log_over_cnt);
    if (err) atomicAdd(&g.stats[ST_ERR_RES_OVERFLOW], 1ull);
  }
}

// ---------------------------------------------------------------------------------------------
// Visitor::visit at level 0 (reader.rs:301-369) with the search queue AND `res` as heaps in HBM: result sets
// beyond the LDS (max(ef_search, k) >= 4 096 with a candidates filter / by_item — k_nns_filtered keeps `res`
// as a sorted LDS array).  The queue takes every accepted point, `res` only what the filter lets through
// (:322-324, :356-360); f_max = f32::MAX while `res` is empty (:337).  Returns 0, or 1 when a heap is full.
// ---------------------------------------------------------------------------------------------
template <int LPR, int NCH>
__device__ int visit_heap(const GraphDev &g, const float4 (&q)[NCH], float qn, int ef, const u32 *eps, int n_eps,
                          QHeap &C, QHeap &R, Visited &vis, u32 *nb_ids, float *nb_d, const u32 *filter, u32 excl,
                          u64 &evals, const unsigned char *qrow) {
  const int ln = threadIdx.x;
  C.size = 0;
  C.top = ~0ull;
  R.size = 0;
  R.top = ~0ull;
  for (int e0 = 0; e0 < n_eps; e0 += 64) { // :316-325 every entry point is queued and visited
    const int ne = n_eps - e0 < 64 ? n_eps - e0 : 64;
    u32 id = ln < ne ? eps[e0 + ln] : 0u;
    bool isnew = visited_insert(vis, id, ln < ne);
    u64 nmask = ballot(isnew);
    visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
    WSYNC();
    if (ln < ne) nb_ids[ln] = id;
    WSYNC();
    dist_rows<LPR, NCH>(g, q, qn, nb_ids, ne, nb_d, qrow);
    evals += (u64)ne;
    WSYNC();
    for (int r = 0; r < ne; r++) {
      const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
      if (!qheap_push(C, ((u64)db << 32) | (u64)(~idr))) return 1;
      if (in_filter(filter, excl, idr) && !qheap_push(R, ~(((u64)db << 32) | (u64)idr))) return 1;
    }
  }
  for (;;) {
    if (C.size == 0) break;
    const u64 top = C.top;
    const float fmax = R.size ? __uint_as_float((u32)((~R.top) >> 32)) : 3.4028235e38f; // :337, once per pop
    if (__uint_as_float((u32)(top >> 32)) > fmax) break;                                // raw f32 compare, :338
    qheap_pop(C);
    const u32 cslot = ~(u32)(top & 0xFFFFFFFFull);
    for (int pass = g.incremental ? 0 : 1; pass < 2; pass++) {
      u32 cap;
      const u32 *nl = pass == 0 ? disk_ids(g, 0u, cslot, cap) : nbr_ids(g, 0u, cslot, cap);
      if (!nl) continue;
      for (u32 c0 = 0; c0 < cap; c0 += 64u) { // lists of more than 64 slots: 64 at a time, in order
        u32 id = c0 + (u32)ln < cap ? nl[c0 + ln] : HNY_SENT;
        bool valid = id != HNY_SENT;
        bool isnew = visited_insert(vis, id, valid); // path.insert(point), :347
        u64 nmask = ballot(isnew);
        if (!nmask) continue;
        visited_log(vis, id, isnew, nmask, __popcll(nmask & ((1ull << ln) - 1ull)));
        if (g.incremental) {
          isnew = isnew && g.has_vec[id] != 0;
          nmask = ballot(isnew);
          if (!nmask) continue;
        }
        const int n_new = __popcll(nmask);
        const int rank = __popcll(nmask & ((1ull << ln) - 1ull));
        WSYNC();
        if (isnew) nb_ids[rank] = id;
        WSYNC();
        dist_rows<LPR, NCH>(g, q, qn, nb_ids, n_new, nb_d, qrow); // :350-353
        evals += (u64)n_new;
        WSYNC();
        for (int r = 0; r < n_new; r++) { // ascending ids, like links.iter()
          const u32 db = uni(fbits(nb_d[r])), idr = uni(nb_ids[r]);
          if ((int)R.size < ef || __uint_as_float(db) < fmax) { // :357
            if (!qheap_push(C, ((u64)db << 32) | (u64)(~idr))) return 1;
            if (!in_filter(filter, excl, idr)) continue;
            const u64 key = ((u64)db << 32) | (u64)idr;
            if ((int)R.size == ef) { // push_pop_max: the new key, unless it is the greatest itself (or ef == 0)
              if (R.size && key < ~R.top) {
                qheap_pop(R);
                if (!qheap_push(R, ~key)) return 1;
              }
            } else if (!qheap_push(R, ~key)) {
              return 1;
            }
          }
        }
      }
    }
  }
  return 0;
}

// k_nns_filtered on heaps: same queries, same results, `res` of any length (NnsArgs.heap = the search queues,
// heap_r = the result heaps; rows of a.cand hold up to rcap hits)
template <int LPR, int NCH>
__global__ __launch_bounds__(64, 4) void k_nns_heap(GraphDev g, NnsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  u32 *nb_ids = reinterpret_cast<u32 *>(smem);
  float *nb_d = reinterpret_cast<float *>(nb_ids + 64);
  u32 *eps = reinterpret_cast<u32 *>(nb_d + 64);
  const int ln = threadIdx.x, t = ln % LPR;
  Visited vis;
  visited_init(vis, a.bits + (size_t)blockIdx.x * a.bits_words, a.bits_words, a.vlog + (size_t)blockIdx.x * a.log_cap,
               a.log_cap, nullptr, 0u);
  QHeap C, R;
  C.h = a.heap + (size_t)blockIdx.x * a.heap_cap;
  C.cap = a.heap_cap;
  R.h = a.heap_r + (size_t)blockIdx.x * a.heap_r_cap;
  R.cap = a.heap_r_cap;
  u64 evals = 0;
  u32 log_over_cnt = 0;
  for (;;) {
    u32 mi = 0;
    if (ln == 0) {
      mi = atomicAdd(a.queue, 1u);
      if (a.cancel && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) mi = 0xFFFFFFFFu;
    }
    mi = uni(mi);
    if (mi >= a.n_members) break;
    const u32 m = a.members ? uni(a.members[mi]) : mi;
    const unsigned char *qrow;
    float qn = 0.f;
    u32 excl = HNY_SENT;
    if (a.by_item) { // :826-828 the stored vector is the query
      excl = uni(a.q_slots[m]);
      qrow = g.rows + (size_t)excl * g.row_stride;
      if (g.norms) qn = g.norms[excl];
    } else {
      qrow = a.q_rows + (size_t)m * a.q_stride;
      if (a.q_norms) qn = a.q_norms[m];
    }
    float4 q[NCH];
    load_row<LPR, NCH>(qrow, t, g.n16, q);
    int n_eps, st = 0;
    u64 rmin = ~0ull;
    if (a.by_item) { // Visitor::new(vec![item], 0, ef, Some(&candidates)), :842
      n_eps = 1;
      if (ln == 0) eps[0] = excl;
      WSYNC();
    } else { // :728-743 greedy descent, no filter, `path` shared and cleared before level 0
      n_eps = (int)a.n_entry_points;
      for (int i = ln; i < n_eps; i += 64) eps[i] = a.entry_points[i];
      WSYNC();
      for (u32 layer = g.max_level; layer >= 1u && !st; layer--) {
        st = walk_layer_heap<LPR, NCH>(g, q, qn, layer, 1, eps, n_eps, C, R, rmin, vis, nb_ids, nb_d, evals, qrow);
        const u32 closest = (u32)(rmin & 0xFFFFFFFFull);
        WSYNC();
        if (ln == 0) eps[0] = closest;
        n_eps = 1;
        if (layer == 1u) {
          if (vis.log_over) log_over_cnt++;
          visited_clear(vis);
        }
        WSYNC();
      }
    }
    if (!st) st = visit_heap<LPR, NCH>(g, q, qn, (int)a.ef_main, eps, n_eps, C, R, vis, nb_ids, nb_d, a.filter, excl, evals, qrow);
    u32 total = 0;
    u64 *row = a.cand + (size_t)m * a.rcap;
    auto drain = [&](u32 at) { // R, ascending, into row[at ..)
      for (u32 i = R.size; i-- > 0u;) {
        const u64 key = ~R.top;
        if (ln == 0) row[at + i] = key;
        qheap_pop(R);
      }
    };
    if (!st) {
      total = R.size;
      if (total > a.rcap) {
        st = 1;
      } else {
        drain(0u);
        if (total < a.k) {
          // exhaustive fallback (:771-795 / :864-890): restart from every item not on `path` yet
          const u32 nwords = (g.n + 31) >> 5;
          const u32 stop = a.by_item ? a.k : a.ef_opt;
          u32 pos = 0;
          visited_flush(vis);
          while (pos < g.n) {
            const u32 wbase = pos >> 5;
            const u32 widx = wbase + (u32)ln;
            u32 unv = 0u;
            if (widx < nwords) {
              unv = ~__hip_atomic_load(&vis.bits[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (ln == 0 && (pos & 31u)) unv &= ~((1u << (pos & 31u)) - 1u);
              if (widx == nwords - 1 && (g.n & 31u)) unv &= (1u << (g.n & 31u)) - 1u;
            }
            const u64 mk = ballot(unv != 0u);
            if (!mk) {
              pos = (wbase + 64u) << 5;
              continue;
            }
            const int l0 = __ffsll((long long)mk) - 1;
            const u32 w0 = (u32)__builtin_amdgcn_readlane((int)unv, l0);
            const u32 slot = ((wbase + (u32)l0) << 5) + (u32)__builtin_ctz(w0);
            pos = slot + 1;
            if (g.incremental && !g.has_vec[slot]) continue;
            int ef2;
            if (a.by_item) ef2 = (int)(a.k - total);                       // :878
            else ef2 = a.ef_opt > total ? (int)(a.ef_opt - total) : 0;     // :783
            WSYNC();
            if (ln == 0) eps[0] = slot;
            WSYNC();
            st = visit_heap<LPR, NCH>(g, q, qn, ef2, eps, 1, C, R, vis, nb_ids, nb_d, a.filter, excl, evals, qrow);
            if (st) break;
            if (total + R.size > a.rcap) {
              st = 1;
              break;
            }
            const u32 got = R.size;
            drain(total); // neighbours.extend(more_nns)
            total += got;
            if (total >= stop) break;
          }
          if (!st) { // drain_asc(): everything that was collected, sorted — through the heap once more
            __threadfence_block();
            WSYNC();
            R.size = 0;
            R.top = ~0ull;
            for (u32 i = 0; i < total && !st; i++) {
              const u64 key = uni64(row[i]);
              if (!qheap_push(R, ~key)) st = 1;
            }
            if (!st) drain(0u);
          }
        }
      }
    }
    if (ln == 0) {
      a.cand_n[m] = st ? 0u : total;
      a.status[m] = st ? 1u : 0u;
    }
    if (vis.log_over) log_over_cnt++;
    visited_clear(vis);
    WSYNC();
  }
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (log_over_cnt) atomicAdd(&g.stats[ST_LOG_OVERFLOW], (u64)log_over_cnt);
  }
}

// brute_force_search (reader.rs:667-711): rank the existing candidates by distance.  The
// BinaryHeap keeps `count` entries and replaces its top only by a strictly smaller distance while
// walking the ids upwards — i.e. it keeps the `count` smallest (bits(d), id) pairs, which does not
// depend on the evaluation order.
template <int LPR, int NCH>
__global__ __launch_bounds__(64, 4) void k_nns_linear(GraphDev g, NnsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *res = reinterpret_cast<u64 *>(smem);
  u32 *nb_ids = reinterpret_cast<u32 *>(res + a.rcap);
  float *nb_d = reinterpret_cast<float *>(nb_ids + 64);
  const int ln = threadIdx.x, t = ln % LPR;
  u64 evals = 0;
  u32 res_err = 0;
  for (;;) {
    u32 mi = 0;
    if (ln == 0) {
      mi = atomicAdd(a.queue, 1u);
      if (a.cancel && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) mi = 0xFFFFFFFFu;
    }
    mi = uni(mi);
    if (mi >= a.n_members) break;
    const u32 m = a.members ? uni(a.members[mi]) : mi;
    const unsigned char *qrow;
    float qn = 0.f;
    if (a.by_item) {
      const u32 qs = uni(a.q_slots[m]);
      qrow = g.rows + (size_t)qs * g.row_stride;
      if (g.norms) qn = g.norms[qs];
    } else {
      qrow = a.q_rows + (size_t)m * a.q_stride;
      if (a.q_norms) qn = a.q_norms[m];
    }
    float4 q[NCH];
    load_row<LPR, NCH>(qrow, t, g.n16, q);
    int res_len = 0;
    for (u32 base = 0; base < a.n_cand_slots; base += 64u) {
      const int nc = (int)(a.n_cand_slots - base < 64u ? a.n_cand_slots - base : 64u);
      WSYNC();
      if (ln < nc) nb_ids[ln] = a.cand_slots[base + ln];
      WSYNC();
      dist_rows<LPR, NCH>(g, q, qn, nb_ids, nc, nb_d, qrow);
      evals += (u64)nc;
      WSYNC();
      const u64 key = ln < nc ? (((u64)fbits(nb_d[ln]) << 32) | nb_ids[ln]) : ~0ull;
      const u64 cur_max = (res_len == (int)a.k && res_len) ? uni(res[res_len - 1]) : ~0ull;
      u64 want = ballot(ln < nc && (res_len < (int)a.k || key < cur_max));
      while (want) {
        const int r = __ffsll((long long)want) - 1;
        want &= want - 1ull;
        const u64 kr = ((u64)(u32)__builtin_amdgcn_readlane((int)(key >> 32), r) << 32) |
                       (u64)(u32)__builtin_amdgcn_readlane((int)(key & 0xFFFFFFFFull), r);
        sorted_insert(res, res_len, kr, (int)a.k, (int)a.rcap, res_err);
      }
    }
    WSYNC();
    for (int e = ln; e < res_len; e += 64) a.cand[(size_t)m * a.rcap + e] = res[e];
    if (ln == 0) {
      a.cand_n[m] = (u32)res_len;
      a.status[m] = 0u;
    }
  }
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (res_err) atomicAdd(&g.stats[ST_ERR_RES_OVERFLOW], 1ull);
  }
}

// ---------------------------------------------------------------------------------------------
// robust_prune (hnsw.rs:565-597) on a list that is already sorted ascending by (bits(d), id).
// list/sel keys: dist bits << 32 | slot.  `exists i in S: bits(d(c,i)*alpha) < bits(dq)` does not
// depend on evaluation order, so S is tested RPI rows at a time with an early exit per chunk.
// ---------------------------------------------------------------------------------------------
// LDS entries of the one-wave prune / add_link kernels' per-list arrays: the larger list capacity, in whole waves
// (one lane per slot up to 64; strict mode and rows beyond 8 KB take longer lists 64 slots at a time)
__host__ __device__ inline u32 wave_capmax(const GraphDev &g) {
  const u32 c = g.M0 > g.M ? g.M0 : g.M;
  return c <= (u32)HNY_MAX_CAP ? (u32)HNY_MAX_CAP : (c + 63u) / 64u * 64u;
}
template <int LPR, int NCH>
__device__ int wave_prune(const GraphDev &g, const u64 *list, int n, int cap, u64 *S, u32 *s_ids,
                          float *tmp_d, u64 &evals) {
  constexpr int RPI = (64 / LPR) * DistGroups<LPR, NCH>::U;
  const int ln = threadIdx.x, t = ln % LPR;
  int s_len = 0;
  for (int ci = 0; ci < n; ci++) {
    if (s_len == cap) break; // :577-579
    const u64 ck = list[ci];
    const u32 cid = (u32)(ck & 0xFFFFFFFFull), cdb = (u32)(ck >> 32);
    float4 c[NCH];
    load_row<LPR, NCH>(g.rows + (size_t)cid * g.row_stride, t, g.n16, c);
    float cn = g.norms ? g.norms[cid] : 0.f;
    bool viol = false;
    for (int k0 = 0; k0 < s_len && !viol; k0 += RPI) {
      int cnt = s_len - k0 < RPI ? s_len - k0 : RPI;
      dist_rows<LPR, NCH>(g, c, cn, s_ids + k0, cnt, tmp_d, g.rows + (size_t)cid * g.row_stride);
      evals += (u64)cnt;
      WSYNC();
      bool v = false;
      if (ln < cnt) {
        float da = tmp_d[ln] * g.alpha; // :585 OrderedFloat(d * alpha) < dist_to_query
        v = fbits(da) < cdb;
      }
      viol = ballot(v) != 0ull;
      WSYNC();
    }
    if (!viol) {
      if (ln == 0) {
        S[s_len] = ck;
        s_ids[s_len] = cid;
      }
      s_len++;
      WSYNC();
    }
  }
  return s_len;
}

template <int LPR, int NCH>
__global__ __launch_bounds__(64) void k_prune(GraphDev g, PruneArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *list = reinterpret_cast<u64 *>(smem);
  u64 *S = list + a.rcap;
  const u32 capmax = wave_capmax(g);
  u32 *s_ids = reinterpret_cast<u32 *>(S + capmax);
  float *tmp_d = reinterpret_cast<float *>(s_ids + capmax);
  const int ln = threadIdx.x;
  u64 evals = 0;
  for (u32 m = a.lo + blockIdx.x; m < a.hi; m += gridDim.x) {
    int n = (int)a.cand_n[m];
    for (int e = ln; e < n; e += 64) list[e] = a.cand[(size_t)m * a.rcap + e];
    WSYNC();
    int s_len = wave_prune<LPR, NCH>(g, list, n, (int)a.cap, S, s_ids, tmp_d, evals);
    u64 *out = a.sel + (size_t)m * a.sel_stride + (size_t)(a.batch_level - a.layer) * (a.cap_sel + 1);
    if (ln == 0) out[0] = (u64)s_len;
    for (int e = ln; e < s_len; e += 64) out[1 + e] = S[e];
    WSYNC();
  }
  if (ln == 0 && evals) atomicAdd(&g.stats[ST_EVALS_PRUNE], evals);
}

// ---------------------------------------------------------------------------------------------
// robust_prune with a 256-thread workgroup and the selected rows staged in LDS.  k_prune (one wave,
// selected rows re-read from HBM for every candidate) fetched 1.6 TB at C2 (profiles/r01 PMC); here
// every candidate row is read from HBM once (prefetched one candidate ahead) and compared against
// the selected rows held in LDS; the 4 waves split the selected rows.  Same distances, same result.
// ---------------------------------------------------------------------------------------------
struct WgPruneLds {
  u64 *S;          // [HNY_MAX_CAP] selected keys
  u32 *s_ids;      // [HNY_MAX_CAP]
  float *s_norm;   // [HNY_MAX_CAP]
  int *surv;       // [2][8] chunk member survived the test against S (by chunk parity)
  u32 *vmask;      // [8] bit j: chunk member violates against chunk member j
  float *cnorm;    // [8]
  unsigned char *cbuf;  // [NW][row_stride] the chunk's candidate rows
  unsigned char *stage; // [SL][row_stride] selected rows
  int SL;          // staged rows (multiple of 64/LPR)
};

// robust_prune for an NW-wave workgroup (NW = 4 or 8).  `exists i in S: bits(d(c,i)*alpha) <
// bits(dq)` does not depend on the order in which S is scanned, so NW consecutive candidates are
// tested concurrently, one per wave, against the selected set as it stood before the chunk (rows
// from LDS, early exit); the survivors are then tested against each other (rows exchanged through
// LDS) and the chunk is resolved in candidate order — exactly the sequential outcome, with 1 to 3
// barriers per NW candidates (see there).  Each wave streams its next candidate row from HBM one chunk ahead.
template <int LPR, int NCH, int NW>
__device__ __forceinline__ int wg_prune(const GraphDev &g, const u64 *list, int n, int cap, const WgPruneLds &L,
                                        u64 &evals) {
  constexpr int RPG = 64 / LPR;
  static_assert(NW == 4 || NW == 8, "chunk of 4 or 8 candidates");
  // (w through readfirstlane: the compiler then knows every per-wave condition below is uniform)
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, t = ln % LPR, sub = ln / LPR;
  const int j4 = fold4_row<LPR>();
  int s_len = 0;
  float4 nxt[NCH];
  float nxt_n = 0.f;
  if (w < n) {
    const u32 c0 = (u32)(list[w] & 0xFFFFFFFFull);
    load_row<LPR, NCH>(g.rows + (size_t)c0 * g.row_stride, t, g.n16, nxt);
    if (g.norms) nxt_n = g.norms[c0];
  }
  __syncthreads();
  for (int base = 0; base < n && s_len < cap; base += NW) {
    const int ci = base + w;
    const bool have = ci < n;
    float4 c[NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) c[k] = nxt[k];
    const float cn = nxt_n;
    const u64 ck = have ? list[ci] : 0ull;
    const u32 cid = (u32)(ck & 0xFFFFFFFFull), cdb = (u32)(ck >> 32);
    // A: publish this wave's candidate row for the intra-chunk tests, start the next one
    unsigned char *cb = L.cbuf + (size_t)w * g.row_stride;
    if (have && sub == 0) {
#pragma unroll
      for (int k = 0; k < NCH; k++) {
        u32 f = (u32)(k * LPR + t);
        if (f < g.n16) *reinterpret_cast<float4 *>(cb + (size_t)f * 16) = c[k];
      }
    }
    if (ci + NW < n) {
      const u32 nx = (u32)(list[ci + NW] & 0xFFFFFFFFull);
      load_row<LPR, NCH>(g.rows + (size_t)nx * g.row_stride, t, g.n16, nxt);
      if (g.norms) nxt_n = g.norms[nx];
    }
    // B: this wave's candidate against S (wave-local)
    bool viol = false;
    const int ngroups = (s_len + RPG - 1) / RPG;
    if (have) {
      // one pass = 4 load groups of S against the candidate.  Two loops: passes whose rows are all
      // staged in LDS touch no register a global load may still be writing, so they do not wait for
      // the next candidate's row (issued in A) — with the two sources mixed per row the compiler put
      // s_waitcnt vmcnt(0) in front of the first LDS read and that prefetch was exposed on every
      // chunk.  The rest of S (beyond L.SL rows, rare) comes from L2 / HBM in the second loop.
      auto pass = [&](const int g0, auto staged_tag) __attribute__((always_inline)) {
        constexpr bool STAGED = decltype(staged_tag)::value;
        const int gend = g0 + 4 < ngroups ? g0 + 4 : ngroups;
        float4 r[4][NCH];
        float pf[4];
        u32 pb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (g0 + j < gend) { // wave-uniform
            int ri = (g0 + j) * RPG + sub;
            if (ri > s_len - 1) ri = s_len - 1;
            if (STAGED || (g0 + j) * RPG < L.SL)
              load_row_lds<LPR, NCH>(L.stage + (size_t)ri * g.row_stride, t, g.n16, r[j]);
            else
              load_row<LPR, NCH>(g.rows + (size_t)L.s_ids[ri] * g.row_stride, t, g.n16, r[j]);
          } else if (!STAGED) {
#pragma unroll
            for (int k = 0; k < NCH; k++) r[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        // (fold4 never adds values of different rows: a staged pass of fewer than 4 groups skips the
        // missing rows' reads and arithmetic instead of working on zeros)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          pf[j] = 0.f;
          pb[j] = 0u;
          if (!STAGED || g0 + j < gend) {
            if (g.mclass == MC_BIN)
              pb[j] = partial_bin<NCH>(c, r[j]);
            else
              pf[j] = partial_f32<NCH>(g.mclass, c, r[j]);
          }
        }
        const int myri = (g0 + j4) * RPG + sub;
        const bool on = myri < s_len;
        const float rn = L.s_norm[on ? myri : 0];
        float d;
        if (g.mclass == MC_BIN)
          d = finalize_bin(g, fold4<LPR, u32>(pb[0], pb[1], pb[2], pb[3]), cn, rn);
        else
          d = finalize_f32(g, fold4<LPR, float>(pf[0], pf[1], pf[2], pf[3]), cn, rn);
        const float da = d * g.alpha; // hnsw.rs:585
        viol = ballot(on && fbits(da) < cdb) != 0ull;
        evals += (u64)((gend - g0) * RPG); // wave-uniform; lane 0's copy is the one that is added up
      };
      int g0 = 0;
      for (; g0 < ngroups && !viol && (g0 + 4 < ngroups ? g0 + 4 : ngroups) * RPG <= L.SL; g0 += 4)
        pass(g0, std::true_type{});
      for (; g0 < ngroups && !viol; g0 += 4) pass(g0, std::false_type{});
    }
    const bool surv = have && !viol;
    // Barriers: one per chunk when at most one candidate survived B and nothing is selected, a second
    // only around the intra-chunk tests (two or more survivors), a third only when S changed.  The
    // survivor flags alternate between two sets, so a wave that runs ahead into the next chunk never
    // overwrites flags a slower wave is still reading (they meet again at that chunk's first barrier).
    int *sv = L.surv + ((base / NW) & 1) * 8;
    if (ln == 0) {
      sv[w] = surv ? 1 : 0;
      L.cnorm[w] = cn;
    }
    __syncthreads();
    u32 svmask = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) svmask |= (sv[j] != 0 ? 1u : 0u) << j;
    svmask = (u32)__builtin_amdgcn_readfirstlane((int)svmask);
    const bool need_c = (svmask & (svmask - 1u)) != 0u; // two or more survivors
    // C: survivors against the earlier survivors of the chunk (folded passes over cbuf, 4 rows each)
    u32 vm = 0;
    if (need_c) {
      if (surv && (svmask & ((1u << w) - 1u)) != 0u) {
#pragma unroll
        for (int q4 = 0; q4 < NW / 4; q4++) {
          if (q4 * 4 < w) { // wave-uniform
            float4 r[4][NCH];
#pragma unroll
            for (int j = 0; j < 4; j++)
              load_row_lds<LPR, NCH>(L.cbuf + (size_t)(q4 * 4 + j) * g.row_stride, t, g.n16, r[j]);
            const int jm = q4 * 4 + j4;
            const float rn = L.cnorm[jm];
            float d;
            if (g.mclass == MC_BIN)
              d = finalize_bin(g, fold4<LPR, u32>(partial_bin<NCH>(c, r[0]), partial_bin<NCH>(c, r[1]),
                                                  partial_bin<NCH>(c, r[2]), partial_bin<NCH>(c, r[3])),
                               cn, rn);
            else
              d = finalize_f32(g, fold4<LPR, float>(partial_f32<NCH>(g.mclass, c, r[0]),
                                                    partial_f32<NCH>(g.mclass, c, r[1]),
                                                    partial_f32<NCH>(g.mclass, c, r[2]),
                                                    partial_f32<NCH>(g.mclass, c, r[3])),
                               cn, rn);
            const float da = d * g.alpha;
            const bool hit = sub == 0 && jm < w && ((svmask >> jm) & 1u) != 0u && fbits(da) < cdb;
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (ballot(hit && j4 == j) != 0ull) vm |= 1u << (q4 * 4 + j);
          }
        }
        evals += (u64)w;
      }
      if (ln == 0) L.vmask[w] = vm;
      __syncthreads();
    }
    // D: resolve the chunk in candidate order (every thread computes the same thing)
    u32 selmask = 0;
    int cnt = s_len;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const u32 vmj = need_c ? L.vmask[j] : 0u;
      if (cnt < cap && ((svmask >> j) & 1u) != 0u && (vmj & selmask) == 0u) { // :577-579, :583-592
        selmask |= 1u << j;
        cnt++;
      }
    }
    selmask = (u32)__builtin_amdgcn_readfirstlane((int)selmask);
    if ((selmask >> w) & 1u) {
      const int pos = s_len + __popc(selmask & ((1u << w) - 1u));
      if (ln == 0) {
        L.S[pos] = ck;
        L.s_ids[pos] = cid;
        L.s_norm[pos] = cn;
      }
      if (sub == 0 && pos < L.SL) {
#pragma unroll
        for (int k = 0; k < NCH; k++) {
          u32 f = (u32)(k * LPR + t);
          if (f < g.n16)
            *reinterpret_cast<float4 *>(L.stage + (size_t)pos * g.row_stride + (size_t)f * 16) = c[k];
        }
      }
    }
    s_len = s_len + __popc(selmask);
    if (selmask != 0u) __syncthreads(); // S changed: the next chunk's B reads it
  }
  return s_len;
}

// capmax: slots of the selected-set arrays = max(M, M0) rounded up to 64 (64, or up to HNY_BIG_CAP)
__host__ __device__ inline u32 wg_capmax(const GraphDev &g) {
  const u32 c = g.M0 > g.M ? g.M0 : g.M;
  return (c + 63u) / 64u * 64u;
}
__host__ __device__ inline size_t wg_prune_lds_bytes(u32 rcap, u32 row_stride, int SL, int NW, u32 capmax) {
  return (size_t)rcap * 8 + (size_t)capmax * (8 + 4 + 4) + 128 + (size_t)(SL + NW) * row_stride;
}

__device__ __forceinline__ WgPruneLds wg_prune_carve(unsigned char *base, int SL, u32 row_stride, int NW, u32 capmax) {
  WgPruneLds L;
  L.S = reinterpret_cast<u64 *>(base);
  L.s_ids = reinterpret_cast<u32 *>(L.S + capmax);
  L.s_norm = reinterpret_cast<float *>(L.s_ids + capmax);
  L.surv = reinterpret_cast<int *>(L.s_norm + capmax);
  L.vmask = reinterpret_cast<u32 *>(L.surv + 16);
  L.cnorm = reinterpret_cast<float *>(L.vmask + 8);
  L.cbuf = reinterpret_cast<unsigned char *>(L.cnorm + 8);
  L.stage = L.cbuf + (size_t)NW * row_stride;
  L.SL = SL;
  return L;
}

// rows up to 3 KB: 4 workgroups of 4 waves per CU is what the LDS carve allows, keep the registers there
constexpr int wg_waves_per_simd(int nch, int nw) { return nw == 4 && nch <= 3 ? 4 : 1; }

template <int LPR, int NCH, int NW, int SP>
__global__ __launch_bounds__(NW * 64, wg_waves_per_simd(NCH, NW)) void k_prune_wg(GraphDev g_in, PruneArgs a, int SL) {
  GraphDev g = g_in;
  specialize<SP>(g);
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *list = reinterpret_cast<u64 *>(smem);
  const bool list_global = SP == 0 && a.list_global != 0; // lists too long for LDS: pruned straight from HBM
  WgPruneLds L = wg_prune_carve(smem + (list_global ? (size_t)0 : (size_t)a.rcap * 8), SL, g.row_stride, NW, wg_capmax(g));
  const int tid = threadIdx.x;
  u64 evals = 0;
  // Workgroups go to the XCDs round robin, so with member = blockIdx + k * gridDim neighbouring members (locality
  // order: overlapping candidate sets) run at the same time on EIGHT different L2s.  xcd_tile != 0 (grid a multiple
  // of 8): the members come in tiles of xcd_tile, tile k belongs to XCD k % 8, and workgroup b takes position
  // b / 8 + k * gridDim / 8 of XCD b % 8's members — the workgroups resident on one XCD then prune neighbouring
  // queries and find each other's candidate rows in their L2.  Members beyond the last full round of tiles keep
  // the plain mapping.  Results are stored per member: only the traffic changes.
  const u32 cnt_all = a.hi - a.lo;
  const u32 T = a.xcd_tile && (gridDim.x & 7u) == 0u ? a.xcd_tile : 0u;
  const u32 tiled = T ? cnt_all / (8u * T) * (8u * T) : 0u;
  for (u32 v = blockIdx.x; v < cnt_all; v += gridDim.x) {
    u32 idx = v;
    if (v < tiled) {
      const u32 x = v & 7u, p = v >> 3;
      idx = ((p / T) * 8u + x) * T + (p % T);
    }
    const u32 mi = a.lo + idx;
    const u32 m = a.perm ? (u32)a.perm[mi - a.lo] : mi; // same locality order as the walk
    const int n = (int)a.cand_n[m];
    if (list_global) {
      list = const_cast<u64 *>(a.cand) + (size_t)m * a.rcap;
    } else {
      for (int e = tid; e < n; e += blockDim.x) list[e] = a.cand[(size_t)m * a.rcap + e];
    }
    __syncthreads();
    int s_len;
    if constexpr (SP != 0) {
      // rows that fill every lane's chunks (768-d, 1024-d, 128-d, 1024 bits ...): with n16 a
      // constant the per-chunk bounds guards of the row loads fold away (~20 % of a pass)
      if (g.n16 == (u32)(LPR * NCH) && g.row_stride == (u32)(LPR * NCH * 16)) {
        GraphDev gf = g;
        gf.n16 = (u32)(LPR * NCH);
        gf.row_stride = (u32)(LPR * NCH * 16);
        s_len = wg_prune<LPR, NCH, NW>(gf, list, n, (int)a.cap, L, evals);
      } else {
        s_len = wg_prune<LPR, NCH, NW>(g, list, n, (int)a.cap, L, evals);
      }
    } else {
      s_len = wg_prune<LPR, NCH, NW>(g, list, n, (int)a.cap, L, evals);
    }
    u64 *out = a.sel + (size_t)m * a.sel_stride + (size_t)(a.batch_level - a.layer) * (a.cap_sel + 1);
    if (tid == 0) out[0] = (u64)s_len;
    for (int e = tid; e < s_len; e += (int)blockDim.x) out[1 + e] = L.S[e];
    __syncthreads();
  }
  if ((tid & 63) == 0 && evals) atomicAdd(&g.stats[ST_EVALS_PRUNE], evals);
}

// ---------------------------------------------------------------------------------------------
// robust_prune for SHORT rows (at most 32 sixteen-byte units = 512 B: 128-d f32, 1024-bit codes): one
// wave per query and EIGHT candidates at a time, one per 8-lane group.  k_prune_wg gives every candidate
// a whole wave and pays 1-3 workgroup barriers per 4 candidates: on C4 / C5 it sat 5x / 15x above the
// time its candidate rows take to stream (384 ms and 107 ms of those builds).  Here a chunk is
//   (1) eight candidate rows in registers (one wave-wide load brings eight 128-B rows), prefetched a
//       chunk ahead;
//   (2) every group scores its candidate against the selected rows, which all groups read from the
//       wave's LDS stage at the same address (a broadcast), until every group has its answer;
//   (3) the survivors are taken in candidate order: the first joins S, the others are scored against
//       that one new row, and so on — `exists i in S: bits(d(c, i) * alpha) < bits(d(c, q))` does not
//       depend on the order in which S is scanned, so this is the sequential outcome (hnsw.rs:577-592).
// No workgroup barrier anywhere; the wave order of a 16- / 32-lane row is computed by 8 lanes exactly
// as dist_rows_narrow does (unit partials added in the butterfly's own order).
// ---------------------------------------------------------------------------------------------
template <int LPRO>
__device__ __forceinline__ float dist8(const GraphDev &g, const float4 (&c)[LPRO / 8], const float4 (&r)[LPRO / 8],
                                       float cn, float rn) {
  constexpr int NQ = LPRO / 8;
  if (g.mclass == MC_BIN) return finalize_bin(g, butterfly_u32<8>(partial_bin<NQ>(c, r)), cn, rn);
  float pu[NQ];
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const float4 ck[1] = {c[k]}, rk[1] = {r[k]};
    pu[k] = partial_f32<1>(g.mclass, ck, rk);
  }
  float pa;
  if constexpr (NQ == 4) pa = (pu[0] + pu[2]) + (pu[1] + pu[3]); // off = 16, then off = 8
  else if constexpr (NQ == 2) pa = pu[0] + pu[1];                // off = 8
  else pa = pu[0];
  return finalize_f32(g, butterfly_f32<8>(pa), cn, rn);
}
__host__ __device__ inline size_t prune_n8_lds_bytes(u32 rcap, u32 row_stride, int SL) {
  return (size_t)rcap * 8 + (size_t)HNY_MAX_CAP * (8 + 4 + 4) + (size_t)(SL + 1) * row_stride;
}
__host__ __device__ inline size_t apply_n8_lds_bytes(u32 row_stride, int SL) {
  return (size_t)HNY_MAX_CAP * (8 + 8 + 8 + 4 + 4) + (size_t)(SL + 1) * row_stride;
}

// the prune of one sorted list by one wave (see above); S / s_ids / s_norm: HNY_MAX_CAP entries, stage: SL rows,
// newrow: one row.  Returns the number of selected entries, left in S.  `list` (LDS, owned by the caller) is
// compacted in place by the prefix filter.
//
// Prefix filter (round 5).  A chunk of eight candidates lasts until its LAST member is decided, and a member that
// survives scans all of S — so the groups whose candidate fell to S[0] or S[1] (most candidates do: the reference
// counts ~4 tests per candidate at C4) idle for the rest of the chunk: k_prune_n8 issued 22.7 vector instructions
// per counted evaluation where a full step needs ~5.  Once S holds K0 >= 2 rows, the remaining candidates are
// therefore first run past those K0 rows alone — one or two steps per chunk of eight, every group busy, rows
// streamed a chunk ahead — and only the survivors (compacted in place, order kept) go through the chunks, starting
// at row K0.  A candidate's tests are the same tests in the same order (rows 0 .. K0-1 now, the rest later): same
// selection, same evaluation count; a survivor's row is fetched a second time.
#ifndef HNY_PRUNE_FILTER
#define HNY_PRUNE_FILTER 1
#endif
#ifndef HNY_PRUNE_FILTER_MIN
#define HNY_PRUNE_FILTER_MIN 40 // candidates left for the filter to be worth its second fetch
#endif
#ifndef HNY_PRUNE_TOUCH
#define HNY_PRUNE_TOUCH 0
#endif
#ifndef HNY_PRUNE_FILTER_ROWS
#define HNY_PRUNE_FILTER_ROWS 4 // rows of S the filter tests (the closest selected neighbours reject the most; 2: C4 prune 0.197 s, 4: 0.189, all: 0.190)
#endif
template <int LPRO>
__device__ __forceinline__ int prune_n8_core(const GraphDev &g, u64 *list, int n, int cap, u64 *S, u32 *s_ids,
                                             float *s_norm, unsigned char *stage, unsigned char *newrow, int SL,
                                             u64 &evals) {
  constexpr int NQ = LPRO / 8;
  const int ln = HNY_LANE, t = ln & 7, gidx = ln >> 3;
  auto load8 = [&](const unsigned char *row, float4 (&r)[NQ]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const u32 f = (u32)(k * 8 + t);
      r[k] = f < g.n16 ? *reinterpret_cast<const float4 *>(row + (size_t)f * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load8_lds = [&](const unsigned char *row, float4 (&r)[NQ]) __attribute__((always_inline)) {
    const unsigned lp = (unsigned)(size_t)row;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const u32 f = (u32)(k * 8 + t);
      if (f < g.n16) {
        const f32x4_t v = *reinterpret_cast<lds_cf4 *>((size_t)(lp + f * 16u));
        r[k] = make_float4(v.x, v.y, v.z, v.w);
      } else {
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  int s_len = 0;
  float4 nxt[NQ];
  float nxt_n = 0.f;
  // the row of candidate ci (this group's member of the chunk after next) on its way into `nxt`
  u32 touched = 0u; // (sink of the touch loads below: never true, keeps them alive)
  auto prefetch = [&](int ci) __attribute__((always_inline)) {
    if (ci < n) {
      const u32 nx = (u32)(list[ci] & 0xFFFFFFFFull);
      load8(g.rows + (size_t)nx * g.row_stride, nxt);
      if (g.norms) nxt_n = g.norms[nx];
    }
#if HNY_PRUNE_TOUCH
    // and the row of the candidate one chunk further on its way into the L2: one dword per 128-B line (lane t of
    // the group touches line t), so that the load above finds it there a chunk later instead of in HBM
    if (ci + 8 < n) {
      const u32 n2 = (u32)(list[ci + 8] & 0xFFFFFFFFull);
      const u32 off = (u32)t * 128u;
      if (off < g.row_stride) touched |= *reinterpret_cast<const u32 *>(g.rows + (size_t)n2 * g.row_stride + off) & 0x7FC00000u;
    }
#endif
  };
#pragma unroll
  for (int k = 0; k < NQ; k++) nxt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  prefetch(gidx);
  // one chunk: candidates [base, base + 8) against S rows [row0, s_len), then its survivors in candidate order
  auto chunk = [&](const int base, const int row0) __attribute__((always_inline)) {
    const int ci = base + gidx;
    const bool have = ci < n;
    float4 c[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) c[k] = nxt[k];
    const float cn = nxt_n;
    const u64 ck = have ? list[ci] : 0ull;
    const u32 cid = (u32)(ck & 0xFFFFFFFFull), cdb = (u32)(ck >> 32);
    prefetch(ci + 8); // the next chunk's rows travel while this one is scored
    // (2) against S as it stands
    // two selected rows per step: `exists i in S` does not care about the order, and the two dependent chains
    // of a step (LDS read -> fma chain -> 8-lane butterfly -> finaliser -> compare) then overlap inside the wave —
    // this loop is bound by their latency, not by issue (k_prune_n8: vector port 0.32 busy)
    bool viol = false;
    for (int j = row0; j < s_len; j += 2) {
      const u64 open = ballot(have && !viol);
      if (!open) break;
      const bool two = j + 1 < s_len; // wave-uniform
      float4 r0[NQ], r1[NQ];
      if (j < SL) load8_lds(stage + (size_t)j * g.row_stride, r0);
      else load8(g.rows + (size_t)s_ids[j] * g.row_stride, r0); // beyond the stage (rare): from L2
      if (two) {
        if (j + 1 < SL) load8_lds(stage + (size_t)(j + 1) * g.row_stride, r1);
        else load8(g.rows + (size_t)s_ids[j + 1] * g.row_stride, r1);
      }
      const float d0 = dist8<LPRO>(g, c, r0, cn, s_norm[j]);
      bool v = fbits(d0 * g.alpha) < cdb; // hnsw.rs:585
      // the counter is the reference's count (a candidate stops at its first violating row), not the work done:
      // row j + 1 is computed for every open candidate, but counted only for those row j did not reject
      u64 counted = open;
      if (two) {
        counted = ballot(have && !viol && !v);
        const float d1 = dist8<LPRO>(g, c, r1, cn, s_norm[j + 1]);
        v = v || fbits(d1 * g.alpha) < cdb;
      }
      viol = viol || v;
      evals += (u64)(__popcll(open) >> 3) + (two ? (u64)(__popcll(counted) >> 3) : 0ull);
    }
    // (3) the survivors, in candidate order
    u64 sv = ballot(have && !viol && t == 0);
    while (sv && s_len < cap) {
      const int l1 = __ffsll((long long)sv) - 1, g1 = l1 >> 3;
      sv &= sv - 1ull;
      if (ln == l1) { // :591 S.push(c)
        S[s_len] = ck;
        s_ids[s_len] = cid;
        s_norm[s_len] = cn;
      }
      if (gidx == g1) {
#pragma unroll
        for (int k = 0; k < NQ; k++) {
          const u32 f = (u32)(k * 8 + t);
          if (f < g.n16) {
            *reinterpret_cast<float4 *>(newrow + (size_t)f * 16) = c[k];
            if (s_len < SL) *reinterpret_cast<float4 *>(stage + (size_t)s_len * g.row_stride + (size_t)f * 16) = c[k];
          }
        }
      }
      WSYNC();
      if (sv) { // the later survivors against the row that has just joined
        float4 r[NQ];
        load8_lds(newrow, r);
        const float d = dist8<LPRO>(g, c, r, cn, s_norm[s_len]);
        const bool out = fbits(d * g.alpha) < cdb;
        evals += (u64)__popcll(sv);
        sv &= ~ballot(out && t == 0);
        WSYNC(); // newrow is rewritten by the next selection
      }
      s_len++;
    }
  };
  int base = 0;
  for (; base < n && s_len < cap; base += 8) {
    if (HNY_PRUNE_FILTER && s_len >= 2 && n - base >= HNY_PRUNE_FILTER_MIN) break;
    chunk(base, 0);
  }
  if (base < n && s_len < cap) {
    // ---- the filter: candidates [base, n) against rows [0, K0) only; survivors compacted to list[base ..)
    const int K0 = s_len < HNY_PRUNE_FILTER_ROWS ? s_len : HNY_PRUNE_FILTER_ROWS;
    int w = base;
    for (int b2 = base; b2 < n; b2 += 8) {
      const int ci = b2 + gidx;
      const bool have = ci < n;
      float4 c[NQ];
#pragma unroll
      for (int k = 0; k < NQ; k++) c[k] = nxt[k];
      const float cn = nxt_n;
      const u64 ck = have ? list[ci] : 0ull;
      const u32 cdb = (u32)(ck >> 32);
      prefetch(ci + 8);
      bool viol = false;
      for (int j = 0; j < K0; j += 2) {
        const u64 open = ballot(have && !viol);
        if (!open) break;
        const bool two = j + 1 < K0; // wave-uniform
        float4 r0[NQ], r1[NQ];
        if (j < SL) load8_lds(stage + (size_t)j * g.row_stride, r0);
        else load8(g.rows + (size_t)s_ids[j] * g.row_stride, r0);
        if (two) {
          if (j + 1 < SL) load8_lds(stage + (size_t)(j + 1) * g.row_stride, r1);
          else load8(g.rows + (size_t)s_ids[j + 1] * g.row_stride, r1);
        }
        const float d0 = dist8<LPRO>(g, c, r0, cn, s_norm[j]);
        bool v = fbits(d0 * g.alpha) < cdb; // hnsw.rs:585
        u64 counted = open;
        if (two) {
          counted = ballot(have && !viol && !v);
          const float d1 = dist8<LPRO>(g, c, r1, cn, s_norm[j + 1]);
          v = v || fbits(d1 * g.alpha) < cdb;
        }
        viol = viol || v;
        evals += (u64)(__popcll(open) >> 3) + (two ? (u64)(__popcll(counted) >> 3) : 0ull);
      }
      const bool keep = have && !viol && t == 0;
      const u64 sv = ballot(keep);
      // (every lane read its list entries above; the writes land at or below the chunk's own first index)
      if (keep) list[w + __popcll(sv & ((1ull << ln) - 1ull))] = ck;
      w += __popcll(sv);
    }
    WSYNC();
    n = w;
    prefetch(base + gidx);
    // ---- the survivors through the chunks: rows [0, K0) are behind them
    for (; base < n && s_len < cap; base += 8) chunk(base, K0);
  }
#if HNY_PRUNE_TOUCH
  if (touched == 0x7FC00001u) evals++; // (never: bit 0 is masked off)
#endif
  return s_len;
}

template <int LPRO, int SP>
__global__ __launch_bounds__(64, 5) void k_prune_n8(GraphDev g_in, PruneArgs a, int SL) {
  static_assert(LPRO == 8 || LPRO == 16 || LPRO == 32, "rows of at most 32 units, 8 lanes each");
  GraphDev g = g_in;
  specialize<SP>(g);
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *list = reinterpret_cast<u64 *>(smem);
  u64 *S = list + a.rcap;
  u32 *s_ids = reinterpret_cast<u32 *>(S + HNY_MAX_CAP);
  float *s_norm = reinterpret_cast<float *>(s_ids + HNY_MAX_CAP);
  unsigned char *stage = reinterpret_cast<unsigned char *>(s_norm + HNY_MAX_CAP); // [SL] selected rows
  unsigned char *newrow = stage + (size_t)SL * g.row_stride;                      // the row that has just joined S
  const int ln = HNY_LANE;
  u64 evals = 0;
  for (u32 mi = a.lo + blockIdx.x; mi < a.hi; mi += gridDim.x) {
    const u32 m = a.perm ? (u32)a.perm[mi - a.lo] : mi; // same locality order as the walk
    const int n = (int)a.cand_n[m];
    for (int e = ln; e < n; e += 64) list[e] = a.cand[(size_t)m * a.rcap + e];
    WSYNC();
    const int s_len = prune_n8_core<LPRO>(g, list, n, (int)a.cap, S, s_ids, s_norm, stage, newrow, SL, evals);
    u64 *out = a.sel + (size_t)m * a.sel_stride + (size_t)(a.batch_level - a.layer) * (a.cap_sel + 1);
    if (ln == 0) out[0] = (u64)s_len;
    if (ln < s_len) out[1 + ln] = S[ln];
    WSYNC();
  }
  if (ln == 0 && evals) atomicAdd(&g.stats[ST_EVALS_PRUNE], evals);
}

// add_link for the targets whose list overflows (hnsw.rs:547-552), short rows: one WAVE per target, the
// self-prune on prune_n8_core (k_apply_wg: a 256-thread workgroup per target around wg_prune)
template <int LPRO, int SP>
__global__ __launch_bounds__(64, 5) void k_apply_n8(GraphDev g_in, ApplyArgs a, int SL) {
  GraphDev g = g_in;
  specialize<SP>(g);
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *lk = reinterpret_cast<u64 *>(smem); // [HNY_MAX_CAP] the node's list
  u64 *sorted = lk + HNY_MAX_CAP;          // [HNY_MAX_CAP]
  u64 *S = sorted + HNY_MAX_CAP;
  u32 *s_ids = reinterpret_cast<u32 *>(S + HNY_MAX_CAP);
  float *s_norm = reinterpret_cast<float *>(s_ids + HNY_MAX_CAP);
  unsigned char *stage = reinterpret_cast<unsigned char *>(s_norm + HNY_MAX_CAP);
  unsigned char *newrow = stage + (size_t)SL * g.row_stride;
  const int ln = HNY_LANE;
  const u32 n_def = *a.n_deferred;
  u64 evals = 0;
  const u32 sw = a.shard_world ? a.shard_world : 1u;
  for (u32 di = a.shard_rank + blockIdx.x * sw; di < n_def; di += gridDim.x * sw) {
    const u32 i0 = a.deferred[di];
    const u64 k0 = a.keys[i0] >> HNY_SEQ_BITS;
    const u32 target = (u32)(k0 & 0x7FFFFFFFull), layer = (u32)(k0 >> 31);
    u32 cap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)target * g.M0;
      dist = g.l0_dist + (size_t)target * g.M0;
      cntp = g.l0_cnt + target;
    } else {
      size_t u = (size_t)g.upper_idx[target] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    const u32 cw = *cntp;
    int cnt = (int)(cw & 0xFFFFu);
    bool frozen = (cw >> 31) != 0u;
    if (ln < cnt) lk[ln] = ((u64)fbits(dist[ln]) << 32) | ids[ln]; // cap <= 64: one entry per lane
    WSYNC();
    for (u32 i = i0; i < a.n_ops && !frozen; i++) {
      const u64 key = a.keys[i];
      if (key == HNY_OP_INVALID || (key >> HNY_SEQ_BITS) != k0) break;
      const u64 val = a.vals[i];
      if ((u32)(val & 0xFFFFFFFFull) == target) continue; // hnsw.rs:530
      if (cnt < (int)cap) {                               // :542-545
        if (ln == 0) lk[cnt] = val;
        cnt++;
        WSYNC();
      } else { // :547-552: the new link is dropped, the full list prunes itself
        const u64 mine = ln < cnt ? lk[ln] : 0ull;
        int rk = 0;
        for (int j = 0; j < cnt; j++) {
          const u64 o = lk[j];
          rk += (o < mine || (o == mine && j < ln)) ? 1 : 0;
        }
        if (ln < cnt) sorted[rk] = mine;
        WSYNC();
        const int s_len = prune_n8_core<LPRO>(g, sorted, cnt, (int)cap, S, s_ids, s_norm, stage, newrow, SL, evals);
        if (ln < s_len) lk[ln] = S[ln];
        cnt = s_len;
        frozen = (s_len == (int)cap); // a full list that prunes to itself can never change again
        WSYNC();
      }
    }
    if ((u32)ln < cap) {
      const bool on = ln < cnt;
      ids[ln] = on ? (u32)(lk[ln] & 0xFFFFFFFFull) : HNY_SENT;
      dist[ln] = on ? __uint_as_float((u32)(lk[ln] >> 32)) : 0.f;
    }
    if (ln == 0) *cntp = (u32)cnt | (frozen ? 0x80000000u : 0u);
    if (a.exch) { // the finished list, for the ranks that did not compute it
      u64 *rec = a.exch + (size_t)(di / sw) * a.exch_stride;
      if (ln == 0) {
        rec[0] = k0;
        rec[1] = (u64)((u32)cnt | (frozen ? 0x80000000u : 0u));
      }
      if (ln < cnt) rec[2 + ln] = lk[ln];
    }
    WSYNC();
  }
  if (ln == 0 && evals) atomicAdd(&g.stats[ST_EVALS_APPLY], evals);
}

// ---------------------------------------------------------------------------------------------
// link ops.  For batch member m (in batch order), layer l from its level down to 0, k-th selected
// (d, n): LINK(q,(d,n),l) then LINK(n,(d,q),l) (hnsw.rs:316-324).  key = layer:4 | target:31 |
// seq:29 — a full 64-bit sort groups ops by target and keeps the reference's sequential order
// inside each target; ops on different targets commute.
// ---------------------------------------------------------------------------------------------
__global__ void k_emit(GraphDev g, EmitArgs a) {
  const u32 n_layers = a.batch_level + 1;
  const u64 total = (u64)a.count * n_layers * a.cap_sel;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 links = 0;
  // grid-stride: one element per thread meant 17 k workgroups per batch, and launching them was
  // the kernel's whole duration; both ops of a pair go out as one 16-byte store
  for (u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    u32 k = (u32)(idx % a.cap_sel);
    u32 li = (u32)((idx / a.cap_sel) % n_layers);
    u32 m = (u32)(idx / ((u64)a.cap_sel * n_layers));
    const u64 *s = a.sel + (size_t)m * a.sel_stride + (size_t)li * (a.cap_sel + 1);
    u64 o = idx * 2;
    ulonglong2 kk, vv;
    if (k < (u32)s[0]) {
      u64 e = s[1 + k];
      u64 nb = e & 0xFFFFFFFFull, db = e >> 32;
      u64 q = a.q_slots[m];
      u64 layer = (u64)(a.batch_level - li);
      kk.x = (layer << (31 + HNY_SEQ_BITS)) | (q << HNY_SEQ_BITS) | o;
      vv.x = (db << 32) | nb;
      kk.y = (layer << (31 + HNY_SEQ_BITS)) | (nb << HNY_SEQ_BITS) | (o + 1);
      vv.y = (db << 32) | q;
      *reinterpret_cast<ulonglong2 *>(a.vals + o) = vv;
      links += 2; // build_stats.incr_link_count(2), hnsw.rs:323
    } else {
      kk.x = HNY_OP_INVALID;
      kk.y = HNY_OP_INVALID;
    }
    *reinterpret_cast<ulonglong2 *>(a.keys + o) = kk;
  }
  // one atomic per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) links += (u64)__shfl_xor((long long)links, off, 64);
  if ((threadIdx.x & 63) == 0 && links) atomicAdd(&g.stats[ST_LINKS], links);
}

// first op of every (layer, target) segment of the sorted link ops -> seg_start (any order: targets are
// independent).  1 024 ops per workgroup and step, ONE slot reservation per workgroup and step: with a
// reservation per wave the 65 k atomics on the one counter were the kernel's whole duration (143 us
// per 4.2 M ops at C2; 4 k of them: the kernel runs at the speed of its 34 MB read).
__global__ __launch_bounds__(256) void k_segments(const u64 *keys, u32 n_ops, u32 *seg_start, u32 *n_seg) {
  __shared__ u32 wsum[4];
  __shared__ u32 base_s;
  const int tid = threadIdx.x, ln = tid & 63, w = tid >> 6;
  for (u32 b0 = blockIdx.x * 1024u; b0 < n_ops; b0 += gridDim.x * 1024u) { // b0: uniform per workgroup
    const u32 i0 = b0 + (u32)tid * 4u;
    u64 k[5];
    k[0] = (i0 > 0u && i0 <= n_ops) ? keys[i0 - 1] : HNY_OP_INVALID;
#pragma unroll
    for (int j = 0; j < 4; j++) k[j + 1] = i0 + (u32)j < n_ops ? keys[i0 + j] : HNY_OP_INVALID;
    u32 flags = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool start = k[j + 1] != HNY_OP_INVALID &&
                         ((i0 + (u32)j) == 0u || (k[j] >> HNY_SEQ_BITS) != (k[j + 1] >> HNY_SEQ_BITS));
      flags |= (start ? 1u : 0u) << j;
    }
    const u32 c = (u32)__popc(flags);
    u32 incl = c; // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const u32 t = (u32)__shfl_up((int)incl, off, 64);
      if (ln >= off) incl += t;
    }
    if (ln == 63) wsum[w] = incl;
    __syncthreads();
    u32 wbase = 0u, total = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < w) wbase += wsum[j];
      total += wsum[j];
    }
    if (tid == 0) base_s = total ? atomicAdd(n_seg, total) : 0u;
    __syncthreads();
    u32 pos = base_s + wbase + incl - c;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if ((flags >> j) & 1u) seg_start[pos++] = i0 + (u32)j;
    __syncthreads();
  }
}

// add_link (hnsw.rs:523-560) for the segments whose list cannot overflow — nearly all of them: one
// THREAD per (layer, target) appends the segment's links in order (:542-545, no dedup; :530 p == q is
// a no-op).  A segment that may overflow is left, whole, to k_apply_wg (`deferred`).  The one-wave-
// per-segment k_apply below spent 295 us per C2 batch walking ~2 M segments of 1-2 ops each.
__global__ __launch_bounds__(256) void k_apply_append(GraphDev g, ApplyArgs a) {
  const u32 n_seg = *a.n_seg;
  for (u32 sg = blockIdx.x * blockDim.x + threadIdx.x; sg < n_seg; sg += gridDim.x * blockDim.x) {
    const u32 i0 = a.seg_start[sg];
    const u64 k0 = a.keys[i0] >> HNY_SEQ_BITS;
    const u32 target = (u32)(k0 & 0x7FFFFFFFull), layer = (u32)(k0 >> 31);
    u32 cap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)target * g.M0;
      dist = g.l0_dist + (size_t)target * g.M0;
      cntp = g.l0_cnt + target;
    } else { // :534 `layers.get(level)` — the target always has this layer (it was found on it)
      const size_t u = (size_t)g.upper_idx[target] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    const u32 cw = *cntp;
    if (cw >> 31) continue; // frozen: full and self-pruned to full, nothing can change it (see k_apply)
    u32 cnt = cw & 0xFFFFu;
    u32 nops = 0;
    for (u32 i = i0; i < a.n_ops; i++) {
      const u64 key = a.keys[i];
      if (key == HNY_OP_INVALID || (key >> HNY_SEQ_BITS) != k0) break;
      nops++;
      if (cnt + nops > cap) break;
    }
    if (cnt + nops > cap) { // may overflow: k_apply_wg replays the whole segment
      a.deferred[atomicAdd(a.n_deferred, 1u)] = i0;
      continue;
    }
    for (u32 i = i0; i < i0 + nops; i++) {
      const u64 val = a.vals[i];
      if ((u32)(val & 0xFFFFFFFFull) == target) continue; // :530 p == q.1
      ids[cnt] = (u32)(val & 0xFFFFFFFFull);
      dist[cnt] = __uint_as_float((u32)(val >> 32));
      cnt++;
    }
    *cntp = cnt;
  }
}

// add_link (hnsw.rs:523-560) for every op of one (layer, target), in order.
template <int LPR, int NCH, int SP>
__global__ __launch_bounds__(64) void k_apply(GraphDev g_in, ApplyArgs a) {
  GraphDev g = g_in;
  specialize<SP>(g);
  // (lists of more than 64 slots — strict mode with M0 up to HNY_BIG_CAP, the reference's fuzz pair (16, 768) in
  // the x86 order — are held whole and taken 64 slots at a time)
  extern __shared__ __align__(16) unsigned char smem[];
  const u32 capmax = wave_capmax(g);
  u64 *lk = reinterpret_cast<u64 *>(smem); // the node's list: dist bits << 32 | slot
  u64 *sorted = lk + capmax;
  u64 *S = sorted + capmax;
  u32 *s_ids = reinterpret_cast<u32 *>(S + capmax);
  float *tmp_d = reinterpret_cast<float *>(s_ids + capmax);
  const int ln = threadIdx.x;
  const u32 n_seg = *a.n_seg;
  u64 evals = 0;
  for (u32 sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
    const u32 i0 = a.seg_start[sg];
    const u64 k0 = a.keys[i0] >> HNY_SEQ_BITS;
    const u32 target = (u32)(k0 & 0x7FFFFFFFull), layer = (u32)(k0 >> 31);
    u32 cap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)target * g.M0;
      dist = g.l0_dist + (size_t)target * g.M0;
      cntp = g.l0_cnt + target;
    } else { // :534 `layers.get(level)` — the target always has this layer (it was found on it)
      size_t u = (size_t)g.upper_idx[target] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    u32 cw = *cntp;
    int cnt = (int)(cw & 0xFFFFu);
    bool frozen = (cw >> 31) != 0u;
    // a frozen list never changes again (see below): nothing to do for any number of ops
    if (frozen) continue;
    if (a.deferred) {
      // count the segment's ops; if the list can overflow, leave the whole segment to k_apply_wg
      u32 nops = 0;
      for (u32 i = i0; i < a.n_ops; i++) {
        const u64 key = a.keys[i];
        if (key == HNY_OP_INVALID || (key >> HNY_SEQ_BITS) != k0) break;
        nops++;
        if (cnt + (int)nops > (int)cap) break;
      }
      if (cnt + (int)nops > (int)cap) {
        if (ln == 0) a.deferred[atomicAdd(a.n_deferred, 1u)] = i0;
        continue;
      }
    }
    for (int e = ln; e < cnt; e += 64) lk[e] = ((u64)fbits(dist[e]) << 32) | ids[e];
    WSYNC();
    for (u32 i = i0; i < a.n_ops; i++) {
      const u64 key = a.keys[i];
      if (key == HNY_OP_INVALID || (key >> HNY_SEQ_BITS) != k0) break;
      const u64 val = a.vals[i];
      if ((u32)(val & 0xFFFFFFFFull) == target) continue; // :530 p == q.1
      if (cnt < (int)cap) {                               // :542-545 append, no dedup
        if (ln == 0) lk[cnt] = val;
        cnt++;
        WSYNC();
      } else if (!frozen) { // :547-552 full: self-prune, the new link is dropped
        for (int e = ln; e < cnt; e += 64) { // stable rank sort by (distance bits, slot), ties in list order
          const u64 mine = lk[e];
          int rk = 0;
          for (int j = 0; j < cnt; j++) {
            const u64 o = lk[j];
            rk += (o < mine || (o == mine && j < e)) ? 1 : 0;
          }
          sorted[rk] = mine;
        }
        WSYNC();
        int s_len = wave_prune<LPR, NCH>(g, sorted, cnt, (int)cap, S, s_ids, tmp_d, evals);
        for (int e = ln; e < s_len; e += 64) lk[e] = S[e];
        cnt = s_len;
        // a full list that prunes to itself can never change again: every later add_link would
        // redo the same prune with the same outcome
        frozen = (s_len == (int)cap);
        WSYNC();
      }
    }
    for (u32 e = (u32)ln; e < cap; e += 64) {
      const bool on = (int)e < cnt;
      ids[e] = on ? (u32)(lk[e] & 0xFFFFFFFFull) : HNY_SENT;
      dist[e] = on ? __uint_as_float((u32)(lk[e] >> 32)) : 0.f;
    }
    if (ln == 0) *cntp = (u32)cnt | (frozen ? 0x80000000u : 0u);
    WSYNC();
  }
  if (ln == 0 && evals) atomicAdd(&g.stats[ST_EVALS_APPLY], evals);
}

// add_link for the segments k_apply deferred (their list overflows): 256 threads per segment, the
// self-prune runs on the LDS-staged wg_prune.
template <int LPR, int NCH, int SP>
__global__ __launch_bounds__(256, wg_waves_per_simd(NCH, 4)) void k_apply_wg(GraphDev g_in, ApplyArgs a, int SL) {
  GraphDev g = g_in;
  specialize<SP>(g);
  extern __shared__ __align__(16) unsigned char smem[];
  const u32 capmax = wg_capmax(g);
  u64 *lk = reinterpret_cast<u64 *>(smem);          // [capmax] the node's list
  u64 *sorted = lk + capmax;                        // [capmax]
  WgPruneLds L = wg_prune_carve(smem + (size_t)2 * capmax * 8, SL, g.row_stride, 4, capmax);
  const int tid = threadIdx.x;
  const u32 n_def = *a.n_deferred;
  u64 evals = 0;
  const u32 sw = a.shard_world ? a.shard_world : 1u;
  for (u32 di = a.shard_rank + blockIdx.x * sw; di < n_def; di += gridDim.x * sw) {
    const u32 i0 = a.deferred[di];
    const u64 k0 = a.keys[i0] >> HNY_SEQ_BITS;
    const u32 target = (u32)(k0 & 0x7FFFFFFFull), layer = (u32)(k0 >> 31);
    u32 cap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)target * g.M0;
      dist = g.l0_dist + (size_t)target * g.M0;
      cntp = g.l0_cnt + target;
    } else {
      size_t u = (size_t)g.upper_idx[target] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    const u32 cw = *cntp;
    int cnt = (int)(cw & 0xFFFFu);
    bool frozen = (cw >> 31) != 0u;
    for (int e = tid; e < cnt; e += 256) lk[e] = ((u64)fbits(dist[e]) << 32) | ids[e];
    __syncthreads();
    for (u32 i = i0; i < a.n_ops && !frozen; i++) {
      const u64 key = a.keys[i];
      if (key == HNY_OP_INVALID || (key >> HNY_SEQ_BITS) != k0) break;
      const u64 val = a.vals[i];
      if ((u32)(val & 0xFFFFFFFFull) == target) continue; // hnsw.rs:530
      if (cnt < (int)cap) {                               // :542-545
        if (tid == 0) lk[cnt] = val;
        cnt++;
        __syncthreads();
      } else { // :547-552
        for (int e = tid; e < cnt; e += 256) { // (cnt <= 256: one entry per thread)
          const u64 mine = lk[e];
          int rk = 0;
          for (int j = 0; j < cnt; j++) {
            u64 o = lk[j];
            rk += (o < mine || (o == mine && j < e)) ? 1 : 0;
          }
          sorted[rk] = mine;
        }
        __syncthreads();
        int s_len;
        if (SP != 0 && g.n16 == (u32)(LPR * NCH) && g.row_stride == (u32)(LPR * NCH * 16)) { // full rows: see k_prune_wg
          GraphDev gf = g;
          gf.n16 = (u32)(LPR * NCH);
          gf.row_stride = (u32)(LPR * NCH * 16);
          s_len = wg_prune<LPR, NCH, 4>(gf, sorted, cnt, (int)cap, L, evals);
        } else {
          s_len = wg_prune<LPR, NCH, 4>(g, sorted, cnt, (int)cap, L, evals);
        }
        for (int e = tid; e < s_len; e += 256) lk[e] = L.S[e];
        cnt = s_len;
        frozen = (s_len == (int)cap);
        __syncthreads();
      }
    }
    for (u32 e = (u32)tid; e < cap; e += 256u) {
      const bool on = (int)e < cnt;
      ids[e] = on ? (u32)(lk[e] & 0xFFFFFFFFull) : HNY_SENT;
      dist[e] = on ? __uint_as_float((u32)(lk[e] >> 32)) : 0.f;
    }
    if (tid == 0) *cntp = (u32)cnt | (frozen ? 0x80000000u : 0u);
    if (a.exch) { // the finished list, for the ranks that did not compute it
      u64 *rec = a.exch + (size_t)(di / sw) * a.exch_stride;
      if (tid == 0) {
        rec[0] = k0;
        rec[1] = (u64)((u32)cnt | (frozen ? 0x80000000u : 0u));
      }
      for (int e = tid; e < cnt; e += 256) rec[2 + e] = lk[e];
    }
    __syncthreads();
  }
  if ((tid & 63) == 0 && evals) atomicAdd(&g.stats[ST_EVALS_APPLY], evals);
}

// multi-GPU: the lists other ranks finished in their share of the deferred segments (k_apply_wg's
// exch records, all-gathered) written into this rank's replica.  One wave per record.
__global__ __launch_bounds__(64) void k_apply_merge(GraphDev g, const u64 *exch, u32 n_def, u32 world, u32 rank,
                                                    u32 per, u32 stride) {
  const int ln = threadIdx.x;
  for (u32 di = blockIdx.x; di < n_def; di += gridDim.x) {
    const u32 owner = di % world;
    if (owner == rank) continue;
    const u64 *rec = exch + ((size_t)owner * per + di / world) * stride;
    const u64 k0 = rec[0];
    const u32 target = (u32)(k0 & 0x7FFFFFFFull), layer = (u32)(k0 >> 31);
    const u32 cw = (u32)rec[1];
    const int cnt = (int)(cw & 0xFFFFu);
    u32 cap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)target * g.M0;
      dist = g.l0_dist + (size_t)target * g.M0;
      cntp = g.l0_cnt + target;
    } else {
      size_t u = (size_t)g.upper_idx[target] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    for (u32 e0 = (u32)ln; e0 < cap; e0 += 64u) {
      const bool on = (int)e0 < cnt;
      const u64 e = on ? rec[2 + e0] : 0ull;
      ids[e0] = on ? (u32)(e & 0xFFFFFFFFull) : HNY_SENT;
      dist[e0] = on ? __uint_as_float((u32)(e >> 32)) : 0.f;
    }
    if (ln == 0) *cntp = cw;
  }
}

// fill_gaps_from_deleted (hnsw.rs:334-415), one wave per surviving old record (layer, slot):
//   bm = (old links  U  old links of every DELETED old neighbour)  -  deleted       (:382-388)
//   |bm| + |new| <= cap :  list = [(0.0, j) for j in bm ascending] ++ new           (:391-400)
//   else               :  list = robust_prune(new ++ [(d(slot, j), j) for j in bm]) (:403-410)
// Scratch: own old links + the old links of every deleted old neighbour = at most cap * (cap + 1)
// ids (1 056 at M0 = 32, 4 160 at M0 = 64) — a mass deletion reaches that — so the LDS arrays are
// sized for the worst case by the launcher (`maxu`), never for a typical one.
__host__ __device__ inline size_t fill_gaps_lds_bytes(u32 maxu) {
  return (size_t)maxu * 4 * 2 + (size_t)(maxu + HNY_MAX_CAP) * 8 * 2 + HNY_MAX_CAP * (8 + 4) + 64 * 4;
}
template <int LPR, int NCH>
__global__ __launch_bounds__(64) void k_fill_gaps(GraphDev g, const u64 *recs, u32 n_recs,
                                                  const unsigned char *deleted, u32 maxu) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64 *keys = reinterpret_cast<u64 *>(smem);            // [maxu + HNY_MAX_CAP]
  u64 *sorted = keys + (maxu + HNY_MAX_CAP);            // [maxu + HNY_MAX_CAP]
  u64 *S = sorted + (maxu + HNY_MAX_CAP);               // [HNY_MAX_CAP]
  u32 *cand = reinterpret_cast<u32 *>(S + HNY_MAX_CAP); // [maxu]
  u32 *bm = cand + maxu;                                // [maxu]
  u32 *s_ids = bm + maxu;                               // [HNY_MAX_CAP]
  float *tmp_d = reinterpret_cast<float *>(s_ids + HNY_MAX_CAP); // [64]
  const int MAXU = (int)maxu;
  const int ln = threadIdx.x, t = ln % LPR;
  u64 evals = 0;
  u32 overflow = 0;
  for (u32 ri = blockIdx.x; ri < n_recs; ri += gridDim.x) {
    const u64 rec = recs[ri];
    const u32 layer = (u32)(rec >> 31), slot = (u32)(rec & 0x7FFFFFFFull);
    u32 cap, dcap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)slot * g.M0;
      dist = g.l0_dist + (size_t)slot * g.M0;
      cntp = g.l0_cnt + slot;
    } else {
      size_t u = (size_t)g.upper_idx[slot] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    const int cnt = (int)(*cntp & 0xFFFFu);
    const u32 *dl = disk_ids(g, layer, slot, dcap);
    // ---- gather: own old links + the old links of deleted old neighbours
    int nu = 0;
    for (u32 j = 0; j < dcap; j++) {
      const u32 x = uni(dl[j]);
      if (x == HNY_SENT) break;
      if (nu < MAXU && ln == 0) cand[nu] = x;
      nu++;
      if (deleted[x]) {
        u32 c2;
        const u32 *xl = disk_ids(g, layer, x, c2);
        if (xl) {
          const u32 y = (u32)ln < c2 ? xl[ln] : HNY_SENT;
          const bool v = y != HNY_SENT;
          const u64 m = ballot(v);
          const int pos = nu + __popcll(m & ((1ull << ln) - 1ull));
          if (v && pos < MAXU) cand[pos] = y;
          nu += __popcll(m);
        }
      }
    }
    if (nu > MAXU) { // cannot happen: maxu is the worst case (old lists hold <= cap ids)
      overflow++;
      continue;
    }
    WSYNC();
    // ---- drop deleted ids and repeats, sort ascending (rank among the kept ones)
    for (int e = ln; e < nu; e += 64) {
      const u32 v = cand[e];
      bool keep = deleted[v] == 0;
      for (int k2 = 0; k2 < e && keep; k2++) keep = (cand[k2] & 0x7FFFFFFFu) != v;
      if (!keep) cand[e] = v | 0x80000000u;
    }
    WSYNC();
    int nb = 0;
    for (int base = 0; base < nu; base += 64) {
      const int e = base + ln;
      const bool kept = e < nu && !(cand[e] >> 31);
      if (kept) {
        const u32 v = cand[e];
        int pos = 0;
        for (int k2 = 0; k2 < nu; k2++) {
          const u32 o = cand[k2];
          pos += (!(o >> 31) && o < v) ? 1 : 0;
        }
        bm[pos] = v;
      }
      nb += __popcll(ballot(kept));
    }
    WSYNC();
    const u32 oi = ln < cnt ? ids[ln] : HNY_SENT;
    const float od = ln < cnt ? dist[ln] : 0.f;
    if (nb + cnt <= (int)cap) { // :391-400 distances are "no longer relevant": 0.0
      WSYNC();
      if (ln < nb) {
        ids[ln] = bm[ln];
        dist[ln] = 0.f;
      }
      if (ln < cnt) {
        ids[nb + ln] = oi;
        dist[nb + ln] = od;
      }
      if (ln == 0) *cntp = (u32)(nb + cnt);
      WSYNC();
      continue;
    }
    // ---- :403-410 score the old links and prune old + new together
    float4 q[NCH];
    load_row<LPR, NCH>(g.rows + (size_t)slot * g.row_stride, t, g.n16, q);
    const float qn = g.norms ? g.norms[slot] : 0.f;
    if (ln < cnt) keys[ln] = ((u64)fbits(od) << 32) | oi;
    for (int base = 0; base < nb; base += 64) {
      const int c = nb - base < 64 ? nb - base : 64;
      dist_rows<LPR, NCH>(g, q, qn, bm + base, c, tmp_d, g.rows + (size_t)slot * g.row_stride);
      evals += (u64)c;
      WSYNC();
      if (ln < c) keys[cnt + base + ln] = ((u64)fbits(tmp_d[ln]) << 32) | bm[base + ln];
      WSYNC();
    }
    const int n = cnt + nb;
    for (int e = ln; e < n; e += 64) {
      const u64 mine = keys[e];
      int rk = 0;
      for (int k2 = 0; k2 < n; k2++) {
        const u64 o = keys[k2];
        rk += (o < mine || (o == mine && k2 < e)) ? 1 : 0;
      }
      sorted[rk] = mine;
    }
    WSYNC();
    const int s_len = wave_prune<LPR, NCH>(g, sorted, n, (int)cap, S, s_ids, tmp_d, evals);
    if ((u32)ln < cap) {
      const bool on = ln < s_len;
      ids[ln] = on ? (u32)(S[ln] & 0xFFFFFFFFull) : HNY_SENT;
      dist[ln] = on ? __uint_as_float((u32)(S[ln] >> 32)) : 0.f;
    }
    if (ln == 0) *cntp = (u32)s_len;
    WSYNC();
  }
  if (ln == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_APPLY], evals);
    if (overflow) atomicAdd(&g.stats[ST_ERR_GAPS_OVERFLOW], (u64)overflow);
  }
}

// fill_gaps_from_deleted for lists of more than 64 slots (64 < M0 <= HNY_BIG_CAP), one 4-wave workgroup
// per surviving old record.  Same outcome as k_fill_gaps; the one-lane-per-slot arrays become loops and
// the gathered set lives in HBM: `bm` (= RoaringBitmap, ascending, no repeats) is built in a per-block
// bitmap over the slots by atomicOr and read back word by word in ascending order (the touched word
// range only; atomic exchange, which also clears it for the next record), the scored list is rank-sorted
// in HBM and robust_prune is wg_prune with its candidate list there.  Sized for M0 = 768 on a small
// index (the reference's fuzz test, src/tests/fuzz.rs:86-87); the bitmap read-back makes it O(n / 32)
// per record, so large indexes belong to the M0 <= 64 kernel.
__host__ __device__ inline size_t fill_gaps_wg_lds_bytes(u32 capmax, u32 row_stride, int SL) {
  return (size_t)capmax * 8 + 256 * 8 + 64 + (size_t)capmax * (8 + 4 + 4) + 128 + (size_t)(SL + 4) * row_stride;
}
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void k_fill_gaps_wg(GraphDev g, const u64 *recs, u32 n_recs,
                                                      const unsigned char *deleted, u32 *bitmaps, u32 words,
                                                      u32 *bm_all, u32 maxb, u64 *keys_all, u64 *sorted_all,
                                                      int SL) {
  extern __shared__ __align__(16) unsigned char smem[];
  const u32 capmax = wg_capmax(g);
  u32 *nl_ids = reinterpret_cast<u32 *>(smem);                 // [capmax] the record's in-memory list
  float *nl_d = reinterpret_cast<float *>(nl_ids + capmax);    // [capmax]
  u32 *wsc = reinterpret_cast<u32 *>(nl_d + capmax);           // [64] ids of a distance pass (+ slack)
  float *wsd = reinterpret_cast<float *>(wsc + 256);           // [64]
  int *misc = reinterpret_cast<int *>(wsd + 256);              // [0] first / [1] last touched word, [2..5] wave sums
  WgPruneLds L = wg_prune_carve(reinterpret_cast<unsigned char *>(misc + 16), SL, g.row_stride, 4, capmax);
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, t = ln % LPR;
  u32 *bitmap = bitmaps + (size_t)blockIdx.x * words;
  u32 *bm = bm_all + (size_t)blockIdx.x * maxb;
  u64 *keys = keys_all + (size_t)blockIdx.x * ((size_t)maxb + capmax);
  u64 *sorted = sorted_all + (size_t)blockIdx.x * ((size_t)maxb + capmax);
  u64 evals = 0;
  for (u32 ri = blockIdx.x; ri < n_recs; ri += gridDim.x) {
    const u64 rec = recs[ri];
    const u32 layer = (u32)(rec >> 31), slot = (u32)(rec & 0x7FFFFFFFull);
    u32 cap, dcap, *ids, *cntp;
    float *dist;
    if (layer == 0) {
      cap = g.M0;
      ids = g.l0_ids + (size_t)slot * g.M0;
      dist = g.l0_dist + (size_t)slot * g.M0;
      cntp = g.l0_cnt + slot;
    } else {
      size_t u = (size_t)g.upper_idx[slot] * g.up_layers + (layer - 1);
      cap = g.M;
      ids = g.up_ids + u * g.M;
      dist = g.up_dist + u * g.M;
      cntp = g.up_cnt + u;
    }
    const int cnt = (int)(*cntp & 0xFFFFu);
    for (int e = tid; e < cnt; e += 256) {
      nl_ids[e] = ids[e];
      nl_d[e] = dist[e];
    }
    if (tid == 0) {
      misc[0] = 0x7FFFFFFF;
      misc[1] = -1;
    }
    __syncthreads();
    // ---- gather (:382-388): own old links + the old links of deleted old neighbours, minus deleted
    const u32 *dl = disk_ids(g, layer, slot, dcap);
    auto mark = [&](u32 y) {
      if (y != HNY_SENT && deleted[y] == 0) {
        atomicOr(&bitmap[y >> 5], 1u << (y & 31));
        atomicMin(&misc[0], (int)(y >> 5));
        atomicMax(&misc[1], (int)(y >> 5));
      }
    };
    if (dl) {
      for (u32 j = (u32)tid; j < dcap; j += 256u) mark(dl[j]);
      for (u32 j = 0; j < dcap; j++) { // (block-uniform)
        const u32 x = dl[j];
        if (x == HNY_SENT) break;
        if (deleted[x]) {
          u32 c2;
          const u32 *xl = disk_ids(g, layer, x, c2);
          if (xl)
            for (u32 t2 = (u32)tid; t2 < c2; t2 += 256u) mark(xl[t2]);
        }
      }
    }
    __syncthreads();
    // ---- bm ascending: read the touched words back in order
    const int wmin = misc[0], wmax = misc[1];
    int nb = 0;
    __syncthreads();
    for (int w0 = wmin; w0 <= wmax; w0 += 256) {
      const int wi = w0 + tid;
      u32 bits = wi <= wmax ? __hip_atomic_exchange(&bitmap[wi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      const int pc = __popc(bits);
      int incl = pc; // inclusive scan inside the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (ln >= off) incl += o;
      }
      if (ln == 63) misc[2 + w] = incl;
      __syncthreads();
      int pos = nb + incl - pc;
      for (int k = 0; k < w; k++) pos += misc[2 + k];
      const int total = misc[2] + misc[3] + misc[4] + misc[5];
      while (bits) {
        const u32 b = (u32)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        if ((u32)pos < maxb) bm[pos] = (u32)wi * 32u + b;
        pos++;
      }
      nb += total;
      __syncthreads();
    }
    if ((u32)nb > maxb) nb = (int)maxb; // cannot happen: maxb = min(slots, cap * (cap + 1))
    __syncthreads();
    if (nb + cnt <= (int)cap) { // :391-400 distances are "no longer relevant": 0.0
      for (int e = tid; e < nb; e += 256) {
        ids[e] = bm[e];
        dist[e] = 0.f;
      }
      for (int e = tid; e < cnt; e += 256) {
        ids[nb + e] = nl_ids[e];
        dist[nb + e] = nl_d[e];
      }
      if (tid == 0) *cntp = (u32)(nb + cnt);
      __syncthreads();
      continue;
    }
    // ---- :403-410 score the old links and prune old + new together
    {
      const unsigned char *qrow = g.rows + (size_t)slot * g.row_stride;
      float4 q[NCH];
      load_row<LPR, NCH>(qrow, t, g.n16, q);
      const float qn = g.norms ? g.norms[slot] : 0.f;
      for (int e = tid; e < cnt; e += 256) keys[e] = ((u64)fbits(nl_d[e]) << 32) | nl_ids[e];
      // (dist_rows takes its lane from threadIdx.x: a one-wave routine, so wave 0 scores the set)
      u32 *my_ids = wsc;
      float *my_d = wsd;
      if (w == 0)
      for (int base = 0; base < nb; base += 64) {
        const int c = nb - base < 64 ? nb - base : 64;
        my_ids[ln] = ln < c ? bm[base + ln] : 0u;
        WSYNC();
        dist_rows<LPR, NCH>(g, q, qn, my_ids, c, my_d, qrow);
        evals += (u64)c;
        WSYNC();
        if (ln < c) keys[cnt + base + ln] = ((u64)fbits(my_d[ln]) << 32) | my_ids[ln];
        WSYNC();
      }
    }
    __syncthreads();
    const int n = cnt + nb;
    for (int e = tid; e < n; e += 256) {
      const u64 mine = keys[e];
      int rk = 0;
      for (int k2 = 0; k2 < n; k2++) {
        const u64 o = keys[k2];
        rk += (o < mine || (o == mine && k2 < e)) ? 1 : 0;
      }
      sorted[rk] = mine;
    }
    __syncthreads();
    int s_len;
    if (g.x86_order) {
      // strict mode: wg_prune carries its own wave-order arithmetic, so wave 0 runs the one-wave prune (dist_rows ->
      // the reference's x86 summation order, lists taken 64 slots at a time) on the workgroup's arrays
      if (w == 0) {
        const int sl = wave_prune<LPR, NCH>(g, sorted, n, (int)cap, L.S, L.s_ids, wsd, evals);
        if (ln == 0) misc[6] = sl;
      }
      __syncthreads();
      s_len = misc[6];
    } else {
      s_len = wg_prune<LPR, NCH, 4>(g, sorted, n, (int)cap, L, evals);
    }
    __syncthreads();
    for (u32 e = (u32)tid; e < cap; e += 256u) {
      const bool on = (int)e < s_len;
      ids[e] = on ? (u32)(L.S[e] & 0xFFFFFFFFull) : HNY_SENT;
      dist[e] = on ? __uint_as_float((u32)(L.S[e] >> 32)) : 0.f;
    }
    if (tid == 0) *cntp = (u32)s_len;
    __syncthreads();
  }
  if ((tid & 63) == 0 && evals) atomicAdd(&g.stats[ST_EVALS_APPLY], evals);
}

// D::distance for explicit pairs of stored items (tests / parity checks)
template <int LPR, int NCH>
__global__ __launch_bounds__(64) void k_pair_distances(GraphDev g, const u32 *pa, const u32 *pb, u32 n,
                                                       float *out) {
  __shared__ u32 ids[64];
  __shared__ float d[64];
  const int ln = threadIdx.x, t = ln % LPR;
  for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
    u32 qa = pa[i];
    float4 q[NCH];
    load_row<LPR, NCH>(g.rows + (size_t)qa * g.row_stride, t, g.n16, q);
    float qn = g.norms ? g.norms[qa] : 0.f;
    if (ln == 0) ids[0] = pb[i];
    WSYNC();
    dist_rows<LPR, NCH>(g, q, qn, ids, 1, d, g.rows + (size_t)qa * g.row_stride);
    WSYNC();
    if (ln == 0) out[i] = d[0];
    WSYNC();
  }
}

// Links record of one node = RoaringBitmap::from_iter(list ids) (hnsw.rs:204-208): ascending,
// deduplicated (slot order == id order).  One wave per list, in place: rank among the first
// occurrences by broadcast compare.  This is also the order Reader::visit iterates a Links bitmap
// in (reader.rs:343-346), so the k-NN search runs on the finalised lists.
__global__ __launch_bounds__(64) void k_finalize_lists(u32 *ids, u32 *cnt_out, u32 n_lists, u32 cap) {
  __shared__ u32 sh[HNY_BIG_CAP];
  __shared__ u32 fst[HNY_BIG_CAP];
  const int ln = threadIdx.x;
  for (u32 li = blockIdx.x; li < n_lists; li += gridDim.x) {
    u32 *row = ids + (size_t)li * cap;
    // (cap <= 64: every loop below runs once, one slot per lane)
    for (u32 e = (u32)ln; e < cap; e += 64u) {
      sh[e] = row[e];
      row[e] = HNY_SENT;
    }
    WSYNC();
    for (u32 e = (u32)ln; e < cap; e += 64u) {
      const u32 v = sh[e];
      bool first = v != HNY_SENT;
      for (u32 j = 0; j < e; j++) first = first && sh[j] != v;
      fst[e] = first ? 1u : 0u;
    }
    WSYNC();
    u32 kept = 0;
    for (u32 e0 = 0; e0 < cap; e0 += 64u) { // uniform trip count: the ballot below needs every lane
      const u32 e = e0 + (u32)ln;
      const bool first = e < cap && fst[e] != 0u;
      if (first) {
        const u32 v = sh[e];
        int pos = 0;
        for (u32 j = 0; j < cap; j++) pos += (fst[j] != 0u && sh[j] < v) ? 1 : 0;
        row[pos] = v; // rank among the first occurrences: every slot was cleared above
      }
      kept += (u32)__popcll(ballot(first));
    }
    if (ln == 0) cnt_out[li] = kept;
    WSYNC();
  }
}

// ---------------------------------------------------------------------------------------------
// add_item side (writer.rs:462-480): Distance::new_header and the bit codecs, on the device.
// Cosine norm = sqrt(dot(v, v)) in the REFERENCE's x86 summation order, so that the stored headers
// are bit-identical to what hannoy itself writes: 32 fma partials + hsum tree for dim >= 32
// (simple_avx.rs:8-13,69-110), 16 unfused partials for 16 <= dim < 32 (simple_sse.rs:10-14,64-110),
// scalar below (simple.rs:81-83).  One half-wave (32 lanes = the 32 partials) per vector.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_norms_x86(const float *v, u32 dim, u64 n, float *out) {
  const int ln = threadIdx.x, j = ln & 31, half = ln >> 5;
  for (u64 vi = (u64)blockIdx.x * 2 + half; vi < n; vi += (u64)gridDim.x * 2) {
    const float *x = v + vi * dim;
    float r;
    if (dim >= 32) {
      const u32 m = dim - dim % 32;
      float acc = 0.f;
      for (u32 i = 0; i < m; i += 32) acc = __builtin_fmaf(x[i + j], x[i + j], acc);
      acc = acc + __shfl_xor(acc, 4, 64); // hsum256: lane k + lane k+4
      acc = acc + __shfl_xor(acc, 2, 64); //          k + k+2
      acc = acc + __shfl_xor(acc, 1, 64); //          0 + 1
      const int b = half * 32;
      const float h1 = __shfl(acc, b, 64), h2 = __shfl(acc, b + 8, 64), h3 = __shfl(acc, b + 16, 64),
                  h4 = __shfl(acc, b + 24, 64);
      r = ((h1 + h2) + h3) + h4;
      for (u32 i = m; i < dim; i++) { // scalar tail, unfused
        float p = x[i] * x[i];
        r = r + p;
      }
    } else if (dim >= 16) {
      const u32 m = dim - dim % 16;
      float acc = 0.f;
      if (j < 16)
        for (u32 i = 0; i < m; i += 16) {
          float p = x[i + j] * x[i + j];
          acc = p + acc;
        }
      acc = acc + __shfl_xor(acc, 2, 64); // hsum128: k + k+2
      acc = acc + __shfl_xor(acc, 1, 64); //          0 + 1
      const int b = half * 32;
      const float h1 = __shfl(acc, b, 64), h2 = __shfl(acc, b + 4, 64), h3 = __shfl(acc, b + 8, 64),
                  h4 = __shfl(acc, b + 12, 64);
      r = ((h1 + h2) + h3) + h4;
      for (u32 i = m; i < dim; i++) {
        float p = x[i] * x[i];
        r = r + p;
      }
    } else {
      r = 0.f;
      for (u32 i = 0; i < dim; i++) {
        float p = x[i] * x[i];
        r = r + p;
      }
    }
    if (j == 0) out[vi] = __builtin_sqrtf(r);
  }
}

// Binary::from_slice (binary.rs:80-94: bit = 0 < bits < 0x8000_0000) / BinaryQuantized::from_slice
// (binary_quantized.rs:80-91: bit = is_sign_positive); dim i -> bit i%64 of u64 word i/64, zero
// padded.  One wave per 64-dim word: the ballot IS the word.
__global__ __launch_bounds__(64) void k_quantize(const float *v, u32 dim, u64 n, int binary_codec,
                                                 u64 *out) {
  const u32 words = (dim + 63) / 64;
  const u64 total = n * words;
  const int ln = threadIdx.x;
  for (u64 w = blockIdx.x; w < total; w += gridDim.x) {
    const u64 vi = w / words;
    const u32 d = (u32)(w % words) * 64 + (u32)ln;
    bool one = false;
    if (d < dim) {
      const u32 bits = __float_as_uint(v[vi * dim + d]);
      one = binary_codec ? (bits < 0x80000000u && bits > 0u) : (bits >> 31) == 0u;
    }
    const u64 word = ballot(one);
    if (ln == 0) out[w] = word;
  }
}

__global__ void k_fill_u32(u32 *p, u32 v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}


template <template <int, int> class Launcher, typename... Args>
hipError_t dispatch_shape(LaunchShape s, Args &&...args) {
#define HNY_CASE(L, C) \
  if (s.lpr == L && s.nch == C) return Launcher<L, C>::run(args...);
#ifdef HNY_ONLY_LPR // diagnostics (scripts/isa_report.py): one row shape only, so that one kernel compiles in seconds
  HNY_CASE(HNY_ONLY_LPR, HNY_ONLY_NCH)
#else
  HNY_CASE(8, 1)
  HNY_CASE(16, 1)
  HNY_CASE(32, 1)
  HNY_CASE(64, 1)
  HNY_CASE(64, 2)
  HNY_CASE(64, 3)
  HNY_CASE(64, 4)
  HNY_CASE(64, 6)
  HNY_CASE(64, 8)
  HNY_CASE(64, 12)
  HNY_CASE(64, 16)
#endif
#undef HNY_CASE
  return hipErrorInvalidValue;
}

// launchers of the four build kernels, general (SP = 0) or specialised for metric SP - 1
template <int SP>
struct Hot {
  template <int L, int C>
  struct Walk {
    static hipError_t run(const GraphDev &g, const WalkArgs &a, int grid, hipStream_t st) {
      size_t lds = hnyk_walk_lds_bytes(a.res_global ? 0u : a.rcap, a.eps_cap) + (size_t)a.vis_slots * 4 +
                   (size_t)a.vis_buckets * 8;
      if constexpr (SP == 0) {
        if (a.eps_cap > 64) {
          hipLaunchKernelGGL((k_walk<L, C, true, 0>), dim3(grid), dim3(64), lds, st, g, a);
          return hipGetLastError();
        }
      }
      if constexpr (SP != 0) {
        if constexpr (C <= HNY_RB_MAX_NCH) { // beam in registers (res <= 128 entries)
          const char *e = getenv("HNY_NO_RB"); // read per launch: tests flip it inside one process
          if (a.rcap <= 128 && !(e && atoi(e) != 0)) {
            // (a ONE-chunk register beam for ef <= 64 — every beam scan a single ballot — was measured on
            // C5: walk 0.692 s against 0.668 s with two chunks, i.e. no gain; the beam scans are not what
            // bounds the short-row walk)
            if constexpr (C == 1 && L <= 32) {
              // build walks on short rows whose result sets never exceed 64 entries (ef <= 64, few entry points —
              // the host decides, WalkArgs.rb_one): ONE 64-entry chunk in registers.  Round 3 measured this form
              // as no gain (0.692 against 0.668 s) while the walk was bound by its global atomics; on the
              // issue-bound walk of round 5 every beam scan, merge rank and eviction round it halves counts.
              if (!a.reader_mode && a.rb_one) {
                hipLaunchKernelGGL((k_walk<L, C, false, SP, false, 1>), dim3(grid), dim3(64), lds, st, g, a);
                return hipGetLastError();
              }
            }
            if (a.reader_mode)
              hipLaunchKernelGGL((k_walk<L, C, false, SP, true, 2>), dim3(grid), dim3(64), lds, st, g, a);
            else
              hipLaunchKernelGGL((k_walk<L, C, false, SP, false, 2>), dim3(grid), dim3(64), lds, st, g, a);
            return hipGetLastError();
          }
          // (a 4-chunk register beam for ef 128..255 was measured and is slower than the LDS beam:
          // 128-d efC=200 walk 0.85 -> 1.00 s, 768-d (C3) 0.62 -> 0.68 s — registers spill)
        }
        if (a.reader_mode) {
          hipLaunchKernelGGL((k_walk<L, C, false, SP, true>), dim3(grid), dim3(64), lds, st, g, a);
          return hipGetLastError();
        }
      }
      hipLaunchKernelGGL((k_walk<L, C, false, SP>), dim3(grid), dim3(64), lds, st, g, a);
      return hipGetLastError();
    }
  };
  template <int L, int C>
  struct PruneWg {
    static hipError_t run(const GraphDev &g, const PruneArgs &a, int SL, int nw, int grid, hipStream_t st) {
      if constexpr (C > 8) {
        return hipErrorInvalidValue; // 4 rows x C chunks do not fit the register file: wave prune
      } else {
        size_t lds = wg_prune_lds_bytes((SP == 0 && a.list_global) ? 0u : a.rcap, g.row_stride, SL, nw, wg_capmax(g));
        if (lds > 65536) { // wide lists (M0 up to HNY_BIG_CAP) next to long rows
          hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prune_wg<L, C, 4, SP>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (rc != hipSuccess) return rc;
        }
        hipLaunchKernelGGL((k_prune_wg<L, C, 4, SP>), dim3(grid), dim3(256), lds, st, g, a, SL);
        return hipGetLastError();
      }
    }
  };
  // short rows (<= 512 B): one wave per query, eight candidates at a time (k_prune_n8)
  static hipError_t prune_n8(const GraphDev &g, const PruneArgs &a, int lpro, int SL, int grid, hipStream_t st) {
    if constexpr (SP == 0) {
      return hipErrorInvalidValue;
    } else {
      const size_t lds = prune_n8_lds_bytes(a.rcap, g.row_stride, SL);
      if (lpro == 8) hipLaunchKernelGGL((k_prune_n8<8, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else if (lpro == 16) hipLaunchKernelGGL((k_prune_n8<16, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else if (lpro == 32) hipLaunchKernelGGL((k_prune_n8<32, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
  }
  static hipError_t apply_n8(const GraphDev &g, const ApplyArgs &a, int lpro, int SL, int grid, hipStream_t st) {
    if constexpr (SP == 0) {
      return hipErrorInvalidValue;
    } else {
      const size_t lds = apply_n8_lds_bytes(g.row_stride, SL);
      if (lpro == 8) hipLaunchKernelGGL((k_apply_n8<8, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else if (lpro == 16) hipLaunchKernelGGL((k_apply_n8<16, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else if (lpro == 32) hipLaunchKernelGGL((k_apply_n8<32, SP>), dim3(grid), dim3(64), lds, st, g, a, SL);
      else return hipErrorInvalidValue;
      return hipGetLastError();
    }
  }
  template <int L, int C>
  struct ApplyWg {
    static hipError_t run(const GraphDev &g, const ApplyArgs &a, int SL, int grid, hipStream_t st) {
      if constexpr (C > 8) {
        return hipErrorInvalidValue;
      } else {
        size_t lds = wg_prune_lds_bytes(2 * wg_capmax(g), g.row_stride, SL, 4, wg_capmax(g));
        if (lds > 65536) {
          hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_apply_wg<L, C, SP>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (rc != hipSuccess) return rc;
        }
        hipLaunchKernelGGL((k_apply_wg<L, C, SP>), dim3(grid), dim3(256), lds, st, g, a, SL);
        return hipGetLastError();
      }
    }
  };
  template <int L, int C>
  struct Apply {
    static hipError_t run(const GraphDev &g, const ApplyArgs &a, int grid, hipStream_t st) {
      const size_t lds = (size_t)wave_capmax(g) * (8 + 8 + 8 + 4) + 64 * 4;
      hipLaunchKernelGGL((k_apply<L, C, SP>), dim3(grid), dim3(64), lds, st, g, a);
      return hipGetLastError();
    }
  };
};
template <int L, int C>
struct NnsFilteredLauncher {
  static hipError_t run(const GraphDev &g, const NnsArgs &a, int grid, hipStream_t st) {
    size_t lds = hnyk_walk_lds_bytes(a.rcap, a.eps_cap) + (size_t)a.vis_slots * 4;
    hipLaunchKernelGGL((k_nns_filtered<L, C>), dim3(grid), dim3(64), lds, st, g, a);
    return hipGetLastError();
  }
};
template <int L, int C>
struct WalkHeapLauncher {
  static hipError_t run(const GraphDev &g, const WalkArgs &a, int grid, hipStream_t st) {
    const size_t lds = 64 * 4 * 2 + (size_t)a.eps_cap * 4;
    hipLaunchKernelGGL((k_walk_heap<L, C>), dim3(grid), dim3(64), lds, st, g, a);
    return hipGetLastError();
  }
};
template <int L, int C>
struct NnsHeapLauncher {
  static hipError_t run(const GraphDev &g, const NnsArgs &a, int grid, hipStream_t st) {
    const size_t lds = 64 * 4 * 2 + (size_t)a.eps_cap * 4;
    hipLaunchKernelGGL((k_nns_heap<L, C>), dim3(grid), dim3(64), lds, st, g, a);
    return hipGetLastError();
  }
};
template <int L, int C>
struct NnsLinearLauncher {
  static hipError_t run(const GraphDev &g, const NnsArgs &a, int grid, hipStream_t st) {
    size_t lds = (size_t)a.rcap * 8 + 64 * 4 * 2;
    hipLaunchKernelGGL((k_nns_linear<L, C>), dim3(grid), dim3(64), lds, st, g, a);
    return hipGetLastError();
  }
};
template <int L, int C>
struct PruneLauncher {
  static hipError_t run(const GraphDev &g, const PruneArgs &a, int grid, hipStream_t st) {
    size_t lds = (size_t)a.rcap * 8 + (size_t)wave_capmax(g) * (8 + 4) + 64 * 4;
    hipLaunchKernelGGL((k_prune<L, C>), dim3(grid), dim3(64), lds, st, g, a);
    return hipGetLastError();
  }
};
template <int L, int C>
struct GapsLauncher {
  static hipError_t run(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                        hipStream_t st) {
    int grid = n_recs < 16384u ? (int)n_recs : 16384;
    const u32 cap = g.M0 > g.M ? g.M0 : g.M;
    const u32 maxu = cap * (cap + 1u);
    const size_t lds = fill_gaps_lds_bytes(maxu);
    if (lds > 65536) {
      hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fill_gaps<L, C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (rc != hipSuccess) return rc;
    }
    hipLaunchKernelGGL((k_fill_gaps<L, C>), dim3(grid), dim3(64), lds, st, g, recs, n_recs, deleted, maxu);
    return hipGetLastError();
  }
};
template <int L, int C>
struct GapsWgLauncher {
  static hipError_t run(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                        u32 *bitmaps, u32 words, u32 *bm, u32 maxb, u64 *keys, u64 *sorted, int SL, int grid,
                        hipStream_t st) {
    if constexpr (C > 8) {
      return hipErrorInvalidValue; // wg_prune: rows of at most 8 KB
    } else {
      const size_t lds = fill_gaps_wg_lds_bytes(wg_capmax(g), g.row_stride, SL);
      if (lds > 65536) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fill_gaps_wg<L, C>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
      }
      hipLaunchKernelGGL((k_fill_gaps_wg<L, C>), dim3(grid), dim3(256), lds, st, g, recs, n_recs, deleted, bitmaps,
                         words, bm, maxb, keys, sorted, SL);
      return hipGetLastError();
    }
  }
};
template <int L, int C>
struct PairLauncher {
  static hipError_t run(const GraphDev &g, const u32 *a, const u32 *b, u32 n, float *out,
                        hipStream_t st) {
    int grid = n < 4096u ? (int)n : 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_pair_distances<L, C>), dim3(grid), dim3(64), 0, st, g, a, b, n, out);
    return hipGetLastError();
  }
};

} // namespace

#define HNY_CAT2(a, b) a##b
#define HNY_CAT(a, b) HNY_CAT2(a, b)
#if HNY_PART != 0
// ---- part 1..7: the build kernels specialised for one metric ----
hipError_t HNY_CAT(hnyk_walk_sp, HNY_PART)(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid,
                                           hipStream_t st) {
  return dispatch_shape<Hot<HNY_PART>::Walk>(s, g, a, grid, st);
}
hipError_t HNY_CAT(hnyk_prune_wg_sp, HNY_PART)(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL,
                                               int nw, int grid, hipStream_t st) {
  return dispatch_shape<Hot<HNY_PART>::PruneWg>(s, g, a, SL, nw, grid, st);
}
hipError_t HNY_CAT(hnyk_prune_n8_sp, HNY_PART)(const GraphDev &g, const PruneArgs &a, int lpro, int SL, int grid,
                                               hipStream_t st) {
  return Hot<HNY_PART>::prune_n8(g, a, lpro, SL, grid, st);
}
hipError_t HNY_CAT(hnyk_apply_n8_sp, HNY_PART)(const GraphDev &g, const ApplyArgs &a, int lpro, int SL, int grid,
                                               hipStream_t st) {
  return Hot<HNY_PART>::apply_n8(g, a, lpro, SL, grid, st);
}
hipError_t HNY_CAT(hnyk_apply_sp, HNY_PART)(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int grid,
                                            hipStream_t st) {
  return dispatch_shape<Hot<HNY_PART>::Apply>(s, g, a, grid, st);
}
hipError_t HNY_CAT(hnyk_apply_wg_sp, HNY_PART)(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL,
                                               int grid, hipStream_t st) {
  return dispatch_shape<Hot<HNY_PART>::ApplyWg>(s, g, a, SL, grid, st);
}
#else
// ---- part 0: the launch entry points ----
// the specialised kernels serve the plain build: wave order, fresh index (HNY_NO_FAST=1 keeps every
// launch on the general kernels)
static bool fast_path(const GraphDev &g) {
  const char *e = getenv("HNY_NO_FAST"); // read per launch: tests flip it inside one process
  const bool off = e && atoi(e) != 0;
  return !off && !g.x86_order && !g.incremental && g.metric >= 0 && g.metric < 7 && g.M0 <= 64u;
}
#define HNY_SP_SWITCH(fn, ...)                 \
  switch (g.metric) {                          \
    case 0: return fn##1(__VA_ARGS__);         \
    case 1: return fn##2(__VA_ARGS__);         \
    case 2: return fn##3(__VA_ARGS__);         \
    case 3: return fn##4(__VA_ARGS__);         \
    case 4: return fn##5(__VA_ARGS__);         \
    case 5: return fn##6(__VA_ARGS__);         \
    default: return fn##7(__VA_ARGS__);        \
  }

size_t hnyk_walk_lds_bytes(u32 rcap, u32 eps_cap) {
  return (size_t)rcap * 8 + HNY_POOL_CAP * 8 + 64 * 4 * 2 + (size_t)eps_cap * 4;
}

hipError_t hnyk_walk(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid, hipStream_t st) {
  if (fast_path(g) && a.eps_cap <= 64 && !a.res_global) { HNY_SP_SWITCH(hnyk_walk_sp, g, a, s, grid, st) }
  return dispatch_shape<Hot<0>::Walk>(s, g, a, grid, st);
}
hipError_t hnyk_walk_heap(const GraphDev &g, const WalkArgs &a, LaunchShape s, int grid, hipStream_t st) {
  return dispatch_shape<WalkHeapLauncher>(s, g, a, grid, st);
}
hipError_t hnyk_nns_filtered(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st) {
  return dispatch_shape<NnsFilteredLauncher>(s, g, a, grid, st);
}
hipError_t hnyk_nns_heap(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st) {
  return dispatch_shape<NnsHeapLauncher>(s, g, a, grid, st);
}
hipError_t hnyk_nns_linear(const GraphDev &g, const NnsArgs &a, LaunchShape s, int grid, hipStream_t st) {
  return dispatch_shape<NnsLinearLauncher>(s, g, a, grid, st);
}
hipError_t hnyk_prune(const GraphDev &g, const PruneArgs &a, LaunchShape s, int grid, hipStream_t st) {
  return dispatch_shape<PruneLauncher>(s, g, a, grid, st);
}
hipError_t hnyk_apply(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int grid, hipStream_t st) {
  if (fast_path(g)) { HNY_SP_SWITCH(hnyk_apply_sp, g, a, s, grid, st) }
  return dispatch_shape<Hot<0>::Apply>(s, g, a, grid, st);
}
// the one-wave prune for short rows: plain build, rows of at most 32 units, lists that fit the LDS
bool hnyk_prune_n8_ok(const GraphDev &g, const PruneArgs &a, LaunchShape s) {
  const char *e = getenv("HNY_PRUNE_N8"); // read per launch: tests and A/B runs flip it inside one process
  if (e && atoi(e) == 0) return false;
  return fast_path(g) && s.nch == 1 && s.lpr <= 32 && !a.list_global && a.rcap <= 512 && a.cap <= (u32)HNY_MAX_CAP;
}
hipError_t hnyk_prune_n8(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL, int grid, hipStream_t st) {
  HNY_SP_SWITCH(hnyk_prune_n8_sp, g, a, s.lpr, SL, grid, st)
}
bool hnyk_apply_n8_ok(const GraphDev &g, LaunchShape s) {
  const char *e = getenv("HNY_PRUNE_N8");
  if (e && atoi(e) == 0) return false;
  return fast_path(g) && s.nch == 1 && s.lpr <= 32 && g.M0 <= (u32)HNY_MAX_CAP && g.M <= (u32)HNY_MAX_CAP;
}
hipError_t hnyk_apply_n8(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL, int grid, hipStream_t st) {
  HNY_SP_SWITCH(hnyk_apply_n8_sp, g, a, s.lpr, SL, grid, st)
}
hipError_t hnyk_prune_wg(const GraphDev &g, const PruneArgs &a, LaunchShape s, int SL, int nw, int grid,
                         hipStream_t st) {
  if (fast_path(g) && nw == 4 && !a.list_global) { HNY_SP_SWITCH(hnyk_prune_wg_sp, g, a, s, SL, nw, grid, st) }
  return dispatch_shape<Hot<0>::PruneWg>(s, g, a, SL, nw, grid, st);
}
hipError_t hnyk_apply_wg(const GraphDev &g, const ApplyArgs &a, LaunchShape s, int SL, int grid,
                         hipStream_t st) {
  if (fast_path(g)) { HNY_SP_SWITCH(hnyk_apply_wg_sp, g, a, s, SL, grid, st) }
  return dispatch_shape<Hot<0>::ApplyWg>(s, g, a, SL, grid, st);
}
hipError_t hnyk_pair_distances(const GraphDev &g, const u32 *a, const u32 *b, u32 n, float *out,
                               LaunchShape s, hipStream_t st) {
  return dispatch_shape<PairLauncher>(s, g, a, b, n, out, st);
}
hipError_t hnyk_fill_gaps(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                          LaunchShape s, hipStream_t st) {
  if (!n_recs) return hipSuccess;
  return dispatch_shape<GapsLauncher>(s, g, recs, n_recs, deleted, st);
}
hipError_t hnyk_fill_gaps_wg(const GraphDev &g, const u64 *recs, u32 n_recs, const unsigned char *deleted,
                             u32 *bitmaps, u32 words, u32 *bm, u32 maxb, u64 *keys, u64 *sorted, int SL, int grid,
                             LaunchShape s, hipStream_t st) {
  if (!n_recs) return hipSuccess;
  return dispatch_shape<GapsWgLauncher>(s, g, recs, n_recs, deleted, bitmaps, words, bm, maxb, keys, sorted, SL,
                                        grid, st);
}
hipError_t hnyk_emit(const GraphDev &g, const EmitArgs &a, hipStream_t st) {
  u64 total = (u64)a.count * (a.batch_level + 1) * a.cap_sel;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(k_emit, dim3((unsigned)std::min<u64>((total + 255) / 256, 2048)), dim3(256), 0, st, g, a);
  return hipGetLastError();
}
hipError_t hnyk_segments(const u64 *keys, u32 n_ops, u32 *seg_start, u32 *n_seg, hipStream_t st) {
  if (!n_ops) return hipSuccess;
  hipLaunchKernelGGL(k_segments, dim3(std::min<u32>((n_ops + 1023) / 1024, 2048u)), dim3(256), 0, st, keys,
                     n_ops, seg_start, n_seg);
  return hipGetLastError();
}
hipError_t hnyk_apply_append(const GraphDev &g, const ApplyArgs &a, hipStream_t st) {
  if (!a.n_ops) return hipSuccess;
  hipLaunchKernelGGL(k_apply_append, dim3(std::min<u32>((a.n_ops / 2 + 255) / 256, 4096u)), dim3(256), 0, st, g, a);
  return hipGetLastError();
}
hipError_t hnyk_finalize_lists(u32 *ids, u32 *cnt_out, u32 n_lists, u32 cap, hipStream_t st) {
  if (!n_lists) return hipSuccess;
  unsigned grid = n_lists < 65536u * 4u ? n_lists : 65536u * 4u;
  hipLaunchKernelGGL(k_finalize_lists, dim3(grid), dim3(64), 0, st, ids, cnt_out, n_lists, cap);
  return hipGetLastError();
}
hipError_t hnyk_norms_x86(const float *v, u32 dim, u64 n, float *out, hipStream_t st) {
  if (!n) return hipSuccess;
  u64 blocks = (n + 1) / 2;
  if (blocks > 262144) blocks = 262144;
  hipLaunchKernelGGL(k_norms_x86, dim3((unsigned)blocks), dim3(64), 0, st, v, dim, n, out);
  return hipGetLastError();
}
hipError_t hnyk_quantize(const float *v, u32 dim, u64 n, int binary_codec, u64 *out, hipStream_t st) {
  if (!n) return hipSuccess;
  u64 blocks = n * ((dim + 63) / 64);
  if (blocks > 1048576) blocks = 1048576;
  hipLaunchKernelGGL(k_quantize, dim3((unsigned)blocks), dim3(64), 0, st, v, dim, n, binary_codec, out);
  return hipGetLastError();
}
hipError_t hnyk_fill_u32(u32 *p, u32 v, size_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, st, p, v, n);
  return hipGetLastError();
}
hipError_t hnyk_sort_pairs(void *temp, size_t &temp_bytes, u64 *keys_in, u64 *keys_out, u64 *vals_in,
                           u64 *vals_out, u32 n, u32 begin_bit, u32 end_bit, hipStream_t st) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n,
                                   begin_bit, end_bit, st);
}
hipError_t hnyk_sort_u32(void *temp, size_t &temp_bytes, u32 *in, u32 *out, u32 n, hipStream_t st) {
  return rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, 32, st);
}
hipError_t hnyk_apply_merge(const GraphDev &g, const u64 *exch, u32 n_def, u32 world, u32 rank, u32 per,
                            u32 stride, hipStream_t st) {
  if (!n_def || world < 2) return hipSuccess;
  hipLaunchKernelGGL(k_apply_merge, dim3(std::min<u32>(n_def, 8192u)), dim3(64), 0, st, g, exch, n_def, world,
                     rank, per, stride);
  return hipGetLastError();
}
hipError_t hnyk_sort_pairs48(void *temp, size_t &temp_bytes, u64 *keys_in, u64 *keys_out, u64 *vals_in,
                             u64 *vals_out, u32 n, hipStream_t st) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n,
                                   0, 48, st);
}
// drain_asc().take(k) (reader.rs:797-798): the k best of every query's sorted result row
__global__ void k_take_topk(const u64 *cand, const u32 *cand_n, u32 rcap, u32 k, u32 n, u64 *out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (u64)n * k) return;
  u32 q = (u32)(i / k), j = (u32)(i % k);
  out[i] = j < cand_n[q] ? cand[(size_t)q * rcap + j] : HNY_OP_INVALID;
}
hipError_t hnyk_take_topk(const u64 *cand, const u32 *cand_n, u32 rcap, u32 k, u32 n, u64 *out,
                          hipStream_t st) {
  u64 total = (u64)n * k;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(k_take_topk, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cand, cand_n,
                     rcap, k, n, out);
  return hipGetLastError();
}
__global__ void k_iota_u64(u64 *p, u32 base, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (u64)(base + i);
}
hipError_t hnyk_iota_u64(u64 *p, u32 base, u32 n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_iota_u64, dim3((n + 255) / 256), dim3(256), 0, st, p, base, n);
  return hipGetLastError();
}

// Device self-test of the cross-lane primitives above against __shfl_xor (one wave; bit i of out[lane]
// = mismatch in check i).  hny_selftest_lane_ops, tests/test_gpu_parity.py.
template <int OFF>
__device__ __forceinline__ u32 lane_selftest_one(u32 v, u32 w) {
  u32 bad = 0;
  if (xshfl_u32<OFF>(v) != (u32)__shfl_xor((int)v, OFF, 64)) bad |= 1u;
  const bool hi = (threadIdx.x & OFF) != 0;
  {
    const u32 keep = hi ? w : v, send = hi ? v : w;
    if (fold_step<OFF, u32>(v, w) != keep + (u32)__shfl_xor((int)send, OFF, 64)) bad |= 2u;
  }
  {
    const float a = __uint_as_float(0x3F800000u | (v & 0x7FFFFFu)), b = __uint_as_float(0x40000000u | (w & 0x7FFFFFu));
    const float keep = hi ? b : a, send = hi ? a : b;
    const float ref = keep + __shfl_xor(send, OFF, 64);
    if (__float_as_uint(fold_step<OFF, float>(a, b)) != __float_as_uint(ref)) bad |= 4u;
    if (__float_as_uint(xshfl<OFF>(a)) != __float_as_uint(__shfl_xor(a, OFF, 64))) bad |= 8u;
  }
  const u64 q = ((u64)v << 32) | w;
  if (xshfl<OFF>(q) != (u64)__shfl_xor((long long)q, OFF, 64)) bad |= 16u;
  return bad;
}
__global__ __launch_bounds__(64) void k_lane_selftest(u32 *out) {
  const u32 v = (threadIdx.x + 1u) * 2654435761u, w = (threadIdx.x + 77u) * 40503u + 0x9E3779B9u;
  u32 bad = 0;
  bad |= lane_selftest_one<1>(v, w) << 0;
  bad |= lane_selftest_one<2>(v, w) << 5;
  bad |= lane_selftest_one<4>(v, w) << 10;
  bad |= lane_selftest_one<8>(v, w) << 15;
  bad |= lane_selftest_one<16>(v, w) << 20;
  bad |= lane_selftest_one<32>(v, w) << 25;
  out[threadIdx.x] = bad;
}
hipError_t hnyk_lane_selftest(u32 *out64, hipStream_t st) {
  hipLaunchKernelGGL(k_lane_selftest, dim3(1), dim3(64), 0, st, out64);
  return hipGetLastError();
}
#endif // HNY_PART == 0
