/*
 * hannoy_oracle.cpp — CPU restatement of hannoy's HNSW build hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see hannoy_oracle.h).  Plain, single-file C++17;
 * every function cites the reference file:line it follows (paths relative to
 * /root/reference/).  Compile with -ffp-contract=off: every fused multiply-add
 * below is an explicit fmaf() exactly where the reference uses
 * _mm256_fmadd_ps, and nowhere else.
 */
#include "hannoy_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <queue>
#include <thread>
#include <vector>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORC_HAVE_AVX 1
#else
#define ORC_HAVE_AVX 0
#endif

namespace {

inline uint32_t f32_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float loadf(const void *p, size_t i) {
  float f;
  std::memcpy(&f, (const uint8_t *)p + 4 * i, 4);
  return f;
}
inline uint64_t load64(const void *p, size_t i) {
  uint64_t w;
  std::memcpy(&w, (const uint8_t *)p + 8 * i, 8);
  return w;
}

/* ------------------------------------------------------------------ */
/* src/spaces/simple_avx.rs — scalar emulation of the AVX+FMA kernels  */
/* ------------------------------------------------------------------ */

/* simple_avx.rs:8-13 hsum256_ps_avx on 8 lanes acc[0..8) */
inline float hsum256_emul(const float *x) {
  float x128[4], x64[2];
  for (int k = 0; k < 4; k++) x128[k] = x[4 + k] + x[k]; /* extractf128(x,1) + cast128(x) */
  for (int k = 0; k < 2; k++) x64[k] = x128[k] + x128[k + 2]; /* x128 + movehl(x128,x128) */
  return x64[0] + x64[1]; /* add_ss(x64, shuffle(x64, 0x55)) */
}
/* simple_sse.rs:10-14 hsum128_ps_sse */
inline float hsum128_emul(const float *x) {
  float x64[2];
  for (int k = 0; k < 2; k++) x64[k] = x[k] + x[k + 2];
  return x64[0] + x64[1];
}

/* simple_avx.rs:69-110 dot_similarity_avx */
float dot_avx_emul(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 32);
  float acc[32];
  for (int j = 0; j < 32; j++) acc[j] = 0.0f;
  for (size_t i = 0; i < m; i += 32)
    for (int j = 0; j < 32; j++) acc[j] = fmaf(loadf(a, i + j), loadf(b, i + j), acc[j]);
  float result = hsum256_emul(acc) + hsum256_emul(acc + 8) + hsum256_emul(acc + 16) +
                 hsum256_emul(acc + 24);
  for (size_t i = m; i < n; i++) {
    float p = loadf(a, i) * loadf(b, i); /* result += a * b — unfused */
    result += p;
  }
  return result;
}
/* simple_avx.rs:17-65 euclid_similarity_avx */
float euclid_avx_emul(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 32);
  float acc[32];
  for (int j = 0; j < 32; j++) acc[j] = 0.0f;
  for (size_t i = 0; i < m; i += 32)
    for (int j = 0; j < 32; j++) {
      float d = loadf(a, i + j) - loadf(b, i + j);
      acc[j] = fmaf(d, d, acc[j]);
    }
  float result = hsum256_emul(acc) + hsum256_emul(acc + 8) + hsum256_emul(acc + 16) +
                 hsum256_emul(acc + 24);
  for (size_t i = m; i < n; i++) {
    float d = loadf(a, i) - loadf(b, i);
    float p = d * d; /* (a - b).powi(2) */
    result += p;
  }
  return result;
}
/* simple_sse.rs:64-110 dot_similarity_sse (mul then add, unfused) */
float dot_sse_emul(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 16);
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.0f;
  for (size_t i = 0; i < m; i += 16)
    for (int j = 0; j < 16; j++) {
      float p = loadf(a, i + j) * loadf(b, i + j);
      acc[j] = p + acc[j];
    }
  float result =
      hsum128_emul(acc) + hsum128_emul(acc + 4) + hsum128_emul(acc + 8) + hsum128_emul(acc + 12);
  for (size_t i = m; i < n; i++) {
    float p = loadf(a, i) * loadf(b, i);
    result += p;
  }
  return result;
}
/* simple_sse.rs:17-61 euclid_similarity_sse */
float euclid_sse_emul(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 16);
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.0f;
  for (size_t i = 0; i < m; i += 16)
    for (int j = 0; j < 16; j++) {
      float d = loadf(a, i + j) - loadf(b, i + j);
      float p = d * d;
      acc[j] = p + acc[j];
    }
  float result =
      hsum128_emul(acc) + hsum128_emul(acc + 4) + hsum128_emul(acc + 8) + hsum128_emul(acc + 12);
  for (size_t i = m; i < n; i++) {
    float d = loadf(a, i) - loadf(b, i);
    float p = d * d;
    result += p;
  }
  return result;
}
/* simple.rs:81-83 dot_product_non_optimized / :49-51 euclidean_distance_non_optimized */
float dot_scalar(const void *a, const void *b, size_t n) {
  float s = 0.0f;
  for (size_t i = 0; i < n; i++) {
    float p = loadf(a, i) * loadf(b, i);
    s = s + p;
  }
  return s;
}
float euclid_scalar(const void *a, const void *b, size_t n) {
  float s = 0.0f;
  for (size_t i = 0; i < n; i++) {
    float d = loadf(a, i) - loadf(b, i);
    float p = d * d;
    s = s + p;
  }
  return s;
}

#if ORC_HAVE_AVX
/* The same two kernels with the real intrinsics (bit-identical to the
 * emulation above; used for speed in the CPU baseline). */
inline float hsum256_real(__m256 x) {
  __m128 x128 = _mm_add_ps(_mm256_extractf128_ps(x, 1), _mm256_castps256_ps128(x));
  __m128 x64 = _mm_add_ps(x128, _mm_movehl_ps(x128, x128));
  __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
  return _mm_cvtss_f32(x32);
}
float dot_avx_real(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 32);
  const float *p1 = (const float *)a, *p2 = (const float *)b;
  __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
  for (size_t i = 0; i < m; i += 32) {
    s1 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + i), _mm256_loadu_ps(p2 + i), s1);
    s2 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + i + 8), _mm256_loadu_ps(p2 + i + 8), s2);
    s3 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + i + 16), _mm256_loadu_ps(p2 + i + 16), s3);
    s4 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + i + 24), _mm256_loadu_ps(p2 + i + 24), s4);
  }
  float result = hsum256_real(s1) + hsum256_real(s2) + hsum256_real(s3) + hsum256_real(s4);
  for (size_t i = m; i < n; i++) {
    float p = loadf(a, i) * loadf(b, i);
    result += p;
  }
  return result;
}
float euclid_avx_real(const void *a, const void *b, size_t n) {
  size_t m = n - (n % 32);
  const float *p1 = (const float *)a, *p2 = (const float *)b;
  __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
  for (size_t i = 0; i < m; i += 32) {
    __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(p1 + i), _mm256_loadu_ps(p2 + i));
    s1 = _mm256_fmadd_ps(d1, d1, s1);
    __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(p1 + i + 8), _mm256_loadu_ps(p2 + i + 8));
    s2 = _mm256_fmadd_ps(d2, d2, s2);
    __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(p1 + i + 16), _mm256_loadu_ps(p2 + i + 16));
    s3 = _mm256_fmadd_ps(d3, d3, s3);
    __m256 d4 = _mm256_sub_ps(_mm256_loadu_ps(p1 + i + 24), _mm256_loadu_ps(p2 + i + 24));
    s4 = _mm256_fmadd_ps(d4, d4, s4);
  }
  float result = hsum256_real(s1) + hsum256_real(s2) + hsum256_real(s3) + hsum256_real(s4);
  for (size_t i = m; i < n; i++) {
    float d = loadf(a, i) - loadf(b, i);
    float p = d * d;
    result += p;
  }
  return result;
}
#endif

/* simple.rs:53-79 dot_product dispatch (x86_64 host with avx+fma+sse) */
float dot_x86(const void *a, const void *b, size_t n) {
  if (n >= 32) {
#if ORC_HAVE_AVX
    return dot_avx_real(a, b, n);
#else
    return dot_avx_emul(a, b, n);
#endif
  }
  if (n >= 16) return dot_sse_emul(a, b, n);
  return dot_scalar(a, b, n);
}
/* simple.rs:19-47 euclidean_distance dispatch */
float euclid_x86(const void *a, const void *b, size_t n) {
  if (n >= 32) {
#if ORC_HAVE_AVX
    return euclid_avx_real(a, b, n);
#else
    return euclid_avx_emul(a, b, n);
#endif
  }
  if (n >= 16) return euclid_sse_emul(a, b, n);
  return euclid_scalar(a, b, n);
}

/* ------------------------------------------------------------------ */
/* WAVE order: restatement of the HIP fast path            */
/* ------------------------------------------------------------------ */
inline uint32_t pow2ceil(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}
inline uint32_t wave_lpr(uint32_t dim) {
  uint32_t dim4 = (dim + 3) / 4;
  uint32_t l = pow2ceil(dim4);
  if (l < 8) l = 8;
  if (l > 64) l = 64;
  return l;
}
enum { WOP_DOT = 0, WOP_EUCLID = 1, WOP_MANHATTAN = 2 };
/* chunks per lane, rounded up to the kernel's template set {1,2,3,4,6,8,12,16} */
inline uint32_t wave_nch(uint32_t dim4, uint32_t lpr) {
  uint32_t n = (dim4 + lpr - 1) / lpr;
  static const uint32_t set[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (uint32_t s : set)
    if (n <= s) return s;
  return n;
}
/* Lane t of an LPR-lane group owns float4 #(c*LPR + t) of the (zero padded) row for c < NCH and
 * runs ONE fma chain over its 4*NCH elements in index order; the LPR partials are then combined
 * with an xor butterfly, offsets LPR/2 ... 1.  (hannoy_amd/csrc/hny_kernels.hip: row_partial,
 * butterfly_f32) */
float wave_reduce_scalar(int op, const void *a, const void *b, uint32_t dim) {
  uint32_t dim4 = (dim + 3) / 4, lpr = wave_lpr(dim);
  uint32_t nch = wave_nch(dim4, lpr);
  float v[64];
  for (uint32_t t = 0; t < lpr; t++) {
    float acc = 0.0f;
    for (uint32_t c = 0; c < nch; c++) {
      uint32_t f = c * lpr + t;
      for (uint32_t j = 0; j < 4; j++) {
        uint32_t e = 4 * f + j;
        float x = e < dim ? loadf(a, e) : 0.0f, y = e < dim ? loadf(b, e) : 0.0f;
        if (op == WOP_DOT) {
          acc = fmaf(x, y, acc);
        } else if (op == WOP_EUCLID) {
          float d = x - y;
          acc = fmaf(d, d, acc);
        } else {
          float d = fabsf(x - y);
          acc = acc + d;
        }
      }
    }
    v[t] = acc;
  }
  for (uint32_t off = lpr / 2; off >= 1; off >>= 1) {
    float w[64];
    for (uint32_t t = 0; t < lpr; t++) w[t] = v[t] + v[t ^ off];
    for (uint32_t t = 0; t < lpr; t++) v[t] = w[t];
  }
  return v[0];
}

#if defined(__AVX2__) && defined(__FMA__)
/* The same arithmetic, eight lanes of the group per AVX register (round 5: the full-size parity runs — 10M x 128 —
 * spend their time here).  A block of 8 lanes t0..t0+7 and chunk c own 32 consecutive floats; an 8 x 4 transpose
 * turns them into J0..J3 (element j of every lane), in lane order (t0 t2 t4 t6 | t1 t3 t5 t7) for all four alike,
 * so each lane's fma chain runs in index order exactly as in wave_reduce_scalar.  Butterfly: offsets >= 8 pair
 * whole blocks; 4, 2, 1 are position swaps i^2, i^1 inside the 128-bit halves and the swap of the halves in that
 * lane order.  a + b == b + a, so every lane ends with the same bits: lane 0 is returned.
 * tests/test_oracle_kat.py checks it against the scalar form bit for bit over dims 1..4096. */
static inline void transpose_8x4(__m256 r0, __m256 r1, __m256 r2, __m256 r3, __m256 (&J)[4]) {
  __m256 a = _mm256_unpacklo_ps(r0, r1), b = _mm256_unpacklo_ps(r2, r3);
  __m256 c = _mm256_unpackhi_ps(r0, r1), d = _mm256_unpackhi_ps(r2, r3);
  J[0] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(a), _mm256_castps_pd(b)));
  J[1] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(a), _mm256_castps_pd(b)));
  J[2] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(c), _mm256_castps_pd(d)));
  J[3] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(c), _mm256_castps_pd(d)));
}
template <int OP>
static float wave_reduce_avx(const float *a, const float *b, uint32_t lpr, uint32_t nch) {
  const uint32_t nb = lpr / 8;
  __m256 acc[8];
  const __m256 sign = _mm256_set1_ps(-0.0f);
  for (uint32_t k = 0; k < nb; k++) {
    __m256 s = _mm256_setzero_ps();
    for (uint32_t c = 0; c < nch; c++) {
      const float *pa = a + 4 * ((size_t)c * lpr + 8 * k), *pb = b + 4 * ((size_t)c * lpr + 8 * k);
      __m256 X[4], Y[4];
      transpose_8x4(_mm256_loadu_ps(pa), _mm256_loadu_ps(pa + 8), _mm256_loadu_ps(pa + 16), _mm256_loadu_ps(pa + 24), X);
      transpose_8x4(_mm256_loadu_ps(pb), _mm256_loadu_ps(pb + 8), _mm256_loadu_ps(pb + 16), _mm256_loadu_ps(pb + 24), Y);
      for (int j = 0; j < 4; j++) {
        if (OP == WOP_DOT) {
          s = _mm256_fmadd_ps(X[j], Y[j], s);
        } else if (OP == WOP_EUCLID) {
          __m256 d = _mm256_sub_ps(X[j], Y[j]);
          s = _mm256_fmadd_ps(d, d, s);
        } else {
          s = _mm256_add_ps(s, _mm256_andnot_ps(sign, _mm256_sub_ps(X[j], Y[j])));
        }
      }
    }
    acc[k] = s;
  }
  for (uint32_t off = nb / 2; off >= 1; off >>= 1) { /* lane offsets lpr/2 .. 8 */
    __m256 w[8];
    for (uint32_t k = 0; k < nb; k++) w[k] = _mm256_add_ps(acc[k], acc[k ^ off]);
    for (uint32_t k = 0; k < nb; k++) acc[k] = w[k];
  }
  __m256 v = acc[0];
  v = _mm256_add_ps(v, _mm256_permute_ps(v, 0x4E));        /* offset 4 */
  v = _mm256_add_ps(v, _mm256_permute_ps(v, 0xB1));        /* offset 2 */
  v = _mm256_add_ps(v, _mm256_permute2f128_ps(v, v, 1));   /* offset 1 */
  return _mm256_cvtss_f32(v);
}
float wave_reduce(int op, const void *a, const void *b, uint32_t dim) {
  const uint32_t dim4 = (dim + 3) / 4, lpr = wave_lpr(dim), nch = wave_nch(dim4, lpr);
  const size_t padded = (size_t)4 * lpr * nch;
  const float *fa = (const float *)a, *fb = (const float *)b;
  float ta[4096], tb[4096];
  if (padded > 4096) return wave_reduce_scalar(op, a, b, dim);
  if (padded != dim) { /* the zero padding the kernel's lanes see beyond the row */
    memcpy(ta, a, (size_t)dim * 4);
    memcpy(tb, b, (size_t)dim * 4);
    memset(ta + dim, 0, (padded - dim) * 4);
    memset(tb + dim, 0, (padded - dim) * 4);
    fa = ta;
    fb = tb;
  }
  if (op == WOP_DOT) return wave_reduce_avx<WOP_DOT>(fa, fb, lpr, nch);
  if (op == WOP_EUCLID) return wave_reduce_avx<WOP_EUCLID>(fa, fb, lpr, nch);
  return wave_reduce_avx<WOP_MANHATTAN>(fa, fb, lpr, nch);
}
#else
float wave_reduce(int op, const void *a, const void *b, uint32_t dim) { return wave_reduce_scalar(op, a, b, dim); }
#endif

/* ------------------------------------------------------------------ */
/* codecs                                                              */
/* ------------------------------------------------------------------ */
inline bool is_binary_metric(int metric) { return metric >= ORC_HAMMING; }
inline size_t vec_bytes(int metric, uint32_t dim) {
  if (is_binary_metric(metric)) return (size_t)((dim + 63) / 64) * 8; /* binary.rs:80-94 pads to u64 words */
  return (size_t)dim * 4;                                             /* f32.rs:9-55 */
}
inline size_t hdr_bytes(int metric) {
  return metric == ORC_HAMMING ? 8 : 4; /* hamming.rs:21-25 idx: usize ; others one f32 */
}

/* popcount(u ^ v) over the whole vector. hamming.rs:55-85 works on u64 words + byte tail
 * (no tail: codecs pad to 8 B); the BQ kernels go byte by byte — same total. */
inline uint32_t xor_popcount(const void *u, const void *v, size_t bytes) {
  uint32_t c = 0;
  size_t w = bytes / 8;
  for (size_t i = 0; i < w; i++) c += (uint32_t)__builtin_popcountll(load64(u, i) ^ load64(v, i));
  const uint8_t *ub = (const uint8_t *)u, *vb = (const uint8_t *)v;
  for (size_t i = w * 8; i < bytes; i++) c += (uint32_t)__builtin_popcount((unsigned)(ub[i] ^ vb[i]));
  return c;
}

/* simple.rs:119-131 dot_product_binary_quantized: per byte ones(!(u^v)) - zeros(!(u^v)) */
inline float bq_dot(const void *u, const void *v, size_t bytes) {
  int32_t pop = (int32_t)xor_popcount(u, v, bytes);
  int32_t bits = (int32_t)(bytes * 8);
  return (float)(bits - 2 * pop);
}

struct Dist {
  int metric;
  int order;
  uint32_t dim;
  size_t vbytes;

  float f32_dot(const void *a, const void *b) const {
    return order == ORC_ORDER_X86 ? dot_x86(a, b, dim) : wave_reduce(WOP_DOT, a, b, dim);
  }
  /* D::distance(p, q) */
  float operator()(const void *pv, const void *ph, const void *qv, const void *qh) const {
    switch (metric) {
      case ORC_COSINE: { /* cosine.rs:40-56 */
        float pn = loadf(ph, 0), qn = loadf(qh, 0);
        float pq = f32_dot(pv, qv);
        float pnqn = pn * qn;
        if (pnqn > 1.1920929e-07f /* f32::EPSILON */) {
          float c = pq / pnqn;
          /* f32::clamp(-1, 1): NaN stays NaN */
          if (c < -1.0f) c = -1.0f;
          if (c > 1.0f) c = 1.0f;
          return (1.0f - c) / 2.0f;
        }
        return 0.0f;
      }
      case ORC_EUCLIDEAN: /* euclidean.rs:42-44 */
        return order == ORC_ORDER_X86 ? euclid_x86(pv, qv, dim)
                                      : wave_reduce(WOP_EUCLID, pv, qv, dim);
      case ORC_MANHATTAN: { /* manhattan.rs:41-43 scalar left-to-right */
        if (order == ORC_ORDER_WAVE) return wave_reduce(WOP_MANHATTAN, pv, qv, dim);
        float s = 0.0f;
        for (uint32_t i = 0; i < dim; i++) {
          float d = fabsf(loadf(pv, i) - loadf(qv, i));
          s = s + d;
        }
        return s;
      }
      case ORC_HAMMING: { /* hamming.rs:44-47 */
        float d = (float)xor_popcount(pv, qv, vbytes);
        return d / (float)(vbytes * 8); /* Binary::len = bytes/8*64, binary.rs:67-69 */
      }
      case ORC_BQ_COSINE: { /* binary_quantized_cosine.rs:44-59 */
        float pn = loadf(ph, 0), qn = loadf(qh, 0);
        float pq = bq_dot(pv, qv, vbytes);
        float pnqn = pn * qn;
        if (pnqn != 0.0f) {
          float c = pq / pnqn;
          return (1.0f - c) / 2.0f;
        }
        return 0.0f;
      }
      case ORC_BQ_EUCLIDEAN: /* binary_quantized_euclidean.rs:76-83 */
        return (float)(xor_popcount(pv, qv, vbytes) * 4u);
      case ORC_BQ_MANHATTAN: /* binary_quantized_manhattan.rs:72-79 */
        return (float)(xor_popcount(pv, qv, vbytes) * 2u);
    }
    return 0.0f;
  }
};

/* ------------------------------------------------------------------ */
/* graph state                                                         */
/* ------------------------------------------------------------------ */
struct Link {
  float d;
  uint32_t id; /* dense slot */
};
/* ordered_float.rs:25-29 + tuple order of ScoredLink (hnsw.rs:30) */
inline uint64_t link_key(const Link &l) { return ((uint64_t)f32_bits(l.d) << 32) | l.id; }

struct NodeList {
  std::vector<Link> links;       /* NodeState<M0>.links (hnsw.rs:33-35) */
  std::atomic<uint8_t> lock{0};  /* stands in for papaya's per-key CAS (hnsw.rs:555) */
  std::atomic<uint8_t> present{0};
};

struct Builder {
  orc_opts o;
  orc_items it;
  Dist dist;
  uint32_t n = 0;
  std::vector<uint8_t> level;
  uint32_t max_level = 0;
  std::vector<uint32_t> entry_points;               /* slots, ascending */
  std::vector<std::vector<int32_t>> idx;            /* [layer][slot] -> list index or -1 */
  std::vector<std::unique_ptr<NodeList[]>> lists;   /* [layer][list index] */
  std::vector<std::vector<uint32_t>> owner;         /* [layer][list index] -> slot */
  std::atomic<uint64_t> n_evals{0}, n_links{0};
  bool threaded = false;
  /* incremental builds only (Appendix B of SURVEY.md): the previous graph as stored in LMDB */
  bool incremental = false;
  std::vector<std::vector<std::vector<uint32_t>>> disk; /* [layer][slot] -> old Links ids (ascending) */
  std::vector<std::vector<uint8_t>> has_disk;           /* [layer][slot] a Links record exists */
  std::vector<uint8_t> has_vec;                         /* slot still has an Item record */

  const uint8_t *vec(uint32_t s) const { return (const uint8_t *)it.vectors + (size_t)s * it.stride; }
  const uint8_t *hdr(uint32_t s) const {
    return (const uint8_t *)it.headers + (size_t)s * it.header_size;
  }
  float d_items(uint32_t a, uint32_t b, uint64_t &ctr) const {
    ctr++;
    return dist(vec(a), hdr(a), vec(b), hdr(b));
  }
  uint32_t cap(uint32_t layer_or_level) const { return layer_or_level == 0 ? o.M0 : o.M; }

  NodeList *list(uint32_t layer, uint32_t slot) {
    if (layer >= idx.size()) return nullptr;
    int32_t li = idx[layer][slot];
    return li < 0 ? nullptr : &lists[layer][li];
  }
  void lock(NodeList *l) {
    if (!threaded) return;
    while (l->lock.exchange(1, std::memory_order_acquire)) {
    }
  }
  void unlock(NodeList *l) {
    if (!threaded) return;
    l->lock.store(0, std::memory_order_release);
  }
};

struct Scratch {
  /* RoaringBitmap of one walk: a bitset over the slots + the words it touched (cleared by that list).  Rounds 1-4
   * kept one u32 epoch stamp per slot: 40 MB per thread at 10M items, a cache miss per neighbour looked at; the
   * bitset of the same index is 1.25 MB. */
  std::vector<uint64_t> bits;
  std::vector<uint32_t> touched;
  uint64_t evals = 0;
  uint64_t evals_walk = 0; /* the hnsw.rs:476,503 call sites alone (schedule-determined) */
  std::vector<uint32_t> nbuf;
  /* diagnostics (ORC_TRACE_EVALS=path, scripts/r3_row_overlap.py): which rows every level-0 ef-walk of the
   * batch-synchronous schedule scores — records [0xFFFFFFFF, query slot, first entry point, slots...] */
  std::vector<uint32_t> trace;
  bool tracing = false;
  uint32_t trace_q = 0;
  void begin(uint32_t n) {
    const size_t w = ((size_t)n + 63) / 64;
    if (bits.size() != w) {
      bits.assign(w, 0);
      touched.clear();
    }
    if (touched.size() > w / 8) {
      std::fill(bits.begin(), bits.end(), 0);
    } else {
      for (uint32_t t : touched) bits[t] = 0;
    }
    touched.clear();
  }
  bool seen(uint32_t s) const { return (bits[s >> 6] >> (s & 63)) & 1u; }
  /* RoaringBitmap::insert → true if newly inserted */
  bool visit(uint32_t s) {
    uint64_t &w = bits[s >> 6];
    const uint64_t m = 1ull << (s & 63);
    if (w & m) return false;
    if (!w) touched.push_back(s >> 6);
    w |= m;
    return true;
  }
};

/* candidates: BinaryHeap<(Reverse<OrderedFloat>, ItemId)> (hnsw.rs:469): max-heap, so the top is
 * the smallest distance and, among equal distances, the LARGEST id. */
struct CandLess {
  bool operator()(const Link &a, const Link &b) const {
    uint32_t da = f32_bits(a.d), db = f32_bits(b.d);
    if (da != db) return da > db; /* Reverse */
    return a.id < b.id;
  }
};

/* hnsw.rs:460-518 walk_layer; returns res as a vector sorted ascending by (bits(d), id) */
template <class QDist>
void walk_layer(Builder &B, Scratch &S, const QDist &qd, const std::vector<uint32_t> &eps,
                uint32_t layer, size_t ef, std::vector<Link> &res) {
  std::priority_queue<Link, std::vector<Link>, CandLess> cand;
  res.clear();
  S.begin(B.n);
  auto res_insert = [&](const Link &l) {
    uint64_t k = link_key(l);
    auto pos = std::lower_bound(res.begin(), res.end(), k,
                                [](const Link &a, uint64_t key) { return link_key(a) < key; });
    res.insert(pos, l);
  };
  const bool tr = S.tracing && layer == 0 && ef > 1;
  if (tr) {
    S.trace.push_back(0xFFFFFFFFu);
    S.trace.push_back(S.trace_q);
    S.trace.push_back(eps.empty() ? 0u : eps[0]);
  }
  for (uint32_t ep : eps) { /* :474-481 — no capacity check on res here */
    Link l{qd(ep, S.evals), ep};
    S.evals_walk++;
    if (tr) S.trace.push_back(ep);
    cand.push(l);
    res_insert(l);
    S.visit(ep);
  }
  while (!cand.empty()) {
    float f = cand.top().d;
    float f_max = res.back().d; /* res.peek_max() :484 */
    if (f > f_max) break;       /* raw f32 compare :485 */
    uint32_t c = cand.top().id;
    cand.pop();
    /* get_neighbours :428-456 — on-disk Links (ascending ids) first, then the in-memory list in
     * insertion order; an absent in-memory entry is created lazily (:449-452) */
    S.nbuf.clear();
    if (B.incremental && layer < B.disk.size() && B.has_disk[layer][c])
      for (uint32_t p : B.disk[layer][c]) S.nbuf.push_back(p);
    if (NodeList *nl = B.list(layer, c)) {
      B.lock(nl);
      if (B.incremental) nl->present.store(1, std::memory_order_relaxed);
      for (const Link &l : nl->links) S.nbuf.push_back(l.id);
      B.unlock(nl);
    }
    /* (hints only, no semantics: at 10M items every stamp, row and header below is a cache miss; issuing them
     * together overlaps the misses the loop would otherwise take one at a time) */
    for (uint32_t p : S.nbuf) __builtin_prefetch(&S.bits[p >> 6]);
    for (uint32_t p : S.nbuf)
      if (!S.seen(p)) {
        const uint8_t *v = B.vec(p);
        for (size_t o = 0; o < B.dist.vbytes; o += 64) __builtin_prefetch(v + o);
        __builtin_prefetch(B.hdr(p));
      }
    if (!cand.empty())
      if (NodeList *nx = B.list(layer, cand.top().id)) __builtin_prefetch(nx);
    for (uint32_t p : S.nbuf) {
      if (!S.visit(p)) continue; /* :493 */
      if (B.incremental && !B.has_vec[p]) continue; /* MissingKey => deleted item, :498-502 */
      float d = qd(p, S.evals);  /* :503 */
      S.evals_walk++;
      if (tr) S.trace.push_back(p);
      if (res.size() < ef || d < f_max) { /* :505, f_max captured once per pop */
        Link l{d, p};
        cand.push(l);
        if (res.size() == ef) { /* push_pop_max :508-509 */
          res_insert(l);
          res.pop_back();
        } else {
          res_insert(l); /* :511 */
        }
      }
    }
  }
}

/* hnsw.rs:565-597 robust_prune */
void robust_prune(Builder &B, std::vector<Link> cands, uint32_t cap, uint64_t &evals,
                  std::vector<Link> &selected) {
  std::sort(cands.begin(), cands.end(),
            [](const Link &a, const Link &b) { return link_key(a) < link_key(b); });
  selected.clear();
  for (size_t ci = 0; ci < cands.size(); ci++) { /* pop from the back of the descending sort = ascending */
    const Link &c = cands[ci];
    if (selected.size() == cap) break;
    if (ci + 2 < cands.size()) { /* hint: the row two candidates ahead */
      const uint8_t *v = B.vec(cands[ci + 2].id);
      for (size_t o = 0; o < B.dist.vbytes; o += 64) __builtin_prefetch(v + o);
      __builtin_prefetch(B.hdr(cands[ci + 2].id));
    }
    bool ok = true;
    for (const Link &i : selected) {
      float d = B.d_items(c.id, i.id, evals);
      float da = d * B.o.alpha;
      if (f32_bits(da) < f32_bits(c.d)) { /* OrderedFloat(d*alpha) < dist_to_query :585 */
        ok = false;
        break;
      }
    }
    static const bool link_stats = std::getenv("ORC_PRUNE_LINK_STATS") != nullptr; /* (not per rejection: getenv scans environ) */
    if (ok) selected.push_back(c);
    else if (link_stats) {
      /* diagnostics (scripts/r3_prune_links.py): could the rejection have been read off STORED links — a
       * selected s that violates and has c in its layer-0 list (with its distance), or sits in c's list? */
      static std::atomic<uint64_t> n_rej{0}, via_s{0}, via_c{0}, via_any{0}, n_seen{0};
      bool fs = false, fc = false;
      for (const Link &i : selected) {
        uint64_t dummy = 0;
        float da = B.d_items(c.id, i.id, dummy) * B.o.alpha;
        if (!(f32_bits(da) < f32_bits(c.d))) continue;
        if (NodeList *nl = B.list(0, i.id))
          for (const Link &l : nl->links) fs = fs || l.id == c.id;
        if (NodeList *nl = B.list(0, c.id))
          for (const Link &l : nl->links) fc = fc || l.id == i.id;
      }
      n_rej++;
      via_s += fs;
      via_c += fc;
      via_any += (fs || fc);
      if ((++n_seen & 0xFFFFF) == 0)
        std::fprintf(stderr, "[prune links] rejected %llu: a violating selected s lists c %.3f, c lists a violating s %.3f, either %.3f\n",
                     (unsigned long long)n_rej.load(), (double)via_s / n_rej, (double)via_c / n_rej, (double)via_any / n_rej);
    }
  }
}

/* hnsw.rs:523-560 add_link */
void add_link(Builder &B, uint32_t p, Link q, uint32_t layer, uint64_t &evals) {
  if (p == q.id) return;              /* :530 */
  NodeList *nl = B.list(layer, p);    /* :534 layer missing → no-op */
  if (!nl) return;
  B.lock(nl);
  nl->present.store(1, std::memory_order_relaxed);
  uint32_t cap = B.cap(layer); /* :540 — the layer being linked */
  if (nl->links.size() < cap) {
    nl->links.push_back(q); /* :542-545, no dedup */
  } else {
    std::vector<Link> pruned; /* :547-552 — q itself is dropped */
    robust_prune(B, nl->links, cap, evals, pruned);
    nl->links.swap(pruned);
  }
  B.unlock(nl);
}

/* hnsw.rs:419-424 add_in_layers_below */
void register_item(Builder &B, uint32_t slot, uint32_t level) {
  for (uint32_t l = 0; l <= level && l < B.idx.size(); l++)
    if (NodeList *nl = B.list(l, slot)) nl->present.store(1, std::memory_order_relaxed);
}

struct Selection {
  std::vector<std::vector<Link>> per_layer; /* index = layer */
};

/* hnsw.rs:291-328 insert, split into its read-only half (search + prune) ... */
void insert_search(Builder &B, Scratch &S, uint32_t q, uint32_t level, Selection &out) {
  S.trace_q = q;
  std::vector<uint32_t> eps(B.entry_points.begin(), B.entry_points.end()); /* :298 */
  auto qd = [&](uint32_t p, uint64_t &ctr) { return B.d_items(q, p, ctr); };
  std::vector<Link> res;
  for (uint32_t l = B.max_level; l > level; l--) { /* :303-307 greedy, ef = 1 */
    walk_layer(B, S, qd, eps, l, 1, res);
    eps.assign(1, res.front().id); /* peek_min */
  }
  out.per_layer.assign(level + 1, {});
  for (int32_t l = (int32_t)level; l >= 0; l--) { /* :312-325 */
    walk_layer(B, S, qd, eps, (uint32_t)l, B.o.ef_construction, res);
    robust_prune(B, res, B.cap(level) /* NB item's top level, :317 */, S.evals, out.per_layer[l]);
    eps.clear();
    for (const Link &s : out.per_layer[l]) eps.push_back(s.id);
  }
}
/* ... and its mutating half (:316-324) */
void insert_apply(Builder &B, uint32_t q, uint32_t level, const Selection &sel, uint64_t &evals,
                  uint64_t &links) {
  for (int32_t l = (int32_t)level; l >= 0; l--)
    for (const Link &s : sel.per_layer[l]) {
      add_link(B, q, s, (uint32_t)l, evals);
      add_link(B, s.id, Link{s.d, q}, (uint32_t)l, evals);
      links += 2; /* build_stats.incr_link_count(2) :323 */
    }
}

} // namespace

struct orc_graph {
  std::vector<uint32_t> rec_item; /* item ids */
  std::vector<uint8_t> rec_layer;
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> nbrs; /* item ids, ascending unique */
  std::vector<uint64_t> raw_offsets;
  std::vector<uint32_t> raw_nbrs;
  std::vector<float> raw_dists;
  std::vector<uint32_t> entry_points; /* item ids */
  uint32_t max_level = 0;
  uint64_t n_evals = 0, n_links = 0, n_evals_walk = 0;
};

extern "C" {

uint32_t orc_batch_size(double frac, uint32_t bmax, uint64_t n_done) {
  if (bmax == 0) return 1;
  double b = std::floor(frac * (double)n_done);
  if (b < 1.0) b = 1.0;
  if (b > (double)bmax) b = (double)bmax;
  return (uint32_t)b;
}

size_t orc_vector_bytes(int32_t metric, uint32_t dim) { return vec_bytes(metric, dim); }
size_t orc_header_bytes(int32_t metric) { return hdr_bytes(metric); }

void orc_encode_vector(int32_t metric, uint32_t dim, const float *v, void *out) {
  if (!is_binary_metric(metric)) {
    std::memcpy(out, v, (size_t)dim * 4);
    return;
  }
  uint8_t *o = (uint8_t *)out;
  for (uint32_t base = 0; base < dim; base += 64) {
    uint64_t word = 0;
    uint32_t cnt = std::min<uint32_t>(64, dim - base);
    for (int32_t k = (int32_t)cnt - 1; k >= 0; k--) { /* chunk.iter().rev() */
      word <<= 1;
      uint32_t bits = f32_bits(v[base + k]);
      if (metric == ORC_HAMMING)
        word += (bits < 0x80000000u && bits > 0u) ? 1 : 0; /* binary.rs:87-89 */
      else
        word += (bits >> 31) == 0 ? 1 : 0; /* is_sign_positive, binary_quantized.rs:86 */
    }
    std::memcpy(o, &word, 8); /* to_ne_bytes */
    o += 8;
  }
}

void orc_make_header(int32_t metric, uint32_t dim, const void *vb, void *out) {
  switch (metric) {
    case ORC_COSINE: { /* cosine.rs:36-38,58-60: norm = sqrt(dot(v,v)) */
      float n = sqrtf(dot_x86(vb, vb, dim));
      std::memcpy(out, &n, 4);
      break;
    }
    case ORC_BQ_COSINE: { /* binary_quantized_cosine.rs:40-42,61-63 */
      float n = sqrtf(bq_dot(vb, vb, vec_bytes(metric, dim)));
      std::memcpy(out, &n, 4);
      break;
    }
    case ORC_HAMMING: { /* hamming.rs:40-42 idx = 0usize */
      uint64_t z = 0;
      std::memcpy(out, &z, 8);
      break;
    }
    default: { /* bias = 0.0 (euclidean.rs:38-40 etc.) */
      float z = 0.0f;
      std::memcpy(out, &z, 4);
    }
  }
}

float orc_distance(int32_t metric, int32_t order, uint32_t dim, const void *pv, const void *ph,
                   const void *qv, const void *qh) {
  Dist d{metric, order, dim, vec_bytes(metric, dim)};
  return d(pv, ph, qv, qh);
}
// the wave-order reduction both ways: [0] the AVX2 form the builds use, [1] the scalar statement of the order
void orc_wave_reduce_both(int32_t op, uint32_t dim, const float *a, const float *b, float out[2]) {
  out[0] = wave_reduce(op, a, b, dim);
  out[1] = wave_reduce_scalar(op, a, b, dim);
}
// the same for many pairs of stored items (test helper: >= 1M-pair distance parity)
void orc_distance_pairs(int32_t metric, int32_t order, uint32_t dim, const void *codes, size_t code_stride,
                        const void *headers, size_t header_stride, uint64_t n_pairs, const uint32_t *a,
                        const uint32_t *b, float *out, int32_t threads) {
  Dist d{metric, order, dim, vec_bytes(metric, dim)};
  const uint8_t *cv = (const uint8_t *)codes, *hv = (const uint8_t *)headers;
  unsigned nt = threads > 0 ? (unsigned)threads : 1u;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      for (uint64_t i = n_pairs * t / nt; i < n_pairs * (t + 1) / nt; i++)
        out[i] = d(cv + (size_t)a[i] * code_stride, hv + (size_t)a[i] * header_stride,
                   cv + (size_t)b[i] * code_stride, hv + (size_t)b[i] * header_stride);
    });
  for (auto &x : th) x.join();
}
float orc_dot(int32_t order, uint32_t dim, const float *a, const float *b) {
  return order == ORC_ORDER_X86 ? dot_x86(a, b, dim) : wave_reduce(WOP_DOT, a, b, dim);
}
float orc_sqeuclid(int32_t order, uint32_t dim, const float *a, const float *b) {
  return order == ORC_ORDER_X86 ? euclid_x86(a, b, dim) : wave_reduce(WOP_EUCLID, a, b, dim);
}
float orc_dot_x86_emulated(uint32_t dim, const float *a, const float *b) {
  if (dim >= 32) return dot_avx_emul(a, b, dim);
  if (dim >= 16) return dot_sse_emul(a, b, dim);
  return dot_scalar(a, b, dim);
}
float orc_sqeuclid_x86_emulated(uint32_t dim, const float *a, const float *b) {
  if (dim >= 32) return euclid_avx_emul(a, b, dim);
  if (dim >= 16) return euclid_sse_emul(a, b, dim);
  return euclid_scalar(a, b, dim);
}

/* hnsw.rs:94-110 get_default_probas */
uint32_t orc_level_probas(uint32_t M, float *out, uint32_t cap) {
  float level_factor = 1.0f / logf((float)M + 1.1920929e-07f);
  uint32_t level = 0;
  for (;;) {
    float proba = expf((float)level * (-1.0f / level_factor)) * (1.0f - expf(-1.0f / level_factor));
    if (proba < 1e-09f) break;
    if (level < cap) out[level] = proba;
    level++;
  }
  return level;
}

/* hnsw.rs:160-185: level groups in order; inside a group either the reference's plain insertion
 * (1 thread or rayon-like) or the batch-synchronous schedule of the GPU build */
static void run_schedule(Builder &B, const orc_opts *opts,
                         const std::vector<std::pair<uint32_t, uint32_t>> &ord, uint64_t n_done,
                         std::vector<Scratch> &scratch, uint64_t &evals, uint64_t &links) {
  const size_t n = ord.size();
  size_t pos = 0;
  const int nthreads = (int)scratch.size();

  while (pos < n) {
    /* hnsw.rs:160 chunk_by level */
    size_t gend = pos;
    while (gend < n && ord[gend].second == ord[pos].second) gend++;

    if (opts->batch_max == 0) {
      if (nthreads == 1) {
        Scratch &S = scratch[0];
        for (size_t i = pos; i < gend; i++) {
          uint32_t q = ord[i].first, lvl = ord[i].second;
          Selection sel;
          /* NB: registration precedes the walks in the reference (:309), harmless here */
          insert_search(B, S, q, lvl, sel);
          register_item(B, q, lvl);
          insert_apply(B, q, lvl, sel, S.evals, links);
        }
      } else {
        /* rayon-like: grp.into_par_iter().try_for_each(insert) :172-185 */
        std::atomic<size_t> next{pos};
        std::vector<std::thread> th;
        std::vector<uint64_t> tlinks(nthreads, 0);
        for (int t = 0; t < nthreads; t++)
          th.emplace_back([&, t]() {
            Scratch &S = scratch[t];
            for (;;) {
              size_t i = next.fetch_add(1);
              if (i >= gend) break;
              uint32_t q = ord[i].first, lvl = ord[i].second;
              /* the reference interleaves walk/prune/link per layer; with concurrent threads the
               * per-layer interleaving is kept so that lower walks see this item's upper links */
              std::vector<uint32_t> eps(B.entry_points.begin(), B.entry_points.end());
              auto qd = [&](uint32_t p, uint64_t &ctr) { return B.d_items(q, p, ctr); };
              std::vector<Link> res, sel;
              for (uint32_t l = B.max_level; l > lvl; l--) {
                walk_layer(B, S, qd, eps, l, 1, res);
                eps.assign(1, res.front().id);
              }
              register_item(B, q, lvl);
              for (int32_t l = (int32_t)lvl; l >= 0; l--) {
                walk_layer(B, S, qd, eps, (uint32_t)l, B.o.ef_construction, res);
                robust_prune(B, res, B.cap(lvl), S.evals, sel);
                eps.clear();
                for (const Link &s : sel) {
                  add_link(B, q, s, (uint32_t)l, S.evals);
                  add_link(B, s.id, Link{s.d, q}, (uint32_t)l, S.evals);
                  eps.push_back(s.id);
                  tlinks[t] += 2;
                }
              }
            }
          });
        for (auto &t : th) t.join();
        for (uint64_t v : tlinks) links += v;
      }
      n_done += gend - pos;
      pos = gend;
    } else {
      /* batch-synchronous schedule: every member of a batch searches the same frozen graph,
       * then links are applied in batch order (DESIGN.md "Batch semantics") */
      size_t bsz = orc_batch_size(opts->batch_frac, opts->batch_max, n_done);
      size_t bend = std::min(gend, pos + bsz);
      size_t cnt = bend - pos;
      std::vector<Selection> sels(cnt);
      const char *trace_path = std::getenv("ORC_TRACE_EVALS"); /* diagnostics: the LAST batch's level-0 walks */
      const bool trace_now = trace_path && bend == ord.size();
      for (auto &sc : scratch) sc.tracing = trace_now;
      if (nthreads == 1) {
        for (size_t i = 0; i < cnt; i++)
          insert_search(B, scratch[0], ord[pos + i].first, ord[pos + i].second, sels[i]);
      } else {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
          th.emplace_back([&, t]() {
            for (;;) {
              size_t i = next.fetch_add(1);
              if (i >= cnt) break;
              insert_search(B, scratch[t], ord[pos + i].first, ord[pos + i].second, sels[i]);
            }
          });
        for (auto &t : th) t.join();
      }
      if (trace_now) {
        if (FILE *f = std::fopen(trace_path, "wb")) {
          for (auto &sc : scratch) {
            if (!sc.trace.empty()) std::fwrite(sc.trace.data(), 4, sc.trace.size(), f);
            sc.trace.clear();
            sc.tracing = false;
          }
          std::fclose(f);
        }
      }
      bool was_threaded = B.threaded;
      B.threaded = false; /* apply is sequential by definition */
      for (size_t i = 0; i < cnt; i++) register_item(B, ord[pos + i].first, ord[pos + i].second);
      if (nthreads == 1 || cnt < 256) {
        for (size_t i = 0; i < cnt; i++)
          insert_apply(B, ord[pos + i].first, ord[pos + i].second, sels[i], evals, links);
      } else {
        /* The same sequence of add_link calls, executed by several threads without changing its outcome: an
         * add_link touches only the lists of its TARGET, so calls on different targets commute; every thread
         * walks the whole sequence in batch order and performs the calls whose target it owns (blocks of 16
         * slots, round robin) — per target the order is the sequential one, no list is shared, no lock taken.
         * (Round 5: at 10M items this phase, on one thread, was most of the oracle's build time.) */
        const uint32_t T = (uint32_t)nthreads;
        std::vector<uint64_t> tev(T, 0), tln(T, 0);
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; t++)
          th.emplace_back([&, t]() {
            uint64_t ev = 0, ln = 0;
            for (size_t i = 0; i < cnt; i++) {
              const uint32_t q = ord[pos + i].first, lvl = ord[pos + i].second;
              const bool mine_q = (q >> 4) % T == t;
              for (int32_t l = (int32_t)lvl; l >= 0; l--)
                for (const Link &sl : sels[i].per_layer[l]) {
                  if (mine_q) {
                    add_link(B, q, sl, (uint32_t)l, ev);
                    ln += 2; /* build_stats.incr_link_count(2) :323, counted by the owner of q */
                  }
                  if ((sl.id >> 4) % T == t) add_link(B, sl.id, Link{sl.d, q}, (uint32_t)l, ev);
                }
            }
            tev[t] = ev;
            tln[t] = ln;
          });
        for (auto &x : th) x.join();
        for (uint32_t t = 0; t < T; t++) {
          evals += tev[t];
          links += tln[t];
        }
      }
      B.threaded = was_threaded;
      n_done += cnt;
      pos = bend;
    }
  }
}

/* [3P] rand_chacha 0.3 ChaCha12Rng (rand 0.8.5 StdRng): 12 rounds, 64-bit block counter in words
 * 12-13, stream id 0, output = successive blocks, words in order. */
namespace {
struct ChaCha12 {
  uint32_t key[8];
  uint64_t ctr = 0;
  uint32_t buf[16];
  int pos = 16;
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static inline void qr(uint32_t *s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
  }
  void refill() {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) st[4 + i] = key[i];
    st[12] = (uint32_t)ctr;
    st[13] = (uint32_t)(ctr >> 32);
    st[14] = st[15] = 0;
    uint32_t w[16];
    std::memcpy(w, st, sizeof w);
    for (int r = 0; r < 6; r++) {
      qr(w, 0, 4, 8, 12); qr(w, 1, 5, 9, 13); qr(w, 2, 6, 10, 14); qr(w, 3, 7, 11, 15);
      qr(w, 0, 5, 10, 15); qr(w, 1, 6, 11, 12); qr(w, 2, 7, 8, 13); qr(w, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) buf[i] = w[i] + st[i];
    ctr++;
    pos = 0;
  }
  uint32_t next_u32() {
    if (pos == 16) refill();
    return buf[pos++];
  }
};
} // namespace

static void draw_levels_impl(const uint8_t *seed32, uint64_t seed_u64, uint64_t skip, uint32_t M, uint64_t n,
                             uint8_t *out);
void orc_draw_levels(const uint8_t *seed32, uint64_t seed_u64, uint32_t M, uint64_t n, uint8_t *out) {
  draw_levels_impl(seed32, seed_u64, 0, M, n, out);
}
void orc_draw_levels_skip(const uint8_t *seed32, uint64_t skip, uint32_t M, uint64_t n, uint8_t *out) {
  draw_levels_impl(seed32, 0, skip, M, n, out);
}
/* rng.gen::<f32>() (rand 0.8.5 Standard, float_impls: 24 random bits times 2^-24) from
 * StdRng::from_seed(seed32) after `skip` u32 words — the generator of the reference's random test
 * vectors (src/tests/writer.rs:137) */
void orc_gen_f32(const uint8_t *seed32, uint64_t skip, uint64_t n, float *out) {
  ChaCha12 rng;
  std::memcpy(rng.key, seed32, 32);
  for (uint64_t i = 0; i < skip; i++) (void)rng.next_u32();
  for (uint64_t i = 0; i < n; i++) out[i] = (float)(rng.next_u32() >> 8) * (1.0f / 16777216.0f);
}
static void draw_levels_impl(const uint8_t *seed32, uint64_t seed_u64, uint64_t skip, uint32_t M, uint64_t n,
                             uint8_t *out) {
  uint8_t seed[32];
  if (seed32) {
    std::memcpy(seed, seed32, 32);
  } else { /* SeedableRng::seed_from_u64: PCG32 expansion */
    uint64_t state = seed_u64;
    for (int c = 0; c < 8; c++) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
      std::memcpy(seed + 4 * c, &x, 4);
    }
  }
  ChaCha12 rng;
  std::memcpy(rng.key, seed, 32);
  for (uint64_t i = 0; i < skip; i++) (void)rng.next_u32();
  float probas[64];
  uint32_t np = orc_level_probas(M, probas, 64);
  /* WeightedIndex::new: running totals before each further weight; Uniform::new(0, total) */
  std::vector<float> cum;
  float total = probas[0];
  for (uint32_t i = 1; i < np; i++) {
    cum.push_back(total);
    total = total + probas[i];
  }
  uint32_t mr = (0xFFFFFFFFu >> 9) | 0x3F800000u;
  float max_rand;
  std::memcpy(&max_rand, &mr, 4);
  max_rand -= 1.0f;
  float scale = total - 0.0f;
  while (scale * max_rand + 0.0f >= total) {
    uint32_t b = f32_bits(scale) - 1;
    std::memcpy(&scale, &b, 4);
  }
  for (uint64_t i = 0; i < n; i++) {
    uint32_t u = (rng.next_u32() >> 9) | 0x3F800000u;
    float v12;
    std::memcpy(&v12, &u, 4);
    float x = (v12 - 1.0f) * scale + 0.0f;
    size_t k = 0; /* cumulative_weights.partition_point(|w| w <= &chosen) */
    while (k < cum.size() && cum[k] <= x) k++;
    out[i] = (uint8_t)k;
  }
}

/* [3P] `levels.sort_unstable_by(|(_, a), (_, b)| b.cmp(a))` (hnsw.rs:268) as Rust's standard library
 * performs it since 1.81 (core::slice::sort::unstable: ipnsort) on `(u32, usize)` pairs — restated from
 * the published algorithm, because the order it leaves EQUAL levels in decides the insertion order and
 * with it the graph (the reference's 100-point snapshots, KAT-9).  Only the steps that can reorder equal
 * elements are spelled out: the existing-run check, the pivot choice (median of 3, recursive from 64
 * elements), the branchless cyclic Lomuto partition and the quicksort driver with its equal-to-ancestor
 * partition.  Slices of <= 32 elements go to the library's small-sort (sorting networks on 8 + insertion
 * + bidirectional merge for 16-byte elements): every step of it is stable, so a stable sort stands in.
 * <= 20 elements: insertion sort (stable).  The heapsort fallback (after 2 * ilog2(n) unbalanced
 * partitions) is restated too but never reached by level data (a handful of distinct keys), hence unpinned. */
namespace {
typedef std::pair<uint32_t, uint32_t> LevelPair; /* (slot or id, level) */
struct RustSort {
  /* is_less(a, b) of the comparator |a, b| b.level.cmp(a.level) == Less  <=>  b.level < a.level */
  static bool lt(const LevelPair &a, const LevelPair &b) { return b.second < a.second; }
  static void small_sort(LevelPair *v, size_t n) {
    std::stable_sort(v, v + n, [](const LevelPair &a, const LevelPair &b) { return lt(a, b); });
  }
  static size_t median3(const LevelPair *v, size_t a, size_t b, size_t c) {
    const bool x = lt(v[a], v[b]), y = lt(v[a], v[c]);
    if (x == y) {
      const bool z = lt(v[b], v[c]);
      return (z ^ x) ? c : b;
    }
    return a;
  }
  static size_t median3_rec(const LevelPair *v, size_t a, size_t b, size_t c, size_t n) {
    if (n * 8 >= 64) {
      const size_t n8 = n / 8;
      a = median3_rec(v, a, a + n8 * 4, a + n8 * 7, n8);
      b = median3_rec(v, b, b + n8 * 4, b + n8 * 7, n8);
      c = median3_rec(v, c, c + n8 * 4, c + n8 * 7, n8);
    }
    return median3(v, a, b, c);
  }
  static size_t choose_pivot(const LevelPair *v, size_t len) {
    const size_t d = len / 8, a = 0, b = d * 4, c = d * 7;
    return len < 64 ? median3(v, a, b, c) : median3_rec(v, a, b, c, d);
  }
  /* partition_lomuto_branchless_cyclic over v[1..len) with the pivot swapped to v[0] */
  /* less_or_equal: the `|a, b| !is_less(b, a)` form of the equal-to-ancestor partition */
  static size_t partition(LevelPair *v, size_t len, size_t pivot_pos, bool less_or_equal) {
    auto less = [less_or_equal](const LevelPair &a, const LevelPair &b) { return less_or_equal ? !lt(b, a) : lt(a, b); };
    std::swap(v[0], v[pivot_pos]);
    const LevelPair pivot = v[0];
    LevelPair *w = v + 1;
    const size_t n = len - 1;
    size_t num_lt = 0;
    if (n) {
      const LevelPair gap_value = w[0];
      size_t gap = 0;
      for (size_t right = 1; right < n; right++) {
        const bool r_lt = less(w[right], pivot);
        w[gap] = w[num_lt];
        w[num_lt] = w[right];
        gap = right;
        num_lt += r_lt ? 1 : 0;
      }
      const bool r_lt = less(gap_value, pivot);
      w[gap] = w[num_lt];
      w[num_lt] = gap_value;
      num_lt += r_lt ? 1 : 0;
    }
    std::swap(v[0], v[num_lt]);
    return num_lt;
  }
  static void quicksort(LevelPair *v, size_t len, const LevelPair *ancestor, uint32_t limit) {
    for (;;) {
      if (len <= 32) {
        small_sort(v, len);
        return;
      }
      if (limit == 0) { /* heapsort::heapsort after 2 * ilog2(n) unbalanced partitions (not reached by level data) */
        for (size_t i = len + len / 2; i-- > 0;) {
          size_t node;
          if (i >= len) {
            node = i - len;
          } else {
            std::swap(v[0], v[i]);
            node = 0;
          }
          const size_t hl = std::min(i, len);
          for (;;) { /* sift_down */
            size_t child = 2 * node + 1;
            if (child >= hl) break;
            if (child + 1 < hl && lt(v[child], v[child + 1])) child++;
            if (!lt(v[node], v[child])) break;
            std::swap(v[node], v[child]);
            node = child;
          }
        }
        return;
      }
      limit--;
      const size_t pp = choose_pivot(v, len);
      if (ancestor && !lt(*ancestor, v[pp])) {
        const size_t num_le = partition(v, len, pp, true);
        v += num_le + 1;
        len -= num_le + 1;
        ancestor = nullptr;
        continue;
      }
      const size_t num_lt = partition(v, len, pp, false);
      quicksort(v, num_lt, ancestor, limit);
      ancestor = v + num_lt; /* the pivot, in its final place */
      v += num_lt + 1;
      len -= num_lt + 1;
    }
  }
  static void sort(std::vector<LevelPair> &vec) {
    LevelPair *v = vec.data();
    const size_t len = vec.size();
    if (len < 2) return;
    if (len <= 20) { /* insertion_sort_shift_left: stable */
      small_sort(v, len);
      return;
    }
    /* find_existing_run: a fully sorted (or strictly descending) input is returned as is (reversed) */
    size_t run = 2;
    const bool desc = lt(v[1], v[0]);
    if (desc) while (run < len && lt(v[run], v[run - 1])) run++;
    else while (run < len && !lt(v[run], v[run - 1])) run++;
    if (run == len) {
      if (desc) std::reverse(vec.begin(), vec.end());
      return;
    }
    uint32_t lg = 0;
    for (size_t x = len | 1; x > 1; x >>= 1) lg++;
    quicksort(v, len, nullptr, 2 * lg);
  }
};
/* the batch-synchronous schedule (batch_max > 1) takes the items of a level group in a fixed pseudo-random
 * order, so that a batch is a sample of the group whatever the id order is (the product: hny_rust_sort.h
 * shuffle_level_groups; any order inside a group is a legitimate parallel execution of hnsw.rs:172-185).
 * Fisher-Yates on splitmix64, seeded by the group's level and size. */
uint64_t splitmix64_next(uint64_t &x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void shuffle_level_groups(std::vector<LevelPair> &v) {
  size_t b = 0;
  while (b < v.size()) {
    size_t e = b;
    while (e < v.size() && v[e].second == v[b].second) e++;
    uint64_t st = 0x68616E6E6F79ull ^ ((uint64_t)v[b].second << 48) ^ (uint64_t)(e - b);
    for (size_t i = e - b; i > 1; i--) std::swap(v[b + i - 1], v[b + (size_t)(splitmix64_next(st) % i)]);
    b = e;
  }
}
/* orc_opts.level_sort: 0 = ties in ascending id order (stable); 1 = Rust >= 1.81 sort_unstable_by */
void sort_levels(std::vector<LevelPair> &v, int32_t level_sort, const orc_opts *o) {
  if (level_sort == 1) RustSort::sort(v);
  else std::stable_sort(v.begin(), v.end(), [](const LevelPair &a, const LevelPair &b) { return a.second > b.second; });
  if (o->batch_max > 1 && !o->no_shuffle) shuffle_level_groups(v);
}
} // namespace

void orc_rust_sort_levels(uint32_t *ids, uint32_t *levels, uint64_t n) {
  std::vector<LevelPair> v(n);
  for (uint64_t i = 0; i < n; i++) v[i] = {ids[i], levels[i]};
  RustSort::sort(v);
  for (uint64_t i = 0; i < n; i++) {
    ids[i] = v[i].first;
    levels[i] = v[i].second;
  }
}

int orc_build(const orc_opts *opts, const orc_items *items, orc_graph **out) {
  if (!opts || !items || !out) return -1;
  if (opts->M == 0 || opts->M0 < opts->M) return -2;
  Builder B;
  B.o = *opts;
  B.it = *items;
  B.dist = Dist{opts->metric, opts->order, opts->dim, vec_bytes(opts->metric, opts->dim)};
  B.n = (uint32_t)items->n;
  B.threaded = opts->threads > 1;
  uint32_t n = B.n;
  auto g = std::make_unique<orc_graph>();

  /* hnsw.rs:141-149 levels (injected) */
  B.level.assign(items->levels, items->levels + n);
  uint32_t cur_max = 0;
  for (uint32_t s = 0; s < n; s++) cur_max = std::max<uint32_t>(cur_max, B.level[s]);

  /* hnsw.rs:268 sort by level desc. The reference's sort is unstable; ties are taken in
   * ascending id order here (= the order KAT-1 implies for small inputs). */
  std::vector<uint32_t> order(n);
  {
    std::vector<LevelPair> lv(n);
    for (uint32_t s = 0; s < n; s++) lv[s] = {s, B.level[s]};
    sort_levels(lv, opts->level_sort, opts);
    for (uint32_t s = 0; s < n; s++) order[s] = lv[s].first;
  }

  if (n > 0) {
    B.max_level = cur_max; /* :272-276 fresh DB: max_level starts at 0 */
    /* layers: one map per level 0..=max_level (:279-281) */
    B.idx.resize(B.max_level + 1);
    B.lists.resize(B.max_level + 1);
    B.owner.resize(B.max_level + 1);
    for (uint32_t l = 0; l <= B.max_level; l++) {
      B.idx[l].assign(n, -1);
      uint32_t cnt = 0;
      for (uint32_t s = 0; s < n; s++)
        if (B.level[s] >= l) {
          B.idx[l][s] = (int32_t)cnt++;
          B.owner[l].push_back(s);
        }
      B.lists[l] = std::unique_ptr<NodeList[]>(new NodeList[cnt]);
    }
    /* :278-287 every item at max_level is an entry point, pre-registered in all layers */
    for (uint32_t s = 0; s < n; s++)
      if (B.level[s] == B.max_level) {
        B.entry_points.push_back(s);
        register_item(B, s, B.max_level);
      }
  }

  uint64_t evals = 0, links = 0;
  int nthreads = std::max(1, opts->threads);
  std::vector<Scratch> scratch(nthreads);
  {
    std::vector<std::pair<uint32_t, uint32_t>> ord(n);
    for (uint32_t i = 0; i < n; i++) ord[i] = {order[i], B.level[order[i]]};
    run_schedule(B, opts, ord, 0, scratch, evals, links);
  }
  uint64_t evals_walk = 0;
  for (auto &s : scratch) {
    evals += s.evals;
    evals_walk += s.evals_walk;
  }

  /* hnsw.rs:191-213 write loop, emitted sorted by (item, layer) = LMDB key order (key.rs:54-66) */
  g->max_level = B.max_level;
  for (uint32_t s : B.entry_points) g->entry_points.push_back(items->ids[s]);
  g->offsets.push_back(0);
  g->raw_offsets.push_back(0);
  std::vector<uint32_t> tmp;
  for (uint32_t s = 0; s < n; s++)
    for (uint32_t l = 0; l < B.idx.size(); l++) {
      NodeList *nl = B.list(l, s);
      if (!nl || !nl->present.load()) continue;
      g->rec_item.push_back(items->ids[s]);
      g->rec_layer.push_back((uint8_t)l);
      tmp.clear();
      for (const Link &k : nl->links) {
        tmp.push_back(items->ids[k.id]);
        g->raw_nbrs.push_back(items->ids[k.id]);
        g->raw_dists.push_back(k.d);
      }
      g->raw_offsets.push_back(g->raw_nbrs.size());
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end()); /* RoaringBitmap::from_iter */
      g->nbrs.insert(g->nbrs.end(), tmp.begin(), tmp.end());
      g->offsets.push_back(g->nbrs.size());
    }
  g->n_evals = evals;
  g->n_evals_walk = evals_walk;
  g->n_links = links;
  *out = g.release();
  return 0;
}

/* Incremental build — Appendix B of SURVEY.md, validated against KAT-2/3/4. */
int orc_build_incremental(const orc_opts *opts, const orc_items *items, const uint32_t *to_insert,
                          uint64_t n_insert, const uint8_t *insert_levels, const uint32_t *to_delete,
                          uint64_t n_delete, const orc_prev_graph *prev, orc_graph **out) {
  if (!opts || !items || !prev || !out) return -1;
  /* universe of ids = current items U deleted items U everything the old graph mentions */
  std::vector<uint32_t> U(items->ids, items->ids + items->n);
  U.insert(U.end(), to_delete, to_delete + n_delete);
  U.insert(U.end(), prev->rec_item, prev->rec_item + prev->n_records);
  if (prev->n_records) U.insert(U.end(), prev->nbrs, prev->nbrs + prev->offsets[prev->n_records]);
  U.insert(U.end(), prev->entry_points, prev->entry_points + prev->n_entry_points);
  std::sort(U.begin(), U.end());
  U.erase(std::unique(U.begin(), U.end()), U.end());
  const uint32_t n = (uint32_t)U.size();
  auto slot = [&](uint32_t id) { return (uint32_t)(std::lower_bound(U.begin(), U.end(), id) - U.begin()); };

  Builder B;
  B.o = *opts;
  B.dist = Dist{opts->metric, opts->order, opts->dim, vec_bytes(opts->metric, opts->dim)};
  B.n = n;
  B.incremental = true;
  const size_t vb = B.dist.vbytes, hb = items->header_size;
  std::vector<uint8_t> vecs((size_t)n * vb, 0), hdrs((size_t)n * hb, 0);
  B.has_vec.assign(n, 0);
  for (uint64_t i = 0; i < items->n; i++) {
    uint32_t s = slot(items->ids[i]);
    std::memcpy(&vecs[(size_t)s * vb], (const uint8_t *)items->vectors + i * items->stride, vb);
    std::memcpy(&hdrs[(size_t)s * hb], (const uint8_t *)items->headers + i * hb, hb);
    B.has_vec[s] = 1;
  }
  B.it.n = n;
  B.it.ids = U.data();
  B.it.vectors = vecs.data();
  B.it.stride = vb;
  B.it.headers = hdrs.data();
  B.it.header_size = hb;
  std::vector<uint8_t> del(n, 0);
  for (uint64_t i = 0; i < n_delete; i++) del[slot(to_delete[i])] = 1;

  /* the previous graph as LMDB holds it */
  const uint32_t MAXL = 15; /* M = 4 draws levels up to 14 (hnsw.rs:94-110) */
  B.disk.assign(MAXL, std::vector<std::vector<uint32_t>>(n));
  B.has_disk.assign(MAXL, std::vector<uint8_t>(n, 0));
  for (uint64_t r = 0; r < prev->n_records; r++) {
    uint32_t l = prev->rec_layer[r], s = slot(prev->rec_item[r]);
    if (l >= MAXL) return -2;
    B.has_disk[l][s] = 1;
    for (uint64_t j = prev->offsets[r]; j < prev->offsets[r + 1]; j++)
      B.disk[l][s].push_back(slot(prev->nbrs[j]));
  }
  /* storage for every (layer, slot); B.idx.size() is the number of layers that "exist"
   * (self.layers.len()) */
  std::vector<std::vector<int32_t>> all_idx(MAXL, std::vector<int32_t>(n));
  std::vector<std::unique_ptr<NodeList[]>> all_lists(MAXL);
  for (uint32_t l = 0; l < MAXL; l++) {
    for (uint32_t s = 0; s < n; s++) all_idx[l][s] = (int32_t)s;
    all_lists[l] = std::unique_ptr<NodeList[]>(new NodeList[n]);
  }
  B.lists = std::move(all_lists);
  auto set_layers = [&](uint32_t count) {
    B.idx.assign(all_idx.begin(), all_idx.begin() + count);
  };

  /* hnsw.rs:141-149: one level per to_insert item, ascending id */
  std::vector<std::pair<uint32_t, uint32_t>> levels; /* (slot, level) */
  uint32_t cur_max_level = 0;
  for (uint64_t i = 0; i < n_insert; i++) {
    levels.push_back({slot(to_insert[i]), insert_levels[i]});
    cur_max_level = std::max<uint32_t>(cur_max_level, insert_levels[i]);
  }
  /* prepare_levels_and_entry_points, hnsw.rs:236-289 */
  uint32_t max_level = prev->max_level;
  std::vector<uint8_t> in_new(n, 0), in_old(n, 0);
  uint32_t n_old = 0, n_new = 0;
  std::vector<uint32_t> del_eps;
  for (uint32_t i = 0; i < prev->n_entry_points; i++) {
    uint32_t s = slot(prev->entry_points[i]);
    if (!in_old[s]) { in_old[s] = 1; n_old++; }
    if (del[s]) del_eps.push_back(s);
    else if (!in_new[s]) { in_new[s] = 1; n_new++; }
  }
  {
    uint32_t l = max_level;
    for (size_t k = 0; k < del_eps.size(); k++) {
      for (;;) {
        if (l < MAXL)
          for (uint32_t s = 0; s < n; s++) { /* iter_layer_links(l): ascending item */
            if (!B.has_disk[l][s]) continue;
            if (!del[s] && !in_new[s]) { /* new_eps.insert(item) returned true */
              in_new[s] = 1;
              n_new++;
              break;
            }
          }
        if (l == 0) break; /* checked_sub */
        l -= 1;
      }
    }
  }
  if (!del_eps.empty() && n_new != n_old) max_level = 0; /* :261-263 */
  for (uint32_t s = 0; s < n; s++)
    if (in_new[s]) levels.push_back({s, max_level}); /* :267 re-index the old entry points */
  sort_levels(levels, opts->level_sort, opts); /* :268 (unstable in Rust; stable == what inputs of <= 20 pairs get) */
  if (cur_max_level > max_level) { /* :272-276 */
    std::fill(in_new.begin(), in_new.end(), 0);
    max_level = cur_max_level;
  }
  B.max_level = max_level;
  set_layers(max_level + 1); /* :279-281 */
  for (auto &pr : levels) {  /* :278, :282-285 upper_layer = take_while(level == max_level) */
    if (pr.second != max_level) break;
    in_new[pr.first] = 1;
    register_item(B, pr.first, max_level);
  }
  for (uint32_t s = 0; s < n; s++)
    if (in_new[s]) B.entry_points.push_back(s); /* :287 */

  /* hnsw.rs:172-185; an update's batches ramp up from one member like a fresh build's (the product's rule,
   * DESIGN.md 1): counting the surviving old records as "already inserted" put a whole update into one batch,
   * whose members cannot see each other.  orc_opts.update_no_ramp: that older rule. */
  uint64_t n_done0 = 0;
  if (opts->update_no_ramp)
      for (uint32_t s = 0; s < n; s++)
        if (B.has_disk[0][s] && !del[s]) n_done0++;
  int nthreads = std::max(1, opts->threads);
  std::vector<Scratch> scratch(nthreads);
  uint64_t evals = 0, links = 0;
  B.threaded = opts->threads > 1;
  run_schedule(B, opts, levels, n_done0, scratch, evals, links);
  B.threaded = false;
  Scratch &S = scratch[0];
  S.evals += evals;
  for (int t = 1; t < nthreads; t++) {
    S.evals += scratch[t].evals;
    S.evals_walk += scratch[t].evals_walk;
  }

  /* fill_gaps_from_deleted, hnsw.rs:334-415 */
  {
    uint32_t need = (uint32_t)B.idx.size();
    for (uint32_t l = 0; l < MAXL; l++)
      for (uint32_t s = 0; s < n; s++)
        if (B.has_disk[l][s]) need = std::max(need, l + 1); /* :352-354 resize layers */
    set_layers(need);
    for (uint32_t s = 0; s < n; s++)      /* iter_links: key order (item, layer) */
      for (uint32_t l = 0; l < MAXL; l++) {
        if (!B.has_disk[l][s] || del[s]) continue; /* :373-375 */
        NodeList *nl = B.list(l, s);
        std::vector<Link> new_links;
        if (nl->present.load()) new_links = nl->links; /* map_guard.get(&id) */
        std::vector<uint8_t> inbm(n, 0);
        for (uint32_t x : B.disk[l][s])
          if (del[x] && B.has_disk[l][x]) /* lmdb.links(x, lvl).unwrap_or_default() */
            for (uint32_t y : B.disk[l][x]) inbm[y] = 1;
        for (uint32_t x : B.disk[l][s]) inbm[x] = 1; /* bitmap |= links */
        std::vector<uint32_t> bm;
        for (uint32_t y = 0; y < n; y++)
          if (inbm[y] && !del[y]) bm.push_back(y); /* bitmap -= to_delete */
        uint32_t thresh = B.cap(l);
        if (bm.size() + new_links.size() <= thresh) { /* :392-400 */
          std::vector<Link> entries;
          for (uint32_t y : bm) entries.push_back(Link{0.0f, y});
          entries.insert(entries.end(), new_links.begin(), new_links.end());
          nl->links.swap(entries);
          nl->present.store(1);
          continue;
        }
        for (uint32_t y : bm) new_links.push_back(Link{B.d_items(s, y, S.evals), y}); /* :403-408 */
        std::vector<Link> pruned;
        robust_prune(B, new_links, B.cap(l), S.evals, pruned);
        nl->links.swap(pruned);
        nl->present.store(1);
      }
  }

  /* DB after the build: old records, overwritten by every in-memory key (hnsw.rs:195-213), minus
   * the Links of deleted items (writer.rs:580, 692-718) */
  auto g = std::make_unique<orc_graph>();
  g->max_level = B.max_level;
  for (uint32_t s : B.entry_points) g->entry_points.push_back(U[s]);
  g->offsets.push_back(0);
  g->raw_offsets.push_back(0);
  std::vector<uint32_t> tmp;
  for (uint32_t s = 0; s < n; s++)
    for (uint32_t l = 0; l < MAXL; l++) {
      if (del[s]) continue;
      NodeList *nl = l < B.idx.size() ? B.list(l, s) : nullptr;
      bool mem = nl && nl->present.load();
      if (!mem && !B.has_disk[l][s]) continue;
      g->rec_item.push_back(U[s]);
      g->rec_layer.push_back((uint8_t)l);
      tmp.clear();
      if (mem) {
        for (const Link &k : nl->links) {
          tmp.push_back(U[k.id]);
          g->raw_nbrs.push_back(U[k.id]);
          g->raw_dists.push_back(k.d);
        }
      } else {
        for (uint32_t y : B.disk[l][s]) tmp.push_back(U[y]);
      }
      g->raw_offsets.push_back(g->raw_nbrs.size());
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      g->nbrs.insert(g->nbrs.end(), tmp.begin(), tmp.end());
      g->offsets.push_back(g->nbrs.size());
    }
  g->n_evals = S.evals;
  g->n_evals_walk = S.evals_walk;
  g->n_links = links;
  *out = g.release();
  return 0;
}

void orc_graph_free(orc_graph *g) { delete g; }
uint64_t orc_graph_n_records(const orc_graph *g) { return g->rec_item.size(); }
uint64_t orc_graph_n_links(const orc_graph *g) { return g->nbrs.size(); }
void orc_graph_export(const orc_graph *g, uint32_t *rec_item, uint8_t *rec_layer, uint64_t *offsets,
                      uint32_t *nbrs) {
  std::memcpy(rec_item, g->rec_item.data(), g->rec_item.size() * 4);
  std::memcpy(rec_layer, g->rec_layer.data(), g->rec_layer.size());
  std::memcpy(offsets, g->offsets.data(), g->offsets.size() * 8);
  if (!g->nbrs.empty()) std::memcpy(nbrs, g->nbrs.data(), g->nbrs.size() * 4);
}
uint64_t orc_graph_n_raw(const orc_graph *g) { return g->raw_nbrs.size(); }
void orc_graph_export_raw(const orc_graph *g, uint64_t *offsets, uint32_t *nbrs, float *dists) {
  std::memcpy(offsets, g->raw_offsets.data(), g->raw_offsets.size() * 8);
  if (!g->raw_nbrs.empty()) {
    std::memcpy(nbrs, g->raw_nbrs.data(), g->raw_nbrs.size() * 4);
    std::memcpy(dists, g->raw_dists.data(), g->raw_dists.size() * 4);
  }
}
uint32_t orc_graph_entry_points(const orc_graph *g, uint32_t *out, uint32_t cap) {
  for (uint32_t i = 0; i < g->entry_points.size() && i < cap; i++) out[i] = g->entry_points[i];
  return (uint32_t)g->entry_points.size();
}
uint32_t orc_graph_max_level(const orc_graph *g) { return g->max_level; }
uint64_t orc_graph_distance_evals(const orc_graph *g) { return g->n_evals; }
uint64_t orc_graph_walk_evals(const orc_graph *g) { return g->n_evals_walk; }
uint64_t orc_graph_links_added(const orc_graph *g) { return g->n_links; }

/* ------------------------------------------------------------------ */
/* Reader::nns() — reader.rs:301-369 (Visitor::visit), 621-640 (should_linear_scan), 642-665    */
/* (nns_by_vec), 667-711 (brute_force_search), 722-800 (hnsw_search), 809-896 (nns_by_item)     */
/* ------------------------------------------------------------------ */
int orc_search_ex(int32_t metric, int32_t order, uint32_t dim, const orc_items *items,
                  uint64_t n_records, const uint32_t *rec_item, const uint8_t *rec_layer,
                  const uint64_t *offsets, const uint32_t *nbrs, const uint32_t *entry_points,
                  uint32_t n_entry_points, uint32_t max_level, uint64_t n_queries, const void *qvecs,
                  size_t qstride, const void *qhdrs, uint32_t k, uint32_t ef_search, int32_t threads,
                  const orc_query_opts *qo, uint32_t *out_ids, float *out_dists,
                  uint32_t *out_counts) {
  uint32_t n = (uint32_t)items->n;
  Dist dist{metric, order, dim, vec_bytes(metric, dim)};
  size_t hsz = items->header_size;
  /* id -> slot (ids ascending) */
  auto slot_of = [&](uint32_t id) -> int64_t {
    const uint32_t *b = items->ids, *e = items->ids + n;
    const uint32_t *p = std::lower_bound(b, e, id);
    return (p != e && *p == id) ? (int64_t)(p - b) : -1;
  };
  /* per layer: slot -> record index */
  std::vector<std::vector<int64_t>> rec_of(max_level + 1, std::vector<int64_t>(n, -1));
  for (uint64_t r = 0; r < n_records; r++) {
    int64_t s = slot_of(rec_item[r]);
    if (s < 0 || rec_layer[r] > max_level) continue;
    rec_of[rec_layer[r]][s] = (int64_t)r;
  }
  std::vector<uint32_t> ep_slots;
  for (uint32_t i = 0; i < n_entry_points; i++) {
    int64_t s = slot_of(entry_points[i]);
    if (s < 0) return -3;
    ep_slots.push_back((uint32_t)s);
  }
  /* QueryBuilder::candidates (reader.rs:200-203) as a mask over the stored items; ids that are not
   * in the database never matter (brute_force_search skips them :686, the graph never reaches them) */
  const bool has_cand = qo && qo->has_candidates;
  const bool by_item = qo && qo->query_items;
  std::vector<uint8_t> cand_mask;
  std::vector<uint32_t> cand_slots; /* ascending */
  if (has_cand) {
    cand_mask.assign(n, 0);
    for (uint64_t i = 0; i < qo->n_candidates; i++) {
      int64_t s = slot_of(qo->candidates[i]);
      if (s >= 0 && !cand_mask[s]) cand_mask[s] = 1;
    }
    for (uint32_t s = 0; s < n; s++)
      if (cand_mask[s]) cand_slots.push_back(s);
  }
  /* should_linear_scan, reader.rs:621-640 */
  bool linear = false;
  if (has_cand && n > 0) {
    uint64_t cl = cand_slots.size();
    bool below_threshold = cl < (uint64_t)qo->linear_below;
    bool below_ratio = (float)cl / (float)n <= qo->linear_below_ratio;
    linear = below_threshold && below_ratio;
  }
  std::atomic<int> err{0};
  auto run_query = [&](uint64_t qi, Scratch &path) {
    out_counts[qi] = 0;
    const uint8_t *qv, *qh;
    uint32_t excl = 0xFFFFFFFFu; /* by_item: candidates.remove(item) :840 */
    /* :652-654 / :822-824 */
    bool never = n == 0 || (has_cand && cand_slots.empty());
    if (by_item) {
      if (never) {
        out_counts[qi] = 0xFFFFFFFFu; /* Ok(None) */
        return;
      }
      int64_t s = slot_of(qo->query_items[qi]);
      if (s < 0) { /* item_vector(..)? else return Ok(None) :826 */
        out_counts[qi] = 0xFFFFFFFFu;
        return;
      }
      /* :826-828 the stored vector re-encoded: same codec bytes, same header */
      qv = (const uint8_t *)items->vectors + (size_t)s * items->stride;
      qh = (const uint8_t *)items->headers + (size_t)s * hsz;
      excl = (uint32_t)s;
    } else {
      if (never) return; /* Done(Vec::new()) */
      qv = (const uint8_t *)qvecs + qi * qstride;
      qh = (const uint8_t *)qhdrs + qi * hsz;
    }
    uint64_t dummy = 0;
    auto qd = [&](uint32_t s) {
      dummy++;
      return dist(qv, qh, (const uint8_t *)items->vectors + (size_t)s * items->stride,
                  (const uint8_t *)items->headers + (size_t)s * hsz);
    };
    auto emit = [&](const std::vector<Link> &sorted) { /* drain_asc().take(count) */
      uint32_t cnt = (uint32_t)std::min<size_t>(k, sorted.size());
      for (uint32_t i = 0; i < cnt; i++) {
        out_ids[qi * k + i] = items->ids[sorted[i].id];
        out_dists[qi * k + i] = sorted[i].d;
      }
      out_counts[qi] = cnt;
    };
    if (linear) { /* brute_force_search, reader.rs:667-711 (by_item: the item itself stays in) */
      /* BinaryHeap<(OrderedFloat, id)> kept at <= count entries: the top (max by bits, then id) is
       * replaced only by a strictly smaller distance (:695), compared by bit pattern */
      std::vector<Link> heap; /* kept sorted ascending by link_key: back() is the top */
      for (uint32_t s : cand_slots) {
        float d = qd(s);
        if (heap.size() >= k) {
          if (!heap.empty() && f32_bits(heap.back().d) > f32_bits(d)) {
            heap.pop_back();
            Link l{d, s};
            heap.insert(std::lower_bound(heap.begin(), heap.end(), link_key(l),
                                         [](const Link &a, uint64_t kk) { return link_key(a) < kk; }),
                        l);
          }
        } else {
          Link l{d, s};
          heap.insert(std::lower_bound(heap.begin(), heap.end(), link_key(l),
                                       [](const Link &a, uint64_t kk) { return link_key(a) < kk; }),
                      l);
        }
      }
      emit(heap); /* into_sorted_vec :706 */
      return;
    }
    std::vector<Link> res;
    auto in_filter = [&](uint32_t s) { return (!has_cand || cand_mask[s]) && s != excl; };
    /* Visitor::visit, reader.rs:301-369 (no cancel) */
    auto visit = [&](const std::vector<uint32_t> &eps, uint32_t level, size_t ef, bool filtered) {
      std::priority_queue<Link, std::vector<Link>, CandLess> sq;
      res.clear();
      auto res_insert = [&](const Link &l) {
        uint64_t key = link_key(l);
        auto pos = std::lower_bound(res.begin(), res.end(), key,
                                    [](const Link &a, uint64_t kk) { return link_key(a) < kk; });
        res.insert(pos, l);
      };
      for (uint32_t ep : eps) {
        Link l{qd(ep), ep};
        sq.push(l);
        path.visit(ep);
        if (!filtered || in_filter(ep)) res_insert(l); /* :322-324 */
      }
      while (!sq.empty()) {
        float f = sq.top().d;
        float f_max = res.empty() ? 3.4028235e38f : res.back().d; /* unwrap_or(f32::MAX) :337 */
        if (f > f_max) break;
        uint32_t c = sq.top().id;
        sq.pop();
        int64_t r = rec_of[level][c];
        if (r < 0) { /* .expect("Links must exist") :343-344 */
          err.store(-4);
          return;
        }
        for (uint64_t j = offsets[r]; j < offsets[r + 1]; j++) {
          int64_t ps = slot_of(nbrs[j]);
          if (ps < 0) {
            err.store(-5);
            return;
          }
          uint32_t p = (uint32_t)ps;
          if (!path.visit(p)) continue;
          float d = qd(p);
          if (res.size() < ef || d < f_max) {
            Link l{d, p};
            sq.push(l);
            if (filtered && !in_filter(p)) continue; /* :356-360 */
            if (res.size() == ef) {
              res_insert(l);
              res.pop_back();
            } else {
              res_insert(l);
            }
          }
        }
      }
    };
    path.begin(n);
    std::vector<uint32_t> eps;
    size_t ef = std::max<size_t>(ef_search, k); /* :746 / :837 */
    const bool filtered = has_cand || by_item;
    if (by_item) {
      eps.assign(1, excl); /* Visitor::new(vec![item], 0, ef, Some(&candidates)) :842 */
    } else {
      eps = ep_slots;
      /* reader.rs:732-741: the path bitmap is shared across the greedy layers */
      for (uint32_t l = max_level; l >= 1; l--) {
        visit(eps, l, 1, false);
        if (err.load()) return;
        eps.assign(1, res.front().id);
      }
      path.begin(n); /* path.clear() :743 */
    }
    visit(eps, 0, ef, filtered);
    if (err.load()) return;
    std::vector<Link> neighbours = res;
    if (neighbours.size() < k) { /* exhaustive fallback :771-795 / :864-890 */
      for (uint32_t s = 0; s < n; s++) {
        if (path.seen(s)) continue;
        size_t ef2;
        if (by_item) ef2 = k - neighbours.size(); /* :878 */
        else ef2 = ef_search > neighbours.size() ? ef_search - neighbours.size() : 0; /* :783 */
        visit(std::vector<uint32_t>{s}, 0, ef2, filtered);
        if (err.load()) return;
        for (const Link &l : res) neighbours.push_back(l);
        if (neighbours.size() >= (by_item ? (size_t)k : (size_t)ef_search)) break;
      }
      std::sort(neighbours.begin(), neighbours.end(),
                [](const Link &a, const Link &b) { return link_key(a) < link_key(b); });
    }
    emit(neighbours); /* drain_asc().take(k) :797 */
  };
  int nt = std::max(1, threads);
  std::atomic<uint64_t> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&]() {
      Scratch path;
      for (;;) {
        uint64_t qi = next.fetch_add(1);
        if (qi >= n_queries || err.load()) break;
        run_query(qi, path);
      }
    });
  for (auto &t : th) t.join();
  return err.load();
}

int orc_search(int32_t metric, int32_t order, uint32_t dim, const orc_items *items,
               uint64_t n_records, const uint32_t *rec_item, const uint8_t *rec_layer,
               const uint64_t *offsets, const uint32_t *nbrs, const uint32_t *entry_points,
               uint32_t n_entry_points, uint32_t max_level, uint64_t n_queries, const void *qvecs,
               size_t qstride, const void *qhdrs, uint32_t k, uint32_t ef_search, int32_t threads,
               uint32_t *out_ids, float *out_dists, uint32_t *out_counts) {
  return orc_search_ex(metric, order, dim, items, n_records, rec_item, rec_layer, offsets, nbrs,
                       entry_points, n_entry_points, max_level, n_queries, qvecs, qstride, qhdrs, k,
                       ef_search, threads, nullptr, out_ids, out_dists, out_counts);
}

/* ------------------------------------------------------------------ */
/* on-disk records                                                     */
/* ------------------------------------------------------------------ */
/* key.rs:57-66: index u16 BE | mode u8 | item u32 BE | layer u8 */
void orc_encode_key(uint16_t index, uint8_t mode, uint32_t item, uint8_t layer, uint8_t out[8]) {
  out[0] = (uint8_t)(index >> 8);
  out[1] = (uint8_t)index;
  out[2] = mode;
  out[3] = (uint8_t)(item >> 24);
  out[4] = (uint8_t)(item >> 16);
  out[5] = (uint8_t)(item >> 8);
  out[6] = (uint8_t)item;
  out[7] = layer;
}

/* [3P] roaring 0.10.9 RoaringBitmap::serialize_into — portable RoaringFormatSpec,
 * SERIAL_COOKIE_NO_RUNCONTAINER (12346), array containers for cardinality <= 4096 else 8 KiB
 * bitmap containers, offset header always present. */
size_t orc_roaring_serialize(const uint32_t *ids, uint64_t n, uint8_t *out) {
  struct C {
    uint16_t key;
    uint64_t begin, end;
  };
  std::vector<C> cs;
  for (uint64_t i = 0; i < n;) {
    uint16_t key = (uint16_t)(ids[i] >> 16);
    uint64_t j = i;
    while (j < n && (uint16_t)(ids[j] >> 16) == key) j++;
    cs.push_back({key, i, j});
    i = j;
  }
  size_t size = 8 + 8 * cs.size();
  for (auto &c : cs) size += (c.end - c.begin) <= 4096 ? 2 * (c.end - c.begin) : 8192;
  if (!out) return size;
  auto w16 = [&](size_t &p, uint16_t v) {
    out[p++] = (uint8_t)v;
    out[p++] = (uint8_t)(v >> 8);
  };
  auto w32 = [&](size_t &p, uint32_t v) {
    for (int k = 0; k < 4; k++) out[p++] = (uint8_t)(v >> (8 * k));
  };
  size_t p = 0;
  w32(p, 12346u);
  w32(p, (uint32_t)cs.size());
  for (auto &c : cs) {
    w16(p, c.key);
    w16(p, (uint16_t)(c.end - c.begin - 1));
  }
  uint32_t off = (uint32_t)(8 + 8 * cs.size());
  for (auto &c : cs) {
    w32(p, off);
    off += (c.end - c.begin) <= 4096 ? (uint32_t)(2 * (c.end - c.begin)) : 8192u;
  }
  for (auto &c : cs) {
    if (c.end - c.begin <= 4096) {
      for (uint64_t i = c.begin; i < c.end; i++) w16(p, (uint16_t)ids[i]);
    } else {
      std::memset(out + p, 0, 8192);
      for (uint64_t i = c.begin; i < c.end; i++) {
        uint16_t lo = (uint16_t)ids[i];
        out[p + (lo >> 3)] |= (uint8_t)(1u << (lo & 7)); /* u64 LE words == little-endian bit array */
      }
      p += 8192;
    }
  }
  return size;
}

static const char *metric_name(int metric) {
  switch (metric) { /* cosine.rs:32-34 etc. */
    case ORC_COSINE: return "cosine";
    case ORC_EUCLIDEAN: return "euclidean";
    case ORC_MANHATTAN: return "manhattan";
    case ORC_HAMMING: return "hamming";
    case ORC_BQ_COSINE: return "binary quantized cosine";
    case ORC_BQ_EUCLIDEAN: return "binary quantized euclidean";
    case ORC_BQ_MANHATTAN: return "binary quantized manhattan";
  }
  return "";
}

size_t orc_encode_kv(const orc_graph *g, const orc_opts *opts, const orc_items *items,
                     uint16_t index, int with_items, uint8_t *out, size_t cap) {
  std::vector<uint8_t> buf;
  auto put = [&](const uint8_t key[8], const std::vector<uint8_t> &val) {
    uint32_t kl = 8, vl = (uint32_t)val.size();
    for (int k = 0; k < 4; k++) buf.push_back((uint8_t)(kl >> (8 * k)));
    buf.insert(buf.end(), key, key + 8);
    for (int k = 0; k < 4; k++) buf.push_back((uint8_t)(vl >> (8 * k)));
    buf.insert(buf.end(), val.begin(), val.end());
  };
  uint8_t key[8];
  std::vector<uint8_t> val;
  /* Metadata (metadata.rs:28-48): name \0 | dims u32 BE | roaring size u32 BE | roaring(items) |
   * entry points (native-endian u32 each, node.rs ItemIds::raw_bytes) | max_level u8 */
  {
    val.clear();
    const char *nm = metric_name(opts->metric);
    val.insert(val.end(), nm, nm + std::strlen(nm));
    val.push_back(0);
    uint32_t dims = opts->dim;
    for (int k = 3; k >= 0; k--) val.push_back((uint8_t)(dims >> (8 * k)));
    size_t rs = orc_roaring_serialize(items->ids, items->n, nullptr);
    for (int k = 3; k >= 0; k--) val.push_back((uint8_t)((uint32_t)rs >> (8 * k)));
    size_t at = val.size();
    val.resize(at + rs);
    orc_roaring_serialize(items->ids, items->n, val.data() + at);
    for (uint32_t ep : g->entry_points) {
      uint8_t b[4];
      std::memcpy(b, &ep, 4);
      val.insert(val.end(), b, b + 4);
    }
    val.push_back((uint8_t)g->max_level);
    orc_encode_key(index, 0, 0, 0, key);
    put(key, val);
  }
  /* Version (version.rs:36-48): 3 x u32 BE = 0.1.3 (Cargo.toml version) */
  {
    val.clear();
    uint32_t v[3] = {0, 1, 3};
    for (uint32_t x : v)
      for (int k = 3; k >= 0; k--) val.push_back((uint8_t)(x >> (8 * k)));
    orc_encode_key(index, 0, 1, 0, key);
    put(key, val);
  }
  /* Links (node.rs:141-144): 0x01 | roaring */
  for (uint64_t r = 0; r < g->rec_item.size(); r++) {
    uint64_t b = g->offsets[r], e = g->offsets[r + 1];
    size_t rs = orc_roaring_serialize(g->nbrs.data() + b, e - b, nullptr);
    val.assign(1 + rs, 0);
    val[0] = 1;
    orc_roaring_serialize(g->nbrs.data() + b, e - b, val.data() + 1);
    orc_encode_key(index, 2, g->rec_item[r], g->rec_layer[r], key);
    put(key, val);
  }
  /* Items (node.rs:136-140): 0x00 | header | vector bytes */
  if (with_items) {
    size_t vb = vec_bytes(opts->metric, opts->dim), hb = items->header_size;
    for (uint64_t s = 0; s < items->n; s++) {
      val.assign(1 + hb + vb, 0);
      std::memcpy(val.data() + 1, (const uint8_t *)items->headers + s * hb, hb);
      std::memcpy(val.data() + 1 + hb, (const uint8_t *)items->vectors + s * items->stride, vb);
      orc_encode_key(index, 3, items->ids[s], 0, key);
      put(key, val);
    }
  }
  if (out && cap >= buf.size()) std::memcpy(out, buf.data(), buf.size());
  return buf.size();
}

} /* extern "C" */
