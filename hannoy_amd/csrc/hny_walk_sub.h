// hny_walk_sub.h — walk_layer (/root/reference/src/hnsw.rs:460-518) + greedy descent (:303-307) for
// SHORT rows (<= 512 B: 128-d f32, 1024-bit codes, every BQ metric up to 4096 dims): FOUR queries
// per wave, each on a 16-lane sub-wave.  Included by hny_kernels.hip inside its anonymous namespace.
//
// Why: with one wave per query a 136-B / 516-B evaluation costs ~40 VALU + ~40 SALU instructions of
// beam / visited / frontier bookkeeping for ~4 instructions of arithmetic (profiles/r01 SQ counters:
// k_walk 0.14 / 0.37 of HBM peak on C5 / C4, bound by instruction issue).  Here one instruction
// stream serves four queries: every "wave-uniform" value of k_walk becomes a value that is uniform
// inside a sub-wave and lives in a VGPR; ballots are taken wave-wide and each sub-wave reads its own
// 16-bit slice.  The four queries advance in lockstep, one expansion per iteration of ONE flattened
// loop; a sub-wave that finishes its query pulls the next member from the work queue by itself, so
// no sub-wave waits for the longest query of a wave.
//
// State per query (sub-wave):
//   beam `res`     RC x 16 sorted keys in registers (entry e -> chunk e / 16, lane e % 16), mirrored
//                  in LDS by every merge (indexed reads: res.max, the popped entry) — same key
//                  format and the same tie pool semantics as k_walk (see Beam there);
//   frontier       the <= 32 neighbour ids of the popped node, two per lane (list positions 2t, 2t+1);
//   visited set    a hash table of the query's own in HBM that stays L2 / Infinity-Cache resident
//                  (16 KB, reused by every query the sub-wave processes).  k_walk's per-wave N-bit
//                  sets are what bounds short rows: on C5 they cost 604 GB of write-backs + as many
//                  line fetches for 847 GB of algorithmic row bytes (profiles/r02 PMC), all of it
//                  random 64-B read-modify-writes.  Layout: buckets of 16 ids = one 64-B line; a
//                  lookup reads the whole home bucket with L1-bypassing loads (1 round trip for all
//                  128 ids of a step), only ids that look new go on to ONE compare-and-swap into an
//                  empty slot of that bucket (the slot after a per-id home position, so that two new
//                  ids of a step rarely race for one slot).  A plain open-addressing table was built
//                  first and was 4x slower than the bitset: its probe loop runs until the LAST of
//                  128 lanes is done, each probe a dependent atomic round trip (29 us per step);
//   query row      1 or 2 float4 per lane.
// Distances: bit-identical to k_walk's wave order.  A row of LPRO = 8 / 16 / 32 sixteen-byte units
// is summed by the same per-unit fma chains; 16 lanes hold two units each when LPRO = 32 and add
// them first, which IS the xor-butterfly's off = 16 step (p[t] + p[t ^ 16]); the remaining steps
// run inside the sub-wave with the folded butterflies of dist_rows.
//
// Anything that does not fit the small per-query capacities (visited table 3/4 full, tie pool > 32,
// res beyond 16 * RC entries, iteration cap) is NOT handled here: the member is appended to a retry
// list and the ordinary one-wave-per-query k_walk, launched right behind this kernel on that list,
// computes it — the result of every member is exact whichever kernel produced it, and the
// evaluation counter only takes completed queries.

#define HNY_SUB_POOL 32

template <int LPRO>
struct SubShape {
  static constexpr int LG = LPRO < 16 ? LPRO : 16; // lanes that share one row
  static constexpr int RPG = 16 / LG;              // rows per sub-wave and load instruction
  static constexpr int NQ = LPRO / LG;             // 16-byte units per lane and row
  static constexpr int U = NQ == 2 ? 4 : 8;        // load instructions in flight
};

__host__ __device__ inline size_t walk_sub_lds_bytes(int rc) {
  // per query: st[16 rc] u64 | pool[32] u64 | nb_ids[32] u32 + nb_d[32] f32 (also the accepted-key
  // list, 32 u64) | eps[32] u32 | dd[32] u32
  return 4 * ((size_t)16 * rc * 8 + HNY_SUB_POOL * 8 + 32 * 8 + 32 * 4 + 32 * 4);
}

typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
// one 64-B bucket of the visited table, fresh from L2: `sc1` loads bypass this CU's L1, which is
// never refreshed by the L2 atomics that fill the table (MI355X_MICROARCH.md, inter-workgroup
// visibility).  The compiler does not count asm loads: vt_wait() must precede the first use.
__device__ __forceinline__ void vt_load_bucket(const u32 *p, u32x4_t &a, u32x4_t &b, u32x4_t &c, u32x4_t &d) {
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
               "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
               "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
               "global_load_dwordx4 %3, %4, off offset:48 sc1"
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
               : "v"(p)
               : "memory");
}
__device__ __forceinline__ void vt_wait(u32x4_t &a, u32x4_t &b, u32x4_t &c, u32x4_t &d, u32x4_t &e, u32x4_t &f,
                                        u32x4_t &g, u32x4_t &h) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
}
// is `id` in the bucket / which of its 16 slots are empty (bit k = slot k)
__device__ __forceinline__ bool vt_scan(const u32x4_t &a, const u32x4_t &b, const u32x4_t &c, const u32x4_t &d, u32 id,
                                        u32 &empty) {
  bool hit = false;
  u32 em = 0u;
#define HNY_VT_ONE(v, k)                        \
  hit = hit || (v) == id;                       \
  em |= ((v) == HNY_SENT ? 1u : 0u) << (k);
  HNY_VT_ONE(a.x, 0) HNY_VT_ONE(a.y, 1) HNY_VT_ONE(a.z, 2) HNY_VT_ONE(a.w, 3)
  HNY_VT_ONE(b.x, 4) HNY_VT_ONE(b.y, 5) HNY_VT_ONE(b.z, 6) HNY_VT_ONE(b.w, 7)
  HNY_VT_ONE(c.x, 8) HNY_VT_ONE(c.y, 9) HNY_VT_ONE(c.z, 10) HNY_VT_ONE(c.w, 11)
  HNY_VT_ONE(d.x, 12) HNY_VT_ONE(d.y, 13) HNY_VT_ONE(d.z, 14) HNY_VT_ONE(d.w, 15)
#undef HNY_VT_ONE
  empty = em;
  return hit;
}

// bits of my sub-wave in a wave-wide ballot
__device__ __forceinline__ u32 sub_slice(u64 m, int sgb) { return (u32)(m >> sgb) & 0xFFFFu; }
// max over the four sub-waves of a sub-wave-uniform value (wave-uniform result, in an SGPR)
__device__ __forceinline__ int sub_wave_max(int v) {
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
            c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

template <int LPRO>
__device__ __forceinline__ float sub_partial_f32(int mclass, const float4 (&q)[SubShape<LPRO>::NQ],
                                                 const float4 (&r)[SubShape<LPRO>::NQ]) {
  constexpr int NQ = SubShape<LPRO>::NQ;
  float4 q0[1] = {q[0]}, r0[1] = {r[0]};
  float p = partial_f32<1>(mclass, q0, r0);
  if constexpr (NQ == 2) { // unit t + 16: the butterfly's off = 16 step, p[t] + p[t ^ 16]
    float4 q1[1] = {q[1]}, r1[1] = {r[1]};
    p = p + partial_f32<1>(mclass, q1, r1);
  }
  return p;
}
template <int LPRO>
__device__ __forceinline__ u32 sub_partial_bin(const float4 (&q)[SubShape<LPRO>::NQ],
                                               const float4 (&r)[SubShape<LPRO>::NQ]) {
  return partial_bin<SubShape<LPRO>::NQ>(q, r);
}

// distances from each sub-wave's query to its rows ids[0..n) -> out[0..n) (both in the sub-wave's
// LDS).  n: sub-wave-uniform; nmax: its maximum over the wave (wave-uniform loop bounds).
template <int LPRO>
__device__ __forceinline__ void dist_rows_sub(const GraphDev &g, const float4 (&q)[SubShape<LPRO>::NQ], float qn,
                                              const u32 *ids, int n, int nmax, float *out) {
  using S = SubShape<LPRO>;
  constexpr int LG = S::LG, RPG = S::RPG, NQ = S::NQ, U = S::U;
  const int t16 = threadIdx.x & 15, t = t16 % LG, sub = t16 / LG;
  const int j4 = fold4_row<LG>();
  for (int k0 = 0; k0 < nmax; k0 += RPG * U) {
    float4 r[U][NQ];
    float rn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      rn[u] = 0.f;
      if (k0 + u * RPG < nmax) { // wave-uniform
        int ri = k0 + u * RPG + sub;
        if (ri > n - 1) ri = n - 1;
        const u32 rid = ri >= 0 ? ids[ri] : 0u; // n == 0: any existing row, the result is not stored
        const unsigned char *p = g.rows + (size_t)rid * g.row_stride;
#pragma unroll
        for (int c = 0; c < NQ; c++) {
          const u32 f = (u32)(c * LG + t);
          r[u][c] = f < g.n16 ? *reinterpret_cast<const float4 *>(p + (size_t)f * 16)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (g.norms) rn[u] = g.norms[rid];
      } else {
#pragma unroll
        for (int c = 0; c < NQ; c++) r[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += 4) {
      if (k0 + u0 * RPG < nmax) { // wave-uniform
        float rnj = rn[u0];
#pragma unroll
        for (int jj = 1; jj < 4; jj++) rnj = (j4 == jj) ? rn[u0 + jj] : rnj;
        float d;
        if (g.mclass == MC_BIN) {
          const u32 pc = fold4<LG, u32>(sub_partial_bin<LPRO>(q, r[u0]), sub_partial_bin<LPRO>(q, r[u0 + 1]),
                                        sub_partial_bin<LPRO>(q, r[u0 + 2]), sub_partial_bin<LPRO>(q, r[u0 + 3]));
          d = finalize_bin(g, pc, qn, rnj);
        } else {
          const float pa = fold4<LG, float>(sub_partial_f32<LPRO>(g.mclass, q, r[u0]),
                                            sub_partial_f32<LPRO>(g.mclass, q, r[u0 + 1]),
                                            sub_partial_f32<LPRO>(g.mclass, q, r[u0 + 2]),
                                            sub_partial_f32<LPRO>(g.mclass, q, r[u0 + 3]));
          d = finalize_f32(g, pa, qn, rnj);
        }
        const int ri = k0 + (u0 + j4) * RPG + sub;
        if ((t & (LG / 4 - 1)) == 0 && ri < n) out[ri] = d;
      }
    }
  }
}

enum { SUB_NEED = 0, SUB_INIT = 1, SUB_RUN = 2, SUB_FIN = 3, SUB_DONE = 4 };

#ifndef HNY_SUB_WPE
#define HNY_SUB_WPE 4
#endif

template <int LPRO, int RC, int SP>
__global__ __launch_bounds__(64, HNY_SUB_WPE) void k_walk_sub(GraphDev g_in, WalkArgs a_in) {
  static_assert(SP != 0, "specialised kernels only");
  using S = SubShape<LPRO>;
  constexpr int NQ = S::NQ, LG = S::LG;
  constexpr int RCAP = 16 * RC;
  GraphDev g = g_in;
  WalkArgs a = a_in;
  specialize<SP>(g);
  extern __shared__ __align__(16) unsigned char smem[];
  const int ln = threadIdx.x, sg = ln >> 4, t16 = ln & 15, sgb = sg << 4;
  const u32 ltm = (1u << t16) - 1u; // lanes below me inside the sub-wave
  unsigned char *base = smem + (size_t)sg * (walk_sub_lds_bytes(RC) / 4);
  u64 *st = reinterpret_cast<u64 *>(base);               // [RCAP] mirror of the beam / merge scratch
  u64 *pool = st + RCAP;                                 // [HNY_SUB_POOL]
  u32 *nb_ids = reinterpret_cast<u32 *>(pool + HNY_SUB_POOL); // [32]
  float *nb_d = reinterpret_cast<float *>(nb_ids + 32);  // [32]
  u64 *accl = reinterpret_cast<u64 *>(nb_ids);           // [32] accepted keys (nb_* are consumed by then)
  u32 *eps = reinterpret_cast<u32 *>(nb_d + 32);         // [32]
  u32 *dd = eps + 32;                                    // [32] first-occurrence scratch
  u32 *vtab = a.vtab + ((size_t)blockIdx.x * 4 + (size_t)sg) * a.vtab_slots;

  // ---- per-query state: uniform inside a sub-wave, different between sub-waves
  int phase = SUB_NEED;
  u32 m = 0, layer = 0, dmax = 0, tie_bits = 0, vcount = 0, vslots = a.vtab_slots, iter = 0;
  int ef = 1, res_len = 0, pool_len = 0, n_weird = 0, n_eps = 0;
  bool dropped = false, bad = false;
  u64 r[RC];
#pragma unroll
  for (int c = 0; c < RC; c++) r[c] = 0ull;
  float4 q[NQ];
#pragma unroll
  for (int c = 0; c < NQ; c++) q[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  float qn = 0.f;
  u64 lkey = 0, qevals = 0, evals = 0;
  u32 n_done = 0, n_gave_up = 0, n_bad_vis = 0, n_bad_pool = 0;
  nb_ids[t16] = 0u;
  nb_ids[t16 + 16] = 0u;
  // the table starts empty (filled once per builder by the host): every walk leaves it empty again
  WSYNC();

  for (;;) {
    // ---- a walk_layer call ended (or the query was given up)
    if (phase == SUB_FIN) {
      const bool last = (layer == a.layer);
      if (bad) {
        if (t16 == 0) a.retry[atomicAdd(a.n_retry, 1u)] = (u64)m;
        n_gave_up++;
        phase = SUB_NEED;
      } else if (last) { // res.into_vec(): ascending, robust_prune sorts it anyway (:573)
#pragma unroll
        for (int c = 0; c < RC; c++)
          if (t16 + 16 * c < res_len)
            a.cand[(size_t)m * a.rcap + 16 * c + t16] = (r[c] & 0xFFFFFFFF00000000ull) | ((r[c] >> 1) & 0x7FFFFFFFull);
        if (t16 == 0) a.cand_n[m] = (u32)res_len;
        evals += qevals;
        n_done++;
        phase = SUB_NEED;
      } else { // :305-306 eps = [closest]
        const u32 closest = (u32)(st[0] >> 1) & 0x7FFFFFFFu;
        WSYNC();
        if (t16 == 0) eps[0] = closest;
        n_eps = 1;
        lkey = (lkey << 16) | (u64)((u32)g.upper_idx[closest] & 0xFFFFu);
        layer--;
        if (layer == a.layer && a.descend_only) {
          if (t16 == 0) {
            a.eps_out[m] = closest;
            a.key_out[m - a.key_base] = lkey & 0xFFFFFFFFFFFFull;
          }
          evals += qevals;
          n_done++;
          phase = SUB_NEED;
        } else {
          ef = layer == a.layer ? (int)a.ef : 1;
          phase = SUB_INIT;
        }
      }
      // walk_layer owns a fresh visited set (hnsw.rs:471): leave the table (the window the walk
      // used) empty
      for (u32 i = (u32)t16 * 4u; i < vslots; i += 64u)
        *reinterpret_cast<uint4 *>(vtab + i) = make_uint4(HNY_SENT, HNY_SENT, HNY_SENT, HNY_SENT);
      vcount = 0;
      bad = false;
    }
    // ---- next member from the work queue
    if (phase == SUB_NEED) {
      u32 qi = 0;
      if (t16 == 0) qi = a.lo + atomicAdd(a.queue, 1u);
      qi = (u32)__shfl((int)qi, sgb, 64);
      if (qi >= a.hi) {
        phase = SUB_DONE;
      } else {
        m = a.perm ? (u32)a.perm[qi - a.lo] : qi;
        const u32 qslot = a.q_slots[m];
        const unsigned char *qrow = g.rows + (size_t)qslot * g.row_stride;
        qn = g.norms ? g.norms[qslot] : 0.f;
#pragma unroll
        for (int c = 0; c < NQ; c++) {
          const u32 f = (u32)(c * LG + (t16 % LG));
          q[c] = f < g.n16 ? *reinterpret_cast<const float4 *>(qrow + (size_t)f * 16)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        WSYNC();
        if (a.first) { // :298 eps = all entry points (the host sends <= 32 of them here)
          n_eps = (int)a.n_entry_points;
          if (t16 < n_eps) eps[t16] = a.entry_points[t16];
          if (t16 + 16 < n_eps) eps[t16 + 16] = a.entry_points[t16 + 16];
          layer = g.max_level;
        } else if (a.eps_in) { // resume after a descend_only launch
          n_eps = 1;
          if (t16 == 0) eps[0] = a.eps_in[m];
          layer = a.layer;
        } else { // :316-321 eps = what was selected on the layer above
          const u64 *sl = a.sel + (size_t)m * a.sel_stride + (size_t)(a.batch_level - (a.layer + 1)) * (a.cap_sel + 1);
          n_eps = (int)sl[0];
          if (t16 < n_eps) eps[t16] = (u32)(sl[1 + t16] & 0xFFFFFFFFull);
          if (t16 + 16 < n_eps) eps[t16 + 16] = (u32)(sl[1 + t16 + 16] & 0xFFFFFFFFull);
          layer = a.layer;
        }
        lkey = 0;
        qevals = 0;
        iter = 0;
        if (a.force_retry && m % a.force_retry == 0u) { // test hook: exercise the hand-over
          bad = true;
          phase = SUB_FIN;
        } else
        if (layer == a.layer && a.descend_only) { // a batch at max_level has no greedy layer: eps stay
          if (t16 == 0) {
            a.eps_out[m] = a.entry_points[0];
            a.key_out[m - a.key_base] = 0ull;
          }
          // stays SUB_NEED: the next iteration pulls another member
        } else {
          ef = layer == a.layer ? (int)a.ef : 1;
          phase = SUB_INIT;
        }
      }
    }
    if (__ballot(phase != SUB_DONE) == 0ull) break;
    if (phase == SUB_INIT) { // walk_layer starts (:468-472)
      res_len = 0;
      pool_len = 0;
      n_weird = 0;
      tie_bits = 0;
      dropped = false;
      // greedy layers visit a few dozen nodes: a 512-entry window of the table is cleared faster
      vslots = ef == 1 && a.vtab_slots > 512u ? 512u : a.vtab_slots;
    }
    WSYNC();

    const bool init = phase == SUB_INIT;
    bool run = phase == SUB_RUN;
    u32 id0 = HNY_SENT, id1 = HNY_SENT;
    float fmax = 0.f;

    // ---- candidates.peek()/pop(): smallest distance bits, larger id first among equals
    // (BinaryHeap<(Reverse<OrderedFloat>, ItemId)>, :469, :483-488)
    if (__ballot(run) != 0ull) {
      int first_un = -1, last = -1;
      bool un[RC];
#pragma unroll
      for (int c = RC - 1; c >= 0; c--) {
        un[c] = run && t16 + 16 * c < res_len && !(r[c] & 1ull);
        const u32 mk = sub_slice(__ballot(un[c]), sgb);
        if (mk) first_un = 16 * c + __ffs((int)mk) - 1; // the lowest chunk wins (descending loop)
      }
      const bool have_a = first_un >= 0;
      const u32 d0 = (u32)(st[have_a ? first_un : 0] >> 32);
      last = first_un;
#pragma unroll
      for (int c = 0; c < RC; c++) {
        const u32 tk = sub_slice(__ballot(have_a && un[c] && (u32)(r[c] >> 32) == d0), sgb);
        if (tk) last = 16 * c + 31 - __clz((int)tk); // the highest chunk wins
      }
      // pop-order key: distance bits ascending, then id DESCENDING
      const u64 ta = have_a ? (((u64)d0 << 32) | (u64)(~(u32)(st[last] & 0xFFFFFFFEull))) : ~0ull;
      u64 tp = ~0ull;
      int pi = -1;
      if (__ballot(run && pool_len > 0) != 0ull) { // tie pool: rare for f32, common for Hamming
        u64 k0 = ~0ull, k1 = ~0ull;
        if (run && t16 < pool_len) {
          const u64 k = pool[t16];
          k0 = (k & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k & 0xFFFFFFFFull));
        }
        if (run && t16 + 16 < pool_len) {
          const u64 k = pool[t16 + 16];
          k1 = (k & 0xFFFFFFFF00000000ull) | (u64)(~(u32)(k & 0xFFFFFFFFull));
        }
        tp = k0 < k1 ? k0 : k1; // keys are distinct (one slot, one key); ~0 = none
        pi = k0 < k1 ? t16 : t16 + 16;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
          const u64 ot = (u64)__shfl_xor((long long)tp, off, 64);
          const int oi = __shfl_xor(pi, off, 64);
          if (ot < tp) {
            tp = ot;
            pi = oi;
          }
        }
        if (tp == ~0ull) pi = -1;
      }
      const bool have_p = pi >= 0;
      const bool use_pool = have_p && (!have_a || tp < ta);
      const u32 fb = (u32)((use_pool ? tp : ta) >> 32);
      bool fin = !have_a && !have_p; // candidates exhausted (or only dropped entries: they break)
      // a dropped ordinary candidate precedes every weird one in pop order and breaks the walk
      fin = fin || (use_pool && weird_bits(fb) && dropped);
      fin = fin || (__uint_as_float(fb) > __uint_as_float(dmax)); // raw f32 compare, :485
      if (run && !fin && ++iter > 200000u) {
        bad = true;
        fin = true;
      }
      if (run && fin) {
        phase = SUB_FIN;
        run = false;
      }
      if (run) {
        u32 cslot;
        if (use_pool) {
          cslot = (~(u32)(tp & 0xFFFFFFFFull)) >> 1;
          const u64 lastk = pool[pool_len - 1];
          WSYNC();
          if (t16 == 0) pool[pi] = lastk;
          pool_len--;
          if (weird_bits(fb)) n_weird--;
          WSYNC();
        } else {
          cslot = (u32)(st[last] >> 1) & 0x7FFFFFFFu;
#pragma unroll
          for (int c = 0; c < RC; c++) r[c] |= (u64)(t16 + 16 * c == last);
        }
        fmax = __uint_as_float(dmax); // f_max captured once per pop (:484)
        // ---- neighbours of c (:491-495): list positions 2t and 2t + 1
        u32 cap;
        const u32 *nl = nbr_ids(g, layer, cslot, cap);
        if ((u32)(2 * t16) < cap) id0 = nl[2 * t16];
        if ((u32)(2 * t16 + 1) < cap) id1 = nl[2 * t16 + 1];
      }
    }
    if (init) { // :474-481 every entry point goes to candidates and res (no capacity check)
      if (2 * t16 < n_eps) id0 = eps[2 * t16];
      if (2 * t16 + 1 < n_eps) id1 = eps[2 * t16 + 1];
    }

    // ---- visited.insert (:493) on the query's own table: read the home buckets of both ids (one
    // round trip), then one compare-and-swap for every id that is not there yet.  Two lanes with the
    // same id: both aim at the same slot (same bucket, same home position), exactly one wins.
    const bool valid0 = id0 != HNY_SENT, valid1 = id1 != HNY_SENT;
    bool isnew0 = false, isnew1 = false;
    {
      const u32 bm = (vslots >> 4) - 1u; // buckets - 1
      const u32 h0 = id0 * 0x9E3779B1u, h1 = id1 * 0x9E3779B1u;
      u32 k0 = (h0 >> 12) & bm, k1 = (h1 >> 12) & bm; // bucket
      const u32 p0 = h0 >> 28, p1 = h1 >> 28;          // home position inside the bucket
      bool pend0 = valid0, pend1 = valid1;
      while (__ballot(pend0 || pend1) != 0ull) {
        u32x4_t a0, a1, a2, a3, b0, b1, b2, b3;
        if (pend0) vt_load_bucket(vtab + (size_t)k0 * 16, a0, a1, a2, a3);
        if (pend1) vt_load_bucket(vtab + (size_t)k1 * 16, b0, b1, b2, b3);
        vt_wait(a0, a1, a2, a3, b0, b1, b2, b3);
        u32 e0 = 0u, e1 = 0u;
        const bool hit0 = pend0 && vt_scan(a0, a1, a2, a3, id0, e0);
        const bool hit1 = pend1 && vt_scan(b0, b1, b2, b3, id1, e1);
        if (hit0) pend0 = false;
        if (hit1) pend1 = false;
        u32 c0 = HNY_SENT - 1u, c1 = HNY_SENT - 1u; // "not tried"
        if (pend0 && e0) { // first empty slot at or after the home position, cyclically
          const u32 rot = ((e0 >> p0) | (e0 << (16u - p0))) & 0xFFFFu;
          c0 = atomicCAS(&vtab[(size_t)k0 * 16 + ((p0 + (u32)__ffs((int)rot) - 1u) & 15u)], HNY_SENT, id0);
        }
        if (pend1 && e1) {
          const u32 rot = ((e1 >> p1) | (e1 << (16u - p1))) & 0xFFFFu;
          c1 = atomicCAS(&vtab[(size_t)k1 * 16 + ((p1 + (u32)__ffs((int)rot) - 1u) & 15u)], HNY_SENT, id1);
        }
        if (pend0) {
          if (!e0) k0 = (k0 + 1u) & bm;   // bucket full: the id lives (or goes) in the next one
          else if (c0 == HNY_SENT) isnew0 = true, pend0 = false;
          else if (c0 == id0) pend0 = false; // the same id in another lane of this step won
          // else: another id took the slot meanwhile: look at the bucket again
        }
        if (pend1) {
          if (!e1) k1 = (k1 + 1u) & bm;
          else if (c1 == HNY_SENT) isnew1 = true, pend1 = false;
          else if (c1 == id1) pend1 = false;
        }
      }
    }
    u32 s0 = sub_slice(__ballot(isnew0), sgb), s1 = sub_slice(__ballot(isnew1), sgb);
    vcount += (u32)(__popc(s0) + __popc(s1));
    if (vcount > vslots / 4u * 3u - 32u) { // the table fills up: the big-visited-set kernel takes over
      if (!bad) n_bad_vis++;
      bad = true;
      phase = SUB_FIN;
      run = false;
      isnew0 = isnew1 = false;
      s0 = s1 = 0u;
    }
    if (init) { // every entry point is scored, visited or not
      isnew0 = valid0;
      isnew1 = valid1;
    } else if (__ballot(run && res_len < ef && (s0 | s1) != 0u) != 0ull) {
      // duplicates inside one list (add_link never dedups, hnsw.rs:521): while res is not full the
      // order of acceptance matters, so the FIRST occurrence must be the one that counts
      const bool ddo = run && res_len < ef;
      dd[2 * t16] = id0;
      dd[2 * t16 + 1] = id1;
      WSYNC();
      int f0 = 2 * t16, f1 = 2 * t16 + 1;
      bool an0 = isnew0, an1 = isnew1;
      for (int j = 0; j < 32; j++) {
        const u32 oj = dd[j];
        const bool nj = (((j & 1) ? s1 : s0) >> (j >> 1)) & 1u;
        if (valid0 && oj == id0) {
          if (j < f0) f0 = j;
          an0 = an0 || nj;
        }
        if (valid1 && oj == id1) {
          if (j < f1) f1 = j;
          an1 = an1 || nj;
        }
      }
      if (ddo) {
        isnew0 = valid0 && an0 && f0 == 2 * t16;
        isnew1 = valid1 && an1 && f1 == 2 * t16 + 1;
      }
      WSYNC();
    }
    if (init || !run) {
      if (!init) isnew0 = isnew1 = false;
    }
    s0 = sub_slice(__ballot(isnew0), sgb);
    s1 = sub_slice(__ballot(isnew1), sgb);
    const int n_new = __popc(s0) + __popc(s1);
    const int nmax = sub_wave_max(n_new);
    if (nmax > 0) {
      // compaction in list order: position 2t precedes 2t + 1 precedes 2(t + 1)
      const int rank0 = __popc(s0 & ltm) + __popc(s1 & ltm);
      const int rank1 = rank0 + (isnew0 ? 1 : 0);
      if (isnew0) nb_ids[rank0] = id0;
      if (isnew1) nb_ids[rank1] = id1;
      WSYNC();
      dist_rows_sub<LPRO>(g, q, qn, nb_ids, n_new, nmax, nb_d); // :476, :503
      qevals += (u64)n_new;
      WSYNC();
      const float myd0 = t16 < n_new ? nb_d[t16] : 0.f, myd1 = t16 + 16 < n_new ? nb_d[t16 + 16] : 0.f;
      const u32 myid0 = t16 < n_new ? nb_ids[t16] : 0u, myid1 = t16 + 16 < n_new ? nb_ids[t16 + 16] : 0u;
      const int ef_eff = init ? 0x7FFFFFFF : ef;
      int room = ef_eff - res_len;
      if (room < 0) room = 0;
      // :505 `res.len() < ef || dist < f_max` — the first `room` new points are taken regardless
      const bool acc0 = t16 < n_new && (t16 < room || myd0 < fmax);
      const bool acc1 = t16 + 16 < n_new && (t16 + 16 < room || myd1 < fmax);
      const u64 key0 = ((u64)fbits(myd0) << 32) | ((u64)myid0 << 1);
      const u64 key1 = ((u64)fbits(myd1) << 32) | ((u64)myid1 << 1);
      const u32 a0 = sub_slice(__ballot(acc0), sgb), a1 = sub_slice(__ballot(acc1), sgb);
      const int A = __popc(a0) + __popc(a1);
      const int Amax = sub_wave_max(A);
      WSYNC();
      if (Amax > 0) {
        // ---- res.push / push_pop_max for all accepted points at once (see beam_merge_rb): res ends
        // up as the new_len smallest keys of (res U accepted)
        const int pos0 = __popc(a0 & ltm), pos1 = __popc(a0) + __popc(a1 & ltm);
        if (acc0) accl[pos0] = key0;
        if (acc1) accl[pos1] = key1;
        WSYNC();
        u64 kc[RC];
        bool inr[RC];
        int sh[RC]; // accepted keys below my res entry of chunk c
#pragma unroll
        for (int c = 0; c < RC; c++) {
          kc[c] = r[c] & ~1ull;
          inr[c] = t16 + 16 * c < res_len;
          sh[c] = 0;
        }
        int mypos0 = 0, mypos1 = 0; // final index of my keys
        for (int i = 0; i < Amax; i++) {
          const bool oni = i < A;
          const u64 ki = oni ? accl[i] : 0ull;
          int below = 0;
#pragma unroll
          for (int c = 0; c < RC; c++) {
            const bool l = oni && inr[c] && kc[c] < ki; // keys are distinct (one slot, one key)
            below += __popc(sub_slice(__ballot(l), sgb));
            sh[c] += (oni && inr[c] && !l) ? 1 : 0;
          }
          mypos0 += (oni && acc0 && ki < key0) ? 1 : 0;
          mypos1 += (oni && acc1 && ki < key1) ? 1 : 0;
          if (acc0 && pos0 == i) mypos0 += below;
          if (acc1 && pos1 == i) mypos1 += below;
        }
        const int total = res_len + A;
        // push while len != ef, push_pop_max at len == ef; a res that starts above ef (entry points are
        // pushed without a capacity check, :474-481) only grows
        int new_len = res_len > ef_eff ? total : (total < ef_eff ? total : ef_eff);
        if (new_len > RCAP) { // does not fit the register beam: the one-wave kernel redoes the query
          bad = true;
          phase = SUB_FIN;
          new_len = RCAP;
        }
        int idx[RC];
        u64 old[RC];
#pragma unroll
        for (int c = 0; c < RC; c++) {
          idx[c] = t16 + 16 * c + sh[c];
          old[c] = r[c];
          if (inr[c] && idx[c] < new_len) st[idx[c]] = r[c];
        }
        if (acc0 && mypos0 < new_len) st[mypos0] = key0;
        if (acc1 && mypos1 < new_len) st[mypos1] = key1;
        WSYNC();
#pragma unroll
        for (int c = 0; c < RC; c++) r[c] = t16 + 16 * c < new_len ? st[t16 + 16 * c] : 0ull;
        res_len = new_len;
        const u32 nd = new_len > 0 ? (u32)(st[new_len - 1] >> 32) : dmax;
        dmax = nd;
        if (__ballot(total != new_len && !bad) != 0ull) { // something fell out of res
          const bool fo = total != new_len && !bad;
          // res.max moved below the ordinary pool entries: they can never be popped before the break
          if (fo && pool_len - n_weird > 0 && tie_bits != nd) {
            dropped = true;
            if (n_weird == 0) {
              pool_len = 0;
            } else {
              const u64 p0 = t16 < pool_len ? pool[t16] : 0ull, p1 = t16 + 16 < pool_len ? pool[t16 + 16] : 0ull;
              const bool w0 = t16 < pool_len && weird_bits((u32)(p0 >> 32));
              const bool w1 = t16 + 16 < pool_len && weird_bits((u32)(p1 >> 32));
              const u32 m0 = sub_slice(__ballot(w0), sgb), m1 = sub_slice(__ballot(w1), sgb);
              WSYNC();
              if (w0) pool[__popc(m0 & ltm)] = p0;
              if (w1) pool[__popc(m0) + __popc(m1 & ltm)] = p1;
              pool_len = __popc(m0) + __popc(m1);
              WSYNC();
            }
          }
          // classify what fell out: an unexpanded entry stays poppable only while its distance ties
          // the new res.max (tie pool) or is "weird"; otherwise it can never be popped (`dropped`)
          bool any_drop = false, any_keep = false;
#pragma unroll
          for (int rd = 0; rd < RC + 2; rd++) {
            const bool ev = rd < RC ? (inr[rd < RC ? rd : 0] && idx[rd < RC ? rd : 0] >= new_len)
                                    : (rd == RC ? (acc0 && mypos0 >= new_len) : (acc1 && mypos1 >= new_len));
            const u64 x = rd < RC ? old[rd < RC ? rd : 0] : (rd == RC ? key0 : key1);
            const u32 xb = (u32)(x >> 32);
            const bool unx = fo && ev && !(x & 1ull);
            const bool w = weird_bits(xb);
            any_drop = any_drop || (unx && !w && xb != nd);
            any_keep = any_keep || (unx && (w || xb == nd));
          }
          if (sub_slice(__ballot(any_drop), sgb)) dropped = true;
          if (__ballot(any_keep) != 0ull) {
#pragma unroll
            for (int rd = 0; rd < RC + 2; rd++) {
              const bool ev = rd < RC ? (inr[rd < RC ? rd : 0] && idx[rd < RC ? rd : 0] >= new_len)
                                      : (rd == RC ? (acc0 && mypos0 >= new_len) : (acc1 && mypos1 >= new_len));
              const u64 x = rd < RC ? old[rd < RC ? rd : 0] : (rd == RC ? key0 : key1);
              const u32 xb = (u32)(x >> 32);
              const bool w = weird_bits(xb);
              const bool keep = fo && ev && !(x & 1ull) && (w || xb == nd);
              const u32 pm = sub_slice(__ballot(keep), sgb);
              if (__ballot(keep) != 0ull) {
                if (sub_slice(__ballot(keep && !w), sgb)) tie_bits = nd;
                int proom = HNY_SUB_POOL - pool_len;
                if (proom < 0) proom = 0;
                const int rank = __popc(pm & ltm);
                const bool put = keep && rank < proom;
                if (put) pool[pool_len + rank] = x & ~1ull;
                const int np = __popc(pm), nput = np < proom ? np : proom;
                n_weird += __popc(sub_slice(__ballot(put && w), sgb));
                pool_len += nput;
                if (np > nput) { // the small tie pool is full: the one-wave kernel redoes the query
                  if (!bad) n_bad_pool++;
                  bad = true;
                  phase = SUB_FIN;
                }
                WSYNC();
              }
            }
          }
        }
      }
    }
    if (init && phase == SUB_INIT) phase = SUB_RUN;
  }
  if (t16 == 0) {
    if (evals) atomicAdd(&g.stats[ST_EVALS_WALK], evals);
    if (n_done) atomicAdd(&g.stats[ST_SUB_DONE], (u64)n_done);
    if (n_gave_up) atomicAdd(&g.stats[ST_SUB_RETRY], (u64)n_gave_up);
    if (n_bad_vis) atomicAdd(&g.stats[ST_SUB_RETRY_VIS], (u64)n_bad_vis);
    if (n_bad_pool) atomicAdd(&g.stats[ST_SUB_RETRY_POOL], (u64)n_bad_pool);
  }
}
