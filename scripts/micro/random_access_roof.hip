// random_access_roof.hip — what the memory system of one MI355X sustains for the short-row walk's access
// shapes, free of any algorithm: (a) random 128-B row gathers (8 lanes x 16 B per row, 8 rows per wave-wide
// load), (b) random returning atomicOr on per-wave bitsets (one lane = one word, 32 lanes per instruction),
// (c) a dependent chain of both, as one expansion of walk_layer issues them (list -> visited -> rows).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/random_access_roof.hip -o gpurun_out/random_access_roof
//   gpurun_out/random_access_roof [rows=5000000] [row_bytes=128] [waves=5120] [bits_words=156250]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 rng(u32 &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// (a) each wave: `iters` times, G wave-wide loads in flight, every 8-lane group reads one random row of
// UNITS x 16 B... (row_bytes / 16 / 8 chunks per lane)
template <int NQ, int G>
__global__ __launch_bounds__(64) void k_rows(const float4 *rows, u32 n_rows, u32 iters, u32 *sink) {
  const int ln = threadIdx.x, t = ln & 7, sub = ln >> 3;
  u32 s = (blockIdx.x * 9781u + 1u) * 2654435761u + (u32)sub * 40503u;  // one stream per 8-lane group
  u32 acc = 0;
  for (u32 it = 0; it < iters; it++) {
    float4 r[G][NQ];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const u32 row = rng(s) % n_rows;
#pragma unroll
      for (int c = 0; c < NQ; c++) r[g][c] = rows[(size_t)row * (NQ * 8) + c * 8 + t];
    }
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int c = 0; c < NQ; c++) acc += __float_as_uint(r[g][c].x) ^ __float_as_uint(r[g][c].w);
  }
  if (acc == 0x12345u) sink[0] = acc;
}

// (b) each wave owns bits[blockIdx][words]; 32 lanes issue one returning atomicOr each on a random word
template <bool ATOMIC>
__global__ __launch_bounds__(64) void k_bits(u32 *bits, u32 words, u32 iters, u32 *sink) {
  const int ln = threadIdx.x;
  u32 *mine = bits + (size_t)blockIdx.x * words;
  u32 s = (blockIdx.x * 64u + ln + 1u) * 2654435761u;
  u32 acc = 0;
  for (u32 it = 0; it < iters; it++) {
    const u32 w = rng(s) % words, b = 1u << (s >> 27);
    if (ln < 32) acc += ATOMIC ? atomicOr(&mine[w], b) : mine[w];
  }
  if (acc == 0x12345u) sink[0] = acc;
}

// (c) the chain of one expansion: a 128-B "list" line at a random node, then 32 atomicOr on the wave's bitset,
// then `new_rows` row gathers; every step waits for the one before (data dependence through the addresses)
template <int NQ>
__global__ __launch_bounds__(64) void k_chain(const float4 *rows, u32 n_rows, const u32 *lists, u32 *bits, u32 words,
                                              u32 iters, u32 new_rows, u32 *sink) {
  const int ln = threadIdx.x, t = ln & 7, sub = ln >> 3;
  u32 *mine = bits + (size_t)blockIdx.x * words;
  u32 node = (blockIdx.x * 7919u) % n_rows;
  u32 acc = 0;
  for (u32 it = 0; it < iters; it++) {
    const u32 id = lists[(size_t)node * 32 + (ln & 31)];           // list of `node`: 32 random ids
    u32 old = 0;
    if (ln < 32) old = atomicOr(&mine[(id >> 5) % words], 1u << (id & 31));
    const u32 pick = __shfl(id ^ (old & 1u), sub + 8 * (int)(it & 1), 64); // a row id that depends on the atomics' return
    u32 x = 0;
    for (u32 k = 0; k < new_rows; k += 8) {
      const u32 row = (pick + k * 7919u) % n_rows;
#pragma unroll
      for (int c = 0; c < NQ; c++) {
        const float4 v = rows[(size_t)row * (NQ * 8) + c * 8 + t];
        x += __float_as_uint(v.x) ^ __float_as_uint(v.w);
      }
    }
    acc += x;
    node = (__shfl(id, 0, 64) + (x & 1u)) % n_rows;                 // the next "pop" depends on the rows
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
  const u32 n_rows = argc > 1 ? (u32)atoll(argv[1]) : 5000000u;
  const u32 row_bytes = argc > 2 ? (u32)atoi(argv[2]) : 128u;
  const int waves_max = argc > 3 ? atoi(argv[3]) : 5120;
  const u32 words = argc > 4 ? (u32)atoll(argv[4]) : (n_rows + 31) / 32;
  const int NQ = row_bytes / 128;
  float4 *rows; u32 *lists, *bits, *sink;
  CK(hipMalloc(&rows, (size_t)n_rows * row_bytes));
  CK(hipMemset(rows, 1, (size_t)n_rows * row_bytes));
  CK(hipMalloc(&sink, 64));
  std::vector<u32> hl((size_t)n_rows * 32);
  u32 s = 12345u;
  for (auto &x : hl) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; x = s % n_rows; }
  CK(hipMalloc(&lists, hl.size() * 4));
  CK(hipMemcpy(lists, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bits, (size_t)waves_max * 2 * words * 4));
  CK(hipMemset(bits, 0, (size_t)waves_max * 2 * words * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1));
                                  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e-3; };
  printf("{\"n_rows\": %u, \"row_bytes\": %u, \"bitset_words_per_wave\": %u, \"results\": [\n", n_rows, row_bytes, words);
  const u32 it = 4000;
  for (int waves : {1280, 2560, 5120, 10240}) {
    if (waves > 2 * waves_max) continue;
    double t;
    if (NQ == 1) t = timed([&] { hipLaunchKernelGGL((k_rows<1, 2>), dim3(waves), dim3(64), 0, 0, rows, n_rows, it, sink); });
    else t = timed([&] { hipLaunchKernelGGL((k_rows<4, 2>), dim3(waves), dim3(64), 0, 0, rows, n_rows, it, sink); });
    double n = (double)waves * it * 16;
    printf(" {\"kind\": \"rows\", \"waves\": %d, \"rows_in_flight_per_wave\": 16, \"G_rows_per_s\": %.2f, \"TB_per_s\": %.3f},\n", waves, n / t * 1e-9, n * row_bytes / t * 1e-12);
    if (NQ == 1) {
      t = timed([&] { hipLaunchKernelGGL((k_rows<1, 8>), dim3(waves), dim3(64), 0, 0, rows, n_rows, it / 4, sink); });
      n = (double)waves * (it / 4) * 64;
      printf(" {\"kind\": \"rows\", \"waves\": %d, \"rows_in_flight_per_wave\": 64, \"G_rows_per_s\": %.2f, \"TB_per_s\": %.3f},\n", waves, n / t * 1e-9, n * row_bytes / t * 1e-12);
    }
    if (waves <= 2 * waves_max) {
      t = timed([&] { hipLaunchKernelGGL((k_bits<true>), dim3(waves), dim3(64), 0, 0, bits, words, it, sink); });
      printf(" {\"kind\": \"atomic_or\", \"waves\": %d, \"G_atomics_per_s\": %.2f},\n", waves, (double)waves * it * 32 / t * 1e-9);
      t = timed([&] { hipLaunchKernelGGL((k_bits<false>), dim3(waves), dim3(64), 0, 0, bits, words, it, sink); });
      printf(" {\"kind\": \"word_load\", \"waves\": %d, \"G_loads_per_s\": %.2f},\n", waves, (double)waves * it * 32 / t * 1e-9);
      for (u32 nr : {8u, 16u}) {
        if (NQ == 1) t = timed([&] { hipLaunchKernelGGL((k_chain<1>), dim3(waves), dim3(64), 0, 0, rows, n_rows, lists, bits, words, it / 4, nr, sink); });
        else t = timed([&] { hipLaunchKernelGGL((k_chain<4>), dim3(waves), dim3(64), 0, 0, rows, n_rows, lists, bits, words, it / 4, nr, sink); });
        printf(" {\"kind\": \"chain\", \"waves\": %d, \"rows_per_expansion\": %u, \"G_expansions_per_s\": %.3f, \"us_per_expansion_per_wave\": %.2f},\n",
               waves, nr, (double)waves * (it / 4) / t * 1e-9, t / (it / 4) * 1e6);
      }
    }
  }
  printf(" {}]}\n");
  return 0;
}
