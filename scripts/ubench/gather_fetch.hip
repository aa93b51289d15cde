// Microbenchmark: what rocprofv3's FETCH_SIZE tallies for the access patterns of bm25_maxscore_kernel's lookups -- random 2 / 4 /
// 8 / 16-byte gathers -- next to the wide coalesced stream MI355X_MICROARCH.md (HBM section) calibrates ("tallied at half").
// Every kernel reads a KNOWN number of bytes from a buffer far larger than the Infinity Cache (default 4 GiB; the gathers'
// addresses are a bijective scramble of the thread index, so no cache line is asked for twice within a launch):
//   stream16 : 16 B per lane, consecutive lanes consecutive -- requested bytes = lines touched x 64
//   gatherN  : one N-byte word per lane at a scrambled line index -- requested bytes = N x loads; lines touched = loads x 64 B
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (a run of its own: scripts/r05/gpu_*.sh); the program prints per kernel
// the bytes requested and the 64-byte lines touched; the factor "bytes of lines touched / FETCH_SIZE as counted" per pattern is
// what profiles/pmc_traffic.json's correction uses (profiles/r05_gather_fetch.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// line index of load i: an odd multiplier modulo a power of two is a bijection -- every load its own 64-byte line
__device__ __forceinline__ uint64_t scramble(uint64_t i, uint64_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask; }

__global__ __launch_bounds__(256) void stream16(const u32x4* __restrict__ p, uint64_t n_vec, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const u32x4 v = __builtin_nontemporal_load(p + i);
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345u) *sink = acc;
}

template <int BYTES>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ base, uint64_t n_loads, uint64_t line_mask, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += stride) {
    const char* a = base + scramble(i, line_mask) * 64ull + ((i * 5ull) & (64ull / BYTES - 1ull)) * BYTES;   // some word of the line
    if (BYTES == 2) acc += *(const uint16_t*)a;
    else if (BYTES == 4) acc += *(const uint32_t*)a;
    else if (BYTES == 8) { const u32x2 v = *(const u32x2*)a; acc += v[0] ^ v[1]; }
    else { const u32x4 v = *(const u32x4*)a; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  }
  if (acc == 0x12345u) *sink = acc;
}

// `gather_fetch sweep [MiB ...]` (round 6): the gather ceiling as a function of the FOOTPRINT the gathers fall into -- 2^26 8-byte
// gathers per launch at line (i x P) mod n_lines (P prime: a bijection on any n_lines it does not divide; every line is asked for
// 2^26 / n_lines times per launch, scattered over the launch) -- from a footprint one XCD's L2 holds to one far beyond the Infinity
// Cache.  1536 MiB is bm25_maxscore_kernel's resident index at C3.  Under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ...` the
// launches are told apart by their order (the program prints the footprints in launch order).
__global__ __launch_bounds__(256) void gather_mod(const char* __restrict__ base, uint64_t n_loads, uint64_t n_lines, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += stride) {
    const uint64_t line = (i * 2654435761ull) % n_lines;
    const u32x2 v = *(const u32x2*)(base + line * 64ull + ((i * 5ull) & 7ull) * 8ull);
    acc += v[0] ^ v[1];
  }
  if (acc == 0x12345u) *sink = acc;
}

// `gather_fetch runs` (round 6): what a gather instruction costs as a function of the DISTINCT LINES its 64 lanes ask for -- lanes
// in runs of `run` share a 64-byte line (8-byte words of it), the runs' lines scattered over the footprint -- at a footprint the
// vector L1 holds (16 KiB), one the L2 holds (2 MiB) and the walk's mix (64 MiB).  If the time follows the lines and not the lanes,
// the vector L1's tag path is what a gather occupies, and lanes that walk adjacent docids are worth arranging.
__global__ __launch_bounds__(256) void gather_runs(const char* __restrict__ base, uint64_t n_loads, uint64_t n_lines, uint32_t run, uint32_t* sink) {
  // (32-bit address arithmetic, powers of two only: a shift, a multiply, a mask per load -- the loop must not be what is measured)
  uint32_t acc = 0;
  const uint32_t stride = gridDim.x * blockDim.x, mask = (uint32_t)n_lines - 1u, sh = 31u - (uint32_t)__builtin_clz(run);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)n_loads; i += stride) {
    const uint32_t line = ((i >> sh) * 2654435761u) & mask;
    const u32x2 v = *(const u32x2*)(base + (line * 64u + ((i & (run - 1u)) & 7u) * 8u));
    acc += v[0] ^ v[1];
  }
  if (acc == 0x12345u) *sink = acc;
}

static int runs(int, char**) {
  const uint64_t kibs[3] = {16, 2048, 65536};
  const uint32_t rr[6] = {1, 2, 4, 8, 16, 64};
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc((void**)&buf, 65536ull << 10));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(buf, 1, 65536ull << 10));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const uint64_t n_loads = 1ull << 26;
  for (int f = 0; f < 3; ++f)
    for (int r = 0; r < 6; ++r) {
      const uint64_t n_lines = (kibs[f] << 10) / 64ull;
      float ms = 0.f, best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_runs, dim3(256 * 16), dim3(256), 0, 0, (const char*)buf, n_loads, n_lines, rr[r], sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double clk = 2.33e9 * 256.0;   // CU-clocks per second
      printf("runs footprint_KiB %llu lanes_per_line %u best_ms %.3f G_loads_per_s %.1f G_lines_per_s %.1f wave_instr_per_CU_clk %.4f lines_per_CU_clk %.3f\n",
             (unsigned long long)kibs[f], rr[r], best, (double)n_loads / (best * 1e-3) / 1e9, (double)n_loads / rr[r] / (best * 1e-3) / 1e9,
             (double)n_loads / 64.0 / (best * 1e-3) / clk, (double)n_loads / rr[r] / (best * 1e-3) / clk);
    }
  CHECK(hipGetLastError());
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}

static int sweep(int argc, char** argv) {
  uint64_t mibs[16] = {2, 16, 64, 128, 256, 512, 1024, 1536, 4096};
  int n = 9;
  if (argc > 2) {
    n = 0;
    for (int i = 2; i < argc && n < 16; ++i) mibs[n++] = strtoull(argv[i], nullptr, 10);
  }
  uint64_t max_mib = 0;
  for (int i = 0; i < n; ++i) max_mib = mibs[i] > max_mib ? mibs[i] : max_mib;
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc((void**)&buf, max_mib << 20));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(buf, 1, max_mib << 20));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const uint64_t n_loads = 1ull << 26;
  for (int i = 0; i < n; ++i) {
    const uint64_t n_lines = (mibs[i] << 20) / 64ull;
    float ms = 0.f, best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {   // (three launches per footprint: the first warms TLBs and caches)
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(gather_mod, dim3(256 * 16), dim3(256), 0, 0, (const char*)buf, n_loads, n_lines, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("sweep footprint_MiB %llu loads %llu visits_per_line %.2f best_ms %.3f G_lines_per_s %.1f line_GBps %.1f\n", (unsigned long long)mibs[i],
           (unsigned long long)n_loads, (double)n_loads / (double)n_lines, best, (double)n_loads / (best * 1e-3) / 1e9, (double)n_loads * 64.0 / (best * 1e-3) / 1e9);
  }
  CHECK(hipGetLastError());
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 's') return sweep(argc, argv);
  if (argc > 1 && argv[1][0] == 'r') return runs(argc, argv);
  const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4ull;
  const uint64_t bytes = gib << 30;                 // a power of two: the scramble is a bijection on its lines
  const uint64_t n_lines = bytes / 64ull;
  const uint64_t n_loads = n_lines / 4ull;          // a quarter of the lines per gather launch: 1 GiB of lines at 4 GiB
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc((void**)&buf, bytes));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid(256 * 16), block(256);
  auto report = [&](const char* name, uint64_t requested, uint64_t lines, float ms) {
    printf("%-10s requested_bytes %llu lines_touched %llu line_bytes %llu ms %.3f line_GBps %.1f\n", name, (unsigned long long)requested,
           (unsigned long long)lines, (unsigned long long)(lines * 64ull), ms, (double)(lines * 64ull) / (ms * 1e-3) / 1e9);
  };
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {   // (the second round is the one to read: the first pays page faults / TLB fills)
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const u32x4*)buf, bytes / 16ull, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    report("stream16", bytes, n_lines, ms);
#define RUN_GATHER(B)                                                                                   \
    CHECK(hipEventRecord(e0));                                                                            \
    hipLaunchKernelGGL((gather<B>), grid, block, 0, 0, (const char*)buf, n_loads, n_lines - 1ull, sink);  \
    CHECK(hipEventRecord(e1));                                                                            \
    CHECK(hipEventSynchronize(e1));                                                                       \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                              \
    report("gather" #B, n_loads * B, n_loads, ms);
    RUN_GATHER(2)
    RUN_GATHER(4)
    RUN_GATHER(8)
    RUN_GATHER(16)
  }
  CHECK(hipGetLastError());
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
