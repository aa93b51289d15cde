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

int main(int argc, char** argv) {
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
