// Microbenchmark: throughput of random-address LDS accumulate flavours on gfx950 (2 x 512-thread
// workgroups per CU, 64 KiB tile each, like the scan kernel).  Prints ns per wave-instruction and
// lane-ops per second.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 4) void k(uint32_t iters, uint64_t* out, uint32_t spread) {
  __shared__ double acc[8192];
  __shared__ uint32_t pad[2048];
  uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 8192; i += 512) acc[i] = 0.0;
  __syncthreads();
  uint32_t x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
  uint64_t* acc64 = (uint64_t*)acc;
  float* acc32 = (float*)acc;
  uint32_t* accu32 = (uint32_t*)acc;
  double sum = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t idx;
    if (spread == 0) idx = (x >> 8) & 8191u;                       // uniform random over the tile
    else idx = ((tid * 4 + (it & 3)) * spread + ((x >> 8) % spread)) & 8191u;  // ascending with random gaps
    if (MODE == 0) unsafeAtomicAdd(&acc[idx], 1.0);                // ds_add_f64
    else if (MODE == 1) atomicAdd((unsigned long long*)&acc64[idx], 1ull);  // ds_add_u64
    else if (MODE == 2) unsafeAtomicAdd(&acc32[idx], 1.0f);        // ds_add_f32
    else if (MODE == 3) atomicAdd(&accu32[idx], 1u);               // ds_add_u32
    else if (MODE == 4) { double v = acc[idx]; acc[idx] = v + 1.0; }  // plain RMW b64
    else if (MODE == 5) acc[idx] = (double)it;                     // plain write b64
    else if (MODE == 6) sum += acc[idx];                           // plain read b64
    else if (MODE == 7) atomicOr((unsigned long long*)&acc64[idx], 1ull << (it & 63));  // ds_or_b64
    else if (MODE == 8) atomicMax((unsigned long long*)&acc64[idx], (unsigned long long)x); // ds_max_u64
    else if (MODE == 9) sum += unsafeAtomicAdd(&acc[idx], 1.0);   // ds_add_rtn_f64
    else if (MODE == 10) sum += (double)atomicExch((unsigned long long*)&acc64[idx], (unsigned long long)x);  // ds_wrxchg_rtn_b64
    else if (MODE == 11) { if (tid & 1) unsafeAtomicAdd(&acc[idx], 1.0); }          // ds_add_f64, half the lanes
    else if (MODE == 12) { if (tid & 1) sum += unsafeAtomicAdd(&acc[idx], 1.0); }   // ds_add_rtn_f64, half the lanes
    else if (MODE == 13) { if ((tid & 3) == 0) unsafeAtomicAdd(&acc[idx], 1.0); }   // ds_add_f64, quarter of the lanes
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = (uint64_t)acc[0] + (uint64_t)sum + pad[0];
}

template <int MODE>
int run(const char* name, uint32_t spread) {
  const uint32_t iters = 4096, blocks = 512 * 4;
  uint64_t* d;
  CHECK(hipMalloc(&d, blocks * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, 64u, d, spread);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, iters, d, spread);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 512 * iters;
  // per CU: 2 blocks resident; wave-instr per CU = ops/64/256
  double ns_per_wave_instr_per_cu = ms * 1e6 / (ops / 64.0 / 256.0);
  printf("%-14s spread=%2u  %8.3f ms  %7.2f Gops/s  %6.1f ns per wave-instr per CU (%.0f cycles @2.1GHz)\n", name, spread, ms,
         ops / ms / 1e6, ns_per_wave_instr_per_cu, ns_per_wave_instr_per_cu * 2.1);
  CHECK(hipFree(d));
  return 0;
}

int main() {
  for (uint32_t spread : {0u, 2u}) {
    run<0>("ds_add_f64", spread); run<1>("ds_add_u64", spread); run<2>("ds_add_f32", spread); run<3>("ds_add_u32", spread);
    run<4>("rmw_b64", spread); run<5>("write_b64", spread); run<6>("read_b64", spread); run<7>("ds_or_b64", spread); run<8>("ds_max_u64", spread);
    run<9>("ds_add_rtn_f64", spread); run<10>("ds_xchg_rtn64", spread); run<11>("add_f64_half", spread); run<12>("add_rtn_half", spread); run<13>("add_f64_quart", spread);
  }
  return 0;
}
