// tests/mockhip/mockhip.c -- TEST INFRASTRUCTURE, not a CPU path of the product: a stand-in for the HIP runtime whose kernels do
// NOTHING (hipLaunchKernel returns at once, device memory is zeroed host memory), preloaded by tests/test_planner_host.py so that
// the HOST side of the library -- segment bookkeeping, the planner, the result unpacking -- can be executed on a box without a GPU.
// Every search under it returns empty results; what is learned is that the host code runs through and what plan it builds
// (NRTGPU_PLAN_TRACE).  The product never loads this: nrtgpu_create fails without a gfx950 device (tests/test_abi.py).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <stdio.h>
static long n_launch = 0;
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { (void)d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int d) {
  (void)d; memset(p, 0, sizeof *p); p->multiProcessorCount = 256; strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->totalGlobalMem = (size_t)288 << 30; p->sharedMemPerBlock = 160 * 1024; p->warpSize = 64; p->maxThreadsPerBlock = 1024; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { void* q = NULL; if (posix_memalign(&q, 256, n ? n : 256)) return hipErrorOutOfMemory; memset(q, 0, n); *p = q; return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned f) { (void)f; return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void* p, size_t n, unsigned f) { (void)p; (void)n; (void)f; return hipSuccess; }
hipError_t hipHostUnregister(void* p) { (void)p; return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned f) { (void)f; *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { (void)k; if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) { (void)k; (void)st; if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { (void)st; memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { (void)f; *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
/* MOCKHIP_SYNC_US: how long a stream synchronisation "takes" (0 by default): lets a test keep searches in flight for a while */
static void mock_sync_delay(void) {
  static int us = -1;
  if (us < 0) { const char* e = getenv("MOCKHIP_SYNC_US"); us = e ? atoi(e) : 0; }
  if (us > 0) usleep((useconds_t)us);
}
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; mock_sync_delay(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned f) { (void)s; (void)e; (void)f; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned f) { (void)f; *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)a; (void)b; *ms = 0.001f; return hipSuccess; }
// mockhip_fail_next_launches(n): the next n kernel launches fail (hipErrorLaunchFailure, also what the launching thread's next
// hipGetLastError() says, once): how a test sees what the library does with a launch that failed
static int fail_launches = 0;
static __thread hipError_t sticky_error = hipSuccess;
void mockhip_fail_next_launches(int n) { __atomic_store_n(&fail_launches, n, __ATOMIC_SEQ_CST); }
hipError_t hipGetLastError(void) { hipError_t e = sticky_error; sticky_error = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { (void)e; return "mock hip"; }
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v) { (void)f; (void)a; (void)v; return hipSuccess; }
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t sm, hipStream_t s) {
  (void)f; (void)g; (void)b; (void)args; (void)sm; (void)s;
  ++n_launch;
  if (__atomic_load_n(&fail_launches, __ATOMIC_SEQ_CST) > 0 && __atomic_fetch_sub(&fail_launches, 1, __ATOMIC_SEQ_CST) > 0) {
    sticky_error = hipErrorLaunchFailure;
    return hipErrorLaunchFailure;
  }
  return hipSuccess;
}
static void* fat_handle[4];
void** __hipRegisterFatBinary(const void* data) { (void)data; return fat_handle; }
void __hipRegisterFunction(void** m, const void* hf, char* df, const char* dn, unsigned tl, void* tid, void* bid, void* bd, void* gd, int* ws) { (void)m; (void)hf; (void)df; (void)dn; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws; }
void __hipRegisterVar(void** m, void* v, char* a, const char* b, int c, size_t d, int e, int f) { (void)m; (void)v; (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; }
void __hipUnregisterFatBinary(void** m) { (void)m; }
static __thread dim3 cfg_g, cfg_b; static __thread size_t cfg_sm; static __thread hipStream_t cfg_s;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sm, hipStream_t s) { cfg_g = g; cfg_b = b; cfg_sm = sm; cfg_s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sm, hipStream_t* s) { *g = cfg_g; *b = cfg_b; *sm = cfg_sm; *s = cfg_s; return hipSuccess; }
long mockhip_launches(void) { return n_launch; }
