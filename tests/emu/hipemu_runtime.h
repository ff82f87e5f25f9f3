// hipemu_runtime.h — host stand-ins for the few HIP runtime calls the backend's C-ABI layer
// makes (TEST INFRASTRUCTURE ONLY, see hipemu.h).  "Device" memory is host memory, streams
// are synchronous.
#pragma once
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; size_t sharedMemPerBlock; char gcnArchName[64]; };

static inline const char *hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = n ? aligned_alloc(256, (n + 255) & ~(size_t)255) : nullptr; return hipSuccess; }
static inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)64 << 30; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)64 << 30; p->sharedMemPerBlock = 160 * 1024;
  strcpy(p->gcnArchName, "hipemu");
  return hipSuccess;
}
#include <time.h>
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = malloc(sizeof(double)); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  *(double *)e = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double *)b - *(double *)a); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = malloc(sizeof(double)); return hipSuccess; }
#define hipEventDisableTiming 0
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemGetAddressRange(void **base, size_t *size, void *p) { *base = p; *size = 1; return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
