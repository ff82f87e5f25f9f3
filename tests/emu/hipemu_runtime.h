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
// Device model: HIPEMU_DEVICES (default 1) pretend devices sharing host memory.  The current device is per host
// thread as in HIP; a stream and an event remember the device that was current when they were created, and
// hipEventRecord rejects an event recorded on a stream of another device (hipErrorInvalidHandle on real HIP) —
// that rule is what the multi-GPU radix tests of the CPU tier check.
#define hipErrorInvalidDevice 101
#define hipErrorInvalidHandle 400
inline thread_local int hipemu_current_device = 0;
static inline int hipemu_device_count() { const char *e = getenv("HIPEMU_DEVICES"); int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= hipemu_device_count()) return hipErrorInvalidDevice; hipemu_current_device = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = hipemu_current_device; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = hipemu_device_count(); return hipSuccess; }
static inline int hipemu_stream_device(hipStream_t s) { return s ? *(int *)s : hipemu_current_device; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = malloc(8); *(int *)*s = hipemu_current_device; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
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
struct hipemu_event { double t; int device; };
static inline hipError_t hipEventCreate(hipEvent_t *e) {
  hipemu_event *ev = (hipemu_event *)malloc(sizeof(hipemu_event)); ev->t = 0; ev->device = hipemu_current_device; *e = ev; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (((hipemu_event *)e)->device != hipemu_stream_device(s)) return hipErrorInvalidHandle;
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  *(double *)e = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double *)b - *(double *)a); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
#define hipEventDisableTiming 0
// Peer model: HIPEMU_NO_PEER=1 makes the pretend devices unable to reach each other's memory; a peer copy between two
// devices then fails unless access was enabled (the library must ask first and take its host-staged path otherwise).
#define hipErrorPeerAccessAlreadyEnabled 704
#define hipErrorPeerAccessUnsupported 217
inline bool hipemu_peer_enabled[16][16] = {};
static inline bool hipemu_no_peer() { const char *e = getenv("HIPEMU_NO_PEER"); return e && atoi(e) != 0; }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer) { *can = (dev != peer && !hipemu_no_peer()) ? 1 : 0; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) {
  if (hipemu_no_peer()) return hipErrorPeerAccessUnsupported;
  bool &e = hipemu_peer_enabled[hipemu_current_device & 15][peer & 15];
  if (e) return hipErrorPeerAccessAlreadyEnabled;
  e = true;
  return hipSuccess;
}
static inline hipError_t hipMemcpyPeerAsync(void *d, int ddev, const void *s, int sdev, size_t n, hipStream_t) {
  if (ddev != sdev && !(hipemu_peer_enabled[ddev & 15][sdev & 15] && hipemu_peer_enabled[sdev & 15][ddev & 15])) return hipErrorPeerAccessUnsupported;
  memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetAddressRange(void **base, size_t *size, void *p) { *base = p; *size = 1; return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
