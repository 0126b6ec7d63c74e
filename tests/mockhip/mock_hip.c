/*
 * mock_hip.c -- a HIP runtime made of host memory and no-ops.  TESTS ONLY (tests/mockgpu.py).
 *
 * The product's host C (dropin.c, combine.c, plan.c, frame_table.c, buffer_pool.c) is compiled unchanged and linked against
 * THIS instead of libamdhip64, with tests/mockhip/mock_launch.cpp standing in for hip_launch.hip: every launch runs the
 * product kernels under the CPU fiber emulator (tests/hipemu), synchronously.  That puts the host logic that otherwise only
 * runs on a GPU box -- flat combining of concurrent drop-in calls, staging of sampled pixels, frame-table publish forms,
 * plans -- under the CPU suite and under ThreadSanitizer.  Nothing here is part of libasciichat_hip.so, and the product has
 * no path to it: the real library fails with ERR_NO_DEVICE where there is no GPU.
 *
 * Model: "device memory" is host memory (64-byte aligned, a guard tail), a mapped host allocation is its own device
 * alias, copies happen at the call, streams and events are tokens, every query says "done".
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdlib.h>
#include <string.h>

static __thread hipError_t g_last = hipSuccess;

static void *mock_alloc(size_t n) {
  void *p = NULL;
  if (posix_memalign(&p, 64, n + 64) != 0)
    return NULL;
  memset(p, 0xA5, n + 64); /* poison: nothing may rely on fresh device memory being zero */
  return p;
}

hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t attr, int dev) {
  (void)dev;
  *v = attr == hipDeviceAttributeMultiprocessorCount ? 8 : 0; /* a small "GPU": row-band policies engage early */
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) {
  const hipError_t e = g_last;
  g_last = hipSuccess;
  return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock HIP error"; }

hipError_t hipMalloc(void **p, size_t n) { return (*p = mock_alloc(n)) ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) {
  free(p);
  return hipSuccess;
}
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t s) {
  (void)s;
  return hipMalloc(p, n);
}
hipError_t hipFreeAsync(void *p, hipStream_t s) {
  (void)s;
  return hipFree(p);
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int flags) {
  (void)flags;
  return (*p = mock_alloc(n)) ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
  free(p);
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attr, const void *ptr) { /* nothing is known about any pointer here */
  (void)attr, (void)ptr;
  return hipErrorInvalidValue;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned int flags) {
  (void)flags;
  *dev = host;
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) {
  (void)k;
  memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st) {
  (void)st;
  return hipMemcpy(d, s, n, k);
}
hipError_t hipMemset(void *d, int v, size_t n) {
  memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t s) {
  (void)s;
  return hipMemset(d, v, n);
}

hipError_t hipStreamCreate(hipStream_t *s) { return (*s = (hipStream_t)malloc(16)) ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int flags) {
  (void)flags;
  return hipStreamCreate(s);
}
hipError_t hipStreamDestroy(hipStream_t s) {
  free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  (void)s;
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
  (void)s;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int flags) {
  (void)s, (void)e, (void)flags;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return (*e = (hipEvent_t)malloc(16)) ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int flags) {
  (void)flags;
  return hipEventCreate(e);
}
hipError_t hipEventDestroy(hipEvent_t e) {
  free(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  (void)e, (void)s;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  (void)e;
  return hipSuccess;
}
/* stream capture / graphs (asciichat_hip_schedule_*): not modelled */
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m) {
  (void)s, (void)m;
  return g_last = hipErrorNotSupported;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) {
  (void)s, (void)g;
  return g_last = hipErrorNotSupported;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *x, hipGraph_t g, hipGraphNode_t *n, char *log, size_t sz) {
  (void)x, (void)g, (void)n, (void)log, (void)sz;
  return g_last = hipErrorNotSupported;
}
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t s) {
  (void)x, (void)s;
  return g_last = hipErrorNotSupported;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t x) {
  (void)x;
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) {
  (void)g;
  return hipSuccess;
}
