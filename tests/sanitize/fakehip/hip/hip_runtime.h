// TEST INFRASTRUCTURE: the handful of HIP runtime calls gamma_amd/csrc/gamma_hip_group.cpp makes (streams, events, device
// buffers, copies), with "device memory" = host memory and every copy done on the calling thread -- so that the group's OWN
// code (member threads, barriers, go / no-go snapshots, error paths) builds with g++ and runs under ThreadSanitizer in a
// container without a GPU (tests/sanitize/Makefile).  Not a HIP implementation; never shipped.
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
#define hipErrorPeerAccessAlreadyEnabled 704
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
#define hipEventDisableTiming 2u
#define hipEventBlockingSync 1u

static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "fake hip error"; }
