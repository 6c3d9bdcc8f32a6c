// TEST-ONLY stand-in for <hip/hip_runtime.h>: the declarations csrc/vdyn_capi.hip uses, nothing else.
// tests/test_host_layer_sanitizers.py compiles the library's own host layer (vdyn_capi.hip, as plain C++ with g++)
// against this header and tests/hipstub/hip_stub.cpp under AddressSanitizer + UndefinedBehaviorSanitizer.  It is a
// stand-in for a LIBRARY the product links (the HIP runtime) inside the product's own test -- not a CPU path of the
// product, not a parity artefact, and it never touches oracle/.
#pragma once
#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorInvalidDevice = 101,
    hipErrorNoDevice = 100,
    hipErrorInvalidResourceHandle = 400,
    hipErrorNotReady = 600,
    hipErrorUnknown = 999
} hipError_t;

typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;

typedef enum hipMemcpyKind {
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4,
    hipMemcpyDeviceToDeviceNoCU = 1024
} hipMemcpyKind;

#define hipStreamDefault 0x00
#define hipStreamNonBlocking 0x01
#define hipEventDefault 0x0
#define hipEventDisableTiming 0x2
#define hipHostMallocDefault 0x0
#define hipHostMallocMapped 0x2
#define hipIpcMemLazyEnablePeerAccess 0x01

typedef struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
} hipDeviceProp_t;

typedef struct hipIpcMemHandle_st {
    char reserved[64];
} hipIpcMemHandle_t;

extern "C" {
hipError_t hipGetDeviceCount(int *count);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device);
hipError_t hipSetDevice(int device);
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);

hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned flags);

hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t event);
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream);
hipError_t hipEventSynchronize(hipEvent_t event);

hipError_t hipMalloc(void **ptr, size_t bytes);
hipError_t hipFree(void *ptr);
hipError_t hipHostMalloc(void **ptr, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *ptr);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream);

hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *handle, void *dev_ptr);
hipError_t hipIpcOpenMemHandle(void **dev_ptr, hipIpcMemHandle_t handle, unsigned flags);
hipError_t hipIpcCloseMemHandle(void *dev_ptr);
}
