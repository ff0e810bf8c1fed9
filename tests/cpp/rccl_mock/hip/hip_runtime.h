/* TEST-ONLY stand-in for <hip/hip_runtime.h> on the include path of the RCCL-mock build of clfan.cpp (see ../rccl/rccl.h):
 * the three runtime names that file uses, over host memory. */
#ifndef RCCL_MOCK_HIP_RUNTIME_H
#define RCCL_MOCK_HIP_RUNTIME_H
#include <stddef.h>
#include <string.h>
typedef struct rccl_mock_stream *hipStream_t;
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s)
{
    (void)k; (void)s;
    if (!dst || !src) return hipErrorInvalidValue;
    memmove(dst, src, n);
    return hipSuccess;
}
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipErrorInvalidValue"; }
#endif
