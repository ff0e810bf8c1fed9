// clhip_common.h -- shared declarations of the gfx950 kernel shim (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cariboulite_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CLHIP_WAVE 64

extern "C" void clhip_set_error(const char *fmt, ...);

#define CLHIP_CHECK(expr)                                                               \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            clhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
            return -1;                                                                  \
        }                                                                               \
    } while (0)

#define CLHIP_CHECK_LAUNCH() CLHIP_CHECK(hipGetLastError())

// RX word fields (caribou_smi.c:338-340): A = bits 29:17, B = bits 13:1.
// S1G: i = A, q = B.  HiF: i = B, q = A.  v_bfe_i32 sign-extends 13 bits.
__device__ __forceinline__ int clhip_field_a(uint32_t w) { return ((int32_t)(w << 2)) >> 19; }
__device__ __forceinline__ int clhip_field_b(uint32_t w) { return ((int32_t)(w << 18)) >> 19; }

static inline size_t clhip_div_up(size_t a, size_t b) { return (a + b - 1) / b; }
