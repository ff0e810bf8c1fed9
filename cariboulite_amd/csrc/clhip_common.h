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

// atan2 for the FM phase-difference demod (finite inputs): octant reduction with one v_rcp_f32, then
// atan(a) = a * P(a^2) on [0,1], P a degree-7 near-minimax fit (max error 1.7e-7 in fp32; the stage's bar is
// 1e-5 of pi).  ~20 instruction slots where the library atan2f takes ~45.  atan2(+-0, x<0) = +-pi, atan2(0,0) = 0.
__device__ __forceinline__ float clhip_atan2f(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(__builtin_fmaxf(mx, 1.17549435e-38f));
    const float s = a * a;
    float p = -0.004668773308f;
    p = __builtin_fmaf(p, s, 0.02416618952f);
    p = __builtin_fmaf(p, s, -0.05936710079f);
    p = __builtin_fmaf(p, s, 0.09906096896f);
    p = __builtin_fmaf(p, s, -0.1401658504f);
    p = __builtin_fmaf(p, s, 0.1996923539f);
    p = __builtin_fmaf(p, s, -0.3333195972f);
    p = __builtin_fmaf(p, s, 0.9999998978f);
    float r = p * a;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}
