// Shared helpers for the gfx950 kernels of libepn_so3conv.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/epn_so3conv.h"

#define EPN_WAVE 64

#define EPN_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t _e = hipGetLastError();         \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

#define EPN_HIP(call)                              \
    do {                                           \
        hipError_t _e = (call);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

static inline hipStream_t epn_stream(epn_stream_t s) { return (hipStream_t)s; }

static inline int epn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Canonical squared norm shared bit-for-bit with oracle/epn_oracle.c:
//   t = a*a; t = fma(b,b,t); t = fma(c,c,t)   (explicit, so -ffp-contract cannot change it)
__device__ __forceinline__ float epn_sq3(float a, float b, float c) {
    float t = __fmul_rn(a, a);
    t = __fmaf_rn(b, b, t);
    t = __fmaf_rn(c, c, t);
    return t;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
