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

// Every launch site records the host stub of the kernel it launches; epn_last_kernel() turns it into the exact template
// instance name (what a profiler will call the kernel), so that a benchmark never has to re-derive the launchers' tile
// choices.  AUX launches (table set-up, operand re-packing, partial-sum reductions) only record themselves when the
// call has launched nothing else.  Cost per launch: two thread-local stores.
namespace epn {
void note_kernel(const void *host_stub, bool aux);   // c_api.hip; thread-local, diagnostic only
bool first_layer_on_valu();                           // c_api.hip: epn_set_kernel_policy(2)
}  // namespace epn
#define EPN_LAUNCH(kern, ...)                                                       \
    do {                                                                            \
        ::epn::note_kernel(reinterpret_cast<const void *>(&kern), false);           \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                                      \
    } while (0)
#define EPN_LAUNCH_AUX(kern, ...)                                                   \
    do {                                                                            \
        ::epn::note_kernel(reinterpret_cast<const void *>(&kern), true);            \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                                      \
    } while (0)

static inline int epn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Canonical squared norm shared bit-for-bit with oracle/epn_oracle.c:
//   t = a*a; t = fma(b,b,t); t = fma(c,c,t)   (explicit, so -ffp-contract cannot change it)
__device__ __forceinline__ float epn_sq3(float a, float b, float c) {
    float t = __fmul_rn(a, a);
    t = __fmaf_rn(b, b, t);
    t = __fmaf_rn(c, c, t);
    return t;
}

// XCD-aware workgroup remap (cdna_hip_programming.md T1, bijective form).  Workgroup b is observed to run on XCD
// b % 8, each XCD with a private 4 MB L2.  Column tiles are ordered cloud-major, and the feature rows a tile gathers
// all lie in its own cloud's slab, so giving every XCD one CONTIGUOUS range of tiles keeps a slab's re-reads inside
// one L2.  Placement only affects speed, never results.
__device__ __forceinline__ unsigned epn_xcd_tile(unsigned bid, unsigned nb) {
    const unsigned xcd = bid & 7u, local = bid >> 3;
    const unsigned q = nb >> 3, r = nb & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
