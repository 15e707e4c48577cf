// Index kernels for gfx950: furthest point sampling, ball query, point gather fwd/bwd.
//
// Semantics follow the reference's CUDA-only kernels (restated on CPU in oracle/epn_oracle.c):
//   vgtk/vgtk/cuda/grouping_cuda_kernel.cu:67-113   ball query
//   vgtk/vgtk/cuda/grouping_cuda_kernel.cu:339-466  FPS (+ __update tie-breaking)
//   vgtk/vgtk/cuda/gathering_cuda_kernel.cu:43-98   gather fwd / bwd
// The launch geometry is NOT the reference's (one block per cloud, one thread per query):
//   * FPS keeps every point and its running min-distance in REGISTERS, reduces (value, tie-key,
//     index) as one 64-bit key with wave shuffles + one LDS hop and ONE barrier per round;
//   * ball query gives each query a 64-lane wavefront that tests 64 support points per step and
//     compacts hits in index order with ballot + prefix popcount (early exit once full);
//   * gather maps the fastest thread index to the output index (coalesced writes).
#include <cmath>

#include "epn_common.h"

namespace {

// ------------------------------------------------------------------------------------ FPS
// Reference tie-breaking, stated as a total order so any reduction tree reproduces it:
// virtual thread t (0 <= t < block, block = min(1024, 2^floor(log2 n))) keeps the FIRST maximum
// among k = t, t+block, ... (strict '>', starting from best=-1, besti=0).  The shared-memory tree
// (offsets block/2 .. 1, "keep idx1 on ties") then prefers, among equal values, the thread whose
// index has a 0 at the lowest differing bit, i.e. the smallest bit-reversed index.
// key = [fkey(value):32][1023 - bitrev10(t):10][besti:22], reduced with max().
__device__ __forceinline__ unsigned long long fps_key(float best, int besti, unsigned t) {
    const unsigned fk = best < 0.0f ? 0u : (__float_as_uint(best) + 1u);
    const unsigned pri = 1023u - (__brev(t) >> 22);
    return ((unsigned long long)fk << 32) | ((unsigned long long)pri << 22) | (unsigned)besti;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

template <int PPT>
__global__ __launch_bounds__(1024) void fps_kernel(const float *__restrict__ xyz, int n, int m,
                                                   int block, int cloud_in_lds,
                                                   int32_t *__restrict__ idxs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);  // [2][16]
    float *cloud_lds = reinterpret_cast<float *>(smem + 2 * 16 * sizeof(unsigned long long));  // [3][n]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
    const float *d = xyz + (size_t)blockIdx.x * 3 * n;
    int32_t *out = idxs + (size_t)blockIdx.x * m;

    // the cloud is re-read once per round (the last winner's coordinates): from LDS when it fits
    const float *cloud = d;
    if (cloud_in_lds) {
        for (int i = tid; i < 3 * n; i += blockDim.x) cloud_lds[i] = d[i];
        cloud = cloud_lds;
    }

    float px[PPT], py[PPT], pz[PPT], tmp[PPT];
    bool live[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * block;
        const bool in = tid < block && k < n;
        px[i] = in ? d[k] : 0.f;
        py[i] = in ? d[n + k] : 0.f;
        pz[i] = in ? d[2 * n + k] : 0.f;
        tmp[i] = 1e10f;
        // reference: `if (mag <= 1e-3) continue;` -- float mag promoted against a double literal
        live[i] = in && !((double)epn_sq3(px[i], py[i], pz[i]) <= 1e-3);
    }
    if (tid == 0) out[0] = 0;
    __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = cloud[old], y1 = cloud[n + old], z1 = cloud[2 * n + old];
        float best = -1.0f;
        int besti = 0;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (live[i]) {
                const float dd = epn_sq3(px[i] - x1, py[i] - y1, pz[i] - z1);
                const float d2 = dd < tmp[i] ? dd : tmp[i];
                tmp[i] = d2;
                if (d2 > best) {
                    best = d2;
                    besti = tid + i * block;
                }
            }
        }
        unsigned long long key = tid < block ? fps_key(best, besti, (unsigned)tid) : 0ull;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = shfl_xor_u64(key, s);
            key = o > key ? o : key;
        }
        if (nwave > 1) {
            unsigned long long *slot = slots + (j & 1) * 16;
            if (lane == 0) slot[wave] = key;
            __syncthreads();
            key = (lane & 15) < nwave ? slot[lane & 15] : 0ull;
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) {
                const unsigned long long o = shfl_xor_u64(key, s);
                key = o > key ? o : key;
            }
        }
        old = (int)(key & 0x3FFFFFu);
        if (tid == 0) out[j] = old;
    }
}

// DPP-reduction FPS (n <= 8192): the m - 1 rounds of FPS are a dependent chain, so the round latency is all that matters
// (32 clouds are 32 workgroups on 256 CUs whatever the geometry).  Every lane keeps PPT points and their running
// min-distances in registers; the arg-max of a round is one 64-bit max per wave done with DPP moves (xor 1, xor 2,
// half-row mirror, row mirror, row_bcast15, row_bcast31: register-to-register, a few cycles each, against ~60 for a
// ds_bpermute shuffle), one LDS slot per wave, ONE barrier, and <= 4 uniform-address LDS reads; the winner's coordinates
// come back as uniform-address LDS reads too.  4 waves per cloud: a single wave issues ~1 VALU instruction per 5 cycles,
// so PPT = 16 on one wave (measured: 670 us for m = 512) loses to PPT = 4 on four.  The reference's tie-breaking is the total order stated above fps_key, extended
// by "inside one virtual thread the smaller index wins" (its strict `>` scan), encoded as key =
// [fkey(value):32][1023 - bitrev10(k mod block):10][0x3FFFFF - k:22] per POINT, so any assignment of points to lanes
// reproduces it.  Round 1's multi-wave kernel took 570 us for m = 512; this one ~90 us.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_max_u64(unsigned &hi, unsigned &lo) {
    const unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const bool take = oh > hi || (oh == hi && ol > lo);
    hi = take ? oh : hi;
    lo = take ? ol : lo;
}

template <int PPT, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void fps_wave_kernel(const float *__restrict__ xyz, int n, int m, int block,
                                                              int32_t *__restrict__ idxs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);             // [2][NWAVE]
    float *cloud = reinterpret_cast<float *>(smem + 2 * 8 * sizeof(unsigned long long));  // [3][n]
    constexpr int T = 64 * NWAVE;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *d = xyz + (size_t)blockIdx.x * 3 * n;
    int32_t *out = idxs + (size_t)blockIdx.x * m;
    for (int i = tid; i < 3 * n; i += T) cloud[i] = d[i];

    float px[PPT], py[PPT], pz[PPT], tmp[PPT];
    unsigned pri[PPT];      // 0 for points that never compete (outside the cloud, or inside the 1e-3 dead zone)
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + T * i;
        const bool in = k < n;
        px[i] = in ? d[k] : 0.f;
        py[i] = in ? d[n + k] : 0.f;
        pz[i] = in ? d[2 * n + k] : 0.f;
        tmp[i] = 1e10f;
        const bool live = in && !((double)epn_sq3(px[i], py[i], pz[i]) <= 1e-3);
        const unsigned t = (unsigned)(k % block);                       // the reference's thread of point k
        pri[i] = live ? (((1023u - (__brev(t) >> 22)) << 22) | (0x3FFFFFu - (unsigned)k)) : 0u;
    }
    if (tid == 0) out[0] = 0;
    __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = cloud[old], y1 = cloud[n + old], z1 = cloud[2 * n + old];
        unsigned long long best = 0ull;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float dd = epn_sq3(px[i] - x1, py[i] - y1, pz[i] - z1);
            const float d2 = fminf(dd, tmp[i]);
            tmp[i] = d2;
            // live points: key = [bits(d2) + 1][pri]; the others keep key 0 (pri == 0 marks them; a live point's pri is
            // never 0 because its low field is 0x3FFFFF - k > 0)
            const unsigned long long key =
                pri[i] ? (((unsigned long long)(__float_as_uint(d2) + 1u) << 32) | pri[i]) : 0ull;
            best = key > best ? key : best;
        }
        unsigned bh = (unsigned)(best >> 32), bl = (unsigned)best;
        dpp_max_u64<0xB1, 0xf>(bh, bl);     // quad_perm [1,0,3,2]: xor 1
        dpp_max_u64<0x4E, 0xf>(bh, bl);     // quad_perm [2,3,0,1]: xor 2
        dpp_max_u64<0x141, 0xf>(bh, bl);    // row_half_mirror: 8 lanes
        dpp_max_u64<0x140, 0xf>(bh, bl);    // row_mirror: 16 lanes
        dpp_max_u64<0x142, 0xa>(bh, bl);    // row_bcast15 into rows 1, 3
        dpp_max_u64<0x143, 0xc>(bh, bl);    // row_bcast31 into rows 2, 3: lane 63 holds the wave maximum
        unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)bh, 63) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)bl, 63);
        if (NWAVE > 1) {
            unsigned long long *slot = slots + (j & 1) * 8;        // two slot sets: one barrier per round suffices
            if ((tid & 63) == 0) slot[wave] = w;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NWAVE; ++q) {
                const unsigned long long o = slot[q];               // uniform address: LDS broadcast
                w = o > w ? o : w;
            }
        }
        old = (w >> 32) == 0ull ? 0 : (int)(0x3FFFFFu - ((unsigned)w & 0x3FFFFFu));   // no live point: besti stays 0
        if (tid == 0) out[j] = old;
    }
}

// ------------------------------------------------------------------------------------ ball query
// canonical squared norm in T (epn_sq3 for float; the same fma chain in double for the fp64 dispatch)
__device__ __forceinline__ float sq3_t(float a, float b, float c) { return epn_sq3(a, b, c); }
__device__ __forceinline__ double sq3_t(double a, double b, double c) {
    double t = __dmul_rn(a, a);
    t = __fma_rn(b, b, t);
    t = __fma_rn(c, c, t);
    return t;
}

template <typename T>   // float, or double (the reference dispatches on both: grouping_cuda_kernel.cu:477)
__global__ __launch_bounds__(256) void ball_query_kernel(const T *__restrict__ new_xyz,
                                                         const T *__restrict__ xyz, int n, int m,
                                                         float radius, int nsample,
                                                         int32_t *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t *row = reinterpret_cast<int32_t *>(smem) + wave * nsample;
    const int bi = blockIdx.y;
    const int j = blockIdx.x * (blockDim.x >> 6) + wave;
    if (j >= m) return;  // whole wave exits together; no block-level barrier below

    const T *q = new_xyz + (size_t)bi * 3 * m;
    const T *s = xyz + (size_t)bi * 3 * n;
    const T qx = q[j], qy = q[m + j], qz = q[2 * m + j];
    // reference: kernel parameter `float radius`, `scalar_t radius2 = radius * radius;` (grouping_cuda_kernel.cu:67,80):
    // the square is ONE rounded FLOAT multiply, widened to T afterwards -- also for T = double
    const T radius2 = (T)__fmul_rn(radius, radius);

    for (int t = lane; t < nsample; t += 64) row[t] = 0;  // reference zero-initialises idx

    int cnt = 0;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) {
            const T d2 = sq3_t(qx - s[k], qy - s[n + k], qz - s[2 * n + k]);
            hit = d2 < radius2;
        }
        const unsigned long long mask = __ballot(hit);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < nsample) row[pos] = k;
        cnt += __popcll(mask);
    }
    if (cnt > nsample) cnt = nsample;
    __builtin_amdgcn_wave_barrier();
    // reference :100-104 -- cyclic repeat of the first cnt hits, only when cnt < nsample-1
    // (cnt == nsample-1 leaves the last slot 0, cnt == 0 leaves the row 0).
    int32_t *o = idx + ((size_t)bi * m + j) * nsample;
    for (int t = lane; t < nsample; t += 64) {
        int v = row[t];
        if (cnt > 0 && cnt < nsample - 1 && t >= cnt) v = row[t % cnt];
        o[t] = v;
    }
}

// ------------------------------------------------------------------------------------ gather
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const T *__restrict__ points,
                                                         const int32_t *__restrict__ idx, int c, int n,
                                                         int m, T *__restrict__ out) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    out[((size_t)bi * c + ci) * m + j] = points[((size_t)bi * c + ci) * n + a];
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_kernel(const T *__restrict__ grad_out,
                                                         const int32_t *__restrict__ idx, int c, int n,
                                                         int m, T *__restrict__ grad_points) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    atomicAdd(grad_points + ((size_t)bi * c + ci) * n + a, grad_out[((size_t)bi * c + ci) * m + j]);
}

// FPS with the running minima in global memory (furthest_point_sampling_cuda_kernel<scalar_t>, grouping_cuda_kernel.cu:351-466):
// the reference's own structure -- `block` virtual threads scanning k = tid, tid + block, ... (no size limit: the strided
// loop of :380-396), a shared-memory tree with its __update tie rule (keep the first on ties) -- with `temp` exactly like
// the reference.  T = double: the fp64 dispatch (no shipped model samples fp64 coordinates); T = float: clouds beyond the
// 32768 points the register-resident kernels above hold (epn_fps_temp_f32).  Neither is a hot path.
template <typename T>
__global__ __launch_bounds__(1024) void fps_temp_kernel(const T *__restrict__ xyz, int n, int m, int block,
                                                        T *__restrict__ temp, int32_t *__restrict__ idxs) {
    __shared__ T dists[1024];
    __shared__ int dists_i[1024];
    const int tid = threadIdx.x;
    const T *d = xyz + (size_t)blockIdx.x * 3 * n;
    T *tmp = temp + (size_t)blockIdx.x * n;
    int32_t *out = idxs + (size_t)blockIdx.x * m;
    for (int k = tid; k < n; k += block) tmp[k] = (T)1e10;
    if (tid == 0) out[0] = 0;
    __syncthreads();
    int old = 0;
    for (int j = 1; j < m; ++j) {
        int besti = 0;
        T best = (T)-1.0;
        const T x1 = d[old], y1 = d[n + old], z1 = d[2 * n + old];
        for (int k = tid; k < n; k += block) {
            const T x2 = d[k], y2 = d[n + k], z2 = d[2 * n + k];
            if ((double)sq3_t(x2, y2, z2) <= 1e-3) continue;       // `mag <= 1e-3`: a double literal (:385-387)
            const T dd = sq3_t(x2 - x1, y2 - y1, z2 - z1);
            const T d2 = dd < tmp[k] ? dd : tmp[k];
            tmp[k] = d2;
            besti = d2 > best ? k : besti;
            best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
        __syncthreads();
        for (int off = block / 2; off >= 1; off >>= 1) {
            if (tid < off) {
                const T v1 = dists[tid], v2 = dists[tid + off];
                const int i1 = dists_i[tid], i2 = dists_i[tid + off];
                dists[tid] = v1 > v2 ? v1 : v2;
                dists_i[tid] = v2 > v1 ? i2 : i1;
            }
            __syncthreads();
        }
        old = dists_i[0];
        if (tid == 0) out[j] = old;
        __syncthreads();
    }
}

int opt_n_threads(int work_size) {  // grouping_cuda_kernel.cu:29-33, same double-precision formula
    const int pow_2 = (int)(std::log((double)work_size) / std::log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    return t < 1 ? 1 : t;
}

}  // namespace

extern "C" int epn_ball_query_f32(const float *new_xyz, const float *xyz, int b, int n, int m,
                                  float radius, int nsample, int32_t *idx, epn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || nsample < 1 || nsample > 4096) return EPN_EINVAL;
    if (b == 0 || m == 0) return 0;
    if (!new_xyz || !xyz || !idx) return EPN_ENULL;
    if (b > 65535) return EPN_EINVAL;
    const int waves = 4;
    dim3 grid(epn_cdiv(m, waves), b);
    EPN_LAUNCH(ball_query_kernel<float>, grid, dim3(64 * waves), waves * nsample * sizeof(int32_t),
                       epn_stream(stream), new_xyz, xyz, n, m, radius, nsample, idx);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_fps_f32(const float *xyz, int b, int n, int m, int32_t *idx, epn_stream_t stream) {
    if (b < 0 || n < 1 || n > 32768 || m < 0) return EPN_EINVAL;
    if (b == 0 || m == 0) return 0;
    if (!xyz || !idx) return EPN_ENULL;
    const int block = opt_n_threads(n);
    hipStream_t st0 = epn_stream(stream);
    if (n <= 256 * 32) {                       // DPP-reduction kernel: 1-4 waves per cloud, <= 1 barrier per round
        const size_t sh = 2 * 8 * sizeof(unsigned long long) + (size_t)3 * n * sizeof(float);
#define EPN_FPSW(P, W)                                                                                               \
    do {                                                                                                             \
        EPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_wave_kernel<P, W>),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                           \
        EPN_LAUNCH((fps_wave_kernel<P, W>), dim3(b), dim3(64 * W), sh, st0, xyz, n, m, block, idx);          \
    } while (0)
        if (n <= 64) EPN_FPSW(1, 1);
        else if (n <= 128) EPN_FPSW(1, 2);
        else if (n <= 256) EPN_FPSW(1, 4);
        else if (n <= 512) EPN_FPSW(2, 4);
        else if (n <= 1024) EPN_FPSW(4, 4);           // 8 waves x 2 points measured the same (279 vs 271 us for m = 512):
        else if (n <= 2048) EPN_FPSW(8, 4);           // the fixed per-round chain (DPP max, LDS hop, barrier) dominates
        else if (n <= 4096) EPN_FPSW(16, 4);
        else EPN_FPSW(32, 4);
#undef EPN_FPSW
        EPN_CHECK_LAUNCH();
        return 0;
    }
    const int threads = block < 64 ? 64 : block;
    const int ppt = epn_cdiv(n, block);
    const int in_lds = (size_t)3 * n * sizeof(float) <= 96 * 1024;
    const size_t shmem = 2 * 16 * sizeof(unsigned long long) + (in_lds ? (size_t)3 * n * sizeof(float) : 0);
    hipStream_t st = epn_stream(stream);
#define EPN_FPS(P)                                                                                  \
    do {                                                                                            \
        EPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_kernel<P>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));      \
        EPN_LAUNCH(fps_kernel<P>, dim3(b), dim3(threads), shmem, st, xyz, n, m, block, in_lds, idx); \
    } while (0)
    if (ppt <= 1) EPN_FPS(1);
    else if (ppt <= 2) EPN_FPS(2);
    else if (ppt <= 4) EPN_FPS(4);
    else if (ppt <= 8) EPN_FPS(8);
    else if (ppt <= 16) EPN_FPS(16);
    else EPN_FPS(32);
#undef EPN_FPS
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_gather_fwd_f32(const float *points, const int32_t *idx, int b, int c, int n, int m,
                                  float *out, epn_stream_t stream) {
    if (b < 0 || c < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || c == 0 || m == 0) return 0;
    if (!points || !idx || !out) return EPN_ENULL;
    if (c > 65535 || b > 65535) return EPN_EINVAL;
    dim3 grid(epn_cdiv(m, 256), c, b);
    EPN_LAUNCH(gather_fwd_kernel<float>, grid, dim3(256), 0, epn_stream(stream), points, idx, c, n, m, out);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_gather_bwd_f32(const float *grad_out, const int32_t *idx, int b, int c, int n, int m,
                                  float *grad_points, epn_stream_t stream) {
    if (b < 0 || c < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || c == 0) return 0;
    if (!grad_points) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, epn_stream(stream)));
    if (m == 0) return 0;
    if (!grad_out || !idx) return EPN_ENULL;
    if (c > 65535 || b > 65535) return EPN_EINVAL;
    dim3 grid(epn_cdiv(m, 256), c, b);
    EPN_LAUNCH(gather_bwd_kernel<float>, grid, dim3(256), 0, epn_stream(stream), grad_out, idx, c, n, m,
                       grad_points);
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// initial_anchor_query (grouping_cuda_kernel.cu:116-167): for every centre and every rotated kernel point, the summed
// influence  max(1 - |c + kappa - x|^2 / sigma, 0)  of the fragment points x within `radius` of the centre, and the
// number of such points.  The reference scatters with two atomics per (point, kernel, centre, anchor) and reads
// xyz[3*pm] before its bounds test; here one workgroup owns one (batch, centre): fragment points inside the ball are
// compacted chunk by chunk into LDS in index order (ballot + prefix, deterministic), every thread owns a fixed set of
// (kernel point, anchor) outputs and sums over the list in registers -- no atomics, fixed summation order.
namespace {

constexpr int AQ_T = 256;
constexpr int AQ_OUT = 8;      // outputs per thread: ks*na <= AQ_T*AQ_OUT (24*60 = 1440)

__device__ __forceinline__ float aq_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double aq_sqrt(double v) { return sqrt(v); }

template <typename T>   // float, or double (AT_DISPATCH_FLOATING_TYPES on xyz.type(), grouping_cuda_kernel.cu:558-563); radius and
// sigma stay FLOAT kernel parameters in either instantiation, as in the reference (:127-128)
__global__ __launch_bounds__(AQ_T) void initial_anchor_query_kernel(const T *__restrict__ centers,
                                                                    const T *__restrict__ xyz,
                                                                    const T *__restrict__ kp, int nc, int m,
                                                                    int na, int ks, float radius, float sigma,
                                                                    T *__restrict__ wts, T *__restrict__ cnt) {
    __shared__ T lst[AQ_T][3];
    __shared__ int wcount[AQ_T / 64];
    const int pn = blockIdx.x, bn = blockIdx.y, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const T *c = centers + (size_t)bn * 3 * nc;
    const T cx = c[pn], cy = c[nc + pn], cz = c[2 * nc + pn];
    const int nout = ks * na;
    T kx[AQ_OUT], ky[AQ_OUT], kz[AQ_OUT], acc[AQ_OUT];
#pragma unroll
    for (int u = 0; u < AQ_OUT; ++u) {
        const int o = t + AQ_T * u;                  // o = kn * na + an
        const bool ok = o < nout;
        kx[u] = ok ? kp[(size_t)o * 3] + cx : (T)0;
        ky[u] = ok ? kp[(size_t)o * 3 + 1] + cy : (T)0;
        kz[u] = ok ? kp[(size_t)o * 3 + 2] + cz : (T)0;
        acc[u] = (T)0;
    }
    int total = 0;
    for (int m0 = 0; m0 < m; m0 += AQ_T) {
        const int pm = m0 + t;
        T x = 0, y = 0, z = 0;
        bool in = false;
        if (pm < m) {
            x = xyz[(size_t)3 * pm]; y = xyz[(size_t)3 * pm + 1]; z = xyz[(size_t)3 * pm + 2];
            const T dx = cx - x, dy = cy - y, dz = cz - z;
            in = aq_sqrt(dx * dx + dy * dy + dz * dz) <= (T)radius;
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = 0, n_in = 0;
#pragma unroll
        for (int w = 0; w < AQ_T / 64; ++w) {
            base += w < wave ? wcount[w] : 0;
            n_in += wcount[w];
        }
        if (in) {
            const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
            lst[slot][0] = x; lst[slot][1] = y; lst[slot][2] = z;
        }
        __syncthreads();
        for (int i = 0; i < n_in; ++i) {
            const T px = lst[i][0], py = lst[i][1], pz = lst[i][2];   // LDS broadcast
#pragma unroll
            for (int u = 0; u < AQ_OUT; ++u) {
                const T dx = kx[u] - px, dy = ky[u] - py, dz = kz[u] - pz;
                const T d = aq_sqrt(dx * dx + dy * dy + dz * dz);     // as written in the reference: sqrt, then square
                const T w = (T)1 - d * d / (T)sigma;
                acc[u] += w > (T)0 ? w : (T)0;
            }
        }
        total += n_in;
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < AQ_OUT; ++u) {
        const int o = t + AQ_T * u;
        if (o < nout) {
            const int kn = o / na, an = o - kn * na;
            const size_t at = (((size_t)bn * ks + kn) * nc + pn) * na + an;
            wts[at] = acc[u];
            cnt[at] = (T)total;
        }
    }
}

}  // namespace

extern "C" int epn_initial_anchor_query_f32(const float *centers, const float *xyz, const float *kernel_points, int b,
                                            int nc, int m, int na, int ks, float radius, float sigma,
                                            float *anchor_weights, float *anchor_ctn, epn_stream_t stream) {
    if (b < 0 || nc < 0 || m < 0 || na < 1 || ks < 1 || !(sigma > 0.f)) return EPN_EINVAL;
    if ((long long)ks * na > (long long)AQ_T * AQ_OUT) return EPN_EINVAL;
    if (b == 0 || nc == 0) return 0;
    if (!centers || !kernel_points || !anchor_weights || !anchor_ctn || (m > 0 && !xyz)) return EPN_ENULL;
    EPN_LAUNCH(initial_anchor_query_kernel<float>, dim3((unsigned)nc, (unsigned)b), dim3(AQ_T), 0, epn_stream(stream),
                       centers, xyz, kernel_points, nc, m, na, ks, radius, sigma, anchor_weights, anchor_ctn);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_initial_anchor_query_f64(const double *centers, const double *xyz, const double *kernel_points, int b,
                                            int nc, int m, int na, int ks, float radius, float sigma,
                                            double *anchor_weights, double *anchor_ctn, epn_stream_t stream) {
    if (b < 0 || nc < 0 || m < 0 || na < 1 || ks < 1 || !(sigma > 0.f)) return EPN_EINVAL;
    if ((long long)ks * na > (long long)AQ_T * AQ_OUT) return EPN_EINVAL;
    if (b == 0 || nc == 0) return 0;
    if (!centers || !kernel_points || !anchor_weights || !anchor_ctn || (m > 0 && !xyz)) return EPN_ENULL;
    EPN_LAUNCH(initial_anchor_query_kernel<double>, dim3((unsigned)nc, (unsigned)b), dim3(AQ_T), 0, epn_stream(stream),
                       centers, xyz, kernel_points, nc, m, na, ks, radius, sigma, anchor_weights, anchor_ctn);
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---- fp64 dispatch of the index / gather extensions (AT_DISPATCH_FLOATING_TYPES in the reference: double callers)
extern "C" int epn_ball_query_f64(const double *new_xyz, const double *xyz, int b, int n, int m, float radius,
                                  int nsample, int32_t *idx, epn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || nsample < 1 || nsample > 4096) return EPN_EINVAL;
    if (b == 0 || m == 0) return 0;
    if (!new_xyz || !xyz || !idx) return EPN_ENULL;
    if (b > 65535) return EPN_EINVAL;
    const int waves = 4;
    dim3 grid(epn_cdiv(m, waves), b);
    EPN_LAUNCH(ball_query_kernel<double>, grid, dim3(64 * waves), waves * nsample * sizeof(int32_t),
                       epn_stream(stream), new_xyz, xyz, n, m, radius, nsample, idx);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_fps_f64(const double *xyz, int b, int n, int m, double *temp, int32_t *idx, epn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || m == 0) return 0;
    if (!xyz || !temp || !idx) return EPN_ENULL;
    const int block = opt_n_threads(n);
    EPN_LAUNCH(fps_temp_kernel<double>, dim3(b), dim3(block), 0, epn_stream(stream), xyz, n, m, block, temp, idx);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_fps_temp_f32(const float *xyz, int b, int n, int m, float *temp, int32_t *idx, epn_stream_t stream) {
    if (b < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || m == 0) return 0;
    if (!xyz || !temp || !idx) return EPN_ENULL;
    const int block = opt_n_threads(n);
    EPN_LAUNCH(fps_temp_kernel<float>, dim3(b), dim3(block), 0, epn_stream(stream), xyz, n, m, block, temp, idx);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_gather_fwd_f64(const double *points, const int32_t *idx, int b, int c, int n, int m, double *out,
                                  epn_stream_t stream) {
    if (b < 0 || c < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || c == 0 || m == 0) return 0;
    if (!points || !idx || !out) return EPN_ENULL;
    if (c > 65535 || b > 65535) return EPN_EINVAL;
    dim3 grid(epn_cdiv(m, 256), c, b);
    EPN_LAUNCH(gather_fwd_kernel<double>, grid, dim3(256), 0, epn_stream(stream), points, idx, c, n, m, out);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_gather_bwd_f64(const double *grad_out, const int32_t *idx, int b, int c, int n, int m,
                                  double *grad_points, epn_stream_t stream) {
    if (b < 0 || c < 0 || n < 1 || m < 0) return EPN_EINVAL;
    if (b == 0 || c == 0) return 0;
    if (!grad_points) return EPN_ENULL;
    EPN_HIP(hipMemsetAsync(grad_points, 0, sizeof(double) * (size_t)b * c * n, epn_stream(stream)));
    if (m == 0) return 0;
    if (!grad_out || !idx) return EPN_ENULL;
    if (c > 65535 || b > 65535) return EPN_EINVAL;
    dim3 grid(epn_cdiv(m, 256), c, b);
    EPN_LAUNCH(gather_bwd_kernel<double>, grid, dim3(256), 0, epn_stream(stream), grad_out, idx, c, n, m,
                       grad_points);
    EPN_CHECK_LAUNCH();
    return 0;
}
