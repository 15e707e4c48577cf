// Stand-alone ablation lab for the NT GEMM main loop (not part of the library): same staging / fragment scheme as
// csrc/gemm.hip with switches for what the loop does.  hipcc --offload-arch=gfx950 -O3 lab.hip -o lab && ./lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
__device__ __forceinline__ void glds16(const void *g, char *l) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// VAR bits: 1 = no global loads in the loop, 2 = no barrier in the loop, 4 = no ds_reads (MFMA only),
//           8 = stage issued right after the barrier (old order), 16 = setprio around MFMAs
template <int WGM, int WGN, int TM, int TN, int VAR>
__global__ __launch_bounds__(64 * WGM * WGN) void k(const float *A, const float *Bt, float *C, long long M, int N, int K) {
    constexpr int NW = WGM * WGN, BM = WGM * TM * 32, BN = WGN * TN * 32, ROWS = BM + BN, NG = ROWS / 8;
    constexpr int GPW = (NG + NW - 1) / NW;
    __shared__ __attribute__((aligned(1024))) char smem[2 * ROWS * 128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_n = (N + BN - 1) / BN;
    const long long m0 = (long long)(blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int nk = K / 32;
    const float *src[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW, r = 8 * g + (lane >> 3), slot = (lane & 7) ^ ((r >> 1) & 7);
        if (r < BM) { long long gr = m0 + r; gr = gr < M ? gr : M - 1; src[i] = A + gr * K + slot * 4; }
        else { int gn = n0 + r - BM; gn = gn < N ? gn : N - 1; src[i] = Bt + (long long)gn * K + slot * 4; }
    }
    f32x4 stg[GPW];
    int wofs[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + i * NW, r = 8 * g + (lane >> 3);
        wofs[i] = r * 128 + (lane & 7) * 16;     // same image as the DMA path (swizzle already in the source address)
    }
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int g = wave + i * NW;
            if (NG % NW == 0 || g < NG) { stg[i] = *reinterpret_cast<const f32x4 *>(src[i]); src[i] += 32; }
        }
    };
    auto lwrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int g = wave + i * NW;
            if (NG % NW == 0 || g < NG) *reinterpret_cast<f32x4 *>(smem + buf * (ROWS * 128) + wofs[i]) = stg[i];
        }
    };
    auto stage_one = [&](int buf, int i) {
        const int g = wave + i * NW;
        if (i < GPW && (NG % NW == 0 || g < NG)) { glds16(src[i], smem + buf * (ROWS * 128) + g * 1024); src[i] += 32; }
    };
    const int wpar = __builtin_amdgcn_readfirstlane((wave / 4) & 1);
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            const int g = wave + i * NW;
            if (NG % NW == 0 || g < NG) { glds16(src[i], smem + buf * (ROWS * 128) + g * 1024); src[i] += 32; }
        }
    };
    const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lj = lane >> 5, fsw = (li >> 1) & 7;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = ((wm * TM + i) * 32 + li) * 128;
#pragma unroll
    for (int i = 0; i < TN; ++i) boff[i] = (BM + (wn * TN + i) * 32 + li) * 128;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (VAR & 32) { gload(); lwrite(0); } else stage(0);
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = f32x4{1.f, 2.f, 3.f, (float)lane};
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i] = f32x4{1.f, 2.f, 3.f, (float)lane};
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (!(VAR & 2)) __syncthreads();
        const char *base = smem + (kt & 1) * (ROWS * 128);
        const bool more = kt + 1 < nk;
        if ((VAR & 8) && !(VAR & 1) && more) stage((kt + 1) & 1);
        if ((VAR & 32) && more) gload();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int so = ((2 * s + lj) ^ fsw) * 16;
            if (!(VAR & 4)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(base + aoff[i] + so);
#pragma unroll
                for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const f32x4 *>(base + boff[i] + so);
            }
            if (VAR & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
                if ((VAR & 64) && more) {       // one load per MFMA group, the two waves of a SIMD on alternating groups
                    const int grp = 4 * s + e;           // 0..15
                    __builtin_amdgcn_sched_barrier(0);
                    if (wpar == 0) { if ((grp & 1) == 1 && grp / 2 < GPW) stage_one((kt + 1) & 1, grp / 2); }
                    else { if ((grp & 1) == 0 && grp >= 2 && grp / 2 - 1 < GPW) stage_one((kt + 1) & 1, grp / 2 - 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(VAR & 64) && !(VAR & 8) && !(VAR & 1) && !(VAR & 32) && s == 0 && e == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) stage((kt + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (VAR & 16) __builtin_amdgcn_s_setprio(0);
        }
        if ((VAR & 32) && more) lwrite((kt + 1) & 1);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lj;
                if (m < M && n < N) C[m * N + n] = acc[i][j][r];
            }
        }
}

#include <map>
#include <string>
#include <algorithm>
static std::map<std::string, std::vector<float>> g_res;
template <int WGM, int WGN, int TM, int TN, int VAR>
void run(const char *name, const float *A, const float *B, float *C, long long M, int N, int K) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const unsigned grid = (unsigned)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((k<WGM, WGN, TM, TN, VAR>), dim3(grid), dim3(64 * WGM * WGN), 0, 0, A, B, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    g_res[name].push_back((float)(2.0 * M * N * K / ms / 1e9));
}

int main() {
    const long long M = 245760; const int N = 256, K = 3072;
    float *A, *B, *C;
    CK(hipMalloc(&A, M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, M * N * 4));
    std::vector<float> h((size_t)N * K);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (long long off = 0; off < M * K; off += (long long)h.size())
        CK(hipMemcpy(A + off, h.data(), std::min<size_t>(h.size(), M * K - off) * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 7; ++rep) {
        run<4, 2, 2, 2, 0>("256x128 8w glds behind first MFMAs", A, B, C, M, N, K);
        run<4, 2, 2, 2, 8>("256x128 8w glds right after barrier", A, B, C, M, N, K);
        run<4, 2, 2, 2, 64>("256x128 8w glds spread + staggered", A, B, C, M, N, K);
        run<4, 2, 2, 4, 64>("256x256 8w glds spread + staggered", A, B, C, M, N, K);
        run<4, 2, 2, 4, 0>("256x256 8w glds behind first MFMAs", A, B, C, M, N, K);
        run<4, 2, 2, 2, 1>("256x128 8w no global loads", A, B, C, M, N, K);
        run<4, 2, 2, 2, 7>("256x128 8w MFMA only", A, B, C, M, N, K);
    }
    for (auto &kv : g_res) {
        auto v = kv.second; std::sort(v.begin(), v.end());
        printf("%-44s median %6.1f  min %6.1f  max %6.1f TF\n", kv.first.c_str(), v[v.size() / 2], v.front(), v.back());
    }
    return 0;
}
