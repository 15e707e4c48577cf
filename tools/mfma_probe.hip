// Hardware probe (run on the GPU box): verifies the MFMA fragment layouts the fused kernels assume and
// times the f32 MFMA forms.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float *A, const float *B, float *D) {  // A[16][4], B[4][16] row-major -> D[16][16]
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((l >> 4) * 4 + i) * 16 + (l & 15)] = c[i];   // row=(l>>4)*4+i, col=l&15
}
__global__ void k32(const float *A, const float *B, float *D) {  // A[32][2], B[2][32] -> D[32][32]
    const int l = threadIdx.x;
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k4(float *D) {  // decode the 4x4x1_16b map: a = 1+lane, b = 100*(1+lane)
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), 100.f * (1 + l), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}
template <int MODE>
__global__ void timing(float *out, int iters, long long *cyc) {
    const int l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    f32x4 c4[4] = {};
    f32x16 c16[4] = {};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) c4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[j], 0, 0, 0);
            if (MODE == 1) c16[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16[j], 0, 0, 0);
            if (MODE == 2) c4[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4[j], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 4; ++j) { s += c4[j][0] + c16[j][0]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
    std::vector<float> A(64), B(64), D(1024), Dh(1024);
    for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7 + 3) % 11) - 5; B[i] = (float)((i * 5 + 1) % 13) - 6; }
    float *dA, *dB, *dD; long long *dc;
    CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 4096 * 64)); CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    // 16x16x4
    hipLaunchKernelGGL(k16, 1, 64, 0, 0, dA, dB, dD); CK(hipMemcpy(Dh.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float r = 0; for (int k = 0; k < 4; ++k) r += A[i * 4 + k] * B[k * 16 + j]; if (r != Dh[i * 16 + j]) ++bad; }
    printf("mfma_f32_16x16x4f32 layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)*4+i col=l&15): %s (%d bad)\n", bad ? "MISMATCH" : "OK", bad);
    hipLaunchKernelGGL(k32, 1, 64, 0, 0, dA, dB, dD); CK(hipMemcpy(Dh.data(), dD, 4096, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float r = 0; for (int k = 0; k < 2; ++k) r += A[i * 2 + k] * B[k * 32 + j]; if (r != Dh[i * 32 + j]) ++bad; }
    printf("mfma_f32_32x32x2f32 layout (A[l&31][l>>5], B[l>>5][l&31], D row=(r&3)+8(r>>2)+4(l>>5) col=l&31): %s (%d bad)\n", bad ? "MISMATCH" : "OK", bad);
    hipLaunchKernelGGL(k4, 1, 64, 0, 0, dD); CK(hipMemcpy(Dh.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("mfma_f32_4x4x1f32 decode: D[lane][reg] = a(la)*b(lb) -> (la,lb)\n");
    int hyp_ok = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        int v = (int)(Dh[l * 4 + i] / 100.f + 0.5f); int la = -1, lb = -1;
        for (int x = 1; x <= 64 && la < 0; ++x) if (v % x == 0 && v / x >= 1 && v / x <= 64) { /* ambiguous; test hypothesis instead */ }
        int ela = (l / 4) * 4 + i, elb = l;      // hypothesis: D[lane l][reg i] = A[lane 4*(l/4)+i] * B[lane l]
        if (v != (1 + ela) * (1 + elb)) hyp_ok = 0;
        (void)la; (void)lb;
    }
    printf("  hypothesis D[l][i] = A(block l/4,row i) * B(block l/4,col l%%4): %s\n", hyp_ok ? "OK" : "MISMATCH");
    if (!hyp_ok) for (int l = 0; l < 8; ++l) printf("  lane %d: %.0f %.0f %.0f %.0f\n", l, Dh[l*4], Dh[l*4+1], Dh[l*4+2], Dh[l*4+3]);
    // timing: 1 wave per SIMD (4 waves/CU), 256 CUs
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[3] = {"16x16x4f32", "32x32x2f32", "4x4x1f32(16 blocks)"};
    const double flops[3] = {2.0 * 16 * 16 * 4, 2.0 * 32 * 32 * 2, 2.0 * 16 * 4 * 4};
    for (int mode = 0; mode < 3; ++mode) for (int wpc = 4; wpc <= 8; wpc += 4) {
        float ms; long long cyc;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(timing<0>, 256 * wpc / 4, 256, 0, 0, dD, iters, dc);
            if (mode == 1) hipLaunchKernelGGL(timing<1>, 256 * wpc / 4, 256, 0, 0, dD, iters, dc);
            if (mode == 2) hipLaunchKernelGGL(timing<2>, 256 * wpc / 4, 256, 0, 0, dD, iters, dc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        double tf = flops[mode] * 4.0 * iters * (256.0 * wpc) / (ms * 1e-3) / 1e12;
        printf("timing %-20s waves/CU=%d: %.3f ms, %.1f TFLOP/s, %.1f clk/instr/wave\n", names[mode], wpc, ms, tf, (double)cyc / (4.0 * iters));
    }
    return 0;
}
