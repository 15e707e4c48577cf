// Hardware probe (run on the GPU box): issue cost, in shader cycles per wave instruction, of the instructions the fixed-point
// transpose of the grouping (csrc/inter_ungroup_cloud.hip) is made of -- is the double-precision adder that converts fp32 to
// 64-bit fixed point really full rate on this part?  One wave per SIMD (4 per workgroup), NI independent chains, so latency is
// hidden and the number is the issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int NI = 8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__global__ __launch_bounds__(256) void probe(int iters, long long *cyc, float *sink, float seed) {
    float a[NI];
    double d[NI];
    unsigned u[NI];
    f32x4 acc[NI];
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
#pragma unroll
    for (int k = 0; k < NI; ++k) { a[k] = seed + k + threadIdx.x; d[k] = a[k]; u[k] = (unsigned)a[k]; acc[k] = f32x4{a[k], 0.f, 0.f, 0.f}; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(seed));
            else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)seed));
            else if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
            else if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"((double)seed));
            else if (OP == 4) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) % NI]), "v"(u[(k + 2) % NI]));
            else if (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) % NI]), "v"(u[(k + 2) % NI]));
            else if (OP == 6) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[k], 0, 0, 0);
            else if (OP == 7) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], seed, acc[k], 0, 0, 0);
            else if (OP == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(a[k]), "v"(a[(k + 1) % NI]));
            else if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) % NI]));
            else asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[k]));
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int k = 0; k < NI; ++k) s += a[k] + (float)d[k] + (float)u[k] + acc[k][0];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    long long *dC;
    float *dS;
    hipMalloc(&dC, 64);
    hipMalloc(&dS, 4);
    const char *names[11] = {"v_add_f32", "v_add_f64", "v_cvt_f64_f32", "v_fma_f64", "v_max3_u32", "v_and_or_b32", "v_mfma_f32_16x16x32_bf16",
                             "v_mfma_f32_16x16x4_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_lshlrev_b64"};
    const int iters = 4000;
    printf("%-28s %12s\n", "instruction", "cycles/instr (one wave per SIMD, 8 independent chains)");
    for (int op = 0; op < 11; ++op) {
        auto launch = [&](int it) {
            switch (op) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 5: hipLaunchKernelGGL(probe<5>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 6: hipLaunchKernelGGL(probe<6>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 7: hipLaunchKernelGGL(probe<7>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 8: hipLaunchKernelGGL(probe<8>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                case 9: hipLaunchKernelGGL(probe<9>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
                default: hipLaunchKernelGGL(probe<10>, dim3(1), dim3(256), 0, 0, it, dC, dS, 1.5f); break;
            }
        };
        launch(100);
        launch(iters);
        hipDeviceSynchronize();
        long long h[4];
        hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost);
        const long long mx = std::max(std::max(h[0], h[1]), std::max(h[2], h[3]));
        printf("%-28s %12.2f\n", names[op], (double)mx / ((double)iters * NI));
    }
    return 0;
}
