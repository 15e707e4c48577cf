// Hardware probe (run on the GPU box): what does a CU sustain in LDS floating-point atomics (`ds_add_f32`, no return)?
// Round 2 measured "~1 lane per clock" inside the scatter kernel and the transpose of the grouping has used plain stores + a
// gather-sum ever since (csrc/inter_mfma.hip: inter_ungroup_shared_kernel).  A transpose that keeps a whole cloud's gradient
// rows in LDS (no global atomics at all) would stand or fall with this rate, so measure it in isolation, per address pattern:
//   linear      lane i -> word i                              (64 distinct banks)
//   rows4       lane (x = lane & 15, j = lane >> 4) -> row[j] * pitch + x, four random rows   (the per-slot store's shape)
//   same        all lanes one word
//   pairs       lanes 2 i, 2 i + 1 -> the same word           (2-way same-address collisions)
// against `ds_write_b32` and `ds_add_u32` on the same addresses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o gpurun_out/lds_atomic_probe && gpurun_out/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int LDS_WORDS = 16384;   // 64 KB
constexpr int NI = 8;              // independent instructions per iteration

template <int OP>
__global__ __launch_bounds__(1024) void probe(const int *addr, int iters, long long *cyc, float *sink) {
    __shared__ float smem[LDS_WORDS];
    for (int i = threadIdx.x; i < LDS_WORDS; i += blockDim.x) smem[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned a[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int w = (addr[lane * NI + k] + wave * 1031) % LDS_WORDS;
        a[k] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(smem + w);
    }
    unsigned a2[NI];                   // 64-bit forms: the same word index, 8-byte elements (half the table)
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int w = (addr[lane * NI + k] + wave * 1031) % (LDS_WORDS / 2);
        a2[k] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(smem + 2 * w);
    }
    const float v = 1.0f;
    const unsigned vu = 1u;
    const unsigned long long v64 = 0x100000001ull;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (OP == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(a[k]), "v"(v) : "memory");
            else if (OP == 1) asm volatile("ds_write_b32 %0, %1" ::"v"(a[k]), "v"(v) : "memory");
            else if (OP == 2) asm volatile("ds_add_u32 %0, %1" ::"v"(a[k]), "v"(vu) : "memory");
            else if (OP == 3) asm volatile("ds_pk_add_f16 %0, %1" ::"v"(a[k]), "v"(vu) : "memory");
            else if (OP == 4) asm volatile("ds_add_u64 %0, %1" ::"v"(a2[k]), "v"(v64) : "memory");
            else asm volatile("ds_write_b64 %0, %1" ::"v"(a2[k]), "v"(v64) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[wave] = t1 - t0;
    __syncthreads();
    if (smem[threadIdx.x] == 12345.678f) sink[0] = 1.0f;
}

int main() {
    struct Case { const char *name; std::vector<int> addr; };
    std::vector<Case> cases;
    auto mk = [&](const char *name, auto f) {
        Case c{name, std::vector<int>(64 * NI)};
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < NI; ++k) c.addr[l * NI + k] = f(l, k);
        cases.push_back(c);
    };
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)(s >> 10); };
    mk("linear", [](int l, int k) { return k * 64 + l; });
    for (int pitch : {16, 20, 32, 36, 64, 68}) {
        static char names[8][32];
        static int ni = 0;
        snprintf(names[ni], 32, "rows4 pitch %d", pitch);
        int rows[NI][4];
        for (int k = 0; k < NI; ++k)
            for (int j = 0; j < 4; ++j) rows[k][j] = rnd() % (LDS_WORDS / 68 - 1);
        Case c{names[ni++], std::vector<int>(64 * NI)};
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < NI; ++k) c.addr[l * NI + k] = rows[k][l >> 4] * pitch + (l & 15);
        cases.push_back(c);
    }
    mk("pairs", [](int l, int k) { return k * 64 + (l >> 1); });
    mk("quads", [](int l, int k) { return k * 64 + (l >> 2); });
    mk("same", [](int l, int k) { return k; });

    int *dA;
    long long *dC;
    float *dS;
    hipMalloc(&dA, 64 * NI * sizeof(int));
    hipMalloc(&dC, 16 * sizeof(long long));
    hipMalloc(&dS, 4);
    int rate = 0, clk = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double scale = 1.0;   // clock64() already counts shader cycles here (ds_write_b32 comes out at 4.1 per instruction)
    const int iters = 1000;
    const char *ops[6] = {"ds_add_f32", "ds_write_b32", "ds_add_u32", "ds_pk_add_f16", "ds_add_u64", "ds_write_b64"};
    printf("shader cycles per wave instruction at the CU's LDS (elapsed / instructions of all waves); wall clock rate %d kHz, shader %d kHz\n", rate, clk);
    printf("%-18s %-14s %10s %10s %10s %10s\n", "pattern", "op", "1 wave", "4 waves", "8 waves", "16 waves");
    for (auto &c : cases) {
        hipMemcpy(dA, c.addr.data(), c.addr.size() * sizeof(int), hipMemcpyHostToDevice);
        for (int op = 0; op < 6; ++op) {
            double res[4];
            int k = 0;
            for (int waves : {1, 4, 8, 16}) {
                auto launch = [&](int it) {
                    if (op == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                    else if (op == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                    else if (op == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                    else if (op == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                    else if (op == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                    else hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64 * waves), 0, 0, dA, it, dC, dS);
                };
                launch(100);
                launch(iters);
                hipDeviceSynchronize();
                std::vector<long long> h(waves);
                hipMemcpy(h.data(), dC, waves * sizeof(long long), hipMemcpyDeviceToHost);
                long long mx = 0;
                for (auto v : h) mx = std::max(mx, v);
                res[k++] = (double)mx * scale / ((double)iters * NI * waves);
            }
            printf("%-18s %-14s %10.2f %10.2f %10.2f %10.2f\n", c.name, ops[op], res[0], res[1], res[2], res[3]);
        }
    }
    return 0;
}
