// Hardware probe (run on the GPU box): what does this part sustain in fp32 global atomics (L2 read-modify-write), as lane
// operations per second and as (instruction x 128-byte line) operations per second?  The transpose of the grouping
// (csrc/inter_mfma.hip: inter_ungroup_shared_kernel) has been priced "0.18 of HBM" for four rounds; round 4 showed that its time
// follows the number of (instruction x cache line) atomic operations at L2, not bytes -- this probe measures the roof of THAT
// quantity so the kernel can be priced against what bounds it ("the transpose of the grouping").
//
// Every wave issues `iters` atomic instructions into a `mb`-megabyte fp32 buffer; the lane -> address map is the variable:
//   lines2    64 consecutive floats at a random 256-byte-aligned offset            (2 lines per instruction: the kernel's shape,
//             one channel per lane, 16 CW = 64 channels of one (destination, anchor) row ... for CW = 4; CW = 2 -> lines1x2)
//   lines1x2  two runs of 32 consecutive floats at two random 128-byte-aligned offsets (2 lines, two destinations)
//   lines4    four runs of 16 floats (64 bytes each) at four random offsets           (the 16-channel chunk form: 4 destinations)
//   lines8    16-byte pieces: 16 runs of 4 floats                                     (round 4's four-channels-per-thread variant)
//   lines64   every lane its own line
//   same      all 64 lanes of all waves on ONE line (serialisation floor)
// Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_rate_probe.hip -o gpurun_out/atomic_rate_probe && gpurun_out/atomic_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *buf, unsigned lines, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned s = wave * 2654435761u + 12345u;         // wave-uniform stream of random line numbers
    for (int it = 0; it < iters; ++it) {
        size_t off;
        if (MODE == 0) off = (size_t)(lcg(s) % (lines / 2)) * 64 + lane;                                   // 2 lines, contiguous
        else if (MODE == 1) { unsigned a = lcg(s), b = lcg(s); off = (size_t)((lane < 32 ? a : b) % lines) * 32 + (lane & 31); }
        else if (MODE == 2) { unsigned r[4] = {lcg(s), lcg(s), lcg(s), lcg(s)}; off = (size_t)(r[lane >> 4] % (lines * 2)) * 16 + (lane & 15); }
        else if (MODE == 3) { unsigned h = s; s = s * 1664525u + 1013904223u; unsigned q = (h ^ ((lane >> 2) * 2246822519u)) * 3266489917u; off = (size_t)((q >> 8) % (lines * 8)) * 4 + (lane & 3); }
        else if (MODE == 4) { unsigned h = s; s = s * 1664525u + 1013904223u; unsigned q = (h ^ (lane * 2246822519u)) * 3266489917u; off = (size_t)((q >> 8) % lines) * 32 + (lane & 31); }
        else off = lane & 31;
        atomicAdd(buf + off, 1.0f);
    }
}

template <int MODE>
static void run(const char *name, int lines_per_instr, float *buf, size_t bytes, int iters) {
    const unsigned lines = (unsigned)(bytes / 128);
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lines, iters / 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lines, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters;
    printf("%-10s %8.1f MB  %8.3f ms  %8.2f G lane-atomics/s  %8.2f G instr/s  %8.2f G (instr x line)/s  %8.1f GB/s of operands\n", name,
           bytes / 1048576.0, ms, instr * 64 / ms / 1e6, instr / ms / 1e6, instr * lines_per_instr / ms / 1e6, instr * 256 / ms / 1e6);
}

int main() {
    for (size_t mb : {16, 128, 512}) {
        float *buf;
        const size_t bytes = mb << 20;
        hipMalloc(&buf, bytes);
        hipMemset(buf, 0, bytes);
        run<0>("lines2", 2, buf, bytes, 2000);
        run<1>("lines1x2", 2, buf, bytes, 2000);
        run<2>("lines4", 4, buf, bytes, 1000);
        run<3>("lines8", 16, buf, bytes, 500);
        run<4>("lines64", 64, buf, bytes, 200);
        if (mb == 16) run<5>("same", 1, buf, bytes, 200);
        hipFree(buf);
    }
    return 0;
}
