// Hardware probe (run on the GPU box): LDS cycles per `ds_read_b64_tr_b16` for candidate images of the weight-gradient GEMMs'
// staged rows (csrc/gemm.hip: gemm_tn_bf16_kernel / gemm_tn_bf16_ring_kernel).  Round 5's PMC files show SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.75-0.88 for those kernels; the guides give the bank of a byte address ((a/4) mod 64) and the two
// 32-lane groups of this instruction, but also say its conflict classes are "hardware-transpose-specific" -- so measure.
//
// A wave reads what the GEMM reads: lane (li = lane & 15, lg = lane >> 4) supplies the address of the 8-byte chunk
// [row = 8 lg + (li >> 2) (+ 4 for the second read)][columns 4 (li & 3) .. + 3] of a 16-column tile; NF tiles per step.
// Every candidate is a function (row, 16-byte slot of the row) -> byte address; the kernel takes the 64 lane addresses of the
// first read, the byte offset of the second one (rows + 4) per lane, and the tile-to-tile offsets as tables.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_tr_probe.hip -o gpurun_out/lds_tr_probe && gpurun_out/lds_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

constexpr int NF = 8;             // fragments (16-column tiles) read per step, as a 64 x 128 wave tile does (TM + TN = 4 + 8 ... 12)
constexpr int LDS_BYTES = 96 * 1024;

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Pattern {
    int lo[64];                   // byte address of lane's first chunk, tile 0
    int hi[64];                   // ... of the chunk four rows further down
    int tile[64][NF];             // per lane: byte offset of tile f relative to tile 0 (swizzles make it lane-dependent)
    int tileh[64][NF];            // ... for the second read (a swizzle keyed on row bit 2 moves rows r and r + 4 differently)
};

__global__ __launch_bounds__(512) void probe(const Pattern *P, int iters, long long *cyc, int *sink) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) reinterpret_cast<int *>(smem)[i] = i;
    __syncthreads();
    unsigned lo[NF], hi[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        lo[f] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(smem + P->lo[lane] + P->tile[lane][f]);
        hi[f] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(smem + P->hi[lane] + P->tileh[lane][f]);
    }
    s16x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        s16x4 v[2 * NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * f]) : "v"(lo[f]));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * f + 1]) : "v"(hi[f]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < 2 * NF; ++f) {
            asm volatile("" : "+v"(v[f]));
            acc ^= v[f];
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc[0] == 12345 && acc[1] == 54321) sink[0] = acc[2];
}

// candidate images: byte address of (row r, 16-byte slot s of the row, byte b inside the slot); tile f, lane chunk c = li & 3 -> slot 2 f + c / 2
typedef std::function<int(int r, int s)> SlotAddr;

static Pattern make(const SlotAddr &at) {
    Pattern p;
    for (int l = 0; l < 64; ++l) {
        const int li = l & 15, lg = l >> 4;
        const int r = 8 * lg + (li >> 2), c = li & 3;
        const int a0 = at(r, c >> 1) + (c & 1) * 8;
        p.lo[l] = a0;
        p.hi[l] = at(r + 4, c >> 1) + (c & 1) * 8;
        for (int f = 0; f < NF; ++f) {
            p.tile[l][f] = at(r, 2 * f + (c >> 1)) + (c & 1) * 8 - a0;
            p.tileh[l][f] = at(r + 4, 2 * f + (c >> 1)) + (c & 1) * 8 - p.hi[l];
        }
    }
    return p;
}

int main() {
    struct Case { std::string name; SlotAddr at; };
    std::vector<Case> cases;
    auto rho = [](int r) { return (r & 3) | (((r >> 3) & 1) << 2); };       // the 8 rows one 32-lane group reads -> 0..7
    for (int pitch : {128, 256, 384, 512, 768, 1024}) {
        cases.push_back({"linear pitch " + std::to_string(pitch), [=](int r, int s) { return r * pitch + s * 16; }});
        cases.push_back({"pitch " + std::to_string(pitch) + " slot ^ 2 rho(row)", [=](int r, int s) { return r * pitch + ((s ^ (2 * rho(r))) * 16); }});
        cases.push_back({"pitch " + std::to_string(pitch) + " slot ^ 2 (row & 7)", [=](int r, int s) { return r * pitch + ((s ^ (2 * (r & 7))) * 16); }});
        cases.push_back({"pitch " + std::to_string(pitch) + " slot ^ (row & 7)", [=](int r, int s) { return r * pitch + ((s ^ (r & 7)) * 16); }});
        for (int pad : {16, 32, 64})
            cases.push_back({"pitch " + std::to_string(pitch) + " + " + std::to_string(pad), [=](int r, int s) { return r * (pitch + pad) + s * 16; }});
    }
    // the guide's conflict-free image of one [32 rows][16 columns] subtile: row pitch 32 bytes, tiles 1 KiB apart
    cases.push_back({"subtile-major [tile][32 rows][32 B]", [](int r, int s) { return (s >> 1) * 1024 + r * 32 + (s & 1) * 16; }});
    cases.push_back({"subtile-major, rows 4-7 <-> 8-11 swapped", [](int r, int s) {
        const int rr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        return (s >> 1) * 1024 + rr * 32 + (s & 1) * 16; }});

    Pattern *dP;
    long long *dC;
    int *dS;
    hipMalloc(&dP, sizeof(Pattern));
    hipMalloc(&dC, 8 * 256 * sizeof(long long));
    hipMalloc(&dS, 4);
    const int iters = 4000;
    printf("%-46s %14s %14s %14s\n", "image", "1 wave", "4 waves/CU", "8 waves/CU");
    printf("%-46s %14s %14s %14s\n", "", "cyc/instr", "LDS cyc/instr", "LDS cyc/instr");
    for (auto &c : cases) {
        Pattern p = make(c.at);
        int maxa = 0;
        for (int l = 0; l < 64; ++l)
            for (int f = 0; f < NF; ++f) {
                maxa = std::max(maxa, p.lo[l] + p.tile[l][f] + 8);
                maxa = std::max(maxa, p.hi[l] + p.tileh[l][f] + 8);
            }
        if (maxa > LDS_BYTES) { printf("%-46s (does not fit)\n", c.name.c_str()); continue; }
        hipMemcpy(dP, &p, sizeof(p), hipMemcpyHostToDevice);
        double res[3];
        int k = 0;
        for (int waves : {1, 4, 8}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 0, 0, dP, 200, dC, dS);      // warm
            hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 0, 0, dP, iters, dC, dS);
            hipDeviceSynchronize();
            std::vector<long long> h(waves);
            hipMemcpy(h.data(), dC, waves * sizeof(long long), hipMemcpyDeviceToHost);
            long long mx = 0;
            for (auto v : h) mx = std::max(mx, v);
            // all `waves` share one CU's LDS: array cycles per instruction = elapsed / (instructions of ALL waves)
            res[k++] = (double)mx / ((double)iters * 2 * NF * (waves == 1 ? 1 : waves));
        }
        printf("%-46s %14.2f %14.2f %14.2f\n", c.name.c_str(), res[0], res[1], res[2]);
    }
    printf("(clock64 ticks at the constant 100 MHz counter are scaled by the runtime's wall_clock rate? see below)\n");
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall_clock_rate %d kHz, shader clock %d kHz: multiply the columns by %.2f for shader cycles if clock64 counts the wall clock\n",
           rate, clk, rate ? (double)clk / rate : 0.0);
    return 0;
}
