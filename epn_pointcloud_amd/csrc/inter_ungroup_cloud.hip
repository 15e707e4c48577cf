// Transpose of the grouping with a whole cloud's gradient rows RESIDENT IN LDS: no global atomics, no zero fill, no conversion
// pass, bitwise repeatable (round 6).
//
//   dF[b, idx[b,p,n], a, c] = sum over (p, n) of  mul[p,n] * sum_k w[b,p,a,k,n] * dG[(b,p,a)][c*ks + k]
//
// replaces  autograd's backward of the gather in inter_zpconv_grouping_naive (vgtk/vgtk/spconv/functional.py:372-390: index_add of
//           the grouped-feature gradient into the input features), as inter_ungroup_shared_kernel (csrc/inter_mfma.hip) does.
//
// Why a second form.  inter_ungroup_shared_kernel pre-reduces the scatter over 8-16 output points in LDS and then issues fp32
// atomics to HBM; tools/atomic_rate_probe.hip + tools/ungroup_atomic_pricing.py (round 6) price it: this part retires one
// 64-byte atomic segment per clock and XCD (1.34 TB/s of atomic operands at best) and the K = 64 instances run at 0.80-0.86 of
// exactly that roof (0.18 of HBM).  More pre-reduction is the only way down, and the limit of pre-reduction is the whole cloud:
// a workgroup that owns ALL p2 output points of one (cloud, anchor) owns every destination row dF[b, :, a, c0 .. c0 + CR) outright
// -- p1 x CR accumulators (p1 <= 1024 input points, CR = 16-128 channels: 64-132 KB) -- and writes them once, as plain stores.
//
// What makes it possible: tools/lds_atomic_probe.hip (profiles/r06_lds_atomic_probe.txt).  `ds_add_f32` retires one wave
// instruction per 192 cycles on gfx950 (3 cycles per LANE: round 2's "2x slower than global atomics"), but the INTEGER forms run
// at the LDS store rate: `ds_add_u32` 4.1 cycles, `ds_add_u64` 6.2 cycles per wave instruction -- 31 x faster.  So the
// accumulators are 64-bit fixed point:
//   * unit = 2^-s with s chosen per call from a device scalar max|dG| and the largest slot multiplicity of the index table so
//     that every single contribution is below 2^50 units; a destination sums at most p2 <= 4096 contributions: below 2^62.
//   * fp32 -> fixed point in THREE instructions through the double-precision adder (full rate on this part):
//     bits(double(x) + 1.5 * 2^52) - bits(1.5 * 2^52), where the second operand has a zero low word -- v_cvt_f64_f32,
//     v_add_f64, v_add_u32 on the high word.  Round to nearest even, symmetric in sign.
//   * integer addition is associative: the result does not depend on the order in which waves arrive.  The sum of the
//     contributions (each rounded once to the unit, <= 2^-43 max|dG| for multiplicity 1-4) is exact, then rounded once to fp32 /
//     bf16 -- where fp32 atomics round after every addition in an order that changes from run to run.
//   * a contribution beyond 2^50 units can only come from an understated maximum: counted in a sticky device counter
//     (epn_inter_ungroup_cloud_range_count), as the f16x2 sentinel does.
//
// Structure: grid (b * na, cin / CR).  Waves are independent (no barrier between zeroing the accumulators and the write-out):
// wave w takes output points w, w + NW, ...; per point it rebuilds the neighbourhood fragments from a per-call slot table (row
// index + multiplicity per neighbour slot, cyclic ball-query padding already folded: uc_slots_kernel), regenerates the
// kernel-influence weights of its anchor by S-MFMA (as the grouping kernels do), and per 16-channel chunk contracts them with
// the dG fragment on the matrix pipe (fp32: v_mfma_f32_16x16x4_f32, bf16: v_mfma_f32_16x16x32_bf16) and adds the 4 NT values of
// every lane to the accumulator rows of their destinations.
#include "inter_device.h"
#include "gemm.h"

namespace epn {
namespace {

__device__ unsigned g_uc_range = 0u;       // waves that saw a contribution beyond what the reported max|dG| allows

constexpr int UC_DUMMY = 4;                // accumulator rows behind the cloud's p1 rows: slots that name no input point, one per lane group
constexpr int UC_LDS_MAX = 160 * 1024;     // LDS of one workgroup: the whole CU

// Per-call tables of the neighbourhoods, one wave per output point (everything about a point that does not depend on the
// anchor -- the main kernel visits every point once per ANCHOR, so it only loads):
//   off[(b * p2 + p) * EW + n]      byte offset of slot n's accumulator row (row * rowb): the input point, or p1 + (n / 4) % 4 for a
//                                   slot that names none (beyond nn, shadow / negative index)
//   hood[((b * p2 + p) * 5 + c) * EW + n]             for neighbour n (= 16 t + x), times its multiplicity m:
//                                   c = 0..2: (xyz[idx[n]] - centre) m;  c = 3: m;  c = 4: alpha m  (alpha = 1 - |g|^2 / sigma)
//                                   -- the S-MFMA operand of lane (x, j) is entry c = j, its accumulator start entry c = 4
// Multiplicity as load_hood (inter_device.h): the ball query pads a row that found cnt < nn neighbours by repeating them
// cyclically (vgtk/vgtk/cuda/grouping_cuda_kernel.cu:100-104); the first occurrence carries the number of slots holding that
// point, the repeats (and slots without a point) 0 -- their weights come out as relu(0) = 0.  mulmax: largest multiplicity.
constexpr int UC_PTS = 8;                  // output points per wave of the table kernel
__global__ __launch_bounds__(256) void uc_slots_kernel(const int32_t *__restrict__ idx, const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz, long long npts, int p1, int p2, int nn, int ew,
                                                       float sigma_inv, unsigned rowb, uint32_t *__restrict__ off,
                                                       float *__restrict__ hood, unsigned *__restrict__ mulmax) {
    // UC_PTS points per wave, their loads issued together: one point per wave was a chain of three dependent global round trips
    // (index row, the row again at lane - cnt, coordinates) per 1.5 KB of table -- 85-91 us per call, 0.5 ms per rotation step
    const int lane = threadIdx.x & 63;
    const long long pt0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * UC_PTS;
    if (pt0 >= npts) return;                                  // wave-uniform
    const bool in = lane < nn;
    int q[UC_PTS];
#pragma unroll
    for (int i = 0; i < UC_PTS; ++i) {
        const long long pt = pt0 + i < npts ? pt0 + i : npts - 1;      // (a ragged tail repeats the last point: same values, same addresses)
        q[i] = in ? idx[pt * nn + lane] : -1;
    }
    bool valid[UC_PTS];
    unsigned mul[UC_PTS];
    float g[UC_PTS][3], cc[UC_PTS][3];
#pragma unroll
    for (int i = 0; i < UC_PTS; ++i) {
        const long long pt = pt0 + i < npts ? pt0 + i : npts - 1;
        const long long bb = pt / p2;
        const int pp = (int)(pt - bb * p2);
        const int first = __shfl(q[i], 0, 64);
        const unsigned long long rep = __ballot(in && lane > 0 && q[i] == first);
        int cnt = rep ? (int)__builtin_ctzll(rep) : nn;
        // only a genuinely cyclic row is de-duplicated (index tensors handed in by the caller may be arbitrary)
        const int qprev = __shfl(q[i], lane >= cnt ? lane - cnt : lane, 64);
        const bool bad = in && lane >= cnt && q[i] != qprev;
        if (__ballot(bad) != 0ull) cnt = nn;
        valid[i] = q[i] >= 0 && q[i] < p1;
        mul[i] = (valid[i] && lane < cnt) ? (unsigned)((nn - 1 - lane) / cnt + 1) : 0u;
        const float *s = xyz + bb * 3 * p1, *c = new_xyz + bb * 3 * p2;
        const int qq = valid[i] ? q[i] : 0;
        g[i][0] = s[qq]; g[i][1] = s[p1 + qq]; g[i][2] = s[2 * p1 + qq];
        cc[i][0] = c[pp]; cc[i][1] = c[p2 + pp]; cc[i][2] = c[2 * p2 + pp];
    }
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < UC_PTS; ++i) {
        const long long pt = pt0 + i < npts ? pt0 + i : npts - 1;
        if (lane < ew) {
            off[pt * ew + lane] = (valid[i] ? (unsigned)q[i] : (unsigned)(p1 + ((lane >> 2) & 3))) * rowb;
            const float gx = g[i][0] - cc[i][0], gy = g[i][1] - cc[i][1], gz = g[i][2] - cc[i][2];
            const float alpha = 1.0f - (gx * gx + gy * gy + gz * gz) * sigma_inv;
            const float mf = (float)mul[i];
            float *h = hood + pt * 5 * ew + lane;                  // [point][entry c][slot]: every store instruction one contiguous run
            h[0] = gx * mf; h[ew] = gy * mf; h[2 * ew] = gz * mf; h[3 * ew] = mf; h[4 * ew] = alpha * mf;
        }
        m = mul[i] > m ? mul[i] : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    // (multiplicity 1 is what uc_scale assumes for a zero slot: recording it made every wave of the launch queue up on one address)
    if (lane == 0 && m > 1u && m > __atomic_load_n(mulmax, __ATOMIC_RELAXED)) atomicMax(mulmax, m);
}

// max|dG| for callers that do not have it (a pass over dG: the GEMM that writes dG can supply it for free, gemm.h c_amax).
// Non-finite elements are left out; *out must be zero before the launch.
template <typename TG>
__global__ __launch_bounds__(256) void uc_absmax_kernel(const TG *__restrict__ src, long long n, unsigned *__restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned a;
        if constexpr (sizeof(TG) == 2) a = ((unsigned)__builtin_bit_cast(unsigned short, src[i]) & 0x7fffu) << 16;
        else a = __builtin_bit_cast(unsigned, src[i]) & 0x7fffffffu;
        m = (a > m && a < 0x7f800000u) ? a : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}

struct UcArgs {
    InterArgs A;               // gout = dG [ncol][cin*ks] (TG), out = dF (fp32 or bf16), rk4 = the rotated-kernel table
    const uint32_t *tab;       // accumulator-row offsets [b][p2][16 NT]
    const float *hood;         // neighbourhood table [b][p2][5][16 NT]
    const unsigned *mulmax;
    const float *dg_amax;      // device scalar max|dG|
    const void *add;           // optional tensor of dF's shape and type added to the result (NULL: none)
    int cr;                    // channels per workgroup (multiple of 16)
    int out_bf16;
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// 2^s with s = 50 - ceil(log2(mulmax * ks)) - (exponent of amax + 1): every contribution mul * sum_k w dG (0 <= w <= 1) is below 2^50 units
__device__ __forceinline__ float uc_scale(float amax, unsigned mulmax, int ks) {
    unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 255u;      // amax < 2^(e - 126)
    e = e < 1u ? 1u : e;
    const unsigned f = (mulmax < 1u ? 1u : mulmax) * (unsigned)ks;
    const int bits = f > 1u ? 32 - __builtin_clz(f - 1u) : 0;            // ceil(log2 f)
    int se = 127 + 50 - bits - ((int)e - 126);
    se = se < 1 ? 1 : (se > 240 ? 240 : se);      // (<= 2^113: multiplicity x scale stays finite; an all-zero dG has e = 1)
    return __builtin_bit_cast(float, (unsigned)se << 23);
}

template <int NT, int KT, typename TG, int NWV, int NL>
__global__ __launch_bounds__(64 * NWV) void inter_ungroup_cloud_kernel(UcArgs P) {
    // LDS: [accumulators (p1 + UC_DUMMY) rows x CR x 8 bytes | 16 bytes: poison flag]
    extern __shared__ __attribute__((aligned(16))) char uc_smem[];
    const InterArgs &A = P.A;
    constexpr int EW = 16 * NT;
    constexpr int NTH = 64 * NWV;
    constexpr int CR = 16 * NL;                 // channels of the workgroup
    constexpr unsigned ROWB = CR * 8u;          // bytes of an accumulator row
    typedef typename std::conditional<sizeof(TG) == 2, bf16x4_t, f32x4>::type frag_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int blk = epn_xcd_tile(blockIdx.x, gridDim.x);
    const int bb = blk / A.na, a = blk - bb * A.na;
    const int c0 = blockIdx.y * CR;
    const int rows = A.p1 + UC_DUMMY;
    int *poison = reinterpret_cast<int *>(uc_smem + (size_t)rows * ROWB);

    {   // zero the accumulators
        const u32x4_t z = {0u, 0u, 0u, 0u};
        u32x4_t *acc4 = reinterpret_cast<u32x4_t *>(uc_smem);
        for (int i = tid; i < rows * (CR / 2); i += NTH) acc4[i] = z;
        if (tid == 0) *poison = 0;
    }
    const float S = uc_scale(*P.dg_amax, *P.mulmax, A.ks);
    const float invS = __builtin_bit_cast(float, (254u - (__builtin_bit_cast(unsigned, S) >> 23)) << 23);
    float rk[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) rk[kt] = A.rk4[((size_t)a * EPN_KS_MAX + 16 * kt + x) * 4 + j];
    const int gss = A.cin * A.ks;
    const uint32_t *tabc = P.tab + (size_t)bb * A.p2 * EW + 4 * j;
    const float *hoodc = P.hood + (size_t)bb * A.p2 * 5 * EW + j * EW + x;    // lane (x, j) -> entry c = j of neighbour 16 t + x
    // dG fragment of (point p, chunk cw): lane (x = channel, j) <- dG[(b, p, a)][(c0 + 16 cw + x) * ks + 16 kt + 4 j .. + 3]
    const TG *dGc = reinterpret_cast<const TG *>(A.gout) + ((size_t)bb * A.p2 * A.na + a) * gss + (size_t)(c0 + x) * A.ks;
    int koff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) koff[kt] = 16 * kt + 4 * (16 * kt + 4 * j < A.ks ? j : 0);
    auto load_dg = [&](int p, int cw, frag_t (&d)[KT]) {
        const TG *src = dGc + (size_t)p * A.na * gss + (size_t)(16 * cw) * A.ks;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if constexpr (sizeof(TG) == 2) d[kt] = *reinterpret_cast<const bf16x4_t *>(src + koff[kt]);
            else d[kt] = ld4f(src + koff[kt]);
        }
    };
    auto load_tab = [&](int p, float (&hb)[NT], float (&ha)[NT], u32x4_t (&e4)[NT]) {
        const uint32_t *t0 = tabc + (size_t)p * EW;
        const float *h0 = hoodc + (size_t)p * 5 * EW;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            hb[t] = h0[16 * t];                           // S-MFMA operand: (g_x, g_y, g_z, 1)[j] m of neighbour 16 t + x
            ha[t] = h0[16 * t + (4 - j) * EW];            // its accumulator start: alpha m of neighbour x
            e4[t] = *reinterpret_cast<const u32x4_t *>(t0 + 16 * t);
        }
    };
    __syncthreads();                                            // accumulators are zero

    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned x8b = (unsigned)(uintptr_t)(lds_char *)uc_smem + 8u * (unsigned)x;
    int chkp = 0;                 // range check, as bit patterns: largest positive contribution of the lane (signed order; NaN / inf on top) ...
    unsigned chkn = 0u;           // ... and the most negative one (unsigned order puts the sign bit on top)
    if (wave < A.p2) {
        float hbn[NT], han[NT];
        u32x4_t e4n[NT];
        // dG fragments are requested D chunks ahead of their use (a ring over the flattened (point, chunk) sequence): with four
        // waves per SIMD -- all a 1024-thread workgroup can have -- a fragment requested one chunk (~500 cycles) ahead arrived
        // an HBM round trip too late, and every wave waited it out once per chunk (v4: 0.90 ms on the 32-channel K = 64 layer)
        constexpr int D = sizeof(TG) == 2 ? NL : (NL < 4 ? NL : 4);
        frag_t ring[D][KT];
        load_tab(wave, hbn, han, e4n);
#pragma unroll
        for (int i = 0; i < D; ++i) load_dg(wave, i, ring[i]);
        for (int p = wave; p < A.p2; p += NWV) {
            // ---- neighbourhood fragments of the point: loads only (uc_slots_kernel), times the scale of the accumulators (both
            // operands of the S-MFMA scale with it: relu commutes with a non-negative factor)
            float gB[NT], alphaN[NT];
            unsigned rowoff[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                gB[t] = hbn[t] * S;
                alphaN[t] = han[t] * S;
#pragma unroll
                for (int r = 0; r < 4; ++r) rowoff[t][r] = e4n[t][r] + x8b;       // LDS address of (row, this lane's column)
            }
            const int pn = p + NWV < A.p2 ? p + NWV : p;       // (the last point re-reads its own entries: cache hits, unused)
            load_tab(pn, hbn, han, e4n);
            // ---- kernel-influence weights of (point, anchor), times multiplicity and scale of their neighbour (folded into the
            // S-MFMA's operands): lane (x, j), register r -> w[k = 16 kt + 4 j + r][n = 16 t + x] -- the A operand of the
            // contraction (row = neighbour x)
            frag_t wgt[NT][KT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4 sk = {alphaN[t], alphaN[t], alphaN[t], alphaN[t]};
                    sk = mfma4(rk[kt], gB[t], sk);
                    if constexpr (sizeof(TG) == 2) {
                        wgt[t][kt] = relu_pack4(sk);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sk[r] = relu_f(sk[r]);
                        wgt[t][kt] = sk;
                    }
                }
            }
#pragma unroll
            for (int cw = 0; cw < NL; ++cw) {
                frag_t dgc[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) dgc[kt] = ring[cw % D][kt];
                if (cw + D < NL) load_dg(p, cw + D, ring[cw % D]);
                else load_dg(pn, cw + D - NL, ring[cw % D]);
                f32x4 tt[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    tt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (sizeof(TG) == 2) {
                        if constexpr (KT == 2) tt[t] = mfma_bf16_k32(wgt[t][0], wgt[t][1], dgc[0], dgc[1], tt[t]);
                        else tt[t] = mfma_bf16_k16(wgt[t][0], dgc[0], tt[t]);
                    } else {
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) tt[t] = mfma4(wgt[t][kt][r], dgc[kt][r], tt[t]);
                    }
                }
                // tt[t]: lane (x = channel, j), register r -> slot n = 16 t + 4 j + r, in accumulator units
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = tt[t][r];
                        const unsigned vb = __builtin_bit_cast(unsigned, v);
                        chkp = (int)vb > chkp ? (int)vb : chkp;
                        chkn = vb > chkn ? vb : chkn;
                        const double dv = (double)v + 6755399441055744.0;                    // 1.5 * 2^52
                        u32x2_t w = __builtin_bit_cast(u32x2_t, dv);
                        w[1] -= 0x43380000u;                                                // bits(1.5 * 2^52): low word zero
                        // (the address as an integer: through a pointer into uc_smem hipcc re-adds the array's -- link-time -- base to
                        // every one of them, two instructions per value and chunk)
                        lds_u64 *dst = (lds_u64 *)(uintptr_t)(rowoff[t][r] + (unsigned)cw * 128u);
                        __hip_atomic_fetch_add((unsigned long long *)dst, __builtin_bit_cast(unsigned long long, w), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            }
        }
    }
    // a contribution at or beyond 2^50 units: the reported maximum was too small (or dG is not finite).  Counted, and the
    // workgroup's rows leave as NaN -- a wrapped integer sum would otherwise pass for a gradient
    if (__builtin_amdgcn_ballot_w64(chkp >= 0x58800000 || chkn >= 0xd8800000u) != 0ull && lane == 0) {    // 2^50
        atomicAdd(&g_uc_range, 1u);
        *poison = 1;
    }
    __syncthreads();
    const float bad = *poison ? __builtin_nanf("") : 0.0f;

    // ---- write-out: dF[b, q, a, c0 .. c0 + CR) = accumulators / 2^s (+ add), one pass of plain stores
    const long long *acc = reinterpret_cast<const long long *>(uc_smem);
    const size_t obase = ((size_t)bb * A.p1 * A.na + a) * A.cin + c0;
    const size_t ostride = (size_t)A.na * A.cin;
    constexpr int CR2 = CR / 2;
    for (int i = tid; i < A.p1 * CR2; i += NTH) {
        const int q = i / CR2, c = 2 * (i - q * CR2);
        const long long v0 = acc[q * CR + c], v1 = acc[q * CR + c + 1];
        float f0 = (float)v0 * invS + bad, f1 = (float)v1 * invS + bad;
        const size_t o = obase + (size_t)q * ostride + c;
        if (P.out_bf16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            if (P.add) {
                const bf16x2_t ad = *reinterpret_cast<const bf16x2_t *>(static_cast<const __bf16 *>(P.add) + o);
                f0 += (float)ad[0]; f1 += (float)ad[1];
            }
            *reinterpret_cast<bf16x2_t *>(reinterpret_cast<__bf16 *>(A.out) + o) = bf16x2_t{(__bf16)f0, (__bf16)f1};
        } else {
            if (P.add) {
                const float2 ad = *reinterpret_cast<const float2 *>(static_cast<const float *>(P.add) + o);
                f0 += ad.x; f1 += ad.y;
            }
            *reinterpret_cast<float2 *>(A.out + o) = make_float2(f0, f1);
        }
    }
}

// LDS of one workgroup at cr channels: accumulators + flag
size_t uc_lds_bytes(const epn_inter_desc *d, int cr) { return (size_t)(d->p1 + UC_DUMMY) * cr * 8 + 16; }

int uc_channels_per_wg(const epn_inter_desc *d, int bf16) {     // 16, 32, 64 or 128: the largest that divides cin and fits
    int best = 0;
    const int cap = d->nn > 32 ? (bf16 ? 64 : 16) : ((!bf16 && d->nn > 16) ? 64 : 128);   // (the wider instances spill under the 128-register cap of 16 waves)
    for (int cr = 16; cr <= cap && cr <= d->cin; cr *= 2)
        if (d->cin % cr == 0 && uc_lds_bytes(d, cr) <= (size_t)UC_LDS_MAX) best = cr;
    return best;
}

template <typename K>
int uc_set_lds(K kern, size_t bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

long long ungroup_cloud_range_take(bool reset) {
    unsigned v = 0;
    hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_uc_range), sizeof(v), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return -(long long)e;
    if (reset && v) {
        const unsigned zero = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_uc_range), &zero, sizeof(zero), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return -(long long)e;
    }
    return (long long)v;
}

bool inter_ungroup_cloud_ok(const epn_inter_desc *d) {
    if (!inter_group_mfma_ok(d)) return false;
    return d->na >= 1 && d->nn <= 64 && d->ks <= 32 && d->p1 + UC_DUMMY <= 65536 && d->p2 <= 4096 && d->cin % 16 == 0 &&
           uc_channels_per_wg(d, 0) > 0 && (long long)d->b * d->p2 * d->na * d->cin * d->ks < (1LL << 40);
}

// workspace behind the rotated-kernel table: offsets [b][p2][16 nt] + neighbourhood table [b][p2][nt][80] + 256 bytes (scalars)
static size_t uc_off_bytes(const epn_inter_desc *d) {
    const int nt = d->nn <= 16 ? 1 : (d->nn <= 32 ? 2 : 4);
    return ((size_t)d->b * d->p2 * 16 * nt * 4 + 255) & ~(size_t)255;
}
size_t inter_ungroup_cloud_extra_bytes(const epn_inter_desc *d) {
    const int nt = d->nn <= 16 ? 1 : (d->nn <= 32 ? 2 : 4);
    return uc_off_bytes(d) + (((size_t)d->b * d->p2 * nt * 80 * 4 + 255) & ~(size_t)255) + 256;
}

int launch_inter_ungroup_cloud(const epn_inter_desc *d, const float *rk4, const void *dG, const float *dg_amax, void *dF,
                               const void *add, int bf16, int out_bf16, void *extra, hipStream_t st) {
    const int nt = d->nn <= 16 ? 1 : (d->nn <= 32 ? 2 : 4), ew = 16 * nt;
    uint32_t *tab = static_cast<uint32_t *>(extra);
    unsigned *mulmax = reinterpret_cast<unsigned *>(static_cast<char *>(extra) + inter_ungroup_cloud_extra_bytes(d) - 256);
    EPN_HIP(hipMemsetAsync(mulmax, 0, 8, st));
    if (!dg_amax) {                                            // no maximum supplied: one pass over dG
        unsigned *slot = mulmax + 1;
        const long long n = (long long)d->b * d->p2 * d->na * d->cin * d->ks;
        const unsigned nb = (unsigned)((n + 256 * 16 - 1) / (256 * 16) < 8192 ? (n + 256 * 16 - 1) / (256 * 16) : 8192);
        if (bf16) EPN_LAUNCH_AUX(uc_absmax_kernel<__bf16>, dim3(nb ? nb : 1), dim3(256), 0, st, static_cast<const __bf16 *>(dG), n, slot);
        else EPN_LAUNCH_AUX(uc_absmax_kernel<float>, dim3(nb ? nb : 1), dim3(256), 0, st, static_cast<const float *>(dG), n, slot);
        EPN_CHECK_LAUNCH();
        dg_amax = reinterpret_cast<const float *>(slot);
    }
    const long long npts = (long long)d->b * d->p2;
    float *hood = reinterpret_cast<float *>(static_cast<char *>(extra) + uc_off_bytes(d));
    EPN_LAUNCH_AUX(uc_slots_kernel, dim3((unsigned)((npts + 4 * UC_PTS - 1) / (4 * UC_PTS))), dim3(256), 0, st, d->ball_idx, d->xyz, d->new_xyz, npts, d->p1, d->p2,
                   d->nn, ew, 1.0f / d->sigma, (unsigned)uc_channels_per_wg(d, bf16) * 8u, tab, hood, mulmax);
    EPN_CHECK_LAUNCH();
    UcArgs P;
    InterArgs &A = P.A;
    A.xyz = d->xyz; A.new_xyz = d->new_xyz; A.idx = d->ball_idx; A.rk4 = rk4;
    A.feats = nullptr; A.W = nullptr; A.gout = static_cast<const float *>(dG); A.out = static_cast<float *>(dF);
    A.sigma_inv = 1.0f / d->sigma;
    A.b = d->b; A.p1 = d->p1; A.p2 = d->p2; A.nn = d->nn; A.na = d->na; A.ks = d->ks; A.cin = d->cin; A.cout = d->cout;
    A.wk = 0; A.packed = 0; A.ncol = (long long)d->b * d->p2 * d->na; A.col_tiles_per_wg = 1;
    P.tab = tab; P.hood = hood; P.mulmax = mulmax; P.dg_amax = dg_amax; P.add = add; P.out_bf16 = out_bf16;
    P.cr = uc_channels_per_wg(d, bf16);
    const size_t lds = uc_lds_bytes(d, P.cr);
    const dim3 grid((unsigned)(d->b * d->na), (unsigned)(d->cin / P.cr));
    const int kt = (d->ks + 15) / 16, nl = P.cr / 16;
    int rc = 0;
#define EPN_UC(NT_, KT_, TG_, NL_)                                                                                   \
    do {                                                                                                             \
        rc = uc_set_lds(inter_ungroup_cloud_kernel<NT_, KT_, TG_, 16, NL_>, lds);                                    \
        if (rc) return rc;                                                                                           \
        EPN_LAUNCH((inter_ungroup_cloud_kernel<NT_, KT_, TG_, 16, NL_>), grid, dim3(64 * 16), lds, st, P);           \
    } while (0)
#define EPN_UC_L(NT_, KT_, TG_)                                                                                      \
    do {                                                                                                             \
        if (nl == 1) EPN_UC(NT_, KT_, TG_, 1); else if (nl == 2) EPN_UC(NT_, KT_, TG_, 2);                           \
        else if (nl == 4) EPN_UC(NT_, KT_, TG_, 4); else EPN_UC(NT_, KT_, TG_, 8);                                   \
    } while (0)
#define EPN_UC_T(NT_, KT_)                                                                                           \
    do {                                                                                                             \
        if (bf16) EPN_UC_L(NT_, KT_, __bf16); else EPN_UC_L(NT_, KT_, float);                                        \
    } while (0)
    if (kt == 1) {
        if (nt == 1) EPN_UC_T(1, 1); else if (nt == 2) EPN_UC_T(2, 1); else EPN_UC_T(4, 1);
    } else {
        if (nt == 1) EPN_UC_T(1, 2); else if (nt == 2) EPN_UC_T(2, 2); else EPN_UC_T(4, 2);
    }
#undef EPN_UC_T
#undef EPN_UC_L
#undef EPN_UC
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
