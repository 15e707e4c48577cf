// InterSO3Conv for a single input channel (cin = 1): the first layer of every shipped model, whose input is the
// occupancy feature of get_occupancy_features (vgtk/vgtk/so3conv/functional.py:25-44).  With one channel the layer
// is not matrix work (24 grouped values and a [cout x 24] weight per column): it is bound by the weight generation
// on the VALU and by the output write.  One lane owns one column (b, p, a):
//   G[k]   = sum_n F[idx[n], a] * relu(1 - |g_n - R_a kappa_k|^2 / sigma)        (functional.py:190-200, expanded)
//   out[o] = sum_k W[o][k] * G[k]                                                  (modules.py:52)
// The weight gradient reduces dOut (x) G over all columns with MFMAs (M = kernel point, N = output channel,
// contraction = columns), the grouped values passing through LDS once.
// Round 4: for features that do not depend on the anchor (the occupancy feature itself) the relu arguments come off the
// matrix pipe instead -- inter_c1_fwd_mfma_kernel below, one wave per output point -- and the host computes the weight
// gradient from the saved grouped values with the library's TN GEMM (ops.InterSO3ConvFn.backward).
#include "conv_internal.h"

namespace epn {
namespace {

struct C1Args {
    const float *xyz, *new_xyz;
    const int32_t *idx;
    const float *rk;      // [na][ks][3]
    const float *feats;   // [b][p1][na]  (cin = 1)
    const float *W;       // fwd: [cout][ks]
    const float *gout;    // bwd: [ncol][cout]
    float *out;           // fwd: [ncol][cout];  bwd: dW [cout][ks]
    float *gsave;         // fwd, optional: the grouped values G[ncol][ks] kept for the weight gradient
    const float *gload;   // bwd, optional: G from the forward pass (no weight generation in the backward pass)
    float sigma_inv;
    int p1, p2, nn, na, ks, cout;
    long long ncol;
    int groups_per_wg;
    unsigned *flag;       // fwd, optional: 0 = the features do not depend on the anchor (matrix-pipe kernel runs), else the VALU kernel
};

// grouped values of one column into g[0..ks)
__device__ __forceinline__ void group_column(const C1Args &A, long long col, float (&g)[EPN_KS_MAX]) {
    const int a = (int)(col % A.na);
    const long long pt = col / A.na;
    const int bb = (int)(pt / A.p2), pp = (int)(pt - (long long)bb * A.p2);
    // expanded form of relu(1 - |g_n - R_a kappa_k|^2 / sigma) (the one the MFMA kernels use, inter_mfma.hip):
    //   alpha_n + beta_k + (2/sigma) g_n . (R_a kappa_k),  alpha_n = 1 - |g_n|^2/sigma,  beta_k = -|kappa_k|^2/sigma
    // 5 VALU operations per (neighbour, kernel point) instead of 9; differs from the literal form by fp32 rounding
    // (<= 5e-7 on w)
    // kernel points in PAIRS: the per-(neighbour, kernel point) arithmetic below is written on two-float vectors so that
    // hipcc emits v_pk_fma_f32 / v_pk_add_f32 (two fp32 operations per lane and instruction on this part): 3.5 instead of 6
    // VALU instructions per weight -- this kernel is nothing but that loop (15k instructions per column at K = 128)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int KP = EPN_KS_MAX / 2;
    f32x2_t rx[KP], ry[KP], rz[KP], beta[KP], g2[KP];
    const float two_si = 2.0f * A.sigma_inv;
#pragma unroll
    for (int k = 0; k < EPN_KS_MAX; ++k) {
        const float *e = A.rk + ((size_t)a * A.ks + (k < A.ks ? k : 0)) * 3;
        const float x = e[0], y = e[1], z = e[2];
        beta[k >> 1][k & 1] = -(x * x + y * y + z * z) * A.sigma_inv;
        rx[k >> 1][k & 1] = two_si * x; ry[k >> 1][k & 1] = two_si * y; rz[k >> 1][k & 1] = two_si * z;
        g2[k >> 1][k & 1] = 0.f;
    }
    const int kpairs = (A.ks + 1) >> 1;          // an odd ks: the pad slot repeats kernel point 0 and is dropped below
    const int32_t *row = A.idx + ((size_t)bb * A.p2 + pp) * A.nn;
    const float *s = A.xyz + (size_t)bb * 3 * A.p1;
    const float *c = A.new_xyz + (size_t)bb * 3 * A.p2;
    const float cx = c[pp], cy = c[A.p2 + pp], cz = c[2 * A.p2 + pp];
    const float *f = A.feats + ((size_t)bb * A.p1) * A.na + a;
    // Neighbours in batches of four (round 4): index -> coordinates -> feature is a chain of dependent loads, and with one
    // neighbour per iteration (and a `continue` the compiler cannot hoist loads across) every neighbour cost a full L2 / HBM
    // round trip per wave -- at two or three waves per SIMD the K = 64 / 128 first layers of the rotation / 3DMatch networks
    // were bound by that latency (1.0 / 1.75 ms), not by the 3.5 VALU instructions per weight.  A shadow index reads row 0
    // with feature value 0 instead of being skipped: same sum.
    constexpr int NB4 = 4;
    for (int n0 = 0; n0 < A.nn; n0 += NB4) {
        int qv[NB4];
        bool okv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            const int q = n0 + u < A.nn ? row[n0 + u] : -1;
            okv[u] = q >= 0 && q < A.p1;       // shadow index: zero feature row (spconv/functional.py:91-95)
            qv[u] = okv[u] ? q : 0;
        }
        float gxv[NB4], gyv[NB4], gzv[NB4], fvv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            gxv[u] = s[qv[u]]; gyv[u] = s[A.p1 + qv[u]]; gzv[u] = s[2 * A.p1 + qv[u]];
            fvv[u] = f[(size_t)qv[u] * A.na];
        }
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
        const float gx = gxv[u] - cx, gy = gyv[u] - cy, gz = gzv[u] - cz;
        const float alpha = 1.0f - (gx * gx + gy * gy + gz * gz) * A.sigma_inv;
        const float fv = okv[u] ? fvv[u] : 0.0f;
        const f32x2_t a2 = {alpha, alpha}, gx2 = {gx, gx}, gy2 = {gy, gy}, gz2 = {gz, gz}, fv2 = {fv, fv};
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            if (kp < kpairs) {
                // same operation order per element as the scalar form: ((alpha + beta) + gx rx) + gy ry) + gz rz
                f32x2_t sv = (a2 + beta[kp]) + gx2 * rx[kp];
                sv = sv + gy2 * ry[kp];
                sv = sv + gz2 * rz[kp];
                // (scalar temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0 for both -- hipcc)
                const float e0 = sv[0], e1 = sv[1];
                const int s0 = __float_as_int(e0), s1 = __float_as_int(e1);
                const f32x2_t w = {__int_as_float(s0 > 0 ? s0 : 0), __int_as_float(s1 > 0 ? s1 : 0)};
                g2[kp] = g2[kp] + fv2 * w;
            }
        }
        }
    }
#pragma unroll
    for (int k = 0; k < EPN_KS_MAX; ++k) g[k] = k < A.ks ? g2[k >> 1][k & 1] : 0.f;
}

__global__ __launch_bounds__(256) void inter_c1_fwd_kernel(C1Args A) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];   // [cout][ks]
    if (A.flag && *A.flag == 0) return;                          // anchor-independent features: inter_c1_fwd_mfma_kernel ran
    for (int i = threadIdx.x; i < A.cout * A.ks; i += blockDim.x) Ws[i] = A.W[i];
    __syncthreads();
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    float g[EPN_KS_MAX];
    group_column(A, col, g);
    if (A.gsave) {   // 96 bytes per lane, consecutive lanes consecutive rows: the weight gradient re-reads this instead of
                     // regenerating 24 x K weights per column (the backward kernel WAS that loop a second time)
        float *gs = A.gsave + col * A.ks;
        if ((A.ks & 3) == 0) {
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; k += 4)
                if (k < A.ks) *reinterpret_cast<f32x4 *>(gs + k) = f32x4{g[k], g[k + 1], g[k + 2], g[k + 3]};
        } else {
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; ++k)
                if (k < A.ks) gs[k] = g[k];
        }
    }
    float *o = A.out + col * A.cout;
    for (int o4 = 0; o4 < A.cout; o4 += 4) {   // cout % 4 == 0 (launcher)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < EPN_KS_MAX; ++k) {
            if (k < A.ks) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += Ws[(o4 + r) * A.ks + k] * g[k];   // LDS broadcast reads
            }
        }
        *reinterpret_cast<f32x4 *>(o + o4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer on the matrix pipe, for features that do not depend on the anchor (the occupancy feature: ones, or
// zeros / ones with use_center) -- checked on the device by c1_feats_check_kernel, anything else takes the kernel above.
//
// The VALU kernel spends 3.5 instructions per (neighbour, kernel point) weight and measured 2.0-2.5 weights per clock and
// SIMD (dependent index -> coordinate loads repeated by the 60 anchor lanes of a point).  The argument of the relu,
//     s[n][(a,k)] = alpha_n + beta_k + (2/sigma) g_n . (R_a kappa_k),
// is a rank-5 product  [g_x g_y g_z alpha 1]_n . [r_x r_y r_z 1 beta]_(a,k):  one v_mfma_f32_16x16x32_bf16 per 16 neighbours
// x 16 (anchor, kernel point) columns with both sides split WITHOUT LOSS into three bf16 pieces (h, m, l) and the six
// products hh, hm, mh, hl, lh, mm of every term laid along the contraction (4 terms x 6 + 3 for beta = 27 of 32 slots):
// fp32-grade s (the dropped ml / lm / ll products are < 2^-24 of a term), 256 values per 16 matrix-pipe cycles.  What is
// left for the VALU per MFMA: four v_max (relu) and two packed FMAs with the neighbours' feature values.
//   * one wave = one output point (all anchors): index row and neighbour coordinates are loaded once per point; the A
//     fragments (nn/16 x 4 registers) stay in registers while the wave walks the 90 column tiles;
//   * the B fragments (constants of the launch: 90 tiles x 1 KB) are built once per workgroup into LDS; 256 persistent
//     workgroups of 8 waves;
//   * D rows = neighbours, so the sum over neighbours is in-lane + one cross-row step (v_permlane{32,16}_swap);
//   * every two anchors (3 column tiles = 48 grouped values) the wave contracts them with W (lane = output channel, its 24
//     weights in registers, grouped values broadcast from LDS) and writes 2 x cout contiguous outputs.
typedef float c1f2 __attribute__((ext_vector_type(2)));
typedef unsigned c1u4 __attribute__((ext_vector_type(4)));
typedef __bf16 c1bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 c1bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned c1_bf(float a) {   // bf16 bits (RNE) of a, in the low half
    const c1f2 v = {a, 0.f};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, c1bf2)) & 0xffffu;
}
__device__ __forceinline__ void c1_split(float xv, unsigned &h, unsigned &m, unsigned &l) {   // xv = h + m + l exactly
    h = c1_bf(xv);
    const float r = xv - __uint_as_float(h << 16);
    m = c1_bf(r);
    l = c1_bf(r - __uint_as_float(m << 16));
}
__device__ __forceinline__ float c1_rows_sum(float v) {   // sum over the four 16-lane rows, result in every lane
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// flag |= "some feats[row][a] differs from feats[row][0]"   (flag zeroed by the launcher); 64 lanes per row, na <= 128
__global__ void c1_feats_check_kernel(const float *__restrict__ feats, long long rows, int na, unsigned *flag) {
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * (blockDim.x >> 6)) {
        const float *f = feats + r * na;
        const float f0 = f[0];
        for (int a = lane; a < na; a += 64) bad = bad || !(f[a] == f0);
    }
    if (bad) *flag = 1u;
}

#define EPN_C1M_KS 24   // kernel points per anchor of the matrix-pipe form (3 column tiles per anchor pair)

template <int NRT, int PTS, int NW>   // row tiles (16 neighbours) per point, points per wave and B fragment, waves
__global__ __launch_bounds__(64 * NW) void inter_c1_fwd_mfma_kernel(C1Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char c1_smem[];
    if (*A.flag != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NW;
    const int x = lane & 15, j = lane >> 4;
    const int ncols = A.na * EPN_C1M_KS, nct = ncols >> 4;           // na even (launcher): nct = 3 na / 2
    c1u4 *Btab = reinterpret_cast<c1u4 *>(c1_smem);                   // [nct][64]
    float *stage = reinterpret_cast<float *>(Btab + (size_t)nct * 64) + (size_t)wave * PTS * 64;       // [PTS][48 + pad]
    float *phis = reinterpret_cast<float *>(Btab + (size_t)nct * 64) + (size_t)nw * PTS * 64 + (size_t)wave * PTS * NRT * 16;
    const float two_si = 2.0f * A.sigma_inv;
    for (int e = tid; e < nct * 64; e += blockDim.x) {
        const int ct = e >> 6, l = e & 63, xx = l & 15, jj = l >> 4;
        const int col = 16 * ct + xx, a = col / EPN_C1M_KS, k = col - a * EPN_C1M_KS;
        const float *r = A.rk + ((size_t)a * A.ks + k) * 3;
        const float rx = r[0], ry = r[1], rz = r[2];
        const float v = jj == 0 ? two_si * rx : jj == 1 ? two_si * ry : jj == 2 ? two_si * rz : 1.0f;
        const float beta = -(rx * rx + ry * ry + rz * rz) * A.sigma_inv;
        unsigned h, m, l3, bh, bm, bl;
        c1_split(v, h, m, l3);
        c1_split(beta, bh, bm, bl);
        c1u4 w;                                                      // B slots: h m h l h m | beta pieces against A's 1.0
        w[0] = h | (m << 16); w[1] = h | (l3 << 16); w[2] = h | (m << 16);
        w[3] = jj == 0 ? (bh | (bm << 16)) : jj == 1 ? bl : 0u;
        Btab[e] = w;
    }
    const int o = lane % A.cout, asel = lane / A.cout, aslots = 64 / A.cout;   // cout in {16, 32, 64} (launcher)
    const int sidx = 16 * j + x;       // this lane's slot of an anchor pair's staging row: column 16 j + x = 24 aa + k
    float Wr[EPN_C1M_KS];
#pragma unroll
    for (int k = 0; k < EPN_C1M_KS; ++k) Wr[k] = A.W[o * A.ks + k];
    __syncthreads();
    const long long npts = A.ncol / A.na;
    const long long nunits = (npts + PTS - 1) / PTS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long u = (long long)blockIdx.x * nw + wave; u < nunits; u += (long long)gridDim.x * nw) {
        c1bf8 af[PTS][NRT];
        c1f2 ph01[PTS][NRT], ph23[PTS][NRT];
        long long ptv[PTS];
#pragma unroll
        for (int p = 0; p < PTS; ++p) {
            const long long pt0 = u * PTS + p;
            const long long pt = pt0 < npts ? pt0 : npts - 1;   // odd tail: the last point twice (same values stored twice)
            ptv[p] = pt;
            const int bb = (int)(pt / A.p2), pp = (int)(pt - (long long)bb * A.p2);
            const int32_t *row = A.idx + (size_t)pt * A.nn;
            const float *s = A.xyz + (size_t)bb * 3 * A.p1;
            const float *c = A.new_xyz + (size_t)bb * 3 * A.p2;
            const float cx = c[pp], cy = c[A.p2 + pp], cz = c[2 * A.p2 + pp];
            const float *f = A.feats + (size_t)bb * A.p1 * A.na;
            int qv[NRT];
            bool okv[NRT];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const int n = 16 * rt + x;
                const int q = n < A.nn ? row[n] : -1;
                okv[rt] = q >= 0 && q < A.p1;          // shadow index: zero feature row (spconv/functional.py:91-95)
                qv[rt] = okv[rt] ? q : 0;
            }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const float gx = s[qv[rt]] - cx, gy = s[A.p1 + qv[rt]] - cy, gz = s[2 * A.p1 + qv[rt]] - cz;
                const float alpha = 1.0f - (gx * gx + gy * gy + gz * gz) * A.sigma_inv;
                const float v = j == 0 ? gx : j == 1 ? gy : j == 2 ? gz : alpha;
                unsigned h, m, l3;
                c1_split(v, h, m, l3);
                c1u4 w;                                              // A slots: h h m h l m | 1.0 against beta's pieces
                w[0] = h | (h << 16); w[1] = m | (h << 16); w[2] = l3 | (m << 16);
                w[3] = j == 0 ? 0x3f803f80u : j == 1 ? 0x00003f80u : 0u;
                if (!okv[rt]) w = c1u4{0u, 0u, 0u, 0u};
                af[p][rt] = __builtin_bit_cast(c1bf8, w);
                if (j == 0) phis[p * NRT * 16 + 16 * rt + x] = okv[rt] ? f[(size_t)qv[rt] * A.na] : 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < PTS; ++p)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(phis + p * NRT * 16 + 16 * rt + 4 * j);   // rows 4j .. 4j+3
                ph01[p][rt] = c1f2{t[0], t[1]};
                ph23[p][rt] = c1f2{t[2], t[3]};
            }
        // per point: where its grouped values / outputs go (offsets inside a point are small ints)
        float *gsp[PTS], *outp[PTS];
#pragma unroll
        for (int p = 0; p < PTS; ++p) {
            gsp[p] = A.gsave ? A.gsave + (size_t)ptv[p] * A.na * EPN_C1M_KS + x : nullptr;
            outp[p] = A.out + (size_t)ptv[p] * A.na * A.cout + o;
        }
        constexpr int NM = 3 * PTS * NRT;            // MFMAs per anchor pair
        c1u4 bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = Btab[t * 64 + lane];
        for (int ap = 0; ap < (A.na >> 1); ++ap) {
            c1bf8 bf[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) bf[t] = __builtin_bit_cast(c1bf8, bw[t]);
            if (ap + 1 < (A.na >> 1)) {              // next pair's B fragments while this pair computes
#pragma unroll
                for (int t = 0; t < 3; ++t) bw[t] = Btab[(3 * (ap + 1) + t) * 64 + lane];
            }
            // software pipeline over the pair's MFMAs: MFMA i+1 is issued before the six VALU instructions that consume MFMA i
            f32x4 dn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][0], bf[0], zero4, 0, 0, 0);
            c1f2 acc = {0.f, 0.f}, acd = {0.f, 0.f};     // (two chains: a packed FMA depending on the previous one costs a wait state)
            float vt[PTS][3];                            // per tile: this lane's sum over its rows of all row tiles
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int t = i / (PTS * NRT), p = (i / NRT) % PTS, rt = i % NRT;
                const f32x4 d = dn;
                if (i + 1 < NM) {
                    const int t1 = (i + 1) / (PTS * NRT), p1 = ((i + 1) / NRT) % PTS, rt1 = (i + 1) % NRT;
                    dn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p1][rt1], bf[t1], zero4, 0, 0, 0);
                }
                if (rt == 0) { acc = c1f2{0.f, 0.f}; acd = c1f2{0.f, 0.f}; }
                // relu on the bit patterns (v_max_i32: one instruction, no canonicalisation), as in the VALU kernel
                const int i0 = __float_as_int(d[0]), i1 = __float_as_int(d[1]), i2 = __float_as_int(d[2]), i3 = __float_as_int(d[3]);
                const c1f2 r01 = {__int_as_float(i0 > 0 ? i0 : 0), __int_as_float(i1 > 0 ? i1 : 0)};
                const c1f2 r23 = {__int_as_float(i2 > 0 ? i2 : 0), __int_as_float(i3 > 0 ? i3 : 0)};
                acc = r01 * ph01[p][rt] + acc;
                acd = r23 * ph23[p][rt] + acd;
                if (rt == NRT - 1) {
                    const c1f2 a2 = acc + acd;
                    vt[p][t] = a2[0] + a2[1];
                }
            }
            // sum over the four 16-lane rows, transposing: lane group j ends with the total of tile j (3 swaps + 3 adds per point)
#pragma unroll
            for (int p = 0; p < PTS; ++p) {
                auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(vt[p][0]), __float_as_uint(vt[p][2]), false, false);
                auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(vt[p][1]), 0u, false, false);
                const float e = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);   // rows 0-1: tile 0 halves summed, rows 2-3: tile 2
                const float f = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);   // rows 0-1: tile 1, rows 2-3: nothing
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(e), __float_as_uint(f), false, false);
                const float g = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);     // row j: total of tile j (row 3: 0)
                stage[p * 64 + sidx] = g;                                             // (row 3 writes the pad slots 48 ...)
                if (A.gsave && j < 3) gsp[p][2 * ap * EPN_C1M_KS + 16 * j] = g;      // 48 contiguous floats per store
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < PTS; ++p)
                for (int aa = asel; aa < 2; aa += aslots) {
                    const float *gs = stage + p * 64 + EPN_C1M_KS * aa;
                    float r = 0.f;
#pragma unroll
                    for (int k4 = 0; k4 < EPN_C1M_KS; k4 += 4) {
                        const f32x4 gv = *reinterpret_cast<const f32x4 *>(gs + k4);      // same address in all lanes of an anchor
#pragma unroll
                        for (int q = 0; q < 4; ++q) r += Wr[k4 + q] * gv[q];
                    }
                    outp[p][(2 * ap + aa) * A.cout] = r;
                }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// dW[o][k] = sum_col dOut[col][o] * G[col][k];  4 waves, each wave contracts its own 64 columns per group.
__global__ __launch_bounds__(256) void inter_c1_bwd_weight_kernel(C1Args A) {
    __shared__ float Gs[256][EPN_KS_MAX + 1];   // [column of the group][kernel point], zero padded to 32
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = lane & 15, j = lane >> 4;
    const int NTO = A.cout >> 4;   // <= 4 (launcher)
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int it = 0; it < A.groups_per_wg; ++it) {
        const long long base = ((long long)blockIdx.x * A.groups_per_wg + it) * 256;
        if (base >= A.ncol) break;
        const long long col = base + threadIdx.x;
        if (A.gload) {
            // the wave's 64 rows of saved G are one contiguous run of 64 ks floats: coalesced loads, scattered to the tile
            const long long w0 = base + wave * 64;
            const long long wn = A.ncol - w0 < 64 ? A.ncol - w0 : 64;       // rows of this wave inside the tensor (may be <= 0)
            const float *src = A.gload + w0 * A.ks;
            const int total = wn > 0 ? (int)wn * A.ks : 0;
            if ((A.ks & 3) == 0 && ((reinterpret_cast<size_t>(src) & 15) == 0)) {
                for (int e = 4 * lane; e < 64 * A.ks; e += 256) {
                    const f32x4 v = e < total ? *reinterpret_cast<const f32x4 *>(src + e) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const int r = e / A.ks, k0 = e - r * A.ks;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Gs[wave * 64 + r][k0 + q] = v[q];
                }
            } else {
                for (int e = lane; e < 64 * A.ks; e += 64) {
                    const int r = e / A.ks;
                    Gs[wave * 64 + r][e - r * A.ks] = e < total ? src[e] : 0.f;
                }
            }
            for (int e = lane; e < 64 * (EPN_KS_MAX - A.ks); e += 64) {       // zero padding up to 32 kernel points
                const int r = e / (EPN_KS_MAX - A.ks);
                Gs[wave * 64 + r][A.ks + e - r * (EPN_KS_MAX - A.ks)] = 0.f;
            }
        } else {
            float g[EPN_KS_MAX];
            if (col < A.ncol) {
                group_column(A, col, g);
            } else {
#pragma unroll
                for (int k = 0; k < EPN_KS_MAX; ++k) g[k] = 0.f;
            }
            // only this wave reads the rows it writes: wave-level ordering is enough
#pragma unroll
            for (int k = 0; k < EPN_KS_MAX; ++k) Gs[threadIdx.x][k] = k < A.ks ? g[k] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        const long long wbase = base + wave * 64;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {   // contraction step: columns wbase + 4s + j
            const long long cc = wbase + 4 * s + j;
            const bool ok = cc < A.ncol;
            float bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = (n < NTO && ok) ? A.gout[cc * A.cout + 16 * n + x] : 0.0f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float af = Gs[wave * 64 + 4 * s + j][16 * m + x];
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n < NTO) acc[m][n] = mfma4(af, bf[n], acc[m][n]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // acc[m][n]: lane (x = o within tile n, j), register r -> k = 16m + 4j + r
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < NTO)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * m + 4 * j + r;
                    if (k < A.ks) atomicAdd(A.out + (size_t)(16 * n + x) * A.ks + k, acc[m][n][r]);
                }
}

C1Args make_c1(const epn_inter_desc *d, const float *rk) {
    C1Args A;
    A.xyz = d->xyz; A.new_xyz = d->new_xyz; A.idx = d->ball_idx; A.rk = rk;
    A.feats = nullptr; A.W = nullptr; A.gout = nullptr; A.out = nullptr; A.gsave = nullptr; A.gload = nullptr;
    A.sigma_inv = 1.0f / d->sigma;
    A.p1 = d->p1; A.p2 = d->p2; A.nn = d->nn; A.na = d->na; A.ks = d->ks; A.cout = d->cout;
    A.ncol = (long long)d->b * d->p2 * d->na;
    A.groups_per_wg = 1;
    A.flag = nullptr;
    return A;
}

}  // namespace

bool inter_c1_fwd_ok(const epn_inter_desc *d) {
    return d->cin == 1 && !d->dense_w && d->ks <= EPN_KS_MAX && d->cout % 4 == 0 &&
           (size_t)d->cout * d->ks * sizeof(float) <= 48 * 1024;
}

bool inter_c1_bwd_weight_ok(const epn_inter_desc *d) {
    return d->cin == 1 && !d->dense_w && d->ks <= EPN_KS_MAX && d->cout % 16 == 0 && d->cout <= 64;
}

// matrix-pipe form: 24 kernel points, anchors in pairs, cout lanes per anchor, the B table + staging within the 160 KB of LDS
static bool c1_mfma_ok(const epn_inter_desc *d) {
    // (epn_set_kernel_policy(2): the VALU kernel for every input -- A/B and cross-check; the library reads no environment)
    return !first_layer_on_valu() && d->ks == EPN_C1M_KS && d->na % 2 == 0 && d->na <= 128 && d->nn >= 1 && d->nn <= 128 &&
           (d->cout == 16 || d->cout == 32 || d->cout == 64);
}

template <int NRT, int PTS, int NW>
static int launch_c1_mfma(const C1Args &A, hipStream_t st) {
    const int nct = A.na * EPN_C1M_KS / 16;
    const size_t lds = (size_t)nct * 1024 + (size_t)NW * PTS * 64 * 4 + (size_t)NW * PTS * NRT * 16 * 4;
    EPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&inter_c1_fwd_mfma_kernel<NRT, PTS, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long units = (A.ncol / A.na + PTS - 1) / PTS;
    const unsigned grid = (unsigned)((units + NW - 1) / NW < 256 ? (units + NW - 1) / NW : 256);
    EPN_LAUNCH((inter_c1_fwd_mfma_kernel<NRT, PTS, NW>), dim3(grid), dim3(64 * NW), lds, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

int launch_inter_c1_fwd(const epn_inter_desc *d, const float *rk, const float *feats, const float *W, float *out,
                        hipStream_t st, float *grouped_save, unsigned *flag) {
    C1Args A = make_c1(d, rk);
    A.feats = feats; A.W = W; A.out = out; A.gsave = grouped_save;
    const bool mf = flag && c1_mfma_ok(d);
    if (mf) {
        // features that do not depend on the anchor (checked here, on the device): the matrix-pipe kernel; the VALU kernel
        // returns at once unless the check found a difference (it is launched first so that epn_last_kernel() names the
        // kernel that does the work for the occupancy feature)
        A.flag = flag;
        EPN_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), st));
        const long long rows = (long long)d->b * d->p1;
        EPN_LAUNCH_AUX(c1_feats_check_kernel, dim3((unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096)), dim3(256), 0, st,
                       feats, rows, d->na, flag);
        EPN_CHECK_LAUNCH();
    }
    const unsigned grid = (unsigned)((A.ncol + 255) / 256);
    EPN_LAUNCH(inter_c1_fwd_kernel, dim3(grid), dim3(256), (size_t)d->cout * d->ks * sizeof(float), st, A);
    EPN_CHECK_LAUNCH();
    if (mf) {
        // waves per workgroup (= per CU: the B table takes 90 KB of LDS): as many as the registers allow -- measured
        // 8 / 12 / 16 waves: K = 32 0.243 / 0.233 / 0.205 ms, K = 64 0.459 / 0.420 / 0.403, K = 128 0.647 / 0.623 / - (152 VGPRs)
        if (d->nn <= 16) return launch_c1_mfma<1, 2, 16>(A, st);
        if (d->nn <= 32) return launch_c1_mfma<2, 2, 16>(A, st);
        if (d->nn <= 64) return launch_c1_mfma<4, 1, 16>(A, st);
        return launch_c1_mfma<8, 1, 12>(A, st);
    }
    return 0;
}

int launch_inter_c1_bwd_weight(const epn_inter_desc *d, const float *rk, const float *feats, const float *dOut,
                               float *dW, hipStream_t st, const float *grouped_saved) {
    C1Args A = make_c1(d, rk);
    A.feats = feats; A.gout = dOut; A.out = dW; A.gload = grouped_saved;
    const long long groups = (A.ncol + 255) / 256;
    // two rounds of workgroups: each ends in cout*ks atomics onto the same 1536 addresses (2048 workgroups: 1.48 ms,
    // of which ~0.9 ms contention; 512: see DESIGN 5)
    long long wgs = groups < 512 ? groups : 512;
    A.groups_per_wg = (int)((groups + wgs - 1) / wgs);
    const unsigned grid = (unsigned)((groups + A.groups_per_wg - 1) / A.groups_per_wg);
    EPN_LAUNCH(inter_c1_bwd_weight_kernel, dim3(grid), dim3(256), 0, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

}  // namespace epn
