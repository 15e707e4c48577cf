// Device-side building blocks shared by the InterSO3Conv kernels (inter_mfma.hip, inter_fx.hip): kernel arguments,
// per-point neighbourhood fragments, kernel-influence weights by S-MFMA, feature-row gathers.
// Fragment layouts (verified on hardware by tools/mfma_probe.hip): lane l, x = l & 15, j = l >> 4:
//   A[m = x][k = j],  B[k = j][n = x],  D[m = 4j + r][n = x]  (r = register 0..3).
#pragma once
#include <type_traits>

#include "conv_internal.h"

namespace epn {
namespace {

constexpr int NW = 4;  // waves per workgroup

struct InterArgs {
    const float *xyz, *new_xyz;
    const int32_t *idx;
    const float *rk4;    // [na][32][4] = ((2/sigma) R_a kappa_k, beta_k), zero / -1e30 padded
    const float *feats;  // fwd: feats_cl [b][p1][na][cin];  bwd_weight: same
    const float *W;      // fwd: W [cout][cin*ks];           bwd_data: WT [cin*ks][cout]
    const float *gout;   // bwd: grad_out_cl [ncol][cout]
    float *out;          // fwd: out_cl [ncol][cout]; bwd_data: grad_feats_cl; bwd_weight: grad_W
    float sigma_inv;
    int b, p1, p2, nn, na, ks, cin, cout, wk;
    long long ncol;
    int col_tiles_per_wg;  // bwd_weight only
    int packed;            // grouping: write G in the packed column order (inter_packed_position)
};


// max(x, 0) as ONE instruction: the integer maximum of the bit patterns (negative floats are negative integers).
// fmaxf(x, 0.0f) compiles to two v_max_f32 (IEEE canonicalisation of the operand first); the weight generation does
// 16-32 of them per column and the grouping kernels are VALU-bound.
__device__ __forceinline__ float relu_f(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup barrier for data exchanged through LDS ONLY.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it also
// waits for every outstanding global access of the wave -- in the LDS-reduced scatters that is the acknowledgement of the fp32
// atomics of the previous anchor step (1-3 us each step, found in the ISA of inter_bwd_f2.hip, round 6) and the prefetched dG
// fragments of the next one.  Here only the LDS queue is drained; global loads are waited for where their values are used.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Feature / grouped-feature storage type of the grouping kernels: float, or __bf16 for the bf16 feature path (weights
// w and the neighbour contraction stay fp32: "bf16 features, fp32 accumulate").
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4f(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 ld4f(const __bf16 *p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t *>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// (write-through / non-temporal stores of the grouped features were measured: sc1, sc0 sc1 and nt all run the grouping
// kernel at 1.4-2.2 TB/s instead of 2.4-3.3 -- the 64- and 32-byte pieces of a 96-byte channel row need the L2 to merge them)
__device__ __forceinline__ void st4f(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void st4f(__bf16 *p, f32x4 v) {
    *reinterpret_cast<bf16x4_t *>(p) = bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

// bf16 feature path: the neighbour contraction runs on the bf16 MFMAs.  The weight MFMA's D fragment (lane (x, j),
// registers 0..3 = four consecutive contraction indices) is, rounded to bf16 and packed, exactly the A fragment of
// v_mfma_f32_16x16x16_bf16 (A[m = x][k = 4j + r]); two such fragments side by side feed v_mfma_f32_16x16x32_bf16
// (k = 8j + e: any bijection works as long as A and B use the same one).  Accumulation stays fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x4_t relu_pack4(f32x4 s) {
    return bf16x4_t{(__bf16)relu_f(s[0]), (__bf16)relu_f(s[1]), (__bf16)relu_f(s[2]), (__bf16)relu_f(s[3])};
}
__device__ __forceinline__ f32x4 mfma_bf16_k16(bf16x4_t a, bf16x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16_k32(bf16x4_t a0, bf16x4_t a1, bf16x4_t b0, bf16x4_t b1, f32x4 c) {
    const bf16x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const bf16x8_t b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Per-point neighbourhood fragments, shared by all columns (anchors) of one output point.
template <int NT>
struct Hood {
    float gA[NT];    // S-MFMA operand: lane (x, j) -> (g_x, g_y, g_z, alpha)[j] of neighbour 16t + x
    int q[NT][4];    // feature-row offset idx*na*cin (floats) for n = 16t + 4j + r, 0 when masked
    bool ok[NT][4];
    // data-gradient scatter only: the ball query pads a row that found cnt < K neighbours by repeating them cyclically
    // (grouping_cuda_kernel.cu:100-104), so slot n and slot n mod cnt are the same point with the same weight.  The
    // scatter adds mul * T once for the first occurrence (mul = number of slots holding that point) and nothing for
    // the repeats: same sum, 13-37 % fewer fp32 atomics on the ModelNet schedule (the atomics bound that kernel).
    float mul[NT][4];
};

template <int NT>
__device__ __forceinline__ void load_hood(const InterArgs &A, int bb, int pp, int x, int j, Hood<NT> &h) {
    const int32_t *row = A.idx + ((size_t)bb * A.p2 + pp) * A.nn;
    const float *s = A.xyz + (size_t)bb * 3 * A.p1;
    const float *c = A.new_xyz + (size_t)bb * 3 * A.p2;
    const float cx = c[pp], cy = c[A.p2 + pp], cz = c[2 * A.p2 + pp];
    // number of distinct neighbours = position of the first repeat of slot 0 (true hits are distinct points)
    const int first = row[0];
    int cnt = A.nn;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + x;
        const bool rep = n > 0 && n < A.nn && row[n] == first;
        const unsigned m16 = (unsigned)(__ballot(rep) & 0xffffull);   // lanes j = 0 carry x = 0..15
        if (m16 != 0u && cnt == A.nn) cnt = 16 * t + __builtin_ctz(m16);
    }
    // only a genuinely cyclic row is de-duplicated (index tensors handed in by the caller may be arbitrary)
    {
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + x;
            bad = bad || (n >= cnt && n < A.nn && row[n] != row[n - cnt]);
        }
        if (__ballot(bad) != 0ull) cnt = A.nn;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + x;
        int qq = n < A.nn ? row[n] : -1;
        const bool valid = qq >= 0 && qq < A.p1;
        qq = valid ? qq : 0;
        const float gx = s[qq] - cx, gy = s[A.p1 + qq] - cy, gz = s[2 * A.p1 + qq] - cz;
        const float alpha = valid ? 1.0f - (gx * gx + gy * gy + gz * gz) * A.sigma_inv : -1e30f;
        h.gA[t] = j == 0 ? gx : (j == 1 ? gy : (j == 2 ? gz : alpha));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n2 = 16 * t + 4 * j + r;
            int q2 = n2 < A.nn ? row[n2] : -1;
            h.ok[t][r] = q2 >= 0 && q2 < A.p1;
            h.q[t][r] = h.ok[t][r] ? q2 * A.na * A.cin : 0;
            h.mul[t][r] = (h.ok[t][r] && n2 < cnt) ? (float)((A.nn - 1 - n2) / cnt + 1) : 0.0f;
        }
    }
}

// Rotated-kernel table entries of one anchor as a lane needs them.  They are LOADS: a kernel that walks the anchors of a
// point must request the next anchor's entries a whole column ahead (with the next column's feature rows) -- read where they
// are used, hipcc waits for them with s_waitcnt vmcnt(0), which (the counter is in order) also waits for every store of the
// previous column and for the feature rows just requested: the grouping kernel ran its stores and its MFMAs strictly one
// after the other (1.75 = 0.97 + 0.8 ms on a 64-channel K = 16 layer) until round 3 found this in the ISA.
template <int KT>
struct RkRow { float rk[KT], beta[KT]; };

template <int KT>
__device__ __forceinline__ void load_rk_row(const InterArgs &A, int a, int x, int j, RkRow<KT> &r) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const float *e = A.rk4 + ((size_t)a * EPN_KS_MAX + 16 * kt + x) * 4;
        r.rk[kt] = e[j];            // lane group j = 3 multiplies alpha by 1: patched in make_weights_from
        r.beta[kt] = e[3];
    }
}

// w[kt][t] (4 registers each) for one column: lane (x, j), register r  ->  k = 16kt + x, n = 16t + 4j + r
template <int NT, int KT>
__device__ __forceinline__ void make_weights_from(const RkRow<KT> &rr, int j, const Hood<NT> &h, f32x4 (&w)[KT][NT]) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const float rk = j == 3 ? 1.0f : rr.rk[kt];
        const float beta = rr.beta[kt];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 s = {beta, beta, beta, beta};
            s = mfma4(h.gA[t], rk, s);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = relu_f(s[r]);
            w[kt][t] = s;
        }
    }
}

template <int NT, int KT>
__device__ __forceinline__ void make_weights(const InterArgs &A, int a, int x, int j, const Hood<NT> &h,
                                             f32x4 (&w)[KT][NT]) {
    RkRow<KT> rr;
    load_rk_row<KT>(A, a, x, j, rr);
    make_weights_from<NT, KT>(rr, j, h, w);
}

// Grouped features of 16 columns x 16 channels into the wave-private LDS tile Gs[col][c_local*ks + k].
// Generic form (any na): neighbourhood fragments are re-derived whenever the output point changes.
template <int NT, int KT, typename TF = float>
__device__ __forceinline__ void group_chunk_generic(const InterArgs &A, long long col0, int ct, int x, int j,
                                                    TF *Gs, int gss) {
    Hood<NT> h;
    int last_pt = -1;
    for (int jc = 0; jc < 16; ++jc) {
        long long col = col0 + jc;
        col = col < A.ncol ? col : A.ncol - 1;
        const int a = (int)(col % A.na);
        const int pt = (int)(col / A.na);  // b*p2 + p
        const int bb = pt / A.p2, pp = pt - bb * A.p2;
        if (pt != last_pt) {
            load_hood<NT>(A, bb, pp, x, j, h);
            last_pt = pt;
        }
        float f[NT][4];
        const TF *fb = reinterpret_cast<const TF *>(A.feats) + (((size_t)bb * A.p1) * A.na + a) * A.cin + 16 * ct + x;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (float)fb[h.q[t][r]];
                f[t][r] = h.ok[t][r] ? v : 0.0f;
            }
        f32x4 w[KT][NT];
        make_weights<NT, KT>(A, a, x, j, h, w);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g = mfma4(w[kt][t][r], f[t][r], g);
            if (16 * kt + 4 * j < A.ks)  // rows k = 16kt + 4j + r of channel x
                st4f(Gs + jc * gss + x * A.ks + 16 * kt + 4 * j, g);
        }
    }
}

// Fast form (na >= 16: a 16-column tile touches at most two output points).  The two neighbourhoods are
// derived ONCE per wave (outside the channel-chunk loop) and the feature rows of the next column are in
// flight while the current column's MFMAs run.
template <int NT>
struct Seg {          // columns [jc0, jc0 + cnt) of the tile: anchors a0.. of one output point
    Hood<NT> h;
    const float *fbase;   // feats + ((b*p1)*na)*cin  (element type TF of the kernel: an opaque base for bf16)
    int a0, jc0, cnt;
};

template <int NT, typename TF = float>
__device__ __forceinline__ void load_f(const InterArgs &A, const Seg<NT> &sg, int a, int coff, float (&f)[NT][4]) {
    const TF *fb = reinterpret_cast<const TF *>(sg.fbase) + (size_t)a * A.cin + coff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) f[t][r] = (float)fb[sg.h.q[t][r]];
}

// bf16 features: raw 16-bit neighbour values of one column, two to a dword, four to a B fragment.  Addressing is a
// wave-uniform row base plus a 32-bit lane offset (neighbour row + channel); masked slots (their offset points at row 0)
// are zeroed by an AND with a per-point mask, so there is no branch and no 64-bit arithmetic per load.
template <int NT>
struct RawHood {
    unsigned off[NT][4];     // element offset idx*na*cin + x of neighbour 16t + 4j + r
    unsigned mask[NT][2];    // 0xffff per valid slot, packed like the values
};

template <int NT>
__device__ __forceinline__ void make_raw_hood(const Seg<NT> &sg, int x, RawHood<NT> &rh) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rh.off[t][r] = (unsigned)sg.h.q[t][r] + (unsigned)x;
        rh.mask[t][0] = (sg.h.ok[t][0] ? 0xffffu : 0u) | (sg.h.ok[t][1] ? 0xffff0000u : 0u);
        rh.mask[t][1] = (sg.h.ok[t][2] ? 0xffffu : 0u) | (sg.h.ok[t][3] ? 0xffff0000u : 0u);
    }
}

template <int NT>
__device__ __forceinline__ void load_f_raw(const unsigned short *__restrict__ rowbase, const RawHood<NT> &rh,
                                           unsigned (&f)[NT][4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) f[t][r] = rowbase[rh.off[t][r]];
}

template <int NT>
__device__ __forceinline__ bf16x4_t pack_f_raw(const RawHood<NT> &rh, const unsigned (&f)[NT][4], int t) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 p;
    p[0] = (f[t][0] | (f[t][1] << 16)) & rh.mask[t][0];
    p[1] = (f[t][2] | (f[t][3] << 16)) & rh.mask[t][1];
    return __builtin_bit_cast(bf16x4_t, p);
}

__device__ __forceinline__ bf16x4_t pack4(f32x4 s) {      // s already clamped at zero
    return bf16x4_t{(__bf16)s[0], (__bf16)s[1], (__bf16)s[2], (__bf16)s[3]};
}

template <int NT, int KT>
__device__ __forceinline__ void group_segment_bf16(const InterArgs &A, const Seg<NT> &sg, int ct, int x, int j,
                                                   __bf16 *Gs, int gss) {
    RawHood<NT> rh;
    make_raw_hood<NT>(sg, x, rh);
    // wave-uniform: first element of channel chunk ct of anchor a0 in this cloud's feature block
    const unsigned short *rb = reinterpret_cast<const unsigned short *>(sg.fbase) + (size_t)sg.a0 * A.cin + 16 * ct;
    unsigned fcur[NT][4], fnext[NT][4];
    RkRow<KT> rcur, rnext;
    load_rk_row<KT>(A, sg.a0, x, j, rcur);
    load_f_raw<NT>(rb, rh, fcur);
    const unsigned lane_st = (unsigned)(x * A.ks + 4 * j);
    for (int i = 0; i < sg.cnt; ++i) {
        const int inext = i + 1 < sg.cnt ? i + 1 : i;   // last column re-reads its own rows (cache hit, result unused)
        load_rk_row<KT>(A, sg.a0 + inext, x, j, rnext);
        load_f_raw<NT>(rb + (size_t)inext * A.cin, rh, fnext);
        f32x4 w[KT][NT];
        make_weights_from<NT, KT>(rcur, j, sg.h, w);    // relu already applied
        rcur = rnext;
        bf16x4_t fb4[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb4[t] = pack_f_raw<NT>(rh, fcur, t);
        __bf16 *grow = Gs + (size_t)(sg.jc0 + i) * gss;   // wave-uniform row of G
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if constexpr (NT % 2 == 0) {
#pragma unroll
                for (int t = 0; t < NT; t += 2)
                    g = mfma_bf16_k32(pack4(w[kt][t]), pack4(w[kt][t + 1]), fb4[t], fb4[t + 1], g);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) g = mfma_bf16_k16(pack4(w[kt][t]), fb4[t], g);
            }
            if (16 * kt + 4 * j < A.ks) st4f(grow + 16 * kt + lane_st, g);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) fcur[t][r] = fnext[t][r];
    }
}

template <int NT, int KT, typename TF = float>
__device__ __forceinline__ void group_segment(const InterArgs &A, const Seg<NT> &sg, int ct, int x, int j,
                                              TF *Gs, int gss) {
    if (sg.cnt <= 0) return;
    if constexpr (sizeof(TF) == 2) {
        group_segment_bf16<NT, KT>(A, sg, ct, x, j, Gs, gss);
        return;
    }
    const int coff = 16 * ct + x;
    float fcur[NT][4], fnext[NT][4];
    RkRow<KT> rcur, rnext;
    load_rk_row<KT>(A, sg.a0, x, j, rcur);
    load_f<NT, TF>(A, sg, sg.a0, coff, fcur);
    for (int i = 0; i < sg.cnt; ++i) {
        const int a = sg.a0 + i;
        const int an = i + 1 < sg.cnt ? a + 1 : a;   // last column re-reads its own rows (cache hit, result unused)
        load_rk_row<KT>(A, an, x, j, rnext);
        load_f<NT, TF>(A, sg, an, coff, fnext);
        f32x4 w[KT][NT];
        make_weights_from<NT, KT>(rcur, j, sg.h, w);
        rcur = rnext;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g = mfma4(w[kt][t][r], sg.h.ok[t][r] ? fcur[t][r] : 0.0f, g);
            if (16 * kt + 4 * j < A.ks)
                st4f(Gs + (sg.jc0 + i) * gss + x * A.ks + 16 * kt + 4 * j, g);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) fcur[t][r] = fnext[t][r];
    }
}

template <int NT, typename TF = float>
__device__ __forceinline__ void make_segments(const InterArgs &A, long long col0, int x, int j, Seg<NT> &s0,
                                              Seg<NT> &s1) {
    long long c0 = col0 < A.ncol ? col0 : A.ncol - 1;
    const int pt0 = (int)(c0 / A.na);
    const int a0 = (int)(c0 - (long long)pt0 * A.na);
    long long ncols = A.ncol - col0;
    ncols = ncols > 16 ? 16 : (ncols < 1 ? 1 : ncols);
    const int n0 = A.na - a0 < (int)ncols ? A.na - a0 : (int)ncols;
    int bb = pt0 / A.p2, pp = pt0 - bb * A.p2;
    load_hood<NT>(A, bb, pp, x, j, s0.h);
    s0.fbase = reinterpret_cast<const float *>(reinterpret_cast<const TF *>(A.feats) + ((size_t)bb * A.p1) * A.na * A.cin);
    s0.a0 = a0; s0.jc0 = 0; s0.cnt = n0;
    const int pt1 = pt0 + 1;
    s1.cnt = (int)ncols - n0;
    s1.a0 = 0; s1.jc0 = n0;
    if (s1.cnt > 0) {
        bb = pt1 / A.p2; pp = pt1 - bb * A.p2;
        load_hood<NT>(A, bb, pp, x, j, s1.h);
        s1.fbase = reinterpret_cast<const float *>(reinterpret_cast<const TF *>(A.feats) + ((size_t)bb * A.p1) * A.na * A.cin);
    } else {
        s1.h = s0.h;
        s1.fbase = s0.fbase;
    }
}

}  // namespace
}  // namespace epn
