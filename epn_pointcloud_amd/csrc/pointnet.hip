// PointnetSO3Conv (vgtk/vgtk/so3conv/modules.py:203-235), the aggregation tail of every shipped model:
//   xyzc      = xyz - mean_p(xyz)
//   ext[c..]  = R_a^T xyzc          (einsum 'aji,bjn->bina'; na == 1: xyzc itself)
//   out[b,o,a]= max_p ( sum_c W[o][c] F[b,c,p,a] + sum_j W[o][C+j] ext_j + bias[o] )
// fused into one pass: the concatenated [C+3]-channel tensor and the [b,co,p,a] embedding never exist in HBM.  The
// layer is small (4 GFLOP at the cls head) and runs on the VALU in exact fp32: one workgroup per (cloud, anchor) and
// block of 128 output channels, thread = output channel, W^T and a 32-point feature tile staged in LDS.
// The backward pass routes dOut through the arg-max point (torch.max backward) without materialising the embedding.
#include "conv_internal.h"

namespace epn {
namespace {

// Sequential arg-max step in ascending point order with torch.max's semantics (the reference's `torch.max(dim=2)`,
// so3conv/modules.py:230): the first maximum wins and NaN IS the maximum -- the first NaN sticks, so a diverged activation
// shows up in the head's output instead of being silently dropped (advisor finding, round 4).
__device__ __forceinline__ bool pn_takes(float v, float best) { return v > best || (v != v && best == best); }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int PN_T = 128;    // threads = output channels per workgroup
constexpr int PN_P = 32;     // points per tile
constexpr int PN_C = 64;     // channels per tile

struct PnArgs {
    const float *feats, *xyz, *anchors, *W, *bias, *gout, *centre_in;
    const int32_t *arg_in;
    float *out, *centre, *dfeats, *dW, *dbias;
    int32_t *arg;
    int b, p, a, c, co, slices;
};

// rotated, centred coordinate of point pp for anchor ai (anchors == nullptr: identity)
__device__ __forceinline__ void ext_xyz(const PnArgs &A, int bb, int ai, int pp, const float (&ctr)[3], float (&e)[3]) {
    const float *s = A.xyz + (size_t)bb * 3 * A.p;
    const float v0 = s[pp] - ctr[0], v1 = s[A.p + pp] - ctr[1], v2 = s[2 * A.p + pp] - ctr[2];
    if (A.anchors) {
        const float *R = A.anchors + (size_t)ai * 9;   // e_i = sum_j R[j][i] v_j
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = R[i] * v0 + R[3 + i] * v1 + R[6 + i] * v2;
    } else {
        e[0] = v0; e[1] = v1; e[2] = v2;
    }
}

__global__ __launch_bounds__(PN_T) void pointnet_fwd_kernel(PnArgs A) {
    __shared__ float Ws[PN_C][PN_T + 1];                          // W^T tile: [channel][output channel]
    __shared__ __attribute__((aligned(16))) float Fs[PN_P][PN_C + 4];
    __shared__ float Es[PN_P][4];
    __shared__ float ctr_s[3];
    const int t = threadIdx.x;
    const int bb = blockIdx.x / A.a, ai = blockIdx.x % A.a;
    const int o = blockIdx.y * PN_T + t;
    const bool o_ok = o < A.co;
    const int ce = A.c + 3;

    if (t < 3) {   // centre of the cloud: plain sequential mean (every workgroup of this cloud gets the same bits)
        const float *s = A.xyz + ((size_t)bb * 3 + t) * A.p;
        float m = 0.f;
        for (int i = 0; i < A.p; ++i) m += s[i];
        m /= (float)A.p;
        ctr_s[t] = m;
        if (ai == 0 && blockIdx.y == 0) A.centre[bb * 3 + t] = m;
    }
    __syncthreads();
    const float ctr[3] = {ctr_s[0], ctr_s[1], ctr_s[2]};
    const float w3[3] = {o_ok ? A.W[(size_t)o * ce + A.c] : 0.f, o_ok ? A.W[(size_t)o * ce + A.c + 1] : 0.f,
                         o_ok ? A.W[(size_t)o * ce + A.c + 2] : 0.f};
    const float bias = (o_ok && A.bias) ? A.bias[o] : 0.f;

    float best = -INFINITY;
    int best_p = 0;
    for (int p0 = 0; p0 < A.p; p0 += PN_P) {
        __syncthreads();
        if (t < PN_P) {
            float e[3] = {0.f, 0.f, 0.f};
            if (p0 + t < A.p) ext_xyz(A, bb, ai, p0 + t, ctr, e);
            Es[t][0] = e[0]; Es[t][1] = e[1]; Es[t][2] = e[2];
        }
        __syncthreads();
        float acc[PN_P];
#pragma unroll
        for (int i = 0; i < PN_P; ++i) acc[i] = bias + w3[0] * Es[i][0] + w3[1] * Es[i][1] + w3[2] * Es[i][2];
        for (int c0 = 0; c0 < A.c; c0 += PN_C) {
            __syncthreads();
            // W^T tile and feature tile: consecutive threads read consecutive channels (coalesced), the padded LDS
            // rows keep the transposed writes conflict-free
            for (int e = t; e < PN_C * PN_T; e += PN_T) {
                const int cl = e % PN_C, j = e / PN_C;
                const int cc = c0 + cl, oo = blockIdx.y * PN_T + j;
                Ws[cl][j] = (cc < A.c && oo < A.co) ? A.W[(size_t)oo * ce + cc] : 0.f;
            }
            for (int e = t; e < PN_C * PN_P; e += PN_T) {
                const int cl = e % PN_C, i = e / PN_C;
                const int cc = c0 + cl, pp = p0 + i;
                Fs[i][cl] = (cc < A.c && pp < A.p) ? A.feats[(((size_t)bb * A.p + pp) * A.a + ai) * A.c + cc] : 0.f;
            }
            __syncthreads();
            for (int c4 = 0; c4 < PN_C; c4 += 4) {
                const float w0 = Ws[c4][t], w1 = Ws[c4 + 1][t], w2 = Ws[c4 + 2][t], w3c = Ws[c4 + 3][t];
#pragma unroll
                for (int i = 0; i < PN_P; ++i) {
                    const f32x4 f = *reinterpret_cast<const f32x4 *>(&Fs[i][c4]);   // LDS broadcast
                    acc[i] += w0 * f[0] + w1 * f[1] + w2 * f[2] + w3c * f[3];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PN_P; ++i)
            if (p0 + i < A.p && pn_takes(acc[i], best)) { best = acc[i]; best_p = p0 + i; }   // first maximum wins
    }
    if (o_ok) {
        const size_t at = ((size_t)bb * A.a + ai) * A.co + o;
        A.out[at] = best;
        A.arg[at] = best_p;
    }
}

// MFMA form of the forward pass (c % 16 == 0): workgroup = (cloud, anchor) x 128 output channels, 4 waves x 32 channels.
// The [64 points x c] feature slab is the A operand straight from global memory (one 16-byte load per lane covers the
// four contraction steps k = 16s + 4j + t), W^T comes through LDS in 64-channel chunks with the same k mapping, the
// three coordinate channels and the bias are added on the VALU, and the max / arg-max over points is a register scan
// plus two cross-lane steps.  Same result as the VALU kernel up to fp32 summation order.
__global__ __launch_bounds__(256) void pointnet_fwd_mfma_kernel(PnArgs A) {
    constexpr int KC = 64, WLD = KC + 4;
    __shared__ __attribute__((aligned(16))) float Ws[128 * WLD];     // [output channel][channel of the chunk]
    __shared__ float Es[64][4];
    __shared__ float ctr_s[3];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int x = lane & 15, j = lane >> 4;
    const int bb = blockIdx.x / A.a, ai = blockIdx.x % A.a;
    const int co0 = blockIdx.y * 128;
    const int ce = A.c + 3;
    if (t < 3) {
        const float *s = A.xyz + ((size_t)bb * 3 + t) * A.p;
        float m = 0.f;
        for (int i = 0; i < A.p; ++i) m += s[i];
        m /= (float)A.p;
        ctr_s[t] = m;
        if (ai == 0 && blockIdx.y == 0) A.centre[bb * 3 + t] = m;
    }
    __syncthreads();
    const float ctr[3] = {ctr_s[0], ctr_s[1], ctr_s[2]};
    // this lane's output channels: co0 + 32 wave + 16 nt + x
    float w3[2][3], bias[2];
    bool cok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int o = co0 + 32 * wave + 16 * nt + x;
        cok[nt] = o < A.co;
        const int oo = cok[nt] ? o : 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) w3[nt][u] = cok[nt] ? A.W[(size_t)oo * ce + A.c + u] : 0.f;
        bias[nt] = (cok[nt] && A.bias) ? A.bias[oo] : 0.f;
    }
    float best[2] = {-INFINITY, -INFINITY};
    int bestp[2] = {0, 0};
    const float *fb = A.feats + ((size_t)bb * A.p * A.a + ai) * A.c;      // + p * a * c
    for (int p0 = 0; p0 < A.p; p0 += 64) {
        __syncthreads();
        if (t < 64) {
            float e[3] = {0.f, 0.f, 0.f};
            if (p0 + t < A.p) ext_xyz(A, bb, ai, p0 + t, ctr, e);
            Es[t][0] = e[0]; Es[t][1] = e[1]; Es[t][2] = e[2];
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *frow[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int pp = p0 + 16 * mt + x;
            pp = pp < A.p ? pp : A.p - 1;                                   // masked below (acc -> -inf)
            frow[mt] = fb + (size_t)pp * A.a * A.c + 4 * j;
        }
        for (int c0 = 0; c0 < A.c; c0 += KC) {
            const int cc = min(KC, A.c - c0);                                // multiple of 16 (launcher)
            __syncthreads();
            for (int e = t; e < 128 * KC; e += 256) {
                const int cl = e % KC, ol = e / KC;
                const int o = co0 + ol;
                Ws[ol * WLD + cl] = (cl < cc && o < A.co) ? A.W[(size_t)o * ce + c0 + cl] : 0.f;
            }
            __syncthreads();
            // the feature fragments of step s + 1 are requested before the MFMAs of step s (round 4): loaded where they are
            // used, every 16-channel step of every wave waited out an L2 / HBM round trip (64 steps per workgroup)
            f32x4 afn[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) afn[mt] = *reinterpret_cast<const f32x4 *>(frow[mt] + c0);
            for (int s16 = 0; s16 < cc; s16 += 16) {
                f32x4 af[4], bf[2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) af[mt] = afn[mt];
                const int sn = s16 + 16 < cc ? s16 + 16 : s16;               // last step re-reads its own fragment (unused)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) afn[mt] = *reinterpret_cast<const f32x4 *>(frow[mt] + c0 + sn);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    bf[nt] = *reinterpret_cast<const f32x4 *>(Ws + (32 * wave + 16 * nt + x) * WLD + s16 + 4 * j);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][u], bf[nt][u], acc[mt][nt], 0, 0, 0);
            }
        }
        // acc[mt][nt][r]: point p0 + 16 mt + 4 j + r, channel co0 + 32 wave + 16 nt + x
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = 16 * mt + 4 * j + r;
                const float e0 = Es[pl][0], e1 = Es[pl][1], e2 = Es[pl][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float v = acc[mt][nt][r] + bias[nt] + w3[nt][0] * e0 + w3[nt][1] * e1 + w3[nt][2] * e2;
                    if (p0 + pl < A.p && pn_takes(v, best[nt])) { best[nt] = v; bestp[nt] = p0 + pl; }   // ascending p: first max
                }
            }
    }
    // combine the four lane groups j (disjoint point sets): larger value wins, ties go to the smaller point index
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int d = 16; d < 64; d <<= 1) {
            const float ov = __shfl_xor(best[nt], d, 64);
            const int op = __shfl_xor(bestp[nt], d, 64);
            const bool on = ov != ov, bn = best[nt] != best[nt];       // NaN is the maximum (torch.max), the first one wins
            if ((on || bn) ? (on && (!bn || op < bestp[nt])) : (ov > best[nt] || (ov == best[nt] && op < bestp[nt]))) {
                best[nt] = ov; bestp[nt] = op;
            }
        }
        if (j == 0 && cok[nt]) {
            const size_t at = ((size_t)bb * A.a + ai) * A.co + co0 + 32 * wave + 16 * nt + x;
            A.out[at] = best[nt];
            A.arg[at] = bestp[nt];
        }
    }
}

// dF[b,p,a,c] = sum_{o : arg[b,a,o] == p} dOut[b,a,o] * W[o][c];  workgroup = (cloud, anchor) x 128-channel block,
// thread = channel, the point tile accumulates in LDS in output-channel order (deterministic, no atomics).
__global__ __launch_bounds__(PN_T) void pointnet_bwd_data_kernel(PnArgs A) {
    constexpr int PT = 64;
    __shared__ float dFs[PT][PN_T];
    const int t = threadIdx.x;
    const int bb = blockIdx.x / A.a, ai = blockIdx.x % A.a;
    const int cc = blockIdx.y * PN_T + t;
    const int ce = A.c + 3;
    const size_t at = ((size_t)bb * A.a + ai) * A.co;
    for (int p0 = 0; p0 < A.p; p0 += PT) {
        for (int i = 0; i < PT; ++i) dFs[i][t] = 0.f;
        // only this thread touches column t: no barrier needed
        // four output channels per step: their arg-max, gradient and weight loads are independent (issued together), only
        // the LDS read-modify-writes of the point tile stay in order (deterministic: output-channel order)
        int o = 0;
        for (; o + 4 <= A.co; o += 4) {
            int ps[4];
            float gw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ps[u] = A.arg_in[at + o + u] - p0;          // wave-uniform
                gw[u] = A.gout[at + o + u] * (cc < A.c ? A.W[(size_t)(o + u) * ce + cc] : 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ps[u] >= 0 && ps[u] < PT) dFs[ps[u]][t] += gw[u];
        }
        for (; o < A.co; ++o) {
            const int ps = A.arg_in[at + o] - p0;
            if (ps < 0 || ps >= PT) continue;
            const float g = A.gout[at + o];
            if (cc < A.c) dFs[ps][t] += g * A.W[(size_t)o * ce + cc];
        }
        if (cc < A.c)
            for (int i = 0; i < PT && p0 + i < A.p; ++i)
                A.dfeats[(((size_t)bb * A.p + p0 + i) * A.a + ai) * A.c + cc] = dFs[i][t];
    }
}

// dW[o][ce] = sum_{b,a} dOut[b,a,o] * ext_feature[b, arg[b,a,o], a, ce];  dbias[o] = sum_{b,a} dOut[b,a,o].
// grid = (output channel, slice of the (b,a) range); thread = (extended) input channel; one atomic per slice.
__global__ __launch_bounds__(256) void pointnet_bwd_weight_kernel(PnArgs A) {
    const int o = blockIdx.x;
    const int ce = A.c + 3;
    const long long nba = (long long)A.b * A.a;
    const long long per = (nba + A.slices - 1) / A.slices;
    const long long q0 = blockIdx.y * per;
    long long q1 = q0 + per;
    q1 = q1 < nba ? q1 : nba;
    // (one block of ce rounded up to whole waves -- the three coordinate channels in a wave beside the feature channels instead
    // of a second pass with 3 live threads -- measured SLOWER: 0.31 -> 0.36 ms per call, fewer workgroups per CU)
    for (int e0 = 0; e0 < ce; e0 += 256) {
        const int e = e0 + threadIdx.x;
        if (e >= ce) continue;                         // (the last pass has 3 live threads: the coordinate channels)
        float acc = 0.f, accb = 0.f;
        auto term = [&](long long q, float g, int ps, float f) {
            accb += g;
            if (e < A.c) {
                acc += g * f;
            } else {
                const int bb = (int)(q / A.a), ai = (int)(q % A.a);
                const float ctr[3] = {A.centre_in[bb * 3], A.centre_in[bb * 3 + 1], A.centre_in[bb * 3 + 2]};
                float x3[3];
                ext_xyz(A, bb, ai, ps, ctr, x3);
                acc += g * x3[e - A.c];
            }
        };
        // eight (gradient, arg-max, gathered feature) triples in flight per thread: the loop was a chain of dependent
        // load latencies (arg-max -> feature row), ~120 of them per thread
        long long q = q0;
        for (; q + 8 <= q1; q += 8) {
            float g[8], f[8];
            int ps[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                g[u] = A.gout[(q + u) * A.co + o];
                ps[u] = A.arg_in[(q + u) * A.co + o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = (int)((q + u) / A.a), ai = (int)((q + u) % A.a);
                f[u] = e < A.c ? A.feats[(((size_t)bb * A.p + ps[u]) * A.a + ai) * A.c + e] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) term(q + u, g[u], ps[u], f[u]);
        }
        for (; q < q1; ++q) {
            const int bb = (int)(q / A.a), ai = (int)(q % A.a);
            const float g = A.gout[q * A.co + o];
            const int ps = A.arg_in[q * A.co + o];
            term(q, g, ps, e < A.c ? A.feats[(((size_t)bb * A.p + ps) * A.a + ai) * A.c + e] : 0.0f);
        }
        atomicAdd(A.dW + (size_t)o * ce + e, acc);
        if (e == 0 && A.dbias) atomicAdd(A.dbias + o, accb);
    }
}

// ---- GEMM-composed form (round 4).  The embedding's feature part Z[b][p][a][o] = sum_c W[o][c] F[b][p][a][c] is a plain
// [b p a, c] x [c, co] contraction: on the library's GEMM (bf16 pipe: bf16 features directly, fp32 ones in the split form) it
// runs 3-4x faster than the fused kernel above, whose workgroups each restage W and whose four waves each load the whole
// feature slab (40 TFLOP/s).  What is left is streaming: the max / arg-max over points with the three coordinate channels
// and the bias added on the way, and in the backward pass the gradient routed through the arg-max point written as the
// (one non-zero per column) dense dZ that two more GEMMs turn into dF and dW.
struct PnMaxArgs {
    const float *Z, *xyz, *anchors, *W, *bias, *gout, *centre_in;
    const int32_t *arg_in;
    float *out, *centre, *dW, *dbias;
    int32_t *arg;
    void *dZ;
    int b, p, a, c, co;
};

__global__ __launch_bounds__(256) void pointnet_max_kernel(PnMaxArgs A) {
    constexpr int PT = 64;
    __shared__ float Es[PT][4];
    __shared__ float ctr_s[3];
    const int t = threadIdx.x;
    const int bb = blockIdx.x / A.a, ai = blockIdx.x % A.a;
    const int o = blockIdx.y * 256 + t;
    const bool ok = o < A.co;
    const int ce = A.c + 3;
    if (t < 3) {   // centre of the cloud: plain sequential mean, as the fused kernels (every workgroup gets the same bits)
        const float *s = A.xyz + ((size_t)bb * 3 + t) * A.p;
        float m = 0.f;
        for (int i = 0; i < A.p; ++i) m += s[i];
        m /= (float)A.p;
        ctr_s[t] = m;
        if (ai == 0 && blockIdx.y == 0) A.centre[bb * 3 + t] = m;
    }
    __syncthreads();
    const float ctr[3] = {ctr_s[0], ctr_s[1], ctr_s[2]};
    const int oo = ok ? o : 0;
    const float w3[3] = {A.W[(size_t)oo * ce + A.c], A.W[(size_t)oo * ce + A.c + 1], A.W[(size_t)oo * ce + A.c + 2]};
    const float bias = A.bias ? A.bias[oo] : 0.f;
    const float *z = A.Z + ((size_t)bb * A.p * A.a + ai) * A.co + oo;        // + p * a * co
    const size_t zs = (size_t)A.a * A.co;
    float best = -INFINITY;
    int bestp = 0;
    for (int p0 = 0; p0 < A.p; p0 += PT) {
        __syncthreads();
        if (t < PT) {
            float e[3] = {0.f, 0.f, 0.f};
            if (p0 + t < A.p) {
                PnArgs X = {};
                X.xyz = A.xyz; X.anchors = A.anchors; X.p = A.p;
                ext_xyz(X, bb, ai, p0 + t, ctr, e);
            }
            Es[t][0] = e[0]; Es[t][1] = e[1]; Es[t][2] = e[2];
        }
        __syncthreads();
        const int np = min(PT, A.p - p0);
        int i = 0;
        for (; i + 8 <= np; i += 8) {          // eight rows in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = z[(size_t)(p0 + i + u) * zs];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w = v[u] + bias + w3[0] * Es[i + u][0] + w3[1] * Es[i + u][1] + w3[2] * Es[i + u][2];
                if (pn_takes(w, best)) { best = w; bestp = p0 + i + u; }     // ascending p: first maximum wins
            }
        }
        for (; i < np; ++i) {
            const float w = z[(size_t)(p0 + i) * zs] + bias + w3[0] * Es[i][0] + w3[1] * Es[i][1] + w3[2] * Es[i][2];
            if (pn_takes(w, best)) { best = w; bestp = p0 + i; }
        }
    }
    if (ok) {
        const size_t at = ((size_t)bb * A.a + ai) * A.co + o;
        A.out[at] = best;
        A.arg[at] = bestp;
    }
}

// dZ[b][p][a][o] = arg[b][a][o] == p ? dOut[b][a][o] : 0 -- every element written (no memset), eight channels per thread
template <typename T>
__global__ __launch_bounds__(256) void pointnet_dz_kernel(PnMaxArgs A) {
    const long long oct = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co8 = A.co >> 3;
    const long long n = (long long)A.b * A.p * A.a * co8;
    if (oct >= n) return;
    const int o = (int)(oct % co8) * 8;
    const long long row = oct / co8;               // (b, p, a)
    const int ai = (int)(row % A.a);
    const long long bp = row / A.a;
    const int pp = (int)(bp % A.p), bb = (int)(bp / A.p);
    const size_t at = ((size_t)bb * A.a + ai) * A.co + o;
    const i32x4 a0 = *reinterpret_cast<const i32x4 *>(A.arg_in + at), a1 = *reinterpret_cast<const i32x4 *>(A.arg_in + at + 4);
    const f32x4 g0 = *reinterpret_cast<const f32x4 *>(A.gout + at), g1 = *reinterpret_cast<const f32x4 *>(A.gout + at + 4);
    float v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v[u] = a0[u] == pp ? g0[u] : 0.f;
        v[4 + u] = a1[u] == pp ? g1[u] : 0.f;
    }
    T *d = static_cast<T *>(A.dZ) + (size_t)row * A.co + o;
    if constexpr (sizeof(T) == 2) {
        bf16x8 h;
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = (__bf16)v[u];
        *reinterpret_cast<bf16x8 *>(d) = h;
    } else {
        *reinterpret_cast<f32x4 *>(d) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// dW[o][c + j] = sum_{b,a} dOut[b,a,o] ext_j[b, arg[b,a,o], a], dbias[o] = sum_{b,a} dOut[b,a,o]: one workgroup per output
// channel, fixed-order tree over its 256 partial sums (deterministic)
__global__ __launch_bounds__(256) void pointnet_bwd_coord_kernel(PnMaxArgs A) {
    __shared__ float red[4][256];
    const int o = blockIdx.x, t = threadIdx.x;
    const long long nba = (long long)A.b * A.a;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    PnArgs X = {};
    X.xyz = A.xyz; X.anchors = A.anchors; X.p = A.p;
    for (long long q = t; q < nba; q += 256) {
        const int bb = (int)(q / A.a), ai = (int)(q % A.a);
        const float g = A.gout[q * A.co + o];
        const int ps = A.arg_in[q * A.co + o];
        const float ctr[3] = {A.centre_in[bb * 3], A.centre_in[bb * 3 + 1], A.centre_in[bb * 3 + 2]};
        float e[3];
        ext_xyz(X, bb, ai, ps, ctr, e);
        acc[0] += g * e[0]; acc[1] += g * e[1]; acc[2] += g * e[2]; acc[3] += g;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) red[u][t] = acc[u];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[u][t] += red[u][t + s];
        __syncthreads();
    }
    if (t < 3) A.dW[(size_t)o * (A.c + 3) + A.c + t] = red[t][0];
    if (t == 3 && A.dbias) A.dbias[o] = red[3][0];
}

PnArgs make_pn(int b, int p, int a, int c, int co) {
    PnArgs A = {};
    A.b = b; A.p = p; A.a = a; A.c = c; A.co = co; A.slices = 1;
    return A;
}

}  // namespace
}  // namespace epn

using namespace epn;

static int check_pn(int b, int p, int a, int c, int co) {
    if (b < 0 || p < 1 || a < 1 || c < 1 || co < 1) return EPN_EINVAL;
    if ((long long)b * p * a * c >= (1LL << 40) || (long long)b * a > 0x7fffffffLL) return EPN_EINVAL;
    return 0;
}

extern "C" int epn_pointnet_so3conv_fwd_f32(const float *feats_cl, const float *xyz, const float *anchors,
                                            const float *W, const float *bias, float *out, int32_t *argmax,
                                            float *centre, int b, int p, int a, int c, int co,
                                            epn_stream_t stream) {
    int rc = check_pn(b, p, a, c, co);
    if (rc) return rc;
    if (b == 0) return 0;
    if (!feats_cl || !xyz || !W || !out || !argmax || !centre) return EPN_ENULL;
    PnArgs A = make_pn(b, p, a, c, co);
    A.feats = feats_cl; A.xyz = xyz; A.anchors = anchors; A.W = W; A.bias = bias; A.out = out; A.arg = argmax;
    A.centre = centre;
    if (c % 16 == 0)
        EPN_LAUNCH(pointnet_fwd_mfma_kernel, dim3((unsigned)(b * a), (unsigned)epn_cdiv(co, 128)), dim3(256), 0,
                           epn_stream(stream), A);
    else
        EPN_LAUNCH(pointnet_fwd_kernel, dim3((unsigned)(b * a), (unsigned)epn_cdiv(co, PN_T)), dim3(PN_T), 0,
                           epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_pointnet_so3conv_bwd_data_f32(const float *grad_out, const int32_t *argmax, const float *W,
                                                 float *grad_feats_cl, int b, int p, int a, int c, int co,
                                                 epn_stream_t stream) {
    int rc = check_pn(b, p, a, c, co);
    if (rc) return rc;
    if (b == 0) return 0;
    if (!grad_out || !argmax || !W || !grad_feats_cl) return EPN_ENULL;
    PnArgs A = make_pn(b, p, a, c, co);
    A.gout = grad_out; A.arg_in = argmax; A.W = W; A.dfeats = grad_feats_cl;
    EPN_LAUNCH(pointnet_bwd_data_kernel, dim3((unsigned)(b * a), (unsigned)epn_cdiv(c, PN_T)), dim3(PN_T), 0,
                       epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_pointnet_so3conv_bwd_weight_f32(const float *grad_out, const int32_t *argmax, const float *feats_cl,
                                                   const float *xyz, const float *anchors, const float *centre,
                                                   float *grad_W, float *grad_bias, int b, int p, int a, int c, int co,
                                                   epn_stream_t stream) {
    int rc = check_pn(b, p, a, c, co);
    if (rc) return rc;
    if (!grad_W) return EPN_ENULL;
    hipStream_t st = epn_stream(stream);
    EPN_HIP(hipMemsetAsync(grad_W, 0, (size_t)co * (c + 3) * sizeof(float), st));
    if (grad_bias) EPN_HIP(hipMemsetAsync(grad_bias, 0, (size_t)co * sizeof(float), st));
    if (b == 0) return 0;
    if (!grad_out || !argmax || !feats_cl || !xyz || !centre) return EPN_ENULL;
    PnArgs A = make_pn(b, p, a, c, co);
    A.gout = grad_out; A.arg_in = argmax; A.feats = feats_cl; A.xyz = xyz; A.anchors = anchors; A.centre_in = centre;
    A.dW = grad_W; A.dbias = grad_bias;
    const long long nba = (long long)b * a;
    A.slices = (int)(nba < 16 ? nba : 16);
    EPN_LAUNCH(pointnet_bwd_weight_kernel, dim3((unsigned)co, (unsigned)A.slices), dim3(256), 0, st, A);
    EPN_CHECK_LAUNCH();
    return 0;
}

static PnMaxArgs make_pnmax(int b, int p, int a, int c, int co) {
    PnMaxArgs A = {};
    A.b = b; A.p = p; A.a = a; A.c = c; A.co = co;
    return A;
}

extern "C" int epn_pointnet_max_f32(const float *Z, const float *xyz, const float *anchors, const float *W,
                                    const float *bias, float *out, int32_t *argmax, float *centre, int b, int p, int a,
                                    int c, int co, epn_stream_t stream) {
    int rc = check_pn(b, p, a, c, co);
    if (rc) return rc;
    if (b == 0) return 0;
    if (!Z || !xyz || !W || !out || !argmax || !centre) return EPN_ENULL;
    PnMaxArgs A = make_pnmax(b, p, a, c, co);
    A.Z = Z; A.xyz = xyz; A.anchors = anchors; A.W = W; A.bias = bias; A.out = out; A.arg = argmax; A.centre = centre;
    EPN_LAUNCH(pointnet_max_kernel, dim3((unsigned)(b * a), (unsigned)epn_cdiv(co, 256)), dim3(256), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}

static int pn_dz(const float *grad_out, const int32_t *argmax, void *dZ, int b, int p, int a, int co, bool bf16,
                 epn_stream_t stream) {
    int rc = check_pn(b, p, a, 1, co);
    if (rc) return rc;
    if (co % 8) return EPN_EINVAL;
    if (b == 0) return 0;
    if (!grad_out || !argmax || !dZ) return EPN_ENULL;
    if (((uintptr_t)grad_out | (uintptr_t)argmax | (uintptr_t)dZ) & 15) return EPN_EINVAL;
    PnMaxArgs A = make_pnmax(b, p, a, 0, co);
    A.gout = grad_out; A.arg_in = argmax; A.dZ = dZ;
    const long long n = (long long)b * p * a * (co / 8);
    if ((n + 255) / 256 > 0x7fffffffLL) return EPN_EINVAL;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (bf16) EPN_LAUNCH(pointnet_dz_kernel<__bf16>, grid, dim3(256), 0, epn_stream(stream), A);
    else EPN_LAUNCH(pointnet_dz_kernel<float>, grid, dim3(256), 0, epn_stream(stream), A);
    EPN_CHECK_LAUNCH();
    return 0;
}
extern "C" int epn_pointnet_dz_f32(const float *grad_out, const int32_t *argmax, float *dZ, int b, int p, int a, int co,
                                   epn_stream_t stream) {
    return pn_dz(grad_out, argmax, dZ, b, p, a, co, false, stream);
}
extern "C" int epn_pointnet_dz_bf16(const float *grad_out, const int32_t *argmax, void *dZ, int b, int p, int a, int co,
                                    epn_stream_t stream) {
    return pn_dz(grad_out, argmax, dZ, b, p, a, co, true, stream);
}

extern "C" int epn_pointnet_bwd_coord_f32(const float *grad_out, const int32_t *argmax, const float *xyz,
                                          const float *anchors, const float *centre, float *grad_W, float *grad_bias,
                                          int b, int p, int a, int c, int co, epn_stream_t stream) {
    int rc = check_pn(b, p, a, c, co);
    if (rc) return rc;
    if (!grad_W) return EPN_ENULL;
    if (b > 0 && (!grad_out || !argmax || !xyz || !centre)) return EPN_ENULL;
    PnMaxArgs A = make_pnmax(b, p, a, c, co);
    A.gout = grad_out; A.arg_in = argmax; A.xyz = xyz; A.anchors = anchors; A.centre_in = centre; A.dW = grad_W;
    A.dbias = grad_bias;
    EPN_LAUNCH(pointnet_bwd_coord_kernel, dim3((unsigned)co), dim3(256), 0, epn_stream(stream), A);   // b == 0: writes zeros
    EPN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention pooling over the anchors of the 3DMatch head (InvOutBlockMVD.forward, SPConvNets/utils/base_so3conv.py:598-606):
//     attn = softmax(logits, dim = anchors);   pooled[b, c, p] = sum_a feats[b, c, p, a] * attn[b, c, p, a]
// on channels-last rows [row = (b, p)][a][c].  Through torch this was a strided softmax (three contiguous copies), four
// multiplications and three reductions of a [64, 128, 64, 60] tensor per step: 2.3 ms of the 35 ms 3DMatch step.  One thread
// per (row, channel): the na logits stay in registers (one read of each input), lanes run along the channels (coalesced).
namespace epn {
namespace {
constexpr int ASP_NA_MAX = 64;

template <int NA_T>   // NA_T = 60: the loops unroll; 0: any na <= ASP_NA_MAX
__global__ __launch_bounds__(256) void anchor_softmax_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ logits,
                                                                      float *__restrict__ attn, float *__restrict__ pooled,
                                                                      long long rows, int na_rt, int c) {
    const int na = NA_T ? NA_T : na_rt;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (row, channel)
    if (i >= rows * c) return;
    const long long row = i / c;
    const int ch = (int)(i - row * c);
    const size_t base = (size_t)row * na * c + ch;
    float l[NA_T ? NA_T : ASP_NA_MAX];
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < (NA_T ? NA_T : ASP_NA_MAX); ++a)
        if (a < na) {
            l[a] = logits[base + (size_t)a * c];
            m = fmaxf(m, l[a]);
        }
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < (NA_T ? NA_T : ASP_NA_MAX); ++a)
        if (a < na) {
            l[a] = __expf(l[a] - m);
            sum += l[a];
        }
    const float inv = 1.0f / sum;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < (NA_T ? NA_T : ASP_NA_MAX); ++a)
        if (a < na) {
            const float w = l[a] * inv;
            attn[base + (size_t)a * c] = w;
            acc += x[base + (size_t)a * c] * w;
        }
    pooled[i] = acc;
}

// dx = dpooled * attn;  dlogits = attn * (g - sum_a attn g),  g = dpooled * x (+ dattn)
template <int NA_T>
__global__ __launch_bounds__(256) void anchor_softmax_pool_bwd_kernel(const float *__restrict__ x, const float *__restrict__ attn,
                                                                      const float *__restrict__ dpooled,
                                                                      const float *__restrict__ dattn, float *__restrict__ dx,
                                                                      float *__restrict__ dlogits, long long rows, int na_rt, int c) {
    const int na = NA_T ? NA_T : na_rt;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c) return;
    const long long row = i / c;
    const int ch = (int)(i - row * c);
    const size_t base = (size_t)row * na * c + ch;
    const float dp = dpooled ? dpooled[i] : 0.f;
    float w[NA_T ? NA_T : ASP_NA_MAX], g[NA_T ? NA_T : ASP_NA_MAX];
    float dot = 0.f;
#pragma unroll
    for (int a = 0; a < (NA_T ? NA_T : ASP_NA_MAX); ++a)
        if (a < na) {
            w[a] = attn[base + (size_t)a * c];
            g[a] = dp * x[base + (size_t)a * c];
            if (dattn) g[a] += dattn[base + (size_t)a * c];
            dot += w[a] * g[a];
        }
#pragma unroll
    for (int a = 0; a < (NA_T ? NA_T : ASP_NA_MAX); ++a)
        if (a < na) {
            if (dx) dx[base + (size_t)a * c] = dp * w[a];
            dlogits[base + (size_t)a * c] = w[a] * (g[a] - dot);
        }
}
}  // namespace
}  // namespace epn

extern "C" int epn_anchor_softmax_pool_fwd_f32(const float *feats_cl, const float *logits_cl, float *attn_cl, float *pooled,
                                               long long rows, int na, int c, epn_stream_t stream) {
    if (rows < 0 || na < 1 || na > epn::ASP_NA_MAX || c < 1) return EPN_EINVAL;
    if (rows == 0) return 0;
    if (!feats_cl || !logits_cl || !attn_cl || !pooled) return EPN_ENULL;
    const long long n = rows * c;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (na == 60)
        EPN_LAUNCH(epn::anchor_softmax_pool_fwd_kernel<60>, grid, blk, 0, epn_stream(stream), feats_cl, logits_cl, attn_cl, pooled, rows, na, c);
    else
        EPN_LAUNCH(epn::anchor_softmax_pool_fwd_kernel<0>, grid, blk, 0, epn_stream(stream), feats_cl, logits_cl, attn_cl, pooled, rows, na, c);
    EPN_CHECK_LAUNCH();
    return 0;
}

extern "C" int epn_anchor_softmax_pool_bwd_f32(const float *feats_cl, const float *attn_cl, const float *grad_pooled,
                                               const float *grad_attn_cl, float *grad_feats_cl, float *grad_logits_cl,
                                               long long rows, int na, int c, epn_stream_t stream) {
    if (rows < 0 || na < 1 || na > epn::ASP_NA_MAX || c < 1) return EPN_EINVAL;
    if (rows == 0) return 0;
    if (!feats_cl || !attn_cl || !grad_logits_cl || (!grad_pooled && !grad_attn_cl)) return EPN_ENULL;
    const long long n = rows * c;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (na == 60)
        EPN_LAUNCH(epn::anchor_softmax_pool_bwd_kernel<60>, grid, blk, 0, epn_stream(stream), feats_cl, attn_cl, grad_pooled, grad_attn_cl,
                   grad_feats_cl, grad_logits_cl, rows, na, c);
    else
        EPN_LAUNCH(epn::anchor_softmax_pool_bwd_kernel<0>, grid, blk, 0, epn_stream(stream), feats_cl, attn_cl, grad_pooled, grad_attn_cl,
                   grad_feats_cl, grad_logits_cl, rows, na, c);
    EPN_CHECK_LAUNCH();
    return 0;
}
