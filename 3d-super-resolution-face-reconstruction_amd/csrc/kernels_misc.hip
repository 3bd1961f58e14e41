// Everything of the SR3 denoise step that is not a convolution: GroupNorm statistics, the
// self-attention core, the noise-level embedding, layout changes and the fused DDPM update.
// All fp32, NHWC activations. gfx950 only (64-lane wavefronts).
#include "sr3_internal.h"
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include <algorithm>

namespace sr3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// =================================================================================================
// GroupNorm statistics -> folded per-(image, channel) affine
//   reference: nn.GroupNorm(groups, C) inside Block / SelfAttention, unet.py:84,119 (eps 1e-5,
//   biased variance over (C/groups)*H*W, groups of contiguous channels of the concatenated input).
// Kernel 1 streams each image slice with 16-byte loads (a thread owns one channel quad, lanes run
// along the channel axis = coalesced rows) and accumulates sum and sum-of-squares per channel in
// fp64 (E[x^2]-mean^2 is then safe: 2^-53 relative), writes per-(image, slice, channel) partials.
// Kernel 2 adds the slices and the channels of each group (any group size, groups may straddle the
// x/skip boundary of a concatenation) and writes scale = rstd*gamma, shift = beta - mean*scale.
// Deterministic: no atomics.
// =================================================================================================
namespace {

constexpr int GN_MAX_SLICES = 64;

__global__ __launch_bounds__(256) void gn_partial_kernel(const TDesc in0, const TDesc in1, int slices,
                                                         double *__restrict__ part) {
    __shared__ double ssum[256][4], ssq[256][4];
    const int C0 = in0.C, C1 = in1.p ? in1.C : 0;
    const int C = C0 + C1, C4 = C >> 2;
    const int W = in0.W, HW = in0.H * in0.W;
    const int slice = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
    const int per = (HW + slices - 1) / slices;
    const int p0 = slice * per;
    const int p1 = min(HW, p0 + per);
    // channel quads are processed in passes of QP quads; threads (pl, q) with q = t % QP
    const int QP = min(C4, 256);
    for (int qbase = 0; qbase < C4; qbase += QP) {
        const int qw = min(QP, C4 - qbase);
        const int plw = 256 / qw;
        const int q = qbase + (t % qw);
        const int pl = t / qw;
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (pl < plw) {
            const int c = q << 2;
            const TDesc &d = (c < C0) ? in0 : in1;
            const int cl = (c < C0) ? c : c - C0;
            const size_t cs = d.C;
            const int Wp = d.Wp();
            int pp = p0 + pl;
            int y = pp / W, x = pp - y * W;
            const float *src = d.p + d.pix(n, 0, 0) * cs + cl;   // interior origin of image n
#pragma unroll 4
            for (; pp < p1; pp += plw) {
                const float4 v = *reinterpret_cast<const float4 *>(src + ((size_t)y * Wp + x) * cs);
                s[0] += v.x; ss[0] = fma((double)v.x, (double)v.x, ss[0]);
                s[1] += v.y; ss[1] = fma((double)v.y, (double)v.y, ss[1]);
                s[2] += v.z; ss[2] = fma((double)v.z, (double)v.z, ss[2]);
                s[3] += v.w; ss[3] = fma((double)v.w, (double)v.w, ss[3]);
                x += plw;
                while (x >= W) { x -= W; ++y; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[t][j] = s[j]; ssq[t][j] = ss[j]; }
        __syncthreads();
        // thread (q', j) adds the pixel lanes of channel 4*(qbase+q') + j
        for (int i = t; i < qw * 4; i += 256) {
            const int qq = i >> 2, j = i & 3;
            double a = 0, b = 0;
            for (int l = 0; l < plw; ++l) { a += ssum[l * qw + qq][j]; b += ssq[l * qw + qq][j]; }
            double *o = part + (((size_t)n * slices + slice) * C + ((qbase + qq) << 2) + j) * 2;
            o[0] = a; o[1] = b;
        }
        __syncthreads();
    }
}

// two partial sources (the halves of a concatenation may come from convs with different tilings).
// grid (image, group quarter): a block reduces groups/GQ groups with 256 / (groups/GQ) slice lanes
// each, so the 128-slice tensors of the 128x128 level are read by 4x as many threads.
constexpr int GN_FIN_GQ = 4;
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double *__restrict__ part0, int C0, int slices0,
                                                          const double *__restrict__ part1, int C1, int slices1,
                                                          int HW, int groups, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ float s_mean[256], s_rstd[256];
    __shared__ double ra[256], rb[256];
    const int n = blockIdx.x, t = threadIdx.x;
    const int C = C0 + C1, Cg = C / groups;
    const int gq = (groups + gridDim.y - 1) / gridDim.y;          // groups of this block
    const int g0 = blockIdx.y * gq, g1 = min(groups, g0 + gq);
    // (group, slice lane): 256 / gpp slice lanes per group, gpp groups per pass
    const int gpp = min(gq, 256);
    const int lanes = 256 / gpp;
    for (int gbase = g0; gbase < g1; gbase += gpp) {
        const int g = gbase + (t % gpp), sl0 = t / gpp;
        double a = 0, b = 0;
        if (g < g1 && sl0 < lanes) {
            for (int cc = 0; cc < Cg; ++cc) {
                const int c = g * Cg + cc;
                const bool first = c < C0;
                const double *pp = first ? part0 : part1;
                const int Cs = first ? C0 : C1, cl = first ? c : c - C0, sl = first ? slices0 : slices1;
                // four loads in flight per thread (a plain loop waits for each one); summed in slice order
                int s = sl0;
                for (; s + 3 * lanes < sl; s += 4 * lanes) {
                    double2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = *reinterpret_cast<const double2 *>(pp + (((size_t)n * sl + s + u * lanes) * Cs + cl) * 2);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a += v[u].x; b += v[u].y; }
                }
                for (; s < sl; s += lanes) {
                    const double *o = pp + (((size_t)n * sl + s) * Cs + cl) * 2;
                    a += o[0]; b += o[1];
                }
            }
        }
        ra[t] = a; rb[t] = b;
        __syncthreads();
        if (t < gpp && gbase + t < g1) {
            double sa = 0, sb = 0;
            for (int l = 0; l < lanes; ++l) { sa += ra[l * gpp + t]; sb += rb[l * gpp + t]; }
            const double cnt = (double)Cg * HW;
            const double mean = sa / cnt;
            const double var = fmax(sb / cnt - mean * mean, 0.0);
            s_mean[t] = (float)mean;
            s_rstd[t] = 1.0f / sqrtf((float)var + eps);
        }
        __syncthreads();
        const int c_lo = gbase * Cg, c_hi = min(g1, gbase + gpp) * Cg;
        for (int c = c_lo + t; c < c_hi; c += blockDim.x) {
            const int gl = c / Cg - gbase;
            const float sc = s_rstd[gl] * gamma[c];
            scale[(size_t)n * C + c] = sc;
            shift[(size_t)n * C + c] = beta[c] - s_mean[gl] * sc;
        }
        __syncthreads();
    }
}

// The same finalize for FEW images with MANY slices (small batches: a 128x128 image on 64x64 tiles leaves 256 slices,
// and the folded apply's prologue — or the kernel above on B x 4 blocks — walks them in 12-16 dependent round trips).
// One block per (image, group): its Cg x slices partial sums are all in flight at once (thread = (channel of the group,
// slice lane)), added per channel in slice order, then per group in channel order.
__global__ __launch_bounds__(256) void gn_finalize_group_kernel(const double *__restrict__ part0, int C0, int slices0,
                                                                const double *__restrict__ part1, int C1, int slices1,
                                                                int HW, int groups, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float eps,
                                                                float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ double2 red[256];
    __shared__ double2 chs[64];
    const int n = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
    const int C = C0 + C1, Cg = C / groups;         // Cg <= 64 (C <= 2048)
    const int lanes = 256 / Cg;                     // slice lanes per channel
    const int cc = t % Cg, l = t / Cg;
    double a = 0, b = 0;
    if (l < lanes) {
        const int c = g * Cg + cc;
        const bool first = c < C0;
        const double *pp = first ? part0 : part1;
        const int Cs = first ? C0 : C1, cl = first ? c : c - C0, sl = first ? slices0 : slices1;
        const double *base = pp + ((size_t)n * sl * Cs + cl) * 2;
        const size_t stride = (size_t)Cs * 2;
        int s = l;
        for (; s + 7 * lanes < sl; s += 8 * lanes) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2 *>(base + (size_t)(s + u * lanes) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
        }
        for (; s < sl; s += lanes) { const double2 v = *reinterpret_cast<const double2 *>(base + (size_t)s * stride); a += v.x; b += v.y; }
    }
    red[t] = make_double2(a, b);
    __syncthreads();
    if (t < Cg) {
        double sa = 0, sb = 0;
        for (int l2 = 0; l2 < lanes; ++l2) { const double2 v = red[l2 * Cg + t]; sa += v.x; sb += v.y; }
        chs[t] = make_double2(sa, sb);
    }
    __syncthreads();
    if (t < Cg) {
        double sa = 0, sb = 0;
        for (int k = 0; k < Cg; ++k) { sa += chs[k].x; sb += chs[k].y; }
        const double cnt = (double)Cg * HW;
        const double mean = sa / cnt;
        const double var = fmax(sb / cnt - mean * mean, 0.0);
        const float rstd = 1.0f / sqrtf((float)var + eps);
        const int c = g * Cg + t;
        const float sc = rstd * gamma[c];
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = beta[c] - (float)mean * sc;
    }
}

} // namespace

static int gn_slices(int B, int HW) {
    int s = 2048 / (B > 0 ? B : 1);
    if (s < 1) s = 1;
    int cap = HW / 64;
    if (cap < 1) cap = 1;
    if (s > cap) s = cap;
    if (s > GN_MAX_SLICES) s = GN_MAX_SLICES;
    return s;
}

// workspace in floats: B * slices * C channels * 2 doubles
size_t gn_workspace_floats(int B, int c_max) { return (size_t)B * GN_MAX_SLICES * c_max * 4; }

void launch_groupnorm_affine(const TDesc &in0, const TDesc &in1, int B, int groups, const float *gamma,
                             const float *beta, float eps, float *part, float *scale, float *shift,
                             hipStream_t s) {
    const int HW = in0.H * in0.W;
    const int slices = gn_slices(B, HW);
    const int C = in0.C + (in1.p ? in1.C : 0);
    double *dpart = reinterpret_cast<double *>(part);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(slices, B), dim3(256), 0, s, in0, in1, slices, dpart);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B, groups % GN_FIN_GQ ? 1 : GN_FIN_GQ), dim3(256), 0, s, dpart, C, slices,
                       (const double *)nullptr, 0, 0, HW, groups, gamma, beta, eps, scale, shift);
}

void launch_groupnorm_finalize(const StatsRef &s0, int C0, const StatsRef &s1, int C1, int B, int HW, int groups,
                               const float *gamma, const float *beta, float eps, float *scale, float *shift,
                               hipStream_t s) {
    // few images, many slices: one block per (image, group) instead of B x 4 blocks
    if ((long)B * GN_FIN_GQ < 128 && std::max(s0.slices, s1.slices) >= 16 && (C0 + C1) / groups <= 64 && ((C0 + C1) % groups) == 0) {
        hipLaunchKernelGGL(gn_finalize_group_kernel, dim3(B, groups), dim3(256), 0, s, s0.p, C0, s0.slices, s1.p, C1, s1.slices,
                           HW, groups, gamma, beta, eps, scale, shift);
        return;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B, groups % GN_FIN_GQ ? 1 : GN_FIN_GQ), dim3(256), 0, s, s0.p, C0, s0.slices,
                       s1.p, C1, s1.slices, HW, groups, gamma, beta, eps, scale, shift);
}

// -------------------------------------------------------------------------------------------------
// GroupNorm apply (+ Swish) (+ channel concat) into a zero-bordered tensor: the activated input of
// a conv (reference Block: GroupNorm -> Swish -> Conv, unet.py:84-87). HBM-bound element-wise pass;
// the VALU work lives here because it would serialise with the f32 MFMAs inside the conv kernel.
// -------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float swish_fast(float x) {
    // x * sigmoid(x); v_exp_f32 / v_rcp_f32 are 1 ulp on gfx950
    return x * __frcp_rn(1.0f + __expf(-x));
}

// e4m3 (OCP) pair conversion: v_cvt_pk_fp8_f32 into the low / high half of a word. The instruction does not saturate
// (|v| > 448 becomes NaN); no clamp here: the callers raise the range flag for exactly those values (|x| > SPLIT_F8_MAX,
// which also bounds x_lo * 2^11 by 256) and the flagged call's results are discarded and recomputed (or the call fails).
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// SPLIT 0: fp32; 1: split-f16 (hi | lo halfs); 2: the F8C variant (hi halfs | e4m3 of lo | e4m3 of hi; ConvParams::f8)
template <int SPLIT>
__device__ __forceinline__ void store8(float *dst, int c, const float (&f)[8], float &absmax) {
    if (SPLIT) {
        // chunk of 32 channels = 128 B: halfs [0,32) hi, [32,64) lo; x = hi + lo to ~2^-22 |x|.
        // Values beyond the fp16 range are not clamped but detected (absmax -> the ovf flag).
        h16x8 hi, lo;
        float lof[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = f[j];
            absmax = fmaxf(absmax, fabsf(g));
            hi[j] = (_Float16)g;
            lof[j] = g - (float)hi[j];
            lo[j] = (_Float16)lof[j];
        }
        _Float16 *hd = reinterpret_cast<_Float16 *>(dst + (c & ~31)) + (c & 31);
        *reinterpret_cast<h16x8 *>(hd) = hi;
        if (SPLIT == 2) {
            constexpr float SL = (float)(1 << SR3_F8_XL), SH = (float)(1 << SR3_F8_XH);
            unsigned char *b8 = reinterpret_cast<unsigned char *>(dst + (c & ~31)) + 64 + (c & 31);
            uint2 l8, h8;
            l8.x = pack4_e4m3(lof[0] * SL, lof[1] * SL, lof[2] * SL, lof[3] * SL);
            l8.y = pack4_e4m3(lof[4] * SL, lof[5] * SL, lof[6] * SL, lof[7] * SL);
            // (the fp32 value itself instead of its fp16 rounding: the same e4m3 number except at rounding ties)
            h8.x = pack4_e4m3(f[0] * SH, f[1] * SH, f[2] * SH, f[3] * SH);
            h8.y = pack4_e4m3(f[4] * SH, f[5] * SH, f[6] * SH, f[7] * SH);
            *reinterpret_cast<uint2 *>(b8) = l8;
            *reinterpret_cast<uint2 *>(b8 + 32) = h8;
        } else {
            *reinterpret_cast<h16x8 *>(hd + 32) = lo;
        }
    } else {
        *reinterpret_cast<float4 *>(dst + c) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(dst + c + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// SPLIT 3: split-f16 halfs in the fragment-major layout (sr3_internal.h fm_*): hi octet at the lane slot of its pixel,
// lo octet 1 KB behind it
__device__ __forceinline__ void store8_fm(const TDesc &out, int n, int y, int x, int c, const float (&f)[8], float &absmax) {
    h16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float g = f[j];
        absmax = fmaxf(absmax, fabsf(g));
        hi[j] = (_Float16)g;
        lo[j] = (_Float16)(g - (float)hi[j]);
    }
    const int yp = y + 1, xp = x + 1;
    const size_t blk = (((size_t)n * (out.H + 2) + yp) * fm_groups(out.W) + (xp >> 4)) * (out.C >> 5) + (c >> 5);
    char *d = reinterpret_cast<char *>(out.p) + blk * 2048 + ((c & 31) >> 3) * 256 + (xp & 15) * 16;
    *reinterpret_cast<h16x8 *>(d) = hi;
    *reinterpret_cast<h16x8 *>(d + 1024) = lo;
}

template <int SPLIT>
__device__ __forceinline__ void store_out(const TDesc &out, int n, int y, int x, int C, int c, const float (&f)[8], float &absmax) {
    if constexpr (SPLIT == 3) store8_fm(out, n, y, x, c, f, absmax);
    else store8<SPLIT>(out.p + out.pix(n, y, x) * C, c, f, absmax);
}

// Where the per-(image, channel) GroupNorm statistics of an apply pass come from: fp64 {sum, sum of
// squares} partials per (image, slice, channel) of one or two source tensors (the halves of a
// concatenation; written by conv epilogues or by gn_partial_kernel), folded with gamma / beta.
struct GnFold {
    const double *p0 = nullptr, *p1 = nullptr;
    int C0 = 0, slices0 = 0, C1 = 0, slices1 = 0;
    const float *gamma = nullptr, *beta = nullptr;
    float eps = 0.f;
    int groups = 0, HW = 0;
};

constexpr int GA_T = 512;      // threads per block of the apply pass

// One block = `ppb` consecutive pixels of one image, all channels; a thread owns 8 consecutive
// channels of a pixel per iteration (two 16-B loads, 16-B stores). The prologue folds the
// GroupNorm FINALIZE into the pass (it used to be a launch of its own in front of every apply):
// per channel it adds the slices, assembles groups from channels (any group size, groups may
// straddle the x / skip boundary of a concatenation), and leaves scale = rstd * gamma,
// shift = beta - mean * scale in LDS. Deterministic (fixed summation order, no atomics).
// MODE 0 copy, 1 affine, 2 affine + Swish; scale/shift != null: take them from memory instead.
template <int MODE, int SPLIT>
__global__ __launch_bounds__(GA_T) void gn_apply_kernel(const TDesc in0, const TDesc in1,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, const GnFold st,
                                                         const TDesc out, const TDesc raw, const int in_split,
                                                         int *ovf, const int ppb) {
    extern __shared__ __attribute__((aligned(16))) float ga_smem[];
    const int C0 = in0.C, C = out.C, C8 = C >> 3;
    const int n = blockIdx.y, t = threadIdx.x;
    float *sc = ga_smem, *sh = ga_smem + C;             // [C] each
    if (MODE != 0) {
        if (scale != nullptr) {
            for (int c = t; c < C; c += GA_T) { sc[c] = scale[(size_t)n * C + c]; sh[c] = shift[(size_t)n * C + c]; }
        } else {
            double2 *chs = reinterpret_cast<double2 *>(ga_smem + 2 * C);    // [C] per-channel totals
            double2 *red = chs + C;                                         // [GA_T]
            float *gm = reinterpret_cast<float *>(red + GA_T), *gr = gm + st.groups;
            const int CP = min(C, GA_T), L = GA_T / CP;                     // slice lanes per channel
            for (int cbase = 0; cbase < C; cbase += CP) {
                const int c = cbase + t % CP, l = t / CP;
                double a = 0, b = 0;
                if (l < L && c < C) {
                    const bool first = c < st.C0;
                    const double *pp = first ? st.p0 : st.p1;
                    const int Cs = first ? st.C0 : st.C1, cl = first ? c : c - st.C0, sl = first ? st.slices0 : st.slices1;
                    const double *base = pp + ((size_t)n * sl * Cs + cl) * 2;
                    const size_t stride = (size_t)Cs * 2;                   // doubles between slices
                    int s = l;
                    // eight loads in flight per thread (a plain loop waits for every load before the next);
                    // summed in slice order: deterministic
                    for (; s + 7 * L < sl; s += 8 * L) {
                        double2 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2 *>(base + (size_t)(s + u * L) * stride);
#pragma unroll
                        for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
                    }
                    for (; s < sl; s += L) {
                        const double2 v = *reinterpret_cast<const double2 *>(base + (size_t)s * stride);
                        a += v.x; b += v.y;
                    }
                }
                red[t] = make_double2(a, b);
                __syncthreads();
                if (t < CP && cbase + t < C) {
                    double sa = 0, sb = 0;
                    for (int l2 = 0; l2 < L; ++l2) { const double2 v = red[l2 * CP + t]; sa += v.x; sb += v.y; }
                    chs[cbase + t] = make_double2(sa, sb);
                }
                __syncthreads();
            }
            const int Cg = C / st.groups;
            for (int g = t; g < st.groups; g += GA_T) {
                double sa = 0, sb = 0;
                for (int cc = 0; cc < Cg; ++cc) { const double2 v = chs[g * Cg + cc]; sa += v.x; sb += v.y; }
                const double cnt = (double)Cg * st.HW;
                const double mean = sa / cnt;
                const double var = fmax(sb / cnt - mean * mean, 0.0);
                gm[g] = (float)mean;
                gr[g] = 1.0f / sqrtf((float)var + st.eps);
            }
            __syncthreads();
            for (int c = t; c < C; c += GA_T) {
                const int g = c / Cg;
                const float v = gr[g] * st.gamma[c];
                sc[c] = v;
                sh[c] = st.beta[c] - gm[g] * v;
            }
        }
        __syncthreads();
    }
    // thread = (pixel lane, channel octet): the octet and its scale / shift stay in registers, the pixel
    // advances by `rows` per iteration (no division in the loop)
    const int W = out.W, HW = out.H * W;
    const int TC = min(C8, GA_T), rows = GA_T / TC;
    const int pl = t / TC;
    if (pl >= rows) return;
    const int pix0 = blockIdx.x * ppb, pix1 = min(HW, pix0 + ppb);
    float absmax = 0.f, absmax_raw = 0.f;
    for (int c = (t - pl * TC) << 3; c < C; c += TC << 3) {       // one pass unless C8 > GA_T
        float scv[8], shv[8];
        if (MODE != 0) {
            const float4 s0 = *reinterpret_cast<const float4 *>(sc + c), s1 = *reinterpret_cast<const float4 *>(sc + c + 4);
            const float4 h0 = *reinterpret_cast<const float4 *>(sh + c), h1 = *reinterpret_cast<const float4 *>(sh + c + 4);
            scv[0] = s0.x; scv[1] = s0.y; scv[2] = s0.z; scv[3] = s0.w; scv[4] = s1.x; scv[5] = s1.y; scv[6] = s1.z; scv[7] = s1.w;
            shv[0] = h0.x; shv[1] = h0.y; shv[2] = h0.z; shv[3] = h0.w; shv[4] = h1.x; shv[5] = h1.y; shv[6] = h1.z; shv[7] = h1.w;
        }
        // in_split bit 0 / 1: in0 / in1 is stored in the split-f16 format (a conv wrote only the twin of
        // its output): x = hi + lo, exact to ~2^-22 |x|
        const bool first = c < C0;
        const int cl = first ? c : c - C0;
        const bool src_split = (in_split >> (first ? 0 : 1)) & 1;
        const TDesc &src = first ? in0 : in1;
        const size_t Cs = src.C;
        // two pixels per iteration: both loads are issued before either is used (twice the bytes in
        // flight per thread; a single-item loop waits for its load before the next one is issued)
        struct Item { f32x4v a, b; };
        auto load = [&](int y, int x) -> Item {
            const float *pixp = src.p + src.pix(n, y, x) * Cs;
            Item it;
            if (src_split) {
                const _Float16 *hp = reinterpret_cast<const _Float16 *>(pixp + (cl & ~31)) + (cl & 31);
                it.a = *reinterpret_cast<const f32x4v *>(hp);            // 8 hi halfs
                it.b = *reinterpret_cast<const f32x4v *>(hp + 32);       // 8 lo halfs
            } else {
                it.a = *reinterpret_cast<const f32x4v *>(pixp + cl);
                it.b = *reinterpret_cast<const f32x4v *>(pixp + cl + 4);
            }
            return it;
        };
        auto finish = [&](const Item &it, int y, int x) {
            float f[8];
            if (src_split) {
                const h16x8 hi = __builtin_bit_cast(h16x8, it.a), lo = __builtin_bit_cast(h16x8, it.b);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (float)hi[j] + (float)lo[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = it.a[j]; f[4 + j] = it.b[j]; }
            }
            if (raw.p != nullptr) store8<(SPLIT ? 1 : 0)>(raw.p + raw.pix(n, y, x) * C, c, f, absmax_raw);
            if (MODE != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], scv[j], shv[j]);
            }
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = swish_fast(f[j]);
            }
            store_out<SPLIT>(out, n, y, x, C, c, f, absmax);
        };
        int pix = pix0 + pl;
        int y = pix / W, x = pix - y * W;
        for (; pix + rows < pix1; pix += 2 * rows) {
            int x2 = x + rows, y2 = y;
            while (x2 >= W) { x2 -= W; ++y2; }
            const Item i0 = load(y, x), i1 = load(y2, x2);
            finish(i0, y, x);
            finish(i1, y2, x2);
            x = x2 + rows; y = y2;
            while (x >= W) { x -= W; ++y; }
        }
        for (; pix < pix1; pix += rows) {       // tail
            finish(load(y, x), y, x);
            x += rows;
            while (x >= W) { x -= W; ++y; }
        }
    }
    if (SPLIT && ovf != nullptr && (absmax > (SPLIT == 2 ? SPLIT_F8_MAX : SPLIT_F16_MAX) || absmax_raw > SPLIT_F16_MAX)) *ovf = 1;
}

} // namespace

namespace {
// Streaming form for LARGE tensors: one (pixel, channel octet) per thread, no loop, scale / shift read from
// memory (written by gn_finalize_kernel). On tensors of hundreds of MB it streams ~10 % faster than the
// folded form above (whose per-block prologue re-reads the partial statistics), which more than pays
// for the separate finalize launch; the folded form wins wherever a launch is latency bound.
// NR = 2: two rows per thread (y and y + H/2), both loads in flight before either is used: +1.4 % on the pass
// (A/B on one box; nontemporal loads of the input measured 8 % slower)
template <int MODE, int SPLIT, int NR>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const TDesc in0, const TDesc in1,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const TDesc out,
                                                            const TDesc raw, const int in_split, int *ovf) {
    // grid: x = chunks of (pixel-in-row, channel octet), y = n * (H / NR) + row
    const int C0 = in0.C, C = out.C, C8 = C >> 3;
    const int Hh = out.H / NR;
    const int n = blockIdx.y / Hh, y0 = blockIdx.y - n * Hh;
    const int item = blockIdx.x * 256 + threadIdx.x;
    // (SPLIT 3: the launcher guarantees C == 64 and W % 32 == 0 — whole blocks, every thread reaches the barrier)
    __shared__ h16x8 fm_stage[SPLIT == 3 ? NR * 8 * 2 * 32 : 1];
    if (SPLIT != 3 && item >= out.W * C8) return;
    const int x = item / C8;
    const int c = (item - x * C8) << 3;
    const bool first = c < C0;
    const int cl = first ? c : c - C0;
    const bool src_split = (in_split >> (first ? 0 : 1)) & 1;
    f32x4v la[NR], lb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int y = y0 + r * Hh;
        const float *pixp = first ? in0.p + in0.pix(n, y, x) * C0 : in1.p + in1.pix(n, y, x) * in1.C;
        const f32x4v *pa, *pb;
        if (src_split) {
            const _Float16 *hp = reinterpret_cast<const _Float16 *>(pixp + (cl & ~31)) + (cl & 31);
            pa = reinterpret_cast<const f32x4v *>(hp); pb = reinterpret_cast<const f32x4v *>(hp + 32);
        } else {
            pa = reinterpret_cast<const f32x4v *>(pixp + cl); pb = reinterpret_cast<const f32x4v *>(pixp + cl + 4);
        }
        la[r] = *pa;
        lb[r] = *pb;
    }
    float scv[8], shv[8];
    if (MODE != 0) {
        const float *scp = scale + (size_t)n * C + c, *shp = shift + (size_t)n * C + c;
        const float4 s0 = *reinterpret_cast<const float4 *>(scp), s1 = *reinterpret_cast<const float4 *>(scp + 4);
        const float4 h0 = *reinterpret_cast<const float4 *>(shp), h1 = *reinterpret_cast<const float4 *>(shp + 4);
        scv[0] = s0.x; scv[1] = s0.y; scv[2] = s0.z; scv[3] = s0.w; scv[4] = s1.x; scv[5] = s1.y; scv[6] = s1.z; scv[7] = s1.w;
        shv[0] = h0.x; shv[1] = h0.y; shv[2] = h0.z; shv[3] = h0.w; shv[4] = h1.x; shv[5] = h1.y; shv[6] = h1.z; shv[7] = h1.w;
    }
    float absmax = 0.f, absmax_raw = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int y = y0 + r * Hh;
        float f[8];
        if (src_split) {
            const h16x8 hi = __builtin_bit_cast(h16x8, la[r]), lo = __builtin_bit_cast(h16x8, lb[r]);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)hi[j] + (float)lo[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = la[r][j]; f[4 + j] = lb[r][j]; }
        }
        if (raw.p != nullptr) store8<(SPLIT ? 1 : 0)>(raw.p + raw.pix(n, y, x) * C, c, f, absmax_raw);
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], scv[j], shv[j]);
        }
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = swish_fast(f[j]);
        }
        if constexpr (SPLIT == 3) {
            // Fragment-major output (C == 64, W % 32 == 0: a block is 32 consecutive pixels x 8 octets). Written straight
            // from this thread mapping every quad would store four 16-byte pieces 256 B apart (+50 % on the whole pass:
            // profiles/README.md finding 66), so the pieces go through LDS: thread (pixel pl, octet j) leaves hi | lo in
            // slot [row][j][h][pl]; after the barrier thread t picks up (j = t >> 5, pl = t & 31) — a quad then writes
            // 64 consecutive bytes of one (chunk, h, q) run of the FM block.
            h16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = f[j];
                absmax = fmaxf(absmax, fabsf(g));
                hi[j] = (_Float16)g;
                lo[j] = (_Float16)(g - (float)hi[j]);
            }
            const int pl = threadIdx.x >> 3, jo = threadIdx.x & 7;
            fm_stage[((r * 8 + jo) * 2 + 0) * 32 + pl] = hi;
            fm_stage[((r * 8 + jo) * 2 + 1) * 32 + pl] = lo;
        } else {
            store_out<SPLIT>(out, n, y, x, C, c, f, absmax);
        }
    }
    if constexpr (SPLIT == 3) {
        __syncthreads();
        const int jo = threadIdx.x >> 5, pl = threadIdx.x & 31;
        const int xq = blockIdx.x * 32 + pl + 1;                    // padded x of this thread's output pixel
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int yp = y0 + r * Hh + 1;
            const size_t blk = (((size_t)n * (out.H + 2) + yp) * fm_groups(out.W) + (xq >> 4)) * 2 + (jo >> 2);
            char *d = reinterpret_cast<char *>(out.p) + blk * 2048 + (jo & 3) * 256 + (xq & 15) * 16;
            *reinterpret_cast<h16x8 *>(d) = fm_stage[((r * 8 + jo) * 2 + 0) * 32 + pl];
            *reinterpret_cast<h16x8 *>(d + 1024) = fm_stage[((r * 8 + jo) * 2 + 1) * 32 + pl];
        }
    }
    if (SPLIT && ovf != nullptr && (absmax > (SPLIT == 2 ? SPLIT_F8_MAX : SPLIT_F16_MAX) || absmax_raw > SPLIT_F16_MAX)) *ovf = 1;
}
} // namespace

void launch_gn_apply_rows(const TDesc &in0, const TDesc &in1, int B, const float *scale, const float *shift,
                          int mode, int split, const TDesc &out, hipStream_t s, const TDesc &raw, int in_split, int *ovf) {
    const int items = out.W * (out.C >> 3);
    const int nr = (out.H % 2) == 0 ? 2 : 1;
    const dim3 grid((items + 255) / 256, B * out.H / nr);
#define SR3_GR(M, S)                                                                                               \
    {                                                                                                              \
        if (nr == 2) hipLaunchKernelGGL((gn_apply_rows_kernel<M, S, 2>), grid, dim3(256), 0, s, in0, in1, scale, shift, out, raw, in_split, ovf); \
        else hipLaunchKernelGGL((gn_apply_rows_kernel<M, S, 1>), grid, dim3(256), 0, s, in0, in1, scale, shift, out, raw, in_split, ovf);         \
    }
#ifdef SR3_EXPERIMENTS      // fragment-major output: the weights-stationary conv experiment only (profiles/README.md finding 66)
    if (split == 3) {
        if (mode == 0) SR3_GR(0, 3) else if (mode == 1) SR3_GR(1, 3) else SR3_GR(2, 3)
    } else
#endif
    if (split == 2) {
        if (mode == 0) SR3_GR(0, 2) else if (mode == 1) SR3_GR(1, 2) else SR3_GR(2, 2)
    } else if (split) {
        if (mode == 0) SR3_GR(0, 1) else if (mode == 1) SR3_GR(1, 1) else SR3_GR(2, 1)
    } else {
        if (mode == 0) SR3_GR(0, 0) else if (mode == 1) SR3_GR(1, 0) else SR3_GR(2, 0)
    }
#undef SR3_GR
}

// pixels per block: enough blocks to keep every CU streaming (>= ~4 blocks of 512 threads per CU over
// the whole grid), at least two items per thread
static int ga_pixels_per_block(int B, int HW, int C8) {
    const int total = exp_int("SR3_GN_BLOCKS", 1024);       // (experiments build reads the variable)
    int P = (total + B - 1) / B;                        // blocks per image wanted
    const int maxP = (HW * C8 + 2 * GA_T - 1) / (2 * GA_T);
    if (P > maxP) P = maxP;
    if (P < 1) P = 1;
    return (HW + P - 1) / P;
}

static void launch_gn_apply_impl(const TDesc &in0, const TDesc &in1, int B, const float *scale, const float *shift,
                                 const GnFold &st, int mode, int split, const TDesc &out, hipStream_t s, const TDesc &raw,
                                 int in_split, int *ovf) {
    const int HW = out.H * out.W, C = out.C;
    const int ppb = ga_pixels_per_block(B, HW, C >> 3);
    const dim3 grid((HW + ppb - 1) / ppb, B);
    // LDS: scale / shift [C] floats; fold scratch: per-channel totals [C] + lanes [GA_T] double2, group mean / rstd
    const size_t lds = mode == 0 ? 16 : ((size_t)2 * C * sizeof(float) +
                       (scale ? 0 : ((size_t)C + GA_T) * sizeof(double2) + (size_t)2 * st.groups * sizeof(float)));
#define SR3_GA(M, S)                                                                                               \
    {                                                                                                              \
        static size_t attr = 48 * 1024;                                                                            \
        if (lds > attr) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gn_apply_kernel<M, S>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
            attr = lds;                                                                                            \
        }                                                                                                          \
        hipLaunchKernelGGL((gn_apply_kernel<M, S>), grid, dim3(GA_T), lds, s, in0, in1, scale, shift, st, out, raw, \
                           in_split, ovf, ppb);                                                                    \
    }
#ifdef SR3_EXPERIMENTS
    if (split == 3) {
        if (mode == 0) SR3_GA(0, 3) else if (mode == 1) SR3_GA(1, 3) else SR3_GA(2, 3)
    } else
#endif
    if (split == 2) {
        if (mode == 0) SR3_GA(0, 2) else if (mode == 1) SR3_GA(1, 2) else SR3_GA(2, 2)
    } else if (split) {
        if (mode == 0) SR3_GA(0, 1) else if (mode == 1) SR3_GA(1, 1) else SR3_GA(2, 1)
    } else {
        if (mode == 0) SR3_GA(0, 0) else if (mode == 1) SR3_GA(1, 0) else SR3_GA(2, 0)
    }
#undef SR3_GA
}

void launch_gn_apply(const TDesc &in0, const TDesc &in1, int B, const float *scale, const float *shift,
                     int mode, int split, const TDesc &out, hipStream_t s, const TDesc &raw, int in_split, int *ovf) {
    launch_gn_apply_impl(in0, in1, B, scale, shift, GnFold(), mode, split, out, s, raw, in_split, ovf);
}

// GroupNorm (statistics already accumulated as partials) + affine (+ Swish) (+ concat) in ONE launch
void launch_gn_fold_apply(const TDesc &in0, const TDesc &in1, int B, const StatsRef &s0, const StatsRef &s1, int groups,
                          const float *gamma, const float *beta, float eps, int mode, int split, const TDesc &out,
                          hipStream_t s, const TDesc &raw, int in_split, int *ovf) {
    GnFold st;
    st.p0 = s0.p; st.C0 = in0.C; st.slices0 = s0.slices;
    st.p1 = s1.p; st.C1 = in1.p ? in1.C : 0; st.slices1 = s1.slices;
    if (in1.p && !s1.p) { st.C0 = in0.C + in1.C; st.C1 = 0; }      // s0 describes the whole concatenation
    st.gamma = gamma; st.beta = beta; st.eps = eps; st.groups = groups; st.HW = in0.H * in0.W;
    launch_gn_apply_impl(in0, in1, B, nullptr, nullptr, st, mode, split, out, s, raw, in_split, ovf);
}

// statistics by the streaming kernel (tensors whose producer could not fuse them): partials of the
// virtual concatenation as ONE source with C0 + C1 channels
StatsRef launch_groupnorm_partials(const TDesc &in0, const TDesc &in1, int B, float *part, hipStream_t s) {
    const int HW = in0.H * in0.W;
    const int slices = gn_slices(B, HW);
    double *dpart = reinterpret_cast<double *>(part);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(slices, B), dim3(256), 0, s, in0, in1, slices, dpart);
    StatsRef r; r.p = dpart; r.slices = slices;
    return r;
}

// =================================================================================================
// Self-attention core (reference unet.py:132-139): one head, scores q.k/sqrt(C), softmax over
// keys, out = P v. qkv is [B][N][3C] (the 1x1 qkv conv output, q|k|v on the channel axis).
// One block = 32 query rows of one image; QK^T and PV on v_mfma_f32_32x32x2_f32, the 32 x N score
// tile lives in LDS, softmax with 8 lanes per row.
// =================================================================================================
namespace {

__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ qkv, int N, int C,
                                                        float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float S[];  // [32][Np + 4]
    const int Np = (N + 31) & ~31;
    const int ld = Np + 4;
    const int q0 = blockIdx.x * 32, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const size_t rs = (size_t)3 * C;  // row stride of qkv
    const float *base = qkv + (size_t)b * N * rs;
    const float sdiv = sqrtf((float)C);

    // ---- scores ----
    const int qrow = q0 + li;
    const float *qp = base + (size_t)(qrow < N ? qrow : 0) * rs + 4 * lh;
    for (int kb = wid; kb < Np / 32; kb += 4) {
        const int krow = kb * 32 + li;
        const float *kp = base + (size_t)(krow < N ? krow : 0) * rs + C + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < C; k0 += 8) {
            float4 a = *reinterpret_cast<const float4 *>(qp + k0);
            float4 k = *reinterpret_cast<const float4 *>(kp + k0);
            if (qrow >= N) a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (krow >= N) k = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, k.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, k.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, k.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, k.w, acc, 0, 0, 0);
        }
        const int col = kb * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            S[row * ld + col] = (col < N) ? acc[r] / sdiv : -INFINITY;
        }
    }
    __syncthreads();

    // ---- softmax over keys, 8 lanes per query row ----
    {
        const int row = tid >> 3, sub = tid & 7;
        float *sr = S + row * ld;
        float mx = -INFINITY;
        for (int c = sub; c < Np; c += 8) mx = fmaxf(mx, sr[c]);
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        float sum = 0.f;
        for (int c = sub; c < Np; c += 8) {
            const float e = expf(sr[c] - mx);
            sr[c] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        for (int c = sub; c < Np; c += 8) sr[c] = sr[c] / sum;
    }
    __syncthreads();

    // ---- out = P v ----
    const float *vp = base + 2 * C;
    for (int cb = wid; cb < C / 32; cb += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < Np; k0 += 8) {
            const float4 pa = *reinterpret_cast<const float4 *>(S + li * ld + k0 + 4 * lh);
            const int t0 = k0 + 4 * lh;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tk = t0 + j;
                v[j] = (tk < N) ? vp[(size_t)tk * rs + cb * 32 + li] : 0.f;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, v[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, v[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, v[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, v[3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < N) out[((size_t)b * N + row) * C + cb * 32 + li] = acc[r];
        }
    }
}

} // namespace

// -------------------------------------------------------------------------------------------------
// Split-f16 form of the same core (precision mode f16x3): q, k, v arrive in the conv's split operand
// format (the qkv projection writes only the twin of its output: per token 3C/32 chunks of 32 hi
// halfs | 32 lo halfs), every product is lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16 with fp32
// accumulation, the probabilities are split in LDS after the fp32 softmax. One block = 32 queries of
// one image; a wave takes 16-key tiles (scores) and 16-channel tiles (P v) round-robin.
//   A fragments (q rows, P rows): lane (l16, q4) holds row l16, k-chunk q4 = 8 consecutive k;
//   B fragments of k rows: the same pattern; of v: 8 consecutive KEYS of channel l16 — gathered as
//   2-byte loads from the 8 key rows (v is stored token-major).
// Output: fp32 [B][N][C] and / or the split twin of it (input format of the out projection).
// -------------------------------------------------------------------------------------------------
namespace {

typedef float f32x4a __attribute__((ext_vector_type(4)));

// compile-time loop (every index a constant in the front end)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Operand ring of two register sets over n steps: the fragments of step i + 1 are in flight while step i multiplies
// (steps past the end re-fetch the last one: unused). fetch(set, step), mult(set, step). Four sets were tried in the
// 64-query form (240 registers): 0.487 against 0.473 ms per step (profiles/README.md finding 43).
template <typename FE, typename MU>
__device__ __forceinline__ void attn_ring2(int n, FE &&fetch, MU &&mult) {
    fetch(std::integral_constant<int, 0>{}, 0);
    for (int c = 0; c < n; c += 2) {
        fetch(std::integral_constant<int, 1>{}, min(c + 1, n - 1));
        mult(std::integral_constant<int, 0>{}, c);
        if (c + 1 < n) {
            fetch(std::integral_constant<int, 0>{}, min(c + 2, n - 1));
            mult(std::integral_constant<int, 1>{}, c + 1);
        }
    }
}

// The same ring with a compile-time step count, fully unrolled, NS register sets deep (NS - 1 steps in flight): without
// a loop back-edge the compiler counts the outstanding loads exactly (`s_waitcnt vmcnt(N)` with N = the loads of the
// younger steps) — in the rolled loop it drained `vmcnt(0)` at the top of every iteration, so one step of MFMAs was all
// that ever covered a load's latency (the block is alone on its CU: nobody else hides it).
template <int N, int NS, typename FE, typename MU>
__device__ __forceinline__ void attn_ring_static(FE &&fetch, MU &&mult) {
    static_for<(NS - 1 < N ? NS - 1 : N)>([&](auto ic) { fetch(std::integral_constant<int, decltype(ic)::value % NS>{}, decltype(ic)::value); });
    __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks the loads next to their uses: nothing in flight)
    static_for<N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i + NS - 1 < N) fetch(std::integral_constant<int, (i + NS - 1) % NS>{}, i + NS - 1);
        __builtin_amdgcn_sched_barrier(0);
        mult(std::integral_constant<int, i % NS>{}, i);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// v^T in the split format: vt[b][c][Np / 32 chunks][32 hi halfs of keys | 32 lo halfs] from the token-major
// v part of qkv (one LDS transpose per 32 keys x 128 channels; done once per image, not once per block of
// queries). Keys >= N are clamped (their probabilities are 0).
__global__ __launch_bounds__(256) void attention_vt_kernel(const float *__restrict__ qkv, int N, int C,
                                                           float *__restrict__ vt) {
    constexpr int VROW = 520;                                   // bytes per staged key row (512 + 8: bank spread)
    __shared__ __attribute__((aligned(16))) char T[32 * VROW];
    const int Np = (N + 31) & ~31;
    const int ks = blockIdx.x, cq = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int nch = C >> 5;
    const size_t rsb = (size_t)3 * C * 4;
    const char *base = reinterpret_cast<const char *>(qkv) + (size_t)b * N * rsb;
    {
        const int key = min(ks * 32 + (tid >> 3), N - 1);
        const char *src = base + (size_t)key * rsb + (size_t)(2 * nch + cq * 4) * 128;
        char *dst = T + (tid >> 3) * VROW + (tid & 7) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int piece = (tid & 7) * 4 + u;                // 16-B piece of the 512-B row; chunk = piece / 8
            double2 v = make_double2(0.0, 0.0);
            if (cq * 4 + (piece >> 3) < nch) v = *reinterpret_cast<const double2 *>(src + piece * 16);
            *reinterpret_cast<double *>(dst + u * 16) = v.x;   // 8-byte stores: the padded rows are 8-B aligned
            *reinterpret_cast<double *>(dst + u * 16 + 8) = v.y;
        }
    }
    __syncthreads();
    const int cw = tid >> 1, hl = tid & 1;                      // channel inside the quarter, hi | lo
    const int c = cq * 128 + cw;
    if (c >= C) return;
    const _Float16 *col = reinterpret_cast<const _Float16 *>(T + (cw >> 5) * 128 + (cw & 31) * 2 + hl * 64);
    h16x8 o[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = col[(g * 8 + j) * (VROW / 2)];
    char *dst = reinterpret_cast<char *>(vt) + (((size_t)b * C + c) * (Np / 32) + ks) * 128 + hl * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<h16x8 *>(dst + g * 16) = o[g];
}

// NTW / NTC: key tiles (scores) and channel tiles (P v) per wave, compile-time maxima; MTQ: 16-query tiles per block
// (32 or 64 queries); NWAVES: waves per block. 64 queries x 8 waves halves the k / v^T bytes a block pulls through
// L2 per query (the kernel is bound by that traffic) at the same number of waves per CU.
// NCH / NKS > 0: channel chunks (C / 32) and key steps (Np / 32) known at compile time — both operand loops fully unrolled
// over deeper register rings (attn_ring_static); 0: run-time counts, two-set rings.
template <int NTW, int NTC, int MTQ, int NWAVES, int NCH = 0, int NKS = 0>
__global__ __launch_bounds__(NWAVES * 64) void attention_split_kernel(const float *__restrict__ qkv, const float *__restrict__ vt,
                                                              int N, int C, float *__restrict__ out,
                                                              float *__restrict__ out_split, int *ovf, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [QB][ld]: fp32 scores, then P as [hi8|lo8] groups
    constexpr int QB = MTQ * 16;
#ifndef SR3_EXPERIMENTS
    dbg = 0;        // timing experiments only (SR3_ATTN_DBG in the experiments build): 1 no scores, 2 no softmax, 4 no P v
#endif
    const int Np = (N + 31) & ~31;
    const int ld = Np + 8;
    // XCD-aware block order (speed only): blocks b and b + 8 share an XCD, so every XCD gets a contiguous
    // range of (image, query block) pairs — the query blocks of one image then read its k and v^T through
    // ONE L2 instead of eight (measured: the kernel was bound by those re-reads)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, loc = bid >> 3, qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int nqb = Np / QB;
    const int b = bid / nqb, q0 = (bid - b * nqb) * QB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l16 = lane & 15, q4 = lane >> 4;
    const size_t rsb = (size_t)3 * C * 4;                        // bytes per token row
    const char *base = reinterpret_cast<const char *>(qkv) + (size_t)b * N * rsb;
    const float sdiv = sqrtf((float)C);
    const int nch = C >> 5;

    // ---- 64-query form: the q rows of the block are staged in LDS once (every wave needs the fragments of all
    // queries: read from global memory they were half of the bytes a block pulled through L2). Row stride
    // 4 C + 32 bytes: the 16-byte fragment reads of a lane group fall on distinct banks. The score tile S takes
    // over the same LDS region once the scores are in registers. ----
    constexpr bool QLDS = MTQ == 4;
    constexpr bool STATIC = NCH > 0 && NKS > 0;
    constexpr int NSET = STATIC ? 4 : 2;     // register sets of the k ring (scores)
    constexpr int NSETV = STATIC ? 3 : 2;    // register sets of the v^T ring (P v)
    constexpr int NSETQ = 2;                 // q fragments (LDS or global): one step ahead
    const int qstride = C * 4 + 32;
    if (QLDS) {
        char *Q = reinterpret_cast<char *>(S);
        const int pieces = C >> 2;                                  // 16-byte pieces per q row (4 C bytes)
        for (int it = tid; it < QB * pieces; it += NWAVES * 64 * 4) {
            f32x4a v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(it + u * NWAVES * 64, QB * pieces - 1);
                const int r = e / pieces, pc = e - r * pieces;
                v[u] = *reinterpret_cast<const f32x4a *>(base + (size_t)min(q0 + r, N - 1) * rsb + pc * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = it + u * NWAVES * 64;
                if (e < QB * pieces) {
                    const int r = e / pieces, pc = e - r * pieces;
                    *reinterpret_cast<f32x4a *>(Q + r * qstride + pc * 16) = v[u];
                }
            }
        }
        __syncthreads();
    }

    // ---- scores: S[query][key] = q . k / sqrt(C). Wave w owns the key tiles w * NTW .. + NTW - 1: the q
    // fragments of a channel chunk are loaded once for all of them ----
    {
        const char *qrow[MTQ];
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
            qrow[mt] = QLDS ? reinterpret_cast<const char *>(S) + (mt * 16 + l16) * qstride + q4 * 16
                            : base + (size_t)min(q0 + mt * 16 + l16, N - 1) * rsb + q4 * 16;
        const int nkt = Np / 16;                                  // key tiles
        const char *krow[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
            krow[i] = base + (size_t)min((wid * NTW + i) * 16 + l16, N - 1) * rsb + (size_t)nch * 128 + q4 * 16;
        f32x4a acc[NTW][MTQ];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][mt][r] = 0.f;
        if (wid * NTW < nkt && !(dbg & 1)) {
            // the fragments of chunk ch + 1 are in flight while chunk ch multiplies (two register sets; rows of
            // tiles past the end are clamped duplicates whose scores are never stored)
            if constexpr (STATIC) {
                // k fragments (global memory) NSET - 1 chunks ahead, q fragments (LDS) read in the step that uses them
                h16x8 bh[NSET][NTW], bl[NSET][NTW];
                auto fetch = [&](auto setc, int ch) {
                    constexpr int set = decltype(setc)::value;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        bh[set][i] = *reinterpret_cast<const h16x8 *>(krow[i] + ch * 128);
                        bl[set][i] = *reinterpret_cast<const h16x8 *>(krow[i] + ch * 128 + 64);
                    }
                };
                auto mult = [&](auto setc, int ch) {
                    constexpr int set = decltype(setc)::value;
                    h16x8 ah[MTQ], al[MTQ];
#pragma unroll
                    for (int mt = 0; mt < MTQ; ++mt) {
                        ah[mt] = *reinterpret_cast<const h16x8 *>(qrow[mt] + ch * 128);
                        al[mt] = *reinterpret_cast<const h16x8 *>(qrow[mt] + ch * 128 + 64);
                    }
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
#pragma unroll
                        for (int mt = 0; mt < MTQ; ++mt) {
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[set][i], acc[i][mt], 0, 0, 0);
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[set][i], acc[i][mt], 0, 0, 0);
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[set][i], acc[i][mt], 0, 0, 0);
                        }
                };
                attn_ring_static<NCH, NSET>(fetch, mult);
            } else {
            h16x8 ah[NSETQ][MTQ], al[NSETQ][MTQ], bh[NSETQ][NTW], bl[NSETQ][NTW];
            auto fetch = [&](auto setc, int ch) {
                constexpr int set = decltype(setc)::value;
#pragma unroll
                for (int mt = 0; mt < MTQ; ++mt) {
                    ah[set][mt] = *reinterpret_cast<const h16x8 *>(qrow[mt] + ch * 128);
                    al[set][mt] = *reinterpret_cast<const h16x8 *>(qrow[mt] + ch * 128 + 64);
                }
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    bh[set][i] = *reinterpret_cast<const h16x8 *>(krow[i] + ch * 128);
                    bl[set][i] = *reinterpret_cast<const h16x8 *>(krow[i] + ch * 128 + 64);
                }
            };
            auto mult = [&](auto setc, int) {
                constexpr int set = decltype(setc)::value;
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int mt = 0; mt < MTQ; ++mt) {
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[set][mt], bh[set][i], acc[i][mt], 0, 0, 0);
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[set][mt], bl[set][i], acc[i][mt], 0, 0, 0);
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[set][mt], bh[set][i], acc[i][mt], 0, 0, 0);
                    }
            };
            attn_ring2(nch, fetch, mult);
            }
        }
        if (QLDS) __syncthreads();          // every wave has read its last q fragment: S may overwrite the q rows
        // C/D map: col = l16 (key), row = 4 q4 + j (query)
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int kt = wid * NTW + i;
            if (kt < nkt) {
                const int col = kt * 16 + l16;
#pragma unroll
                for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        S[(mt * 16 + 4 * q4 + j) * ld + col] = (col < N) ? acc[i][mt][j] / sdiv : -INFINITY;
            }
        }
    }
    __syncthreads();

    // ---- softmax over keys (fp32, 8 lanes per query row), then P -> [8 hi halfs | 8 lo halfs] per 8 keys, in place ----
    if (!(dbg & 2)) {
        constexpr int RPP = NWAVES * 8, NRP = QB / RPP;     // query rows per pass of the block, passes
        const int sub = tid & 7;
        float sums[NRP];
#pragma unroll
        for (int rp = 0; rp < NRP; ++rp) {
            float *sr = S + ((tid >> 3) + rp * RPP) * ld;
            float mx = -INFINITY;
            for (int c = sub; c < Np; c += 8) mx = fmaxf(mx, sr[c]);
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            float sum = 0.f;
            for (int c = sub; c < Np; c += 8) {
                const float e = expf(sr[c] - mx);
                sr[c] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4);
            sums[rp] = sum;
        }
        __syncthreads();                    // every lane of a row has finished its strided pass
#pragma unroll
        for (int rp = 0; rp < NRP; ++rp) {
            float *sr = S + ((tid >> 3) + rp * RPP) * ld;
            const float sum = sums[rp];
            for (int g = sub; g < Np / 8; g += 8) {
                float *gp = sr + g * 8;
                const f32x4a v0 = *reinterpret_cast<const f32x4a *>(gp), v1 = *reinterpret_cast<const f32x4a *>(gp + 4);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                h16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = f[j] / sum;
                    hi[j] = (_Float16)pv;
                    lo[j] = (_Float16)(pv - (float)hi[j]);
                }
                *reinterpret_cast<h16x8 *>(gp) = hi;
                *reinterpret_cast<h16x8 *>(gp + 4) = lo;
            }
        }
    }
    __syncthreads();

    // ---- out = P v: A = P rows from LDS, B = rows of v^T (8 consecutive keys per lane and channel); wave w
    // owns the channel tiles w, w + NWAVES, ... ----
    const unsigned psel = split_pair_selector(l16 & 1);
    unsigned range_bits = 0;
    const int nct = C >> 4, nks = Np >> 5;
    const char *vrow[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
        const int c = min((wid + NWAVES * t) * 16 + l16, C - 1);
        vrow[t] = reinterpret_cast<const char *>(vt) + ((size_t)b * C + c) * nks * 128 + q4 * 16;
    }
    f32x4a acc[NTC][MTQ];
#pragma unroll
    for (int t = 0; t < NTC; ++t)
#pragma unroll
        for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][mt][r] = 0.f;
    {
        // v^T fragments of key step ks + 1 in flight while step ks multiplies (channel tiles past the end are
        // clamped duplicates that are never stored)
        h16x8 vh[NSETV][NTC], vl[NSETV][NTC];
        auto fetch = [&](auto setc, int ks) {
            constexpr int set = decltype(setc)::value;
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
                vh[set][t] = *reinterpret_cast<const h16x8 *>(vrow[t] + ks * 128);
                vl[set][t] = *reinterpret_cast<const h16x8 *>(vrow[t] + ks * 128 + 64);
            }
        };
        auto mult = [&](auto setc, int ks) {
            constexpr int set = decltype(setc)::value;
            h16x8 ah[MTQ], al[MTQ];
#pragma unroll
            for (int mt = 0; mt < MTQ; ++mt) {
                const float *pp = S + (mt * 16 + l16) * ld + (ks * 4 + q4) * 8;
                ah[mt] = *reinterpret_cast<const h16x8 *>(pp);
                al[mt] = *reinterpret_cast<const h16x8 *>(pp + 4);
            }
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int mt = 0; mt < MTQ; ++mt) {
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], vh[set][t], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], vl[set][t], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], vh[set][t], acc[t][mt], 0, 0, 0);
                }
        };
        if constexpr (STATIC) attn_ring_static<NKS, NSETV>(fetch, mult);
        else if (!(dbg & 4)) attn_ring2(nks, fetch, mult);
    }
    // C/D map: col = l16 (channel), row = 4 q4 + j (query)
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
        const int ch0 = (wid + NWAVES * t) * 16;
        if (wid + NWAVES * t < nct) {
#pragma unroll
            for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = q0 + mt * 16 + 4 * q4 + j;
                    const bool ok = row < N;
                    const size_t o = ((size_t)b * N + (ok ? row : 0)) * C + ch0 + l16;
                    if (out != nullptr && ok) out[o] = acc[t][mt][j];
                    if (out_split != nullptr) {
                        // every lane takes part in the DPP exchange; only rows of the image are stored
                        const unsigned word = split_pair_word(acc[t][mt][j], psel, range_bits);
                        if (ok) reinterpret_cast<unsigned *>(out_split)[(o & ~(size_t)31) + ((l16 & 1) ? 16 : 0) + ((o & 31) >> 1)] = word;
                    }
                }
        }
    }
    if (ovf != nullptr && split_range_overflow(range_bits)) *ovf = 1;
}

} // namespace

bool attention_split_supported(int N, int C) { return N <= 512 && C <= 512 && (C % 32) == 0; }
size_t attention_vt_floats(int B, int N, int C) { return (size_t)B * C * ((N + 31) & ~31); }

// qkv_split: [B][N][3C] in the split operand format; vt: scratch of attention_vt_floats(B, N, C) floats;
// out (fp32) and / or out_split ([B][N][C] split) may be null
double launch_attention_split(const float *qkv_split, float *vt, int B, int N, int C, float *out, float *out_split,
                              int *ovf, hipStream_t s) {
    const int Np = (N + 31) & ~31;
    hipLaunchKernelGGL(attention_vt_kernel, dim3(Np / 32, (C + 127) / 128, B), dim3(256), 0, s, qkv_split, N, C, vt);
    // 64 queries x 8 waves per block where that still leaves a block for every CU (config 3: B = 64, 256 tokens);
    // else 32 queries x 4 waves
    static const int q64 = exp_int("SR3_ATTN_Q64", 1);
    const bool big = q64 && (Np % 64) == 0 && (long)B * (Np / 64) >= 256 && (Np / 16 + 7) / 8 <= 2 && (C / 16 + 7) / 8 <= 4;
    const int QB = big ? 64 : 32, NW = big ? 8 : 4;
    const size_t lds = std::max((size_t)QB * (Np + 8) * sizeof(float), big ? (size_t)QB * (C * 4 + 32) : (size_t)0);
    const int ntw = (Np / 16 + NW - 1) / NW, ntc = (C / 16 + NW - 1) / NW;
    const int dbg = exp_int("SR3_ATTN_DBG", 0);
#define SR3_AT(A, B_, MQ, W_, CH_, KS_)                                                                            \
    {                                                                                                              \
        static size_t attr = 0;                                                                                    \
        if (lds > attr) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attention_split_kernel<A, B_, MQ, W_, CH_, KS_>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
            attr = lds;                                                                                            \
        }                                                                                                          \
        hipLaunchKernelGGL((attention_split_kernel<A, B_, MQ, W_, CH_, KS_>), dim3((Np / QB) * B), dim3(W_ * 64), lds, s, \
                           qkv_split, vt, N, C, out, out_split, ovf, dbg);                                         \
    }
    static const int attn_static = exp_int("SR3_ATTN_STATIC", 1);    // A/B (experiments build): rolled two-set rings
    if (big && attn_static && C == 512 && Np == 256 && dbg == 0) SR3_AT(2, 4, 4, 8, 16, 8)     // config 3: unrolled, deep rings
    else if (big) SR3_AT(2, 4, 4, 8, 0, 0)       // (at most 256 tokens: 2 key tiles per wave)
    else if (ntw <= 2 && ntc <= 2) SR3_AT(2, 2, 2, 4, 0, 0)
    else if (ntw <= 4 && ntc <= 8) SR3_AT(4, 8, 2, 4, 0, 0)
    else SR3_AT(8, 8, 2, 4, 0, 0)
#undef SR3_AT
    return 4.0 * (double)B * N * N * C;
}

double launch_attention(const float *qkv, int B, int N, int C, float *out, hipStream_t s) {
    const int Np = (N + 31) & ~31;
    const size_t lds = (size_t)32 * (Np + 4) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(Np / 32, B), dim3(256), lds, s, qkv, N, C, out);
    return 4.0 * (double)B * N * N * C;
}

// =================================================================================================
// Noise-level embedding (reference unet.py:18-31 PositionalEncoding, :179-184 noise_level_mlp,
// :34-50 FeatureWiseAffine linears of every ResnetBlock, concatenated in module order).
// =================================================================================================
namespace {

__global__ __launch_bounds__(256) void noise_embed_kernel(const EmbedParams p) {
    extern __shared__ float sm[];  // pe[dim] | h[4dim] | te[dim]
    const int dim = p.dim, hid = 4 * dim;
    float *pe = sm, *h = sm + dim, *te = sm + dim + hid;
    const int n = blockIdx.x, t = threadIdx.x;
    const float nl = p.noise_level[(size_t)n * p.nl_stride];
    const int count = dim / 2;
    for (int k = t; k < count; k += blockDim.x) {
        const float step = (float)k / (float)count;
        const float e = nl * expf(-9.210340371976184f * step);
        pe[k] = sinf(e);
        pe[count + k] = cosf(e);
    }
    __syncthreads();
    for (int j = t; j < hid; j += blockDim.x) {
        float a = p.b1[j];
        const float4 *w = reinterpret_cast<const float4 *>(p.w1 + (size_t)j * dim);
        const float4 *v = reinterpret_cast<const float4 *>(pe);
        // 16-byte loads, same k order as a scalar loop (fmaf chain): results unchanged
        for (int k = 0; k < dim / 4; ++k) {
            const float4 ww = w[k], vv = v[k];
            a = fmaf(ww.x, vv.x, a); a = fmaf(ww.y, vv.y, a); a = fmaf(ww.z, vv.z, a); a = fmaf(ww.w, vv.w, a);
        }
        h[j] = a / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int j = t; j < dim; j += blockDim.x) {
        float a = p.b2[j];
        const float4 *w = reinterpret_cast<const float4 *>(p.w2 + (size_t)j * hid);
        const float4 *v = reinterpret_cast<const float4 *>(h);
        for (int k = 0; k < hid / 4; ++k) {
            const float4 ww = w[k], vv = v[k];
            a = fmaf(ww.x, vv.x, a); a = fmaf(ww.y, vv.y, a); a = fmaf(ww.z, vv.z, a); a = fmaf(ww.w, vv.w, a);
        }
        te[j] = a;
        if (p.temb && blockIdx.y == 0) p.temb[(size_t)n * dim + j] = a;
    }
    __syncthreads();
    // the FeatureWiseAffine outputs are spread over blockIdx.y (every block recomputes the tiny MLP above:
    // one block per image left all 7360 dot products of the yml UNet to 256 threads, 86 us per step)
    const int per = (p.total + gridDim.y - 1) / gridDim.y;
    const int j0 = blockIdx.y * per, j1 = min(p.total, j0 + per);
    for (int j = j0 + t; j < j1; j += blockDim.x) {
        float a = p.nfb[j];
        const float4 *w = reinterpret_cast<const float4 *>(p.nfw + (size_t)j * dim);
        const float4 *v = reinterpret_cast<const float4 *>(te);
        for (int k = 0; k < dim / 4; ++k) {
            const float4 ww = w[k], vv = v[k];
            a = fmaf(ww.x, vv.x, a); a = fmaf(ww.y, vv.y, a); a = fmaf(ww.z, vv.z, a); a = fmaf(ww.w, vv.w, a);
        }
        p.chan_bias[(size_t)n * p.total + j] = a;
    }
}

} // namespace

void launch_noise_embed(const EmbedParams &p, int B, hipStream_t s) {
    const size_t lds = (size_t)(6 * p.dim) * sizeof(float);
    const int slices = p.total > 0 ? (p.total + 255) / 256 : 1;
    hipLaunchKernelGGL(noise_embed_kernel, dim3(B, slices), dim3(256), lds, s, p);
}

// =================================================================================================
// Layout changes, Philox normal stream, fused DDPM update
// =================================================================================================
namespace {

__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, int C, const TDesc dst, int coff,
                                    size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*H*W, c fastest
    if (i >= total) return;
    const int HW = dst.H * dst.W;
    const int c = (int)(i % C);
    const size_t np = i / C;          // n*HW + p
    const int n = (int)(np / HW), pp = (int)(np - (size_t)n * HW);
    const int y = pp / dst.W, x = pp - y * dst.W;
    dst.p[dst.pix(n, y, x) * dst.C + coff + c] = in[((size_t)n * C + c) * HW + pp];
}

__global__ void nhwc_to_nchw_kernel(const TDesc src, int coff, int C, float *out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // NCHW order
    if (i >= total) return;
    const int HW = src.H * src.W;
    const int pp = (int)(i % HW);
    const size_t nc = i / HW;
    const int n = (int)(nc / C);
    const int c = (int)(nc - (size_t)n * C);
    const int y = pp / src.W, x = pp - y * src.W;
    out[i] = src.p[src.pix(n, y, x) * src.C + coff + c];
}

// Philox4x32-10 (Salmon et al. 2011). CPU twin: oracle/philox.py.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Standard normal number `elem` of draw `draw` for image `image`: counter = (elem/4, draw,
// image_lo, image_hi), key = seed; Box-Muller on the two 24-bit uniform pairs.
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t image, uint32_t draw,
                                               uint32_t elem) {
    uint32_t r[4];
    philox4x32_10(elem >> 2, draw, (uint32_t)image, (uint32_t)(image >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const int pair = (elem >> 1) & 1;
    const float u1 = ((float)(r[2 * pair] >> 8) + 0.5f) * 5.9604644775390625e-08f;      // 2^-24
    const float u2 = ((float)(r[2 * pair + 1] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * logf(u1));
    const float th = 6.283185307179586f * u2;
    return (elem & 1) ? rad * sinf(th) : rad * cosf(th);
}

__global__ void philox_normal_kernel(uint64_t seed, uint64_t image, uint32_t draw, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = philox_normal(seed, image, draw, (uint32_t)i);
}

__global__ void init_state_kernel(const TDesc state, int xoff, int C, const float *noise, uint64_t seed,
                                  uint64_t image_offset, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW, NCHW order
    if (i >= total) return;
    const int HW = state.H * state.W;
    const int pp = (int)(i % HW);
    const size_t nc = i / HW;
    const int n = (int)(nc / C);
    const int c = (int)(nc - (size_t)n * C);
    const int y = pp / state.W, x = pp - y * state.W;
    const float z = noise ? noise[i] : philox_normal(seed, image_offset + n, 0u, (uint32_t)(c * HW + pp));
    state.p[state.pix(n, y, x) * state.C + xoff + c] = z;
}

// One p_sample tail (reference diffusion.py:144-151 predict_start_from_noise, :175-176 clamp,
// :153-162 q_posterior, :182-187 p_sample), element-wise in the reference's operation order.
// One thread per (image, channel, pixel) in NCHW order (a thread per pixel with its three Philox / Box-Muller
// evaluations in sequence measured 1.5x slower: the kernel is bound by that arithmetic, not by its accesses).
__global__ void ddpm_update_kernel(const UpdateParams u, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW, NCHW order
    if (i >= total) return;
    const StepArgs sa = *u.args;
    const int HW = u.state.H * u.state.W;
    const int pp = (int)(i % HW);
    const size_t nc = i / HW;
    const int n = (int)(nc / u.C);
    const int c = (int)(nc - (size_t)n * u.C);
    const int y = pp / u.state.W, xx = pp - y * u.state.W;
    const size_t pix = u.state.pix(n, y, xx);
    const size_t si = pix * u.state.C + u.xoff + c;
    const float x = u.state.p[si];
    const float e = u.eps.p[u.eps.pix(n, y, xx) * u.eps.C + c];
    float x0 = __fsub_rn(__fmul_rn(sa.a, x), __fmul_rn(sa.b, e));
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    float v = __fadd_rn(__fmul_rn(sa.c1, x0), __fmul_rn(sa.c2, x));
    if (sa.sigma != 0.f) {
        const float z = sa.noise ? sa.noise[i]
                                 : philox_normal(sa.seed, sa.image_offset + n, sa.draw, (uint32_t)(c * HW + pp));
        v = __fadd_rn(v, __fmul_rn(z, sa.sigma));
    }
    u.state.p[si] = v;
    if (u.packed != nullptr) {          // the first conv reads the state as packed split-f16 pixels (kernels_edge.hip)
        _Float16 *pk = reinterpret_cast<_Float16 *>(u.packed) + pix * 16 + u.xoff + c;
        const _Float16 hi = (_Float16)v;        // |v| is bounded by the clamp of x0 and the noise — an injected noise slab may
        pk[0] = hi;                             // still be anything: detected like every other store of the format
        pk[8] = (_Float16)(v - (float)hi);
        if (u.ovf != nullptr && ((unsigned)__builtin_bit_cast(unsigned short, hi) & 0x7C00u) == 0x7C00u) *u.ovf = 1;
    }
    if (sa.frame) sa.frame[i] = v;
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

} // namespace

void launch_nchw_to_nhwc(const float *in, int B, int C, const TDesc &dst, int coff, hipStream_t s) {
    const size_t total = (size_t)B * C * dst.H * dst.W;
    if (!total) return;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblk(total)), dim3(256), 0, s, in, C, dst, coff, total);
}
void launch_nhwc_to_nchw(const TDesc &src, int coff, int B, int C, float *out, hipStream_t s) {
    const size_t total = (size_t)B * C * src.H * src.W;
    if (!total) return;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk(total)), dim3(256), 0, s, src, coff, C, out, total);
}
void launch_ddpm_update(const UpdateParams &p, int B, hipStream_t s) {
    const size_t total = (size_t)B * p.C * p.state.H * p.state.W;
    hipLaunchKernelGGL(ddpm_update_kernel, dim3(nblk(total)), dim3(256), 0, s, p, total);
}
void launch_init_state(const TDesc &state, int xoff, int C, const float *noise, uint64_t seed,
                       uint64_t image_offset, int B, hipStream_t s) {
    const size_t total = (size_t)B * C * state.H * state.W;
    hipLaunchKernelGGL(init_state_kernel, dim3(nblk(total)), dim3(256), 0, s, state, xoff, C, noise, seed,
                       image_offset, total);
}
void launch_philox_normal(uint64_t seed, uint64_t image, uint32_t draw, int n, float *out,
                          hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(nblk(n)), dim3(256), 0, s, seed, image, draw, n, out);
}

} // namespace sr3
