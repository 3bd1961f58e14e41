// Conv2d (1x1 / 3x3, stride 1|2, optional nearest-x2 upsample, optional channel concat) as an
// implicit GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the SR3 UNet (reference model/sr/sr3_modules/unet.py):
//   Block        GroupNorm -> Swish -> Conv3x3            :80-91   (GN+Swish folded into the A-tile fill)
//   ResnetBlock  + FeatureWiseAffine bias, + residual     :94-110  (fused epilogue)
//   Upsample / Downsample                                 :58-74   (index remap in the gather)
//   torch.cat((x, skip), 1)                               :261     (dual-pointer K range)
//   SelfAttention.qkv / .out 1x1 convs                    :120-121
//
// GEMM view: M = B*Hout*Wout output pixels, N = Cout, K = ks*ks*Cin.  A (activations, NHWC) is
// gathered per (tap, 32-channel chunk) with the zero padding applied AFTER the folded
// GroupNorm+Swish; B (weights) is pre-packed [tap][Cout][Cin] so both operands are K-contiguous.
// Block = 256 threads = 4 waves; each wave owns a (32*MI) x (32*NI) tile of 32x32 accumulators.
// LDS rows are padded to 36 floats: a lane's ds_read_b128 of 4 consecutive k lands on a distinct
// 16-B slot for every row of its 16-lane group (row stride 144 B = 9 slots, 9 odd).
// The 4 k-values a lane reads are fed to 4 consecutive MFMAs; lane half h supplies k = 8kk+4h+j to
// MFMA j of group kk for both operands, so the k-permutation is consistent between A and B.
#include "sr3_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace sr3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector (not HIP's float4 struct: struct copies through a register array become
// memcpys via scratch memory)
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;          // channels per K-step
constexpr int LDSK = BK + 4;    // padded LDS row (floats)

// compile-time loop: every index is a constant in the front end, so register arrays are split
// into scalars before any loop pass (runtime-indexed arrays end up in scratch, guide rule 20)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float swish_f(float x) {
    // x * sigmoid(x); v_exp_f32 / v_rcp_f32 are 1 ulp on gfx950
    return x * __frcp_rn(1.0f + __expf(-x));
}

// MODE 0: raw input, 1: per-(image, channel) affine (GroupNorm folded), 2: affine + Swish
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;  // float4 rows per thread for the A / B tile
    static_assert(MI >= 1 && NI >= 1, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [2][BM][LDSK]
    float *Bs = smem + 2 * BM * LDSK;       // [2][BN][LDSK]

    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int M = p.B * HWo;
    const int tilesN = (p.Cout + BN - 1) / BN;

    // XCD-aware block remap (bijective): blocks b and b+8 share an XCD (speed only), so give each
    // XCD a contiguous range of logical tiles; the n-tiles of one m-tile then share one L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, loc = bid >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;

    const int tid = threadIdx.x;
    const int q = tid & 7;     // float4 column inside the 32-wide K chunk
    const int r0 = tid >> 3;   // 0..31
    const int lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;

    const int pad = p.ks >> 1;
    const int Hv = p.Hin << p.up2, Wv = p.Win << p.up2;
    const int taps = p.ks * p.ks;
    const int nk = taps * (Cin / BK);

    // per-thread row bookkeeping (constant over K)
    int a_n[AR], a_uy[AR], a_ux[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < M) {
            const int n = m / HWo;
            const int rem = m - n * HWo;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            a_n[i] = n;
            a_uy[i] = oy * p.stride - pad;
            a_ux[i] = ox * p.stride - pad;
        } else {
            a_n[i] = 0;
            a_uy[i] = -(1 << 20);
            a_ux[i] = -(1 << 20);
        }
    }
    // weight rows are clamped (rows past Cout are never stored); per-thread element offsets
    const float *b_ptr[BR];
    static_for<BR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        b_ptr[i] = p.w + (size_t)min(n0 + r0 + 32 * i, p.Cout - 1) * Cin + 4 * q;
    });

    f32x4 ra[AR], rsc[AR], rsh[AR], rb[BR];
    unsigned vmask = 0;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // The three pipeline stages are macros, not lambdas: arrays captured by reference in a lambda
    // were left in scratch memory by hipcc (ROCm 7.2), which serialised every K-step on the loads.

    // Branch-free tile fetch: out-of-window taps load a clamped (valid) address and are zeroed
    // when the tile is written to LDS, so the main loop is one basic block.
#define SR3_ISSUE_LOADS(KIDX)                                                                      \
    {                                                                                              \
        const int kidx_ = (KIDX);                                                                  \
        const int cc_ = kidx_ / taps;                                                              \
        const int tap_ = kidx_ - cc_ * taps;                                                       \
        const int c0_ = cc_ * BK;                                                                  \
        const int dy_ = tap_ / p.ks, dx_ = tap_ - dy_ * p.ks;                                      \
        const bool first_ = c0_ < p.C0;                                                            \
        const float *src_ = first_ ? p.in0 : p.in1;                                                \
        const int Cs_ = first_ ? p.C0 : p.C1;                                                      \
        const int cl_ = first_ ? c0_ : c0_ - p.C0;                                                 \
        vmask = 0;                                                                                 \
        static_for<AR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            const int uy = a_uy[i] + dy_, ux = a_ux[i] + dx_;                                      \
            const bool ok = (unsigned)uy < (unsigned)Hv && (unsigned)ux < (unsigned)Wv;            \
            const int iy = min(max(uy, 0), Hv - 1) >> p.up2, ix = min(max(ux, 0), Wv - 1) >> p.up2; \
            const size_t off = ((size_t)(a_n[i] * p.Hin + iy) * p.Win + ix) * Cs_ + cl_ + 4 * q;   \
            ra[i] = *reinterpret_cast<const f32x4 *>(src_ + off);                                 \
            vmask |= ok ? (1u << i) : 0u;                                                          \
            if (MODE != 0) {                                                                       \
                const size_t go = (size_t)a_n[i] * Cin + c0_ + 4 * q;                              \
                rsc[i] = *reinterpret_cast<const f32x4 *>(p.gn_scale + go);                       \
                rsh[i] = *reinterpret_cast<const f32x4 *>(p.gn_shift + go);                       \
            }                                                                                      \
        });                                                                                        \
        static_for<BR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr[i] + (size_t)tap_ * p.Cout * Cin + c0_); \
        });                                                                                        \
    }

#define SR3_STAGE_TO_LDS(BUF)                                                                      \
    {                                                                                              \
        float *Ad = As + (BUF) * BM * LDSK;                                                        \
        float *Bd = Bs + (BUF) * BN * LDSK;                                                        \
        static_for<AR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            f32x4 v = ra[i];                                                                      \
            if (MODE != 0) {                                                                       \
                v.x = fmaf(v.x, rsc[i].x, rsh[i].x);                                               \
                v.y = fmaf(v.y, rsc[i].y, rsh[i].y);                                               \
                v.z = fmaf(v.z, rsc[i].z, rsh[i].z);                                               \
                v.w = fmaf(v.w, rsc[i].w, rsh[i].w);                                               \
            }                                                                                      \
            if (MODE == 2) {                                                                       \
                v.x = swish_f(v.x); v.y = swish_f(v.y);                                            \
                v.z = swish_f(v.z); v.w = swish_f(v.w);                                            \
            }                                                                                      \
            const bool ok = (vmask >> i) & 1u;                                                     \
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; \
            *reinterpret_cast<f32x4 *>(Ad + (r0 + 32 * i) * LDSK + 4 * q) = v;                    \
        });                                                                                        \
        static_for<BR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            *reinterpret_cast<f32x4 *>(Bd + (r0 + 32 * i) * LDSK + 4 * q) = rb[i];                \
        });                                                                                        \
    }

#define SR3_MMA_GROUPS(CUR, KK0, KK1)                                                              \
    {                                                                                              \
        const float *Ab = As + (CUR) * BM * LDSK + (wm * WM + li) * LDSK + 4 * lh;                 \
        const float *Bb = Bs + (CUR) * BN * LDSK + (wn * WN + li) * LDSK + 4 * lh;                 \
        _Pragma("unroll") for (int kk = (KK0); kk < (KK1); ++kk) {                                 \
            f32x4 av[MI], bv[NI];                                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                      \
                av[mi] = *reinterpret_cast<const f32x4 *>(Ab + mi * 32 * LDSK + kk * 8);          \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                      \
                bv[ni] = *reinterpret_cast<const f32x4 *>(Bb + ni * 32 * LDSK + kk * 8);          \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                    \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].x, bv[ni].x, acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].y, bv[ni].y, acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].z, bv[ni].z, acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].w, bv[ni].w, acc[mi][ni], 0, 0, 0); \
            }                                                                                      \
        }                                                                                          \
    }

    // Pipeline: while tile kt is multiplied out of LDS buffer kt&1, tile kt+1 (already in
    // registers) is transformed and written to the other buffer in the shadow of the MFMAs, then
    // the global loads of tile kt+2 are issued. Loads past the end re-fetch the last tile (unused).
    SR3_ISSUE_LOADS(0)
    SR3_STAGE_TO_LDS(0)
    SR3_ISSUE_LOADS(min(1, nk - 1))
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        SR3_MMA_GROUPS(cur, 0, BK / 16)
        SR3_STAGE_TO_LDS(cur ^ 1)
        SR3_MMA_GROUPS(cur, BK / 16, BK / 8)
        SR3_ISSUE_LOADS(min(kt + 2, nk - 1))
        __syncthreads();
    }
#undef SR3_ISSUE_LOADS
#undef SR3_STAGE_TO_LDS
#undef SR3_MMA_GROUPS

    // ---- epilogue -------------------------------------------------------------------------
    // LDS is free now: one image index per tile row (one integer division per thread).
    int *rowimg = reinterpret_cast<int *>(smem);
    if (p.chan_bias != nullptr) {
        if (tid < BM) rowimg[tid] = min(m0 + tid, M - 1) / HWo;
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * WN + ni * 32 + li;
        const int nc = min(n, p.Cout - 1);
        const float bs = p.bias ? p.bias[nc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rbase = wm * WM + mi * 32 + 4 * lh;
            float add[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) add[r] = bs;
            if (p.resid != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(m0 + rbase + (r & 3) + 8 * (r >> 2), M - 1);
                    add[r] += p.resid[(size_t)m * p.Cout + nc];
                }
            }
            if (p.chan_bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int img = rowimg[rbase + (r & 3) + 8 * (r >> 2)];
                    add[r] += p.chan_bias[(size_t)img * p.chan_bias_stride + nc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + rbase + (r & 3) + 8 * (r >> 2);
                if (m < M && n < p.Cout) p.out[(size_t)m * p.Cout + n] = acc[mi][ni][r] + add[r];
            }
        }
    }
}

// =================================================================================================
// Wave-specialised variant: 512 threads = 4 consumer waves (MFMA + LDS fragment reads only) and
// 4 producer waves (global gather, GroupNorm/Swish transform, LDS writes). The hardware places the
// waves of a workgroup round-robin over the 4 SIMDs, so each SIMD hosts one consumer and one
// producer per resident block; the matrix pipe and the VALU run side by side and the consumer's
// instruction stream never waits on global memory. Same tiles, same LDS image, same numerics
// (identical k order per accumulator) as conv_igemm_f32. One barrier per K-step:
//   step k: consumers multiply tile k out of buffer k&1 | producers write tile k+1 into buffer
//   (k+1)&1 (free since the barrier of step k-1) and issue the global loads of tile k+2.
// =================================================================================================
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(512, 4) void conv_igemm_ws_f32(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 consumer waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [2][BM][LDSK]
    float *Bs = smem + 2 * BM * LDSK;       // [2][BN][LDSK]
    int *rowimg = reinterpret_cast<int *>(smem + 2 * (BM + BN) * LDSK);  // [BM]

    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int M = p.B * HWo;
    const int tilesN = (p.Cout + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, loc = bid >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;
    const int taps = p.ks * p.ks;
    const int nk = taps * (Cin / BK);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    if (wid >= 4) {
        // ------------------------------- producer waves -------------------------------------
        // Address arithmetic is kept to 32-bit / 24-bit integer ops: the producers share their
        // SIMD's vector issue port with the consumer's MFMAs, and quarter-rate 64-bit multiplies
        // in this loop measurably slowed the matrix pipe.
        if (p.dbg & 16) __builtin_amdgcn_s_setprio(2);
        const int tid = threadIdx.x - 256;
        const int q = tid & 7, r0 = tid >> 3;
        const int pad = p.ks >> 1;
        const int Hv1 = (p.Hin << p.up2) - 1, Wv1 = (p.Win << p.up2) - 1;
        if (tid < BM) rowimg[tid] = min(m0 + tid, M - 1) / HWo;

        // per row: image base pointers (both concat halves), GroupNorm row pointers, window origin
        const float *a_p0[AR], *a_p1[AR], *g_sc[AR], *g_sh[AR];
        int a_uy[AR], a_ux[AR];
        static_for<AR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = m0 + r0 + 32 * i;
            int n = 0, uy = -(1 << 20), ux = -(1 << 20);
            if (m < M) {
                n = m / HWo;
                const int rem = m - n * HWo;
                const int oy = rem / p.Wout;
                const int ox = rem - oy * p.Wout;
                uy = oy * p.stride - pad;
                ux = ox * p.stride - pad;
            }
            a_uy[i] = uy;
            a_ux[i] = ux;
            const size_t img = (size_t)n * p.Hin * p.Win;
            a_p0[i] = p.in0 + img * p.C0 + 4 * q;
            a_p1[i] = p.C1 ? p.in1 + img * p.C1 + 4 * q : a_p0[i];
            if (MODE != 0) {
                g_sc[i] = p.gn_scale + (size_t)n * Cin + 4 * q;
                g_sh[i] = p.gn_shift + (size_t)n * Cin + 4 * q;
            }
        });
        const float *b_ptr[BR];
        static_for<BR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            b_ptr[i] = p.w + (size_t)min(n0 + r0 + 32 * i, p.Cout - 1) * Cin + 4 * q;
        });
        const int tapstride = p.Cout * Cin;   // floats between taps of the packed weights
        f32x4 ra[AR], rsc[AR], rsh[AR], rb[BR];
        unsigned vmask = 0;
        // (tap, chunk) counters advanced incrementally: no division in the loop
        int l_tap = 0, l_c0 = 0, l_dy = 0, l_dx = 0;

#define SR3_ISSUE_LOADS_NEXT()                                                                     \
    {                                                                                              \
        const bool first_ = l_c0 < p.C0;                                                           \
        const int Cs_ = first_ ? p.C0 : p.C1;                                                      \
        const int cl_ = first_ ? l_c0 : l_c0 - p.C0;                                               \
        vmask = 0;                                                                                 \
        static_for<AR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            const int uy = a_uy[i] + l_dy, ux = a_ux[i] + l_dx;                                    \
            const bool ok = (unsigned)uy <= (unsigned)Hv1 && (unsigned)ux <= (unsigned)Wv1;        \
            const int iy = min(max(uy, 0), Hv1) >> p.up2, ix = min(max(ux, 0), Wv1) >> p.up2;      \
            const int pix = __mul24(iy, p.Win) + ix;                                               \
            const unsigned off = __umul24((unsigned)pix, (unsigned)Cs_) + (unsigned)cl_;                  \
            ra[i] = *reinterpret_cast<const f32x4 *>((first_ ? a_p0[i] : a_p1[i]) + off);          \
            vmask |= ok ? (1u << i) : 0u;                                                          \
            if (MODE != 0) {                                                                       \
                rsc[i] = *reinterpret_cast<const f32x4 *>(g_sc[i] + l_c0);                         \
                rsh[i] = *reinterpret_cast<const f32x4 *>(g_sh[i] + l_c0);                         \
            }                                                                                      \
        });                                                                                        \
        const unsigned woff_ = (unsigned)l_tap * (unsigned)tapstride + (unsigned)l_c0;             \
        static_for<BR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr[i] + woff_);                            \
        });                                                                                        \
        /* advance to the next (chunk, tap): tap inner, chunk outer; saturate at the last tile */  \
        if (l_tap + 1 < taps) {                                                                    \
            ++l_tap;                                                                               \
            if (++l_dx == p.ks) { l_dx = 0; ++l_dy; }                                              \
        } else if (l_c0 + BK < Cin) {                                                              \
            l_tap = 0; l_dx = 0; l_dy = 0; l_c0 += BK;                                             \
        }                                                                                          \
    }

#define SR3_STAGE_TO_LDS(BUF)                                                                      \
    {                                                                                              \
        float *Ad = As + (BUF) * BM * LDSK;                                                        \
        float *Bd = Bs + (BUF) * BN * LDSK;                                                        \
        static_for<AR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            f32x4 v = ra[i];                                                                       \
            if (MODE != 0) {                                                                       \
                v.x = fmaf(v.x, rsc[i].x, rsh[i].x);                                               \
                v.y = fmaf(v.y, rsc[i].y, rsh[i].y);                                               \
                v.z = fmaf(v.z, rsc[i].z, rsh[i].z);                                               \
                v.w = fmaf(v.w, rsc[i].w, rsh[i].w);                                               \
            }                                                                                      \
            if (MODE == 2) {                                                                       \
                v.x = swish_f(v.x); v.y = swish_f(v.y);                                            \
                v.z = swish_f(v.z); v.w = swish_f(v.w);                                            \
            }                                                                                      \
            const bool ok = (vmask >> i) & 1u;                                                     \
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; \
            *reinterpret_cast<f32x4 *>(Ad + (r0 + 32 * i) * LDSK + 4 * q) = v;                     \
        });                                                                                        \
        static_for<BR>([&](auto ic) {                                                              \
            constexpr int i = decltype(ic)::value;                                                 \
            *reinterpret_cast<f32x4 *>(Bd + (r0 + 32 * i) * LDSK + 4 * q) = rb[i];                 \
        });                                                                                        \
    }

        SR3_ISSUE_LOADS_NEXT()          // tile 0
        SR3_STAGE_TO_LDS(0)
        SR3_ISSUE_LOADS_NEXT()          // tile 1 (or tile 0 again if nk == 1)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk && !(p.dbg & 1)) {
                if (!(p.dbg & 4)) {
                    SR3_STAGE_TO_LDS((kt & 1) ^ 1)
                } else {
                    _Pragma("unroll") for (int i = 0; i < AR; ++i) asm volatile("" ::"v"(ra[i]));
                    _Pragma("unroll") for (int i = 0; i < BR; ++i) asm volatile("" ::"v"(rb[i]));
                }
                if (!(p.dbg & 2)) {
                    SR3_ISSUE_LOADS_NEXT()   // tile kt + 2
                }
                if (p.dbg >> 8) {           // experiment: extra independent VALU work per step
                    float e0 = (float)kt, e1 = e0 + 1.f, e2 = e0 + 2.f, e3 = e0 + 3.f;
                    for (int j = 0; j < (p.dbg >> 8); ++j) {
                        e0 = fmaf(e0, 1.0001f, 0.5f); e1 = fmaf(e1, 1.0001f, 0.5f);
                        e2 = fmaf(e2, 1.0001f, 0.5f); e3 = fmaf(e3, 1.0001f, 0.5f);
                    }
                    asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(e3));
                }
            }
            __syncthreads();
        }
#undef SR3_ISSUE_LOADS_NEXT
#undef SR3_STAGE_TO_LDS
        return;
    }

    // ----------------------------------- consumer waves -----------------------------------------
    if (p.dbg & 8) __builtin_amdgcn_s_setprio(2);
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // Fragment reads run one 8-k group ahead of the MFMAs (two register sets); the reads of the
    // next tile's first group are issued right after the barrier and land under the last group's
    // 16 MFMAs, so the consumer never waits on LDS latency.
    const float *Abase = As + (wm * WM + li) * LDSK + 4 * lh;
    const float *Bbase = Bs + (wn * WN + li) * LDSK + 4 * lh;
    f32x4 fa[2][MI], fb[2][NI];
#define SR3_FRAG_READ(SET, CUR, KK)                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) fa[SET][mi] =                            \
            *reinterpret_cast<const f32x4 *>(Abase + (CUR) * BM * LDSK + mi * 32 * LDSK + (KK) * 8); \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) fb[SET][ni] =                            \
            *reinterpret_cast<const f32x4 *>(Bbase + (CUR) * BN * LDSK + ni * 32 * LDSK + (KK) * 8); \
    }
#define SR3_FRAG_MMA(SET)                                                                          \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                        \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][mi].x, fb[SET][ni].x, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][mi].y, fb[SET][ni].y, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][mi].z, fb[SET][ni].z, acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][mi].w, fb[SET][ni].w, acc[mi][ni], 0, 0, 0); \
        }                                                                                          \
    }
    __syncthreads();
    SR3_FRAG_READ(0, 0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        SR3_FRAG_READ(1, cur, 1)
        SR3_FRAG_MMA(0)
        SR3_FRAG_READ(0, cur, 2)
        SR3_FRAG_MMA(1)
        SR3_FRAG_READ(1, cur, 3)
        SR3_FRAG_MMA(0)
        __syncthreads();                       // every read of tile kt has been issued and waited
        SR3_FRAG_READ(0, cur ^ 1, 0)           // next tile (garbage after the last one: unused)
        SR3_FRAG_MMA(1)
    }
#undef SR3_FRAG_READ
#undef SR3_FRAG_MMA
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue (consumer waves; rowimg was written by the producers before the first barrier)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * WN + ni * 32 + li;
        const int nc = min(n, p.Cout - 1);
        const float bs = p.bias ? p.bias[nc] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rbase = wm * WM + mi * 32 + 4 * lh;
            float add[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) add[r] = bs;
            if (p.resid != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(m0 + rbase + (r & 3) + 8 * (r >> 2), M - 1);
                    add[r] += p.resid[(size_t)m * p.Cout + nc];
                }
            }
            if (p.chan_bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int img = rowimg[rbase + (r & 3) + 8 * (r >> 2)];
                    add[r] += p.chan_bias[(size_t)img * p.chan_bias_stride + nc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + rbase + (r & 3) + 8 * (r >> 2);
                if (m < M && n < p.Cout) p.out[(size_t)m * p.Cout + n] = acc[mi][ni][r] + add[r];
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int MODE>
void launch_inst_ws(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t lds = ((size_t)2 * (BM + BN) * LDSK + BM) * sizeof(float);
    auto kern = conv_igemm_ws_f32<BM, BN, WGM, WGN, MODE>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    const int tilesM = (M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(512), lds, s, p);
}

template <int BM, int BN, int WGM, int WGN, int MODE>
void launch_inst(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDSK * sizeof(float);
    auto kern = conv_igemm_f32<BM, BN, WGM, WGN, MODE>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    const int tilesM = (M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(256), lds, s, p);
}

int conv_impl() {
    static int impl = -1;
    if (impl < 0) {
        const char *e = getenv("SR3_CONV_IMPL");
        impl = e ? atoi(e) : 2;
    }
    return impl;
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const ConvParams &p, hipStream_t s) {
    if (conv_impl() == 1) {
        if (p.gn_scale == nullptr) launch_inst<BM, BN, WGM, WGN, 0>(p, s);
        else if (!p.swish) launch_inst<BM, BN, WGM, WGN, 1>(p, s);
        else launch_inst<BM, BN, WGM, WGN, 2>(p, s);
    } else {
        if (p.gn_scale == nullptr) launch_inst_ws<BM, BN, WGM, WGN, 0>(p, s);
        else if (!p.swish) launch_inst_ws<BM, BN, WGM, WGN, 1>(p, s);
        else launch_inst_ws<BM, BN, WGM, WGN, 2>(p, s);
    }
}

} // namespace

double launch_conv(const ConvParams &p_in, hipStream_t s) {
    ConvParams p = p_in;
    if (const char *e = getenv("SR3_CONV_DBG")) p.dbg = atoi(e);
    const long M = (long)p.B * p.Hout * p.Wout;
    const int Cin = p.C0 + p.C1;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    const long want = 512;  // 256 CUs x 2 resident blocks
    if (p.Cout <= 32) {
        launch_cfg<128, 32, 4, 1>(p, s);
    } else if (p.Cout <= 64 || (p.Cout % 128) != 0) {
        if (blocks(128, 64) >= want) launch_cfg<128, 64, 2, 2>(p, s);
        else launch_cfg<64, 64, 2, 2>(p, s);
    } else {
        if (blocks(128, 128) >= want) launch_cfg<128, 128, 2, 2>(p, s);
        else if (blocks(128, 64) >= want) launch_cfg<128, 64, 2, 2>(p, s);
        else launch_cfg<64, 64, 2, 2>(p, s);
    }
    return 2.0 * (double)M * p.Cout * (double)(p.ks * p.ks) * Cin;
}

void pack_conv_weight(const float *oihw, int Cout, int Cin, int ks, int CinPad, float *dst) {
    const int taps = ks * ks;
    for (int t = 0; t < taps; ++t)
        for (int o = 0; o < Cout; ++o) {
            float *d = dst + ((size_t)t * Cout + o) * CinPad;
            for (int i = 0; i < Cin; ++i) d[i] = oihw[((size_t)o * Cin + i) * taps + t];
            for (int i = Cin; i < CinPad; ++i) d[i] = 0.f;
        }
}

} // namespace sr3
