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

namespace sr3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;          // channels per K-step
constexpr int LDSK = BK + 4;    // padded LDS row (floats)

__device__ __forceinline__ float swish_f(float x) {
    // x * sigmoid(x); v_exp_f32 / v_rcp_f32 are 1 ulp on gfx950
    return x * __frcp_rn(1.0f + __expf(-x));
}

template <int BM, int BN, int WGM, int WGN>
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

    float4 ra[AR], rsc[AR], rsh[AR], rb[BR];
    unsigned vmask = 0;
    const bool has_gn = p.gn_scale != nullptr;

    auto issue_loads = [&](int kidx) {
        const int cc = kidx / taps;
        const int tap = kidx - cc * taps;
        const int c0 = cc * BK;
        const int dy = tap / p.ks, dx = tap - dy * p.ks;
        const float *src;
        int Cs, cl;
        if (c0 < p.C0) {
            src = p.in0; Cs = p.C0; cl = c0;
        } else {
            src = p.in1; Cs = p.C1; cl = c0 - p.C0;
        }
        vmask = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int uy = a_uy[i] + dy, ux = a_ux[i] + dx;
            const bool ok = (unsigned)uy < (unsigned)Hv && (unsigned)ux < (unsigned)Wv;
            if (ok) {
                const int iy = uy >> p.up2, ix = ux >> p.up2;
                const size_t off = ((size_t)(a_n[i] * p.Hin + iy) * p.Win + ix) * Cs + cl + 4 * q;
                ra[i] = *reinterpret_cast<const float4 *>(src + off);
                vmask |= 1u << i;
                if (has_gn) {
                    const size_t go = (size_t)a_n[i] * Cin + c0 + 4 * q;
                    rsc[i] = *reinterpret_cast<const float4 *>(p.gn_scale + go);
                    rsh[i] = *reinterpret_cast<const float4 *>(p.gn_shift + go);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + r0 + 32 * i;
            if (n < p.Cout) {
                rb[i] = *reinterpret_cast<const float4 *>(p.w + ((size_t)tap * p.Cout + n) * Cin + c0 + 4 * q);
            } else {
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    auto stage_to_lds = [&](int buf) {
        float *Ad = As + buf * BM * LDSK;
        float *Bd = Bs + buf * BN * LDSK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vmask & (1u << i)) {
                v = ra[i];
                if (has_gn) {
                    v.x = fmaf(v.x, rsc[i].x, rsh[i].x);
                    v.y = fmaf(v.y, rsc[i].y, rsh[i].y);
                    v.z = fmaf(v.z, rsc[i].z, rsh[i].z);
                    v.w = fmaf(v.w, rsc[i].w, rsh[i].w);
                    if (p.swish) {
                        v.x = swish_f(v.x); v.y = swish_f(v.y);
                        v.z = swish_f(v.z); v.w = swish_f(v.w);
                    }
                }
            }
            *reinterpret_cast<float4 *>(Ad + (r0 + 32 * i) * LDSK + 4 * q) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            *reinterpret_cast<float4 *>(Bd + (r0 + 32 * i) * LDSK + 4 * q) = rb[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue_loads(0);
    stage_to_lds(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) issue_loads(kt + 1);

        const float *Ab = As + cur * BM * LDSK + (wm * WM + li) * LDSK + 4 * lh;
        const float *Bb = Bs + cur * BN * LDSK + (wn * WN + li) * LDSK + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 av[MI], bv[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                av[mi] = *reinterpret_cast<const float4 *>(Ab + mi * 32 * LDSK + kk * 8);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bv[ni] = *reinterpret_cast<const float4 *>(Bb + ni * 32 * LDSK + kk * 8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].x, bv[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].y, bv[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].z, bv[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi].w, bv[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (more) stage_to_lds(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * WN + ni * 32 + li;
        if (n >= p.Cout) continue;
        const float bs = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float v = acc[mi][ni][r] + bs;
                    if (p.chan_bias) v += p.chan_bias[(size_t)(m / HWo) * p.chan_bias_stride + n];
                    const size_t o = (size_t)m * p.Cout + n;
                    if (p.resid) v += p.resid[o];
                    p.out[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDSK * sizeof(float);
    auto kern = conv_igemm_f32<BM, BN, WGM, WGN>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    const int tilesM = (M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(256), lds, s, p);
}

} // namespace

double launch_conv(const ConvParams &p, hipStream_t s) {
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
