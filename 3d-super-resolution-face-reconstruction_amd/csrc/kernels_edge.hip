// The two badly shaped convolutions of the SR3 UNet, each as a kernel of its own (gfx950):
//
//   final_conv  = GroupNorm -> Swish -> Conv3x3(C -> 3)   reference unet.py:80-91 (Block), :229 (final_conv)
//                 3 output channels: as an implicit GEMM 29 of 32 MFMA columns were padding and the
//                 activated input cost an HBM pass of its own. Here: ONE kernel reads the raw tensor,
//                 applies the folded GroupNorm affine + Swish in registers and contracts with exact fp32
//                 FMAs (3.6 GFLOP per 64 images: VALU work, no matrix cores, no LDS staging of pixels).
//
//   downs.0     = Conv3x3(6 -> 64) on cat([cond, x_t])     reference unet.py:193-194, diffusion.py:170
//                 6 input channels: as an implicit GEMM every K-step multiplied 26 zero channels. Here
//                 the sampler state is ALSO kept as 8-channel pixels in the split-f16 operand format
//                 (16 B hi | 16 B lo per pixel), so the three dx taps of a row are 24 consecutive
//                 k-values of ONE v_mfma_f32_16x16x32_f16: 3 K-steps instead of 9 x 32-channel steps,
//                 A fragments straight from global memory (no LDS), weights resident in registers.
#include "sr3_internal.h"
#include <math.h>
#include <type_traits>

namespace sr3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float swish_fast(float x) {
    // x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (1 ulp each on gfx950)
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// -------------------------------------------------------------------------------------------------
// final_conv. Thread = (pixel group, channel quad): a group of 16 lanes owns a 2 x 4 block of output
// pixels, lane j of the group handles input channels [4j, 4j + 4) (+ 64 per pass) of the 4 x 6 input
// window — its loads are one coalesced 256-B pixel row per 16 lanes — and accumulates partial sums for
// all 2 x 4 x COUT outputs; a 4-step butterfly over the 16 lanes adds the channel quads at the end.
// Zero padding is applied AFTER the activation (the reference pads the activated tensor): window
// pixels outside the image contribute 0, not swish(shift).
// -------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256, 3) void final_conv_kernel(const TDesc x, const float *__restrict__ scale,
                                                         const float *__restrict__ shift,
                                                         const float *__restrict__ wq,     // [9][C][4] (cout padded to 4)
                                                         const float *__restrict__ bias, const TDesc out) {
    extern __shared__ __attribute__((aligned(16))) float fc_w[];       // [9][C][4]
    const int C = x.C, H = x.H, W = x.W;
    for (int i = threadIdx.x; i < 9 * C; i += 256)
        reinterpret_cast<f32x4 *>(fc_w)[i] = reinterpret_cast<const f32x4 *>(wq)[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int x0 = blockIdx.x * 32 + (pg & 7) * 4, y0 = blockIdx.y * 4 + (pg >> 3) * 2, n = blockIdx.z;
    float acc[2][4][COUT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[a][b][o] = 0.f;
    // window columns x0 - 1 + p (p = 0..5) and rows y0 - 1 + r (r = 0..3): validity and clamped
    // coordinates (clamped ones stay inside the zero-bordered storage; their values are masked)
    bool cv[6];
    int cx[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int xx = x0 - 1 + p;
        cv[p] = xx >= 0 && xx < W;
        cx[p] = min(max(xx, -1), W);
    }
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + l16 * 4;
        const f32x4 scv = *reinterpret_cast<const f32x4 *>(scale + (size_t)n * C + c);
        const f32x4 shv = *reinterpret_cast<const f32x4 *>(shift + (size_t)n * C + c);
        const float sc[4] = {scv.x, scv.y, scv.z, scv.w}, sh[4] = {shv.x, shv.y, shv.z, shv.w};
        // compile-time loops over the window rows and taps: every array index is a constant in the
        // front end (runtime-indexed register arrays end up in scratch memory). The window rows are
        // software-pipelined: row r + 1 is in flight while row r is activated and multiplied.
        float tw[2][6][4];
        auto load_row = [&](auto bufc, int r) {
            constexpr int buf = decltype(bufc)::value;
            const int cy = min(max(y0 - 1 + r, -1), H);
            static_for<6>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x.p + x.pix(n, cy, cx[p]) * (size_t)C + c);
                tw[buf][p][0] = v.x; tw[buf][p][1] = v.y; tw[buf][p][2] = v.z; tw[buf][p][3] = v.w;
            });
        };
        // the 18 (window row, tap) steps in order: pair = step / 3 -> (r, dy) = ((pair + 1) / 2, pair / 2), dx = step % 3;
        // the weights of step s + 1 are fetched from LDS while step s multiplies (two register sets)
        float wv[2][4][4];                  // wv[set][j][o]: weight of input channel c + j for output o
        auto load_w = [&](auto setc, int tap) {
            constexpr int set = decltype(setc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(fc_w + ((size_t)tap * C + c + j) * 4);
                wv[set][j][0] = w4.x; wv[set][j][1] = w4.y; wv[set][j][2] = w4.z; wv[set][j][3] = w4.w;
            }
        };
        load_row(std::integral_constant<int, 0>{}, 0);
        load_w(std::integral_constant<int, 0>{}, 0);
        static_for<4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int cur = r & 1;
            if constexpr (r < 3) load_row(std::integral_constant<int, cur ^ 1>{}, r + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int yy = y0 - 1 + r;
            const bool rv = yy >= 0 && yy < H;
            float (&t)[6][4] = tw[cur];
            static_for<6>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const bool ok = rv && cv[p];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = swish_fast(fmaf(t[p][j], sc[j], sh[j]));
                    t[p][j] = ok ? v : 0.f;
                }
            });
            // scheduling fences keep one step's work together (without them the compiler hoists the loads
            // and weights of all four rows: 256 VGPRs + spills)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int pair_lo = r == 0 ? 0 : 2 * r - 1, pair_hi = r == 3 ? 5 : 2 * r;   // (r, dy) pairs of this row
            static_for<(pair_hi - pair_lo + 1) * 3>([&](auto sc_) {
                constexpr int step = pair_lo * 3 + decltype(sc_)::value;
                constexpr int dy = (step / 3) / 2, dx = step % 3, oy = r - dy, set = step & 1;
                if constexpr (step < 17) {
                    constexpr int nd = ((step + 1) / 3) / 2, nx = (step + 1) % 3;
                    load_w(std::integral_constant<int, set ^ 1>{}, nd * 3 + nx);
                }
#pragma unroll
                for (int ox = 0; ox < 4; ++ox)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int o = 0; o < COUT; ++o)
                            acc[oy][ox][o] = fmaf(t[ox + dx][j], wv[set][j][o], acc[oy][ox][o]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    // add the 16 channel quads: DPP butterfly inside each row of 16 lanes (quad swaps, half mirror, row
    // mirror: every lane ends with the full sum), then lanes 0..3 of a group store column ox = lane
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                float v = acc[a][b][o];
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
                acc[a][b][o] = v;
            }
    if (l16 < 4) {
        const int xx = x0 + l16;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int yy = y0 + a;
            if (xx < W && yy < H) {
                float *o = out.p + out.pix(n, yy, xx) * (size_t)out.C;
#pragma unroll
                for (int oc = 0; oc < COUT; ++oc) {
                    const float v = l16 == 0 ? acc[a][0][oc] : l16 == 1 ? acc[a][1][oc] : l16 == 2 ? acc[a][2][oc] : acc[a][3][oc];
                    o[oc] = v + bias[oc];
                }
            }
        }
    }
}

} // namespace

// -------------------------------------------------------------------------------------------------
// final_conv, split-f16 mode: the conv is linear, so  out[p][o] = sum_tap y[p + tap][tap][o]  with
// y[q][tap][o] = sum_c act[q][c] w[tap][c][o]  computed ONCE per input pixel — a [pixels] x [C] x [27 -> 32]
// GEMM on v_mfma_f32_16x16x32_f16 whose A fragments come straight from the activation: lane (l16, q)
// loads 8 consecutive channels of pixel l16 (32 B), applies the folded GroupNorm affine + Swish and
// the hi / lo split in registers (every activation is evaluated once; the VALU form above evaluates
// a 4 x 6 window per 2 x 4 outputs, 3x the tensor). Block = 8 x 32 output pixels: y of the 10 x 34
// halo region (22 M-tiles of 16 pixels) goes to LDS, then every thread adds its pixel's 9 taps.
// -------------------------------------------------------------------------------------------------
template <int KSTEPS>
__global__ __launch_bounds__(256) void final_conv_mfma_kernel(const TDesc x, const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              const h16x8 *__restrict__ wfr,   // [kstep][nt 2][hi|lo][lane 64]
                                                              const float w_unscale, const float *__restrict__ bias,
                                                              const TDesc out, int *ovf) {
    constexpr int HR = 10, HC = 34, NPIX = HR * HC, MTILES = (NPIX + 15) / 16, YLD = 33;
    unsigned range_bits = 0;        // running max of the hi halfs' exponent fields: 0x7C00 = beyond hi + lo, or not finite
    __shared__ float Y[MTILES * 16 * YLD];
    const int C = x.C, H = x.H, W = x.W, Cout = out.C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8, n = blockIdx.z;
    // weight fragments and this lane's GroupNorm scale / shift (channels 8q .. 8q + 7 of every 32-channel K-step)
    h16x8 bh[KSTEPS][2], bl[KSTEPS][2];
    float sc[KSTEPS][8], sh[KSTEPS][8];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bh[ks][nt] = wfr[((ks * 2 + nt) * 2 + 0) * 64 + lane];
            bl[ks][nt] = wfr[((ks * 2 + nt) * 2 + 1) * 64 + lane];
        }
        const float *sp = scale + (size_t)n * C + ks * 32 + q * 8, *hp = shift + (size_t)n * C + ks * 32 + q * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(sp), s1 = *reinterpret_cast<const f32x4 *>(sp + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(hp), h1 = *reinterpret_cast<const f32x4 *>(hp + 4);
        sc[ks][0] = s0.x; sc[ks][1] = s0.y; sc[ks][2] = s0.z; sc[ks][3] = s0.w; sc[ks][4] = s1.x; sc[ks][5] = s1.y; sc[ks][6] = s1.z; sc[ks][7] = s1.w;
        sh[ks][0] = h0.x; sh[ks][1] = h0.y; sh[ks][2] = h0.z; sh[ks][3] = h0.w; sh[ks][4] = h1.x; sh[ks][5] = h1.y; sh[ks][6] = h1.z; sh[ks][7] = h1.w;
    }
    for (int mt = wave; mt < MTILES; mt += 4) {
        const int p = min(mt * 16 + l16, NPIX - 1);              // halo-region pixel of this lane's A row
        const int hy = p / HC, hx = p - hy * HC;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;   // zero padding AFTER the activation
        const float *src = x.p + x.pix(n, min(max(yy, -1), H), min(max(xx, -1), W)) * (size_t)C + q * 8;
        f32x4 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + ks * 32), v1 = *reinterpret_cast<const f32x4 *>(src + ks * 32 + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            h16x8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = swish_fast(fmaf(f[e], sc[ks][e], sh[ks][e]));
                a = ok ? a : 0.f;
                ah[e] = (_Float16)a;
                al[e] = (_Float16)(a - (float)ah[e]);
                const unsigned eb = (unsigned)__builtin_bit_cast(unsigned short, ah[e]) & 0x7C00u;
                range_bits = eb > range_bits ? eb : range_bits;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks][nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks][nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks][nt], acc[nt], 0, 0, 0);
            }
        }
        // C/D map: col = l16 (+ 16 nt) = tap * 3 + o, row = 4 q + j = pixel of the M-tile
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) Y[(mt * 16 + 4 * q + j) * YLD + nt * 16 + l16] = acc[nt][j] * w_unscale;
    }
    if (ovf != nullptr && split_range_overflow(range_bits)) *ovf = 1;      // (the activations are split on the fly: same contract)
    __syncthreads();
    const int oy = threadIdx.x >> 5, ox = threadIdx.x & 31;
    const int yy = y0 + oy, xx = x0 + ox;
    if (yy < H && xx < W) {
        float o3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float *yp = Y + ((oy + dy) * HC + ox + dx) * YLD + (dy * 3 + dx) * 3;
                for (int o = 0; o < Cout; ++o) o3[o] += yp[o];
            }
        float *op = out.p + out.pix(n, yy, xx) * (size_t)Cout;
        for (int o = 0; o < Cout; ++o) op[o] = o3[o] + bias[o];
    }
}

bool final_conv_supported(int C, int Cout) { return (C % 64) == 0 && Cout >= 1 && Cout <= 4; }
// MFMA form (split-f16 mode): 27 (tap, output) columns in two 16-wide tiles
bool final_conv_mfma_supported(int C, int Cout) { return Cout == 3 && (C == 32 || C == 64 || C == 128); }
size_t final_conv_mfma_weight_floats(int C) { return (size_t)(C / 32) * 2 * 2 * 64 * 4; }

// OIHW [3][C][3][3] -> B fragments [kstep][nt][hi|lo][lane][8 halfs]: lane (l16, q) holds column nt*16 + l16 =
// tap*3 + o (columns >= 27 zero), k = channels kstep*32 + 8q .. + 7; scaled by 2^k (max|w| 2^k in [1024, 2048))
float pack_final_conv_mfma_weight(const float *oihw, int C, float *dst_as_float) {
    const int k = split_scale_exponent(oihw, (size_t)3 * C * 9);
    const float scl = ldexpf(1.0f, k);
    _Float16 *d = reinterpret_cast<_Float16 *>(dst_as_float);
    for (int ks = 0; ks < C / 32; ++ks)
        for (int nt = 0; nt < 2; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int l16 = lane & 15, q = lane >> 4, col = nt * 16 + l16;
                for (int e = 0; e < 8; ++e) {
                    float v = 0.f;
                    if (col < 27) {
                        const int tap = col / 3, o = col - tap * 3, c = ks * 32 + q * 8 + e;
                        v = oihw[((size_t)o * C + c) * 9 + tap] * scl;
                    }
                    const _Float16 hi = (_Float16)v;
                    d[((((size_t)ks * 2 + nt) * 2 + 0) * 64 + lane) * 8 + e] = hi;
                    d[((((size_t)ks * 2 + nt) * 2 + 1) * 64 + lane) * 8 + e] = (_Float16)(v - (float)hi);
                }
            }
    return ldexpf(1.0f, -k);
}

void launch_final_conv_mfma(const TDesc &x, int B, const float *scale, const float *shift, const float *wfr, float w_unscale,
                            const float *bias, const TDesc &out, hipStream_t s, int *ovf) {
    const dim3 grid((x.W + 31) / 32, (x.H + 7) / 8, B);
    const h16x8 *w = reinterpret_cast<const h16x8 *>(wfr);
    switch (x.C / 32) {
    case 1: hipLaunchKernelGGL(final_conv_mfma_kernel<1>, grid, dim3(256), 0, s, x, scale, shift, w, w_unscale, bias, out, ovf); break;
    case 2: hipLaunchKernelGGL(final_conv_mfma_kernel<2>, grid, dim3(256), 0, s, x, scale, shift, w, w_unscale, bias, out, ovf); break;
    default: hipLaunchKernelGGL(final_conv_mfma_kernel<4>, grid, dim3(256), 0, s, x, scale, shift, w, w_unscale, bias, out, ovf); break;
    }
}

// OIHW [Cout][C][3][3] -> [tap][C][4] (output channel padded to 4 floats)
void pack_final_conv_weight(const float *oihw, int Cout, int C, float *dst) {
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < C; ++c)
            for (int o = 0; o < 4; ++o)
                dst[((size_t)t * C + c) * 4 + o] = o < Cout ? oihw[((size_t)o * C + c) * 9 + t] : 0.f;
}

void launch_final_conv(const TDesc &x, int B, const float *scale, const float *shift, const float *wq, const float *bias,
                       const TDesc &out, hipStream_t s) {
    const size_t lds = (size_t)9 * x.C * 4 * sizeof(float);
    const dim3 grid((x.W + 31) / 32, (x.H + 3) / 4, B);
#define SR3_FC(N)                                                                                                  \
    {                                                                                                              \
        static size_t attr = 48 * 1024;                                                                            \
        if (lds > attr) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(final_conv_kernel<N>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
            attr = lds;                                                                                            \
        }                                                                                                          \
        hipLaunchKernelGGL(final_conv_kernel<N>, grid, dim3(256), lds, s, x, scale, shift, wq, bias, out);         \
    }
    switch (out.C) {
    case 1: SR3_FC(1) break;
    case 2: SR3_FC(2) break;
    case 3: SR3_FC(3) break;
    default: SR3_FC(4) break;
    }
#undef SR3_FC
}

// -------------------------------------------------------------------------------------------------
// downs.0 on the packed sampler state (split-f16 arithmetic only).
//   xp   [B][H+2][W+2] pixels of 16 halfs: channels 0..7 hi | channels 0..7 lo (zero border, channels
//        >= in_channel zero)
//   wci  [dy 3][nt][hi|lo][lane 64][8 halfs]: B fragments of v_mfma_f32_16x16x32_f16 — lane (l16, q)
//        holds output channel nt*16 + l16, k-chunk q = tap dx (q = 3: zeros), 8 halfs = input channels
// Block = 4 waves x 64 consecutive pixels (4 M-tiles of 16) = 256 pixels of one image; A fragments are
// two 16-B global loads per lane and (dy, M-tile): pixel x0 + l16 + q - 1 of row y + dy - 1.
// Epilogue as conv_epilogue16: bias, fp32 output and / or split-f16 twin, fused GroupNorm statistics
// (one slice per block), range check of the twin.
// -------------------------------------------------------------------------------------------------
namespace {

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_in_kernel(const _Float16 *__restrict__ xp, const h16x8 *__restrict__ wci,
                                                         const float *__restrict__ bias, const float w_unscale,
                                                         const TDesc out, const TDesc out_split, const int out_f32,
                                                         double *__restrict__ stats, const int stats_slices, int *ovf,
                                                         const int nblocks) {
    constexpr int MT = 4;
    __shared__ double2 red[16][NT * 16];
#ifndef SR3_CI_ROWSTORE
#define SR3_CI_ROWSTORE 1
#endif
#if SR3_CI_ROWSTORE
    constexpr int TLD = NT * 16 + 4;                     // floats per staged pixel row (+4: bank spread, 16-B aligned)
    __shared__ __attribute__((aligned(16))) float tstage[4 * 16 * TLD];
#endif
    const int H = out.H, W = out.W, Hp = H + 2, Wp = W + 2, HW = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int Cout = NT * 16;
    unsigned range_bits = 0;             // running max of the hi halfs' exponent fields (split_pair_word)
    // resident weight fragments
    h16x8 bh[3][NT], bl[3][NT];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[dy][nt] = wci[((dy * NT + nt) * 2 + 0) * 64 + lane];
            bl[dy][nt] = wci[((dy * NT + nt) * 2 + 1) * 64 + lane];
        }
    // persistent blocks: the weight fragments (24.6 KB per wave for 64 output channels) are fetched once per
    // wave, not once per 64 pixels
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int pix0 = blk * 256 + wave * 64;                 // first pixel (n*HW + y*W + x) of this wave
    const int n = pix0 / HW;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
    int ty[MT], tx[MT];                                      // tile origin (row, first column)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rem = pix0 + mt * 16 - n * HW;
        ty[mt] = rem / W;
        tx[mt] = rem - ty[mt] * W;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        h16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // padded coordinates of pixel (ty - 1 + dy, tx + l16 - 1 + q) are (ty + dy, tx + l16 + q)
            const _Float16 *pp = xp + (((size_t)n * Hp + ty[mt] + dy) * Wp + tx[mt] + l16 + q) * 16;
            ah[mt] = *reinterpret_cast<const h16x8 *>(pp);
            al[mt] = *reinterpret_cast<const h16x8 *>(pp + 8);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[dy][nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[dy][nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[dy][nt], acc[mt][nt], 0, 0, 0);
            }
    }
    // ---- epilogue: C/D map col = l16 (+ 16 nt), row = 4 q + j. Whole-tile phases, each behind ONE test
    // of its option (a per-element form costs two branches per value) ----
    unsigned rb[MT];                                         // element offset of (row 4q, column 0) per M-tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rb[mt] = (unsigned)out.pix(n, ty[mt], tx[mt] + 4 * q) * (unsigned)Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bs = bias ? bias[nt * 16 + l16] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][nt][j] = fmaf(acc[mt][nt][j], w_unscale, bs);
    }
    if (out_f32) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) out.p[rb[mt] + (unsigned)(j * Cout + nt * 16) + (unsigned)l16] = acc[mt][nt][j];
    }
    if (out_split.p != nullptr) {
#if SR3_CI_ROWSTORE
        // twin through an LDS transpose: the accumulator layout gives a lane 4 pixels x 1 channel (4-byte stores,
        // 32-B runs); staged as [pixel][channel] a lane owns 8 consecutive channels of a pixel and stores its hi
        // and lo halfs as two 16-B pieces of the 128-B chunk
        float *T = tstage + wave * (16 * TLD);
        constexpr int OCT = NT * 2;               // channel octets per pixel
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {         // one 16-pixel M-tile at a time (4.4 KB of LDS per wave)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) T[(4 * q + j) * TLD + nt * 16 + l16] = acc[mt][nt][j];
            // a wave only reads what it wrote itself: no block barrier, just its own LDS writes completed
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
#pragma unroll
            for (int k = 0; k < (16 * OCT + 63) / 64; ++k) {
                const int item = lane + 64 * k;
                if (item < 16 * OCT) {
                    const int px = item / OCT, oct = item - px * OCT;
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(T + px * TLD + oct * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(T + px * TLD + oct * 8 + 4);
                    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    h16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        hi[e] = (_Float16)f[e];
                        lo[e] = (_Float16)(f[e] - (float)hi[e]);
                        const unsigned eb = (unsigned)__builtin_bit_cast(unsigned short, hi[e]) & 0x7C00u;
                        range_bits = eb > range_bits ? eb : range_bits;
                    }
                    const size_t pixel = out.pix(n, ty[mt], tx[mt] + px);
                    _Float16 *dst = reinterpret_cast<_Float16 *>(out_split.p + pixel * Cout + (oct >> 2) * 32) + (oct & 3) * 8;
                    *reinterpret_cast<h16x8 *>(dst) = hi;
                    *reinterpret_cast<h16x8 *>(dst + 32) = lo;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // the reads are done before the next M-tile overwrites T
        }
#else
        // per 32-channel chunk 32 hi halfs | 32 lo halfs; lanes l16, l16 ^ 1 hold neighbouring channels: the
        // even lane stores both hi halfs, the odd lane both lo halfs (split_pair_word)
        unsigned *tw = reinterpret_cast<unsigned *>(out_split.p);
        const unsigned psel = split_pair_selector(l16 & 1);
        const unsigned lane_word = ((l16 & 1) ? 16u : 0u) + ((unsigned)l16 >> 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned word = split_pair_word(acc[mt][nt][j], psel, range_bits);
                    // o = rb + j*Cout + nt*16 + l16: chunk base (o & ~31), word (odd ? 16 : 0) + (o & 31) / 2
                    const unsigned ob = rb[mt] + (unsigned)(j * Cout + (nt >> 1) * 32);
                    tw[ob + (unsigned)((nt & 1) * 8) + lane_word] = word;
                }
#endif
    }
    if (stats != nullptr) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double st1 = 0.0, st2 = 0.0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double v = (double)acc[mt][nt][j];
                    st1 += v;
                    st2 = fma(v, v, st2);
                }
            red[wave * 4 + q][nt * 16 + l16] = make_double2(st1, st2);
        }
    }
    if (stats != nullptr) {
        __syncthreads();
        if (threadIdx.x < Cout) {
            double a = 0, b = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double2 v = red[i][threadIdx.x]; a += v.x; b += v.y; }
            const int slice = (blk * 256 - n * HW) / 256;
            double *o = stats + (((size_t)n * stats_slices + slice) * Cout + threadIdx.x) * 2;
            o[0] = a; o[1] = b;
        }
        __syncthreads();        // red[] is reused by the next block of pixels
    }
    }
    if (ovf != nullptr && split_range_overflow(range_bits)) *ovf = 1;
}

// fp32 state tensor (first 8 channels of [B][H+2][W+2][C]) -> packed split pixels (interior only)
__global__ void pack_state_kernel(const TDesc x, _Float16 *__restrict__ xp, int total, int *ovf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int HW = x.H * x.W;
    const int n = i / HW, rem = i - n * HW, y = rem / x.W, xx = rem - y * x.W;
    const size_t pix = x.pix(n, y, xx);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(x.p + pix * x.C), b = *reinterpret_cast<const f32x4 *>(x.p + pix * x.C + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    h16x8 hi, lo;
    float absmax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        absmax = fmaxf(absmax, fabsf(f[j]));
        hi[j] = (_Float16)f[j];
        lo[j] = (_Float16)(f[j] - (float)hi[j]);
    }
    *reinterpret_cast<h16x8 *>(xp + pix * 16) = hi;
    *reinterpret_cast<h16x8 *>(xp + pix * 16 + 8) = lo;
    if (ovf != nullptr && absmax > SPLIT_F16_MAX) *ovf = 1;
}

} // namespace

bool conv_in_supported(int Cin, int Cout, int H, int W) {
    // Cout a multiple of 32: the split-f16 twin of the output is stored in 32-channel chunks
    return Cin <= 8 && (Cout == 32 || Cout == 64) && (W % 16) == 0 && ((H * W) % 256) == 0;
}

// OIHW [Cout][Cin][3][3] -> fragment layout, scaled by 2^k (max|w| 2^k in [1024, 2048)); returns 2^-k
float pack_conv_in_weight(const float *oihw, int Cout, int Cin, float *dst_as_float) {
    const int NT = Cout / 16;
    const int k = split_scale_exponent(oihw, (size_t)Cout * Cin * 9);
    const float sc = ldexpf(1.0f, k);
    _Float16 *d = reinterpret_cast<_Float16 *>(dst_as_float);
    for (int dy = 0; dy < 3; ++dy)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int l16 = lane & 15, q = lane >> 4, co = nt * 16 + l16;
                for (int j = 0; j < 8; ++j) {
                    float v = 0.f;
                    if (q < 3 && j < Cin) v = oihw[(((size_t)co * Cin + j) * 3 + dy) * 3 + q] * sc;
                    const _Float16 hi = (_Float16)v;
                    d[((((size_t)dy * NT + nt) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    d[((((size_t)dy * NT + nt) * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                }
            }
    return ldexpf(1.0f, -k);
}

size_t conv_in_weight_floats(int Cout) { return (size_t)3 * (Cout / 16) * 2 * 64 * 4; }   // 8 halfs = 4 floats per lane

void launch_pack_state(const TDesc &x, int B, float *xp, hipStream_t s, int *ovf) {
    const int total = B * x.H * x.W;
    hipLaunchKernelGGL(pack_state_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, reinterpret_cast<_Float16 *>(xp), total, ovf);
}

void launch_conv_in(const float *xp, const float *wci, const float *bias, float w_unscale, int B, const TDesc &out,
                    const TDesc &out_split, int out_f32, double *stats, int stats_slices, int *ovf, hipStream_t s) {
    const int blocks = B * out.H * out.W / 256;
    const _Float16 *x = reinterpret_cast<const _Float16 *>(xp);
    const h16x8 *w = reinterpret_cast<const h16x8 *>(wci);
    const int grid = blocks < 512 ? blocks : 512;          // 2 resident blocks of 4 waves per CU (230 registers at 64 output channels)
#define SR3_CI(N) hipLaunchKernelGGL(conv_in_kernel<N>, dim3(grid), dim3(256), 0, s, x, w, bias, w_unscale, out, out_split, out_f32, stats, stats_slices, ovf, blocks)
    switch (out.C / 16) {
    case 1: SR3_CI(1); break;
    case 2: SR3_CI(2); break;
    case 3: SR3_CI(3); break;
    default: SR3_CI(4); break;
    }
#undef SR3_CI
}

} // namespace sr3
