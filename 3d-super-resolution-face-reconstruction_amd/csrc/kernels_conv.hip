// Conv2d (1x1 / 3x3, stride 1|2, optional nearest-x2 upsample, optional channel concat) as an
// implicit GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the SR3 UNet (reference model/sr/sr3_modules/unet.py):
//   Block conv3x3 (input already GroupNorm+Swish'ed by gn_apply)        :80-91
//   ResnetBlock  + FeatureWiseAffine bias, + residual (fused epilogue)   :94-110
//   Upsample / Downsample (index remap in the gather)                    :58-74
//   torch.cat((x, skip), 1) (dual-pointer K range)                       :261
//   SelfAttention.qkv / .out 1x1 convs                                   :120-121
//
// GEMM view: M = B*Hout*Wout output pixels, N = Cout, K = ks*ks*Cin; one K-step = one
// (tap, 32-channel chunk). Weights are pre-packed [tap][Cout][Cin], so both operands are
// K-contiguous 128-byte rows.
//
// Measured fact that shapes this kernel: on gfx950 the f32-input MFMA executes at the f32 vector
// rate and does NOT overlap with VALU work on the same SIMD (every extra v_fma in a co-resident
// wave costs ~4.8 cycles of matrix time; profiles/r01_notes.md). So the kernel is built to issue
// almost no vector instructions besides the MFMAs:
//   * activations are stored with a 1-pixel zero border (TDesc), so the im2col gather needs no
//     bounds checks, and GroupNorm+Swish is applied by a separate HBM-bound pass;
//   * tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//     ds_write, no transform;
//   * 512-thread blocks are wave-specialised: 4 consumer waves (ds_read_b128 + MFMA only) and
//     4 producer waves (address arithmetic + DMA). The hardware deals a workgroup's waves
//     round-robin over the 4 SIMDs, so every SIMD hosts one consumer and one producer per block;
//     2 blocks are resident per CU (2 x 64 KiB LDS).
// LDS image: rows of 32 floats (128 B) without padding (the DMA writes 1 KiB lane-linear), 16-B
// chunk c of row r stored at position c ^ ((r >> 1) & 7): the swizzle is applied on the per-lane
// SOURCE address and again on the fragment read (conflict-free ds_read_b128).
// Each lane reads 4 consecutive k per ds_read_b128 and feeds 4 MFMAs; lane half h supplies
// k = 8kk + 4h + j to MFMA j of group kk for both operands.
#include "sr3_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <algorithm>

namespace sr3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector (not HIP's float4 struct: struct copies through a register array become
// memcpys via scratch memory)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace {

// timing experiments (tools/conv_bench.py) exist only in builds with -DSR3_EXPERIMENTS: the product
// library ignores SR3_CONV_DBG and carries no experiment branches in its kernels
#ifdef SR3_EXPERIMENTS
#define SR3_DBG(p) ((p).dbg)
#else
#define SR3_DBG(p) 0
#endif

constexpr int BK = 32;            // channels per K-step
constexpr int ROWF = 32;          // floats per LDS row (128 B, unpadded)

// compile-time loop: every index is a constant in the front end, so register arrays are split
// into scalars before any loop pass (runtime-indexed arrays end up in scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// producer side of a pipeline barrier: wait until at most N of this wave's LDS-DMA instructions
// are still in flight, then the workgroup barrier. A raw s_barrier is used because __syncthreads()
// would drain every outstanding DMA (vmcnt(0)).
template <int N>
__device__ __forceinline__ void producer_sync() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void dma16(const float *g, float *lds_wave_base) {
    // 64 lanes x 16 B -> 1 KiB at lds_wave_base (wave-uniform) + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The same from a wave-uniform base + a per-lane 32-bit byte offset, with NO vector arithmetic in front of the DMA:
// written as base + zext(offset) the optimiser reassociates the sum, hoists tap-invariant 64-bit parts into VGPR
// pairs (18 pairs in the unrolled 3x3 loop) and pays a 64-bit vector add per DMA — vector instructions of the
// producer waves compete with the consumers' MFMAs for issue (same box, B = 64 step: 17.45 -> 17.07 ms in f16x3,
// 44.4 -> 40.8 ms in f32 mode; profiles/README.md finding 48).
//  BUF: buffer_load ... lds — a resource descriptor over the base (scalar registers only), the offset goes into
//       the instruction as it is. Best where the MFMAs are the critical path (the 128-row tiles).
//  !BUF: plain global_load_lds on base + offset, addresses left to the optimiser. Best where the producers' own
//       issue rate is the critical path (the 64x64 tile of the small batches): the descriptor set-up of the buffer
//       form, or pinning global_load_lds to its (SGPR base) + (VGPR offset) form with opaque operands (one v_mov per
//       DMA), cost config 1 (8 -> 16, B = 4) 2-3 %.
template <bool BUF>
__device__ __forceinline__ void dma16s(const char *sbase, unsigned voff, float *lds_wave_base) {
    if constexpr (BUF) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(sbase), 0, -1, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, (int)voff, 0, 0, 0);
    } else {
        dma16(reinterpret_cast<const float *>(sbase + voff), lds_wave_base);
    }
}

// KS: 1 | 2 | 3 (the tap loop is unrolled); nearest x2 upsampling never reaches the kernel (launch_conv_up2)
// m / (Hout*Wout) and rem / Wout of the tile address set-up: shifts when the sizes are powers of two
__device__ __forceinline__ int div_hw(const ConvParams &p, int m, int HWo) { return p.hw_shift >= 0 ? (m >> p.hw_shift) : m / HWo; }
__device__ __forceinline__ int div_w(const ConvParams &p, int rem, int W) { return p.w_shift >= 0 ? (rem >> p.w_shift) : rem / W; }

// kernel-side view of the parameters for this block's sub-pixel phase (ConvParams::phases)
__device__ __forceinline__ ConvParams phase_params(const ConvParams &in, int ph) {
    ConvParams p = in;
    if (in.phases > 1) {
        p.org_y = p.out_oy = ph >> 1;
        p.org_x = p.out_ox = ph & 1;
        p.w = in.w + (size_t)ph * in.phase_w_stride;
        p.stats_slice0 = in.stats_slice0 + ph * in.phase_slices;
        if (in.part) p.part = in.part + (size_t)ph * in.phase_part_stride;
    }
    return p;
}

// residual element o (= pixel * C + channel) of a tensor stored in the split-f16 format: hi + lo
__device__ __forceinline__ float load_split(const float *base, unsigned o) {
    const _Float16 *hp = reinterpret_cast<const _Float16 *>(base + (o & ~31u)) + (o & 31u);
    return (float)hp[0] + (float)hp[32];
}

// Consumer epilogue shared by the conv kernels: split-K partials, or accumulator + bias +
// FeatureWiseAffine channel bias + residual into the zero-bordered output, plus the fused
// GroupNorm statistics (per-column fp64 sums left in LDS for the producer threads).
// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int BM, int BN, int WGM, int WGN, int MI, int NI>
__device__ __forceinline__ void conv_epilogue_impl(const ConvParams &p, f32x16 (&acc)[MI][NI], float *smem,
                                                   const int *rowpix, const int *rowimg, int m0, int n0, int M,
                                                   int wm, int wn, int li, int lh, int split, int nsplit) {
    constexpr int WM = BM / WGM, WN = BN / WGN;
    const int Cout = p.out.C;
    if (nsplit > 1) {
        // split-K: raw partial sums to part[split][m][n]; the reduce kernel finishes the epilogue
        float *pp = p.part + (size_t)split * M * Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * WN + ni * 32 + li;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < M && n < Cout) pp[(size_t)m * Cout + n] = acc[mi][ni][r];
                }
        }
        return;
    }
    // processed 4 accumulator registers (4 consecutive rows) at a time to keep registers low
    unsigned split_range = 0;          // running max of the stored hi halfs' exponent fields (range check; catches NaN too)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * WN + ni * 32 + li;
        const int nc = min(n, Cout - 1);
        const float bs = p.bias ? p.bias[nc] : 0.f;
        double st1 = 0.0, st2 = 0.0;       // fused GroupNorm statistics of this lane's column
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int rbase = wm * WM + mi * 32 + 8 * rq + 4 * lh;
                float add[4];
                unsigned o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    add[j] = bs;
                    o[j] = (unsigned)rowpix[rbase + j] * (unsigned)Cout + (unsigned)nc;
                }
                if (p.resid.p != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) add[j] += p.resid_split ? load_split(p.resid.p, o[j]) : p.resid.p[o[j]];
                }
                if (p.chan_bias != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        add[j] += p.chan_bias[(size_t)rowimg[rbase + j] * p.chan_bias_stride + nc];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + rbase + j;
                    const float v = acc[mi][ni][4 * rq + j] + add[j];
                    if (m < M && n < Cout && p.out_f32) p.out.p[o[j]] = v;
                    if (p.out_split.p != nullptr) {
                        // twin in the conv input format: per 32-channel chunk 32 hi halfs | 32 lo halfs.
                        // Lanes li, li^1 hold neighbouring channels: the even lane stores both hi halfs,
                        // the odd lane both lo halfs (one 4-byte store per lane instead of two 2-byte ones).
                        // no clamp: a value beyond the fp16 range is DETECTED (split_absmax -> ConvParams::ovf,
                        // the API call then fails) instead of being silently saturated
                        const float g = v;
                        const _Float16 hi = (_Float16)g;
                        {
                            const unsigned eb = (unsigned)__builtin_bit_cast(unsigned short, hi) & 0x7C00u;
                            split_range = eb > split_range ? eb : split_range;
                        }
                        const _Float16 lo = (_Float16)(g - (float)hi);
                        const unsigned own = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                             ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                        const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
                        const bool odd = li & 1;
                        const unsigned word = odd ? ((oth >> 16) | (own & 0xFFFF0000u)) : ((own & 0xFFFFu) | (oth << 16));
                        // Cout % 32 == 0 for every tensor that has a twin, so n < Cout holds for both lanes
                        if (m < M && n < Cout) {
                            unsigned *hd = reinterpret_cast<unsigned *>(p.out_split.p + (o[j] & ~31u));
                            hd[(odd ? 16 : 0) + ((o[j] & 31u) >> 1)] = word;
                        }
                    }
                    if (p.stats != nullptr) { st1 += (double)v; st2 = fma((double)v, (double)v, st2); }
                }
            }
        }
        if (p.stats != nullptr)
            reinterpret_cast<double2 *>(smem)[(wm * 2 + lh) * BN + wn * WN + ni * 32 + li] = make_double2(st1, st2);
    }
    if (p.ovf != nullptr && split_range_overflow(split_range)) *p.ovf = 1;
    if (p.stats != nullptr) __syncthreads();
}

template <int BM, int BN, int WGM, int WGN, int MI, int NI>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f32x16 (&acc)[MI][NI], float *smem,
                                              const int *rowpix, const int *rowimg, int m0, int n0, int M, int wm,
                                              int wn, int li, int lh, int split, int nsplit) {
    conv_epilogue_impl<BM, BN, WGM, WGN, MI, NI>(p, acc, smem, rowpix, rowimg, m0, n0, M, wm, wn, li, lh, split, nsplit);
}

// Epilogue for 16x16 accumulator tiles (v_mfma_f32_16x16x32_f16): C/D map col = lane & 15,
// row = 4 * (lane >> 4) + r. Same duties as conv_epilogue (no split-K: the halo kernels never split;
// they also guarantee M % BM == 0 and Cout % BN == 0, so nothing is masked).
// Written as a sequence of whole-tile phases, each behind ONE test of its (uniform) option — values
// stay in the accumulator registers in between — instead of testing every option per element: the
// per-element form compiled to ~140 branches with a wait for its loads in every basic block and cost a
// 128x64 tile as much as a third of its K loop.
// colbias[BN] (LDS, staged by the producers at kernel start so that no global load sits on the
// epilogue's critical path): bias + FeatureWiseAffine bias when the whole tile lies in one image
// (one_img: the flag the producers leave behind the table); otherwise bias only and the per-image part is
// gathered per row. BN may be a column slice of the block's tile (in-place split-K: n0 and colbias advanced by it).
// QRED (persistent kernel): the four row groups of a wave are added by lane exchange first, only lanes q == 0
// write [wm][column] entries (a quarter of the staging space) and the caller does the barrier
// GNF (producer-side GroupNorm of the output, ConvParams::gnf_*): the column statistics go out FIRST (phase 6 moves in
// front of the stores), the wave then sits through the four barriers of gnf_producer_tail, picks up scale / shift of its
// columns from LDS, applies swish(scale * v + shift) to the accumulators and stores them as the split-f16 tensor only.
template <int BM, int BN, int WGM, int WGN, int MT, int NT, bool QRED = false, bool GNF = false>
__device__ __forceinline__ void conv_epilogue16(const ConvParams &p, f32x4 (&acc)[MT][NT], float *smem,
                                                const int *rowpix, const int *rowimg, int m0, int n0, int M, int wm,
                                                int wn, int l16, int q, const float *colbias, const bool one_img) {
    constexpr int WM = BM / WGM, WN = BN / WGN;
    static_assert(!(GNF && QRED), "the persistent experiment has no producer-side GroupNorm");
    const unsigned Cout = (unsigned)p.out.C;
    const unsigned ncol = (unsigned)(n0 + wn * WN + l16);       // column of nt = 0; + 16 per nt
    // element offset of (row, column 0) for this lane's 4 rows of every row tile
    unsigned rb[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int4 rp = *reinterpret_cast<const int4 *>(rowpix + wm * WM + mt * 16 + 4 * q);
        rb[mt][0] = (unsigned)rp.x * Cout; rb[mt][1] = (unsigned)rp.y * Cout;
        rb[mt][2] = (unsigned)rp.z * Cout; rb[mt][3] = (unsigned)rp.w * Cout;
    }
    // 1. column bias
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bs = colbias[wn * WN + nt * 16 + l16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mt][nt][j] += bs;
    }
    __builtin_amdgcn_sched_barrier(0);
    // 2. FeatureWiseAffine bias per row (tiles that span two images: the 8x8 level)
    if (p.chan_bias != nullptr && !one_img) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int4 ri = *reinterpret_cast<const int4 *>(rowimg + wm * WM + mt * 16 + 4 * q);
            const int im[4] = {ri.x, ri.y, ri.z, ri.w};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mt][nt][j] += p.chan_bias[(size_t)im[j] * p.chan_bias_stride + ncol + nt * 16];
                __builtin_amdgcn_sched_barrier(0);      // 4 loads in flight at a time: bounds the registers
            }
        }
    }
    // 3. residual (fp32 tensor, or hi + lo of a split-only tensor)
    if (p.resid.p != nullptr) {
        if (p.resid_split) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][nt][j] += load_split(p.resid.p, rb[mt][j] + ncol + nt * 16);
                __builtin_amdgcn_sched_barrier(0);      // one column of tiles in flight at a time
            }
        } else {
            const char *rbase = reinterpret_cast<const char *>(p.resid.p);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mt][nt][j] += *reinterpret_cast<const float *>(rbase + (rb[mt][j] + ncol + nt * 16) * 4u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (GNF) {
        // 6'. statistics of the values block2's GroupNorm sees (bias and FeatureWiseAffine bias included), then the
        //     hand-off (gnf_producer_tail: barriers A, A2, A3, Y, X1) and the normalisation in registers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double st1 = 0.0, st2 = 0.0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double v = (double)acc[mt][nt][j];
                    st1 += v; st2 = fma(v, v, st2);
                }
            reinterpret_cast<double2 *>(smem)[(wm * 4 + q) * BN + wn * WN + nt * 16 + l16] = make_double2(st1, st2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();        // A: column sums staged
        __syncthreads();        // A2: this block's slice is on its way to memory
        __syncthreads();        // A3: arrival counted
        __syncthreads();        // Y: (last block: slices folded | others: group ready)
        __syncthreads();        // X1: scale / shift of this N-tile in LDS
        const float2 *ab_lds = reinterpret_cast<const float2 *>(smem + WGM * 4 * BN * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float2 ab = ab_lds[wn * WN + nt * 16 + l16];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = fmaf(ab.x, acc[mt][nt][j], ab.y);
                    acc[mt][nt][j] = x * __frcp_rn(1.0f + __expf(-x));      // Swish (unet.py:53-55), as gn_apply does it
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The statistics phase (6) runs in FRONT of the stores (4, 5): the producer threads' tail (adding the staged column sums,
    // writing the slice) then runs under the consumers' store phase instead of after it, and the block leaves its CU
    // slot that much earlier: 16.887 -> 16.780 ms per B = 64 step (three alternating pairs, -DSR3_EARLY_STATS=0 | 1
    // builds of the same source; profiles/README.md finding 61)
#ifndef SR3_EARLY_STATS
#define SR3_EARLY_STATS 1
#endif
    constexpr bool EARLY = SR3_EARLY_STATS && !QRED && !GNF;
    auto stats_phase = [&]() {
    // 6. fused GroupNorm statistics of the stored values: fp64 column sums, handed to the producers
        if (!GNF && p.stats != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double st1 = 0.0, st2 = 0.0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double v = (double)acc[mt][nt][j];
                        st1 += v; st2 = fma(v, v, st2);
                    }
                if (QRED) {
                    st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
                    st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
                    if (q == 0) reinterpret_cast<double2 *>(smem)[wm * BN + wn * WN + nt * 16 + l16] = make_double2(st1, st2);
                } else {
                    reinterpret_cast<double2 *>(smem)[(wm * 4 + q) * BN + wn * WN + nt * 16 + l16] = make_double2(st1, st2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!QRED) __syncthreads();
        }
    };
    if constexpr (EARLY) stats_phase();
    // 4. fp32 output (32-bit byte offsets: every tensor is < 4 GiB)
    if (!GNF && p.out_f32) {
        char *obase = reinterpret_cast<char *>(p.out.p);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float *>(obase + (rb[mt][j] + ncol + nt * 16) * 4u) = acc[mt][nt][j];
        __builtin_amdgcn_sched_barrier(0);
    }
    // 5. split-f16 twin: per 32-channel chunk 32 hi halfs | 32 lo halfs. Lanes l16, l16 ^ 1 hold
    //    neighbouring channels: the even lane stores both hi halfs, the odd lane both lo halfs
    //    (split_pair_word: DPP exchange + one v_perm_b32; range check instead of a silent clamp).
    //    Element o = rb + 16 (cb16 + nt) + l16 with cb16 = (n0 + wn * WN) / 16: chunk base (o & ~31) floats,
    //    word (odd ? 16 : 0) + (o & 31) / 2 inside the chunk.
    if (p.out_split.p != nullptr) {
        const unsigned psel = split_pair_selector(l16 & 1);
        char *tlane = reinterpret_cast<char *>(p.out_split.p) + (((l16 & 1) ? 16u : 0u) + ((unsigned)l16 >> 1)) * 4u;
        const unsigned cb16 = (unsigned)(n0 + wn * WN) >> 4;
        unsigned range_bits = 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned t16 = cb16 + (unsigned)nt;                       // wave-uniform
            const unsigned coff = ((t16 >> 1) * 32u + (t16 & 1u) * 8u) * 4u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned word = split_pair_word(acc[mt][nt][j], psel, range_bits);
                    *reinterpret_cast<unsigned *>(tlane + rb[mt][j] * 4u + coff) = word;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (p.ovf != nullptr && split_range_overflow(range_bits)) *p.ovf = 1;
    }
    if constexpr (!EARLY) stats_phase();
}

// conv_epilogue16's phase structure for 32x32 accumulator tiles (the F8C consumers of the x-halo kernel): C/D map
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5). Whole tiles (M % BM == 0, Cout % BN == 0), no
// split-K. colbias / one_img as in conv_epilogue16; the statistics go to the producers FIRST (finding 61), staged as
// [wm * 2 + lh][column] entries (producer_stats_tail<BM, BN, WGM, 2>). The per-element form (conv_epilogue_impl) cost
// these kernels ~17 % of their time inside the step (finding 64).
template <int BM, int BN, int WGM, int WGN, int MI, int NI>
__device__ __forceinline__ void conv_epilogue32(const ConvParams &p, f32x16 (&acc)[MI][NI], float *smem,
                                                const int *rowpix, const int *rowimg, int wm, int wn, int li, int lh,
                                                int n0, const float *colbias, const bool one_img) {
    constexpr int WM = BM / WGM, WN = BN / WGN;
    const unsigned Cout = (unsigned)p.out.C;
    const unsigned ncol = (unsigned)(n0 + wn * WN + li);        // column of ni = 0; + 32 per ni
    // rows of accumulator registers 4 rq .. 4 rq + 3 of row tile mi: wm * WM + mi * 32 + 8 rq + 4 lh + j
    auto rows4 = [&](const int *tab, int mi, int rq) {
        return *reinterpret_cast<const int4 *>(tab + wm * WM + mi * 32 + 8 * rq + 4 * lh);
    };
    // 1. column bias
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const float bs = colbias[wn * WN + ni * 32 + li];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bs;
    }
    __builtin_amdgcn_sched_barrier(0);
    // 2. FeatureWiseAffine bias per row (tiles that span several images)
    if (p.chan_bias != nullptr && !one_img) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int4 ri = rows4(rowimg, mi, rq);
                const int im[4] = {ri.x, ri.y, ri.z, ri.w};
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mi][ni][4 * rq + j] += p.chan_bias[(size_t)im[j] * p.chan_bias_stride + ncol + ni * 32];
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    // 3. residual (fp32 tensor, or hi + lo of a split-only tensor)
    if (p.resid.p != nullptr) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int4 rp = rows4(rowpix, mi, rq);
                const unsigned rb[4] = {(unsigned)rp.x * Cout, (unsigned)rp.y * Cout, (unsigned)rp.z * Cout, (unsigned)rp.w * Cout};
                if (p.resid_split) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[mi][ni][4 * rq + j] += load_split(p.resid.p, rb[j] + ncol + ni * 32);
                } else {
                    const char *rbase = reinterpret_cast<const char *>(p.resid.p);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mi][ni][4 * rq + j] += *reinterpret_cast<const float *>(rbase + (rb[j] + ncol + ni * 32) * 4u);
                }
                __builtin_amdgcn_sched_barrier(0);      // 4 rows x NI columns in flight at a time
            }
    }
    // 6. fused GroupNorm statistics of the stored values, handed to the producers before the stores
    if (p.stats != nullptr) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            double st1 = 0.0, st2 = 0.0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = (double)acc[mi][ni][r];
                    st1 += v; st2 = fma(v, v, st2);
                }
            reinterpret_cast<double2 *>(smem)[(wm * 2 + lh) * BN + wn * WN + ni * 32 + li] = make_double2(st1, st2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // 4. fp32 output, 5. split-f16 twin (lanes li, li ^ 1 hold neighbouring channels: one 4-byte word per lane)
    const unsigned psel = split_pair_selector(li & 1);
    char *obase = reinterpret_cast<char *>(p.out.p);
    char *tlane = reinterpret_cast<char *>(p.out_split.p) + (((li & 1) ? 16u : 0u) + ((unsigned)li >> 1)) * 4u;
    const unsigned cb = (unsigned)(n0 + wn * WN) * 4u;          // byte offset of the wave's first 32-channel chunk
    unsigned range_bits = 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int4 rp = rows4(rowpix, mi, rq);
            const unsigned rb[4] = {(unsigned)rp.x * Cout, (unsigned)rp.y * Cout, (unsigned)rp.z * Cout, (unsigned)rp.w * Cout};
            if (p.out_f32) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float *>(obase + (rb[j] + ncol + ni * 32) * 4u) = acc[mi][ni][4 * rq + j];
            }
            if (p.out_split.p != nullptr) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned word = split_pair_word(acc[mi][ni][4 * rq + j], psel, range_bits);
                        *reinterpret_cast<unsigned *>(tlane + rb[j] * 4u + cb + ni * 128u) = word;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    if (p.out_split.p != nullptr && p.ovf != nullptr && split_range_overflow(range_bits)) *p.ovf = 1;
}

// Producer side of the fused statistics: after the consumers' sums are in LDS, the producer
// threads add them per column and write this tile's per-channel partials.
template <int BM, int BN, int WGM, int PER_WAVE = 2>
__device__ __forceinline__ void producer_stats_tail(const ConvParams &p, const float *smem, int m0, int n0, int HWo) {
    __syncthreads();
    const double2 *red = reinterpret_cast<const double2 *>(smem);
    const int col = threadIdx.x - 256;
    const int Cout = p.out.C;
    if (col < BN && n0 + col < Cout) {
        double a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < WGM * PER_WAVE; ++j) { const double2 v = red[j * BN + col]; a += v.x; b += v.y; }
        const int n = m0 / HWo, slice = p.stats_slice0 + (m0 - n * HWo) / BM;
        double *o = p.stats + (((size_t)n * p.stats_slices + slice) * Cout + n0 + col) * 2;
        o[0] = a; o[1] = b;
    }
}

// Producer side of the producer-side GroupNorm (ConvParams::gnf_*), run by the block's 256 producer threads while the
// consumer waves sit in the matching barriers of conv_epilogue16<..., GNF>. NSL = staged entries per column.
// Hand-off forms (MI355X_MICROARCH.md, inter-workgroup visibility): every byte that crosses blocks is written by a
// write-through (sc1) store and read by a cache-bypassing (sc1) global load; every storing wave waits vmcnt(0), then a
// workgroup barrier, then ONE lane adds to the group's counter (agent scope). The counter of a group of G blocks runs
// 0 -> G (arrivals; the add that returns G - 1 marks the last block) -> 2G (the last block's "ready", after its
// scale / shift stores have completed) -> 3G (departures: every block adds 1 after it has fetched scale / shift; the
// add that returns 3G - 1 stores 0, ready for the next launch — nobody polls any more by then).
// No wait can hang by construction: a block only waits for blocks of its own group, which launch_conv orders so that
// they are dispatched together (see conv_gnf_supported); should the hardware ever dispatch differently, the poll gives
// up after SR3_WAIT_TICKS of the 100 MHz real-time counter (5 ms), raises SR3_FLAG_GNF_TIMEOUT in *ovf and the grid drains.
typedef unsigned __attribute__((address_space(1))) *gnf_cnt_ptr;
typedef const double __attribute__((address_space(1))) *gnf_cdbl_ptr;
typedef double __attribute__((address_space(1))) *gnf_dbl_ptr;
typedef const float __attribute__((address_space(1))) *gnf_cflt_ptr;
typedef float __attribute__((address_space(1))) *gnf_flt_ptr;
template <int BM, int BN, int NSL>
__device__ __forceinline__ void gnf_producer_tail(const ConvParams &p, float *smem, int m0, int n0, int HWo, int tid) {
    const double2 *red = reinterpret_cast<const double2 *>(smem);
    float2 *ab_lds = reinterpret_cast<float2 *>(smem + NSL * BN * 4);                  // [BN]
    double2 *fin = reinterpret_cast<double2 *>(smem + NSL * BN * 4 + 2 * BN);          // [256 / BN][BN]
    const int Cout = p.out.C;
    const int img = m0 / HWo, TMI = HWo / BM, slice = (m0 - img * HWo) / BM;
    const int tilesN = Cout / BN, ntile = n0 / BN;
    const unsigned G = (unsigned)TMI;
    // counter and ready word of the group on cache lines of their own (the pollers' loads must not queue in front of
    // the other blocks' arrival adds)
    gnf_cnt_ptr cnt = (gnf_cnt_ptr)(p.gnf_cnt + ((size_t)img * tilesN + ntile) * 64);
    gnf_cnt_ptr rdy = cnt + 32;
    __syncthreads();                                    // A: the consumers' column sums are staged
    if (tid < BN) {
        double a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < NSL; ++j) { const double2 v = red[j * BN + tid]; a += v.x; b += v.y; }
        gnf_dbl_ptr o = (gnf_dbl_ptr)(p.stats + (((size_t)img * TMI + slice) * Cout + n0 + tid) * 2);
        __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the slice has reached the coherence point ...
    __syncthreads();                                    // A2: ... in every storing wave
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1u) {
            // last arrival: every block of the group counted itself after its slice had completed
            __hip_atomic_store(rdy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(rdy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(16);
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > SR3_WAIT_TICKS) {     // 5 ms: never in a healthy run
                    if (p.ovf != nullptr) atomicOr(p.ovf, SR3_FLAG_GNF_TIMEOUT);
                    break;
                }
            }
        }
    }
    __syncthreads();                                    // A3: all G slices of this image are complete
    // EVERY block folds the G slices of its N-tile's channels itself (in slice order: the same bits in every block):
    // thread (channel c, part) adds slices part, part + PARTS, ... with all its loads in flight at once. One hop less
    // than "the last block folds and publishes" (no second flag, no scale / shift round trip).
    constexpr int PARTS = 256 / BN;
    {
        const int c = tid % BN, part = tid / BN;
        gnf_cdbl_ptr base = (gnf_cdbl_ptr)(p.stats + ((size_t)img * TMI * Cout + n0 + c) * 2);
        const size_t sstride = (size_t)Cout * 2;
        double sa = 0, sb = 0;
        int sidx = part;
        for (; sidx + 15 * PARTS < TMI; sidx += 16 * PARTS) {
            double va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                va[u] = __hip_atomic_load(base + (size_t)(sidx + u * PARTS) * sstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vb[u] = __hip_atomic_load(base + (size_t)(sidx + u * PARTS) * sstride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { sa += va[u]; sb += vb[u]; }
        }
        for (; sidx + 3 * PARTS < TMI; sidx += 4 * PARTS) {
            double va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                va[u] = __hip_atomic_load(base + (size_t)(sidx + u * PARTS) * sstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vb[u] = __hip_atomic_load(base + (size_t)(sidx + u * PARTS) * sstride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa += va[u]; sb += vb[u]; }
        }
        for (; sidx < TMI; sidx += PARTS) {
            sa += __hip_atomic_load(base + (size_t)sidx * sstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sb += __hip_atomic_load(base + (size_t)sidx * sstride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        fin[part * BN + c] = make_double2(sa, sb);
    }
    __syncthreads();                                    // Y
    if (tid < BN) {
        // whole groups lie inside an N-tile: channel tid adds its group's channels and parts in a fixed order
        const int cg = Cout / p.gnf_groups, g0 = (tid / cg) * cg;
        double sa = 0, sb = 0;
        for (int cc = 0; cc < cg; ++cc)
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) { const double2 v = fin[pt * BN + g0 + cc]; sa += v.x; sb += v.y; }
        const double count = (double)cg * HWo;
        const double mean = sa / count;
        const double var = fmax(sb / count - mean * mean, 0.0);
        const float rstd = 1.0f / sqrtf((float)var + p.gnf_eps);
        const float sc = rstd * p.gnf_gamma[n0 + tid];
        ab_lds[tid] = make_float2(sc, p.gnf_beta[n0 + tid] - (float)mean * sc);
    }
    __syncthreads();                                    // X1: scale / shift of this N-tile in LDS
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // departed
        if (old == 2u * G - 1u) {       // every block of the group has read the slices: ready for the next launch
            __hip_atomic_store(rdy, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// PREC 0: exact f32 (v_mfma_f32_32x32x2_f32); PREC 1: split-f16, 3 x v_mfma_f32_32x32x16_f16
// NS: LDS pipeline stages (power of two or 3); the DMA of tile k+NS-1 is issued while tile k is
// multiplied, so NS-2 tiles stay in flight across a barrier (counted vmcnt + raw s_barrier)
template <int BM, int BN, int WGM, int WGN, int KS, int PREC, int NS>
__global__ __launch_bounds__(512, (((BM + BN) * ROWF * 4 * NS + 8 * BM) * 3 <= 160 * 1024 ? 6 : 4)) void conv_igemm_dma_f32(const ConvParams p_in) {
    const ConvParams p = phase_params(p_in, blockIdx.z);
    static_assert(WGM * WGN == 4, "4 consumer waves per block");
    constexpr bool DMA_BUF = !(BM == 64 && BN == 64);   // (see dma16s)
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;     // DMA instructions per producer wave and K-step
    constexpr int STAGE = (BM + BN) * ROWF;       // floats per pipeline stage
    static_assert(MI >= 1 && NI >= 1 && (WM % 16) == 0, "wave tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [2 stages][A: BM rows | B: BN rows][32 floats], then per-row tables for the epilogue
    int *rowpix = reinterpret_cast<int *>(smem + NS * STAGE);   // [BM] padded output pixel index
    int *rowimg = rowpix + BM;                                  // [BM] image index

    const int C0 = p.in0.C, C1 = p.in1.p ? p.in1.C : 0;
    const int Cin = C0 + C1;
    const int Cout = p.out.C;
    const int HWo = p.Hout * p.Wout;
    const int M = p.B * HWo;
    const int tilesN = (Cout + BN - 1) / BN;
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
    constexpr int TAPS = KS * KS;
    // split-K: this block reduces chunks [cb, ce) of the input channels (all taps) and chunks [c2b, c2e) of the
    // fused 1x1 term (res_conv) — both divided, so the splits carry equal numbers of K-steps
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int nchunk = Cin / BK;
    const int cb = (int)((long)nchunk * split / nsplit) * BK, ce = (int)((long)nchunk * (split + 1) / nsplit) * BK;
    const int C2a = p.in2.p ? p.in2.C : 0, C2t = C2a + (p.in2b.p ? p.in2b.C : 0);
    const int nchunk2 = C2t / BK;
    const int c2b = (int)((long)nchunk2 * split / nsplit) * BK, c2e = (int)((long)nchunk2 * (split + 1) / nsplit) * BK;
    const int nk = TAPS * ((ce - cb) / BK) + (c2e - c2b) / BK;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wid >= 4) {
        // ------------------------------- producer waves -------------------------------------
        const int w = wid - 4;
        const int tid = threadIdx.x - 256;
        // epilogue tables: one row per thread (clamped rows are never stored)
        if (tid < BM) {
            const int m = min(m0 + tid, M - 1);
            const int n = div_hw(p, m, HWo);
            const int rem = m - n * HWo;
            const int oy = div_w(p, rem, p.Wout);
            rowpix[tid] = (int)p.out.pix(n, oy * p.out_step + p.out_oy, (rem - oy * p.Wout) * p.out_step + p.out_ox);
            rowimg[tid] = n;
        }
        // DMA instruction i of this wave fills tile rows (4i + w) * 8 + (lane >> 3), lane & 7 is
        // the 16-B position inside the row; the source chunk is the swizzled one.
        // Addresses are (uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset that is
        // constant over the whole K loop), so a K-step costs the producers 8 DMA instructions
        // and a few scalar adds — no vector ALU work (it would stall the co-resident MFMAs).
        const int rsub = lane >> 3;
        const unsigned schunk16 = (unsigned)(((lane & 7) ^ ((((w & 1) << 2) | (lane >> 4)) & 7)) * 16);
        constexpr int cpad = KS >> 1;
        const int tpad = p.in0.pad;
        const int Hp = p.in0.Hp(), Wp = p.in0.Wp();

        unsigned vA0[AR], vA1[AR];            // offsets of the window origin (in0 / in1)
        static_for<AR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = min(m0 + (4 * i + w) * 8 + rsub, M - 1);
            const int n = div_hw(p, m, HWo);
            const int rem = m - n * HWo;
            const int oy = div_w(p, rem, p.Wout);
            const int ox = rem - oy * p.Wout;
            const unsigned pixbase = (unsigned)((n * Hp + oy * p.stride - cpad + tpad + p.org_y) * Wp +
                                                ox * p.stride - cpad + tpad + p.org_x);
            vA0[i] = pixbase * (unsigned)C0 * 4u + schunk16;
            vA1[i] = pixbase * (unsigned)C1 * 4u + schunk16;
        });
        unsigned vB[BR], vB2[BR], vA2[AR], vA2b[AR];
        static_for<BR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int n = min(n0 + (4 * i + w) * 8 + rsub, Cout - 1);
            vB[i] = (unsigned)n * (unsigned)Cin * 4u + schunk16;
            vB2[i] = (unsigned)n * (unsigned)C2t * 4u + schunk16;
        });
        static_for<AR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = min(m0 + (4 * i + w) * 8 + rsub, M - 1);
            const int n = div_hw(p, m, HWo);
            const int rem = m - n * HWo;
            const int oy = div_w(p, rem, p.Wout);
            vA2[i] = C2t ? (unsigned)p.in2.pix(n, oy, rem - oy * p.Wout) * (unsigned)C2a * 4u + schunk16 : 0u;
            vA2b[i] = C2t > C2a ? (unsigned)p.in2b.pix(n, oy, rem - oy * p.Wout) * (unsigned)(C2t - C2a) * 4u + schunk16 : 0u;
        });
        const size_t tapstride = (size_t)Cout * Cin;   // floats between taps of the packed weights

        // step kt: consumers multiply tile kt out of stage kt&1 while tile kt+1 streams into the
        // other stage (free since the barrier that ended step kt-1); the barrier's implied
        // vmcnt(0) makes the DMA data visible before anybody reads it.
        int k = 0;
        for (int c0 = cb; c0 < ce; c0 += BK) {
            const bool first = c0 < C0;
            const int Cs = first ? C0 : C1;
            const char *abase = reinterpret_cast<const char *>((first ? p.in0.p : p.in1.p) + (first ? c0 : c0 - C0));
            const char *wbase = reinterpret_cast<const char *>(p.w + c0);
            static_for<TAPS>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int dy = tap / KS, dx = tap % KS;
                float *Ad = smem + (k % NS) * STAGE + w * 256;
                float *Bd = Ad + BM * ROWF;
                if (!(SR3_DBG(p) & 1) || k == 0) {
                    const char *ab = abase + (size_t)(dy * Wp + dx) * Cs * 4;
                    if (SR3_DBG(p) & 2) ab = reinterpret_cast<const char *>(p.in0.p);   // experiment: cache-hot source
                    if (first) {
                        static_for<AR>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            dma16s<DMA_BUF>(ab, vA0[i], Ad + i * 1024);
                        });
                    } else {
                        static_for<AR>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            dma16s<DMA_BUF>(ab, vA1[i], Ad + i * 1024);
                        });
                    }
                    const char *wb = (SR3_DBG(p) & 2) ? reinterpret_cast<const char *>(p.w) : wbase + (size_t)tap * tapstride * 4;
                    static_for<BR>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        dma16s<DMA_BUF>(wb, vB[i], Bd + i * 1024);
                    });
                }
                if (k >= NS - 2) producer_sync<(NS - 2) * (AR + BR)>();
                ++k;
            });
        }
        // fused 1x1 term: K-steps over the channels of in2, read at the output pixel
        for (int c0 = c2b; c0 < c2e; c0 += BK) {
            float *Ad = smem + (k % NS) * STAGE + w * 256;
            float *Bd = Ad + BM * ROWF;
            const bool first2 = c0 < C2a;
            const char *ab = reinterpret_cast<const char *>(first2 ? p.in2.p + c0 : p.in2b.p + (c0 - C2a));
            const char *wb = reinterpret_cast<const char *>(p.w2 + c0);
            if (first2) {
                static_for<AR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dma16s<DMA_BUF>(ab, vA2[i], Ad + i * 1024);
                });
            } else {
                static_for<AR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dma16s<DMA_BUF>(ab, vA2b[i], Ad + i * 1024);
                });
            }
            static_for<BR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                dma16s<DMA_BUF>(wb, vB2[i], Bd + i * 1024);
            });
            if (k >= NS - 2) producer_sync<(NS - 2) * (AR + BR)>();
            ++k;
        }
        // drain: the last NS-2 tiles are still in flight
        static_for<NS - 2>([&](auto rc) {
            constexpr int r = NS - 3 - decltype(rc)::value;     // NS-3 ... 0 tiles may stay in flight
            if (r < nk) producer_sync<r * (AR + BR)>();
        });
        __syncthreads();
        // (in-place split-K: the consumer waves of the tile's last block write the statistics themselves)
        if (p.stats != nullptr && nsplit == 1) producer_stats_tail<BM, BN, WGM>(p, smem, m0, n0, HWo);
        return;
    }

    // ----------------------------------- consumer waves -----------------------------------------
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int swz = (li >> 1) & 7;          // == ((row >> 1) & 7) for every row this lane reads
    const float *Abase = smem + (wm * WM + li) * ROWF;
    const float *Bbase = smem + BM * ROWF + (wn * WN + li) * ROWF;
    if constexpr (PREC == 0) {
        // Fragment reads run one 8-k group ahead of the MFMAs (two register sets); the reads of
        // the next tile's first group are issued right after the barrier and land under the last
        // group's MFMAs, so the consumer never waits on LDS latency.
        int koff[BK / 8];                   // float offset of chunk (2kk + lh) after swizzling
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) koff[kk] = (((2 * kk + lh) ^ swz) & 7) * 4;
        f32x4 fa[2][MI], fb[2][NI];
#define SR3_FRAG_READ(SET, CUR, KK)                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) fa[SET][mi] =                            \
            *reinterpret_cast<const f32x4 *>(Abase + (CUR) * STAGE + mi * 32 * ROWF + koff[KK]);   \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) fb[SET][ni] =                            \
            *reinterpret_cast<const f32x4 *>(Bbase + (CUR) * STAGE + ni * 32 * ROWF + koff[KK]);   \
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
            const int cur = kt % NS, nxt = (kt + 1) % NS;                  // LDS stage of tile kt / kt+1
            constexpr bool sync = true;
            SR3_FRAG_READ(1, cur, 1)
            SR3_FRAG_MMA(0)
            SR3_FRAG_READ(0, cur, 2)
            SR3_FRAG_MMA(1)
            SR3_FRAG_READ(1, cur, 3)
            SR3_FRAG_MMA(0)
            if (sync) __syncthreads();         // every read of this stage has been issued and waited
            SR3_FRAG_READ(0, nxt, 0)           // next tile (stale data after the last one: unused)
            SR3_FRAG_MMA(1)
        }
#undef SR3_FRAG_READ
#undef SR3_FRAG_MMA
    } else {
        // split-f16: a row is 32 hi halfs (16-B chunks 0..3) | 32 lo halfs (chunks 4..7). For the
        // 16-wide K block s (0|1), lane half h holds k = 16s + 8h + j: hi chunk 2s+h, lo chunk
        // 4+2s+h. Per 32x32 tile and K block: acc += Al*Bh + Ah*Bl + Ah*Bh (fp32 accumulate).
        int hoff[2], loff[2];               // float offsets of the swizzled chunks
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            hoff[sb] = (((2 * sb + lh) ^ swz) & 7) * 4;
            loff[sb] = (((4 + 2 * sb + lh) ^ swz) & 7) * 4;
        }
        h16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
#define SR3_FRAG_READ(SET, CUR, SB)                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                        \
            ah[SET][mi] = *reinterpret_cast<const h16x8 *>(Abase + (CUR) * STAGE + mi * 32 * ROWF + hoff[SB]); \
            al[SET][mi] = *reinterpret_cast<const h16x8 *>(Abase + (CUR) * STAGE + mi * 32 * ROWF + loff[SB]); \
        }                                                                                          \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                        \
            bh[SET][ni] = *reinterpret_cast<const h16x8 *>(Bbase + (CUR) * STAGE + ni * 32 * ROWF + hoff[SB]); \
            bl[SET][ni] = *reinterpret_cast<const h16x8 *>(Bbase + (CUR) * STAGE + ni * 32 * ROWF + loff[SB]); \
        }                                                                                          \
    }
#define SR3_FRAG_MMA(SET)                                                                          \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                        \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[SET][mi], bh[SET][ni], acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[SET][mi], bl[SET][ni], acc[mi][ni], 0, 0, 0); \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[SET][mi], bh[SET][ni], acc[mi][ni], 0, 0, 0); \
        }                                                                                          \
    }
        __syncthreads();
        SR3_FRAG_READ(0, 0, 0)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt % NS, nxt = (kt + 1) % NS;
            constexpr bool sync = true;
            SR3_FRAG_READ(1, cur, 1)
            SR3_FRAG_MMA(0)
            if (sync) __syncthreads();         // every read of this stage has been issued and waited
            SR3_FRAG_READ(0, nxt, 0)
            SR3_FRAG_MMA(1)
        }
#undef SR3_FRAG_READ
#undef SR3_FRAG_MMA
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= p.w_unscale;
    }

    if constexpr (BM == 64 && BN == 64) {
    if (nsplit > 1 && p.tile_cnt != nullptr) {
        // In-place split-K (64x64 tiles; launch_conv guarantees whole tiles in M and N): every block leaves its
        // partial sums in part[split]; the block that arrives LAST at the tile's counter adds the partials of all
        // splits in split order (so the result does not depend on which block that is) and runs the whole
        // epilogue — no second kernel, and the fused GroupNorm statistics keep the one-slice-per-M-tile layout of
        // an unsplit conv. The producer waves have retired by now (a barrier counts live waves only).
        // Visibility across the XCDs' L2s without a device-scope fence (a release fence writes the WHOLE L2 back:
        // measured +22 us per conv): the partials are stored and loaded as relaxed device-scope atomics — write-
        // through stores / loads that bypass the non-coherent cache levels (sc1) — and ordered against the counter
        // by an explicit `s_waitcnt vmcnt(0)` in every consumer wave before the barrier (see below).
        // Addresses: wave-uniform 64-bit base + one 32-bit lane offset (the partial buffer is far below 4 GiB).
        const size_t MC = (size_t)M * Cout;
        const unsigned lane_off = ((unsigned)(4 * lh) * (unsigned)Cout + (unsigned)li) * 4u;
        const float *tile0 = p.part + (size_t)(m0 + wm * WM) * Cout + n0 + wn * WN;      // wave-uniform
        auto elem = [&](int sp, int mi, int ni, int r) -> float * {
            const float *ub = tile0 + (size_t)sp * MC + (size_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * Cout + ni * 32;
            return reinterpret_cast<float *>(reinterpret_cast<uintptr_t>(ub) + lane_off);
        };
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(elem(split, mi, ni, r), acc[mi][ni][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The guarantee relied on: an sc1 (agent-scope, write-through) store is counted by vmcnt until the coherence
        // point (the memory side past the XCD's L2) has acknowledged it, so `s_waitcnt vmcnt(0)` in EVERY consumer wave,
        // then the barrier, then the counter RMW orders "partials visible device-wide" before "block counted". A
        // workgroup-scope release fence alone does NOT emit that wait (without tgsplit it needs none; the round-2 code
        // compiled to lgkmcnt(0) only), hence the explicit instruction.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the stores have completed ...
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();                            // ... in all four consumer waves, before the tile counts this block
        int *last_flag = reinterpret_cast<int *>(smem) + NS * STAGE - 4;   // (far behind the statistics staging area)
        if (threadIdx.x == 0) {
            unsigned *cnt = p.tile_cnt + (size_t)blockIdx.z * gridDim.x + blockIdx.x;
            const unsigned arrived = atomicAdd(cnt, 1u);
            const int last = arrived == (unsigned)(nsplit - 1);
            if (last) *cnt = 0u;                    // ready for the next launch (nobody else touches it any more)
            *last_flag = last;
        }
        __syncthreads();
        if (*last_flag == 0) return;
        // U splits x the whole accumulator tile in flight per round trip, added in split order
        constexpr int U = 4 / (MI * NI);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        for (int sp = 0; sp < nsplit; sp += U) {
            float t[U][MI][NI][16];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            t[u][mi][ni][r] = __hip_atomic_load(elem(min(sp + u, nsplit - 1), mi, ni, r), __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (sp + u < nsplit) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[mi][ni][r] += t[u][mi][ni][r];
                }
        }
        conv_epilogue<BM, BN, WGM, WGN, MI, NI>(p, acc, smem, rowpix, rowimg, m0, n0, M, wm, wn, li, lh, 0, 1);
        if (p.stats != nullptr) {                   // (the epilogue ended with a barrier behind the staged column sums)
            const double2 *red = reinterpret_cast<const double2 *>(smem);
            const int col = threadIdx.x;
            if (col < BN) {
                double a = 0, b = 0;
#pragma unroll
                for (int j = 0; j < WGM * 2; ++j) { const double2 t2 = red[j * BN + col]; a += t2.x; b += t2.y; }
                const int n = m0 / HWo, slice = p.stats_slice0 + (m0 - n * HWo) / BM;
                double *o = p.stats + (((size_t)n * p.stats_slices + slice) * Cout + n0 + col) * 2;
                o[0] = a; o[1] = b;
            }
        }
        return;
    }
    }
    conv_epilogue<BM, BN, WGM, WGN, MI, NI>(p, acc, smem, rowpix, rowimg, m0, n0, M, wm, wn, li, lh, split, nsplit);
}

// =================================================================================================
// x-halo variant (split-f16, 3x3, stride 1, no upsample): the three dx taps of one (chunk, dy) read
// the same pixels shifted by one, so the A operand is staged ONCE per (chunk, dy) as row segments
// with a 1-pixel halo on each side and the consumers address it at row offset dx. A-side LDS-DMA
// instructions drop from 9 to ~3.1 per chunk (the DMA issue cost is what limits split-f16 mode).
//   tile = BM consecutive output pixels = nseg segments of SEG = min(W, BM) pixels of one image row;
//   A stage = nseg x (SEG + 2) rows of 128 B, row R of segment s = padded pixel (n, y + dy, x0 + R);
//   consumer row for tile row r and tap dx: r + 2 * (r / SEG) + dx.
// A ring: 2 stages, group g = (chunk, dy) lives in stage g & 1 and is issued two K-steps ahead (at
// the dx = 1 step of the previous group); B ring: 2 stages, one tile per K-step as before.
// The fused 1x1 term (in2) uses plain BM-row A tiles in the same A ring.
// =================================================================================================
// MS: MFMA shape of the consumers, 32 (v_mfma_f32_32x32x16_f16) or 16 (v_mfma_f32_16x16x32_f16: same
// FLOP per cycle, but the chip holds a higher clock on it under load — MI355X_MICROARCH.md, DVFS (7))
// GNF: producer-side GroupNorm of the output (ConvParams::gnf_*; MS == 16 only)
// SPK: in-place split-K instantiation (ConvParams::splits > 1; MS == 16, KS == 3). A template parameter, not a run-time
// branch: with the split code compiled into the one kernel the register allocation of the unsplit K loop changed
// (scratch 76 -> 430 bytes per lane) and EVERY x-halo conv ran 20 % slower (same box: 16.35 -> 19.29 ms per B = 64 step).
template <int BM, int BN, int WGM, int WGN, int SEGMIN, int KS, int MS, bool GNF = false, bool SPK = false, bool F8C = false>
__global__ __launch_bounds__(512, (BN == 64 && MS == 16 && BM == 128) ? 6 : 4) void conv3x3_halo_h3(const ConvParams p_in) {
    static_assert(!F8C || (MS == 32 && KS == 3 && !GNF && !SPK), "fp8 correction products: 3x3 convs on the 32x32 consumers");
    static_assert(!GNF || (MS == 16 && KS == 3), "producer-side GroupNorm: 3x3 convs on the 16x16x32 consumers");
    static_assert(!SPK || (MS == 16 && KS == 3 && !GNF), "in-place split-K: 3x3 convs on the 16x16x32 consumers");
    const ConvParams p = phase_params(p_in, blockIdx.z);
    static_assert(WGM * WGN == 4, "4 consumer waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int HALO = KS - 1, TAPS = KS * KS;         // KS 3: 3x3 conv; KS 2: one sub-pixel phase of the upsample conv
    constexpr int RA = (BM + HALO * (BM / SEGMIN) + 7) / 8 * 8;   // rows of one A stage (worst case, whole DMA instructions)
    constexpr int ARH = (RA / 8 + 3) / 4;               // halo DMA instructions per producer wave
    constexpr int ASTG = RA * ROWF, BSTG = BN * ROWF;
    static_assert((RA % 8) == 0, "A stage rows");

    // (128-byte alignment: the pattern K loop derives a row's lo-chunk address as hi ^ 64; every ring row is 128 bytes and
    // both rings start at multiples of 128 bytes)
    extern __shared__ __attribute__((aligned(128))) float smem[];
    static_assert((ASTG * 4) % 128 == 0 && (BSTG * 4) % 128 == 0, "ring stages are whole 128-byte rows");
    float *Aring = smem;                    // [2][RA][32]
    float *Bring = smem + 2 * ASTG;         // [2][BN][32]
    int *rowpix = reinterpret_cast<int *>(smem + 2 * ASTG + 2 * BSTG);
    int *rowimg = rowpix + BM;
    float *colbias = reinterpret_cast<float *>(rowimg + BM);    // [BN] + 1 flag word (conv_epilogue16)

    const int C0 = p.in0.C, C1 = p.in1.p ? p.in1.C : 0;
    const int Cin = C0 + C1;
    const int Cout = p.out.C;
    const int W = p.Wout;
    const int HWo = p.Hout * W;
    const int M = p.B * HWo;
    const int tilesN = (Cout + BN - 1) / BN;
    const int nsplit = SPK ? p.splits : 1;     // in-place split-K (end of the consumer path)
    int bid = blockIdx.x;
    if (GNF) {
        // Block order of the producer-side GroupNorm: the blocks of one image must be dispatched together (they wait
        // for each other), whole images per XCD so that no group straddles two XCDs' dispatch sequences.
        //  band == 0 (an image is at most 32 blocks): XCD x = blockIdx.x & 7 takes images x, x + 8, ... one after the
        //    other (the grid is padded to whole rounds of eight images; surplus blocks leave at once);
        //  band > 0 (128x128-pixel level: 128 M-tiles per image): XCD x takes M-tiles [x * band, (x + 1) * band) of
        //    EVERY image, images in order — an image is spread over the eight XCDs, band x tilesN blocks on each, and
        //    neighbouring image rows still share an L2.
        const int xcd = bid & 7, loc = bid >> 3, TMI = HWo / BM;
        if (p.gnf_band > 0) {
            const int per = p.gnf_band * tilesN;
            const int image = loc / per, r = loc - image * per;
            const int j = r / tilesN;
            bid = (image * TMI + xcd * p.gnf_band + j) * tilesN + (r - j * tilesN);
        } else {
            const int per = TMI * tilesN;
            const int image = (loc / per) * 8 + xcd;
            if (image >= p.B) return;
            bid = image * per + (loc % per);
        }
    } else if (SPK) {
        // split-K: the grid is (tiles padded to a multiple of 8) x splits in ONE dimension; XCD x = blockIdx.x & 7 takes
        // tiles [x * tpx, (x + 1) * tpx), the splits of a tile adjacent in its dispatch sequence — the blocks that wait
        // for each other at the end of the kernel are always dispatched together, whatever the tile count
        const int xcd = bid & 7, loc = bid >> 3;
        const int tpx = (int)gridDim.x / (8 * nsplit);
        bid = (xcd * tpx + loc / nsplit) * nsplit + loc % nsplit;
        if (bid / nsplit >= (M / BM) * tilesN) return;          // padding
    } else {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, loc = bid >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int split = bid % nsplit;
    bid /= nsplit;
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;
    const int SEG = min(W, BM), SEGP = SEG + HALO, nseg = BM / SEG;
    const int rows_a = nseg * SEGP;
    const int C2a = p.in2.p ? p.in2.C : 0, C2 = C2a + (p.in2b.p ? p.in2b.C : 0);
    // split-K (in place, see the end of the kernel): this block reduces the input-channel chunks
    // [cb, ce) (all taps) and the chunks [c2b, c2e) of the fused 1x1 term — both divided, like the generic kernel
    const int nchunk = Cin / BK, nchunk2 = C2 / BK;
    const int cb = SPK ? (int)((long)nchunk * split / nsplit) * BK : 0, ce = SPK ? (int)((long)nchunk * (split + 1) / nsplit) * BK : Cin;
    const int c2b = SPK ? (int)((long)nchunk2 * split / nsplit) * BK : 0, c2e = SPK ? (int)((long)nchunk2 * (split + 1) / nsplit) * BK : C2;
    const int nkh = TAPS * ((ce - cb) / BK);           // halo-phase K-steps
    const int nk = nkh + (c2e - c2b) / BK;
    const int G = nkh / KS;                             // A groups of the halo phase
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wid >= 4) {
        // ------------------------------- producer waves -------------------------------------
#if defined(SR3_CONV_VARIANT) && SR3_CONV_VARIANT == 3
        __builtin_amdgcn_s_setprio(1);      // experiment: the DMA waves win issue arbitration
#endif
        const int w = wid - 4;
        const int tid = threadIdx.x - 256;
        // Epilogue tables (row -> output pixel / image, column bias). They are not read before the epilogue, and the
        // bias loads have a memory round trip of their own: with SR3_LATE_TABLES they are set up behind the DMAs of
        // K-step 0 instead of in front of the first DMA (profiles/README.md finding 61).
#ifndef SR3_LATE_TABLES
#define SR3_LATE_TABLES 0
#endif
        auto setup_tables = [&]() {
            if (tid < BM) {
                const int m = min(m0 + tid, M - 1);
                const int n = div_hw(p, m, HWo);
                const int rem = m - n * HWo;
                const int oy = div_w(p, rem, W);
                rowpix[tid] = (int)p.out.pix(n, oy * p.out_step + p.out_oy, (rem - oy * W) * p.out_step + p.out_ox);
                rowimg[tid] = n;
            }
            if constexpr (MS == 16 || F8C) {    // column bias of the epilogue, fetched now, read from LDS later
                const int img_a = div_hw(p, m0, HWo), img_b = div_hw(p, min(m0 + BM - 1, M - 1), HWo);
                const bool one = img_a == img_b;
                if (tid < BN) {
                    const int nc = min(n0 + tid, Cout - 1);
                    float v = p.bias ? p.bias[nc] : 0.f;
                    if (p.chan_bias != nullptr && one) v += p.chan_bias[(size_t)img_a * p.chan_bias_stride + nc];
                    colbias[tid] = v;
                }
                if (tid == 0) reinterpret_cast<int *>(colbias)[BN] = one ? 1 : 0;
            }
        };
        if constexpr (!SR3_LATE_TABLES) setup_tables();
        const int rsub = lane >> 3;
        const unsigned schunk16 = (unsigned)(((lane & 7) ^ ((((w & 1) << 2) | (lane >> 4)) & 7)) * 16);
        const int Hp = p.in0.Hp(), Wp = p.in0.Wp();
        // halo rows: DMA instruction i of this wave fills A-stage rows (4i + w) * 8 + rsub
        unsigned vH0[ARH], vH1[ARH];
        static_for<ARH>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int R = (4 * i + w) * 8 + rsub;
            int sg = R / SEGP, jx = R - sg * SEGP;
            if (sg >= nseg) { sg = nseg - 1; jx = 0; }              // unused tail rows: any valid pixel
            const int m = m0 + sg * SEG;
            const int n = div_hw(p, m, HWo);
            const int rem = m - n * HWo;
            const int y = div_w(p, rem, W), x0 = rem - y * W;
            // KS 3: padded coordinates of (y - 1 + dy, x0 - 1 + jx) are (y + dy, x0 + jx); KS 2: the phase's
            // window starts at padded (y + org_y, x0 + org_x); dy is added per group
            const unsigned pix = (unsigned)((n * Hp + y + p.org_y) * Wp + x0 + jx + p.org_x);
            vH0[i] = pix * (unsigned)C0 * 4u + schunk16;
            vH1[i] = pix * (unsigned)C1 * 4u + schunk16;
        });
        unsigned vB[BR], vB2[BR], vA2[AR], vA2b[AR];
        static_for<BR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int n = min(n0 + (4 * i + w) * 8 + rsub, Cout - 1);
            vB[i] = (unsigned)n * (unsigned)Cin * 4u + schunk16;
            vB2[i] = (unsigned)n * (unsigned)C2 * 4u + schunk16;
        });
        static_for<AR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = min(m0 + (4 * i + w) * 8 + rsub, M - 1);
            const int n = div_hw(p, m, HWo);
            const int rem = m - n * HWo;
            const int oy = div_w(p, rem, W);
            vA2[i] = C2 ? (unsigned)p.in2.pix(n, oy, rem - oy * W) * (unsigned)C2a * 4u + schunk16 : 0u;
            vA2b[i] = C2 > C2a ? (unsigned)p.in2b.pix(n, oy, rem - oy * W) * (unsigned)(C2 - C2a) * 4u + schunk16 : 0u;
        });
        const size_t tapstride = (size_t)Cout * Cin;

        // issue the halo tile of group (c0, dy) into A stage (ga & 1)
        int ga = 0;
#define SR3_ISSUE_HALO(C0A, DY)                                                                    \
    {                                                                                              \
        const int c0a_ = (C0A);                                                                    \
        const bool first_ = c0a_ < C0;                                                             \
        const int Cs_ = first_ ? C0 : C1;                                                          \
        const char *ab_ = reinterpret_cast<const char *>((first_ ? p.in0.p : p.in1.p) + (first_ ? c0a_ : c0a_ - C0)) + \
                          (size_t)(DY) * Wp * Cs_ * 4;                                             \
        float *Ad_ = Aring + (ga & 1) * ASTG + w * 256;                                            \
        /* (a uniform branch instead of a per-lane select of the offset in front of every DMA) */   \
        if (first_) {                                                                              \
            static_for<ARH>([&](auto ic) {                                                         \
                constexpr int i = decltype(ic)::value;                                             \
                if ((4 * i + w) * 8 < rows_a) dma16s<true>(ab_, vH0[i], Ad_ + i * 1024);                 \
            });                                                                                    \
        } else {                                                                                   \
            static_for<ARH>([&](auto ic) {                                                         \
                constexpr int i = decltype(ic)::value;                                             \
                if ((4 * i + w) * 8 < rows_a) dma16s<true>(ab_, vH1[i], Ad_ + i * 1024);                 \
            });                                                                                    \
        }                                                                                          \
        ++ga;                                                                                      \
    }

        int nh = 0;                                     // halo DMA instructions of this wave per group
        static_for<ARH>([&](auto ic) { if ((4 * decltype(ic)::value + w) * 8 < rows_a) ++nh; });
        int k = 0;
        SR3_ISSUE_HALO(cb, 0)
        for (int c0 = cb; c0 < ce; c0 += BK) {
            const char *wbase = reinterpret_cast<const char *>(p.w + c0);
            static_for<TAPS>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int dy = tap / KS, dx = tap % KS;
                const bool feed = !(SR3_DBG(p) & 1) || k == 0;      // experiment: operands only for the first K-step
                // experiment bit 7 (timing only, results wrong): every other block issues no weight DMA after its first K-step —
                // what the launch would cost if co-resident blocks SHARED their weight tiles (37 % fewer L2 bytes on a 128x128
                // tile) without being tied to each other's barriers (profiles/README.md finding 72)
                const bool feed_b = feed && (!(SR3_DBG(p) & 128) || !(blockIdx.x & 1) || k == 0);
                // The B tile of this K-step goes out FIRST: the consumers wait for it at the very next barrier. The A
                // halo group issued at dx == 1 is not read before the K-step after next, so it goes out behind the B
                // tile and stays in flight across this K-step's barrier (counted wait: the wave's nh youngest DMAs).
                float *Bd = Bring + (k & 1) * BSTG + w * 256;
                const char *wb = wbase + (size_t)tap * tapstride * 4;
                if (feed_b) {
                    static_for<BR>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        dma16s<true>(wb, vB[i], Bd + i * 1024);
                    });
                }
                bool halo_now = false;
                if (dx == 1 && feed) {  // the stage of group g-1 is free once K-step KS*g - 1 has been read
                    if (dy < KS - 1) {
                        SR3_ISSUE_HALO(c0, dy + 1)
                        halo_now = true;
                    } else if (c0 + BK < ce) {
                        SR3_ISSUE_HALO(c0 + BK, 0)
                        halo_now = true;
                    }
                }
#ifdef SR3_EXPERIMENTS
                {   // experiment (profiles/README.md, GroupNorm-apply fusion): bits 8..15 = N dummy vector operations per
                    // K-step in every producer lane (every 8th one a v_exp_f32), the VALU load an in-kernel
                    // a*x+b / Swish / hi-lo split of the A tile would add next to the f16 MFMAs
                    // eight independent chains (the real work has that much parallelism: ~17 elements per lane)
                    const int nv = (p.dbg >> 8) & 0xFF;
                    float z[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) z[u] = (float)(lane + u);
                    for (int i = 0; i < nv; i += 8) {
#pragma unroll
                        for (int u = 0; u < 7; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z[u]) : "v"(1.000001f));
                        z[7] = __expf(z[7] * 1e-6f);
                    }
                    float zs = 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) zs += z[u];
                    if (zs == 1234.5f) rowimg[0] = 1;     // keep the chains alive
                }
#endif
                if constexpr (SR3_LATE_TABLES && tap == 0) {
                    if (c0 == cb) setup_tables();
                }
                if (!(SR3_DBG(p) & 8)) {                            // experiment bit 3: no barriers (timing only)
                    if (dx == 1 && halo_now) {
                        static_for<ARH + 1>([&](auto nc) {
                            if (nh == decltype(nc)::value) producer_sync<decltype(nc)::value>();
                        });
                    } else {
                        producer_sync<0>();
                    }
                }
                ++k;
            });
        }
#undef SR3_ISSUE_HALO
        // fused 1x1 term: plain BM-row A tiles continue in the A ring
        for (int c0 = c2b; c0 < c2e; c0 += BK) {
            float *Ad = Aring + (ga & 1) * ASTG + w * 256;
            float *Bd = Bring + (k & 1) * BSTG + w * 256;
            const bool first2 = c0 < C2a;
            const char *ab = reinterpret_cast<const char *>(first2 ? p.in2.p + c0 : p.in2b.p + (c0 - C2a));
            const char *wb = reinterpret_cast<const char *>(p.w2 + c0);
            if (first2) {
                static_for<AR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dma16s<true>(ab, vA2[i], Ad + i * 1024);
                });
            } else {
                static_for<AR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dma16s<true>(ab, vA2b[i], Ad + i * 1024);
                });
            }
            static_for<BR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                dma16s<true>(wb, vB2[i], Bd + i * 1024);
            });
            producer_sync<0>();
            ++k;
            ++ga;
        }
        __syncthreads();
        if constexpr (GNF) {
            gnf_producer_tail<BM, BN, WGM * 4>(p, smem, m0, n0, HWo, tid);
            return;
        }
        // (in-place split-K: the consumer waves of the tile's last block write the statistics themselves)
        if (p.stats != nullptr && nsplit == 1) producer_stats_tail<BM, BN, WGM, MS == 16 ? 4 : 2>(p, smem, m0, n0, HWo);
        return;
    }

    // ----------------------------------- consumer waves -----------------------------------------
    if constexpr (MS == 16) {
        // 16x16x32 tiles: lane (l16, q) holds row/col l16 and the 8 halfs of k-chunk q (hi: 16-B chunk
        // q, lo: chunk 4 + q of the 128-B row) — one MFMA covers the whole 32-channel K-step.
        // Registers: all A fragments of the K-step (MT x {hi, lo}) + two B column buffers; column nt+1
        // is fetched while column nt is multiplied; after the barrier the last column's MFMAs release
        // the A fragments one row tile at a time and the next K-step's are fetched into them.
        constexpr int MT = WM / 16, NT = WN / 16;
        static_assert((NT % 2) == 0, "column buffers alternate");
        const int l16 = lane & 15, q = lane >> 4;
        const int wm = wid / WGN, wn = wid % WGN;
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
        const int swzB = (l16 >> 1) & 7;                    // rows 16 apart share (row >> 1) & 7
        const float *Bbase = Bring + (wn * WN + l16) * ROWF;
        const int bho = ((q ^ swzB) & 7) * 4, blo = (((4 + q) ^ swzB) & 7) * 4;
        int rhalo[MT], rplain[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            rplain[mt] = wm * WM + mt * 16 + l16;
            rhalo[mt] = rplain[mt] + HALO * (rplain[mt] / SEG);
        }
        h16x8 ah[MT], al[MT], bqh[2], bql[2];
#ifndef SR3_CONV_VARIANT
#define SR3_CONV_VARIANT 1
#endif
#if SR3_CONV_VARIANT == 2
        __builtin_amdgcn_s_setprio(1);      // consumers win issue arbitration against the DMA waves
#endif
        const float *aptr_h[MT], *aptr_l[MT];   // variant >= 1: fragment addresses of the NEXT K-step, computed early
        // ONESEG: the WM rows of a wave lie in one row segment (SEG is a power of two >= SEGMIN), so row tile mt
        // is tile 0 + 16 * mt ring rows with the same swizzle term: one address pair per K-step, the other row
        // tiles are constant offsets of the reads (half the address arithmetic, two registers less)
        constexpr bool ONESEG = (SEGMIN % WM) == 0;
#define SR3_AADDR(MTI, KT)                                                                         \
    if (!ONESEG || (MTI) == 0) {                                                                   \
        const int kt_ = (KT);                                                                      \
        const bool halo_ = kt_ < nkh;                                                              \
        const int g_ = kt_ / KS;                                                                   \
        const int astage_ = halo_ ? (g_ & 1) : ((G + kt_ - nkh) & 1);                              \
        const int R_ = halo_ ? rhalo[MTI] + (kt_ - KS * g_) : rplain[MTI];                         \
        const int sw_ = (R_ >> 1) & 7;                                                             \
        const float *Ar_ = Aring + astage_ * ASTG + R_ * ROWF;                                     \
        aptr_h[MTI] = Ar_ + ((q ^ sw_) & 7) * 4;                                                   \
        aptr_l[MTI] = Ar_ + (((4 + q) ^ sw_) & 7) * 4;                                             \
    }
#define SR3_AREAD2(MTI)                                                                            \
    {                                                                                              \
        const int ai_ = ONESEG ? 0 : (MTI), ao_ = ONESEG ? (MTI) * 16 * ROWF : 0;                  \
        ah[MTI] = *reinterpret_cast<const h16x8 *>(aptr_h[ai_] + ao_);                             \
        al[MTI] = *reinterpret_cast<const h16x8 *>(aptr_l[ai_] + ao_);                             \
    }
#define SR3_AREAD(MTI, KT)                                                                         \
    {                                                                                              \
        const int kt_ = (KT);                                                                      \
        const bool halo_ = kt_ < nkh;                                                              \
        const int g_ = kt_ / KS;                                                                   \
        const int astage_ = halo_ ? (g_ & 1) : ((G + kt_ - nkh) & 1);                              \
        const int R_ = halo_ ? rhalo[MTI] + (kt_ - KS * g_) : rplain[MTI];                         \
        const int sw_ = (R_ >> 1) & 7;                                                             \
        const float *Ar_ = Aring + astage_ * ASTG + R_ * ROWF;                                     \
        ah[MTI] = *reinterpret_cast<const h16x8 *>(Ar_ + ((q ^ sw_) & 7) * 4);                     \
        al[MTI] = *reinterpret_cast<const h16x8 *>(Ar_ + (((4 + q) ^ sw_) & 7) * 4);               \
    }
#define SR3_BREAD(BUF, NTI, KT)                                                                    \
    {                                                                                              \
        const float *Bb_ = Bbase + ((KT) & 1) * BSTG + (NTI) * 16 * ROWF;                          \
        bqh[BUF] = *reinterpret_cast<const h16x8 *>(Bb_ + bho);                                    \
        bql[BUF] = *reinterpret_cast<const h16x8 *>(Bb_ + blo);                                    \
    }
#define SR3_MMA16(MTI, NTI, BUF)                                                                   \
    {                                                                                              \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[MTI], bqh[BUF], acc[MTI][NTI], 0, 0, 0); \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[MTI], bql[BUF], acc[MTI][NTI], 0, 0, 0); \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[MTI], bqh[BUF], acc[MTI][NTI], 0, 0, 0); \
    }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) SR3_AREAD(mt, 0)
        SR3_BREAD(0, 0, 0)
#ifdef SR3_EXPERIMENTS
        // timing experiments (results are wrong): bit 2 no fragment reads after the first K-step,
        // bit 3 no barriers, bit 4 no MFMAs
        const bool x_rd = !(p.dbg & 4), x_bar = !(p.dbg & 8), x_mma = !(p.dbg & 16);
#else
        constexpr bool x_rd = true, x_bar = true, x_mma = true;
#endif
        int kt = 0;
#if SR3_CONV_VARIANT >= 1
        if constexpr (ONESEG) {
            // Halo-phase K loop WITHOUT address arithmetic. The fragment addresses of K-step kt = KS g + dx depend on kt
            // through (B stage kt & 1, A stage g & 1, row shift dx) only: a pattern of period 2 KS. The loop is unrolled
            // over one period, so every read is one of 2 KS + 2 lane-constant base registers (row shift dx enters the
            // swizzle term, hence one A base pair per dx) plus an immediate for stage / row tile / column tile. Vector
            // instructions compete with the MFMAs for issue; this removes 12 of the consumers' 14 per K-step. The last
            // K-steps (the last 3 or 6 of the halo phase, the fused 1x1 K-steps) run through the general loop below.
            constexpr int PER = 2 * KS;
            // LDS byte addresses (the dynamic LDS segment starts at 0 and every row is 128-byte aligned, so the lo
            // chunk of a row — chunk index ^ 4 — is the hi address ^ 64: one base register per dx and one for B)
            typedef const h16x8 __attribute__((address_space(3))) *lds_frag;
            auto lds_addr = [](const float *pp) { return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)pp; };
            unsigned avh[KS];
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const int R_ = rhalo[0] + dx, sw_ = (R_ >> 1) & 7;
                avh[dx] = lds_addr(Aring + R_ * ROWF + ((q ^ sw_) & 7) * 4);
            }
            const unsigned bvh = lds_addr(Bbase + bho);
            // one K-step of the pattern at position u of the period (the fragments behind it follow the pattern too)
            auto kstep = [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int BS = u & 1, UN = (u + 1) % PER, BSN = (u + 1) & 1, ASN = (UN / KS) & 1, DXN = UN % KS;
                // (80-register kernel: the lo addresses are recomputed where they are used — one v_xor — instead
                // of being hoisted out of the loop as four more live registers, which spilled fragments)
                auto lo_of = [](unsigned hi) {
                    unsigned lo;
                    if constexpr (BN == 64) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(lo) : "v"(hi));
                    else lo = hi ^ 64u;
                    return lo;
                };
                const unsigned bvl = lo_of(bvh);
#pragma unroll
                for (int nt = 0; nt < NT - 1; ++nt) {
                    bqh[(nt + 1) & 1] = *(lds_frag)(bvh + (BS * BSTG + (nt + 1) * 16 * ROWF) * 4);
                    bql[(nt + 1) & 1] = *(lds_frag)(bvl + (BS * BSTG + (nt + 1) * 16 * ROWF) * 4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) SR3_MMA16(mt, nt, nt & 1)
                }
                __syncthreads();
                bqh[0] = *(lds_frag)(bvh + BSN * BSTG * 4);
                bql[0] = *(lds_frag)(bvl + BSN * BSTG * 4);
                const unsigned avl_ = lo_of(avh[DXN]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    SR3_MMA16(mt, NT - 1, (NT - 1) & 1)
                    __builtin_amdgcn_sched_barrier(0);
                    ah[mt] = *(lds_frag)(avh[DXN] + (ASN * ASTG + mt * 16 * ROWF) * 4);
                    al[mt] = *(lds_frag)(avl_ + (ASN * ASTG + mt * 16 * ROWF) * 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (strictly less: the K-step behind a pass must still be one of the pattern. Peeling the last period so that
            // it runs here too — only its last K-step fetching through the general path — measured 0.5 % SLOWER over the
            // step than leaving the last 3 or 6 halo K-steps to the general loop, which measured 0.5 % faster than no
            // pattern loop at all; same box, three alternating pairs each.)
            for (; kt + PER < nkh; kt += PER) static_for<PER>([&](auto uc) { kstep(uc); });
        }
#endif
        for (; kt < nk; ++kt) {
            const int kn = min(kt + 1, nk - 1);     // after the last K-step: re-read, unused
#if SR3_CONV_VARIANT >= 1
            // the next K-step's A addresses are computed in the shadow of the first columns' MFMAs and
            // the reads are pinned right behind the last MFMA that uses each fragment (left to itself
            // the scheduler sinks address arithmetic and reads to the top of the next iteration, in
            // front of the MFMAs that wait for them)
#pragma unroll
            for (int nt = 0; nt < NT - 1; ++nt) {
                SR3_BREAD((nt + 1) & 1, nt + 1, kt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) SR3_MMA16(mt, nt, nt & 1)
                if (nt == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) SR3_AADDR(mt, kn)
                }
            }
            __syncthreads();
            SR3_BREAD(0, 0, kn)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                SR3_MMA16(mt, NT - 1, (NT - 1) & 1)
                __builtin_amdgcn_sched_barrier(0);
                SR3_AREAD2(mt)
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int nt = 0; nt < NT - 1; ++nt) {
                if (x_rd) SR3_BREAD((nt + 1) & 1, nt + 1, kt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) if (x_mma) SR3_MMA16(mt, nt, nt & 1)
            }
            if (x_bar) __syncthreads();             // every read of K-step kt has been issued and waited
            if (x_rd) SR3_BREAD(0, 0, kn)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (x_mma) SR3_MMA16(mt, NT - 1, (NT - 1) & 1)
                if (x_rd) SR3_AREAD(mt, kn)
            }
#endif
        }
#undef SR3_AREAD
#undef SR3_AREAD2
#undef SR3_AADDR
#undef SR3_BREAD
#undef SR3_MMA16
        if constexpr (SPK) {
        {
            // In-place split-K as a reduce-scatter (launch_conv: deep-K convs over few 128x128 tiles — the 8x8 level at
            // B = 64). Every block leaves its raw partial sums in part[tile][split] in the wave's own register order
            // (1 KiB per wave instruction), counts itself on the tile's counter and waits until all nsplit blocks of the
            // tile have (they are adjacent in the grid: dispatched together; the poll is bounded all the same). Then
            // block `split` adds COLUMN SLICE `split` of all partials in split order — bit-identical whatever the
            // arrival order — and runs the epilogue and the statistics for those BN / nsplit columns: the tail of a
            // tile is shared by its blocks instead of falling on the last one. Hand-off form as in the 64x64 kernel:
            // sc1 stores, s_waitcnt vmcnt(0) in every storing wave, barrier, one agent-scope add; sc1 loads.
            // The producer waves have retired (a barrier counts live waves only).
            const size_t tstride = (size_t)4 * MT * NT * 256;                     // floats of one partial tile image
            float *pbase = p.part + (size_t)bid * nsplit * tstride + ((size_t)wid * MT * NT * 64 + lane) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(pbase + (size_t)split * tstride + (size_t)(mt * NT + nt) * 256);
                    const unsigned long long lo2 = (unsigned long long)__float_as_uint(acc[mt][nt][0]) | ((unsigned long long)__float_as_uint(acc[mt][nt][1]) << 32);
                    const unsigned long long hi2 = (unsigned long long)__float_as_uint(acc[mt][nt][2]) | ((unsigned long long)__float_as_uint(acc[mt][nt][3]) << 32);
                    __hip_atomic_store(dst, lo2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, hi2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the partials have reached the coherence point ...
            __syncthreads();                                        // ... in all four consumer waves
            unsigned *cnt = p.tile_cnt + (size_t)bid;
            if (threadIdx.x == 0) {
                const unsigned arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived + 1u < (unsigned)nsplit) {
                    // Bounded wait (SR3_WAIT_TICKS of the constant 100 MHz counter = 5 ms). Liveness assumption: the
                    // nsplit blocks of a tile are neighbours in the grid and the chip has a free slot for each of them —
                    // true on an otherwise idle GPU (512 blocks for 512 slots at the 8x8 level of B = 64). If a co-tenant
                    // kernel holds the slots a sibling needs, the wait gives up, the flag makes the API replay the work
                    // with ConvParams::no_halo_split (a correct result either way, never a hang).
                    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nsplit) {
                        __builtin_amdgcn_s_sleep(8);
                        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > SR3_WAIT_TICKS) {
                            if (p.ovf != nullptr) atomicOr(p.ovf, SR3_FLAG_GNF_TIMEOUT);
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            const float *colb = colbias;
            const bool one_img = reinterpret_cast<const int *>(colbias)[BN] != 0;
            auto slice = [&](auto ntq_c) {
                constexpr int NTQ = decltype(ntq_c)::value;             // column tiles of this block's slice
                f32x4 aq[MT][NTQ];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int u = 0; u < NTQ; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) aq[mt][u][r] = 0.f;
                for (int sp = 0; sp < nsplit; ++sp) {
                    unsigned long long t[MT][NTQ][2];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int u = 0; u < NTQ; ++u) {
                            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(
                                pbase + (size_t)sp * tstride + (size_t)(mt * NT + split * NTQ + u) * 256);
                            t[mt][u][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            t[mt][u][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int u = 0; u < NTQ; ++u) {
                            aq[mt][u][0] += __uint_as_float((unsigned)t[mt][u][0]);
                            aq[mt][u][1] += __uint_as_float((unsigned)(t[mt][u][0] >> 32));
                            aq[mt][u][2] += __uint_as_float((unsigned)t[mt][u][1]);
                            aq[mt][u][3] += __uint_as_float((unsigned)(t[mt][u][1] >> 32));
                        }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int u = 0; u < NTQ; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) aq[mt][u][r] *= p.w_unscale;
                constexpr int BNQ = 16 * NTQ;
                const int n0q = n0 + split * BNQ;
                conv_epilogue16<BM, BNQ, WGM, 1, MT, NTQ>(p, aq, smem, rowpix, rowimg, m0, n0q, M, wm, 0, l16, q, colb + split * BNQ, one_img);
                if (p.stats != nullptr) {
                    // (the epilogue ended with a barrier behind the staged column sums.) A tile may cover several whole
                    // images (8x8 level: HWo = 64 < BM): the row groups of image i of the tile are entries
                    // [i * per, (i + 1) * per) of the WGM * 4 staged ones, one slice per image; else one slice per tile
                    const double2 *red = reinterpret_cast<const double2 *>(smem);
                    const int imgs = HWo < BM ? BM / HWo : 1, per = (WGM * 4) / imgs;
                    for (int e = threadIdx.x; e < imgs * BNQ; e += 256) {
                        const int im = e / BNQ, col = e - im * BNQ;
                        double a = 0, b = 0;
                        for (int j = 0; j < per; ++j) { const double2 t2 = red[(im * per + j) * BNQ + col]; a += t2.x; b += t2.y; }
                        const int m_img = m0 + im * (BM / imgs);
                        const int n = m_img / HWo, slc = p.stats_slice0 + (m_img - n * HWo) / BM;
                        double *o = p.stats + (((size_t)n * p.stats_slices + slc) * Cout + n0q + col) * 2;
                        o[0] = a; o[1] = b;
                    }
                }
            };
            if (nsplit == 2) slice(std::integral_constant<int, NT / 2>{});
            else slice(std::integral_constant<int, (NT / 4 > 0 ? NT / 4 : 1)>{});
            if (threadIdx.x == 0) {                 // departed; the last one leaves the counter at zero for the next launch
                const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == 2u * (unsigned)nsplit - 1u) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] *= p.w_unscale;
        conv_epilogue16<BM, BN, WGM, WGN, MT, NT, false, GNF>(p, acc, smem, rowpix, rowimg, m0, n0, M, wm, wn, l16, q, colbias,
                                                              reinterpret_cast<const int *>(colbias)[BN] != 0);
        return;
    }
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // B rows are plain: the swizzle term is a per-lane constant
    const int swzB = (li >> 1) & 7;
    const float *Bbase = Bring + (wn * WN + li) * ROWF;
    int hoffB[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) hoffB[sb] = (((2 * sb + lh) ^ swzB) & 7) * 4;
    // A rows: halo row of tile row r for dx = 0 is r + 2 * (r / SEG); plain row (in2 phase) is r.
    // (Precomputing per-(mi, dx) offsets and unrolling the three taps was measured slower: more
    // registers, lower occupancy for the 128x64 tile.)
    int rhalo[MI], rplain[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        rplain[mi] = wm * WM + mi * 32 + li;
        rhalo[mi] = rplain[mi] + HALO * (rplain[mi] / SEG);
    }
    // second halves of the rows (lo halfs, or fp8 operands in F8C K-steps): the two 16-byte reads of a K-step are
    // concatenated into the fp8 MFMA's 8-register operand (the compiler loads them into adjacent registers: no moves)
    h16x8 ah[2][MI], bh[2][NI];
    i32x4 alq[2][MI], blq[2][NI];
    // fragment reads of K-step KT, 16-wide K block SB into register set SET
#define SR3_HREAD(SET, KT, SB)                                                                     \
    {                                                                                              \
        const int kt_ = (KT);                                                                      \
        const bool halo_ = kt_ < nkh;                                                              \
        const int g_ = kt_ / KS;                                                                   \
        const int astage_ = halo_ ? (g_ & 1) : ((G + kt_ - nkh) & 1);                              \
        const int dx_ = kt_ - KS * g_;                                                             \
        const float *Ab_ = Aring + astage_ * ASTG;                                                 \
        const float *Bb_ = Bbase + (kt_ & 1) * BSTG;                                               \
        /* second half of a row: chunk 4 + 2 SB + lh. f16x3: the lo halfs of K block SB. F8C halo K-steps: the same two    \
           chunks ARE the lane's fp8 operand — v_mfma_scale_f32_32x32x64_f8f6f4 takes bytes 0..15 of a lane as K elements    \
           16 lh .. 16 lh + 15 of scale block 0 and bytes 16..31 as those of scale block 1 (measured: tools/f8_mfma_probe) */ \
        const int c2_ = 4 + 2 * (SB) + lh;                                                         \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                        \
            const int R_ = halo_ ? rhalo[mi] + dx_ : rplain[mi];                                   \
            const int sw_ = (R_ >> 1) & 7;                                                         \
            ah[SET][mi] = *reinterpret_cast<const h16x8 *>(Ab_ + R_ * ROWF + (((2 * (SB) + lh) ^ sw_) & 7) * 4);     \
            alq[SET][mi] = *reinterpret_cast<const i32x4 *>(Ab_ + R_ * ROWF + ((c2_ ^ sw_) & 7) * 4); \
        }                                                                                          \
        const int lob_ = ((c2_ ^ swzB) & 7) * 4;                                                   \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                        \
            bh[SET][ni] = *reinterpret_cast<const h16x8 *>(Bb_ + ni * 32 * ROWF + hoffB[SB]);      \
            blq[SET][ni] = *reinterpret_cast<const i32x4 *>(Bb_ + ni * 32 * ROWF + lob_);          \
        }                                                                                          \
    }
    // F8C (ConvParams::f8): in the halo K-steps the second half of a row holds fp8 operands — (xl8 | xh8) for A, (wh8 | wl8)
    // for B = scale blocks 0 | 1 of the instruction — and ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile and K-step
    // computes xl*wh + xh*wl (64 cycles instead of the 128 of four f16 MFMAs); the power-of-two operand scales (SR3_F8_*)
    // come back through the block-scale operands (scale block j is taken from lane half lh = j). Fused 1x1 K-steps stay f16x3.
#define SR3_HMMA(SET, F8K)                                                                         \
    {                                                                                              \
        if (F8C && (F8K)) {                                                                        \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                      \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[SET][mi], bh[SET][ni], acc[mi][ni], 0, 0, 0); \
            if ((SET) == 0) {                                                                      \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                  \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                  \
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                 \
                        __builtin_shufflevector(alq[0][mi], alq[1][mi], 0, 1, 2, 3, 4, 5, 6, 7),           \
                        __builtin_shufflevector(blq[0][ni], blq[1][ni], 0, 1, 2, 3, 4, 5, 6, 7), acc[mi][ni], 0, 0, 0, f8_scale_a, 0, f8_scale_b); \
            }                                                                                      \
        } else {                                                                                   \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                      \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                    \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, alq[SET][mi]), bh[SET][ni], acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[SET][mi], __builtin_bit_cast(h16x8, blq[SET][ni]), acc[mi][ni], 0, 0, 0); \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[SET][mi], bh[SET][ni], acc[mi][ni], 0, 0, 0); \
            }                                                                                      \
        }                                                                                          \
    }
    // E8M0 block scales of the fp8 product: scale block 0 (from lanes lh = 0) is xl8 * wh8, block 1 (lh = 1) is xh8 * wl8
    const int f8_scale_a = lh == 0 ? 127 - SR3_F8_XL : 127 - SR3_F8_XH, f8_scale_b = lh == 0 ? 127 - SR3_F8_WH : 127 - SR3_F8_WL;
    (void)f8_scale_a; (void)f8_scale_b;
    __syncthreads();
    SR3_HREAD(0, 0, 0)
    int kt = 0;
    if constexpr (F8C) {
        // halo K-steps: corrections on the fp8 path (issued with set 0). A loop of its own — a run-time choice between the
        // two MFMA sequences inside one loop made the compiler spill the accumulators at the merge points.
        for (; kt < nkh; ++kt) {
            SR3_HREAD(1, kt, 1)
            SR3_HMMA(0, true)
            __syncthreads();
            SR3_HREAD(0, min(kt + 1, nk - 1), 0)
            SR3_HMMA(1, true)
        }
    }
    for (; kt < nk; ++kt) {
        SR3_HREAD(1, kt, 1)
        SR3_HMMA(0, false)
        __syncthreads();                       // every read of tile kt has been issued and waited
        SR3_HREAD(0, min(kt + 1, nk - 1), 0)   // next tile (re-reads the last one at the end: unused)
        SR3_HMMA(1, false)
    }
#undef SR3_HREAD
#undef SR3_HMMA
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= p.w_unscale;
    if constexpr (F8C)
        conv_epilogue32<BM, BN, WGM, WGN, MI, NI>(p, acc, smem, rowpix, rowimg, wm, wn, li, lh, n0, colbias,
                                                  reinterpret_cast<const int *>(colbias)[BN] != 0);
    else
        conv_epilogue<BM, BN, WGM, WGN, MI, NI>(p, acc, smem, rowpix, rowimg, m0, n0, M, wm, wn, li, lh, 0, 1);
}

#ifdef SR3_EXPERIMENTS
// =================================================================================================
// EXPERIMENT (libsr3hip_exp.so only, SR3_PERSIST=1; results are correct — the conv / UNet / sampler GPU tests
// pass with it — but it is 1.4x SLOWER than the one-tile kernel: 0.35 vs 0.25 ms on the 128x128-pixel
// 64-channel conv, +1.0 ms per B=64 step; profiles/README.md finding 38). Kept as the measured record of
// the "hide the tile prologue behind the previous epilogue" idea.
// Persistent form of the x-halo kernel (16x16x32 consumers, four waves stacked along M): a block walks
// over tiles v = blockIdx.x, + gridDim.x, ... . What it buys: while the consumer waves run the epilogue
// of tile i, the producer waves set up tile i + 1 (tables, addresses) and put its first A halo group
// and first B tile in flight — the LDS rings are free once the last K-step of tile i has been read —
// so a tile's launch / set-up / first-DMA latency (10 % of a 128x128-pixel, 64-channel conv) hides
// behind the previous epilogue. What makes it fit the same registers and LDS as the one-tile kernel:
// the consumer state is tile independent (fragment row offsets depend on the lane only), the per-tile
// tables are double-buffered (1.3 KB), and the statistics hand-off is reduced over the four row groups
// of a wave by lane exchange, so it fits B stage 1 (free between two tiles) instead of a region of its own.
// Barriers per tile, identical on both sides: nk K-step barriers + F (end of K loop) + E (statistics
// staged; only with fused statistics). The consumers' first barrier of a tile is the producers' K-step-0
// barrier of that tile.
// =================================================================================================
template <int BM, int BN, int SEGMIN, int KS>
__global__ __launch_bounds__(512, (BN == 64) ? 6 : 4) void conv3x3_halo_pt(const ConvParams p_in, const int ntiles,
                                                                             const int stagger) {
    const ConvParams p = phase_params(p_in, blockIdx.z);
    // Resident blocks of a CU that start together and walk over equal tiles stay in lockstep: all in their K
    // loops, then all in their epilogues (MI355X_MICROARCH.md, "try a stagger"). The k-th wave of blocks
    // (blockIdx.x / stagger_group) starts k * stagger shader cycles late so the phases interleave.
    if (stagger > 0) {
        const int late = (blockIdx.x / (gridDim.x / 3 > 0 ? gridDim.x / 3 : 1)) * stagger;
        const long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < late) __builtin_amdgcn_s_sleep(16);
    }
    constexpr int WGM = 4, WGN = 1;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int HALO = KS - 1, TAPS = KS * KS;
    constexpr int RA = (BM + HALO * (BM / SEGMIN) + 7) / 8 * 8;
    constexpr int ARH = (RA / 8 + 3) / 4;
    constexpr int ASTG = RA * ROWF, BSTG = BN * ROWF;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int TBL = 2 * BM + BN + 4;            // ints / floats of the epilogue tables: rowpix | rowimg | colbias + flag
    static_assert((NT % 2) == 0 && (RA % 8) == 0, "tile shape");
    static_assert((size_t)WGM * BN * sizeof(double2) <= (size_t)BSTG * sizeof(float), "statistics staging fits one B stage");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Aring = smem;                    // [2][RA][32]
    float *Bring = smem + 2 * ASTG;         // [2][BN][32]
    float *tables = smem + 2 * ASTG + 2 * BSTG;     // [TBL], ONE set (a second one would cost the third block per CU: LDS is
                                                    // allocated in 1280-B granules): written by the producers right behind a
                                                    // tile's K-step-0 barrier, when every consumer has left the previous epilogue
    float *stat_stage = Bring + BSTG;       // B stage 1: free from the end of a tile's K loop to K-step 1 of the next

    const int C0 = p.in0.C, C1 = p.in1.p ? p.in1.C : 0;
    const int Cin = C0 + C1;
    const int Cout = p.out.C;
    const int W = p.Wout;
    const int HWo = p.Hout * W;
    const int M = p.B * HWo;
    const int tilesN = Cout / BN;
    const int SEG = min(W, BM), SEGP = SEG + HALO, nseg = BM / SEG;
    const int rows_a = nseg * SEGP;
    const int nkh = TAPS * (Cin / BK);
    const int C2a = p.in2.p ? p.in2.C : 0, C2 = C2a + (p.in2b.p ? p.in2b.C : 0);
    const int nk = nkh + C2 / BK;
    const int G = nkh / KS;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool has_stats = p.stats != nullptr;
    // XCD-aware order of the virtual tile ids (speed only): ids v and v + 8 run on one XCD
    auto tile_of = [&](int v, int &m0, int &n0) {
        const int xcd = v & 7, loc = v >> 3, qq = ntiles >> 3, rr = ntiles & 7;
        const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
        m0 = (bid / tilesN) * BM;
        n0 = (bid % tilesN) * BN;
    };

    if (wid >= 4) {
        // ------------------------------- producer waves -------------------------------------
        const int w = wid - 4;
        const int tid = threadIdx.x - 256;
        const int rsub = lane >> 3;
        const unsigned schunk16 = (unsigned)(((lane & 7) ^ ((((w & 1) << 2) | (lane >> 4)) & 7)) * 16);
        const int Hp = p.in0.Hp(), Wp = p.in0.Wp();
        const size_t tapstride = (size_t)Cout * Cin;
        unsigned vH0[ARH], vH1[ARH], vB[BR], vB2[BR], vA2[AR], vA2b[AR];
        int m0 = 0, n0 = 0;
        // epilogue tables of tile (tm0, tn0)
        auto setup_tables = [&](int tm0, int tn0) {
            int *rowpix = reinterpret_cast<int *>(tables);
            int *rowimg = rowpix + BM;
            float *colbias = reinterpret_cast<float *>(rowimg + BM);
            const int m0 = tm0, n0 = tn0;
            if (tid < BM) {
                const int m = min(m0 + tid, M - 1);
                const int n = div_hw(p, m, HWo);
                const int rem = m - n * HWo;
                const int oy = div_w(p, rem, W);
                rowpix[tid] = (int)p.out.pix(n, oy * p.out_step + p.out_oy, (rem - oy * W) * p.out_step + p.out_ox);
                rowimg[tid] = n;
            }
            {
                const int img_a = div_hw(p, m0, HWo), img_b = div_hw(p, min(m0 + BM - 1, M - 1), HWo);
                const bool one = img_a == img_b;
                if (tid < BN) {
                    const int nc = min(n0 + tid, Cout - 1);
                    float v = p.bias ? p.bias[nc] : 0.f;
                    if (p.chan_bias != nullptr && one) v += p.chan_bias[(size_t)img_a * p.chan_bias_stride + nc];
                    colbias[tid] = v;
                }
                if (tid == 0) reinterpret_cast<int *>(colbias)[BN] = one ? 1 : 0;
            }
        };
        // DMA addresses of tile (m0, n0)
        auto setup = [&]() {
            static_for<ARH>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int R = (4 * i + w) * 8 + rsub;
                int sg = R / SEGP, jx = R - sg * SEGP;
                if (sg >= nseg) { sg = nseg - 1; jx = 0; }
                const int m = m0 + sg * SEG;
                const int n = div_hw(p, m, HWo);
                const int rem = m - n * HWo;
                const int y = div_w(p, rem, W), x0 = rem - y * W;
                const unsigned pix = (unsigned)((n * Hp + y + p.org_y) * Wp + x0 + jx + p.org_x);
                vH0[i] = pix * (unsigned)C0 * 4u + schunk16;
                vH1[i] = pix * (unsigned)C1 * 4u + schunk16;
            });
            static_for<BR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int n = min(n0 + (4 * i + w) * 8 + rsub, Cout - 1);
                vB[i] = (unsigned)n * (unsigned)Cin * 4u + schunk16;
                vB2[i] = (unsigned)n * (unsigned)C2 * 4u + schunk16;
            });
            static_for<AR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int m = min(m0 + (4 * i + w) * 8 + rsub, M - 1);
                const int n = div_hw(p, m, HWo);
                const int rem = m - n * HWo;
                const int oy = div_w(p, rem, W);
                vA2[i] = C2 ? (unsigned)p.in2.pix(n, oy, rem - oy * W) * (unsigned)C2a * 4u + schunk16 : 0u;
                vA2b[i] = C2 > C2a ? (unsigned)p.in2b.pix(n, oy, rem - oy * W) * (unsigned)(C2 - C2a) * 4u + schunk16 : 0u;
            });
        };
        int ga = 0;
#define SR3_ISSUE_HALO(C0A, DY)                                                                    \
    {                                                                                              \
        const int c0a_ = (C0A);                                                                    \
        const bool first_ = c0a_ < C0;                                                             \
        const int Cs_ = first_ ? C0 : C1;                                                          \
        const char *ab_ = reinterpret_cast<const char *>((first_ ? p.in0.p : p.in1.p) + (first_ ? c0a_ : c0a_ - C0)) + \
                          (size_t)(DY) * Wp * Cs_ * 4;                                             \
        float *Ad_ = Aring + (ga & 1) * ASTG + w * 256;                                            \
        /* (a uniform branch instead of a per-lane select of the offset in front of every DMA) */   \
        if (first_) {                                                                              \
            static_for<ARH>([&](auto ic) {                                                         \
                constexpr int i = decltype(ic)::value;                                             \
                if ((4 * i + w) * 8 < rows_a) dma16s<true>(ab_, vH0[i], Ad_ + i * 1024);                 \
            });                                                                                    \
        } else {                                                                                   \
            static_for<ARH>([&](auto ic) {                                                         \
                constexpr int i = decltype(ic)::value;                                             \
                if ((4 * i + w) * 8 < rows_a) dma16s<true>(ab_, vH1[i], Ad_ + i * 1024);                 \
            });                                                                                    \
        }                                                                                          \
        ++ga;                                                                                      \
    }
        // first operands of a tile: halo group (chunk 0, dy 0) into A stage 0, B tile of K-step 0 into B stage 0
        auto issue_first = [&]() {
            ga = 0;
            SR3_ISSUE_HALO(0, 0)
            const char *wb = reinterpret_cast<const char *>(p.w);
            float *Bd = Bring + w * 256;
            static_for<BR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                dma16s<true>(wb, vB[i], Bd + i * 1024);
            });
        };
        int v = blockIdx.x;
        if (v < ntiles) {
            tile_of(v, m0, n0);
            setup();
            issue_first();
        }
        for (; v < ntiles; v += gridDim.x) {
            const int m0_cur = m0, n0_cur = n0;
            int k = 0;
            for (int c0 = 0; c0 < Cin; c0 += BK) {
                const char *wbase = reinterpret_cast<const char *>(p.w + c0);
                static_for<TAPS>([&](auto tc) {
                    constexpr int tap = decltype(tc)::value;
                    constexpr int dy = tap / KS, dx = tap % KS;
                    if (dx == 1) {          // the stage of group g-1 is free once K-step KS*g - 1 has been read
                        if (dy < KS - 1) {
                            SR3_ISSUE_HALO(c0, dy + 1)
                        } else if (c0 + BK < Cin) {
                            SR3_ISSUE_HALO(c0 + BK, 0)
                        }
                    }
                    if (k > 0) {            // (K-step 0's B tile went out with the tile's first operands)
                        float *Bd = Bring + (k & 1) * BSTG + w * 256;
                        const char *wb = wbase + (size_t)tap * tapstride * 4;
                        static_for<BR>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            dma16s<true>(wb, vB[i], Bd + i * 1024);
                        });
                    }
                    producer_sync<0>();
                    if (k == 0) setup_tables(m0_cur, n0_cur);
                    ++k;
                });
            }
            // fused 1x1 term: plain BM-row A tiles continue in the A ring
            for (int c0 = 0; c0 < C2; c0 += BK) {
                float *Ad = Aring + (ga & 1) * ASTG + w * 256;
                float *Bd = Bring + (k & 1) * BSTG + w * 256;
                const bool first2 = c0 < C2a;
                const char *ab = reinterpret_cast<const char *>(first2 ? p.in2.p + c0 : p.in2b.p + (c0 - C2a));
                const char *wb = reinterpret_cast<const char *>(p.w2 + c0);
                if (first2) {
                    static_for<AR>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        dma16s<true>(ab, vA2[i], Ad + i * 1024);
                    });
                } else {
                    static_for<AR>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        dma16s<true>(ab, vA2b[i], Ad + i * 1024);
                    });
                }
                static_for<BR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    dma16s<true>(wb, vB2[i], Bd + i * 1024);
                });
                producer_sync<0>();
                ++k;
                ++ga;
            }
            producer_sync<0>();             // F: the consumers have read the last K-step
            // next tile: tables, addresses and first operands while the consumers run this tile's epilogue
            const int vn = v + gridDim.x;
            if (vn < ntiles) {
                tile_of(vn, m0, n0);
                setup();
                issue_first();
            }
            if (has_stats) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();               // E: this tile's column sums are staged (not a DMA wait)
                const double2 *red = reinterpret_cast<const double2 *>(stat_stage);
                if (tid < BN) {
                    double a = 0, b = 0;
#pragma unroll
                    for (int j = 0; j < WGM; ++j) { const double2 t2 = red[j * BN + tid]; a += t2.x; b += t2.y; }
                    const int n = m0_cur / HWo, slice = p.stats_slice0 + (m0_cur - n * HWo) / BM;
                    double *o = p.stats + (((size_t)n * p.stats_slices + slice) * Cout + n0_cur + tid) * 2;
                    o[0] = a; o[1] = b;
                }
            }
        }
#undef SR3_ISSUE_HALO
        return;
    }

    // ----------------------------------- consumer waves -----------------------------------------
    const int l16 = lane & 15, q = lane >> 4;
    const int wm = wid, wn = 0;
    const int swzB = (l16 >> 1) & 7;
    const float *Bbase = Bring + (wn * WN + l16) * ROWF;
    const int bho = ((q ^ swzB) & 7) * 4, blo = (((4 + q) ^ swzB) & 7) * 4;
    static_assert(SEGMIN % WM == 0, "a wave's rows lie in one row segment");
    const int rplain0 = wm * WM + l16;
    const int rhalo0 = rplain0 + HALO * (rplain0 / SEG);
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        int m0, n0;
        tile_of(v, m0, n0);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
        h16x8 ah[MT], al[MT], bqh[2], bql[2];
        const float *aptr_h, *aptr_l;
        // (SEGMIN >= WM: the WM rows of a wave lie in one row segment, so row tile mt is tile 0 + 16 * mt ring
        // rows with the same swizzle term — one address pair per K-step, the other tiles are constant offsets)
#define SR3_AADDR(MTI, KT)                                                                         \
    if ((MTI) == 0) {                                                                              \
        const int kt_ = (KT);                                                                      \
        const bool halo_ = kt_ < nkh;                                                              \
        const int g_ = kt_ / KS;                                                                   \
        const int astage_ = halo_ ? (g_ & 1) : ((G + kt_ - nkh) & 1);                              \
        const int R_ = halo_ ? rhalo0 + (kt_ - KS * g_) : rplain0;                                 \
        const int sw_ = (R_ >> 1) & 7;                                                             \
        const float *Ar_ = Aring + astage_ * ASTG + R_ * ROWF;                                     \
        aptr_h = Ar_ + ((q ^ sw_) & 7) * 4;                                                        \
        aptr_l = Ar_ + (((4 + q) ^ sw_) & 7) * 4;                                                  \
    }
#define SR3_AREAD2(MTI)                                                                            \
    {                                                                                              \
        ah[MTI] = *reinterpret_cast<const h16x8 *>(aptr_h + (MTI) * 16 * ROWF);                    \
        al[MTI] = *reinterpret_cast<const h16x8 *>(aptr_l + (MTI) * 16 * ROWF);                    \
    }
#define SR3_BREAD(BUF, NTI, KT)                                                                    \
    {                                                                                              \
        const float *Bb_ = Bbase + ((KT) & 1) * BSTG + (NTI) * 16 * ROWF;                          \
        bqh[BUF] = *reinterpret_cast<const h16x8 *>(Bb_ + bho);                                    \
        bql[BUF] = *reinterpret_cast<const h16x8 *>(Bb_ + blo);                                    \
    }
#define SR3_MMA16(MTI, NTI, BUF)                                                                   \
    {                                                                                              \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[MTI], bqh[BUF], acc[MTI][NTI], 0, 0, 0); \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[MTI], bql[BUF], acc[MTI][NTI], 0, 0, 0); \
        acc[MTI][NTI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[MTI], bqh[BUF], acc[MTI][NTI], 0, 0, 0); \
    }
        __syncthreads();                            // K-step 0 of this tile has landed
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { SR3_AADDR(mt, 0) SR3_AREAD2(mt) }
        SR3_BREAD(0, 0, 0)
        for (int kt = 0; kt < nk; ++kt) {
            const int kn = min(kt + 1, nk - 1);     // after the last K-step: re-read, unused
#pragma unroll
            for (int nt = 0; nt < NT - 1; ++nt) {
                SR3_BREAD((nt + 1) & 1, nt + 1, kt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) SR3_MMA16(mt, nt, nt & 1)
                if (nt == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) SR3_AADDR(mt, kn)
                }
            }
            __syncthreads();
            SR3_BREAD(0, 0, kn)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                SR3_MMA16(mt, NT - 1, (NT - 1) & 1)
                __builtin_amdgcn_sched_barrier(0);
                SR3_AREAD2(mt)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef SR3_AADDR
#undef SR3_AREAD2
#undef SR3_BREAD
#undef SR3_MMA16
        // The epilogue re-reads its parameters from the kernel-argument segment through a pointer the optimiser
        // cannot see through: otherwise ~15 pointers / scalars are hoisted out of the tile loop, stay live across
        // the K loop, and the 80-register budget of this tile shape spills inside the K loop. (KS == 3: no
        // sub-pixel phases, so the phase-adjusted copy equals the argument itself.)
        typedef const unsigned __attribute__((address_space(4))) *KArgW;
        typedef unsigned __attribute__((may_alias)) AliasWord;      // (the copy is read back as floats and pointers)
        KArgW pq = (KArgW)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pq));
        ConvParams pl;
        static_assert(sizeof(ConvParams) % 4 == 0, "word copy");
#pragma unroll
        for (int i = 0; i < (int)(sizeof(ConvParams) / 4); ++i) reinterpret_cast<AliasWord *>(&pl)[i] = pq[i];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] *= pl.w_unscale;
        const int *rowpix = reinterpret_cast<const int *>(tables);
        const int *rowimg = rowpix + BM;
        const float *colbias = reinterpret_cast<const float *>(rowimg + BM);
        // (same for the lane-derived offsets of the epilogue: made opaque per tile so they are recomputed here
        // instead of being carried through the K loop)
        int l16e = l16, qe = q;
        asm volatile("" : "+v"(l16e), "+v"(qe));
        conv_epilogue16<BM, BN, WGM, WGN, MT, NT, true>(pl, acc, stat_stage, rowpix, rowimg, m0, n0, M, wm, wn, l16e, qe, colbias,
                                                        reinterpret_cast<const int *>(colbias)[BN] != 0);
        if (has_stats) __syncthreads();             // E: column sums staged for the producer threads
    }
}

template <int BM, int BN, int SEGMIN, int KS>
void launch_halo_pt(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    static int cus = 0;
    constexpr int RA = (BM + (KS - 1) * (BM / SEGMIN) + 7) / 8 * 8;
    constexpr size_t lds = ((size_t)2 * RA * ROWF + 2 * BN * ROWF + 2 * BM + BN + 4) * sizeof(float);
    auto kern = conv3x3_halo_pt<BM, BN, SEGMIN, KS>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    const int ntiles = (M / BM) * (p.out.C / BN);
    // resident blocks per CU: LDS comes in 1280-byte granules (160 KiB = 128 of them)
    const int granules = ((int)lds + 1279) / 1280;
    const int per_cu = std::min(128 / granules, BN == 64 ? 3 : 2);
    int grid = cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    static const int stagger = exp_int("SR3_PT_STAGGER", 0);
    hipLaunchKernelGGL(kern, dim3(grid, 1, p.phases), dim3(512), lds, s, p, ntiles, stagger);
}

#endif  // SR3_EXPERIMENTS

template <int BM, int BN, int WGM, int WGN, int SEGMIN, int KS, int MS, bool GNF = false, bool SPK = false, bool F8C = false>
void launch_halo(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    constexpr int RA = (BM + (KS - 1) * (BM / SEGMIN) + 7) / 8 * 8;
    constexpr size_t lds = ((size_t)2 * RA * ROWF + 2 * BN * ROWF + 2 * BM + BN + 4) * sizeof(float);
    // (the GroupNorm hand-off stages its sums, scale / shift and the fold scratch in the rings, free after the K loop)
    static_assert(!GNF || ((size_t)WGM * 4 * BN * 16 + BN * 8 + 4096 + 16 <= ((size_t)2 * RA * ROWF + 2 * BN * ROWF) * sizeof(float)), "GNF staging fits the rings");
    auto kern = conv3x3_halo_h3<BM, BN, WGM, WGN, SEGMIN, KS, MS, GNF, SPK, F8C>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    int grid = (M / BM) * ((p.out.C + BN - 1) / BN);
    if (GNF && p.gnf_band == 0) grid = (p.B + 7) / 8 * 8 * ((p.Hout * p.Wout) / BM) * (p.out.C / BN);   // whole rounds of eight images
    if (SPK) grid = (grid + 7) / 8 * 8 * p.splits;     // in-place split-K: see the kernel's block order
    hipLaunchKernelGGL(kern, dim3(grid, 1, p.phases), dim3(512), lds, s, p);
}

// consumer MFMA shape of the halo kernels: 16x16x32 with the four consumer waves stacked along M
// (wave tile 32 x BN: all A fragments of a K-step are 4 registers x 4, the B columns stream through
// two buffers); SR3_MFMA16=0 selects the 32x32x16 consumers (2 x 2 wave grid) for A/B measurements
static bool halo_mfma16() {
    static const int v = exp_int("SR3_MFMA16", 1);
    return v != 0;
}

#ifdef SR3_EXPERIMENTS
// persistent-tile form of the halo kernel (conv3x3_halo_pt, experiment): SR3_PERSIST=1
static bool halo_persistent() {
    static const int v = exp_int("SR3_PERSIST", 0);
    return v != 0;
}
#endif

// SR3_NO_HALO=1 (product safety switch): no x-halo kernel anywhere — and therefore no F8C path and no halo split-K
static bool halo_off() {
    static const int off = env_int("SR3_NO_HALO", 0);
    return off != 0;
}

// preconditions of the x-halo kernel for tile height BM
static bool halo_ok(const ConvParams &p, int BM, int segmin, int bn, bool split_ok = false) {
    if (halo_off() || p.prec != 1 || (p.ks != 3 && p.ks != 2) || p.stride != 1 || p.up2 || (p.splits > 1 && !split_ok) || p.in0.pad != 1) return false;
    const int W = p.Wout;
    if (p.in0.W != W || p.in0.H != p.Hout) return false;
    const int seg = W < BM ? W : BM;
    if (seg < segmin || (W % seg) || (BM % seg)) return false;
    const long M = (long)p.B * p.Hout * W;
    // no masking anywhere in the halo kernels: whole tiles in M and (for the tile width bn the caller picks) in N
    return (M % BM) == 0 && (p.out.C % bn) == 0 && (!p.in2.p || (p.in2.C % 32) == 0) && (!p.in2b.p || (p.in2b.C % 32) == 0);
}

template <int BM, int BN, int WGM, int WGN, int KS, int PREC, int NS>
void launch_inst2(const ConvParams &p, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t lds = ((size_t)NS * (BM + BN) * ROWF + 2 * BM) * sizeof(float);
    auto kern = conv_igemm_dma_f32<BM, BN, WGM, WGN, KS, PREC, NS>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = p.B * p.Hout * p.Wout;
    const int tilesM = (M + BM - 1) / BM, tilesN = (p.out.C + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN, p.splits > 1 ? p.splits : 1, p.phases), dim3(512), lds, s, p);
}

template <int BM, int BN, int WGM, int WGN, int KS, int PREC>
void launch_inst(const ConvParams &p, hipStream_t s) {
    // stages per tile shape (A/B in profiles/README.md): the 64x64 tile is used where few blocks
    // are resident (small M), so it gets a 4-deep ring (2 blocks/CU); the larger tiles run 2-3
    // blocks per CU with 2 stages (3 stages x 2 blocks measured the same or slower)
    constexpr int NS = (BM + BN) <= 128 ? 4 : 2;
    launch_inst2<BM, BN, WGM, WGN, KS, PREC, NS>(p, s);
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const ConvParams &p, hipStream_t s) {
    if (p.prec == 0) {
        if (p.ks == 1) launch_inst<BM, BN, WGM, WGN, 1, 0>(p, s);
        else if (p.ks == 2) launch_inst<BM, BN, WGM, WGN, 2, 0>(p, s);
        else launch_inst<BM, BN, WGM, WGN, 3, 0>(p, s);
    } else {
        if (p.ks == 1) launch_inst<BM, BN, WGM, WGN, 1, 1>(p, s);
        else if (p.ks == 2) launch_inst<BM, BN, WGM, WGN, 2, 1>(p, s);
        else launch_inst<BM, BN, WGM, WGN, 3, 1>(p, s);
    }
}

} // namespace

// Preconditions (checked by the callers in sr3_api.hip): channels multiples of 32, 3x3 inputs
// zero-bordered (pad 1), up2 only with ks 3 / stride 1 / single input, every tensor < 4 GiB
// (32-bit byte offsets in the DMA addressing).
// split-K second pass: out = sum_s part[s] + bias + FeatureWiseAffine bias + residual, plus the fused
// GroupNorm statistics of the result (a split conv cannot produce them itself: its blocks hold partial
// sums). One block = TP consecutive pixels of one image x all channels: thread (pixel lane, channel quad)
// walks the pixels of its lane; per-channel fp64 sums go through LDS to one statistics slice per block
// (ConvParams::stats layout, stats_slices = splitk_stats_slices(HWo) per phase).
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p_in, int M, int HWo, int TP) {
    __shared__ double2 red[256][4];
    const ConvParams p = phase_params(p_in, blockIdx.y);
    const int Cout = p.out.C, cq = Cout >> 2;
    const int cqs = min(cq, 256), lanes_p = 256 / cqs;
    const int t = threadIdx.x, pl = t / cqs;
    const int m_lo = blockIdx.x * TP;
    const int img = m_lo / HWo;
    double st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0};
    unsigned range_bits = 0;
    for (int qb = t - pl * cqs; qb < cq && pl < lanes_p; qb += cqs) {     // one pass unless Cout > 1024
        const int n = qb << 2;
        for (int pp = pl; pp < TP; pp += lanes_p) {
            const int m = m_lo + pp;
            if (m >= M) break;
            // four partials in flight per thread (a plain loop waits for each load); summed in split order
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *pm = p.part + (size_t)m * Cout + n;
            const size_t sstride = (size_t)M * Cout;
            int sp = 0;
            for (; sp + 3 < p.splits; sp += 4) {
                float4 b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const float4 *>(pm + (size_t)(sp + u) * sstride);
#pragma unroll
                for (int u = 0; u < 4; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
            }
            for (; sp < p.splits; ++sp) {
                const float4 b = *reinterpret_cast<const float4 *>(pm + (size_t)sp * sstride);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            // (per pixel: a block of TP pixels straddles two images when HWo % TP != 0 — 14x14, 20x20 ... levels)
            const int im = m / HWo;
            const int rem = m - im * HWo, oy = rem / p.Wout;
            const size_t o = p.out.pix(im, oy * p.out_step + p.out_oy, (rem - oy * p.Wout) * p.out_step + p.out_ox) * Cout + n;
            float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (p.bias) v[j] += p.bias[n + j];
                if (p.chan_bias) v[j] += p.chan_bias[(size_t)im * p.chan_bias_stride + n + j];
                if (p.resid.p) v[j] += p.resid_split ? load_split(p.resid.p, (unsigned)(o + j)) : p.resid.p[o + j];
                if (p.out_f32) p.out.p[o + j] = v[j];
                if (p.out_split.p != nullptr) {
                    const _Float16 hi = (_Float16)v[j];         // detected, not clamped (range_bits)
                    const unsigned e = (unsigned)__builtin_bit_cast(unsigned short, hi) & 0x7C00u;
                    range_bits = e > range_bits ? e : range_bits;
                    _Float16 *hd = reinterpret_cast<_Float16 *>(p.out_split.p + ((o + j) & ~(size_t)31)) + ((o + j) & 31);
                    hd[0] = hi;
                    hd[32] = (_Float16)(v[j] - (float)hi);
                }
                st1[j] += (double)v[j];
                st2[j] = fma((double)v[j], (double)v[j], st2[j]);
            }
        }
        if (p.stats != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[t][j] = make_double2(st1[j], st2[j]);
        }
    }
    if (p.ovf != nullptr && split_range_overflow(range_bits)) *p.ovf = 1;
    if (p.stats != nullptr && cq <= 256) {
        __syncthreads();
        // thread (channel quad q, j) adds the pixel lanes of channel 4q + j
        for (int i = t; i < cq * 4; i += 256) {
            const int q = i >> 2, j = i & 3;
            double sa = 0, sb = 0;
            for (int l = 0; l < lanes_p; ++l) { const double2 v = red[l * cqs + q][j]; sa += v.x; sb += v.y; }
            const int slice = p.stats_slice0 + (m_lo - img * HWo) / TP;
            double *o = p.stats + (((size_t)img * p.stats_slices + slice) * Cout + i) * 2;
            o[0] = sa; o[1] = sb;
        }
    }
}

// pixels per block of the split-K reduce (at most 64 statistics slices per image and phase)
int splitk_reduce_tp(int HWo) { const int tp = HWo / 64; return tp < 1 ? 1 : tp; }
// 0: the reduce pass cannot produce the statistics of this shape (the caller's statistics kernel runs)
int splitk_stats_slices(int HWo, int Cout) {
    const int tp = splitk_reduce_tp(HWo);
    return ((HWo % tp) == 0 && Cout <= 1024 && (Cout & 3) == 0) ? HWo / tp : 0;
}

// tile choice: 0 = 128x32, 1 = 128x64, 2 = 64x64, 3 = 128x128
static int conv_tile_choice(long M, int Cout) {
    static const int force = exp_int("SR3_CONV_TILE", -1);   // experiments build only
    if (force >= 0) return force;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
    const long want = 512;  // 256 CUs x 2 resident blocks
    if (Cout <= 32) return 0;
    if (Cout <= 64 || (Cout % 128) != 0) return blocks(128, 64) >= want ? 1 : 2;
    if (blocks(128, 128) >= want) return 3;
    return blocks(128, 64) >= want ? 1 : 2;
}

int conv_tile_m(long M, int Cout) { return conv_tile_choice(M, Cout) == 2 ? 64 : 128; }

int conv_splits(long M, int Cout, int Cin) {
    static const int off = exp_int("SR3_NO_SPLITK", 0);
    if (off || (Cout & 3)) return 1;
    if (const int f = exp_int("SR3_FORCE_SPLITS", 0)) return f;
    static const int bm[4] = {128, 128, 64, 128}, bn[4] = {32, 64, 64, 128};
    const int t = conv_tile_choice(M, Cout);
    const long tiles = ((M + bm[t] - 1) / bm[t]) * ((Cout + bn[t] - 1) / bn[t]);
    const int nchunk = Cin / BK;
    // (tuned on B = 1 at 128x128 and config 1, with the in-place fix-up: profiles/README.md finding 42)
    const int tmin = exp_int("SR3_SPLIT_TMIN", 256), target = exp_int("SR3_SPLIT_TARGET", 256), chmin = exp_int("SR3_SPLIT_CHMIN", 2);
    if (tiles >= tmin || nchunk < 2 * chmin) return 1;
    int s = 2;
    while (s * 2 <= nchunk / chmin && tiles * s * 2 <= target && s < 16) s *= 2;
    return s;
}

// Deep-K 3x3 / stride-1 split-f16 convs over FEW output tiles (the 8x8 level at B = 64: M = 4096 pixels, 512 channels):
// instead of 64x64 tiles of the generic kernel (LDS traffic per MFMA twice that of a 128x128 tile) they run on the
// 128x128 x-halo tile with the K range split over `return value` blocks per tile, added IN PLACE by the block that
// arrives last (conv3x3_halo_h3, end of the consumer path) — no reduce pass, statistics in the unsplit layout (one
// slice per 128-row tile, or one per image where a tile covers several whole images). 0 or 1: not for this shape.
// Assumes ks 3, stride 1, prec 1, pad 1 (launch_conv checks them); sr3_api.hip sizes the partial buffer with it.
int conv_halo_splits(long M, int H, int W, int Cout, int Cin) {
    static const int force = env_int("SR3_HALO_SPLITS", -1);    // product switch: 0 off, 2 | 4 forced
    if (force == 0 || halo_off() || !halo_mfma16()) return 0;
    const int HWo = H * W;
    if (W < 8 || (M % 128) || (Cout % 128) || (Cin % 32)) return 0;
    const int seg = W < 128 ? W : 128;
    if ((W % seg) || (128 % seg)) return 0;
    if (HWo >= 128 ? (HWo % 128) != 0 : ((128 % HWo) != 0 || (HWo % 32) != 0)) return 0;
    if (conv_tile_choice(M, Cout) != 2) return 0;                  // enough 128-row tiles already: no split needed
    const long tiles = (M / 128) * (Cout / 128);
    const int nchunk = Cin / 32;
    if (tiles < 64 || tiles > CONV_TILE_COUNTERS || nchunk < 8) return 0;
    int sp = 2;
    while (sp * 2 <= nchunk / 4 && tiles * sp * 2 <= 512 && sp < 4) sp *= 2;      // (the reduce-scatter tail handles 2 or 4 splits)
    if (force == 2 || force == 4) sp = force;
    return sp;
}

// F8C ("f16f8" mode): 3x3 / stride-1 split-f16 convs that run on the 128x128 x-halo tile without split-K at the 32x32- and
// 16x16-pixel levels — the MFMA-bound shapes, where the 32x32 consumers with the corrections on the fp8 path measured
// 14-20 % faster than the f16x3 kernel (profiles/README.md finding 64); the 64x64- and 128x128-pixel levels are bound by
// operand movement and gain nothing. Mirrors launch_conv's choice: callers format the conv's input accordingly.
bool conv_f8_supported(int B, int H, int W, int Cout, int Cin) {
    static const int off = exp_int("SR3_NO_F8C", 0);
    static const int maxhw = exp_int("SR3_F8C_MAX_HW", 1024);
    if (halo_off()) return false;            // (the F8C consumers exist in the x-halo kernel only)
    const long M = (long)B * H * W;
    const int HWo = H * W;
    if (off || (Cin % 32) || (Cout % 128) || (M % 128) || HWo > maxhw || HWo < 128 || (HWo % 128)) return false;
    const int seg = W < 128 ? W : 128;
    if (seg < 8 || (W % seg) || (128 % seg)) return false;
    if (conv_tile_choice(M, Cout) != 3) return false;
    return conv_halo_splits(M, H, W, Cout, Cin) <= 1;
}

bool conv_split_inplace(long M, int HWo, int Cout, int Cin, int phases) {
    static const int off = env_int("SR3_NO_INPLACE_SPLIT", 0);
    if (off || conv_splits(M, Cout, Cin) <= 1) return false;
    static const int bm[4] = {128, 128, 64, 128}, bn[4] = {32, 64, 64, 128};
    const int t = conv_tile_choice(M, Cout);
    const long tiles = ((M + bm[t] - 1) / bm[t]) * ((Cout + bn[t] - 1) / bn[t]);
    // the 64x64-tile kernel only; whole tiles per image (the statistics slices are per M-tile of an image) and in N
    // (nothing is masked in the fix-up), and a counter for every tile
    return t == 2 && (HWo % bm[t]) == 0 && (Cout % bn[t]) == 0 && tiles * phases <= CONV_TILE_COUNTERS;
}

// Producer-side GroupNorm (ConvParams::gnf_*): which halo kernel would run it — 0 none, 1 the 128x64 tile, 2 the 128x128
// tile with row segments of 32+ pixels, 3 the 128x128 tile with shorter segments.
// EXPERIMENT (libsr3hip_exp.so, SR3_GNF=1 [SR3_GNF_MAX_HW=pixels]): correct (all sampler / UNet / full-size tests pass with
// it) but it does not pay — the cross-CU hand-off costs a block about what the saved apply pass costs the chip
// (profiles/README.md finding 54), and its liveness rests on the observed dispatch order. The product library never
// takes this path.
static int gnf_kernel_choice(const ConvParams &p, int groups) {
#ifndef SR3_EXPERIMENTS
    (void)p; (void)groups;
    return 0;
#else
    static const int on = exp_int("SR3_GNF", 0);
    if (!on || !halo_mfma16() || p.prec != 1 || p.ks != 3 || p.stride != 1 || p.up2 || p.phases > 1 || groups <= 0) return 0;
    if (p.resid.p != nullptr || p.in2.p != nullptr || p.stats == nullptr) return 0;
    const int Cout = p.out.C, HWo = p.Hout * p.Wout;
    const long M = (long)p.B * HWo;
    if ((Cout % groups) != 0 || (HWo % 128) != 0) return 0;             // whole 128-row tiles inside one image
    const int cg = Cout / groups;
    const int Cin = p.in0.C + (p.in1.p ? p.in1.C : 0);
    if (conv_splits(M, Cout, Cin) > 1) return 0;
    int which = 0, bn = 0;
    switch (conv_tile_choice(M, Cout)) {
    case 1: if (halo_ok(p, 128, 32, 64)) { which = 1; bn = 64; } break;
    case 3:
        if (halo_ok(p, 128, 32, 128)) { which = 2; bn = 128; }
        else if (halo_ok(p, 128, 8, 128)) { which = 3; bn = 128; }
        break;
    default: break;
    }
    if (!which || (bn % cg) != 0 || p.stats_slices != HWo / 128) return 0;
    if ((long)p.B * (Cout / bn) > CONV_GNF_COUNTERS) return 0;
    static const int max_hw = exp_int("SR3_GNF_MAX_HW", 1 << 30);   // A/B: only images of at most this many pixels
    if (HWo > max_hw) return 0;
    // every (image, N-tile) group of TMI blocks must be able to be resident together: the standard block order keeps an
    // image's TMI x tilesN blocks on one XCD (32 CUs x 2 blocks at least); larger images use the band order, which
    // needs TMI to divide over the eight XCDs and a grid the round-robin deals evenly
    const int TMI = HWo / 128, tilesN = Cout / bn;
    if (TMI * tilesN > 32 && (TMI % 8) != 0) return 0;
    if (TMI * tilesN > 32 && (TMI / 8) * tilesN > 32) return 0;
    return which;
#endif
}
static int gnf_band_for(const ConvParams &p, int which) {
    const int HWo = p.Hout * p.Wout, TMI = HWo / 128, tilesN = p.out.C / (which == 1 ? 64 : 128);
    return TMI * tilesN > 32 ? TMI / 8 : 0;
}
bool conv_gnf_supported(const ConvParams &p, int groups) { return gnf_kernel_choice(p, groups) != 0; }

// first launch request of this thread that could not be honoured (nothing was launched for it); the API entry points
// turn it into an error return (conv_take_error)
static thread_local const char *g_conv_error = nullptr;
const char *conv_take_error() { const char *e = g_conv_error; g_conv_error = nullptr; return e; }

void launch_conv(const ConvParams &p_in, hipStream_t s) {
    if (p_in.up2) { launch_conv_up2(p_in, s); return; }     // weights must be in phase form (make_up2_phase_weights)
    ConvParams p = p_in;
    p.dbg = exp_int("SR3_CONV_DBG", 0);
    const long M = (long)p.B * p.Hout * p.Wout;
    {
        auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
        p.hw_shift = lg(p.Hout * p.Wout);
        p.w_shift = lg(p.Wout);
    }
    if (p.part == nullptr) p.splits = 1;
    if (p.in_fm) {
        // the caller wrote this conv's input fragment-major (it asked conv_ws_shape_ok): 64 -> 64 channels over many
        // 128-pixel tiles — the weights-stationary persistent kernel, and no other kernel can read that layout
        p.splits = 1;
        if (conv_ws_supported(p)) launch_conv_ws(p, s);
        else g_conv_error = "internal: fragment-major input handed to a conv the weights-stationary kernel does not support";
        return;
    }
    if (p.prec == 1 && p.ks == 3 && p.stride == 1 && !p.up2 && p.phases == 1 && p.part != nullptr && p.tile_cnt != nullptr &&
        p.in0.pad == 1 && p.in0.W == p.Wout && p.in0.H == p.Hout && p.gnf_gamma == nullptr && !p.no_halo_split) {
        const int hs = conv_halo_splits(M, p.Hout, p.Wout, p.out.C, p.in0.C + (p.in1.p ? p.in1.C : 0));
        const int HWo = p.Hout * p.Wout;
        const bool stats_ok = p.stats == nullptr || p.stats_slices == (HWo >= 128 ? HWo / 128 : 1);
        if (hs > 1 && stats_ok) {
            p.splits = hs;
            if (halo_ok(p, 128, 32, 128, true)) { launch_halo<128, 128, 4, 1, 32, 3, 16, false, true>(p, s); return; }
            if (halo_ok(p, 128, 8, 128, true)) { launch_halo<128, 128, 4, 1, 8, 3, 16, false, true>(p, s); return; }
            p.splits = p_in.splits;
        }
    }
    const bool inplace = p.splits > 1 && p.tile_cnt != nullptr &&
                         conv_split_inplace(M, p.Hout * p.Wout, p.out.C, p.in0.C + (p.in1.p ? p.in1.C : 0), p.phases);
    if (!inplace) p.tile_cnt = nullptr;
    // split-K, two-kernel form: the blocks hold partial sums, the statistics come out of the reduce pass (one slice
    // per TP pixels); they need whole tiles per image and at most 1024 channels, else the caller's statistics kernel runs
    double *reduce_stats = nullptr;
    if (p.splits > 1 && !inplace) {
        const int HWo = p.Hout * p.Wout;
        if (p.stats && splitk_stats_slices(HWo, p.out.C) > 0) reduce_stats = p.stats;
        p.stats = nullptr;
    }
#ifdef SR3_EXPERIMENTS
    if (p.gnf_gamma != nullptr) {
        // producer-side GroupNorm: the caller asked conv_gnf_supported() first
        const int which = gnf_kernel_choice(p, p.gnf_groups);
        if (which == 0) { g_conv_error = "internal: producer-side GroupNorm requested for an unsupported conv"; return; }
        p.gnf_band = gnf_band_for(p, which);
        if (which == 1) launch_halo<128, 64, 4, 1, 32, 3, 16, true>(p, s);
        else if (which == 2) launch_halo<128, 128, 4, 1, 32, 3, 16, true>(p, s);
        else launch_halo<128, 128, 4, 1, 8, 3, 16, true>(p, s);
        return;
    }
#endif
    if (p.f8) {
        // the caller asked conv_f8_supported() first and wrote the input / passes the weights in the F8C format
        // (conv_f8_supported is the single source of truth; a mismatch is a library bug, reported through the API's
        // error path — nothing is launched, the process is never aborted)
        if (!(p.prec == 1 && p.ks == 3 && p.phases == 1 && p.splits <= 1 && halo_ok(p, 128, 8, 128))) {
            g_conv_error = "internal: fp8 correction products requested for a conv the F8C kernel does not support";
            return;
        }
        launch_halo<128, 128, 2, 2, 8, 3, 32, false, false, true>(p, s);
        return;
    }
    switch (conv_tile_choice(M, p.out.C)) {
    case 0: launch_cfg<128, 32, 4, 1>(p, s); break;
    case 1:
        if (halo_ok(p, 128, 32, 64)) {
#ifdef SR3_EXPERIMENTS
            if (halo_mfma16() && halo_persistent() && p.ks == 3) launch_halo_pt<128, 64, 32, 3>(p, s);
            else
#endif
            if (halo_mfma16()) { if (p.ks == 3) launch_halo<128, 64, 4, 1, 32, 3, 16>(p, s); else launch_halo<128, 64, 4, 1, 32, 2, 16>(p, s); }
            else { if (p.ks == 3) launch_halo<128, 64, 2, 2, 32, 3, 32>(p, s); else launch_halo<128, 64, 2, 2, 32, 2, 32>(p, s); }
        }
        else launch_cfg<128, 64, 2, 2>(p, s);
        break;
    case 2: launch_cfg<64, 64, 2, 2>(p, s); break;
    default:
        if (halo_mfma16() && halo_ok(p, 128, 32, 128)) {   // rows of 32+ pixels: one row segment per wave (ONESEG)
            if (p.ks == 3) launch_halo<128, 128, 4, 1, 32, 3, 16>(p, s); else launch_halo<128, 128, 4, 1, 32, 2, 16>(p, s);
        }
        else if (halo_ok(p, 128, 8, 128)) {
            if (halo_mfma16()) { if (p.ks == 3) launch_halo<128, 128, 4, 1, 8, 3, 16>(p, s); else launch_halo<128, 128, 4, 1, 8, 2, 16>(p, s); }
            else { if (p.ks == 3) launch_halo<128, 128, 2, 2, 8, 3, 32>(p, s); else launch_halo<128, 128, 2, 2, 8, 2, 32>(p, s); }
        }
        else launch_cfg<128, 128, 2, 2>(p, s);
        break;
    }
    if (p.splits > 1 && !inplace) {
        const int HWo = p.Hout * p.Wout, TP = splitk_reduce_tp(HWo);
        p.stats = reduce_stats;
        if (reduce_stats) p.phase_slices = splitk_stats_slices(HWo, p.out.C);   // slices of one sub-pixel phase (phases > 1)
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((M + TP - 1) / TP), p.phases), dim3(256), 0, s, p,
                           (int)M, HWo, TP);
    }
}

void launch_conv_up2(const ConvParams &p_in, hipStream_t s) {
    const int H = p_in.Hout / 2, W = p_in.Wout / 2, Cout = p_in.out.C;
    const int Cin = p_in.in0.C + (p_in.in1.p ? p_in.in1.C : 0);
    const long Ml = (long)p_in.B * H * W;
    ConvParams p = p_in;            // the four phases are one launch: blockIdx.z picks (py, px)
    p.ks = 2; p.stride = 1; p.up2 = 0;
    p.Hout = H; p.Wout = W;
    p.out_step = 2;
    p.phases = 4;
    p.phase_w_stride = (size_t)4 * Cout * Cin;
    p.phase_slices = (H * W) / conv_tile_m(Ml, Cout);
    p.splits = p.part ? conv_splits(Ml, Cout, Cin) : 1;
    p.phase_part_stride = p.splits > 1 ? (size_t)p.splits * Ml * Cout : 0;
    launch_conv(p, s);
}

void make_up2_phase_weights(const float *w9, int Cout, int CinPad, float *dst) {
    // nearest x2 then 3x3: output row 2y+py reads upsampled rows 2y+py-1..2y+py+1, i.e. source
    // rows (2y+py+d-1)>>1 for d = 0..2: py = 0 -> {y-1, y, y}, py = 1 -> {y, y, y+1}. Window row
    // r2 (0|1) of the phase is source row y-1+py+r2; the same holds for columns.
    const size_t plane = (size_t)Cout * CinPad;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int r2 = 0; r2 < 2; ++r2)
                for (int c2 = 0; c2 < 2; ++c2) {
                    float *d = dst + ((size_t)(py * 2 + px) * 4 + r2 * 2 + c2) * plane;
                    for (size_t i = 0; i < plane; ++i) {
                        double acc = 0.0;
                        for (int dy = 0; dy < 3; ++dy) {
                            if (((py + dy + 1) >> 1) != py + r2) continue;     // source row offset (+1) of tap dy
                            for (int dx = 0; dx < 3; ++dx)
                                if (((px + dx + 1) >> 1) == px + c2) acc += (double)w9[(size_t)(dy * 3 + dx) * plane + i];
                        }
                        d[i] = (float)acc;
                    }
                }
}

namespace {
// one thread = 4 channels of a 32-channel chunk: split weights (32 hi halfs | 32 lo halfs) -> F8C weights
// (32 hi halfs | 32 x e4m3(hi * 2^SR3_F8_WH) | 32 x e4m3(lo * 2^SR3_F8_WL))
__global__ __launch_bounds__(256) void make_f8_weights_kernel(const float *__restrict__ split, float *__restrict__ dst, size_t chunks) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= chunks * 8) return;
    const size_t ch = t >> 3;
    const int j = (int)(t & 7) * 4;
    const _Float16 *src = reinterpret_cast<const _Float16 *>(split + ch * 32);
    _Float16 *dh = reinterpret_cast<_Float16 *>(dst + ch * 32);
    unsigned char *d8 = reinterpret_cast<unsigned char *>(dst + ch * 32) + 64;
    float h[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = (float)src[j + u]; l[u] = (float)src[32 + j + u]; dh[j + u] = src[j + u]; }
    const float sh = ldexpf(1.0f, SR3_F8_WH), sl = ldexpf(1.0f, SR3_F8_WL);
    auto cl = [](float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f); };
    int wh = __builtin_amdgcn_cvt_pk_fp8_f32(cl(h[0] * sh), cl(h[1] * sh), 0, false);
    wh = __builtin_amdgcn_cvt_pk_fp8_f32(cl(h[2] * sh), cl(h[3] * sh), wh, true);
    int wl = __builtin_amdgcn_cvt_pk_fp8_f32(cl(l[0] * sl), cl(l[1] * sl), 0, false);
    wl = __builtin_amdgcn_cvt_pk_fp8_f32(cl(l[2] * sl), cl(l[3] * sl), wl, true);
    *reinterpret_cast<int *>(d8 + j) = wh;
    *reinterpret_cast<int *>(d8 + 32 + j) = wl;
}
} // namespace

void launch_make_f8_weights(const float *split, float *dst, size_t chunks, hipStream_t s) {
    hipLaunchKernelGGL(make_f8_weights_kernel, dim3((unsigned)((chunks * 8 + 255) / 256)), dim3(256), 0, s, split, dst, chunks);
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

// fp32 packed rows [rows][CinPad] -> per 32-channel chunk: 32 hi halfs | 32 lo halfs of w * 2^k,
// k chosen so that max|w| * 2^k is in [1024, 2048): hi and lo are then normal fp16 numbers for all
// but vanishing weights and w = (hi + lo) * 2^-k to ~2^-22 relative. Returns 2^-k.
int split_scale_exponent(const float *packed, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(packed[i]));
    if (!(mx > 0.f)) return 0;
    int e;
    frexpf(mx, &e);              // mx = f * 2^e, f in [0.5, 1)
    return 11 - e;               // mx * 2^k in [1024, 2048)
}

float split_conv_weight(const float *packed, size_t rows, int CinPad, float *dst) {
    return split_conv_weight_k(packed, rows, CinPad, split_scale_exponent(packed, rows * (size_t)CinPad), dst);
}

float split_conv_weight_k(const float *packed, size_t rows, int CinPad, int k, float *dst) {
    const float sc = ldexpf(1.0f, k);
    _Float16 *d = reinterpret_cast<_Float16 *>(dst);
    for (size_t r = 0; r < rows; ++r)
        for (int c0 = 0; c0 < CinPad; c0 += 32) {
            const float *src = packed + r * CinPad + c0;
            _Float16 *o = d + (r * CinPad + c0) * 2;
            for (int j = 0; j < 32; ++j) {
                const float v = src[j] * sc;
                const _Float16 hi = (_Float16)v;
                o[j] = hi;
                o[32 + j] = (_Float16)(v - (float)hi);
            }
        }
    return ldexpf(1.0f, -k);
}

} // namespace sr3
