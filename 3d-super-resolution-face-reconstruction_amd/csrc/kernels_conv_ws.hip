// Weights-stationary 3x3 conv for the 64 -> 64 channel layers of the full-resolution level (split-f16 arithmetic).
//
// Replaces, for the seven 128x128-pixel Cin = Cout = 64 convs of a p_sample step (reference unet.py:80-110: block1 / block2
// of downs.1, downs.2 and block2 of ups.16-18 incl. their fused 1x1 res_conv / identity-skip K-steps), the x-halo kernel
// of kernels_conv.hip on its 128x64 tile.
//
// Why a kernel of its own (profiles/README.md, round 4): on the LDS-ring kernel these layers run at 0.11 of the f16 MFMA
// peak — K is 18 K-steps short, every one of the 8192 blocks re-fetches the whole 147 KB weight tensor through L2 into its
// rings (1.2 GB of L2 -> LDS traffic per launch against 0.27 GB of activations), every K-step ends in a workgroup barrier
// behind a one-K-step-deep DMA pipeline, and a third of a block's life is prologue + epilogue that its co-resident blocks
// do not hide. Here the roles are turned round:
//   * ONE persistent block per CU keeps the WHOLE weight tensor in LDS (18 K-steps x 64 rows x 128 B = 144 KB of the
//     160 KB) for the lifetime of the launch;
//   * the A operand never touches LDS: the 16x16x32 MFMA's A fragment of a lane is 16 contiguous bytes of one pixel's
//     32-channel chunk (hi halfs) + 16 more (lo halfs), so every wave loads its fragments straight from the zero-bordered
//     NHWC tensor into registers (buffer_load_dwordx4, voffset constant per 32-pixel sub-tile, tap / chunk as scalar and
//     immediate offsets — no vector address arithmetic in the K loop), DEPTH K-steps ahead of their use;
//   * a wave owns its 32 pixels x 64 channels completely: NO barrier and no shared pipeline state inside the launch (one
//     barrier after the weight load) — eight independent wave pipelines per CU, two per SIMD, which de-phase by
//     themselves: one wave's epilogue (VALU + stores) runs under the other's MFMAs;
//   * a wave processes whole 128-pixel tiles (four 32-pixel sub-tiles in sequence) and keeps the fused GroupNorm
//     statistics of the tile in fp64 registers: the statistics slice layout is the x-halo kernel's (one slice per
//     128-row tile), no LDS staging, no atomics, deterministic.
// Products, accumulation order inside a K-step and the K-step order (chunk-major, then dy, dx; fused 1x1 K-steps last) are
// those of the x-halo kernel's 16x16x32 consumers, so results agree with it to fp32 summation order.
#include "sr3_internal.h"

#include <type_traits>

namespace sr3 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_C = 64;                       // Cin == Cout == 64
constexpr int WS_KSTEPS = 18;                  // 9 taps x 2 chunks of 32 channels
constexpr int WS_LDS_BYTES = WS_KSTEPS * WS_C * 128;      // 147,456
constexpr int WS_WAVES = 8;
constexpr int WS_DEPTH = 6;                    // K-steps of A fragments in flight per wave (16 registers each)

template <int N, class F>
__device__ __forceinline__ void ws_static_for(F &&f) {
    if constexpr (N > 0) {
        ws_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ h16x8 ws_load16(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

struct WsFrag { h16x8 ah[2], al[2]; };

// One persistent block per CU, 8 waves. Tiles of 128 consecutive output pixels (whole tiles per image, W % 32 == 0).
// Wave g (XCD-major numbering, so that the tiles of an image stay in one XCD's L2) takes tiles [g * tpw, (g + 1) * tpw).
__global__ __launch_bounds__(512, 2) void conv3x3_ws64_kernel(const ConvParams p, const int tiles_total, const int tpw) {
    extern __shared__ __attribute__((aligned(128))) float wlds[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int l16 = lane & 15, q = lane >> 4;

    // ---- weights -> LDS, once: row R = kstep * 64 + cout (128 B = 32 hi halfs | 32 lo halfs of one 32-channel chunk),
    //      16-byte slot s of a row holds source chunk s ^ ((cout >> 1) & 7) (conflict-free ds_read_b128 of the fragments)
    {
        const char *wsrc = reinterpret_cast<const char *>(p.w);
#pragma unroll 2
        for (int i = 0; i < WS_LDS_BYTES / (512 * 16); ++i) {
            const int L = i * 512 + tid;
            const int R = L >> 3, s = L & 7;
            const int ks = R >> 6, o = R & 63;
            const int chunk = ks / 9, tap = ks - chunk * 9;      // K-step order of the consumers: chunk-major, then the tap
            const u32x4 v = *reinterpret_cast<const u32x4 *>(wsrc + ((size_t)(tap * WS_C + o) * WS_C + chunk * 32) * 4 +
                                                            ((s ^ ((o >> 1) & 7)) << 4));
            *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(wlds) + R * 128 + s * 16) = v;
        }
    }
    __syncthreads();            // the only barrier of the launch

    // ---- this wave's tiles
    const int nblk = (int)gridDim.x;
    const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int per_xcd = (nblk + 7) >> 3;
    const int g = (xcd * per_xcd + loc) * WS_WAVES + wid;
    int tile = g * tpw;
    const int tile_end = min(tile + tpw, tiles_total);
    if (tile >= tile_end) return;

    const int W = p.Wout, HWo = p.Hout * W;
    const int Hp = p.in0.Hp(), Wp = p.in0.Wp();
    const int C2a = p.in2.p ? p.in2.C : 0, C2 = C2a + (p.in2b.p ? p.in2b.C : 0);
    const int n2 = C2 >> 5;                                     // fused 1x1 K-steps
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in0.p), 0, -1, 0x00020000);
    // row offsets of the three dy taps (scalar): padded coordinates of input pixel (y - 1 + dy, x - 1 + dx) are (y + dy, x + dx)
    const unsigned s_dy[3] = {0u, (unsigned)Wp * 256u, (unsigned)Wp * 512u};

    // B fragment addresses in LDS: lane (l16, q) reads row (nt * 16 + l16) of the K-step's 64, hi slot q ^ sw, lo slot (4 + q) ^ sw
    const unsigned sw = (unsigned)(l16 >> 1) & 7u;
    const unsigned b_hi = (unsigned)l16 * 128u + (((unsigned)q ^ sw) & 7u) * 16u;
    typedef const h16x8 __attribute__((address_space(3))) *lds_frag;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)wlds;

    unsigned range_bits = 0;
    const unsigned psel = split_pair_selector(l16 & 1);

    // sub-tile cursor: sub-tile s (0..3) of tile t covers output pixels m = t * 128 + s * 32 + [0, 32)
    auto voff_of = [&](int m) -> unsigned {         // per-lane byte offset of (row l16 of row tile 0, tap (0, 0), chunk q)
        const int n = p.hw_shift >= 0 ? (m >> p.hw_shift) : m / HWo;
        const int rem = m - n * HWo;
        const int y = p.w_shift >= 0 ? (rem >> p.w_shift) : rem / W;
        const int x0 = rem - y * W;
        return (unsigned)((n * Hp + y) * Wp + x0 + l16) * 256u + (unsigned)q * 16u;
    };

    WsFrag ring[WS_DEPTH];
    // issue the A fragment loads of halo K-step ks (static) for the sub-tile whose lane offset is vo
    auto issue = [&](auto ksc, unsigned vo) {
        constexpr int ks = decltype(ksc)::value;
        // K-step order of the x-halo kernel: chunk-major, then dy, then dx
        constexpr int chunk = ks / 9, tap = ks % 9, dy = tap / 3, dx = tap % 3;
        constexpr unsigned imm = dx * 256u + chunk * 128u;
        WsFrag &f = ring[ks % WS_DEPTH];
        // (fenced: left alone the scheduler reverses the order of a group of K-steps' loads, and the first K-step then
        // waits for the last load — vmcnt(0) — instead of for its own four)
        __builtin_amdgcn_sched_barrier(0);
        f.ah[0] = ws_load16(rs_a, vo + imm, s_dy[dy]);
        f.al[0] = ws_load16(rs_a, vo + imm + 64u, s_dy[dy]);
        f.ah[1] = ws_load16(rs_a, vo + imm + 4096u, s_dy[dy]);
        f.al[1] = ws_load16(rs_a, vo + imm + 4096u + 64u, s_dy[dy]);
        __builtin_amdgcn_sched_barrier(0);
    };

    unsigned vo_cur = voff_of(tile * 128);
    ws_static_for<WS_DEPTH>([&](auto kc) { issue(kc, vo_cur); });

    for (; tile < tile_end; ++tile) {
        const int m_tile = tile * 128;
        const int img = p.hw_shift >= 0 ? (m_tile >> p.hw_shift) : m_tile / HWo;
        // column bias of the tile: conv bias + FeatureWiseAffine bias of the tile's image (a tile never spans two images)
        float cb[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float v = p.bias ? p.bias[nt * 16 + l16] : 0.f;
            if (p.chan_bias != nullptr) v += p.chan_bias[(size_t)img * p.chan_bias_stride + nt * 16 + l16];
            cb[nt] = v;
        }
        double st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0};
        for (int sub = 0; sub < 4; ++sub) {
            const int m_sub = m_tile + sub * 32;
            const bool last_sub = sub == 3 && tile + 1 >= tile_end;
            const unsigned vo_next = last_sub ? vo_cur : voff_of(m_sub + 32);      // (tiles of a wave are consecutive)
            f32x4 acc[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
            // ---------------- halo K-steps: weights from LDS, A fragments from the register ring
            ws_static_for<WS_KSTEPS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                const unsigned bh = lds0 + (unsigned)ks * 8192u + b_hi;
                const unsigned bl = bh ^ 64u;
                h16x8 bqh[4], bql[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bqh[nt] = *(lds_frag)(bh + nt * 2048);
                    bql[nt] = *(lds_frag)(bl + nt * 2048);
                }
                const WsFrag &f = ring[ks % WS_DEPTH];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.al[mt], bqh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[mt], bql[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[mt], bqh[nt], acc[mt][nt], 0, 0, 0);
                    }
                // the MFMAs have read the ring slot: refill it with K-step ks + DEPTH of this sub-tile (the loads are
                // issued while the matrix pipe works through the 24 MFMAs above)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks + WS_DEPTH < WS_KSTEPS) issue(std::integral_constant<int, ks + WS_DEPTH>{}, vo_cur);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---------------- fused 1x1 K-steps (res_conv over x || skip, or the identity skip as 2^k I): both operands
            //                  straight from memory (weights: 16-49 KB, L1 / L2 resident)
            if (n2 > 0) {
                const int n = img;
                const int rem = m_sub - n * HWo;
                const int y = p.w_shift >= 0 ? (rem >> p.w_shift) : rem / W;
                const int x0 = rem - y * W;
                const unsigned pixa = (unsigned)p.in2.pix(n, y, x0 + l16);
                for (int k2 = 0; k2 < n2; ++k2) {
                    const int c0 = k2 * 32;
                    const bool first = c0 < C2a;
                    const char *ab = first ? reinterpret_cast<const char *>(p.in2.p + c0)
                                           : reinterpret_cast<const char *>(p.in2b.p + (c0 - C2a));
                    const unsigned Cs = first ? (unsigned)C2a : (unsigned)(C2 - C2a);
                    const unsigned pix = first ? pixa : (unsigned)p.in2b.pix(n, y, x0 + l16);
                    h16x8 ah[2], al[2], bqh[4], bql[4];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const char *a = ab + ((size_t)(pix + mt * 16) * Cs) * 4 + q * 16;
                        ah[mt] = *reinterpret_cast<const h16x8 *>(a);
                        al[mt] = *reinterpret_cast<const h16x8 *>(a + 64);
                    }
                    const char *wb = reinterpret_cast<const char *>(p.w2 + c0) + q * 16;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const char *b = wb + (size_t)(nt * 16 + l16) * C2 * 4;
                        bqh[nt] = *reinterpret_cast<const h16x8 *>(b);
                        bql[nt] = *reinterpret_cast<const h16x8 *>(b + 64);
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bqh[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bql[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bqh[nt], acc[mt][nt], 0, 0, 0);
                        }
                }
            }
            // ---------------- the next sub-tile's first fragments go out BEFORE the epilogue and fly under it
            if (!last_sub) {
                ws_static_for<WS_DEPTH>([&](auto kc) { issue(kc, vo_next); });
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- epilogue: C/D map col = l16 (+ 16 nt), row = 4 q + r (+ 16 mt)
            {
                const int n = img;
                const int rem = m_sub - n * HWo;
                const int y = p.w_shift >= 0 ? (rem >> p.w_shift) : rem / W;
                const int x0 = rem - y * W;
                const unsigned ob = (unsigned)p.out.pix(n, y, x0 + 4 * q) * (unsigned)WS_C;     // element offset of row 4q, column 0
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = fmaf(acc[mt][nt][r], p.w_unscale, cb[nt]);
                            acc[mt][nt][r] = v;
                            st1[nt] += (double)v;
                            st2[nt] = fma((double)v, (double)v, st2[nt]);
                        }
                if (p.out_f32) {
                    char *obase = reinterpret_cast<char *>(p.out.p);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                *reinterpret_cast<float *>(obase + (ob + (unsigned)((mt * 16 + r) * WS_C + nt * 16 + l16)) * 4u) = acc[mt][nt][r];
                }
                if (p.out_split.p != nullptr) {
                    // per 32-channel chunk 32 hi halfs | 32 lo halfs; lanes l16, l16 ^ 1 hold neighbouring channels: the even
                    // lane stores both hi halfs, the odd lane both lo halfs (split_pair_word, sr3_internal.h)
                    char *tlane = reinterpret_cast<char *>(p.out_split.p) + (((l16 & 1) ? 16u : 0u) + ((unsigned)l16 >> 1)) * 4u;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const unsigned coff = ((unsigned)(nt >> 1) * 32u + (unsigned)(nt & 1) * 8u) * 4u;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const unsigned word = split_pair_word(acc[mt][nt][r], psel, range_bits);
                                *reinterpret_cast<unsigned *>(tlane + (ob + (unsigned)((mt * 16 + r) * WS_C)) * 4u + coff) = word;
                            }
                    }
                }
            }
            vo_cur = vo_next;
        }
        // ---- statistics slice of the tile: add the four row groups (lanes q) in a fixed order, lanes q == 0 write
        if (p.stats != nullptr) {
            const int slice = p.stats_slice0 + (m_tile - img * HWo) / 128;
            double *o = p.stats + (((size_t)img * p.stats_slices + slice) * WS_C) * 2;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                double a = st1[nt], b = st2[nt];
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                if (q == 0) { o[(nt * 16 + l16) * 2] = a; o[(nt * 16 + l16) * 2 + 1] = b; }
            }
        }
    }
    if (p.ovf != nullptr && split_range_overflow(range_bits)) *p.ovf = 1;
}

}  // namespace

// Shapes the weights-stationary kernel takes (launch_conv asks; everything else stays on the LDS-ring kernels)
bool conv_ws_supported(const ConvParams &p) {
    static const int off = exp_int("SR3_NO_WS", 0);
    if (off || p.prec != 1 || p.f8 || p.ks != 3 || p.stride != 1 || p.up2 || p.phases != 1 || p.splits > 1) return false;
    if (p.gnf_gamma != nullptr || p.resid.p != nullptr || p.dbg != 0) return false;
    if (p.in0.C != WS_C || p.in1.p != nullptr || p.out.C != WS_C || p.in0.pad != 1) return false;
    if (p.in0.W != p.Wout || p.in0.H != p.Hout || p.out_step != 1 || p.org_x || p.org_y) return false;
    const int W = p.Wout, HWo = p.Hout * W;
    if ((W % 32) != 0 || (HWo % 128) != 0) return false;
    if (p.in2.p && ((p.in2.C % 32) != 0 || p.in2.H != p.Hout || p.in2.W != W)) return false;
    if (p.in2b.p && (!p.in2.p || (p.in2b.C % 32) != 0 || p.in2b.H != p.Hout || p.in2b.W != W)) return false;
    if (p.in2.p && p.w2 == nullptr) return false;
    if (p.stats != nullptr && p.stats_slices != HWo / 128) return false;
    // at least one 128-pixel tile for every wave of a chip-wide launch (each block loads the 144 KB weight tensor once)
    const long tiles = (long)p.B * HWo / 128;
    return tiles >= 2048;
}

void launch_conv_ws(const ConvParams &p, hipStream_t s) {
    static int cus = 0;
    if (!cus) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_ws64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        cus = (cus + 7) / 8 * 8;
    }
    const int tiles = (int)((long)p.B * p.Hout * p.Wout / 128);
    const int waves = cus * WS_WAVES;
    const int tpw = (tiles + waves - 1) / waves;
    hipLaunchKernelGGL(conv3x3_ws64_kernel, dim3(cus), dim3(512), WS_LDS_BYTES, s, p, tiles, tpw);
}

}  // namespace sr3
