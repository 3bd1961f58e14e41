// EXPERIMENT (libsr3hip_exp.so only, SR3_WS=1; profiles/README.md finding 66): built, correct on whole-image-aligned tile
// ranges, measured SLOWER than the x-halo kernel it was meant to replace (0.31 vs 0.26 ms per 128x128 64 -> 64 conv at B = 64,
// plus 0.05 ms on the GroupNorm apply pass that has to write the fragment-major layout). The product library compiles none
// of it. Kept as the measured record of the "persistent block per CU" form VERDICT r3 asked for, with its in-kernel timeline.
//
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
//     32-channel chunk (hi halfs) + 16 more (lo halfs), so every wave loads its fragments straight from memory into
//     registers, two (chunk, dy) groups ahead of their use. The activated input is stored FRAGMENT-MAJOR for that
//     (sr3_internal.h fm_*; written by the GroupNorm apply pass, split = 3): the fragment of 16 pixels is 1 KB of consecutive
//     memory and the wave's buffer_load_dwordx4 is lane-linear — read pixel-major (64 B per pixel, pixels 256 B apart) the
//     texture path took 2.4x as long and was the bound. Only ALIGNED 16-pixel blocks are loaded (three per group and half:
//     6 KB for three K-steps); the fragments of the taps dx = 1, 2 are made from them by DPP row shifts (a DPP row is the
//     16 pixels of one channel octet). ONE per-lane offset (lane * 16) serves every A load of the launch; sub-tile, row
//     and block are scalar offsets, chunk and hi / lo immediates: no vector address arithmetic in the K loop;
//   * a wave owns its 32 pixels x 64 channels completely: NO barrier and no shared pipeline state inside the launch (one
//     barrier after the weight load) — eight independent wave pipelines per CU, two per SIMD, which de-phase by
//     themselves: one wave's epilogue (VALU + stores) runs under the other's MFMAs;
//   * a wave processes whole 128-pixel tiles (four 32-pixel sub-tiles in sequence) and keeps the fused GroupNorm
//     statistics of the tile in fp64 registers: the statistics slice layout is the x-halo kernel's (one slice per
//     128-row tile), no LDS staging, no atomics, deterministic.
// Products, accumulation order inside a K-step and the K-step order (chunk-major, then dy, dx; fused 1x1 K-steps last) are
// those of the x-halo kernel's 16x16x32 consumers, so results agree with it to fp32 summation order.
#include "sr3_internal.h"

#ifndef SR3_EXPERIMENTS
namespace sr3 {
// product build: the kernel does not exist; no conv input is ever written fragment-major
bool conv_ws_shape_ok(int, int, int, int, int) { return false; }
bool conv_ws_supported(const ConvParams &) { return false; }
void launch_conv_ws(const ConvParams &, hipStream_t) {}
}  // namespace sr3
#else

#include <stdio.h>
#include <algorithm>
#include <type_traits>
#include <vector>

namespace sr3 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_C = 64;                       // Cin == Cout == 64
constexpr int WS_KSTEPS = 18;                  // 9 taps x 2 chunks of 32 channels
constexpr int WS_LDS_BYTES = WS_KSTEPS * WS_C * 128;      // 147,456
#ifndef SR3_WS_WAVES
#define SR3_WS_WAVES 8
#endif
constexpr int WS_WAVES = SR3_WS_WAVES;         // 8 (two per SIMD, 256 registers), 12 (three, 168) or 16 (four, 128)
constexpr int WS_THREADS = WS_WAVES * 64;
#ifndef SR3_WS_DYN
#define SR3_WS_DYN 1
#endif
constexpr bool WS_DYN = SR3_WS_DYN != 0;       // tiles handed out by per-XCD counters (ConvParams::tile_cnt) instead of statically
constexpr int WS_GROUPS = 6;                   // (chunk, dy) groups of a sub-tile: three K-steps (dx) each
#ifndef SR3_WS_MT
#define SR3_WS_MT 4
#endif
constexpr int WS_MT = SR3_WS_MT;               // 16-pixel row tiles per wave sub-tile: 2 (32 pixels) or 4 (64 pixels)
constexpr int WS_SUBPIX = 16 * WS_MT, WS_NSUB = 128 / WS_SUBPIX;
#ifndef SR3_WS_RING
#define SR3_WS_RING (SR3_WS_MT == 2 ? 3 : 2)
#endif
constexpr int WS_RING = SR3_WS_RING;           // groups of A blocks in registers per wave (8 (MT + 1) registers each): 2 or 3

template <int N, class F>
__device__ __forceinline__ void ws_static_for(F &&f) {
    if constexpr (N > 0) {
        ws_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ u32x4 ws_load16(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// The A operand of one (chunk, dy) group: three 16-pixel blocks of the fragment-major tensor (pixels x0 .. x0 + 47 of the
// padded row; the last one only for its first two pixels: MT + 1 blocks for MT row tiles), hi and lo halfs — each block is 1 KB of consecutive memory and
// IS the MFMA fragment of its 16 pixels for tap dx = 0 (lane (l16, q) = slot q * 16 + l16 of the block).
struct WsGroup { u32x4 h[WS_MT + 1], l[WS_MT + 1]; };

// Fragment of a row tile for tap dx from the aligned blocks: lane l16 needs pixel l16 + dx of block `a`, or pixel
// l16 + dx - 16 of the next block `nx`. The 16 lanes of one q are one DPP row, so a pixel shift is a row shift: two DPP
// moves per register (row_shr from the next block into the lanes that the row_shl of this block leaves untouched).
template <int DX>
__device__ __forceinline__ h16x8 ws_shift(const u32x4 a, const u32x4 nx) {
    if constexpr (DX == 0) {
        return __builtin_bit_cast(h16x8, a);
    } else {
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int u = __builtin_amdgcn_update_dpp((int)nx[d], (int)nx[d], 0x110 + (16 - DX), 0xF, 0xF, true);   // row_shr:16-DX (other lanes: don't care)
            r[d] = (unsigned)__builtin_amdgcn_update_dpp(u, (int)a[d], 0x100 + DX, 0xF, 0xF, false);           // row_shl:DX
        }
        return __builtin_bit_cast(h16x8, r);
    }
}

// In-kernel timeline (ConvParams::dbg bit 6): per wave the shader cycles spent in the K loops (halo + fused 1x1 K-steps), in
// the epilogues and in total, the sub-tile count and the 100 MHz real-time ticks — launch_conv_ws prints the averages
// (a build of its own, -DSR3_WS_TIMELINE=1: the counters cost 14 registers the 64-pixel form does not have)
#ifndef SR3_WS_TIMELINE
#define SR3_WS_TIMELINE 0
#endif
#if SR3_WS_TIMELINE
#define WS_STAMP(v) if (p.dbg & 64) { v = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#define WS_TL_ADD if (p.dbg & 64) { tl_k += tl_b - tl_a; tl_e += tl_c - tl_b; ++tl_n; }
#else
#define WS_STAMP(v)
#define WS_TL_ADD
#endif

// One persistent block per CU, 8 waves. Tiles of 128 consecutive output pixels (whole tiles per image; W is 32, 64 or 128,
// so a tile is 4, 2 or 1 whole image rows and a 32-pixel sub-tile never leaves its row).
// Wave g (XCD-major numbering, so that the tiles of an image stay in one XCD's L2) takes tiles [g * tpw, (g + 1) * tpw).
__global__ __launch_bounds__(WS_THREADS, WS_WAVES / 4) void conv3x3_ws64_kernel(const ConvParams p, const int tiles_total, const int tpw) {
    extern __shared__ __attribute__((aligned(128))) float wlds[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int l16 = lane & 15, q = lane >> 4;
#if SR3_WS_TIMELINE
    const long long tr_launch = (long long)__builtin_amdgcn_s_memrealtime();      // (timeline: the wave's first instruction)
#endif

    // ---- weights -> LDS, once: row R = kstep * 64 + cout (128 B = 32 hi halfs | 32 lo halfs of one 32-channel chunk),
    //      16-byte slot s of a row holds source chunk s ^ ((cout >> 1) & 7) (conflict-free ds_read_b128 of the fragments).
    //      All 18 loads of a thread are in flight before the first LDS write (one round trip, L2-resident after the first block).
    {
        const char *wsrc = reinterpret_cast<const char *>(p.w);
        u32x4 wv[WS_LDS_BYTES / (WS_THREADS * 16)];
#pragma unroll
        for (int i = 0; i < WS_LDS_BYTES / (WS_THREADS * 16); ++i) {
            const int L = i * WS_THREADS + tid;
            const int R = L >> 3, s = L & 7;
            const int ks = R >> 6, o = R & 63;
            const int chunk = ks / 9, tap = ks - chunk * 9;      // K-step order of the consumers: chunk-major, then the tap
            wv[i] = *reinterpret_cast<const u32x4 *>(wsrc + ((size_t)(tap * WS_C + o) * WS_C + chunk * 32) * 4 + ((s ^ ((o >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < WS_LDS_BYTES / (WS_THREADS * 16); ++i) {
            const int L = i * WS_THREADS + tid;
            *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(wlds) + (L >> 3) * 128 + (L & 7) * 16) = wv[i];
        }
    }
    __syncthreads();            // the only barrier of the launch

    // ---- this wave's tiles
    const int nblk = (int)gridDim.x;
    const int xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int per_xcd = (nblk + 7) >> 3;
    const int g = (xcd * per_xcd + loc) * WS_WAVES + wid;
    // static: wave g takes tiles [g * tpw, (g + 1) * tpw). dynamic (WS_DYN): XCD x owns tiles [x * tpx, (x + 1) * tpx) and its
    // waves take them one at a time from the XCD's counter (ConvParams::tile_cnt[x]; word 8 counts finished waves: the last
    // one leaves all nine words at zero for the next launch) — the waves of a launch then finish together whatever their
    // luck with the matrix pipe and the memory system (static: wave end times spread over 148 … 213 us)
    const int tpx = (tiles_total + 7) >> 3;
    auto grab = [&]() -> int {
        int t = 0;
        if (lane == 0) t = (int)__hip_atomic_fetch_add(p.tile_cnt + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        const int gt = xcd * tpx + t;
        return (t < tpx && gt < tiles_total) ? gt : -1;
    };
    auto wave_done = [&]() {
        if (WS_DYN && lane == 0) {
            const unsigned total = gridDim.x * WS_WAVES;
            const unsigned old = __hip_atomic_fetch_add(p.tile_cnt + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == total - 1u)
                for (int i = 0; i < 9; ++i) __hip_atomic_store(p.tile_cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    int tile = WS_DYN ? grab() : g * tpw;
    const int tile_end = WS_DYN ? tiles_total : min(tile + tpw, tiles_total);
    if (tile < 0 || tile >= tile_end) { wave_done(); return; }
    int tile_next = -1;

    const int W = p.Wout, HWo = p.Hout * W;
    const int Hp = p.in0.Hp();
    const int G = fm_groups(W);
    const unsigned rowb = (unsigned)G * 4096u;                  // bytes of one padded row of the FM tensor (64 channels)
    const int C2a = p.in2.p ? p.in2.C : 0, C2 = C2a + (p.in2b.p ? p.in2b.C : 0);
    const int n2 = C2 >> 5;                                     // fused 1x1 K-steps
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in0.p), 0, -1, 0x00020000);
    const unsigned v_lane = (unsigned)lane * 16u;               // the ONE per-lane offset of every A load of the launch
    // byte offset of the 16-pixel block that holds padded pixel (n, y, x0) — uniform
    auto sub_base = [&](int m) -> unsigned {
        const int n = p.hw_shift >= 0 ? (m >> p.hw_shift) : m / HWo;
        const int rem = m - n * HWo;
        const int y = p.w_shift >= 0 ? (rem >> p.w_shift) : rem / W;
        return (unsigned)((n * Hp + y) * G + ((rem - y * W) >> 4)) * 4096u;
    };

    // B fragment addresses in LDS: lane (l16, q) reads row (nt * 16 + l16) of the K-step's 64, hi slot q ^ sw, lo slot (4 + q) ^ sw
    const unsigned sw = (unsigned)(l16 >> 1) & 7u;
    const unsigned b_hi = (unsigned)l16 * 128u + (((unsigned)q ^ sw) & 7u) * 16u;
    typedef const h16x8 __attribute__((address_space(3))) *lds_frag;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)wlds;

    unsigned range_bits = 0;
    const unsigned psel = split_pair_selector(l16 & 1);

    WsGroup ring[WS_RING];
    // issue the six block loads of group gi (static: chunk-major, then dy — the x-halo kernel's K-step order) of the sub-tile
    // whose first block is at sb. Every load is lane-linear: voffset = lane * 16, everything else scalar / immediate.
    auto issue = [&](auto gic, unsigned sb) {
        constexpr int gi = decltype(gic)::value;
        constexpr int chunk = gi / 3, dy = gi % 3;
        constexpr unsigned imm = chunk * 2048u;
        WsGroup &f = ring[gi % WS_RING];
        const unsigned so = sb + (unsigned)dy * rowb;
        // (fenced: left alone the scheduler reorders a group of loads and the first use waits for the last of them)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < WS_MT + 1; ++b) {
            f.h[b] = ws_load16(rs_a, v_lane + imm, so + (unsigned)b * 4096u);
            f.l[b] = ws_load16(rs_a, v_lane + imm + 1024u, so + (unsigned)b * 4096u);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#if SR3_WS_TIMELINE
    long long tl_a = 0, tl_b = 0, tl_c = 0, tl_k = 0, tl_e = 0, tl_n = 0;
    const long long tl_0 = (long long)__builtin_amdgcn_s_memtime();
    const long long tr_0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    unsigned sb_cur = sub_base(tile * 128);
    ws_static_for<WS_RING>([&](auto gc) { issue(gc, sb_cur); });
    // The two waves of a SIMD (w and w + 4) would run in lock-step — both in their K loops (sharing the matrix pipe), then
    // both in their epilogues (pipe idle). Half a K loop of delay for the upper four lets one wave's epilogue run under
    // the other's MFMAs.
    if (wid >= 12) __builtin_amdgcn_s_sleep(120);
    else if (wid >= 8) __builtin_amdgcn_s_sleep(80);
    else if (wid >= 4) __builtin_amdgcn_s_sleep(40);

    for (; tile >= 0 && tile < tile_end; tile = tile_next) {
        // the tile after this one: the next of the static range, or the XCD counter's (taken now: its latency hides under the tile)
        tile_next = WS_DYN ? grab() : (tile + 1 < tile_end ? tile + 1 : -1);
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
        for (int sub = 0; sub < WS_NSUB; ++sub) {
            const int m_sub = m_tile + sub * WS_SUBPIX;
            const bool last_sub = sub == WS_NSUB - 1 && tile_next < 0;
            const unsigned sb_next = last_sub ? sb_cur : sub_base(sub == WS_NSUB - 1 ? tile_next * 128 : m_sub + WS_SUBPIX);
            f32x4 acc[WS_MT][4];
#pragma unroll
            for (int mt = 0; mt < WS_MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
            WS_STAMP(tl_a)
            // ---------------- halo K-steps: weights from LDS, A fragments from the block ring (dx = 0 as loaded, 1 and 2 by DPP)
            ws_static_for<WS_GROUPS>([&](auto gic) {
                constexpr int gi = decltype(gic)::value;
                const WsGroup &f = ring[gi % WS_RING];
                ws_static_for<3>([&](auto dxc) {
                    constexpr int dx = decltype(dxc)::value;
                    constexpr int ks = gi * 3 + dx;
                    const unsigned bh = lds0 + (unsigned)ks * 8192u + b_hi;
                    const unsigned bl = bh ^ 64u;
                    h16x8 bqh[4], bql[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        bqh[nt] = *(lds_frag)(bh + nt * 2048);
                        bql[nt] = *(lds_frag)(bl + nt * 2048);
                    }
                    // row tile by row tile: its two shifted fragments (8 registers) are made right in front of their twelve
                    // MFMAs; within a row tile the three products of one accumulator (al*bh, ah*bl, ah*bh — the x-halo kernel's
                    // order) are four MFMAs apart, so no MFMA waits for the result of the one in front of it
#pragma unroll
                    for (int mt = 0; mt < WS_MT; ++mt) {
                        const h16x8 ah = ws_shift<dx>(f.h[mt], f.h[mt + 1]);
                        const h16x8 al = ws_shift<dx>(f.l[mt], f.l[mt + 1]);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bqh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bql[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bqh[nt], acc[mt][nt], 0, 0, 0);
                    }
                });
                // the group's registers are free: refill them with group gi + RING of this sub-tile, or with a group of the
                // next sub-tile (its first three go out here, before the epilogue, and fly under it)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (gi + WS_RING < WS_GROUPS) issue(std::integral_constant<int, gi + WS_RING>{}, sb_cur);
                else if (!last_sub) issue(std::integral_constant<int, gi + WS_RING - WS_GROUPS>{}, sb_next);
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
                    h16x8 bqh[4], bql[4];
                    const char *wb = reinterpret_cast<const char *>(p.w2 + c0) + q * 16;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const char *b = wb + (size_t)(nt * 16 + l16) * C2 * 4;
                        bqh[nt] = *reinterpret_cast<const h16x8 *>(b);
                        bql[nt] = *reinterpret_cast<const h16x8 *>(b + 64);
                    }
                    // (row tile by row tile: 8 operand registers at a time; these few K-steps are not software-pipelined)
#pragma unroll
                    for (int mt = 0; mt < WS_MT; ++mt) {
                        const char *a = ab + ((size_t)(pix + mt * 16) * Cs) * 4 + q * 16;
                        const h16x8 ah = *reinterpret_cast<const h16x8 *>(a);
                        const h16x8 al = *reinterpret_cast<const h16x8 *>(a + 64);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bqh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bql[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bqh[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
            WS_STAMP(tl_b)
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
                    for (int mt = 0; mt < WS_MT; ++mt)
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
                    for (int mt = 0; mt < WS_MT; ++mt)
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
                        for (int mt = 0; mt < WS_MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const unsigned word = split_pair_word(acc[mt][nt][r], psel, range_bits);
                                *reinterpret_cast<unsigned *>(tlane + (ob + (unsigned)((mt * 16 + r) * WS_C)) * 4u + coff) = word;
                            }
                    }
                }
            }
            sb_cur = sb_next;
            WS_STAMP(tl_c)
            WS_TL_ADD
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
    wave_done();
#if SR3_WS_TIMELINE
    if ((p.dbg & 64) && p.part != nullptr && lane == 0) {
        long long *o = reinterpret_cast<long long *>(p.part) + (size_t)g * 8;
        const long long tr_end = (long long)__builtin_amdgcn_s_memrealtime();
        o[0] = tl_k; o[1] = tl_e; o[2] = (long long)__builtin_amdgcn_s_memtime() - tl_0;
        o[3] = tl_n | ((tr_end - tr_0) << 16);      // [15:0] sub-tiles, [63:16] 100 MHz ticks
        o[4] = tr_launch; o[5] = tr_0; o[6] = tr_end;       // absolute 100 MHz stamps: first instruction, after the barrier, end
    }
#endif
}

}  // namespace

// Shapes the weights-stationary kernel takes: the engine asks first (conv_ws_shape_ok) and writes the conv's input
// fragment-major; launch_conv then checks the launch itself (conv_ws_supported)
bool conv_ws_shape_ok(int B, int H, int W, int Cin, int Cout) {
    static const int on = exp_int("SR3_WS", 0);
    if (!on || env_int("SR3_NO_HALO", 0) || Cin != WS_C || Cout != WS_C) return false;
    if (!(W == 32 || W == 64 || W == 128) || W < WS_SUBPIX || ((H * W) % 128) != 0) return false;
    // at least one 128-pixel tile for every wave of a chip-wide launch (each block loads the 144 KB weight tensor once)
    return (long)B * H * W / 128 >= 2048 && fm_floats(B, WS_C, H, W) * sizeof(float) < (1ull << 32);
}

bool conv_ws_supported(const ConvParams &p) {
    if (!p.in_fm || p.prec != 1 || p.f8 || p.ks != 3 || p.stride != 1 || p.up2 || p.phases != 1 || p.splits > 1) return false;
    if (p.gnf_gamma != nullptr || p.resid.p != nullptr || (p.dbg & ~64) != 0) return false;
    if (p.in0.C != WS_C || p.in1.p != nullptr || p.out.C != WS_C || p.in0.pad != 1) return false;
    if (p.in0.W != p.Wout || p.in0.H != p.Hout || p.out_step != 1 || p.org_x || p.org_y) return false;
    const int W = p.Wout, HWo = p.Hout * W;
    if (!conv_ws_shape_ok(p.B, p.Hout, W, p.in0.C, p.out.C)) return false;
    if (p.in2.p && ((p.in2.C % 32) != 0 || p.in2.H != p.Hout || p.in2.W != W)) return false;
    if (p.in2b.p && (!p.in2.p || (p.in2b.C % 32) != 0 || p.in2b.H != p.Hout || p.in2b.W != W)) return false;
    if (p.in2.p && p.w2 == nullptr) return false;
    if (WS_DYN && p.tile_cnt == nullptr) return false;
    return p.stats == nullptr || p.stats_slices == HWo / 128;
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
    if (SR3_WS_TIMELINE && (p.dbg & 64)) {       // timeline run: a buffer of its own, synchronous, averages to stderr
        ConvParams q = p;
        long long *buf = nullptr;
        (void)hipMalloc(&buf, (size_t)waves * 8 * sizeof(long long));
        (void)hipMemset(buf, 0, (size_t)waves * 8 * sizeof(long long));
        q.part = reinterpret_cast<float *>(buf);
        hipLaunchKernelGGL(conv3x3_ws64_kernel, dim3(cus), dim3(WS_THREADS), WS_LDS_BYTES, s, q, tiles, tpw);
        (void)hipStreamSynchronize(s);
        std::vector<long long> h((size_t)waves * 8);
        (void)hipMemcpy(h.data(), buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(buf);
        double k = 0, e = 0, t = 0, n = 0, rt = 0; int cnt = 0;
        long long t_first = 0, t_last_start = 0, t_bar_min = 0, t_bar_max = 0, t_end_min = 0, t_end_max = 0;
        for (int i = 0; i < waves; ++i) if (h[i * 8 + 3] > 0) {
            k += h[i * 8]; e += h[i * 8 + 1]; t += h[i * 8 + 2]; n += h[i * 8 + 3] & 0xFFFF; rt += (double)(h[i * 8 + 3] >> 16);
            const long long a = h[i * 8 + 4], b = h[i * 8 + 5], c2 = h[i * 8 + 6];
            if (!cnt) { t_first = t_last_start = a; t_bar_min = t_bar_max = b; t_end_min = t_end_max = c2; }
            t_first = std::min(t_first, a); t_last_start = std::max(t_last_start, a);
            t_bar_min = std::min(t_bar_min, b); t_bar_max = std::max(t_bar_max, b);
            t_end_min = std::min(t_end_min, c2); t_end_max = std::max(t_end_max, c2);
            ++cnt;
        }
        static int printed = 0;
        if (cnt && printed++ < 3)
            fprintf(stderr, "ws timeline: %d waves, %.1f sub-tiles each; per sub-tile K loop %.0f cycles, epilogue %.0f; per wave total %.0f cycles "
                    "(K %.1f %%, epilogue %.1f %%, rest = weight load + first loads %.1f %%); %.1f us per wave on the 100 MHz counter => s_memtime ticks at %.0f MHz\n",
                    cnt, n / cnt, k / n, e / n, t / cnt, 100 * k / t, 100 * e / t, 100 * (t - k - e) / t, rt / cnt / 100.0, t / (rt / 100.0));
        if (cnt && printed <= 3)
            fprintf(stderr, "ws launch timeline (us from the first wave's first instruction): last wave starts %.1f; barrier passed %.1f .. %.1f; "
                    "waves end %.1f .. %.1f\n", (t_last_start - t_first) / 100.0, (t_bar_min - t_first) / 100.0, (t_bar_max - t_first) / 100.0,
                    (t_end_min - t_first) / 100.0, (t_end_max - t_first) / 100.0);
        return;
    }
    hipLaunchKernelGGL(conv3x3_ws64_kernel, dim3(cus), dim3(WS_THREADS), WS_LDS_BYTES, s, p, tiles, tpw);
}

}  // namespace sr3
#endif  // SR3_EXPERIMENTS
