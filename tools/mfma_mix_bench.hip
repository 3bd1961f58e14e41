// Micro-benchmark (development tool, not part of the product): what the consumer loop of the halo conv kernels would
// sustain if the two CORRECTION products of the split-f16 scheme (xl*wh + xh*wl) ran on the fp8 matrix path.
//
//   variant 0  f16x3   per 16x16 tile and 32-channel K-step: 3 x v_mfma_f32_16x16x32_f16            (48 cycles)
//   variant 1  f16+f8  1 x v_mfma_f32_16x16x32_f16 per K-step + 1 x v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) per
//              PAIR of K-steps (K = 128 = {xl8 | xh8} x 2 K-steps against {wh8 | wl8} x 2); the column tiles
//              alternate which K-steps they pair, so every K-step issues NT f16 + NT/2 fp8 MFMAs  (32 cycles avg)
//   variant 2  f16x2   2 x f16 MFMA (the round-1 two-product mode: same nominal cycles as variant 1)
//   variant 3  f16x1   1 x f16 MFMA with variant 1's LDS reads (what the reads alone allow)
//
// Shape of the real loop: 4 consumer waves per block stacked along M (wave tile 32 x 128: MT = 2, NT = 8), two blocks per
// CU, every operand fragment re-read from LDS with ds_read_b128 each K-step (A fragments per K-step, B columns streamed),
// fp32 accumulators. No DMA, no barriers: this is the ceiling of the arithmetic + LDS-read part only (finding 27's
// "no DMA after K-step 0" column), on random operands (the clock under load depends on the data).
//
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/mfma_mix_bench tools/mfma_mix_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int ROWB = 128;                 // bytes per LDS row (32 channels: 64 B f16 hi + 64 B second half)
constexpr int ROWS = 128;                 // rows per operand stage (A: 128 pixels, B: 128 output channels)
constexpr int STG = ROWS * ROWB;          // 16 KB
constexpr int MT = 2, NT = 8;

template <int V>
__global__ __launch_bounds__(256, 2) void loop_kernel(const uint4 *__restrict__ src, float *__restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(128))) char lds[];      // [A stage 0][A stage 1][B 0][B 1][B 2]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 5 * STG / 16; i += 256) reinterpret_cast<uint4 *>(lds)[i] = src[(blockIdx.x & 7) * (5 * STG / 16) + i];
    __syncthreads();
    const int l16 = lane & 15, q = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment byte offsets inside a row (the real kernels swizzle the 16-byte chunks by the row; same here)
    const int sw = (l16 >> 1) & 7;
    const int o_hi = ((q ^ sw) & 7) * 16, o_lo = (((4 + q) ^ sw) & 7) * 16;
    // fp8 fragment of a K-step pair: lanes q = 0,1 take the 32 bytes {xl8 | xh8} (second half of the row) of the OLDER
    // K-step, lanes q = 2,3 of the newer one -> per lane: stage select by (q >> 1), 32-byte half select by (q & 1)
    // (the two 16-byte chunks of that half: 4 + 2 (q & 1) and the next one, swizzled by the row like every other chunk)
    const int o_f8a = (((4 + 2 * (q & 1)) ^ sw) & 7) * 16, o_f8b = (((5 + 2 * (q & 1)) ^ sw) & 7) * 16;
    const char *Abase = lds + (w * 32 + l16) * ROWB;
    const char *Bbase = lds + 2 * STG + l16 * ROWB;
    auto kstep = [&](auto parity, int k) {
        constexpr int PAR = decltype(parity)::value;
        const int as = (k & 1) * STG, bs = (k % 3) * STG;             // stage of this K-step
        const int as_old = ((k + 1) & 1) * STG, bs_old = ((k + 2) % 3) * STG;   // stage of K-step k - 1
        h16x8 ah[MT], al[MT];
        i32x8 a8[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const char *r = Abase + mt * 16 * ROWB;
            ah[mt] = *reinterpret_cast<const h16x8 *>(r + as + o_hi);
            if (V == 0 || V == 2) al[mt] = *reinterpret_cast<const h16x8 *>(r + as + o_lo);
            if (V == 1 || V == 3) {
                const char *p8 = r + (q >= 2 ? as : as_old);
                const i32x4 lo4 = *reinterpret_cast<const i32x4 *>(p8 + o_f8a), hi4 = *reinterpret_cast<const i32x4 *>(p8 + o_f8b);
                a8[mt] = i32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const char *c = Bbase + nt * 16 * ROWB;
            const h16x8 bh = *reinterpret_cast<const h16x8 *>(c + bs + o_hi);
            if (V == 0) {
                const h16x8 bl = *reinterpret_cast<const h16x8 *>(c + bs + o_lo);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                }
            } else if (V == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                if (((nt ^ PAR) & 1) == 0) {        // this column tile closes a K-step pair now
                    const char *p8 = c + (q >= 2 ? bs : bs_old);
                    const i32x4 lo4 = *reinterpret_cast<const i32x4 *>(p8 + o_f8a), hi4 = *reinterpret_cast<const i32x4 *>(p8 + o_f8b);
                    const i32x8 b8 = i32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    if (V == 1) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)      // e4m3 x e4m3, constant block scales 2^-15 x 1 (E8M0 112, 127)
                            acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[mt], b8, acc[mt][nt], 0, 0, 0, 112, 0, 127);
                    } else {
                        asm volatile("" ::"v"(b8), "v"(a8[0]), "v"(a8[1]));   // the reads stay, the fp8 MFMAs do not
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
    };
    for (int k = 0; k < steps; k += 2) {
        kstep(std::integral_constant<int, 0>{}, k);
        kstep(std::integral_constant<int, 1>{}, k + 1);
    }
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
    out[blockIdx.x * 256 + tid] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32x32 consumers, 2 x 2 wave grid (wave tile 64 x 64: MI = NI = 2), as in the F8C path of conv3x3_halo_h3:
//   W = 0  f16x3: per 32x32 tile and K-step 6 x v_mfma_f32_32x32x16_f16 (192 cycles)
//   W = 1  f16+f8: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 (128 cycles); the fp8 operand of a lane is
//          its two lo chunks (4 + lh, 6 + lh) — the layout the instruction wants (finding 64)
template <int W>
__global__ __launch_bounds__(256, 2) void loop32_kernel(const uint4 *__restrict__ src, float *__restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(128))) char lds[];      // [A stage 0][A stage 1][B 0][B 1]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4 * STG / 16; i += 256) reinterpret_cast<uint4 *>(lds)[i] = src[(blockIdx.x & 7) * (5 * STG / 16) + i];
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5, wm = w >> 1, wn = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int sw = (li >> 1) & 7;
    const char *Abase = lds + (wm * 64 + li) * ROWB;
    const char *Bbase = lds + 2 * STG + (wn * 64 + li) * ROWB;
    for (int k = 0; k < steps; ++k) {
        const int st = (k & 1) * STG;
        h16x8 ah[2][2], bh[2][2];
        i32x4 al[2][2], bl[2][2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char *ra = Abase + st + t * 32 * ROWB, *rb = Bbase + st + t * 32 * ROWB;
                ah[sb][t] = *reinterpret_cast<const h16x8 *>(ra + (((2 * sb + lh) ^ sw) & 7) * 16);
                al[sb][t] = *reinterpret_cast<const i32x4 *>(ra + (((4 + 2 * sb + lh) ^ sw) & 7) * 16);
                bh[sb][t] = *reinterpret_cast<const h16x8 *>(rb + (((2 * sb + lh) ^ sw) & 7) * 16);
                bl[sb][t] = *reinterpret_cast<const i32x4 *>(rb + (((4 + 2 * sb + lh) ^ sw) & 7) * 16);
            }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sb][mi], bh[sb][ni], acc[mi][ni], 0, 0, 0);
                    if (W == 0) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, al[sb][mi]), bh[sb][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[sb][mi], __builtin_bit_cast(h16x8, bl[sb][ni]), acc[mi][ni], 0, 0, 0);
                    }
                }
                if (W == 1)
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                        __builtin_shufflevector(al[0][mi], al[1][mi], 0, 1, 2, 3, 4, 5, 6, 7),
                        __builtin_shufflevector(bl[0][ni], bl[1][ni], 0, 1, 2, 3, 4, 5, 6, 7), acc[mi][ni], 0, 0, 0, 112, 0, 127);
            }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int W>
static double run32(const uint4 *src, float *out, int blocks, int steps, int reps) {
    const size_t smem = 4 * STG;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(loop32_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    loop32_kernel<W><<<blocks, 256, smem>>>(src, out, steps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) loop32_kernel<W><<<blocks, 256, smem>>>(src, out, steps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int V>
static double run(const uint4 *src, float *out, int blocks, int steps, int reps) {
    const size_t smem = 5 * STG;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(loop_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    loop_kernel<V><<<blocks, 256, smem>>>(src, out, steps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) loop_kernel<V><<<blocks, 256, smem>>>(src, out, steps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, steps = argc > 2 ? atoi(argv[2]) : 4096, reps = 5;
    const size_t n16 = (size_t)8 * 5 * STG / 16;
    std::vector<uint32_t> h(n16 * 4);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (size_t r = 0; r < h.size() / 32; ++r) {          // one 128-byte row: 32 f16 in [-2, 2) then 64 fp8 bytes
        uint16_t *row16 = reinterpret_cast<uint16_t *>(&h[r * 32]);
        uint8_t *row8 = reinterpret_cast<uint8_t *>(&h[r * 32]) + 64;
        for (int i = 0; i < 32; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t x = (uint32_t)(st >> 33);
            row16[i] = (uint16_t)(((x & 1) << 15) | (((x >> 1) % 4 + 12) << 10) | ((x >> 8) & 0x3FF));   // exponents 12..15
        }
        for (int i = 0; i < 64; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            row8[i] = (uint8_t)((st >> 40) & 0xF7 & ~0x40);                                               // finite e4m3, small exponents
        }
    }
    uint4 *src;
    float *out;
    CHECK(hipMalloc(&src, h.size() * 4));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char *names[6] = {"16x16 f16x3 (3 f16 MFMA)", "16x16 f16+f8 (1 f16 + 1/2 fp8 K=128)", "16x16 f16x2 (2 f16 MFMA)", "16x16 f16x1 + the f8 reads",
                            "32x32 f16x3 (6 f16 MFMA)", "32x32 f16+f8 (2 f16 + 1 fp8 K=64)"};
    double ms[6];
    for (int rep = 0; rep < 2; ++rep) {
        ms[0] = run<0>(src, out, blocks, steps, reps);
        ms[1] = run<1>(src, out, blocks, steps, reps);
        ms[2] = run<2>(src, out, blocks, steps, reps);
        ms[3] = run<3>(src, out, blocks, steps, reps);
        ms[4] = run32<0>(src, out, blocks, steps, reps);
        ms[5] = run32<1>(src, out, blocks, steps, reps);
        for (int v = 0; v < 6; ++v) {
            const double flop = 2.0 * 128 * 128 * 32 * (double)steps * blocks;      // algorithmic: one product per MAC
            printf("pass %d  %-40s %8.3f ms  %7.1f ns per K-step  %7.1f TFLOP/s algorithmic  (%.2fx of f16x3)\n", rep, names[v], ms[v],
                   ms[v] * 1e6 / steps, flop / (ms[v] * 1e-3) * 1e-12, ms[0] / ms[v]);
        }
    }
    return 0;
}
