// Pre-processing step in front of the sampler: the reference's conditioning
// image is the LR crop upsampled with PIL's 8-bit bicubic filter
// (datasets/tool/prepare_data.py:24-47 -> Pillow Resample.c), then ToTensor and x*2-1
// (datasets/util.py:76-83). This file does the same on the device, bit-exactly: fixed-point
// coefficients (22 fractional bits) computed on the host exactly as Pillow's precompute_coeffs /
// normalize_coeffs_8bpc, a horizontal pass rounded to 8 bits, then a vertical pass rounded to
// 8 bits and written as the fp32 NCHW tensor the sampler takes. HBM-bound byte work; no MFMA.
#include "sr3_internal.h"
#include <math.h>
#include <vector>

namespace sr3 {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// in [B][H][Win][3] u8 -> out [B][H][Wout][3] u8
__global__ void resample_h_kernel(const uint8_t *__restrict__ in, int H, int Win, int Wout,
                                  const int *__restrict__ bounds, const int *__restrict__ kk, int ksize,
                                  uint8_t *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*H*Wout*3
    if (i >= total) return;
    const int c = (int)(i % 3);
    const size_t t = i / 3;
    const int xx = (int)(t % Wout);
    const size_t row = t / Wout;                                       // b*H + y
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int *k = kk + (size_t)xx * ksize;
    const uint8_t *src = in + (row * Win + x0) * 3 + c;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) ss += (int)src[3 * x] * k[x];
    out[i] = (uint8_t)clip8(ss);
}

// in [B][Hin][W][3] u8 -> tensor [B][3][Hout][W] fp32 in [-1,1] (+ optional u8 HWC copy)
__global__ void resample_v_kernel(const uint8_t *__restrict__ in, int Hin, int Hout, int W,
                                  const int *__restrict__ bounds, const int *__restrict__ kk, int ksize,
                                  float *__restrict__ out_nchw, uint8_t *__restrict__ out_u8, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*Hout*W*3, HWC order
    if (i >= total) return;
    const int c = (int)(i % 3);
    size_t t = i / 3;
    const int x = (int)(t % W);
    t /= W;
    const int yy = (int)(t % Hout);
    const size_t b = t / Hout;
    int v;
    if (bounds) {
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int *k = kk + (size_t)yy * ksize;
        const uint8_t *src = in + ((b * Hin + y0) * W + x) * 3 + c;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < n; ++y) ss += (int)src[(size_t)3 * W * y] * k[y];
        v = clip8(ss);
    } else {
        v = in[((b * Hin + yy) * W + x) * 3 + c];
    }
    if (out_u8) out_u8[i] = (uint8_t)v;
    // ToTensor: u8 -> float / 255; then * (max - min) + min with (min, max) = (-1, 1)
    const float f = __fdiv_rn((float)v, 255.0f);
    out_nchw[((b * 3 + c) * Hout + yy) * W + x] = __fadd_rn(__fmul_rn(f, 2.0f), -1.0f);
}

double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

} // namespace

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2)
int bicubic_coeffs(int in_size, int out_size, std::vector<int> &bounds, std::vector<int> &kk) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            const double k = ww != 0.0 ? w[x] / ww : w[x];
            kk[(size_t)xx * ksize + x] = k < 0 ? (int)(-0.5 + k * (1 << PRECISION_BITS)) : (int)(0.5 + k * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

void launch_resample_h(const uint8_t *in, int B, int H, int Win, int Wout, const int *bounds, const int *kk,
                       int ksize, uint8_t *out, hipStream_t s) {
    const size_t total = (size_t)B * H * Wout * 3;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, H, Win, Wout,
                       bounds, kk, ksize, out, total);
}

void launch_resample_v(const uint8_t *in, int B, int Hin, int Hout, int W, const int *bounds, const int *kk,
                       int ksize, float *out_nchw, uint8_t *out_u8, hipStream_t s) {
    const size_t total = (size_t)B * Hout * W * 3;
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, Hin, Hout, W,
                       bounds, kk, ksize, out_nchw, out_u8, total);
}

} // namespace sr3
