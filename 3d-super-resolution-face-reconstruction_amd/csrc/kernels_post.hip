// Post-processing step behind the sampler: what the reference does on the host
// between `super_resolution` and the MICA/ArcFace encoder, done on the device so the SR images
// never leave HBM. HBM-bound byte work; no MFMA.
//   u8 chain   (model/sr3d/model.py:372-386, :462-471): tensor2img (core/metrics.py:16-42) ->
//              cv2.resize(.., (224, 224)) [INTER_LINEAR, 8-bit fixed point] -> images / 255 and
//              cv2.dnn.blobFromImages(.., 1/127.5, (112, 112), 127.5, swapRB=True) (:127-131).
//   tensor chain (model/sr3d/model.py:474-483): tensor2tensor_img * 255 (core/metrics.py:44-50) ->
//              create_tensor_blob (:105-124): (x - 127.5) / 127.5, F.interpolate(bilinear,
//              align_corners=False) to 112, channel swap.
// cv2 is a third-party dependency that is not installed here: its INTER_LINEAR for 8-bit images
// is restated from OpenCV 4.x modules/imgproc/src/resize.cpp (coefficient set-up in resize(),
// HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,FixedPtCast<.., 22>>, and the
// "scale == 2 -> INTER_AREA" shortcut with ResizeAreaFastVec) — parity unpinned (no cv2 fixture).
#include "sr3_internal.h"
#include <math.h>
#include <vector>

namespace sr3 {

namespace {
constexpr int COEF_BITS = 11;                 // INTER_RESIZE_COEF_BITS
constexpr int COEF_ONE = 1 << COEF_BITS;      // INTER_RESIZE_COEF_SCALE

// tensor2img for one value: clamp(-1,1) -> (v+1)/2 -> *255 -> round half to even -> u8
__device__ __forceinline__ uint8_t to_u8(float v) {
    v = fminf(fmaxf(v, -1.f), 1.f);
    const float t = __fmul_rn(__fdiv_rn(__fadd_rn(v, 1.f), 2.f), 255.f);
    return (uint8_t)(int)rintf(t);
}

// [B][3][H][W] fp32 -> [B][H][W][3] u8
__global__ void tensor2img_kernel(const float *__restrict__ in, int HW, uint8_t *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW pixels
    if (i >= total) return;
    const size_t b = i / HW, p = i - b * HW;
    const float *s = in + b * 3 * HW + p;
    uint8_t *d = out + i * 3;
    d[0] = to_u8(s[0]);
    d[1] = to_u8(s[HW]);
    d[2] = to_u8(s[2 * (size_t)HW]);
}

// tables: xofs[Wd], xa[Wd][2], yofs[Hd], yb[Hd][2] (ints)
// src [B][Hs][Ws][3] u8 -> dst [B][Hd][Wd][3] u8 (+ optional images [B][3][Hd][Wd] = dst / 255)
__global__ void resize_linear_u8_kernel(const uint8_t *__restrict__ src, int Hs, int Ws, int Hd, int Wd,
                                        const int *__restrict__ tab, uint8_t *__restrict__ dst,
                                        float *__restrict__ images, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*Hd*Wd pixels
    if (i >= total) return;
    const int dx = (int)(i % Wd);
    const size_t t = i / Wd;
    const int dy = (int)(t % Hd);
    const size_t b = t / Hd;
    const int *xofs = tab, *xa = tab + Wd, *yofs = tab + 3 * Wd, *yb = tab + 3 * Wd + Hd;
    const int sx = xofs[dx], a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
    const int sx1 = min(sx + 1, Ws - 1);                  // a1 == 0 whenever sx + 1 is outside
    const int sy = yofs[dy], b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    const int y0 = min(max(sy, 0), Hs - 1), y1 = min(max(sy + 1, 0), Hs - 1);
    const uint8_t *r0 = src + (b * Hs + y0) * (size_t)Ws * 3, *r1 = src + (b * Hs + y1) * (size_t)Ws * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;     // horizontal pass, 11 fractional bits
        const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        const uint8_t u = (uint8_t)v;
        dst[i * 3 + c] = u;
        if (images) images[((b * 3 + c) * Hd + dy) * (size_t)Wd + dx] = (float)((double)u / 255.0);
    }
}

// blobFromImages tail: src [B][Hs][Ws][3] u8 (Hs = f*Hb) -> out [B][3][Hb][Wb] fp32, channel c of the
// blob = image channel 2-c (swapRB), value (avg - mean) * scale; f == 2: 2x2 area average
// (a+b+c+d+2)>>2, f == 1: the pixel itself
__global__ void blob_kernel(const uint8_t *__restrict__ src, int Hb, int Wb, int f, float mean, float scale,
                            float *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*Hb*Wb
    if (i >= total) return;
    const int x = (int)(i % Wb);
    const size_t t = i / Wb;
    const int y = (int)(t % Hb);
    const size_t b = t / Hb;
    const int Ws = Wb * f, Hs = Hb * f;
    const uint8_t *p = src + ((b * Hs + (size_t)y * f) * Ws + (size_t)x * f) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v = p[c];
        if (f == 2) v = (v + p[3 + c] + p[(size_t)Ws * 3 + c] + p[(size_t)Ws * 3 + 3 + c] + 2) >> 2;
        out[((b * 3 + (2 - c)) * Hb + y) * (size_t)Wb + x] = __fmul_rn(__fsub_rn((float)v, mean), scale);
    }
}

// tensor chain: [B][3][H][W] fp32 in [-1,1] -> [B][3][Hb][Wb], channel swap, torch bilinear
// (align_corners=False: src = max(scale*(dst+0.5)-0.5, 0), scale = (float)in/out; upsample
// kernels evaluate wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d))
__device__ __forceinline__ float norm255(float v) {
    v = fminf(fmaxf(v, -1.f), 1.f);
    const float t = __fmul_rn(__fdiv_rn(__fadd_rn(v, 1.f), 2.f), 255.f);       // tensor2tensor_img * 255
    return __fdiv_rn(__fsub_rn(t, 127.5f), 127.5f);                             // create_tensor_blob
}
__global__ void tensor_blob_kernel(const float *__restrict__ in, int H, int W, int Hb, int Wb, float sh, float sw,
                                   float *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*3*Hb*Wb (output order)
    if (i >= total) return;
    const int x = (int)(i % Wb);
    size_t t = i / Wb;
    const int y = (int)(t % Hb);
    t /= Hb;
    const int co = (int)(t % 3);
    const size_t b = t / 3;
    const float ry = fmaxf(__fsub_rn(__fmul_rn(sh, (float)y + 0.5f), 0.5f), 0.f);
    const float rx = fmaxf(__fsub_rn(__fmul_rn(sw, (float)x + 0.5f), 0.5f), 0.f);
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float *s = in + (b * 3 + (2 - co)) * (size_t)H * W;
    const float a = norm255(s[(size_t)y0 * W + x0]), bb = norm255(s[(size_t)y0 * W + x1]);
    const float c = norm255(s[(size_t)y1 * W + x0]), d = norm255(s[(size_t)y1 * W + x1]);
    const float top = __fadd_rn(__fmul_rn(lx0, a), __fmul_rn(lx1, bb));
    const float bot = __fadd_rn(__fmul_rn(lx0, c), __fmul_rn(lx1, d));
    out[i] = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
}

inline unsigned nblk(size_t total) { return (unsigned)((total + 255) / 256); }
} // namespace

// OpenCV resize(): per destination index the source index and the two 11-bit weights
void cv_linear_coeffs(int in_size, int out_size, bool horizontal, std::vector<int> &ofs, std::vector<int> &ab) {
    ofs.resize(out_size);
    ab.resize(2 * (size_t)out_size);
    const double inv_scale = (double)out_size / in_size;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < out_size; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (horizontal) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= in_size - 1) { f = 0.f; s = in_size - 1; }
        }
        ofs[d] = s;                                        // vertical: rows are clamped at use
        const float c0 = 1.f - f, c1 = f;
        ab[2 * d] = (int)lrintf(c0 * (float)COEF_ONE);     // saturate_cast<short>: round half to even
        ab[2 * d + 1] = (int)lrintf(c1 * (float)COEF_ONE);
    }
}

void launch_tensor2img(const float *in_nchw, int B, int H, int W, uint8_t *out_hwc, hipStream_t s) {
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(tensor2img_kernel, dim3(nblk(total)), dim3(256), 0, s, in_nchw, H * W, out_hwc, total);
}
void launch_resize_linear_u8(const uint8_t *src, int B, int Hs, int Ws, int Hd, int Wd, const int *tab, uint8_t *dst,
                             float *images, hipStream_t s) {
    const size_t total = (size_t)B * Hd * Wd;
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(nblk(total)), dim3(256), 0, s, src, Hs, Ws, Hd, Wd, tab, dst,
                       images, total);
}
void launch_blob(const uint8_t *src, int B, int Hb, int Wb, int f, float mean, float scale, float *out, hipStream_t s) {
    const size_t total = (size_t)B * Hb * Wb;
    hipLaunchKernelGGL(blob_kernel, dim3(nblk(total)), dim3(256), 0, s, src, Hb, Wb, f, mean, scale, out, total);
}
void launch_tensor_blob(const float *in_nchw, int B, int H, int W, int Hb, int Wb, float *out, hipStream_t s) {
    const size_t total = (size_t)B * 3 * Hb * Wb;
    const float sh = (float)H / (float)Hb, sw = (float)W / (float)Wb;
    hipLaunchKernelGGL(tensor_blob_kernel, dim3(nblk(total)), dim3(256), 0, s, in_nchw, H, W, Hb, Wb, sh, sw, out, total);
}

} // namespace sr3
