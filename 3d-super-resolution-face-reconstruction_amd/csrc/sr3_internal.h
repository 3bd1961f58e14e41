// Internal declarations shared by the HIP translation units of libsr3hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr3 {

// ---- conv as implicit GEMM (kernels_conv.hip) -------------------------------------------------
// Activations NHWC fp32. The input is the virtual channel-concatenation in0 ‖ in1
// (reference unet.py:261) seen through an optional nearest x2 upsample (unet.py:61); an optional
// per-(image, channel) affine + Swish (GroupNorm folded, unet.py:84-85) is applied while the tile
// is staged, before the zero padding of the 3x3 window.
struct ConvParams {
    const float *in0;
    const float *in1;       // may be null (C1 == 0)
    int C0, C1;             // channels of in0 / in1; both multiples of 32
    int B, Hin, Win;        // stored input size
    int Hout, Wout;
    int ks, stride, up2;    // ks 1|3 (pad = ks/2), stride 1|2, up2 0|1
    const float *w;         // packed [ks*ks][Cout][Cin]
    const float *bias;      // [Cout] or null
    const float *chan_bias; // [B][chan_bias_stride] (+ offset applied by caller) or null
    int chan_bias_stride;
    const float *resid;     // [B][Hout][Wout][Cout] or null
    const float *gn_scale;  // [B][Cin] or null
    const float *gn_shift;
    int swish;
    float *out;             // [B][Hout][Wout][Cout]
    int Cout;
    int dbg = 0;            // timing experiments only (tools/conv_bench.py); 0 in product code
};
// returns the algorithmic FLOPs (2*MAC) of the launch
double launch_conv(const ConvParams &p, hipStream_t s);
// host helper: OIHW -> [tap][Cout][CinPad] (zero pad input channels up to CinPad)
void pack_conv_weight(const float *oihw, int Cout, int Cin, int ks, int CinPad, float *dst);

// ---- GroupNorm statistics (kernels_misc.hip) ---------------------------------------------------
// part: workspace of at least gn_workspace_floats(B, groups) floats
size_t gn_workspace_floats(int B, int groups);
void launch_groupnorm_affine(const float *in0, int C0, const float *in1, int C1, int B, int HW,
                             int groups, const float *gamma, const float *beta, float eps,
                             float *part, float *scale, float *shift, hipStream_t s);

// ---- attention core ----------------------------------------------------------------------------
double launch_attention(const float *qkv, int B, int N, int C, float *out, hipStream_t s);

// ---- noise-level embedding ---------------------------------------------------------------------
struct EmbedParams {
    const float *noise_level; // noise_level[n * nl_stride]; stride 0 broadcasts one scalar
    int nl_stride;
    int dim;                  // inner_channel
    const float *w1, *b1;     // [4dim][dim], [4dim]
    const float *w2, *b2;     // [dim][4dim], [dim]
    const float *nfw, *nfb;   // [total][dim], [total]
    int total;
    float *temb;              // [B][dim]
    float *chan_bias;         // [B][total]
};
void launch_noise_embed(const EmbedParams &p, int B, hipStream_t s);

// ---- layout + DDPM update ----------------------------------------------------------------------
void launch_nchw_to_nhwc(const float *in, int B, int C, int H, int W, float *out, int Cdst,
                         int coff, hipStream_t s);
void launch_nhwc_to_nchw(const float *in, int B, int C, int H, int W, int Csrc, int coff,
                         float *out, hipStream_t s);
void launch_fill_zero(float *p, size_t n, hipStream_t s);
struct UpdateParams {
    float *state;       // [B][HW][Cs] NHWC; x lives in channels [xoff, xoff+3)
    int Cs, xoff, C;    // C = image channels (3)
    const float *eps;   // [B][HW][Ce] NHWC (Ce >= C)
    int Ce;
    const float *noise; // NCHW [B][C][HW] or null -> Philox
    float a, b, c1, c2, sigma; // recip, recipm1, coef1, coef2, exp(0.5*logvar) (0 at t == 0)
    uint64_t seed, image_offset;
    uint32_t draw;
    float *frame;       // NCHW [B][C][HW] or null: copy of the updated image
};
void launch_ddpm_update(const UpdateParams &p, int B, int HW, hipStream_t s);
// x <- noise (NCHW buffer or Philox draw 0) into state channels
void launch_init_state(float *state, int Cs, int xoff, int C, const float *noise, uint64_t seed,
                       uint64_t image_offset, int B, int HW, hipStream_t s);
void launch_philox_normal(uint64_t seed, uint64_t image, uint32_t draw, int n, float *out,
                          hipStream_t s);

} // namespace sr3
