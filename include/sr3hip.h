/*
 * sr3hip.h — C-ABI of libsr3hip.so: the MI355X (gfx950) SR3 iterative-refinement sampler.
 *
 * This is the drop-in boundary for ONE path of zouiner/3d-super-resolution-Face-reconstruction:
 * GaussianDiffusion.p_sample_loop driving the noise-level-conditioned UNet
 * (reference: model/sr/sr3_modules/diffusion.py:189-215, model/sr/sr3_modules/unet.py:235-265).
 * The reference is pure Python/PyTorch and has no FFI of its own; the Python object protocol it
 * exposes (define_G / GaussianDiffusion / UNet) is mirrored by the ctypes host layer in
 * 3d-super-resolution-face-reconstruction_amd/, which binds exactly the entry points below.
 *
 * Conventions
 *   - plain C: opaque context, plain pointers and sizes, int return codes (0 = ok, <0 = error,
 *     message via sr3_last_error()).  No torch types.
 *   - every `*_dev` pointer is a DEVICE pointer on the context's GPU (e.g. tensor.data_ptr()).
 *   - public tensors are fp32, NCHW contiguous (the reference's layout); the library keeps NHWC
 *     internally.
 *   - one context per process/GPU, not thread-safe, all work is stream-ordered on the context's
 *     stream (sr3_set_stream); calls return without synchronising unless stated.
 */
#ifndef SR3HIP_H
#define SR3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR3_MAX_MULTS 8
#define SR3_MAX_ATTN_RES 8
#define SR3_NAME_MAX 192

typedef struct sr3_ctx sr3_ctx;

/* Mirrors the keyword arguments of UNet.__init__ (reference model/sr/sr3_modules/unet.py:161-174)
 * as filled by define_G (reference model/sr/networks.py:83-101). */
typedef struct sr3_unet_cfg {
    int32_t in_channel;      /* 6 for conditional SR3 (cond ‖ x), 3 unconditional */
    int32_t out_channel;     /* 3 */
    int32_t inner_channel;   /* 64 in every reference yml; must be a multiple of 32 here */
    int32_t norm_groups;     /* 32 */
    int32_t n_mults;
    int32_t channel_mults[SR3_MAX_MULTS];
    int32_t n_attn_res;
    int32_t attn_res[SR3_MAX_ATTN_RES];
    int32_t res_blocks;
    int32_t image_size;      /* only decides attention placement (unet.py:192-207) */
    float dropout;           /* accepted for API parity; eval semantics (identity) only */
} sr3_unet_cfg;

/* ---- lifecycle -------------------------------------------------------------------------- */

/* replaces: UNet.__init__ + .cuda()  (unet.py:161-233, model/sr3d/model.py:51) */
int sr3_create(const sr3_unet_cfg *cfg, int device, sr3_ctx **out);
void sr3_destroy(sr3_ctx *ctx);
/* message of the last failing call on this thread ("" if none) */
const char *sr3_last_error(void);
/* Return codes: 0 = ok, < 0 = error (sr3_last_error), and one positive "ok, with a warning" code:
 * the call's result is complete and valid, but the split-f16 arithmetic left its range and (part of) the
 * call was recomputed in exact f32 — message via sr3_last_warning(). See sr3_set_range_policy. */
#define SR3_OK_F32_FALLBACK 1
/* ... and a second one: the result is complete and valid, but part of the call was REPLAYED in the same arithmetic on
 * another kernel path. Cause: the in-place split-K convs (deep-K convs over few tiles — the 8x8 level at B = 64) let the
 * blocks of one output tile wait for each other; that needs a free CU slot for each of them, which an otherwise idle GPU
 * always has. If another kernel (another stream or process: e.g. the ArcFace encoder that follows the SR stage,
 * model/sr3d/model.py:372-393) holds those slots, the bounded wait (5 ms) gives up, the work is replayed on the path
 * without inter-block waits and the context stays on that path. Never a hang, never a failed call.
 * sr3_replay_calls counts them; SR3_HALO_SPLITS=0 in the environment selects the non-waiting path from the start. */
#define SR3_OK_REPLAYED 2
const char *sr3_last_warning(void);
/* work is enqueued on `hip_stream` (a hipStream_t; NULL = the context's own stream) */
int sr3_set_stream(sr3_ctx *ctx, void *hip_stream);
int sr3_synchronize(sr3_ctx *ctx);
/* Device-side ordering against another stream of the host (no host synchronisation): the context's stream
 * waits for everything enqueued so far on `other_stream` / `other_stream` waits for everything enqueued so far
 * on the context's stream. (A host that keeps its tensors on its own stream — torch's default stream cannot
 * be captured into a hipGraph, so the library then runs on its own stream — calls the first before handing
 * inputs over and the second before reading results.) */
int sr3_wait_for_stream(sr3_ctx *ctx, void *other_stream);
int sr3_stream_wait_for_ctx(sr3_ctx *ctx, void *other_stream);
/* Arithmetic of the convolutions: 0 = exact f32 on v_mfma_f32_32x32x2_f32 (default; bit-faithful
 * fp32 products), 1 = split-f16 ("f16x3"): operands stored as hi + lo halfs, three
 * v_mfma_f32_32x32x16_f16 per product with fp32 accumulation — fp32-equivalent accuracy (error of
 * the same size as fp32 accumulation itself) at ~5x the matrix rate.
 * 2 = "f16f8": mode 1 with the two CORRECTION products (x_lo*w_hi + x_hi*w_lo) of the MFMA-bound 3x3 convs (32x32- and
 * 16x16-pixel levels at full batch) on the fp8 matrix path: operands additionally stored as OCP e4m3 with power-of-two
 * scales, one v_mfma_scale_f32_32x32x64_f8f6f4 instead of four f16 MFMAs per 32x32 tile and K-step. Correction terms
 * carry 2^-11 of the product, so 3 mantissa bits there cost ~2^-16 relative: 5e-5..7e-5 from the reference over whole
 * sampler runs (bar 1e-3) instead of 4e-6 if EVERY conv took that path (CPU emulation); measured with the eligible
 * layers: 1e-5. Activations beyond 448 (e4m3) in those layers raise the range flag, as 65504 does in mode 1; the default
 * policy then repeats the work in mode 1 first, in f32 if that overflows too. No reference counterpart (unet.py
 * computes in fp32). */
int sr3_set_precision(sr3_ctx *ctx, int prec);
/* 1 when mode 2 runs a 3x3 / stride-1 conv of this shape with fp8 correction products, else 0 (tests, tools) */
int sr3_conv_f8_supported(int B, int H, int W, int Cout, int Cin);

/* ---- weights: reference state_dict names and layouts ------------------------------------- */

/* replaces: nn.Module.state_dict() key enumeration for `denoise_fn.*`
 * (names are relative to the UNet, e.g. "downs.1.res_block.block1.block.3.weight"). */
int sr3_num_params(sr3_ctx *ctx);
int sr3_param_info(sr3_ctx *ctx, int index, char *name, int name_cap, int64_t *shape4, int *ndim);
/* replaces: load_state_dict for one tensor (lib/trainer_temp.py:175-179). `host` is fp32 in the
 * reference layout (Conv2d OIHW, Linear [out,in], vectors [C]); repacked to kernel layout inside. */
int sr3_load_weight(sr3_ctx *ctx, const char *name, const float *host, const int64_t *shape,
                    int ndim);
/* number of parameters that have never been loaded (0 = ready) */
int sr3_weights_missing(sr3_ctx *ctx);

/* ---- the denoiser: UNet.forward(x, noise_level)  (unet.py:235-265) ------------------------ */

/* x_dev: [B, in_channel, H, W] NCHW; noise_level_dev: [B]; out_dev: [B, out_channel, H, W]. */
int sr3_unet_forward(sr3_ctx *ctx, const float *x_dev, const float *noise_level_dev, int B, int H,
                     int W, float *out_dev);

/* ---- the sampler: GaussianDiffusion.{set_new_noise_schedule,p_sample_loop} ---------------- */

/* replaces: the fp32 buffers registered by set_new_noise_schedule (diffusion.py:93-142). The host
 * computes them in float64 exactly as the reference does and passes the fp32 casts:
 *   noise_level[T+1]  = float32(sqrt_alphas_cumprod_prev)      (diffusion.py:108-109,166-167)
 *   recip[T]          = sqrt_recip_alphas_cumprod               (:125-126)
 *   recipm1[T]        = sqrt_recipm1_alphas_cumprod             (:127-128)
 *   logvar[T]         = posterior_log_variance_clipped          (:137-138)
 *   coef1[T], coef2[T]= posterior_mean_coef1/2                  (:139-142) */
int sr3_set_schedule(sr3_ctx *ctx, int T, const float *noise_level, const float *recip,
                     const float *recipm1, const float *logvar, const float *coef1,
                     const float *coef2);

/* replaces: p_sample_loop (diffusion.py:189-215) for a whole batch.
 *   cond_dev   [B,3,H,W] conditioning image (x_in) or NULL for the unconditional branch (:193-201)
 *   noise_dev  NULL -> device Philox4x32-10 + Box-Muller keyed by (seed, image index+image_offset,
 *              draw, element); else [(T), B, C, H, W]: slab 0 is the initial image (torch.randn,
 *              :205), slab k (1..T-1) is the randn_like of loop iteration k-1 (t = T-k, :186)
 *   out_dev    [B,C,H,W] final images (every image, not only ret_img[-1])
 *   frames_dev NULL or [n_frames,B,C,H,W]: the image after every step i with i % sample_inter == 0
 *              (:192,209-211), sample_inter = 1 | (T/10); n_frames = sr3_num_frames(ctx)
 * Stream-ordered. In f32 mode it returns after enqueueing; in split-f16 mode it synchronises every
 * T/10 steps and at the end to read the range-check flag (see sr3_range_check below). */
int sr3_sample(sr3_ctx *ctx, const float *cond_dev, int B, int H, int W, const float *noise_dev,
               uint64_t seed, uint64_t image_offset, float *out_dev, float *frames_dev);
int sr3_num_frames(sr3_ctx *ctx);
/* Largest batch ONE sr3_sample / sr3_unet_forward call takes at H x W (every activation tensor below 4 GiB: 32-bit DMA
 * offsets; ~250 images at 128x128 with the yml UNet). Larger batches — the reference's validation loop is 15 samples x N
 * images, lib/trainer_temp.py:441-446; BASELINE configs[3] is 512 images — are run as equal chunks with image_offset
 * advancing, which the Python facade does by itself (GaussianDiffusion.sample_batch). < 0: error. */
int sr3_max_batch(sr3_ctx *ctx, int H, int W);
/* One p_sample step t on the library-resident state (used by bench.py to time exact step counts):
 * sr3_sample_begin loads cond + initial noise, sr3_sample_step runs step t, sr3_sample_end copies
 * the current image out. noise_slab_dev may be NULL (Philox). */
int sr3_sample_begin(sr3_ctx *ctx, const float *cond_dev, int B, int H, int W,
                     const float *init_noise_dev, uint64_t seed, uint64_t image_offset);
int sr3_sample_step(sr3_ctx *ctx, int t, const float *noise_slab_dev);
int sr3_sample_end(sr3_ctx *ctx, float *out_dev);
/* Range check of the split-f16 arithmetic (sr3_set_precision(ctx, 1)): the reference computes in
 * fp32 (diffusion.py:164-180, unet.py:235-265) and has no such limit, so a value that does not fit
 * the hi + lo fp16 operand format (|v| > 65504, or a NaN) must never pass silently. Every kernel that stores
 * or builds that format (conv epilogues and split-K fix-ups, GroupNorm apply, attention, the final conv's
 * on-the-fly split, state packing and the DDPM update's packed state) raises a device flag instead of clamping.
 *   sr3_unet_forward, sr3_sample — the calls that own their inputs — FINISH the call like the reference would:
 *     sr3_unet_forward evaluates the forward again in exact f32; sr3_sample reads the flag every
 *     T/10 steps (one stream synchronisation each), keeps a copy of the sampler state of the last
 *     in-range boundary and, when the flag trips, replays from that boundary to the end in exact f32
 *     (noise draws and frame slots are functions of t, so the replay is exact). Both then return
 *     SR3_OK_F32_FALLBACK (> 0) with sr3_last_warning() set; sr3_fallback_calls counts them.
 *   sr3_set_range_policy(ctx, 1) ("strict") restores the failing behaviour: those calls return < 0
 *     and the message names the f32 mode as the remedy.
 *   sr3_sample_end (step API: the caller owns the noise slabs, the library cannot replay) always
 *     fails on overflow; sr3_range_check does the same on demand between sr3_sample_step calls:
 *     0 = in range, < 0 = overflow since the last check (flag cleared). Both synchronise the stream.
 *   In mode 2 ("f16f8") the flag is also raised by an activation beyond the fp8 operand range (|v| > 448) in a conv on
 *     the fp8 correction path; the message then names mode 1 (f16x3) as the first remedy, mode 0 as the second, and the
 *     default policy of sr3_unet_forward / sr3_sample retries in exactly that order.
 *   The same flag word carries the "in-place split-K wait gave up" bit (SR3_OK_REPLAYED above): sr3_unet_forward and
 *     sr3_sample replay by themselves; sr3_sample_end / sr3_range_check fail with a message that says to repeat the call. */
int sr3_range_check(sr3_ctx *ctx);
int sr3_set_range_policy(sr3_ctx *ctx, int strict);
int sr3_fallback_calls(sr3_ctx *ctx);
int sr3_replay_calls(sr3_ctx *ctx);
/* TEST HOOK (tests/test_gpu_round4.py): device address of the context's flag word (bit 0: range overflow, bit 1: an
 * in-place split-K wait gave up), so that a test kernel on another stream can raise a bit in the middle of a running
 * sr3_sample call and the replay logic is exercised deterministically. Not for production use. */
void *sr3_test_flag_address(sr3_ctx *ctx);

/* The documented CPU twin of the device RNG is oracle/philox.py; this dumps the device stream for
 * comparison: n floats of draw `draw` for image `image`. */
int sr3_philox_normal(sr3_ctx *ctx, uint64_t seed, uint64_t image, uint32_t draw, int n,
                      float *out_dev);

/* ---- measurement ------------------------------------------------------------------------- */

/* When enabled every kernel launch is bracketed by HIP events on the context's stream and
 * accumulated per kernel family. */
int sr3_profile_enable(sr3_ctx *ctx, int on);
int sr3_profile_reset(sr3_ctx *ctx);
/* family: 0 conv_igemm, 1 groupnorm, 2 attention, 3 embed, 4 ddpm_update/layout.
 * Synchronises the stream. flops = algorithmic 2*MAC of the launches (0 for non-GEMM families). */
int sr3_profile_get(sr3_ctx *ctx, int family, double *total_ms, int64_t *launches, double *flops);
#define SR3_N_FAMILIES 5
/* per distinct conv launch shape: launches, total/avg ms, GFLOP per launch, TFLOP/s (CSV) */
int sr3_profile_dump_csv(sr3_ctx *ctx, const char *path);

/* Kernel micro-benchmark: average ms of `iters` launches of one conv shape on scratch buffers
 * and, in *apply_ms, of the
 * GroupNorm apply pass that precedes it (mode 0 copy, 1 affine, 2 affine + Swish). */
int sr3_bench_conv(sr3_ctx *ctx, int B, int Hin, int Win, int C0, int C1, int Cout, int ks, int stride,
                   int up2, int mode, int with_resid, int with_chan_bias, int iters, float *avg_ms,
                   float *apply_ms);

/* ---- single ops through the same kernels (parity tests call these) ------------------------ */

/* Conv2d over NHWC device tensors. in1_dev may be NULL (C1 = 0); channel order is in0 ‖ in1
 * (torch.cat((x, skip), 1), unet.py:261). weight_host is OIHW [Cout, C0+C1, ks, ks], bias_host
 * NULL or [Cout]. up2 = nearest x2 upsample before the conv (unet.py:58-65); stride 1|2
 * (unet.py:68-74). gn_scale/gn_shift NULL or [B, C0+C1]: per (image, channel) affine applied to the
 * input before zero padding, followed by Swish if `swish`. chan_bias_dev NULL or [B, Cout]
 * (FeatureWiseAffine, unet.py:34-50); resid_dev NULL or [B,Hout,Wout,Cout]. */
int sr3_op_conv2d(sr3_ctx *ctx, const float *in0_dev, int C0, const float *in1_dev, int C1, int B,
                  int Hin, int Win, const float *weight_host, const float *bias_host, int Cout,
                  int ks, int stride, int up2, const float *gn_scale_dev,
                  const float *gn_shift_dev, int swish, const float *chan_bias_dev,
                  const float *resid_dev, float *out_dev);
/* GroupNorm statistics folded with the affine: scale[b,c] = rstd*gamma[c],
 * shift[b,c] = beta[c] - mean*rstd*gamma[c]  (torch GroupNorm, eps 1e-5; unet.py:84,119). */
int sr3_op_groupnorm_affine(sr3_ctx *ctx, const float *in0_dev, int C0, const float *in1_dev,
                            int C1, int B, int H, int W, int groups, const float *gamma_host,
                            const float *beta_host, float *scale_dev, float *shift_dev);
/* SelfAttention core (unet.py:132-139): qkv_dev [B, N, 3C] (q|k|v along the last axis) -> out
 * [B, N, C]; softmax(q.k / sqrt(C)) v, one head. */
int sr3_op_attention(sr3_ctx *ctx, const float *qkv_dev, int B, int N, int C, float *out_dev);
/* noise_level_mlp + every FeatureWiseAffine linear (unet.py:23-31,179-184,39,49): noise_level_dev
 * [B] -> chan_bias_dev [B, total] where total = sr3_chan_bias_total(ctx) (blocks in module order). */
int sr3_op_noise_embed(sr3_ctx *ctx, const float *noise_level_dev, int B, float *temb_dev,
                       float *chan_bias_dev);
int sr3_chan_bias_total(sr3_ctx *ctx);
/* NCHW <-> NHWC copies */
int sr3_op_nchw_to_nhwc(sr3_ctx *ctx, const float *in_dev, int B, int C, int H, int W,
                        float *out_dev);
int sr3_op_nhwc_to_nchw(sr3_ctx *ctx, const float *in_dev, int B, int C, int H, int W,
                        float *out_dev);

/* ---- pre-processing in front of the sampler ------------------------------------------------- */

/* replaces: the PIL 8-bit bicubic upsample that builds the conditioning image
 * (datasets/tool/prepare_data.py:24-47: Image.resize(size, BICUBIC)) followed by ToTensor and
 * x*2-1 (datasets/util.py:76-83). in: uint8 [B,Hin,Win,3] (HWC, RGB); out: fp32 [B,3,Hout,Wout]
 * in [-1,1]; out_u8 (optional) the resized uint8 image. Bit-exact with Pillow 12.2. Synchronous. */
int sr3_preprocess_bicubic(sr3_ctx *ctx, const uint8_t *in_hwc_dev, int B, int Hin, int Win, int Hout,
                           int Wout, float *out_nchw_dev, uint8_t *out_u8_hwc_dev);

/* ---- post-processing behind the sampler ------------------------------------------------------ */

/* replaces the host chain between the sampler and the MICA / ArcFace encoder
 * (model/sr3d/model.py:372-386 and :462-471):
 *   img_u8  = Metrics.tensor2img(SR)                     core/metrics.py:16-42  [B,H,W,3] RGB
 *   up_u8   = cv2.resize(img_u8, (up, up))               model/sr3d/model.py:380 [B,up,up,3]
 *   images  = up_u8.transpose(2,0,1) / 255               model/sr3d/model.py:383-385 [B,3,up,up] fp32
 *   arcface = cv2.dnn.blobFromImages([up_u8], 1/127.5, (blob, blob), 127.5, swapRB=True)[0]
 *                                                         model/sr3d/model.py:127-131 [B,3,blob,blob]
 * sr_nchw: fp32 [B,3,H,W] in [-1,1] (the sampler's output). Every output pointer is optional
 * (null = not wanted); all are device pointers. up == 0 skips the resize (arcface is then built
 * from img_u8). cv2's 8-bit INTER_LINEAR (and its scale-2 INTER_AREA shortcut inside
 * blobFromImages) is restated from OpenCV 4.x resize.cpp — parity unpinned, cv2 is not installed.
 * Synchronous. */
int sr3_postprocess_u8(sr3_ctx *ctx, const float *sr_nchw_dev, int B, int H, int W, int up, int blob,
                       uint8_t *img_u8_dev, uint8_t *up_u8_dev, float *images_dev, float *arcface_dev);

/* replaces the tensor chain of model3 (model/sr3d/model.py:474-483):
 *   arcface = create_tensor_blob(tensor2tensor_img(SR) * 255)   model/sr3d/model.py:105-124,
 *                                                                core/metrics.py:44-50
 * i.e. clamp, [0,255] scaling, (x-127.5)/127.5, torch bilinear resize (align_corners=False) to
 * blob x blob, RGB->BGR. out: fp32 [B,3,blob,blob]. Asynchronous on the context's stream. */
int sr3_postprocess_tensor_blob(sr3_ctx *ctx, const float *sr_nchw_dev, int B, int H, int W, int blob,
                                float *arcface_dev);

/* ---- device memory helpers (so hosts without torch can drive the library) ---------------- */
int sr3_dev_malloc(sr3_ctx *ctx, uint64_t bytes, void **out_dev);
int sr3_dev_free(sr3_ctx *ctx, void *dev);
int sr3_memcpy_h2d(sr3_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes);
int sr3_memcpy_d2h(sr3_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes);
/* bytes of device memory the context currently holds (weights + workspace) */
uint64_t sr3_device_bytes(sr3_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SR3HIP_H */
