// Internal declarations shared by the HIP translation units of libsr3hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

namespace sr3 {

// ---- environment switches ---------------------------------------------------------------------------
// PRODUCT switches (INTEGRATION.md lists them; each is read once per process and then baked into captured graphs):
//   SR3_NO_GRAPH=1          every kernel launched individually, no hipGraph replay (hosts that cannot capture)
//   SR3_NO_HALO=1           generic implicit-GEMM kernel instead of the x-halo kernels (safety switch; slower)
//   SR3_HALO_SPLITS=0|2|4   in-place split-K of deep-K convs on 128x128 x-halo tiles: off / forced
//   SR3_NO_INPLACE_SPLIT=1  split-K always as conv + reduce kernel
// Everything else is an A/B switch of the development build (-DSR3_EXPERIMENTS, build.py --experiments): the product
// library does not read those variables at all, so a stray one cannot change kernels or numerics.
inline int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#ifdef SR3_EXPERIMENTS
inline int exp_int(const char *name, int dflt) { return env_int(name, dflt); }
inline double exp_double(const char *name, double dflt) { const char *e = getenv(name); return e ? atof(e) : dflt; }
#else
inline int exp_int(const char *, int dflt) { return dflt; }
inline double exp_double(const char *, double dflt) { return dflt; }
#endif

// Activation tensor: NHWC fp32 with an optional 1-pixel zero border ("pad") stored around every
// image, so a 3x3 window never needs a bounds check: pixel (n, y, x) lives at
//   p + (((n * (H + 2*pad)) + y + pad) * (W + 2*pad) + x + pad) * C.
// Borders are zeroed once when the workspace is created and never written again.
struct TDesc {
    float *p = nullptr;
    int C = 0, H = 0, W = 0, pad = 0;
    __host__ __device__ int Hp() const { return H + 2 * pad; }
    __host__ __device__ int Wp() const { return W + 2 * pad; }
    __host__ __device__ size_t pix(int n, int y, int x) const {
        return ((size_t)n * Hp() + y + pad) * Wp() + x + pad;
    }
    size_t floats(int B) const { return (size_t)B * Hp() * Wp() * C; }
};

// ---- conv as implicit GEMM (kernels_conv.hip) -------------------------------------------------
// Input = channel concatenation in0 ‖ in1 (reference unet.py:261), already activated where the
// reference applies GroupNorm+Swish first (launch_gn_apply), seen through an optional nearest x2
// upsample (unet.py:61). 3x3 convs require pad == 1 on the inputs.
struct ConvParams {
    TDesc in0, in1;         // in1.p == nullptr if none; same H, W, pad as in0; C multiples of 32
    int B = 0;
    int Hout = 0, Wout = 0;
    int ks = 3, stride = 1, up2 = 0;   // ks 1 | 2 | 3; ks == 2 (stride 1) reads the window (oy+org_y+{0,1}, ox+org_x+{0,1})
    // sub-pixel phases of the upsample conv (launch_conv_up2): window origin shift in padded
    // coordinates, and the output pixel written for window (oy, ox) is (oy*out_step + out_oy, ...)
    int org_y = 0, org_x = 0;
    int out_step = 1, out_oy = 0, out_ox = 0;
    int stats_slice0 = 0;              // first statistics slice of this launch
    // log2(Hout*Wout) / log2(Wout) when they are powers of two (set by launch_conv, -1 otherwise): the
    // per-lane address set-up of a tile then uses shifts instead of integer divisions
    int hw_shift = -1, w_shift = -1;
    // phases > 1 (= 4): one launch runs the four sub-pixel phases of an upsample conv on blockIdx.z
    // (grid.y of the split-K reduce): phase ph = (py, px) sets org_* = out_o* = (py, px) and advances
    // w by ph * phase_w_stride floats, stats_slice0 by ph * phase_slices, part by ph * phase_part_stride
    int phases = 1, phase_slices = 0;
    size_t phase_w_stride = 0, phase_part_stride = 0;
    const float *w = nullptr;          // packed [ks*ks][Cout][Cin]
    // optional fused 1x1 term (ResnetBlock.res_conv, unet.py:102-103,110): out += in2 (*) w2, read at
    // the output pixel; same precision format and (for prec 1) the same weight scale as w
    TDesc in2;                         // p == nullptr if none; same H, W as the output
    TDesc in2b;                        // optional second part: the 1x1 input is in2 ‖ in2b (x ‖ skip, unet.py:261)
    const float *w2 = nullptr;         // packed [Cout][in2.C + in2b.C]
    const float *bias = nullptr;       // [Cout] or null
    const float *chan_bias = nullptr;  // [B][chan_bias_stride] (+ offset applied by caller) or null
    int chan_bias_stride = 0;
    // optional fused GroupNorm statistics of the OUTPUT: per (image, M-tile-in-image, channel)
    // {sum, sum of squares} in fp64 at stats[((n * stats_slices + slice) * Cout + c) * 2]; requires
    // Hout*Wout % BM == 0 for the tile the launcher picks (launch_conv_tile_m tells)
    double *stats = nullptr;
    int stats_slices = 0;
    TDesc resid;            // p == nullptr if none; same geometry as out
    int resid_split = 0;    // 1: resid is stored in the split-f16 format (hi + lo), not fp32
    TDesc out;              // C = Cout
    // optional twin of the output in the split-f16 input format (same geometry as out): lets the
    // next conv read this tensor directly (Down/Upsample input, fused res_conv operand)
    TDesc out_split;
    int out_f32 = 1;        // 0: only out_split is written (out.p then only provides the geometry)
    // prec 0: exact f32 MFMA; inputs / weights are fp32.
    // prec 1: split-f16 ("f16x3"): every 32-channel chunk of the inputs and of the packed weights is
    //         stored as 32 hi halfs | 32 lo halfs (x = hi + lo to ~2^-22), the product is
    //         hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation; weights are
    //         pre-scaled by 2^k (w_unscale = 2^-k is applied to the accumulator in the epilogue).
    int prec = 0;
    // f8 = 1 (prec 1 only; "f16f8" mode, conv_f8_supported() tells for which shapes): in0 / in1 and w are in the F8C
    // variant of the split format — per 32-channel chunk 32 hi halfs | 32 x e4m3(lo * 2^SR3_F8_XL) | 32 x e4m3(hi * 2^SR3_F8_XH)
    // for activations, 32 hi halfs | 32 x e4m3(hi * 2^SR3_F8_WH) | 32 x e4m3(lo * 2^SR3_F8_WL) for weights — and the two
    // correction products xl*wh + xh*wl run as ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile and K-step (half the
    // cycles of the four f16 MFMAs they replace; ~6e-5 instead of ~4e-6 from the reference over a sampler run).
    // in2 / in2b / w2 (fused 1x1 term) stay in the plain split format.
    int f8 = 0;
    float w_unscale = 1.0f;
    // split-K for small problems (few tiles, deep K): `splits` blocks share one output tile, each
    // reducing a contiguous range of 32-channel chunks into part[split][M][Cout]; a second kernel
    // adds the partials and applies the epilogue. splits <= 1: single pass.
    int splits = 1;
    float *part = nullptr;
    // in-place split-K (conv_split_inplace() tells when launch_conv uses it): one zero-initialised counter per
    // output tile and sub-pixel phase; the last block to arrive at a tile adds the partials and runs the
    // epilogue itself (no reduce kernel; fused statistics in the unsplit layout: one slice per M-tile).
    // nullptr: always the two-kernel form.
    unsigned *tile_cnt = nullptr;
    // 1: never the in-place split-K of the 128x128 x-halo tile (conv_halo_splits) — set for the rest of a context's
    // life once one of its bounded inter-block waits gave up (SR3_FLAG_GNF_TIMEOUT): the conv then runs unsplit on the
    // generic 64x64 tile, whose blocks never wait for each other
    int no_halo_split = 0;
    // 1: in0 is stored FRAGMENT-MAJOR (fm_* below) — the input layout of the weights-stationary kernel
    // (kernels_conv_ws.hip); launch_conv runs such a conv on that kernel or reports an error, never on another one
    int in_fm = 0;
    // split-f16 range check: any value stored in the split format (out_split) with |v| > 65504 (or
    // non-finite) sets *ovf = 1; the API call that ran the launch then fails (never a silent clamp)
    int *ovf = nullptr;
    // Producer-side GroupNorm of the OUTPUT (gnf_gamma != nullptr): the output h of a ResnetBlock's first conv is read
    // by nothing but block2's GroupNorm + Swish (unet.py:105-110, 84-87), so the conv normalises it ITSELF and stores
    // swish(scale * h + shift) straight in the next conv's operand format (out_split = that conv's activated input,
    // out_f32 = 0): h never reaches memory and no apply pass runs. Inside the launch the blocks of one (image, N-tile)
    // group — they hold whole GroupNorm groups of that image between them — publish their fp64 partial statistics
    // (ConvParams::stats, one slice per M-tile, write-through stores), count themselves on gnf_cnt[image * tilesN + nt];
    // the block that arrives last folds the slices in slice order (bit-identical whichever block it is), publishes
    // scale / shift in gnf_ab and signals; the others poll the counter (one lane, s_sleep) and fetch gnf_ab.
    // Requirements (conv_gnf_supported): split-f16 x-halo kernel with 16x16x32 consumers, whole 128-row tiles inside one
    // image, whole groups inside an N-tile, no split-K, and every group's blocks co-resident on their XCD (the block
    // order is chosen for it: gnf_band).
    const float *gnf_gamma = nullptr, *gnf_beta = nullptr;   // [Cout] of the GroupNorm that FOLLOWS this conv
    int gnf_groups = 0;
    float gnf_eps = 1e-5f;
    unsigned *gnf_cnt = nullptr;       // per (image, N-tile) 64 words: [0] counter, [32] ready word (two cache lines), zero between launches
    float *gnf_ab = nullptr;           // [B][Cout][2] scale | shift
    int gnf_band = 0;                  // > 0: M-tiles of an image per XCD (band block order, set by launch_conv)
    int dbg = 0;            // timing experiments only (tools/conv_bench.py); 0 in product code
};
// bit raised in *ConvParams::ovf when a bounded inter-block wait gives up (in-place split-K of the x-halo kernel; the
// producer-side GroupNorm experiment): the launch's result is invalid; the API replays the work with
// ConvParams::no_halo_split set (sr3_api.hip: range_read)
constexpr int SR3_FLAG_GNF_TIMEOUT = 2;
// bound of such a wait in ticks of s_memrealtime (constant 100 MHz on gfx950, MI355X_MICROARCH.md): 2^19 = 5.2 ms — the
// blocks of one tile are dispatched back to back and the whole conv takes < 0.1 ms on an idle chip
constexpr long long SR3_WAIT_TICKS = 1ll << 19;
// largest magnitude the split-f16 format (hi + lo, both fp16) can hold
constexpr float SPLIT_F16_MAX = 65504.0f;

// ---- split-f16 twin stores from MFMA accumulators (conv epilogues) -----------------------------------
// Two lanes that hold neighbouring channels (lane ^ 1) exchange halfs so that each lane stores ONE 4-byte
// word: the even lane both hi halfs, the odd lane both lo halfs. own = hi | lo << 16, oth = the partner's
// (DPP quad_perm [1,0,3,2]); v_perm_b32 picks {own.hi16?..} by a per-lane selector:
//   even: [own.b0, own.b1, oth.b0, oth.b1]   odd: [oth.b2, oth.b3, own.b2, own.b3]
// (v_perm_b32 D = bytes of {src0 = oth : bytes 4..7, src1 = own : bytes 0..3}).
// range: running unsigned max of (hi bits & 0x7C00) — 0x7C00 at the end means some hi half was inf / nan,
// i.e. |value| beyond what hi + lo can hold (>= 65520) or not finite: two VALU operations per value.
__device__ __forceinline__ unsigned split_pair_selector(bool odd) { return odd ? 0x03020706u : 0x05040100u; }
__device__ __forceinline__ unsigned split_pair_word(float v, unsigned selector, unsigned &range) {
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const unsigned own = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    const unsigned e = own & 0x7C00u;
    range = e > range ? e : range;
    const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    return __builtin_amdgcn_perm(oth, own, selector);
}
__device__ __forceinline__ bool split_range_overflow(unsigned range) { return range == 0x7C00u; }
// F8C operand scaling (powers of two; activations beyond e4m3's 448 raise the range flag). The CPU emulation of these
// formats (tests/emulate_operand_formats.py) gives the same sampler error for activation exponents 0, 2 and 4.
constexpr int SR3_F8_XH = 0, SR3_F8_XL = 11;      // xh8 = e4m3(xh), xl8 = e4m3(xl * 2048)   (|xl| <= 2^-11 |xh|: <= 256 at the limit)
constexpr int SR3_F8_WH = -3, SR3_F8_WL = 9;      // weights are pre-scaled into [1024, 2048): wh8 < 256, |wl| <= 0.5 -> wl8 <= 256
constexpr float SPLIT_F8_MAX = 448.0f;
// true when launch_conv runs this 3x3 / stride-1 split-f16 conv with the fp8 correction products if ConvParams::f8 is set
// (the caller then writes the conv's input with split format 2 and passes the F8C weights)
bool conv_f8_supported(int B, int H, int W, int Cout, int Cin);
// split weights [chunks][32 hi | 32 lo] -> F8C weights [chunks][32 hi | 32 wh8 | 32 wl8] (device to device)
void launch_make_f8_weights(const float *split, float *dst, size_t chunks, hipStream_t s);
void launch_conv(const ConvParams &p, hipStream_t s);
// launch_conv never aborts the process: a request it cannot honour (a caller / library bug) launches nothing and leaves
// a message here; returns it once (nullptr if none) — the C-ABI entry points fail the call with it
const char *conv_take_error();
// Weights-stationary kernel for the 64 -> 64 channel 3x3 convs of the full-resolution level (kernels_conv_ws.hip). Its
// waves load their MFMA A fragments straight from memory, so the activated input is stored FRAGMENT-MAJOR ("FM"): the
// zero-bordered tensor [B][H + 2][W + 2][C] regrouped per padded row into groups of 16 consecutive pixels, per group and
// 32-channel chunk one 2 KB block  [hi | lo][q = 0..3][pixel 0..15][16 B]  — the 16 bytes (8 halfs of channel octet q) that
// lane (pixel, q) of a 16x16x32 MFMA holds, so one fragment of 16 pixels is 1 KB of consecutive memory (a wave's load is
// lane-linear: perfectly coalesced; the pixel-major layout costs the texture path 2.4x, profiles/README.md finding 66).
// Rows are padded to whole groups; borders and group padding stay zero like every activation border.
//   byte offset of (n, padded y, padded x, channel c), hi half of its octet:
//     ((((n * (H + 2) + yp) * G + (xp >> 4)) * (C / 32) + (c >> 5)) * 2048 + ((c & 31) >> 3) * 256 + (xp & 15) * 16,  lo: + 1024
__host__ __device__ inline int fm_groups(int W) { return (W + 2 + 15) >> 4; }                      // G: 16-pixel groups per padded row
inline size_t fm_floats(int B, int C, int H, int W) { return (size_t)B * (H + 2) * fm_groups(W) * (C / 32) * 512; }
// shape test shared by the engine (which writes the conv's input in FM form) and launch_conv: 3x3 / stride 1 / split-f16
// conv of this shape runs on the weights-stationary kernel when its input is FM (ConvParams::in_fm)
bool conv_ws_shape_ok(int B, int H, int W, int Cin, int Cout);
bool conv_ws_supported(const ConvParams &p);
void launch_conv_ws(const ConvParams &p, hipStream_t s);
// Upsample (nearest x2) + conv3x3 (unet.py:58-65) as four sub-pixel phases: output pixels of
// parity (py, px) see only a 2x2 window of the low-resolution input, with the 3x3 taps that land
// on the same source pixel pre-added (make_up2_phase_weights) — 16 instead of 36 MACs per
// low-resolution pixel, channel pair and 2x2 output block. p describes the conv at the OUTPUT
// resolution (Hout, Wout = 2H, 2W; in0 = low-resolution input); p.w = phase weights
// [4 phases][4 taps][Cout][CinPad]; stats_slices must be 4 * (H*W / conv_tile_m(B*H*W, Cout)).
void launch_conv_up2(const ConvParams &p, hipStream_t s);
// packed [9][Cout][CinPad] -> [py*2+px][dy2*2+dx2][Cout][CinPad]
void make_up2_phase_weights(const float *packed9, int Cout, int CinPad, float *dst);
// BM of the tile launch_conv will use for this problem (so callers can size / enable fused stats)
int conv_tile_m(long M, int Cout);
// number of K-splits launch_conv wants for this problem (1 = none); Cin per tap, multiple of 32
int conv_splits(long M, int Cout, int Cin);
// true when a split conv of this shape adds its partials in place (ConvParams::tile_cnt given): its fused statistics
// then have the unsplit layout, HWo / conv_tile_m() slices per image; HWo = pixels of one image (and phase)
bool conv_split_inplace(long M, int HWo, int Cout, int Cin, int phases = 1);
// true when launch_conv can run this 3x3 / stride-1 / split-f16 conv with the producer-side GroupNorm of its output
// (ConvParams::gnf_*): p as it will be launched, without the gnf fields
bool conv_gnf_supported(const ConvParams &p, int groups);
constexpr int CONV_GNF_COUNTERS = 8192;     // capacity of ConvParams::gnf_cnt (images x N-tiles of one launch)
// K-splits of the 128x128 x-halo tile for deep-K 3x3 / stride-1 split-f16 convs over few tiles (kernels_conv.hip); <= 1: none
int conv_halo_splits(long M, int H, int W, int Cout, int Cin);
constexpr int CONV_TILE_COUNTERS = 8192;    // capacity of ConvParams::tile_cnt (tiles x phases of one launch)
// a split conv's fused GroupNorm statistics come out of its reduce pass: slices per image (and per
// sub-pixel phase) for an output of HWo pixels; the caller sizes / strides ConvParams::stats with it
// (0: not available for this shape)
int splitk_stats_slices(int HWo, int Cout);
// host helper: OIHW -> [tap][Cout][CinPad] (zero pad input channels up to CinPad)
void pack_conv_weight(const float *oihw, int Cout, int Cin, int ks, int CinPad, float *dst);
// host helper: fp32 packed weights -> split-f16 layout (same byte size), returns the unscale factor
float split_conv_weight(const float *packed, size_t rows, int CinPad, float *dst);

// ---- GroupNorm (kernels_misc.hip) -------------------------------------------------------------
// statistics over the virtual concatenation in0 ‖ in1 -> folded affine scale/shift [B][C]
size_t gn_workspace_floats(int B, int c_max);   // c_max = widest normalised tensor
void launch_groupnorm_affine(const TDesc &in0, const TDesc &in1, int B, int groups, const float *gamma,
                             const float *beta, float eps, float *part, float *scale, float *shift,
                             hipStream_t s);
// statistics already accumulated by the producing convs (ConvParams::stats): only the reduce
struct StatsRef { const double *p = nullptr; int slices = 0; };
void launch_groupnorm_finalize(const StatsRef &s0, int C0, const StatsRef &s1, int C1, int B, int HW, int groups,
                               const float *gamma, const float *beta, float eps, float *scale, float *shift,
                               hipStream_t s);
// out[n,y,x,:] = act(concat(in0,in1)[n,y,x,:] * scale[n,:] + shift[n,:]); mode 0 copy, 1 affine,
// 2 affine + Swish. out.C == in0.C + in1.C; writes the interior only. split = 1 stores every
// 32-channel chunk as 32 hi halfs | 32 lo halfs (the conv's prec 1 input format), split = 2 as
// 32 hi halfs | 32 x e4m3(lo * 2^SR3_F8_XL) | 32 x e4m3(hi * 2^SR3_F8_XH) (ConvParams::f8; `raw` stays format 1),
// split = 3 as format 1 in the fragment-major layout (fm_*: out.p must hold fm_floats(); ConvParams::in_fm).
// raw (optional, p != nullptr): additionally stores the un-normalised concatenation in the same
// format (the input of a fused res_conv).
// in_split: bit 0 / bit 1 = in0 / in1 is itself stored in the split-f16 format (split-only tensors)
// ovf: range-check flag of the split format (see ConvParams::ovf), may be null
void launch_gn_apply(const TDesc &in0, const TDesc &in1, int B, const float *scale, const float *shift,
                     int mode, int split, const TDesc &out, hipStream_t s, const TDesc &raw = TDesc(),
                     int in_split = 0, int *ovf = nullptr);
// streaming form of the apply pass for large tensors (one item per thread; scale / shift from memory)
void launch_gn_apply_rows(const TDesc &in0, const TDesc &in1, int B, const float *scale, const float *shift,
                          int mode, int split, const TDesc &out, hipStream_t s, const TDesc &raw = TDesc(),
                          int in_split = 0, int *ovf = nullptr);
// GroupNorm finalize folded into the apply pass (one launch per normalised tensor): s0 / s1 are the fp64
// partial statistics of in0 / in1 (ConvParams::stats layout; from conv epilogues or
// launch_groupnorm_partials, which returns the virtual concatenation as one source: pass it as s0
// with s1 empty)
void launch_gn_fold_apply(const TDesc &in0, const TDesc &in1, int B, const StatsRef &s0, const StatsRef &s1, int groups,
                          const float *gamma, const float *beta, float eps, int mode, int split, const TDesc &out,
                          hipStream_t s, const TDesc &raw = TDesc(), int in_split = 0, int *ovf = nullptr);
StatsRef launch_groupnorm_partials(const TDesc &in0, const TDesc &in1, int B, float *part, hipStream_t s);
// common power-of-two scale for several weight tensors: returns k with max|w| * 2^k in [1024, 2048)
int split_scale_exponent(const float *packed, size_t n);
float split_conv_weight_k(const float *packed, size_t rows, int CinPad, int k, float *dst);

// ---- attention core ----------------------------------------------------------------------------
double launch_attention(const float *qkv, int B, int N, int C, float *out, hipStream_t s);
// split-f16 form: qkv in the conv's split operand format ([B][N][3C], 32-channel chunks of hi | lo halfs);
// out (fp32 [B][N][C]) and / or out_split (the same tensor in the split format) may be null
bool attention_split_supported(int N, int C);
size_t attention_vt_floats(int B, int N, int C);     // scratch for v^T
double launch_attention_split(const float *qkv_split, float *vt, int B, int N, int C, float *out, float *out_split,
                              int *ovf, hipStream_t s);

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
// NCHW [B][C][H][W] -> channels [coff, coff+C) of dst (NHWC TDesc)
void launch_nchw_to_nhwc(const float *in, int B, int C, const TDesc &dst, int coff, hipStream_t s);
// channels [coff, coff+C) of src -> NCHW
void launch_nhwc_to_nchw(const TDesc &src, int coff, int B, int C, float *out, hipStream_t s);
// Per-step values live in device memory (written by a tiny async copy before every step), so the
// kernel arguments of a step never change and the whole step can be replayed as one hipGraph.
struct StepArgs {
    float nl;                   // noise level fed to the embedding (diffusion.py:166-167)
    float a, b, c1, c2, sigma;  // recip, recipm1, coef1, coef2, exp(0.5*logvar) (0 at t == 0)
    uint32_t draw, pad_;
    const float *noise;         // NCHW [B][C][HW] or null -> Philox
    float *frame;               // NCHW [B][C][HW] or null: copy of the updated image
    uint64_t seed, image_offset;
};
struct UpdateParams {
    float *packed = nullptr;   // optional packed split-f16 copy of the state (conv_in_kernel): 16 halfs per padded pixel
    TDesc state;        // x lives in channels [xoff, xoff+C)
    int xoff, C;        // C = image channels (3)
    TDesc eps;          // eps.C >= C
    const StepArgs *args;
    int *ovf = nullptr; // range check of the packed copy (ConvParams::ovf contract)
};
void launch_ddpm_update(const UpdateParams &p, int B, hipStream_t s);
// x <- noise (NCHW buffer or Philox draw 0) into state channels
void launch_init_state(const TDesc &state, int xoff, int C, const float *noise, uint64_t seed,
                       uint64_t image_offset, int B, hipStream_t s);
void launch_philox_normal(uint64_t seed, uint64_t image, uint32_t draw, int n, float *out,
                          hipStream_t s);

// ---- edge convolutions (kernels_edge.hip) ----------------------------------------------------------
// final_conv = GroupNorm affine + Swish + Conv3x3(C -> Cout <= 4) in one fp32 VALU kernel: x is the raw
// fp32 zero-bordered tensor, scale / shift [B][C] the folded GroupNorm (gn_finalize), wq the weights as
// [9][C][4] (pack_final_conv_weight), out [B][H][W][Cout] unpadded
bool final_conv_supported(int C, int Cout);
void pack_final_conv_weight(const float *oihw, int Cout, int C, float *dst);
void launch_final_conv(const TDesc &x, int B, const float *scale, const float *shift, const float *wq, const float *bias,
                       const TDesc &out, hipStream_t s);

// the same conv in split-f16 mode as a per-pixel [C] x [27 -> 32] MFMA GEMM + a 9-tap gather (kernels_edge.hip)
bool final_conv_mfma_supported(int C, int Cout);
size_t final_conv_mfma_weight_floats(int C);
float pack_final_conv_mfma_weight(const float *oihw, int C, float *dst_as_float);
void launch_final_conv_mfma(const TDesc &x, int B, const float *scale, const float *shift, const float *wfr, float w_unscale,
                            const float *bias, const TDesc &out, hipStream_t s, int *ovf = nullptr);
// downs.0 (Conv3x3 in_channel <= 8 -> Cout) on the packed split-f16 state: xp = [B][H+2][W+2] pixels of
// 16 halfs (8 channels hi | lo; + 16 floats of slack behind the last pixel), wci from pack_conv_in_weight
bool conv_in_supported(int Cin, int Cout, int H, int W);
float pack_conv_in_weight(const float *oihw, int Cout, int Cin, float *dst_as_float);
size_t conv_in_weight_floats(int Cout);
void launch_pack_state(const TDesc &x, int B, float *xp, hipStream_t s, int *ovf);
void launch_conv_in(const float *xp, const float *wci, const float *bias, float w_unscale, int B, const TDesc &out,
                    const TDesc &out_split, int out_f32, double *stats, int stats_slices, int *ovf, hipStream_t s);

// ---- pre-processing: PIL-exact 8-bit bicubic resize -> sampler input tensor (kernels_pre.hip) ----
} // namespace sr3
#include <vector>
namespace sr3 {
int bicubic_coeffs(int in_size, int out_size, std::vector<int> &bounds, std::vector<int> &kk);   // returns ksize
void launch_resample_h(const uint8_t *in, int B, int H, int Win, int Wout, const int *bounds, const int *kk,
                       int ksize, uint8_t *out, hipStream_t s);
// bounds == nullptr: no vertical resize (Hin == Hout), only the tensor conversion
void launch_resample_v(const uint8_t *in, int B, int Hin, int Hout, int W, const int *bounds, const int *kk,
                       int ksize, float *out_nchw, uint8_t *out_u8, hipStream_t s);

// ---- post-processing: tensor2img / cv2-style 8-bit linear resize / ArcFace blob (kernels_post.hip) ----
void cv_linear_coeffs(int in_size, int out_size, bool horizontal, std::vector<int> &ofs, std::vector<int> &ab);
void launch_tensor2img(const float *in_nchw, int B, int H, int W, uint8_t *out_hwc, hipStream_t s);
// tab = xofs[Wd] | xa[Wd][2] | yofs[Hd] | yb[Hd][2]; images (optional) = dst / 255 as [B][3][Hd][Wd]
void launch_resize_linear_u8(const uint8_t *src, int B, int Hs, int Ws, int Hd, int Wd, const int *tab, uint8_t *dst,
                             float *images, hipStream_t s);
// src is f x the blob size (f = 1 | 2); out [B][3][Hb][Wb], channels swapped, (avg - mean) * scale
void launch_blob(const uint8_t *src, int B, int Hb, int Wb, int f, float mean, float scale, float *out, hipStream_t s);
void launch_tensor_blob(const float *in_nchw, int B, int H, int W, int Hb, int Wb, float *out, hipStream_t s);

} // namespace sr3
