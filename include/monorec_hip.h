/*
 * monorec_hip.h - C ABI of libmonorec_hip.so: the MI355X (gfx950) kernels behind the MonoRec
 * cost-volume inference path.
 *
 * The reference (Brummi/MonoRec) is pure Python on PyTorch; its "FFI" for this path is the set of
 * ATen operator calls inside MonoRecModel.forward.  Each entry point below replaces one group of
 * those calls; the comment on every function names the reference lines it stands in for
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain C: raw device pointers, ints/floats and a `void* stream` (a hipStream_t); no torch types.
 *   - the caller owns every buffer; all tensors are dense fp32 NCHW on the current HIP device.
 *   - every launch is asynchronous on `stream`; nothing here allocates, frees or synchronises,
 *     so all entry points may be recorded into a hipGraph (stream capture).
 *   - return value: 0 on success, a positive hipError_t, or a negative MR_ERR_* code. Nothing throws
 *     across the boundary. mr_error_string() maps a code to text.
 */
#ifndef MONOREC_HIP_H
#define MONOREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_ABI_VERSION 19

#define MR_COMPUTE_F32  0
#define MR_COMPUTE_BF16 1
#define MR_COMPUTE_BF16X3 2

#define MR_ERR_BAD_ARGUMENT (-1)
#define MR_ERR_UNSUPPORTED  (-2)
#define MR_ERR_LDS_BUDGET   (-3)

#define MR_MAX_SOURCES 3
#define MR_MAX_FRAMES  8
#define MR_MAX_HEADS   4   /* one-channel 3x3 heads per mr_depth_heads_f32 launch (DepthModule has 4 predictors) */
#define MR_MAX_VOTE_MASKS 8   /* masks voted over by mr_pointcloud_append_f32 (the reference buffers 5) */

/* ---- activation codes for mr_conv2d_f32 (epilogue, applied after bias [+ residual]) ---- */
enum {
    MR_ACT_NONE = 0,
    MR_ACT_RELU = 1,            /* torchvision BasicBlock relu                                   */
    MR_ACT_LEAKY_RELU = 2,      /* LeakyReLU(slope = act_p0), 0 <= act_p0 <= 1 (else MR_ERR_UNSUPPORTED)  model/layers.py:303,330,391 */
    MR_ACT_SIGMOID = 3,         /* MaskModule.classifier      model/monorec/monorec_model.py:342  */
    MR_ACT_ABS_TANH_AFFINE = 4  /* |tanh(x)| then (1-p)*act_p0 + p*act_p1  monorec_model.py:556,717 */
};

/* ---- how the conv reads its (virtual) input plane from the stored source plane ---- */
enum {
    MR_IN_DIRECT = 0,     /* input == source                                                          */
    MR_IN_UPSAMPLE2 = 1,  /* input(y,x) = source(y/2,x/2): nn.Upsample(scale_factor=2) layers.py:349   */
    MR_IN_MAXPOOL2 = 2    /* input(y,x) = max of the 2x2 source block: nn.MaxPool2d(2) monorec_model.py:304 */
};

/* ---- value transform applied to in-bounds input samples while staging ---- */
enum {
    MR_TF_NONE = 0,
    MR_TF_RESNET_NORM = 1 /* ((x + 0.5) - 0.45) / 0.225  monorec_model.py:691 + :120 */
};

/*
 * One convolution launch.  Replaces nn.Conv2d / PadSameConv2d+Conv2d / Upsample+pad+Conv2d /
 * MaxPool2d+Conv2d / one output phase of ConvTranspose2d(k4,s2) (+ folded eval BatchNorm, bias,
 * residual add and activation) of model/layers.py:241-252,289-356,380-400 and the torchvision
 * ResNet-18 trunk used by model/monorec/monorec_model.py:118-129.
 *
 * Input = channel concatenation (torch.cat(..., dim=1), monorec_model.py:372-380,531,541-545) of
 * `num_src` NCHW sources sharing batch and plane size; they are read in place, never materialised.
 * Output pixel (oy,ox) of the conv grid is written to plane position
 * (oy*out_step_h + out_off_h, ox*out_step_w + out_off_w) of channel dst_channel_offset + co of a
 * (batch, dst_total_channels, dst_plane_h, dst_plane_w) tensor - step 2 is used by the four phases of
 * the transposed convolution.
 */
typedef struct mr_conv_desc {
    /* input */
    const float* src[MR_MAX_SOURCES];
    int32_t src_channels[MR_MAX_SOURCES];
    int32_t num_src;
    int32_t batch;
    int32_t src_h, src_w;            /* stored plane size of every source                      */
    int32_t in_mode;                 /* MR_IN_*                                                */
    int32_t in_transform;            /* MR_TF_*                                                */
    /* filter geometry; taps read input (oy*stride_h - pad_top + ky, ox*stride_w - pad_left + kx), zero outside */
    int32_t kh, kw, stride_h, stride_w, pad_top, pad_left;
    int32_t out_h, out_w;            /* conv grid size                                          */
    /* output placement */
    float* dst;
    int32_t out_channels;
    int32_t dst_total_channels, dst_channel_offset;
    int32_t dst_plane_h, dst_plane_w;
    int32_t out_step_h, out_step_w, out_off_h, out_off_w;
    /* parameters */
    const float* packed_weights;     /* from mr_conv_pack_weights_f32 (device copy)              */
    const float* bias;               /* out_channels floats or NULL                              */
    const float* residual;           /* same geometry as dst (incl. channel offset) or NULL      */
    int32_t activation;              /* MR_ACT_*                                                 */
    float act_p0, act_p1;
    /* schedule */
    int32_t cout_blocks_per_wg;      /* MB: 16-channel output blocks per workgroup, one of 1,2,3,4,6 */
    int32_t pixel_blocks_per_wave;   /* NB: 16-pixel row segments per wave, one of 1,2,4            */
    int32_t chunk_channels;          /* CK: input channels staged per LDS chunk: 8,16,32,64,128    */
    int32_t split_k;                 /* >= 1; > 1 needs `workspace`                               */
    float* workspace;                /* split_k * phases * batch * ceil16(out_channels) * out_h * out_w floats */
    /* optional output phases (the 4 parities of ConvTranspose2d(k=4,s=2), model/layers.py:389):
     * num_phases == 4 runs four filter sets in ONE launch; phase p uses phase_weights[p], its own
     * pad_top/pad_left and output offset, and the common kh/kw/stride/out_step.  num_phases <= 1: ignored. */
    int32_t num_phases;
    const float* phase_weights[4];
    int32_t phase_pad_top[4], phase_pad_left[4], phase_out_off_h[4], phase_out_off_w[4];
    /* schedule, continued: waves per workgroup, 4 (0 = default) or 8.  8 waves share one LDS tile of twice the
     * rows: same LDS and DMA traffic per output as 4 waves with 2x pixel_blocks_per_wave, but two waves per
     * SIMD from every workgroup, which hides the chunk-fill stalls.  8 needs the direct-read dwordx4 path
     * (in_mode DIRECT, no in_transform, src_w % 4 == 0); otherwise MR_ERR_UNSUPPORTED. */
    int32_t waves_per_wg;
    /* MR_COMPUTE_F32 (0, default): v_mfma_f32_16x16x4_f32, exact fp32 products - the path that meets the 1e-4 parity bar.
     * MR_COMPUTE_BF16: v_mfma_f32_16x16x16_bf16 - packed_weights / phase_weights from mr_conv_pack_weights_bf16, the
     * activations (still fp32 in memory) are rounded to bf16 (nearest even) as the B fragment is formed, fp32 accumulate:
     * the numerics of "bf16 weights and activations, fp32 accumulate" (BASELINE configs[4]).  chunk_channels in
     * {16, 32, 64, 128}; LDS-DMA staged inputs only (in_mode DIRECT / UPSAMPLE2, no in_transform).
     * MR_COMPUTE_BF16X3 (EXPERIMENTAL, not yet validated on hardware): fp32-class accuracy on the bf16 matrix cores - every
     * operand is split into hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits together) and a*b is evaluated as
     * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (three v_mfma_f32_16x16x16_bf16 per 16 channels instead of
     * four v_mfma_f32_16x16x4_f32, which run at 1/16 of the rate).  Weights from mr_conv_pack_weights_bf16x3; the same
     * restrictions as MR_COMPUTE_BF16.  CPU emulation over the whole network: depth error 4e-6 (fp32 path 1.3e-6, bar 1e-4). */
    int32_t compute_dtype;
    /* per-phase filter sizes (num_phases == 4): phase p sweeps phase_kh[p] x phase_kw[p] taps of phase_weights[p] (packed with
     * that size); 0 = the common kh / kw, which must be the maximum over the phases (it sizes the input tile).  This is how
     * layers.Upconv (nearest x2 -> pad(0,1,0,1) -> conv2x2, model/layers.py:349-356) runs on the LOW-resolution input: output
     * parity (py, px) is a (1+py) x (1+px) convolution with the 2x2 filter's rows / columns summed where both fall on the same
     * input pixel - 9 instead of 16 multiply-adds per 2x2 output block. */
    int32_t phase_kh[4], phase_kw[4];
    /* schedule, continued: 1 = the waves of a workgroup split K instead of the pixels - all of them sweep the same
     * pixel_blocks_per_wave blocks of 16 pixels, each every waves_per_wg-th k-step of a chunk, and the partial sums are reduced
     * through LDS in a fixed order.  waves_per_wg times more workgroups for layers with few output pixels, without the workspace
     * round trip and the finishing launch of split_k (which must be 1 here; MR_COMPUTE_F32, LDS-DMA staged inputs only). */
    int32_t k_split_waves;
} mr_conv_desc;

/* number of floats of the packed weight image for a conv with the given source split and schedule
 * (cout_blocks_per_wg, chunk_channels must equal the values later put into mr_conv_desc) */
size_t mr_conv_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                    int32_t kh, int32_t kw, int32_t cout_blocks_per_wg, int32_t chunk_channels);

/*
 * Host-side repack of an (out_channels, sum(src_channels), kh, kw) fp32 weight (nn.Conv2d layout,
 * row-major) into the MFMA A-fragment stream the kernel consumes. `dst` is host memory of
 * mr_conv_packed_weight_floats() floats; upload it once per layer.
 */
int mr_conv_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels,
                             int32_t num_src, int32_t kh, int32_t kw, int32_t cout_blocks_per_wg,
                             int32_t chunk_channels, float* dst);

/* The same two functions for MR_COMPUTE_BF16 launches: sources padded to multiples of 16 channels, weights rounded to bf16
 * (nearest even), 4 bf16 per lane and k-step (layout in csrc/conv_layout.h).  `dst` is still sized in floats. */
size_t mr_conv_packed_weight_floats_bf16(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                         int32_t kh, int32_t kw, int32_t cout_blocks_per_wg, int32_t chunk_channels);
int mr_conv_pack_weights_bf16(const float* weight, int32_t out_channels, const int32_t* src_channels,
                              int32_t num_src, int32_t kh, int32_t kw, int32_t cout_blocks_per_wg,
                              int32_t chunk_channels, float* dst);

/* ... and for MR_COMPUTE_BF16X3 launches: per lane and k-step 4 bf16 `hi` followed by 4 bf16 `lo` (16 bytes), weight image twice
 * the bf16 one (= the fp32 size for 16-aligned channel counts). */
size_t mr_conv_packed_weight_floats_bf16x3(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                           int32_t kh, int32_t kw, int32_t cout_blocks_per_wg, int32_t chunk_channels);
int mr_conv_pack_weights_bf16x3(const float* weight, int32_t out_channels, const int32_t* src_channels,
                                int32_t num_src, int32_t kh, int32_t kw, int32_t cout_blocks_per_wg,
                                int32_t chunk_channels, float* dst);

/* bytes of dynamic LDS the launch will request (for planning / tests); negative MR_ERR_* if invalid */
int64_t mr_conv2d_lds_bytes(const mr_conv_desc* desc);

/* launch (replaces the reference lines listed above mr_conv_desc) */
int mr_conv2d_f32(const mr_conv_desc* desc, void* stream);

/*
 * 3x3, stride 1, zero padding 1 convolution (PadSameConv2d(3) + nn.Conv2d(3): every ConvReLU of the MaskModule,
 * model/monorec/monorec_model.py:296-343, model/layers.py:317-335; the 3x3 stride-1 convolutions of the ResNet trunk) as Winograd
 * F(2x2, 3x3) on the fp32 matrix cores: 16 multiplies per (input channel, output channel) and 2x2 output tile instead of 36.
 * Same conventions as mr_conv2d_f32: up to MR_MAX_SOURCES sources concatenated on channels and read in place, dense NCHW fp32,
 * bias / residual / activation (MR_ACT_NONE, MR_ACT_RELU, MR_ACT_LEAKY_RELU) in the epilogue, dst = (batch, out_channels, height,
 * width).  width % 4 == 0.  Results differ from the direct convolution by the rounding of the transforms (~1e-6 relative).
 */
typedef struct mr_wino_desc {
    const float* src[MR_MAX_SOURCES];
    int32_t src_channels[MR_MAX_SOURCES];
    int32_t num_src;
    int32_t batch, height, width;
    float* dst;
    int32_t out_channels;
    const float* packed_weights;     /* from mr_wino_pack_weights_f32 with the same cout_blocks_per_wave (device copy) */
    const float* bias;               /* out_channels floats or NULL */
    const float* residual;           /* same shape as dst or NULL */
    int32_t activation;
    float act_p0;
    int32_t cout_blocks_per_wave;    /* 1 or 2: a workgroup (8 waves, 8 x 32 output pixels) produces 32 or 64 output channels */
    int32_t variant;                 /* mr_conv3x3_winograd_f32 only: 0 input transform through an LDS buffer (thread = channel x tile),
                                        1 input transform in the registers of the lane that feeds it to the matrix core (no V buffer,
                                        one barrier per chunk; bit-identical results).  Which is faster is measured per layer shape.
                                        2 = 1 for out_channels = 32 a + r, 0 < r <= 16, cout_blocks_per_wave 1: the r tail channels are
                                        produced by workgroups of 16 x 32 pixels x 16 channels instead of a half-empty 32-channel group
                                        (the 48-channel layers); packed_weights then from mr_wino_pack_weights_tail_f32. */
    /* ---- mr_conv1d_cooktoom_f32 only (every other entry point wants them 0): strided source views and a column-split destination, which is
     * how the stride-2 halves of layers.ConvReLU2 (model/layers.py:289-314 with stride (2,1) / (1,2); monorec_model.py:489-501) run as stride-1
     * forms over [even samples | odd samples] (monorec_amd/cooktoom.py: stride2_as_stride1). */
    int32_t src_row_pitch;           /* floats between consecutive rows of a source as the launch sees it; 0 = width (dense).  2 * width with
                                        height = H / 2 reads every second row of an (.., H, width) tensor: src[s] = its first element for the even
                                        rows, + width floats for the odd rows */
    int32_t src_plane_floats;        /* floats between consecutive channel planes of a source in memory; 0 = height * width (dense) */
    int32_t dst_split_columns;       /* axis 1 only: 1 = the destination is stored de-interleaved by column parity, dst = (2, batch, out_channels,
                                        height, width / 2): [0] the even columns, [1] the odd ones - two dense tensors the 1 x k stride-(1,2) half
                                        of the pair reads as its two sources */
} mr_wino_desc;
size_t mr_wino_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t cout_blocks_per_wave);
/* weight: (out_channels, sum(src_channels), 3, 3) fp32 host memory; the transformed filters G g G^T are formed in double */
int mr_wino_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                             int32_t cout_blocks_per_wave, float* dst);
/* variant 2: the full 32-channel groups as above (cout_blocks_per_wave 1), then the tail group of 1..16 channels */
size_t mr_wino_packed_weight_floats_tail(int32_t out_channels, const int32_t* src_channels, int32_t num_src);
int mr_wino_pack_weights_tail_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst);
int64_t mr_conv3x3_winograd_lds_bytes(const mr_wino_desc* desc);   /* dynamic LDS of the launch, or a negative MR_ERR_* code */
int mr_conv3x3_winograd_f32(const mr_wino_desc* desc, void* stream);

/*
 * The same 3x3 stride-1 convolution as Winograd F(4x4, 3x3): 36 multiplies per (input channel, output channel) and 4x4 outputs instead of
 * 144 (F(2x2, 3x3): 64) - the F(4, 3) Cook-Toom form of cooktoom_1d.h in both directions (points 0, +-1, +-2, infinity; transform
 * coefficients up to 8, transformed weights formed in double and rounded once).  Effect on the path's outputs measured before the kernel was
 * written (oracle/numerics_study_winograd.py: depth moves by 2.4e-7 with every 3x3 stride-1 layer of the mask and depth nets evaluated this
 * way).  Same descriptor and conventions as mr_conv3x3_winograd_f32 (`cout_blocks_per_wave` and `variant` ignored); a workgroup (8 waves)
 * produces 16 x 64 output pixels x 32 output channels and uses 153 KB of LDS, so it pays where the layer has >= 256 such workgroups.
 *
 * In the product library again since ABI 17 (ABI 16 had moved it to the diagnostic build: at c2 it buys nothing end to end); measured on
 * the c3 and configs[4] shapes it is 12-15 % ahead of the best F(2x2,3x3) variant on every full- and half-resolution 3x3 layer
 * (tools/sessions/r04_s18.sh), and only those table entries select it.
 */
size_t mr_wino44_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src);
int mr_wino44_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst);
int64_t mr_conv3x3_winograd44_lds_bytes(const mr_wino_desc* desc);
int mr_conv3x3_winograd44_f32(const mr_wino_desc* desc, void* stream);

/* The same F(4x4, 3x3) convolution with the 36 positions of a tile split over two waves (csrc/conv_wino44s.hip, round 5): a wave holds the 18
 * accumulator sets of one half of the vertical transform index and forms a partial output transform; the halves are added through LDS after the K
 * loop.  8 x 64 output pixels x 32 channels per workgroup, K in chunks of 4 channels, 70.5 KB of LDS and <= 128 registers: TWO workgroups per CU
 * (mr_conv3x3_winograd44_f32: one, whose load / transform / MFMA phases add up instead of overlapping).  Same products and transformed weights
 * as that kernel; the output differs by the rounding of a two-term partial sum.  Selected per layer shape by the measured table (code 41). */
size_t mr_wino44s_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src);
int mr_wino44s_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst);
int64_t mr_conv3x3_winograd44s_lds_bytes(const mr_wino_desc* desc);
int mr_conv3x3_winograd44s_f32(const mr_wino_desc* desc, void* stream);

/* The same F(4x4, 3x3) convolution with ONE WAVE PER SIMD (csrc/conv_wino44w.hip, round 6, ABI 19): 4 waves of 512 registers on 16 x 64 output pixels x 32
 * channels; a wave owns a tile row and BOTH 16-channel blocks (72 accumulator sets), so the input transform - VALU work that runs on the same ALUs as
 * the fp32 MFMAs - is done once per 72 MFMAs instead of once per 36, and the patch reads of the next channel quad are in flight while the current
 * quad multiplies; K in single-quad stages through a ring of four (partial vmcnt waits).  Packed weights of mr_wino44_pack_weights_f32; the same
 * products as mr_conv3x3_winograd44_f32, summed in the same order per output (bit-identical results).  Table code 51. */
int64_t mr_conv3x3_winograd44w_lds_bytes(const mr_wino_desc* desc);
int mr_conv3x3_winograd44w_f32(const mr_wino_desc* desc, void* stream);

/* Exported by the DIAGNOSTIC build only (python -m monorec_amd.build --timeline -> libmonorec_hip_timeline.so, selected with MR_HIP_LIBRARY):
 * bit 0 = the F(2,7) instantiations of mr_conv1d_cooktoom_f32 are present (no measured table entry ever selected them). */
#ifdef MR_DIAGNOSTIC_LIBRARY
int mr_diagnostic_forms(void);
#endif

/*
 * ---- bf16 MFMA path with channel-blocked bf16 activation storage (BASELINE configs[4]; MonoRecModel(hip_bf16=True)) --------------------
 * Layout "B8" of an activation: (batch, ceil(C / 8), H, W, 8) bf16 - one 16-byte group holds 8 consecutive channels of one pixel, padded
 * channels are zero.  mr_conv2d_b8 stands in for the nn.Conv2d / ConvTranspose2d(4,2) / Upconv layers of MaskModule and DepthModule
 * (model/monorec/monorec_model.py:345-385,526-557; model/layers.py:289-356,380-400): sources concatenated on channels and read in place,
 * each either B8 or dense fp32 NCHW (rounded to bf16, nearest even, while it is staged); v_mfma_f32_16x16x32_bf16 with fp32 accumulation;
 * bias and activation (none / ReLU / LeakyReLU(act_p0)) in the epilogue; destination B8 (rounded once) or fp32 NCHW, written at
 * (oy * out_step_h + phase_out_off_h, ox * out_step_w + phase_out_off_w) of a (dst_plane_h, dst_plane_w) plane.  Taps read input
 * (oy * stride_h - phase_pad_top + ky, ox * stride_w - phase_pad_left + kx), zero outside.  num_phases = 4 runs the four output parities of
 * a transposed / upsampling layer in one launch (per-phase filter size <= kh x kw, padding, offset, weights); num_phases <= 1 uses entry 0.
 * Not within the 1e-4 parity bar (bf16 operands): the accuracy bar of this mode is tests/test_gpu_model.py::test_c5_shape_in_fp32_and_bf16.
 */
#define MR_LAYOUT_F32_NCHW 0
#define MR_LAYOUT_BF16_B8  1
typedef struct mr_b8_conv_desc {
    const void* src[MR_MAX_SOURCES];
    int32_t src_channels[MR_MAX_SOURCES];
    int32_t src_layout[MR_MAX_SOURCES];      /* MR_LAYOUT_* */
    int32_t num_src, batch, src_h, src_w;
    int32_t kh, kw, stride_h, stride_w;      /* kh / kw: the maximum over the phases */
    int32_t out_h, out_w;                    /* conv grid size (per phase) */
    void* dst;
    int32_t dst_layout, out_channels;
    int32_t dst_plane_h, dst_plane_w, out_step_h, out_step_w;
    const float* bias;                       /* out_channels floats or NULL */
    int32_t activation;
    float act_p0;
    int32_t cout_blocks_per_wg;              /* MB: 1..4 blocks of 16 output channels per workgroup */
    int32_t pixel_blocks_per_wave;           /* NB: 1, 2, 4 */
    int32_t waves_per_wg;                    /* 4 or 8 */
    int32_t num_phases;
    const void* phase_weights[4];            /* from mr_b8_pack_weights with that phase's filter size (device copies) */
    int32_t phase_kh[4], phase_kw[4];        /* 0 = kh / kw */
    int32_t phase_pad_top[4], phase_pad_left[4], phase_out_off_h[4], phase_out_off_w[4];
} mr_b8_conv_desc;
size_t mr_b8_packed_weight_bytes(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t kh, int32_t kw, int32_t mb);
/* weight: (out_channels, sum(src_channels), kh, kw) fp32 in host memory -> bf16 A-fragment stream in host memory `dst` */
int mr_b8_pack_weights(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t kh, int32_t kw,
                       int32_t mb, void* dst);
int64_t mr_conv2d_b8_lds_bytes(const mr_b8_conv_desc* desc);
int mr_conv2d_b8(const mr_b8_conv_desc* desc, void* stream);
/* nn.MaxPool2d(2) of the next MaskModule encoder stage and the maximum over the frames (monorec_model.py:357-365) on B8 tensors:
 * src (frames, planes, h, w) 16-byte groups, planes = batch * channel blocks -> pooled (frames, planes, h/2, w/2), fmax (planes, h, w) */
int mr_pool2x2_framemax_b8(const void* src, void* pooled, void* fmax, int32_t frames, int64_t planes, int32_t h, int32_t w, void* stream);
int mr_max_over_frames_b8(const void* src, void* dst, int32_t frames, int64_t groups16, void* stream);
/* The two producers of the nets' fp32 inputs, writing a B8 copy on the side so that the first layers read no fp32 volume:
 * mr_cost_volume_b8_f32 = mr_cost_volume_mode_f32 of the default configuration (3x3 patch, sfcv * mask; 32 / 48 / 64 depth steps) with
 * sfcv_b8[f] = (batch, num_depths / 8, height, width, 8) bf16 per frame next to the dense single-frame volumes (which stay the outputs).
 * Being the entry point of the bf16 configuration only (accuracy bar 1e-2 / 1e-3 on the depth), it forms the 3x3 window sums separably and
 * multiplies by fp32(1/9) where mr_cost_volume_f32 keeps the reference's summation order and division: -17 % instructions; the volumes differ
 * from mr_cost_volume_f32's by <= 1e-4 (the fused volume by more where the frame weights nearly cancel, as between any two summation orders),
 * validity (the zeros of the volumes) exactly equal;
 * mr_mask_classifier_b8_f32 = mr_mask_classifier_f32 with the masked volume (monorec_model.py:713) also as
 * cost_volume_b8 = (batch, num_depths / 8, plane, 8) bf16 (num_depths % 16 == 0). */
int mr_cost_volume_b8_f32(const float* keyframe, const float* const* frames, int32_t num_frames, const float* kinv, const float* proj,
                          const float* depths, int32_t batch, int32_t num_depths, int32_t height, int32_t width, float alpha,
                          const float* channel_weights, int32_t use_ssim, const float* pixel_depths, float* cost_volume,
                          float* const* sfcv, void* const* sfcv_b8, void* stream);
/* mr_cost_volume_b8_f32 for a caller that does not hand out fp32 single-frame volumes (MonoRecModel(hip_bf16=True, hip_lean_outputs=True)):
 * the fusion kernel writes the fused volume and the B8 copies only; sfcv[f] are SCRATCH (they end up holding the raw per-frame costs, not
 * monorec_model.py:251's volumes) - D * F * H * W * 4 bytes of HBM writes less per keyframe (403 MB at BASELINE configs[4]). */
int mr_cost_volume_b8_lean_f32(const float* keyframe, const float* const* frames, int32_t num_frames, const float* kinv, const float* proj,
                               const float* depths, int32_t batch, int32_t num_depths, int32_t height, int32_t width, float alpha,
                               const float* channel_weights, int32_t use_ssim, const float* pixel_depths, float* cost_volume,
                               float* const* sfcv_scratch, void* const* sfcv_b8, void* stream);
int mr_mask_classifier_b8_f32(const float* features, const float* weight, const float* bias, int32_t batch, int32_t channels, int64_t plane,
                              float* cv_mask, float* cost_volume, int32_t num_depths, void* cost_volume_b8, void* stream);
/* layout conversions: dense fp32 (n, c, hw) <-> B8 (n, ceil(c/8), hw, 8) bf16 */
int mr_f32_nchw_to_b8(const float* src, void* dst, int32_t n, int32_t c, int64_t hw, void* stream);
int mr_b8_to_f32_nchw(const void* src, float* dst, int32_t n, int32_t c, int64_t hw, void* stream);

/*
 * 3x1 / 1x3, stride 1, zero padding 1 along the filter axis (PadSameConv2d + nn.Conv2d of layers.ConvReLU2, model/layers.py:289-314: the
 * second pair of every DepthModule encoder stage, dec{1,2}.1, dec4.0 - model/monorec/monorec_model.py:485-513) as 1-D Winograd F(2, 3):
 * 4 multiplies per (input channel, output channel) and 2 outputs instead of 6.  Same descriptor and conventions as
 * mr_conv3x3_winograd_f32 (sources concatenated on channels and read in place, bias / activation in the epilogue, dst = (batch,
 * out_channels, height, width), width % 4 == 0); no residual; `variant` ignored; cout_blocks_per_wave 1..4: a workgroup (8 waves, 8 x 32
 * output pixels) produces 16 x that many output channels.  axis 0: 1 x 3 (along x), weight (out, in, 1, 3); axis 1: 3 x 1 (along y),
 * weight (out, in, 3, 1) - three taps per (out, in) in memory either way.
 */
size_t mr_wino1d_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t cout_blocks_per_wave);
int mr_wino1d_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                               int32_t cout_blocks_per_wave, float* dst);
int64_t mr_conv1d3_winograd_lds_bytes(const mr_wino_desc* desc);
int mr_conv1d3_winograd_f32(const mr_wino_desc* desc, int32_t axis, void* stream);

/*
 * The larger Cook-Toom forms of the same layers: F(m, r) = m outputs along the filter axis from m + r - 1 inputs with m + r - 1
 * multiplies per (input channel, output channel).  (m, r) = (4, 3): 6 instead of 12 multiplies per 4 outputs of the 3-tap layers;
 * (2, 7) / (4, 7): 8 / 10 instead of 14 / 28 for the 7 x 1 and 1 x 7 stride-1 layers of DepthModule.enc.0.0
 * (model/monorec/monorec_model.py:487-500, layers.ConvReLU2 model/layers.py:289-314).  Interpolation points 0, +-1, +-2, +-1/2, +-4,
 * infinity; transform coefficients are dyadic rationals, the transformed weights are formed in double and rounded once.  Effect on the
 * path's outputs measured before the kernel was written (oracle/numerics_study_winograd.py: depth moves by <= 4e-7).  Descriptor and
 * conventions as mr_conv1d3_winograd_f32 (zero padding (r - 1) / 2 either side, width % 4 == 0, no residual); weight (out, in, 1, r) for
 * axis 0, (out, in, r, 1) for axis 1; cout_blocks_per_wave 1..4 (1..3 for (4, 7)).  A workgroup (8 waves) produces 8 rows x 16 m
 * columns (axis 0) or 4 m rows x 32 columns (axis 1).  MR_ERR_BAD_ARGUMENT for any other (m, r).
 */
size_t mr_cooktoom1d_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t cout_blocks_per_wave,
                                          int32_t m, int32_t r);
int mr_cooktoom1d_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                   int32_t cout_blocks_per_wave, int32_t m, int32_t r, float* dst);
int64_t mr_conv1d_cooktoom_lds_bytes(const mr_wino_desc* desc, int32_t axis, int32_t m, int32_t r);
int mr_conv1d_cooktoom_f32(const mr_wino_desc* desc, int32_t axis, int32_t m, int32_t r, void* stream);

/*
 * layers.Upconv (model/layers.py:349-356: nn.Upsample(2, nearest) -> pad (0,1,0,1) -> nn.Conv2d(2); the first layer of every MaskModule
 * decoder stage, model/monorec/monorec_model.py:318-338) with 4 multiplies per (input channel, output channel) and 2x2 output block
 * instead of 16 (9 as four parity phases of mr_conv2d_f32): the block of input position (y, x) is a bilinear form of the 2x2 input patch
 * at (y, x), evaluated on differences of neighbouring inputs.  Same descriptor as the functions above: sources (batch, C_s, height, width)
 * are the LOW-resolution input, dst = (batch, out_channels, 2 height, 2 width), bias / activation in the epilogue, no residual,
 * cout_blocks_per_wave 1 or 2 (16 or 32 output channels per workgroup), width % 4 == 0.
 */
int mr_upconv_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                               int32_t cout_blocks_per_wave, float* dst);      /* size: mr_wino1d_packed_weight_floats */
int mr_upconv2x2_winograd_f32(const mr_wino_desc* desc, void* stream);

/*
 * nn.ConvTranspose2d(kernel 4, stride 2) + the centre crop of layers.Refine (model/layers.py:380-400: the decoder stages of the
 * DepthModule, model/monorec/monorec_model.py:503-513) as Winograd F(2x2, 2x2): each of the four output parities is a 2x2 stride-1
 * convolution on the low-resolution input (9 multiplies per 2x2 parity tile instead of 16; transform coefficients 0 / +-1 only).
 * Same descriptor as mr_conv3x3_winograd_f32: sources (batch, C_s, height, width) read in place, dst = (batch, out_channels,
 * 2 * height, 2 * width), bias / activation in the epilogue, no residual; cout_blocks_per_wave 1, 2 or 4 (32 / 64 / 128 output channels
 * per workgroup); packed_weights from mr_wino_t_pack_weights_f32 with the same value.  width % 4 == 0.  `variant` as for
 * mr_conv3x3_winograd_f32 (0 transform through LDS, 1 in registers, 2 = 1 with 16-channel tail workgroups).
 */
size_t mr_wino_t_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t cout_blocks_per_wave);
/* weight: the nn.ConvTranspose2d tensor (sum(src_channels), out_channels, 4, 4), fp32 host memory */
int mr_wino_t_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                               int32_t cout_blocks_per_wave, float* dst);
/* variant 2 (out_channels = 32 a + r, 0 < r <= 16, cout_blocks_per_wave 1): full 32-channel groups, then the tail group of r channels */
size_t mr_wino_t_packed_weight_floats_tail(int32_t out_channels, const int32_t* src_channels, int32_t num_src);
int mr_wino_t_pack_weights_tail_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst);
int64_t mr_convt4x4s2_winograd_lds_bytes(const mr_wino_desc* desc);
int mr_convt4x4s2_winograd_f32(const mr_wino_desc* desc, void* stream);

/*
 * Fused plane-sweep cost volume.  Replaces CostVolumeModule.forward per-pixel work,
 * model/monorec/monorec_model.py:193-271 (+ model/layers.py:63-71 point_projection,
 * :119-137 SSIM, F.grid_sample x2, F.conv3d), for use_mono, use_ssim=True, sfcv_mult_mask=True,
 * patch_size=3.  The 4x4 pose/intrinsics algebra of :171,198,207 stays on the host:
 *   kinv   : batch x 9      inverse(keyframe_intrinsics)[:3,:3], row-major
 *   proj   : batch x F x 12 (K_f @ (inverse(pose_f) @ pose_kf))[:3,:4], row-major
 *   depths : D depth hypotheses (1/linspace(inv_max, inv_min, D)), far to near
 * keyframe: (batch,3,H,W); frames[f]: (batch,3,H,W), all in [-0.5,0.5].
 * Outputs: cost_volume (batch,D,H,W); sfcv[f] (batch,D,H,W).
 * D >= 2 (any parity: the kernels take the planes in pairs, an odd D's last pair repeats the last hypothesis and drops its second plane), F <= MR_MAX_FRAMES.  sfcv[f] doubles as scratch for the raw matching cost between the two
 * internal launches (sad kernel, per-pixel fusion kernel); no other workspace is needed.
 */
int mr_cost_volume_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                       const float* kinv, const float* proj, const float* depths,
                       int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                       float alpha, const float* channel_weights /* 3 host floats */,
                       float* cost_volume, float* const* sfcv, void* stream);

/* The same with the photometric term selected like CostVolumeModule's use_ssim (monorec_model.py:227-243):
 * 1 = SSIM distance (what mr_cost_volume_f32 does), 0 = absolute difference, 2 = 0.85 SSIM + 0.15 absolute difference,
 * 3 = absolute difference averaged over 3x3 (zero padded); and, when pixel_depths != NULL, per-pixel depth hypotheses
 * (batch, num_depths, H, W) instead of the num_depths shared ones - data_dict["cv_depths"], monorec_model.py:181-182
 * (`depths` may then be NULL); and sfcv_mult_mask = 0 masks the single-frame volumes per depth plane by
 * (any channel of the warped pixel != 0) | (warped pixel == keyframe pixel) instead of by the all-depth validity
 * (monorec_model.py:252-253; needs num_depths >= num_frames), 1 = default. */
int mr_cost_volume_mode_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                            const float* kinv, const float* proj, const float* depths,
                            int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                            float alpha, const float* channel_weights, int32_t use_ssim,
                            const float* pixel_depths, int32_t sfcv_mult_mask,
                            float* cost_volume, float* const* sfcv, void* stream);

/* Opt-in of the fp32 path (MonoRecModel(hip_cv_separable=True)): mr_cost_volume_mode_f32 of the default configuration (3x3 patch,
 * sfcv * mask) with the 3x3 window sums of layers.SSIM (model/layers.py:123-131) and of the box stage (monorec_model.py:246-248) formed
 * SEPARABLY (every row keeps its horizontal sums; a window is (top + mid) + cur) and x * fp32(1/9) for x / 9 - another rounding of the same
 * nine-term sums, -17 % instructions in the sad kernel.  Validity (the zeros of the volumes) is exactly mr_cost_volume_f32's; single-frame
 * volumes differ by <= 1e-4, the depth by <= 2e-6 (inside north_star's 1e-4-on-depth bar; the DEFAULT stays the exact-order kernel, whose
 * volumes agree with the reference to 5e-7).  Per-pixel depths / sizes without an exact constant division run the exact kernels. */
int mr_cost_volume_relaxed_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                               const float* kinv, const float* proj, const float* depths,
                               int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                               float alpha, const float* channel_weights, int32_t use_ssim, const float* pixel_depths,
                               float* cost_volume, float* const* sfcv, void* stream);

/* mr_cost_volume_mode_f32 on the round-1 kernels (LDS-tiled sad kernel + three-pass fusion kernel) whatever the options: the
 * default configuration (use_ssim 1, sfcv_mult_mask 1) otherwise runs the LDS-free marching kernel + register-resident fusion
 * kernel, whose results are bit-identical.  Kept as the A/B reference of that claim (tests/test_gpu_kernels.py) and for timing. */
int mr_cost_volume_tiled_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                             const float* kinv, const float* proj, const float* depths,
                             int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                             float alpha, const float* channel_weights, int32_t use_ssim,
                             const float* pixel_depths, int32_t sfcv_mult_mask,
                             float* cost_volume, float* const* sfcv, void* stream);

/* The same with a P x P matching patch, `cv_patch_size` of MonoRecModel (monorec_model.py:138-142,247): the photometric term is
 * averaged over a zero-padded patch_size x patch_size box and the border radius becomes patch_size / 2 + 1 (:139).
 * patch_size odd, 1..7; 3 is the tuned kernel of mr_cost_volume_mode_f32, the others run a generic (slower) variant. */
int mr_cost_volume_patch_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                             const float* kinv, const float* proj, const float* depths,
                             int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                             float alpha, const float* channel_weights, int32_t use_ssim,
                             const float* pixel_depths, int32_t sfcv_mult_mask, int32_t patch_size,
                             float* cost_volume, float* const* sfcv, void* stream);

/* nn.MaxPool2d(kernel 3, stride 2, padding 1) of the torchvision ResNet stem (monorec_model.py:124) */
int mr_maxpool3x3s2_f32(const float* src, float* dst, int32_t planes, int32_t in_h, int32_t in_w, void* stream);

/* nn.MaxPool2d(2) between the MaskModule encoder stages (monorec_model.py:304-316). in_h even, in_w % 4 == 0.
 * (mr_conv2d_f32 can also pool while staging - MR_IN_MAXPOOL2 - but the separate pass + DMA-staged conv is
 * faster on MI355X.) */
int mr_maxpool2x2_f32(const float* src, float* dst, int64_t planes, int32_t in_h, int32_t in_w, void* stream);

/* MonoRecModel.forward returns tensors the caller owns (the reference's outputs are fresh tensors: monorec_model.py:256-279,
 * 690, 713-727; create_pointcloud.py:79-98 keeps `result` of five keyframes and multiplies one in place).  The path computes into
 * resident buffers, so forward() ends with ONE launch that copies every output into the caller's memory: `num_segments`
 * (<= MR_MAX_COPY_SEGMENTS) independent device-to-device copies; src / dst 16-byte aligned, bytes a positive multiple of 16. */
#define MR_MAX_COPY_SEGMENTS 24
typedef struct mr_copy_segment {
    const void* src;
    void* dst;
    int64_t bytes;
} mr_copy_segment;
int mr_copy_segments(const mr_copy_segment* segments, int32_t num_segments, void* stream);

/* Host helper of the cost-volume launch: 1 when the 3-instruction reciprocal sequence the kernels use for `u / (W - 1)`, `v / (H - 1)`
 * (model/layers.py:67-68) reproduces the correctly rounded fp32 quotient for every dividend (exhaustive over a binade, cached per
 * divisor), 0 when the launch must use IEEE division for this divisor.  Exposed so that tests can pin the verdicts. */
int mr_exact_const_division(float divisor);

/* `num` (<= MR_MAX_GATHER) small fp32 tensors of `floats_each` elements each, anywhere in device memory, -> dst[num][floats_each],
 * one launch.  MonoRecModel uses it to bring the 4x4 pose / intrinsics matrices of a forward (monorec_model.py:160-171: they feed
 * torch.inverse / matmul, which this implementation runs with the reference's CPU operators) into device-writable pinned host memory. */
#define MR_MAX_GATHER (2 + 2 * MR_MAX_FRAMES)
int mr_gather_small_f32(const float* const* srcs, int32_t num, int32_t floats_each, float* dst, void* stream);

/* A run of consecutive convolution launches behind ONE host call: item i names an entry point above (MR_LAUNCH_*) and its descriptor
 * (mr_conv_desc for MR_LAUNCH_CONV2D, mr_wino_desc otherwise; `arg` = the axis of mr_conv1d3_winograd_f32, or
 * axis | m << 4 | r << 8 for MR_LAUNCH_COOKTOOM_1D = mr_conv1d_cooktoom_f32).  Launched in order on
 * `stream`; stops at the first failure, returns its code and - if `failed_index` is given - its position.  (No reference equivalent:
 * the reference issues one ATen call per layer from Python; this is the launch-overhead side of MonoRecModel.forward.) */
#define MR_LAUNCH_CONV2D  0
#define MR_LAUNCH_WINO3X3 1
#define MR_LAUNCH_WINO_T  2
#define MR_LAUNCH_WINO_1D 3
#define MR_LAUNCH_UPCONV  4
#define MR_LAUNCH_COOKTOOM_1D 5
#define MR_LAUNCH_WINO44  6
#define MR_LAUNCH_CONV_B8 7
#define MR_LAUNCH_WINO44S 8
#define MR_LAUNCH_WINO44W 9
typedef struct mr_launch_item {
    int32_t kind;
    int32_t arg;
    const void* desc;
} mr_launch_item;
int mr_run_launches(const mr_launch_item* items, int32_t num, void* stream, int32_t* failed_index);

/* ResnetEncoder input normalisation ((x + 0.5) - 0.45) / 0.225, elementwise (monorec_model.py:691 + :120);
 * count % 4 == 0.  (MR_TF_RESNET_NORM does the same while staging inside mr_conv2d_f32.) */
int mr_resnet_normalize_f32(const float* src, float* dst, int64_t count, void* stream);

/* elementwise max over F stacked tensors: torch.max(cv_feats[i], x) (monorec_model.py:365).
 * src is (F, count) contiguous. */
int mr_max_over_frames_f32(const float* src, float* dst, int32_t num_frames, int64_t count, void* stream);

/* SimpleMaskModule input (monorec_model.py:448-449): stacked.sum(0) / (stacked != 0).sum(0).clamp_min(1) over the F
 * single-frame cost volumes.  src is (F, count) contiguous, count % 4 == 0. */
int mr_nonzero_mean_over_frames_f32(const float* src, float* dst, int32_t num_frames, int64_t count, void* stream);

/* Both consumers of a MaskModule encoder stage output in one pass: pooled = MaxPool2d(2) per frame (next stage input,
 * monorec_model.py:304-316) and frame_max = max over the frames (cv_feats[i], :365).
 * src (F, planes, in_h, in_w) -> pooled (F, planes, in_h/2, in_w/2), frame_max (planes, in_h, in_w);
 * planes = batch * channels, in_h even, in_w % 4 == 0. */
int mr_pool2x2_framemax_f32(const float* src, float* pooled, float* frame_max, int32_t num_frames, int64_t planes,
                            int32_t in_h, int32_t in_w, void* stream);

/* cost_volume = (1 - cv_mask) * cost_volume (monorec_model.py:713); mask (batch,1,H,W), cv (batch,D,H,W).
 * dst may alias cv. */
int mr_apply_mask_f32(const float* cv, const float* mask, float* dst, int32_t batch, int32_t num_depths,
                      int64_t plane, void* stream);

/* MaskModule.classifier = nn.Conv2d(C, 1, kernel_size=1) + nn.Sigmoid (monorec_model.py:340-343, applied :383) fused with the
 * mask multiply that follows it in MonoRecModel.forward, cost_volume = (1 - cv_mask) * cost_volume (:713):
 *   cv_mask[b,0,p] = sigmoid(bias[0] + sum_c weight[c] * features[b,c,p]);   cost_volume[b,d,p] *= 1 - cv_mask[b,0,p]  (in place)
 * features (batch, channels, plane) with plane = H*W even, weight (channels) = the conv's (1,C,1,1) tensor, cv_mask (batch,1,plane),
 * cost_volume (batch, num_depths, plane) or NULL (mask only: pretrain_mode 2, :712,723).  One HBM-bound launch instead of a
 * 1-of-16-rows MFMA launch + mr_apply_mask_f32. */
int mr_mask_classifier_f32(const float* features, const float* weight, const float* bias, int32_t batch, int32_t channels,
                           int64_t plane, float* cv_mask, float* cost_volume, int32_t num_depths, void* stream);

/* DepthModule.predictors[i] = PadSameConv2d(3) + nn.Conv2d(C_i, 1, 3) (monorec_model.py:520-523), predict_depth's abs(tanh(x))
 * (:554-557) and the inverse-depth affine (1 - p) * act_p0 + p * act_p1 of MonoRecModel.forward (:716-717), for up to
 * MR_MAX_HEADS heads in ONE launch (their inputs differ in size; nothing downstream reads the outputs, so they run together once
 * the decoder is done).  Per head: src (batch, channels, height, width), weight (1, channels, 3, 3) as nn.Conv2d stores it,
 * bias (1), dst (batch, 1, height, width); zero padding 1 on every side (PadSameConv2d of k=3, s=1: model/layers.py:249-251). */
typedef struct mr_head_desc {
    const float* src;
    const float* weight;
    const float* bias;
    float* dst;
    int32_t batch, channels, height, width;
} mr_head_desc;
int mr_depth_heads_f32(const mr_head_desc* heads, int32_t num_heads, float act_p0, float act_p1, void* stream);

/* Fused sparse depth metrics, one launch for all seven metrics of configs/evaluate/eval_monorec.json:53-61
 * (model/metric_functions/sparse_metrics.py:136-252 with utils/util.py:36-118; pred_all_valid=True, no cv-mask):
 * prediction/target (batch,1,H,W) inverse depths; roi = {y0,y1,x0,x1} host ints or NULL; max_distance <= 0 = None.
 * sums: batch x 8 doubles on the device, per sample [n_valid, sum|dp-dg|/dg, sum(dp-dg)^2/dg, sum(dp-dg)^2,
 * sum(log dp - log dg)^2, n(thresh<1.25), n(thresh<1.25^2), n(thresh<1.25^3)]. */
int mr_sparse_metric_sums_f32(const float* prediction, const float* target, int32_t batch, int32_t height,
                              int32_t width, const int32_t* roi, float max_distance, double* sums, void* stream);

/* ---- point-cloud path: create_pointcloud.py + utils/ply_utils.py -------------------------------
 *
 * Static-scene mask of one keyframe, create_pointcloud.py:76-77:
 *   mask = (cv_mask >= threshold); out = (F.conv2d(mask, ones(mask_fill+1, mask_fill+1), padding=mask_fill//2) < 1)
 * i.e. 1 where no moving pixel lies within +-mask_fill/2 (zero padding), else 0.  cv_mask, out: (batch,1,H,W);
 * mask_fill even (reference: 32, threshold .1). */
int mr_static_mask_f32(const float* cv_mask, float* out, int32_t batch, int32_t height, int32_t width,
                       float threshold, int32_t mask_fill, void* stream);

/* One PLYSaver.add_depthmap (utils/ply_utils.py:34-53) preceded by the mask vote of create_pointcloud.py:90-92:
 *   vote   = (sum_k static_masks[k]) > vote_above            (num_masks = 0: no masking; reference: 5 masks, > 4)
 *   depth  = 1 / (inv_depth * vote);  keep = min_d <= depth <= max_d, inside roi (y0,y1,x0,x1; python slice
 *            semantics; NULL = whole image), and uniform > dropout when `uniform` (the torch.rand_like of :45) is given
 *   point  = pose @ [depth * (kinv @ (x, y, 1)); 1]           (model/layers.py:56-61, ply_utils.py:47-49)
 *   colour = (image + .5) * 255
 * Kept points are appended in the reference's order (batch, then pixel row-major) as 6 floats x y z r g b to
 * `records` starting at record *cursor; *cursor (device memory) is advanced by the number of kept points even
 * beyond `capacity_records` (nothing is written past the capacity - the caller checks).  No host synchronisation.
 * inv_depth/uniform (batch,1,H,W), image (batch,3,H,W), kinv batch x 9 (inverse(intrinsics)[:3,:3]), pose batch x 16. */
int mr_pointcloud_append_f32(const float* inv_depth, const float* const* static_masks, int32_t num_masks,
                             float vote_above, const float* image, const float* kinv, const float* pose,
                             const float* uniform, float dropout, float min_d, float max_d, const int32_t* roi,
                             int32_t batch, int32_t height, int32_t width, float* records, int64_t capacity_records,
                             int64_t* cursor, void* stream);

/* ---- input pipeline: KittiOdometryDataset.preprocess_image, kitti_odometry_dataset.py:120-134 ------
 *
 * img.crop(box) -> img.resize((W,H), Image.BILINEAR) -> float32 / 255 - .5 -> CHW (a 1-channel image is stacked 3x),
 * Pillow's resampling (Resample.c: scaled triangle filter, 22-bit fixed-point weights, horizontal pass into an 8-bit
 * intermediate, then vertical) reproduced bit for bit.
 *
 * Host side, once per (size, box): the coefficient tables of one axis for a crop [in0, in1) of an axis of in_size
 * pixels resized to out_size.  ksize = mr_resample_ksize_bilinear(in0, in1, out_size); bounds: out_size x 2 ints
 * (first source index, tap count), coeffs: out_size x ksize ints (unused taps 0).  For a cropped image pass
 * in_size = in1 - in0, in0 = 0 (Image.crop materialises the crop before the resize). */
int32_t mr_resample_ksize_bilinear(int32_t in0, int32_t in1, int32_t out_size);
int mr_resample_coeffs_bilinear(int32_t in_size, int32_t in0, int32_t in1, int32_t out_size, int32_t* bounds,
                                int32_t* coeffs);

/* Device side: src = decoded image in device memory, uint8, interleaved (src_h, src_w, channels), channels 1 or 3;
 * box = integer crop (x0, y0, x1, y1) as Image.crop rounds it; tables (device memory) for the cropped size;
 * max_tile_rows = max over output row tiles [16t, 16t+16) of the source rows they touch
 * (vbounds[last].first + vbounds[last].count - vbounds[first].first); dst (3, out_h, out_w) fp32. */
int mr_preprocess_image_u8_f32(const uint8_t* src, int32_t src_h, int32_t src_w, int32_t channels,
                               int64_t row_stride_bytes, const int32_t* box, int32_t out_h, int32_t out_w,
                               const int32_t* hbounds, const int32_t* hcoeffs, int32_t hksize,
                               const int32_t* vbounds, const int32_t* vcoeffs, int32_t vksize,
                               int32_t max_tile_rows, float* dst, void* stream);

/* Sparse lidar ground truth, preprocess_depth_annotated_lidar (kitti_odometry_dataset.py:184-211): the 16-bit depth PNG
 * (depth * 256, 0 = no return) -> inverse depth 256 / value scattered to the nearest cell of the (out_h, out_w) grid of
 * the cropped (box = x0, y0, x1, y1; NULL = none) and rescaled image; colliding samples: last one in row-major source
 * order wins, like numpy's fancy assignment.  owner_scratch: out_h * out_w ints of device scratch.  dst (out_h, out_w). */
int mr_lidar_inverse_depth_u16_f32(const uint16_t* depth_png, int32_t src_h, int32_t src_w, const int32_t* box,
                                   int32_t out_h, int32_t out_w, int32_t* owner_scratch, float* dst, void* stream);

/* Sparse D(V)SO ground truth, preprocess_depth_dso (kitti_odometry_dataset.py:156-182): the 16-bit PNG (0 = no point) ->
 * inverse depth orig_w * value / (0.54 * focal_x * 65535) scattered like mr_lidar_inverse_depth_u16_f32; source coordinates
 * are rescaled to the original image size (orig_h, orig_w) first and the crop box is tested on those (NULL = none). */
int mr_dso_inverse_depth_u16_f32(const uint16_t* depth_png, int32_t src_h, int32_t src_w, int32_t orig_h, int32_t orig_w,
                                 double focal_x, const int32_t* box, int32_t out_h, int32_t out_w,
                                 int32_t* owner_scratch, float* dst, void* stream);

int mr_abi_version(void);
const char* mr_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* MONOREC_HIP_H */
