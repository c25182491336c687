/*
 * tedspad_hip.h -- C ABI of libtedspad_hip.so, the MI355X (gfx950) kernels behind the
 * TeD-SPAD video-encoder hot path (I3D clip feature extraction, UNet anonymizer, losses).
 *
 * The reference has NO native layer: every op below is reached there through a stock
 * torch.nn module (cuDNN).  Each entry point cites the reference call it replaces.
 * Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     the name says host; `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, a negative TEDSPAD_E* code otherwise and
 *     never throws; tedspad_last_error() gives the message (thread-local).
 *   - no hidden allocation, no synchronisation: launches are asynchronous on `stream`
 *     and are hipGraph-capturable.
 *   - activations are channels-last (N,T,H,W,C) 16-bit (f16 or bf16, see `dtype`),
 *     accumulation and the BN/residual/ReLU epilogue are fp32.
 */
#ifndef TEDSPAD_HIP_H
#define TEDSPAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEDSPAD_ABI_VERSION 5   /* 5: tedspad_count_saturated added (round 6); 2: the BatchNorm entries take `zdtype` / `groups` / `dbias`, multi-job refresh entries (round 2); 3: tedspad_conv_extras.nosat / .nchunk_src, the fp32-clip stem entry (round 4); 4: tedspad_bneck_l1_* removed, tedspad_frames_crop_resize_tp added (round 5) */

enum { TEDSPAD_F16 = 0, TEDSPAD_BF16 = 1, TEDSPAD_F32 = 2 /* only where an argument says so (the BatchNorm `zdtype`) */ };
enum { TEDSPAD_OK = 0, TEDSPAD_EINVAL = -1, TEDSPAD_ELAUNCH = -2, TEDSPAD_EUNSUPPORTED = -3 };

/* Geometry of one convolution in channels-last form.  2-D convs (UNet) use t = kt = 1. */
typedef struct tedspad_conv_desc {
    int32_t n, t, h, w;          /* input pixels                                              */
    int32_t cin;                 /* channels read per pixel, multiple of 8                    */
    int32_t ldx;                 /* elements between consecutive input pixels (>= cin)        */
    int32_t cout;                /* output channels written, multiple of 8                    */
    int32_t ldy;                 /* elements between consecutive output pixels (concat slices) */
    int32_t ldres;               /* same for the residual tensor; ignored if residual == NULL */
    int32_t kt, kh, kw;          /* kernel                                                    */
    int32_t st, sh, sw;          /* stride                                                    */
    int32_t pt, ph, pw;          /* FRONT zero padding (TF-SAME is asymmetric: i3d.py:82-106) */
    int32_t to, ho, wo;          /* output pixels                                             */
    int32_t relu;                /* 1: ReLU after scale/shift(+residual)                      */
    int32_t dtype;               /* TEDSPAD_F16 | TEDSPAD_BF16                                */
    int32_t tile_cfg;            /* 0: built-in heuristic; 1..tedspad_conv_num_tile_cfgs(): forced
                                    (the host autotuner's analogue of cudnn.benchmark, train_anonymizer.py:28) */
} tedspad_conv_desc;

typedef struct tedspad_pool_desc {
    int32_t n, t, h, w, c;       /* input, c multiple of 8                                    */
    int32_t ldx, ldy;            /* pixel strides (elements)                                  */
    int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
    int32_t to, ho, wo;
    int32_t pad_zero;            /* 1: padded taps contribute 0 (MaxPool3dSamePadding, i3d.py:41-45) */
    int32_t dtype;
} tedspad_pool_desc;

int32_t     tedspad_abi_version(void);
const char *tedspad_last_error(void);

/* ---- packing helpers (HOST side, pure CPU) ------------------------------------------------ */

/* K = kt*kh*kw*cin rounded up to the kernel's K tile; rows of the packed weight matrix. */
int32_t tedspad_conv_kpad(const tedspad_conv_desc *d);
int32_t tedspad_conv_cout_pad(const tedspad_conv_desc *d);
/* Number of tile configurations of the conv kernel (valid tile_cfg values are 1..this). */
int32_t tedspad_conv_num_tile_cfgs(void);
/* Number of int32 pairs in the K-chunk table (= kpad / 8). */
int32_t tedspad_conv_ktab_entries(const tedspad_conv_desc *d);
/* Fills host_out[2*entries]: {element offset of the tap+channel chunk, packed (dt,dh,dw,valid)}. */
int32_t tedspad_conv_build_ktab(const tedspad_conv_desc *d, int32_t *host_out);

/* ---- device launchers --------------------------------------------------------------------- */

/*
 * y[n,to,ho,wo,co] = act( scale[co] * sum_{dt,dh,dw,ci} x[n, to*st-pt+dt, ..., ci] * w[co][(dt,dh,dw,ci)]
 *                         + shift[co] (+ residual[n,to,ho,wo,co]) )
 * Replaces nn.Conv3d/Conv2d + BatchNorm(eval) + ReLU (+ residual add):
 *   Unit3D.forward            aux_code/models/i3d.py:89-120
 *   Bottleneck.forward        aux_code/models/large_i3d.py:61-84
 *   I3Res50 stem              aux_code/models/large_i3d.py:229-231
 *   DoubleConv / OutConv      aux_code/models/unet_parts.py:8-25,71-77
 * w_packed: [cout_pad][kpad] 16-bit, K ordered (dt,dh,dw,ci), zero padded.
 * ktab: device copy of tedspad_conv_build_ktab.  scale/shift: fp32 [cout_pad].
 * act: relu if d->relu; `sigmoid` != 0 applies a logistic instead (UNet output, unet_model.py:37).
 */
int32_t tedspad_conv_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed,
                         const int32_t *ktab, const float *scale, const float *shift,
                         const void *residual, void *y, int32_t sigmoid, void *stream);

/* Optional epilogue extras used by the training path (all-zero / NULL = plain tedspad_conv_fwd):
 *   mask   : 16-bit tensor shaped like the output; out = mask > 0 ? out : 0. Fuses the ReLU backward
 *            into the data-gradient conv that produces d(input) (the input IS the ReLU output).
 *   stats  : fp32 [2][stats_ld], pre-zeroed; receives per-channel sum and sum of squares of
 *            scale*conv+shift over all output pixels (BatchNorm batch statistics, train mode:
 *            train_anonymizer.py:73,139) via one atomic per channel per workgroup.
 *   out_strided: output pixel (n,to,ho,wo) is written at (n, to*ost+oot, ho*osh+ooh, wo*osw+oow) of a
 *            (tf,hf,wf) tensor: lets the stride-s data gradient run as s^d dense sub-convolutions
 *            (one per output parity class) that interleave their results in place. */
typedef struct tedspad_conv_extras {
    const void *mask;
    float      *stats;
    float      *y32;     /* optional fp32 output (pixel stride ldy32); `y` may then be NULL. Train-mode BatchNorm keeps
                            the pre-normalisation conv output in fp32: rounding it to 16 bits before subtracting the
                            batch mean flips ReLU branches (DESIGN.md "training precision"). */
    int32_t     ldmask, stats_ld, ldy32;
    int32_t     out_strided, ost, osh, osw, oot, ooh, oow, tf, hf, wf;
    int32_t     fold_hw, fold_c, fold_ldy;  /* fold_hw > 0 (tile_cfg 25 / 26 only, plain epilogue): the conv's output channels hold cout / fold_c output FRAMES of
                                fold_c channels: channel n of output row m = sample*fold_hw + px (fold_hw = to*ho*wo) is stored as channel n % fold_c of row
                                (sample * (cout / fold_c) + n / fold_c) * fold_hw + px with row stride fold_ldy (d->ldy is not used); scale / shift are fold_c long; fold_c a multiple of 128 (an epilogue pass of the kernel). Lets a 3x1x1 'same' conv on a
                                2-frame tensor (layer3 / layer4 of I3Res50 after maxpool2) run as ONE GEMM with K = 2*cin over both frames -- out[0] =
                                [W1 W2].[x0;x1], out[1] = [W0 W1].[x0;x1] -- instead of K = 3*cin with a third of the products on zero padding */
    int32_t     stats_rows;  /* 0: one set of statistics over all output rows. > 0 (>= 256): output rows [g*stats_rows, (g+1)*stats_rows) are
                                statistics GROUP g and accumulate into stats + g*2*stats_ld -- the three clips of a training step run through the
                                network as ONE batch while their BatchNorms keep separate batch statistics (train_anonymizer.py:169-175) */
    int32_t     nosat;       /* 1: f16 results are NOT clamped to +-65504 (the training path: an overflow stays inf / NaN, as under the reference's fp16
                                autocast, train_anonymizer.py:78, so that a loss scale's non-finite check sees it). 0: saturate (inference) */
    int32_t     nchunk_src;  /* > 0 (= cin / 64, <= 8; tile_cfg 32 / 33 only, kt = 1): the input is a GATHERED CONCATENATION -- the 64-channel chunk k of every
                                input pixel is read from chunk_src[k] (pointer to that chunk's first channel at pixel 0, 16-byte aligned) with pixel stride
                                chunk_ld[k]; bit k of chunk_up: that source is a (n, h/2, w/2) tensor read through a nearest x2 upsample (pixel (h, w) <-
                                (h >> 1, w >> 1); h, w even). `x` / d->ldx are not read. This is `torch.cat([F.interpolate(x, scale_factor=2, mode='nearest'),
                                *skips], dim=1)` of the default anonymizer's decoder blocks (arch='unet++', aux_code/model_loaders.py:17-30; smp
                                decoders/unetplusplus/decoder.py DecoderBlock.forward) without the upsampled tensor or the concat buffer ever being written */
    int32_t     chunk_up;
    int32_t     chunk_ld[8];
    const void *chunk_src[8];
} tedspad_conv_extras;

int32_t tedspad_conv_fwd_ex(const tedspad_conv_desc *d, const void *x, const void *w_packed,
                            const int32_t *ktab, const float *scale, const float *shift,
                            const void *residual, void *y, int32_t sigmoid,
                            const tedspad_conv_extras *ex, void *stream);

/* Bottleneck tail + temporal max-pool in one launch: y = MaxPool3d((2,1,1), stride (2,1,1)) of
 * ReLU(conv1x1x1(x)*scale + shift + residual) -- `bn3 / += residual / relu` of the last layer1 block followed by
 * `maxpool2` (large_i3d.py:77-84,139,235). `d` describes the convolution (1x1x1, stride 1, cin 64 or 128, to/ho/wo = its
 * own output extent); y is (n, t/2, h, w) with pixel stride d->ldy. Persistent kernel of tile_cfg 19: the un-pooled
 * tensor is never written. */
int32_t tedspad_conv_pool_t2_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                 const float *shift, const void *residual, void *y, void *stream);

/* First bottleneck of a stage whose two branches are both 1x1x1 stride-1 convs with cin = 64 (layer1.0 of I3Res50):
 * y = ReLU((conv(x, w)*scale + shift) + (conv(x2, w2)*scale2 + shift2)) -- `bn3(conv3(out))`, `downsample(x)`, `+=`, `relu`
 * of large_i3d.py:77-84 in one persistent launch (tile_cfg 19's kernel with two sources): the downsample tensor is never
 * written. `d` describes the first conv (cin = 64; d->relu applies to the sum); x2 has the same pixel grid, pixel stride
 * ldx2, 64 channels; both weight matrices are packed [cout_pad][64]. Each branch is accumulated and scaled in fp32. */
int32_t tedspad_conv_pw_dual_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                 const float *shift, const void *x2, int32_t ldx2, const void *w2_packed,
                                 const float *scale2, const float *shift2, void *y, void *stream);

/* First bottleneck of layer2/3/4 (large_i3d.py:77-84 with a STRIDED downsample branch) as one GEMM on the ping-pong kernel:
 * y = act([W3*s3 | Wd*sd] . [x ; x2 sampled with stride (sh2, sw2)] + shift). `d` describes the first conv (1x1x1, stride 1,
 * cin % 64 == 0, cout % 256 == 0); x2 is (n, t, h2, w2) with cin2 % 64 == 0 channels and pixel stride ldx2; w_packed is the
 * [cout_pad][cin + cin2] matrix with the BatchNorm scales already folded into the rows (scale = ones, shift = b3 + bd). */
int32_t tedspad_conv_p8_dual_fwd(const tedspad_conv_desc *d, const void *x, const void *x2, int32_t cin2, int32_t ldx2, int32_t h2, int32_t w2,
                                 int32_t sh2, int32_t sw2, const void *w_packed, const float *scale, const float *shift, void *y,
                                 void *stream);

/* Tail of a layer1 bottleneck in one launch (csrc/conv_bneck.hip; aux_code/models/large_i3d.py:49-54,69-84):
 *     y = act( bn3(conv3(relu(bn2(conv2(x))))) + residual )                         (plain block)
 *     y = act( bn3(conv3(relu(bn2(conv2(x))))) + bn_d(conv_d(x2)) )                 (first block: downsample branch, stride 1)
 * conv2 = `d2`: a stride-1 'same' 1 x kh x kw convolution with 64 input and 64 output channels (weights `w2_packed` in the
 * tedspad_conv_fwd layout, BatchNorm folded to scale2 / shift2); conv3 / conv_d are 1x1x1 with 64 input channels and cout3 (a multiple of
 * 64, <= 512) output channels. `w3p` is [cout3][KB * 64] 16-bit (KB = 2 with the second source): columns
 * ((a*2 + s)*2 + h)*8 + j = conv3 weight of input channel 32 a + 16 s + 8 (j >> 2) + 4 h + (j & 3) (the k order in which an MFMA
 * accumulator tile is consumed as the next MFMA's operand), columns 64 + c = conv_d weight of input channel c. scale3 / scale_d are the two
 * BatchNorm scales, shift3 the sum of the shifts. The 64-channel tensor between conv2 and conv3 is never written.
 * Residual and result rows go through wave-private LDS images, so that every global access moves whole 128-byte lines (variant bits 0 / 1, which
 * selected that form until ABI 3, are ignored).
 * variant bit 2 (plain block only, t even): MaxPool3d((2,1,1), stride (2,1,1)) of the block's output fused (large_i3d.py:139):
 * y[n][t/2][h][w][ldy] = max over the frame pair; a workgroup runs conv2 for the same 256 pixels of both frames and stores once. */
int32_t tedspad_bneck_tail_fwd(const tedspad_conv_desc *d2, const void *x, const void *w2_packed, const float *scale2, const float *shift2,
                               const void *w3p, const float *scale3, const float *shift3, int32_t cout3, const void *residual, int32_t ldres,
                               const void *x2, int32_t ldx2, const float *scale_d, void *y, int32_t ldy, int32_t relu, int32_t variant, void *stream);

/* WHOLE bottleneck of a plain I3Res50 layer3 block (csrc/conv_bneck_frame.hip) -- replaces the `Bottleneck.forward` of
 * aux_code/models/large_i3d.py:61-84 for blocks without `downsample` on 14 x 14 frames (cin = 1024, cmid = 256):
 *     y = act( bn3(conv3( relu(bn2(conv2( relu(bn1(conv1(x))) ))) )) + x )
 * conv1 1x1x1 (steps1 = cin / 32) or 3x1x1 'same' on TWO-frame clips in its folded form (steps1 = 2 * cin / 32, t == 2: output frame f =
 * Wa . x[frame 0] + Wb . x[frame 1] with (Wa, Wb) = (W[:,:,1], W[:,:,2]) for f = 0 and (W[:,:,0], W[:,:,1]) for f = 1); conv2 1x3x3 stride 1 pad 1;
 * conv3 1x1x1. One workgroup owns a whole frame; both cmid-channel tensors stay in LDS. x / y: (n, t, h, w, cin) 16-bit channels-last with pixel
 * strides ldx / ldy (y must not alias x).
 * Weights are streams of 16 KB SLOT IMAGES, one per 32-deep K step and 256 output channels: for a matrix Wm[rows][K], row block rb, K step ks
 *     img[wc 0..3][j 0..3][lane 0..63][kk 0..7] = Wm[256 rb + 64 wc + r(j, lane & 15)][32 ks + 8 (lane >> 4) + kk]
 * (lane-linear v_mfma_f32_16x16x32 fragments) with r(j, i) = 16 (i >> 2) + 4 j + (i & 3) for conv1 / conv2 (a lane ends with 16 consecutive channels of a pixel:
 * the LDS rows of the mid tensors) and r(j, i) = 4 i + j for conv3 (weights as the B operand: consecutive lanes end with consecutive channels, so residual
 * loads and stores are fully coalesced).
 *   w1_even / w1_odd: steps1 images of conv1 for even / odd frames (the same buffer twice for the 1x1x1 form), K = ci (folded form: frame * cin + ci);
 *   w23: 72 images of conv2 (K = (dh*3 + dw) * cmid + ci), 4 x 8 images of conv3 (row block = 256 output channels, K = ci) and 2 padding images (the ring
 *        fetches two steps past the end).
 * scale / shift: the three folded BatchNorms (cmid, cmid, cin floats, 16-byte aligned). ldy must equal ldx; relu must be non-zero (large_i3d.py:84). */
int32_t tedspad_bneck_frame_fwd(const void *x, int32_t ldx, void *y, int32_t ldy, int32_t n, int32_t t, int32_t h, int32_t w, int32_t cin, int32_t cmid,
                                const void *w1_even, const void *w1_odd, int32_t steps1, const void *w23, const float *scale1, const float *shift1,
                                const float *scale2, const float *shift2, const float *scale3, const float *shift3, int32_t relu, int32_t dtype,
                                void *stream);
int32_t tedspad_bneck_frame_lds_bytes(void);   /* dynamic LDS of the kernel above (one workgroup per CU) */

/* Measurement aid (bench.py `roofline.peak_at_clock`): `workgroups` x 256 threads run 4 x iters v_mfma_f32_32x32x16_f16 per wave; workgroup b writes
 * out[2b] = elapsed shader cycles (s_memtime) and out[2b+1] = elapsed 100 MHz ticks (s_memrealtime): the sustained clock under matrix load is
 * 100 MHz x out[2b] / out[2b+1]. out: 2 x workgroups uint64. */
int32_t tedspad_clock_probe(int32_t iters, int32_t workgroups, void *out, void *stream);

/* Deterministic mode for the training kernels (the reference has no such switch: torch's convolution backward is non-deterministic too unless
 * torch.use_deterministic_algorithms is set). on != 0: the workgroups of a launch pass through their float-atomic sections -- BatchNorm batch statistics in
 * the conv epilogues (tedspad_conv_fwd_ex with `stats`), the channel sums of tedspad_bn_bwd_reduce, the weight-gradient flush of tedspad_conv_wgrad -- one at a
 * time, in blockIdx order (csrc/det_gate.h): bit-identical sums from run to run, at the price of serialising those sections. The caller must keep such
 * launches of one kernel family off concurrent streams, use fixed tile configurations, and form a conv bias gradient with tedspad_bn_bwd_reduce instead of
 * tedspad_bn_bwd_apply's fused one (ted_spad_amd.engine.set_deterministic does all of it). Synchronises the device. tedspad_deterministic_giveups: how many
 * workgroups stopped waiting for their turn since the mode was set (0 unless something is wrong; results are then not reproducible, nothing hangs). */
int32_t tedspad_set_deterministic(int32_t on);
int32_t tedspad_deterministic_giveups(void);

/* Persistent Cin = 3 stem (csrc/conv_stem_pt.hip): conv1 5x7x7 / stride 2 / pad (2,3,3) + bn1 + ReLU of large_i3d.py:133-137,229-231
 * with the temporal half of maxpool1 (MaxPool3d((2,3,3), 2), large_i3d.py:138,232) fused: y[n][to/2][ho][wo][64] =
 * max over the output frame pair (to, to+1) of act(conv * scale + shift); the spatial 3x3 / 2 half of the pool is a
 * tedspad_maxpool_fwd with kt = 1 on y.
 * tedspad_clip_to_tp lays the fp32 (n, c <= 3, t, h, w) clip (W even) out as x_tp[n][tp][h][b][w/2][24] 16-bit: the 48-byte record of
 * pixel (h, 2*wq + b) for output-frame pair tp holds value dt*3 + ci = x[n][ci][4*tp - pad_t + dt][h][2*wq + b], dt = 0..7, zeros
 * outside the clip (stride_t = 2). Output frame 2*tp reads the 32 bytes at offset 0 of a record, frame 2*tp + 1 those at offset 12.
 * w_img (tedspad_stem_pt_wimg_bytes() bytes): [tap = dh*7 + dw][co 0..63][half][8] 16-bit, value v = dt*3 + ci (zero for v >= kt*3)
 * of half v / 8, the two halves of a row stored swapped when (co >> 4) & 1 (the kernel's conflict-free LDS image, copied linearly).
 * t_pairs = output frame pairs (floor(To / 2)); ho = ceil(h / 2), wo = w / 2; nwg = persistent workgroups (0: 256);
 * variant bit 1: 8 waves per workgroup, the output channels split between the two waves of a SIMD (same results); bits 8-10:
 * timing ablations (wrong results, see the kernel). */
int32_t tedspad_clip_to_tp(const float *x, void *x_tp, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc,
                           int64_t st, int64_t sh, int64_t sw, int32_t pad_t, int32_t stride_t, int32_t t_pairs, int32_t dtype, void *stream);
int32_t tedspad_stem_pt_wimg_bytes(void);
int32_t tedspad_stem_pt_fwd(const void *x_tp, const void *w_img, const float *scale, const float *shift, void *y, int32_t n, int32_t t_pairs,
                            int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t ldy, int32_t relu, int32_t nwg,
                            int32_t variant, int32_t dtype, void *stream);
/* The same stem with the WHOLE maxpool1 fused (MaxPool3d((2,3,3), stride 2, no padding), large_i3d.py:138,232): ReLU always;
 * y[n][t_pairs][hp][wp][ldy], hp = (ceil(h/2) - 3) / 2 + 1, wp = (w/2 - 3) / 2 + 1. Workgroups walk column strips of 8 x 16 patches top
 * to bottom, pool each patch over columns in registers and over rows through LDS (the rows a window shares with the patch above are
 * carried); the pooled column a patch shares with its right neighbour is joined by a second small launch from `side`
 * (tedspad_stem_pt_side_bytes() bytes of scratch, 16-byte aligned). The 112 x 112 stem tensor never reaches HBM. */
/* variant bit 2 (with the 8 waves of bit 1): the same taps on v_mfma_f32_16x16x32 -- one MFMA sums a PAIR of taps (K = 32); the chip holds a ~12 % higher
 * clock on that shape under load. w_img then is the tap-pair image (tedspad_stem_pt_wimg16_bytes() bytes): [pair 0..24][co][chunk q][8] 16-bit with chunk
 * q = 2 * (tap of the pair) + (8-value half of its 16 values dt*3 + ci), chunk q of row co stored at chunk q ^ (2 * ((co >> 3) & 1)); pairs in the order
 * phase 0 (dh even): ((dh, 1), (dh, 2)), ((dh, 3), (dh, 4)), ((dh, 5), (dh, 6)) for dh = 0, 2, 4, 6, then ((0, 0), (2, 0)), ((4, 0), (6, 0)); phase 1 (dh odd)
 * likewise for dh = 1, 3, 5, then ((1, 0), (3, 0)), ((5, 0), zeros). Sums the 49 taps in another order than the 32x32x16 form (fp32: one f16 step apart). */
int32_t tedspad_stem_pt_wimg16_bytes(void);
int64_t tedspad_stem_pt_side_bytes(int32_t n, int32_t t_pairs, int32_t h, int32_t w);
int32_t tedspad_stem_pt_pool_fwd(const void *x_tp, const void *w_img, const float *scale, const float *shift, void *y, void *side, int32_t n,
                                 int32_t t_pairs, int32_t h, int32_t w, int32_t hp, int32_t wp, int32_t ldy, int32_t nwg, int32_t variant,
                                 int32_t dtype, void *stream);
/* The pool-fused 16x16x32 stem WITHOUT the layout pass: x is the fp32 (n, c, t, h, w) clip batch exactly as the reference hands it to
 * `I3Res50.extract_features` / `forward` (large_i3d.py:229-232,251-254; produced by dali_extraction.py:38-50 / st_feature_extraction.py:18-26), element
 * strides sn, sc, st, sh, sw = 1, rows 16-byte aligned (w % 4 == 0, x and every stride multiples of 4 elements). The persistent workgroups build their halo
 * images from it themselves (register-staged 16-byte loads requested a whole patch ahead, converted to the 16-bit dtype on the way into LDS): the 9.6 MB
 * per clip of tedspad_clip_to_tp records are never written or read. Same arithmetic, same results as tedspad_clip_to_tp + tedspad_stem_pt_pool_fwd(variant 6).
 * w_img16: the tap-pair image; variant: bits 8..10 timing ablations only. */
int32_t tedspad_stem_pt_pool_clip_fwd(const float *x, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc, int64_t st, int64_t sh,
                                      int64_t sw, int32_t pad_t, int32_t stride_t, int32_t t_pairs, const void *w_img16, const float *scale,
                                      const float *shift, void *y, void *side, int32_t hp, int32_t wp, int32_t ldy, int32_t nwg, int32_t variant,
                                      int32_t dtype, void *stream);

/* The block that ends the default anonymizer's decoder (arch = 'unet++', aux_code/model_loaders.py:17-30; smp 0.3.3 UnetPlusPlusDecoder blocks['x_0_3'] =
 * DecoderBlock(64, 0, 32) followed by SegmentationHead(32, 3, kernel_size 3)) in ONE launch, at full resolution:
 *     y = conv3x3(relu(bn(conv3x3(relu(bn(conv3x3(interpolate(x, 2, 'nearest'))))))) ) + bias         64 -> 32 -> 32 -> 3
 * x: (n, h/2, w/2, 64) channels-last 16-bit, pixel stride ldx; y: (n, 3, h, w) fp32 NCHW (the model's output as the extraction loops consume it,
 * dali_extraction.py:169-173); both 32-channel tensors stay in LDS. w_img: tedspad_unetpp_tail_wimg_bytes() bytes, the three weight tensors in the kernel's
 * LDS image: [hc 2][tap 9][co 32][32 ci] | [tap 9][co 32][32 ci] | [tap 9][co 16 (3 used)][32 ci] 16-bit, the four 16-byte pieces of a 64-byte row stored at
 * piece ^ ((co >> 1) & 3) (unetpp.py packs it); scale / shift: the folded BatchNorms (32 values each), bias3: 3 values. */
int32_t tedspad_unetpp_tail_wimg_bytes(void);
int32_t tedspad_unetpp_tail_fwd(const void *x, int32_t ldx, float *y, int32_t n, int32_t h, int32_t w, const void *w_img, const float *scale1,
                                const float *shift1, const float *scale2, const float *shift2, const float *bias3, int32_t dtype, void *stream);

/* Weight gradient: dw[co][k] += sum over output pixels of dy[m][co] * x[m @ tap(k)][ci(k)], fp32, in the
 * packed [cout_pad][kpad] layout of the forward weights (k ordered (dt,dh,dw,ci)). `dw` must be
 * zeroed by the caller (hipMemsetAsync on the same stream); accumulation uses float atomics.
 * `d` is the FORWARD descriptor of the convolution (d->ldy = pixel stride of dy). Replaces the
 * autograd of nn.Conv3d/Conv2d in loss.backward(): train_anonymizer.py:122,190-191. */
int32_t tedspad_conv_wgrad(const tedspad_conv_desc *d, const void *x, const void *dy, const int32_t *ktab,
                           float *dw, void *stream);

/* nn.MaxPool3d / MaxPool3dSamePadding / nn.MaxPool2d: large_i3d.py:138-139, i3d.py:13-45, unet_parts.py:34 */
int32_t tedspad_maxpool_fwd(const tedspad_pool_desc *d, const void *x, void *y, void *stream);
/* Same, also recording the window-local index of the FIRST maximum of every output element
 * (uint8, laid out (n,to,ho,wo,c) contiguous) for tedspad_maxpool_bwd. */
int32_t tedspad_maxpool_fwd_idx(const tedspad_pool_desc *d, const void *x, void *y, uint8_t *idx, void *stream);

/* AdaptiveAvgPool3d(1) / AvgPool3d([2,7,7]) over `spatial` pixels: large_i3d.py:146,262; i3d.py:293,340.
 * x: (n, spatial, c) 16-bit with pixel stride ldx -> y fp32 (n, c). */
int32_t tedspad_global_avgpool_fwd(const void *x, float *y, int32_t n, int32_t spatial, int32_t c,
                                   int32_t ldx, int32_t dtype, void *stream);

/* nn.AvgPool3d(kernel_size=[kt,kh,kw], stride=(1,1,1)) of InceptionI3d.extract_features (aux_code/models/i3d.py:293-295,336-340) on a
 * Mixed_5c map larger than the kernel: x (n,t,h,w,c) 16-bit channels-last, pixel stride ldx -> y fp32 (n,c,t-kt+1,h-kh+1,w-kw+1). */
int32_t tedspad_avgpool3d_s1_fwd(const void *x, float *y, int32_t n, int32_t t, int32_t h, int32_t w, int32_t c, int32_t ldx, int32_t kt,
                                 int32_t kh, int32_t kw, int32_t dtype, void *stream);

/* fp32 NCTHW clip (as ft.extract_features takes it: large_i3d.py:249) -> 16-bit NTHWC with the
 * channel dim zero-padded to cpad (4: stem pixel-pair form; 8: UNet).  x strides in elements. */
int32_t tedspad_clip_to_channels_last(const float *x, void *y, int32_t n, int32_t c, int32_t t,
                                      int32_t h, int32_t w, int64_t sn, int64_t sc, int64_t st_,
                                      int64_t sh, int64_t sw, int32_t cpad, int32_t dtype, void *stream);

/* 16-bit NTHWC (pixel stride ldx, first c channels) -> fp32 NCTHW contiguous (module boundary out). */
int32_t tedspad_channels_last_to_nchw(const void *x, float *y, int32_t n, int32_t c, int32_t t,
                                      int32_t h, int32_t w, int32_t ldx, int32_t dtype, void *stream);

/* y[b,n] = act( scale[n] * sum_k x[b,k] * w[n,k] + shift[n] ), all fp32 (scale/shift may be NULL).
 * Replaces nn.Linear (+ BatchNorm1d eval + ReLU): I3Res50.fc large_i3d.py:147,245; mlp.fc1/bn1/fc2/bn2
 * aux_code/model_loaders.py:242-253. */
int32_t tedspad_linear_fwd(const float *x, const float *w, const float *scale, const float *shift, float *y,
                           int32_t B, int32_t K, int32_t N, int32_t relu, void *stream);

/* y[b,:] = x[b,:] / max(||x[b,:]||_2, eps): nn.functional.normalize(p=2, dim=1), model_loaders.py:253. */
int32_t tedspad_l2_normalize_rows(const float *x, float *y, int32_t B, int32_t N, float eps, void *stream);

/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) + F.pad to the skip size
 * (unet_parts.py:50,56-62), written straight into its channel slice of the concat buffer (:67).
 * x: (n,h,w,c) pixel stride ldx -> y: (n,ho,wo,c) pixel stride ldy; rows/cols outside
 * [pad_top, pad_top+2h) x [pad_left, pad_left+2w) are zero-filled. */
int32_t tedspad_upsample_bilinear2x_fwd(const void *x, void *y, int32_t n, int32_t h, int32_t w, int32_t c,
                                        int32_t ldx, int32_t ldy, int32_t ho, int32_t wo, int32_t pad_top,
                                        int32_t pad_left, int32_t dtype, void *stream);

/* ---- losses of the anonymizer training step: value AND input gradients in one launch ---------- */

/* NTXentLoss(device, N, temperature, use_cosine)(zis, zjs)  -- aux_code/nt_xent_original.py:49-70.
 * zis, zjs: fp32 (N, C); loss: fp32[1]; dzis/dzjs: fp32 (N, C) = dloss/dz (both or neither NULL).
 * 2N <= 64, even C <= 256 (reference call sites: N = 12, C = 128, T = 0.1, dot similarity). */
int32_t tedspad_ntxent_fwd_bwd(const float *zis, const float *zjs, float *loss, float *dzis, float *dzjs,
                               int32_t N, int32_t C, float temperature, int32_t use_cosine, void *stream);

/* nn.TripletMarginLoss(margin, p=2, eps=1e-6), mean reduction -- train_anonymizer.py:349-350,115.
 * a,p,n: fp32 (B, C); row_ws: fp32[B] workspace; da/dp/dn may all be NULL (value only). */
int32_t tedspad_triplet_fwd_bwd(const float *a, const float *p, const float *n, float *loss, float *row_ws,
                                float *da, float *dp, float *dn, int32_t B, int32_t C, float margin, float eps,
                                void *stream);

/* nn.CrossEntropyLoss() (mean) -- train_anonymizer.py:347,107. logits fp32 (B, C), labels int64 (B). */
int32_t tedspad_cross_entropy_fwd_bwd(const float *logits, const int64_t *labels, float *loss, float *row_ws,
                                      float *dlogits, int32_t B, int32_t C, void *stream);

/* ---- training path: train-mode BatchNorm around the conv kernels, backward of the pooling / resize ops ----
 * Reference: autograd of the torch.nn modules of aux_code/models/{large_i3d,unet_parts}.py under
 * fa_model.train() / ft_model.train() (anonymization_training/train_anonymizer.py:73-75,137-139). */

/* tedspad_bn_finalize + tedspad_scale_shift_act as ONE launch: y = act((z - mean) * gamma * invstd + beta (+ res)) with mean / invstd from the
 * batch sums `stats` ([2][stats_ld]: sum, sum of squares over `count` values per channel); writes mean / invstd (C floats each, kept for the
 * backward pass) and updates running_mean / running_var in place (momentum, unbiased variance; NULL: not tracked). z: (pixels, Cz)
 * with Cz >= C channels per pixel (channels >= C come out as 0); zdtype TEDSPAD_F32, or `dtype` (16-bit conv output, ldz % 8 == 0 -- what
 * the reference's autocast region holds in front of its BatchNorms, train_anonymizer.py:78,151; the batch sums come from the conv's fp32
 * accumulators either way). nn.BatchNorm{2,3}d in train() mode. */
int32_t tedspad_bn_train_apply(const void *z, int32_t zdtype, const float *stats, int32_t stats_ld, int64_t count, const float *gamma, const float *beta,
                               float eps, float momentum, float *running_mean, float *running_var, float *mean, float *invstd, int32_t C,
                               const void *res, void *y, int64_t pixels, int32_t Cz, int32_t ldz, int32_t ldres, int32_t ldy, int32_t relu,
                               int32_t groups, int32_t dtype, void *stream);
/* groups (here and in the two backward entries below): the tensors hold `groups` consecutive blocks of `pixels` rows, each normalised with its OWN
 * statistics (stats / sums: [groups][2][ld], mean / invstd: [groups][Cz]); the running statistics take the groups' momentum updates in order. */

/* From the per-channel sum / sum-of-squares the conv epilogue accumulated (tedspad_conv_extras.stats) over
 * `count` pixels: batch mean / biased variance -> scale = gamma*invstd, shift = beta - mean*scale; updates
 * running_mean / running_var in place (momentum, unbiased variance) when they are not NULL. */
int32_t tedspad_bn_finalize(const float *stats, int32_t stats_ld, int64_t count, const float *gamma, const float *beta,
                            float eps, float momentum, float *running_mean, float *running_var, float *scale,
                            float *shift, float *mean, float *invstd, int32_t C, void *stream);

/* y = act(z*scale[c] + shift[c] (+ res)); z fp32 (pixel stride ldz), res / y 16-bit channels-last. */
int32_t tedspad_scale_shift_act(const float *z, const float *scale, const float *shift, const void *res, void *y,
                                int64_t pixels, int32_t C, int32_t ldz, int32_t ldres, int32_t ldy, int32_t relu,
                                int32_t dtype, void *stream);

/* sums[0][c] += sum_px g, sums[1][c] += sum_px g*xhat  with g = dy*(y>0 if relu), xhat = (z-mean)*invstd
 * (second row skipped when z == NULL: plain per-channel sum = bias gradient). `sums` pre-zeroed.
 * relu with y == NULL (units without a residual input): the mask is recomputed as z*s + b > 0 with the forward pass's own s = gamma*invstd,
 * b = beta - mean*s (gamma / beta: C floats) -- the 16-bit output is not re-read (2 of the 8 bytes per element this pass moves). */
int32_t tedspad_bn_bwd_reduce(const void *dy, const void *y, const void *z, int32_t zdtype, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, float *sums, int32_t sums_ld, int64_t pixels, int32_t C, int32_t lddy, int32_t ldy,
                              int32_t ldz, int32_t relu, int32_t groups, int32_t dtype, void *stream);

/* dz = gamma*invstd*(g - sums[0]/M - xhat*sums[1]/M); optionally dres = g (gradient of a fused residual input); y == NULL with relu as above.
 * dbias ([dbias_slots][C] floats, pre-zeroed, may be NULL): the rows together += sum over pixels of dz -- the gradient of the bias of the conv
 * in front of the BatchNorm (nn.Conv2d(bias=True) + BatchNorm2d in unet_parts.py DoubleConv), gathered here instead of by another pass over dz;
 * workgroup b adds into row b % dbias_slots (several rows: thousands of atomics on one address serialise), the caller sums the rows.
 * C <= 3520 (the kernel keeps 16 bytes of folded terms per channel in LDS). */
int32_t tedspad_bn_bwd_apply(const void *dy, const void *y, const void *z, int32_t zdtype, const float *mean, const float *invstd,
                             const float *gamma, const float *beta, const float *sums, int32_t sums_ld, void *dz, void *dres,
                             float *dbias, int32_t dbias_slots, int64_t pixels, int32_t C, int32_t lddy, int32_t ldy, int32_t ldz, int32_t lddz,
                             int32_t lddres, int32_t relu, int32_t groups, int32_t dtype, void *stream);

/* dx[i] = (add ? add[i] : 0) + sum over pooling windows o containing i of dy[o]*[argmax(o) == i]; `d` = forward
 * desc, idx from tedspad_maxpool_fwd_idx (first maximum wins, as torch). relu_mask != 0 additionally zeroes dx
 * where x <= 0 (x is a ReLU output: folds that ReLU's backward in). */
int32_t tedspad_maxpool_bwd(const tedspad_pool_desc *d, const void *x, const uint8_t *idx, const void *dy, int32_t lddy,
                            const void *add, int32_t ldadd, void *dx, int32_t lddx, int32_t relu_mask, void *stream);

/* dx[n,p,c] = dfeat[n,c] / spatial, zeroed where mask <= 0 (mask may be NULL). */
int32_t tedspad_global_avgpool_bwd(const float *dfeat, const void *mask, int32_t ldmask, void *dx, int32_t n,
                                   int32_t spatial, int32_t c, int32_t lddx, int32_t dtype, void *stream);

/* F.interpolate(scale_factor=2, mode="nearest") of the UNet++ decoder blocks (segmentation_models_pytorch 0.3.3,
 * decoders/unetplusplus/decoder.py, DecoderBlock.forward; the anonymizer of model_loaders.py:17-30), written straight into its channel
 * slice of the block's concat buffer: y[n][2h+a][2w+b][0..c) = x[n][h][w][0..c), 16-bit elements, pixel strides ldx / ldy.
 * tedspad_copy_channels copies a channel slice (npix pixels x c channels) between two channels-last buffers: the dense skip pathway
 * concatenates some tensors into two different blocks' inputs (torch.cat in UnetPlusPlusDecoder.forward). */
int32_t tedspad_upsample_nearest2x_fwd(const void *x, void *y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, int32_t ldy,
                                       void *stream);
int32_t tedspad_copy_channels(const void *x, void *y, int64_t npix, int32_t c, int32_t ldx, int32_t ldy, void *stream);
/* Their backward passes (the autograd of F.interpolate(nearest) and of a tensor with several consumers; the train-mode anonymizer,
 * train_anonymizer.py:73-123): dx[n][h][w] (+)= the fp32 sum of the 2 x 2 block of dy; y[:, 0..c) += x[:, 0..c). */
int32_t tedspad_upsample_nearest2x_bwd(const void *dy, void *dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldy, int32_t ldx,
                                       int32_t accumulate, int32_t dtype, void *stream);
int32_t tedspad_add_channels(const void *x, void *y, int64_t npix, int32_t c, int32_t ldx, int32_t ldy, int32_t dtype, void *stream);

/* Backward of tedspad_upsample_bilinear2x_fwd: dy is the (n,ho,wo,c) slice the forward wrote, dx is (n,h,w,c). */
int32_t tedspad_upsample_bilinear2x_bwd(const void *dy, void *dx, int32_t n, int32_t h, int32_t w, int32_t c,
                                        int32_t lddy, int32_t lddx, int32_t ho, int32_t wo, int32_t pad_top,
                                        int32_t pad_left, int32_t dtype, void *stream);

/* fp32 (n,c,thw) gradient -> 16-bit (n,thw,8) channels-last; if y_sigmoid != NULL multiplies by y*(1-y)
 * (backward of the UNet's sigmoid output, unet_model.py:37). */
int32_t tedspad_nchw_grad_to_channels_last(const float *dy, const float *y_sigmoid, void *out, int32_t n, int32_t c,
                                           int64_t thw, int32_t dtype, void *stream);

/* 16-bit (n,t,h,w,ldx) first c channels -> fp32 tensor with arbitrary element strides (gradient w.r.t. a clip view). */
int32_t tedspad_channels_last_to_nchw_strided(const void *x, float *y, int32_t n, int32_t c, int32_t t, int32_t h,
                                              int32_t w, int32_t ldx, int64_t sn, int64_t sc, int64_t st_, int64_t sh,
                                              int64_t sw, int32_t dtype, void *stream);

/* ---- fp32 head ops of the training step (B x C matrices with tiny B): mlp / fc of wrapper_i3d ---- */
/* nn.BatchNorm1d in train mode (+ ReLU): model_loaders.py:252-253 under ft_model.train(). */
int32_t tedspad_bn1d_train_fwd(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                               float *running_mean, float *running_var, float *y, float *mean, float *invstd,
                               int32_t B, int32_t C, int32_t relu, void *stream);
int32_t tedspad_bn1d_train_bwd(const float *dy, const float *x, const float *y, const float *mean, const float *invstd,
                               const float *gamma, float *dx, float *dgamma, float *dbeta, int32_t B, int32_t C,
                               int32_t relu, void *stream);
/* backward of tedspad_l2_normalize_rows (x = its INPUT). */
int32_t tedspad_l2_normalize_rows_bwd(const float *x, const float *dy, float *dx, int32_t B, int32_t N, float eps,
                                      void *stream);
/* out = a * b * scale (dropout mask application: I3Res50.drop, large_i3d.py:148,242). */
int32_t tedspad_mul_f32(const float *a, const float *b, float *out, int64_t n, float scale, void *stream);

/* Device-side weight packing: fp32 (co, ci, kt, kh, kw) parameter -> the 16-bit [rows_pad][kpad] matrix of
 * tedspad_conv_fwd. mode 0: forward matrix (rows = co; K ordered (dt,dh,dw,c) over the kernel-form channels
 * cink / width taps kwk; pair_shift >= 0 selects the stem's pixel-pair form, -1 plain zero-padded channels).
 * mode 1: data-gradient matrix of one parity class (rows = kernel-form input channels, K ordered (et,eh,ew,co));
 * dgrad_geo = host int32[9] {Et,Eh,Ew, ct,ch,cw, st,sh,sw}: tap = c + s*(E-1-e) per dim. `scale` (per co, may be
 * NULL) is multiplied in (folded eval-mode BatchNorm). */
int32_t tedspad_pack_conv_weights(const float *w, const float *scale, void *out, int32_t co, int32_t ci, int32_t kt,
                                  int32_t kh, int32_t kw, int32_t cink, int32_t kwk, int32_t pair_shift, int32_t mode,
                                  int32_t rows, int32_t rows_pad, int32_t kpad, const int32_t *dgrad_geo, int32_t dtype,
                                  void *stream);

/* ---- multi-job weight refresh (csrc/pack.hip) -------------------------------------------------------------------
 * After an optimizer step every 16-bit weight image of the updated network is stale: the reference simply reads its fp32
 * parameters again (nn.Conv3d / FrozenBN in large_i3d.py:8-38, 61-84; train_anonymizer.py:123,193 `optimizer.step()`), here the
 * images and the folded BatchNorm vectors are rewritten IN PLACE by ONE launch per kind, from a job table that stays on the
 * device (every address in it is static). `jobs`: HOST array (the library validates it and fills block0 / nblocks);
 * `table_dev`: njobs * sizeof(job) device bytes; upload != 0 copies jobs -> table_dev on `stream` first (needed once, and again
 * only when a job changed). */
typedef struct tedspad_pack_job {           /* one tedspad_pack_conv_weights call */
    const float *w, *scale;
    void *out;
    int32_t co, ci, kt, kh, kw, cink, kwk, pair_shift, mode, rows, rows_pad, kpad;
    int32_t geo[9];                         /* mode 1 (data gradient): {Et,Eh,Ew, ct,ch,cw, st,sh,sw} */
    int32_t dtype;
    int32_t block0, nblocks;                /* filled by the library */
} tedspad_pack_job;
int32_t tedspad_pack_multi(tedspad_pack_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream);

typedef struct tedspad_fold_job {           /* one tedspad_bn_fold call, written zero-padded to n (and n2) floats */
    const float *gamma, *beta, *mean, *var, *conv_bias;   /* gamma NULL: no BatchNorm -- scale = 1, shift = conv_bias (or 0) */
    float *scale, *shift;                   /* n floats each (either may be NULL) */
    float *scale2, *shift2;                 /* optional second copy, n2 floats each */
    double eps;
    int32_t C, n, n2, reserved;
} tedspad_fold_job;
int32_t tedspad_fold_multi(tedspad_fold_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream);

/* packed fp32 weight-gradient accumulators of tedspad_conv_wgrad ([co_pad][kpad], K ordered (dt,dh,dw,c) over cink channels) ->
 * the parameters' gradients in the reference layout (co, ci, kt, kh, kw): grad = (accumulate ? grad : 0) + dw * row_scale[co]. */
typedef struct tedspad_wgrad_unpack_job {
    const float *dw;
    float *grad;
    const float *row_scale;                 /* per output channel, may be NULL */
    int32_t co, ci, kt, kh, kw, cink, kpad, accumulate;
    int32_t block0, nblocks;                /* filled by the library */
} tedspad_wgrad_unpack_job;
int32_t tedspad_wgrad_unpack_multi(tedspad_wgrad_unpack_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream);

/* Eval-mode BatchNorm folded to y = x*scale + shift (fp64 inside, rounded once): scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale (+ conv_bias*scale). FrozenBN / .eval() semantics (large_i3d.py:8-38, i3d.py:113-116). */
int32_t tedspad_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, const float *conv_bias,
                        double eps, int32_t C, float *scale, float *shift, void *stream);

/* ---- the steps either side of the encoder ---------------------------------------- */
/* Antialiased bilinear resize weights of one axis, as torch builds them for F.interpolate(mode='bilinear',
 * antialias=True, align_corners=False) -- the call behind torchvision F.resize(antialias=True) on a float tensor
 * (feature_extraction/dali_extraction.py:49). Host functions: `table` receives out_size entries of
 * (2 + taps) 32-bit words {first input index, tap count, float weights[taps]}; copy it to the device. */
int32_t tedspad_resize_aa_taps(int32_t in_size, int32_t out_size);
int32_t tedspad_resize_aa_table(int32_t in_size, int32_t out_size, int32_t *table);

/* DALIDataloader.val_augmentations (dali_extraction.py:38-50) for one crop box: frames (T,H,W,C) interleaved,
 * uint8 or float (in_is_float) -> value / divisor (255) -> crop [y0,y0+ch) x [x0,x0+cw) -> antialiased bilinear
 * resize to (oh,ow) with the two device tables above (for ch->oh and cw->ow) -> optional horizontal flip (ten-crop)
 * -> fp32 out[t*so_t + c*so_c + y*so_h + x*so_w] (element strides: (T,C,h,w) as the reference returns it, or
 * straight into a (C,T,h,w) clip of the encoder's batch). */
int32_t tedspad_frames_crop_resize(const void *frames, int32_t in_is_float, int32_t T, int32_t H, int32_t W, int32_t C,
                                   int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t oh, int32_t ow,
                                   const int32_t *ytab, const int32_t *xtab, float divisor, int32_t flip, float *out,
                                   int64_t so_t, int64_t so_c, int64_t so_h, int64_t so_w, void *stream);

/* The same frames -> /divisor -> crop -> antialiased resize (-> flip), written as the INPUT RECORDS of the persistent stem (tedspad_clip_to_tp's layout,
 * X[n][tp][oh][b][ow/2][24] 16-bit) instead of an fp32 clip: the step between the decoder's frames and I3Res50's conv1 (dali_extraction.py:38-50 ->
 * large_i3d.py:229) without the 9.6 MB per clip of fp32 in between. Clip n of n_clips takes source frames first + n*clip_step + f*frame_step,
 * f = 0 .. t_clip-1 (HybridValPipe: sequence_length 16, stride fix_skip = 2, step 32; dali_extraction.py:62-73); frames outside [0, T) are zero frames
 * (pad_sequences). pad_t / stride_t / t_pairs: the stem's temporal geometry as in tedspad_clip_to_tp. Every value is the fp32 value
 * tedspad_frames_crop_resize computes, rounded to `dtype` as tedspad_clip_to_tp rounds it: records bit-identical to crop_resize + clip_to_tp. */
int32_t tedspad_frames_crop_resize_tp(const void *frames, int32_t in_is_float, int32_t T, int32_t H, int32_t W, int32_t C, int32_t n_clips,
                                      int32_t first, int32_t clip_step, int32_t frame_step, int32_t t_clip, int32_t y0, int32_t x0, int32_t ch,
                                      int32_t cw, int32_t oh, int32_t ow, const int32_t *ytab, const int32_t *xtab, float divisor, int32_t flip,
                                      void *records, int32_t pad_t, int32_t stride_t, int32_t t_pairs, int32_t dtype, void *stream);

/* `shanghai_frames_dataset.augmentation` (feature_extraction/shanghai_dl.py:27-40): uint8 frames (T,H,W,C) -> crop box -> Pillow's
 * two-pass BILINEAR resize with its 8-bit intermediate image (what torchvision's resize does for a PIL image) -> to_tensor (/255)
 * -> fp32 out[t*so_t + c*so_c + y*so_h + x*so_w]. ytab / xtab: device tables of (2 + taps) int32 per output index
 * {first input index, count, coefficients with 22 fractional bits}, built by the host as libImaging/Resample.c does
 * (ted_spad_amd/preprocess.pil_table). Bit-exact with Pillow (tests/test_hip_feed.py). */
int32_t tedspad_frames_crop_resize_pil(const void *frames, int32_t T, int32_t H, int32_t W, int32_t C, int32_t y0, int32_t x0, int32_t ch,
                                       int32_t cw, int32_t oh, int32_t ow, const int32_t *ytab, int32_t ytaps, const int32_t *xtab,
                                       int32_t xtaps, float *out, int64_t so_t, int64_t so_c, int64_t so_h, int64_t so_w, void *stream);

/* MGFN feature feed (anomaly_detection_mgfn/datasets/dataset.py:65-100): feat (T, ncrops, F) fp32.
 * length > 0 (train): out (ncrops, length, F+1) = process_feat (utils/utils.py:34-42: means over the
 * numpy.linspace(0,T,length+1,dtype=int) segments, a single row where a segment is empty) + L2 magnitude channel.
 * length == 0 (test_mode): out (T, ncrops, F+1) = the rows + magnitude channel. */
int32_t tedspad_segment_pool_mag(const float *feat, int32_t T, int32_t ncrops, int32_t F, int32_t length, float *out,
                                 void *stream);

/* f16 head-room made observable (large_i3d.py:78-79 runs in fp32 / autocast fp16, where an overflow is an inf; the inference stores here SATURATE at
 * +-65504 instead, silently): counts, in a channels-last 16-bit tensor of `rows` pixels x `c` channels (pixel stride ldx), the elements AT the largest finite
 * value of the type (f16: what a clamped store writes; bf16: never) into out2[0] and the non-finite ones into out2[1] (both ADDED to). */
int32_t tedspad_count_saturated(const void *x, int64_t rows, int32_t c, int32_t ldx, int32_t dtype, uint32_t *out2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TEDSPAD_HIP_H */
