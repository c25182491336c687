// Shared between the conv kernels: kernel parameter block, constants.
#pragma once
#include "common.h"

namespace tedspad {

struct ConvKP {
    const uint16_t *x;
    const uint16_t *w;
    const int2 *ktab;
    const float *scale;
    const float *shift;
    const uint16_t *res;
    uint16_t *y;
    int M, Cout, Kpad, nk;
    int Ti, Hi, Wi, ldx;
    int To, Ho, Wo, ldy, ldres;
    int kt, kh, kw;
    int st, sh, sw, pt, ph, pw;
    int relu, sigmoid, pointwise;
    int tiles_n;
    float inv_wo, inv_ho, inv_to;   // fp32 reciprocals for the row decode
    int cin, utap;                  // utap: cin % 64 == 0, K-tile -> tap is arithmetic (no table)
    // optional epilogue extras (training path)
    const uint16_t *mask;   // out = mask > 0 ? out : 0   (ReLU backward fused into the dgrad that produces d(input))
    float *stats;           // [2][stats_ld]: per-channel sum / sum of squares of the pre-activation (BatchNorm batch statistics)
    float *y32;             // optional fp32 copy of the output (train-mode BN keeps the pre-normalisation conv output exact)
    int ldmask, stats_ld, ldy32;
    int fold_hw, fold_c, fold_f;   // ping-pong kernel only: output channel n of row m = sample * fold_hw + px is stored as channel n % fold_c of row
                            // (sample * fold_f + n / fold_c) * fold_hw + px (fold_f = cout / fold_c output FRAMES folded into the channel dimension);
                            // scale / shift are fold_c long. 0: off
    int stats_rows;         // rows per statistics GROUP (0: all rows one group); group g accumulates into stats + g * 2 * stats_ld. A tile may straddle ONE boundary
    int ostrided;           // output pixel (n,to,ho,wo) -> (n, to*ost+oot, ho*osh+ooh, wo*osw+oow) of a (TF,HF,WF) tensor
    int ost, osh, osw, oot, ooh, oow, TF, HF, WF;
    // second source of the dual pointwise launch (conv_pw.hip, DUAL): x2 has the pixel grid of x, cin2 = 64
    const uint16_t *x2, *w2;
    const float *scale2, *shift2;
    int ldx2;
    int nk1, Hi2, Wi2, sh2, sw2;     // conv_p8.hip DUAL: K tiles of the first source; grid and spatial stride of the second
    float sat;              // clamp of the f16 stores: 65504 (saturate), +inf on the training path (tedspad_conv_extras.nosat)
};

constexpr int BK = 64;                  // K elements per LDS tile row (8 chunks of 16 bytes)
constexpr int KTAB_MAX_BYTES = 10240;   // Kpad <= 10240 (one int2 per 8 K elements)
constexpr int KTAB_SMALL_BYTES = 1024;  // "short-K" configs: Kpad <= 1024, smaller LDS -> 2-3 workgroups per CU


// conv_pw.hip: persistent pointwise kernel for 1x1x1 stride-1 convolutions with cin = 64 / 128 (tile_cfg 19).
int32_t launch_conv_pw(int dtype, const ConvKP &p, hipStream_t s, bool pool_t = false);

// conv_p8.hip: ping-pong 256 x 256 kernel (two waves per SIMD one barrier apart) for cin % 64 == 0, cout % 256 == 0 (tile_cfg 25).
int32_t launch_conv_p8(int dtype, const ConvKP &p, hipStream_t s, int mf = 32);   // mf 16: tile_cfg 26 (16x16x32 MFMA)

// conv_flat.hip: flat-halo kernel for stride-1 'same' 1 x kh x kw convs with cin = 64, cout <= 64 (tile_cfg 27).
int32_t launch_conv_flat(int dtype, const ConvKP &p, int cin, hipStream_t s);
// ... and its temporal sibling for stride-1 'same' kt x 1 x 1 convs with cin % 64 == 0, cout <= 64, T <= 4 (tile_cfg 28).
int32_t launch_conv_tflat(int dtype, const ConvKP &p, int N, int cin, hipStream_t s);

// conv_patch.hip: 16 x 16 patch-halo kernel for stride-1 'same' 1 x kh x kw convs with cin % 64 == 0, cout <= 128 on wide frames (tile_cfg 32).
// gathered concatenation (tedspad_conv_extras.nchunk_src): per 64-channel chunk its own source tensor, optionally read through a nearest x2 upsample
struct PatchSrc {
    const uint16_t *ptr[8];
    int ld[8];
    int up, n;
};
int32_t launch_conv_patch(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, int mode = 0, const PatchSrc *src = nullptr);
int32_t launch_conv_patch2(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src = nullptr, int flat = 0);   // tile_cfg 38 / 39: two patches (two flat tiles) per workgroup   // mode 1: tile_cfg 33 (256 consecutive pixels), 2: tile_cfg 34 (temporal)

// conv_patch3.hip: PERSISTENT two-patch kernel (tile_cfg 40): 8 waves, one workgroup per CU, halo double-buffered, weight ring of six stages (resident for cin <= 64),
// epilogue straight from the accumulators; 1 x 3 x 3 'same' convs with 32 < cout <= 64
int32_t launch_conv_patch3(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src = nullptr);
void patch3_set_det(int on);

}  // namespace tedspad
