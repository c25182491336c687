// The two HBM-bound steps either side of the encoder, for gfx950:
//   * frames -> /255 -> crop box -> antialiased bilinear resize (-> optional horizontal flip) -> fp32 clip
//     (DALIDataloader.val_augmentations, feature_extraction/dali_extraction.py:38-50)
//   * (T, ncrops, F) features -> 32-segment mean pooling + L2 magnitude channel
//     (process_feat, anomaly_detection_mgfn/utils/utils.py:34-42; Dataset.__getitem__, datasets/dataset.py:65-100)
// Both are pure bandwidth work: every global access below is a run of consecutive bytes / floats across the
// lanes of a wave, intermediates live in LDS, and grids are one workgroup per output row (>> 256 workgroups).
#include <math.h>

#include "common.h"

namespace tedspad {
namespace {

// ------------------------------------------------------------------------------------------------------------
// antialiased separable bilinear resize, weights as torch builds them for F.interpolate(mode='bilinear',
// antialias=True, align_corners=False) -- which is what torchvision's F.resize(antialias=True) calls on a
// float tensor (dali_extraction.py:49). Table entry of output index i: {xmin, xsize, w[0..taps)}.
// ------------------------------------------------------------------------------------------------------------
inline int aa_taps(int in, int out) {
    const float scale = (float)in / (float)out;
    const float support = scale >= 1.f ? scale : 1.f;
    return (int)ceilf(support) * 2 + 1;
}

inline void aa_table(int in, int out, int32_t *tab) {
    const int taps = aa_taps(in, out);
    const float scale = (float)in / (float)out;
    const float support = scale >= 1.f ? scale : 1.f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    for (int i = 0; i < out; i++) {
        int32_t *e = tab + (size_t)i * (2 + taps);
        float *w = (float *)(e + 2);
        const float center = scale * ((float)i + 0.5f);
        long lo = (long)(center - support + 0.5f);
        long hi = (long)(center + support + 0.5f);
        if (lo < 0) lo = 0;
        if (hi > in) hi = in;
        int n = (int)(hi - lo);
        if (n > taps) n = taps;
        float total = 0.f;
        for (int j = 0; j < taps; j++) w[j] = 0.f;
        for (int j = 0; j < n; j++) {
            float x = ((float)(j + lo) - center + 0.5f) * invscale;
            x = fabsf(x);
            w[j] = x < 1.f ? 1.f - x : 0.f;
            total += w[j];
        }
        if (total != 0.f)
            for (int j = 0; j < n; j++) w[j] /= total;
        e[0] = (int32_t)lo;
        e[1] = n;
    }
}

struct ResizeKP {
    const void *in;
    float *out;
    const int32_t *ytab, *xtab;
    int T, H, W, C;          // frames are (T, H, W, C) with C interleaved (DALI / decoder layout)
    int y0, x0, ch, cw;      // crop box
    int oh, ow, ytaps, xtaps;
    int flip;
    float div;               // 255: applied as a true division per sample, like `video / 255.`
    long so_t, so_c, so_h, so_w;
};

template <typename In> __device__ __forceinline__ float load_px(const In *p, float div);
template <> __device__ __forceinline__ float load_px<uint8_t>(const uint8_t *p, float div) { return (float)(*p) / div; }
template <> __device__ __forceinline__ float load_px<float>(const float *p, float div) { return *p / div; }

// One workgroup per (frame, output row). Pass 1 (vertical): every lane owns interleaved (x, c) elements of the
// cropped input row span, consecutive lanes = consecutive bytes, and reduces the <= ytaps input rows of this output
// row into LDS. Pass 2 (horizontal): lanes own output pixels and read their taps from LDS.
template <typename In>
__global__ __launch_bounds__(256) void crop_resize_aa_kernel(const ResizeKP p) {
    extern __shared__ float row[];      // cw * C
    const int t = blockIdx.x / p.oh, oy = blockIdx.x % p.oh;
    const int32_t *ye = p.ytab + (size_t)oy * (2 + p.ytaps);
    const int ymin = ye[0], yn = ye[1];
    const float *wy = (const float *)(ye + 2);
    const int span = p.cw * p.C;
    const In *base = (const In *)p.in + (((long)t * p.H + p.y0 + ymin) * p.W + p.x0) * p.C;
    const long rstride = (long)p.W * p.C;
    for (int e = threadIdx.x; e < span; e += 256) {
        float acc = 0.f;
        for (int j = 0; j < yn; j++) acc += wy[j] * load_px<In>(base + j * rstride + e, p.div);
        row[e] = acc;
    }
    __syncthreads();
    for (int ox = threadIdx.x; ox < p.ow; ox += 256) {
        const int32_t *xe = p.xtab + (size_t)ox * (2 + p.xtaps);
        const int xmin = xe[0], xn = xe[1];
        const float *wx = (const float *)(xe + 2);
        const int oxw = p.flip ? p.ow - 1 - ox : ox;
        for (int c = 0; c < p.C; c++) {
            float acc = 0.f;
            for (int j = 0; j < xn; j++) acc += wx[j] * row[(xmin + j) * p.C + c];
            p.out[t * p.so_t + c * p.so_c + oy * p.so_h + oxw * p.so_w] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// uint8 frames -> crop box -> Pillow's two-pass BILINEAR resize (8-bit intermediate image) -> /255 -> fp32:
// `shanghai_frames_dataset.augmentation` (feature_extraction/shanghai_dl.py:27-40: to_pil_image, center_crop, resize(antialias),
// to_tensor). Pillow (libImaging/Resample.c, 8 bits per channel) resamples HORIZONTALLY first, rounds that image to uint8, then
// vertically, with fixed-point coefficients of 22 fractional bits: out = clip8((2^21 + sum px * kk) >> 22). The integer tables
// {first index, count, kk[0..taps)} are built on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do (preprocess.py).
// One workgroup per (frame, output row): a thread owns output pixels (x, c) and runs the horizontal pass of the <= ytaps
// temporary rows it needs itself (every input row feeds ~2 output rows: 2x redundant integer MACs on a memory-bound op).
// ------------------------------------------------------------------------------------------------------------
struct ResizeU8KP {
    const uint8_t *in;
    float *out;
    const int32_t *ytab, *xtab;
    int T, H, W, C, y0, x0, oh, ow, ytaps, xtaps;
    long so_t, so_c, so_h, so_w;
};

__device__ __forceinline__ int clip8_fixed(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void crop_resize_pil_kernel(const ResizeU8KP p) {
    const int t = blockIdx.x / p.oh, oy = blockIdx.x % p.oh;
    const int32_t *ye = p.ytab + (size_t)oy * (2 + p.ytaps);
    const int ymin = ye[0], yn = ye[1];
    const uint8_t *base = p.in + (((long)t * p.H + p.y0 + ymin) * p.W + p.x0) * p.C;
    const long rstride = (long)p.W * p.C;
    for (int e = threadIdx.x; e < p.ow * p.C; e += 256) {
        const int ox = e / p.C, c = e - ox * p.C;
        const int32_t *xe = p.xtab + (size_t)ox * (2 + p.xtaps);
        const int xmin = xe[0], xn = xe[1];
        int acc = 1 << 21;
        for (int j = 0; j < yn; j++) {
            const uint8_t *r = base + j * rstride + (long)xmin * p.C + c;
            int hs = 1 << 21;
            for (int i = 0; i < xn; i++) hs += (int)r[(long)i * p.C] * xe[2 + i];
            acc += clip8_fixed(hs) * ye[2 + j];                  // the horizontal pass's uint8 pixel of temporary row ymin + j
        }
        p.out[t * p.so_t + c * p.so_c + oy * p.so_h + ox * p.so_w] = (float)clip8_fixed(acc) / 255.f;   // to_tensor: uint8 -> float / 255
    }
}

// ------------------------------------------------------------------------------------------------------------
// process_feat + magnitude. One workgroup per (crop, segment); lanes stride over F (consecutive floats).
// length > 0 : out (ncrops, length, F+1): row s = mean of feature rows r[s]..r[s+1]-1 (or row r[s] if empty),
//              r = linspace(0, T, length+1) truncated to int exactly as numpy does it in float64.
// length == 0: test-mode layout out (T, ncrops, F+1): the rows themselves + magnitude.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_pool_mag_kernel(const float *feat, float *out, int T, int ncrops, int F, int length) {
    __shared__ float red[256];
    const int seg = blockIdx.x, crop = blockIdx.y;
    int r0, r1;
    if (length > 0) {
        const double step = (double)T / (double)length;          // numpy.linspace: start + arange * step, last = stop
        r0 = (int)((double)seg * step);
        r1 = seg + 1 == length ? T : (int)((double)(seg + 1) * step);
    } else {
        r0 = seg;
        r1 = seg + 1;
    }
    float *o = length > 0 ? out + ((size_t)crop * length + seg) * (F + 1) : out + ((size_t)seg * ncrops + crop) * (F + 1);
    float ss = 0.f;
    for (int f = threadIdx.x; f < F; f += 256) {
        float v;
        if (r1 > r0) {
            float acc = 0.f;
            for (int r = r0; r < r1; r++) acc += feat[((size_t)r * ncrops + crop) * F + f];
            v = acc / (float)(r1 - r0);
        } else {
            v = feat[((size_t)r0 * ncrops + crop) * F + f];
        }
        o[f] = v;
        ss += v * v;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) o[F] = sqrtf(red[0]);
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_resize_aa_taps(int32_t in_size, int32_t out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    return aa_taps(in_size, out_size);
}

extern "C" int32_t tedspad_resize_aa_table(int32_t in_size, int32_t out_size, int32_t *table) {
    TS_REQUIRE(in_size > 0 && out_size > 0 && table, "tedspad_resize_aa_table: bad arguments");
    aa_table(in_size, out_size, table);
    return TEDSPAD_OK;
}

extern "C" int32_t tedspad_frames_crop_resize(const void *frames, int32_t in_is_float, int32_t T, int32_t H, int32_t W, int32_t C,
                                              int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t oh, int32_t ow,
                                              const int32_t *ytab, const int32_t *xtab, float divisor, int32_t flip, float *out,
                                              int64_t so_t, int64_t so_c, int64_t so_h, int64_t so_w, void *stream) {
    TS_REQUIRE(frames && out && ytab && xtab, "tedspad_frames_crop_resize: null pointer");
    TS_REQUIRE(T > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && oh > 0 && ow > 0, "tedspad_frames_crop_resize: bad sizes");
    TS_REQUIRE(y0 >= 0 && x0 >= 0 && ch > 0 && cw > 0 && y0 + ch <= H && x0 + cw <= W,
               "tedspad_frames_crop_resize: crop box (%d,%d,%d,%d) outside the %dx%d frame", y0, x0, ch, cw, H, W);
    TS_REQUIRE(divisor != 0.f, "tedspad_frames_crop_resize: divisor must be non-zero");
    const size_t lds = (size_t)cw * C * sizeof(float);
    TS_REQUIRE(lds <= 64 * 1024, "tedspad_frames_crop_resize: crop width %d x %d channels exceeds the 64 KB LDS row buffer", cw, C);
    ResizeKP p;
    p.in = frames; p.out = out; p.ytab = ytab; p.xtab = xtab;
    p.T = T; p.H = H; p.W = W; p.C = C; p.y0 = y0; p.x0 = x0; p.ch = ch; p.cw = cw; p.oh = oh; p.ow = ow;
    p.ytaps = aa_taps(ch, oh); p.xtaps = aa_taps(cw, ow); p.flip = flip; p.div = divisor;
    p.so_t = so_t; p.so_c = so_c; p.so_h = so_h; p.so_w = so_w;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)((long)T * oh));
    if (in_is_float) hipLaunchKernelGGL(crop_resize_aa_kernel<float>, g, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(crop_resize_aa_kernel<uint8_t>, g, dim3(256), lds, s, p);
    return check_launch("tedspad_frames_crop_resize");
}

extern "C" int32_t tedspad_segment_pool_mag(const float *feat, int32_t T, int32_t ncrops, int32_t F, int32_t length, float *out,
                                            void *stream) {
    TS_REQUIRE(feat && out && T > 0 && ncrops > 0 && F > 0 && length >= 0, "tedspad_segment_pool_mag: bad arguments");
    const dim3 g((unsigned)(length > 0 ? length : T), (unsigned)ncrops);
    hipLaunchKernelGGL(segment_pool_mag_kernel, g, dim3(256), 0, (hipStream_t)stream, feat, out, T, ncrops, F, length);
    return check_launch("tedspad_segment_pool_mag");
}

extern "C" int32_t tedspad_frames_crop_resize_pil(const void *frames, int32_t t, int32_t h, int32_t w, int32_t c, int32_t y0, int32_t x0, int32_t ch,
                                                  int32_t cw, int32_t oh, int32_t ow, const int32_t *ytab, int32_t ytaps, const int32_t *xtab,
                                                  int32_t xtaps, float *out, int64_t so_t, int64_t so_c, int64_t so_h, int64_t so_w, void *stream) {
    TS_REQUIRE(frames && ytab && xtab && out && t > 0 && h > 0 && w > 0 && c > 0 && c <= 4 && oh > 0 && ow > 0 && ytaps > 0 && xtaps > 0,
               "tedspad_frames_crop_resize_pil: bad arguments");
    TS_REQUIRE(y0 >= 0 && x0 >= 0 && ch > 0 && cw > 0 && y0 + ch <= h && x0 + cw <= w, "tedspad_frames_crop_resize_pil: crop box outside the frame");
    TS_REQUIRE((long)t * oh < (1L << 31), "tedspad_frames_crop_resize_pil: too many rows");
    ResizeU8KP p;
    p.in = (const uint8_t *)frames; p.out = out; p.ytab = ytab; p.xtab = xtab;
    p.T = t; p.H = h; p.W = w; p.C = c; p.y0 = y0; p.x0 = x0; p.oh = oh; p.ow = ow; p.ytaps = ytaps; p.xtaps = xtaps;
    p.so_t = so_t; p.so_c = so_c; p.so_h = so_h; p.so_w = so_w;
    hipLaunchKernelGGL(crop_resize_pil_kernel, dim3((unsigned)((long)t * oh)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("tedspad_frames_crop_resize_pil");
}
