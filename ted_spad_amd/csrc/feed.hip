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

// bytes a[0..3] as one dword (byte k in bits 8k..): aligned 4-byte loads (one, or two joined by v_alignbyte_b32 when `a` is not 4-byte aligned) where they stay
// inside the buffer, single bytes at its very end. `a & 3` is wave-uniform for the callers below (uniform row pointer + 4 * lane group).
__device__ __forceinline__ unsigned load_u8x4(const uint8_t *a, const uint8_t *end) {
    const unsigned sh = (unsigned)((uintptr_t)a & 3u);
    if (a + 8 <= end) {
        const unsigned *ap = reinterpret_cast<const unsigned *>(a - sh);
        const unsigned w0 = ap[0];
        if (sh == 0) return w0;
        return __builtin_amdgcn_alignbyte(ap[1], w0, sh);
    }
    unsigned w = 0;
    for (int k = 0; k < 4; ++k)
        if (a + k < end) w |= (unsigned)a[k] << (8 * k);
    return w;
}

template <typename In> __device__ __forceinline__ float load_px(const In *p, float div);
template <> __device__ __forceinline__ float load_px<uint8_t>(const uint8_t *p, float div) { return (float)(*p) / div; }
template <> __device__ __forceinline__ float load_px<float>(const float *p, float div) { return *p / div; }

// The two passes, shared by the fp32-clip kernel and the stem-record kernel below (one source expression per value: the two kernels give the same bits).
// Pass 1 (vertical): every lane owns interleaved (x, c) elements of the cropped input row span, consecutive lanes = consecutive bytes, and reduces the
// <= ytaps input rows of this output row into LDS. Pass 2 (horizontal): a lane owns an output pixel and reads its taps from LDS.
template <typename In>
__device__ __forceinline__ void aa_vpass(const In *base, long rstride, const int32_t *ye, int span, float div, float *row) {
    const int yn = ye[1];
    const float *wy = (const float *)(ye + 2);
    for (int e = threadIdx.x; e < span; e += 256) {
        float acc = 0.f;
        for (int j = 0; j < yn; j++) acc += wy[j] * load_px<In>(base + j * rstride + e, div);
        row[e] = acc;
    }
}
__device__ __forceinline__ float aa_hpass(const float *row, const int32_t *xe, int C, int c) {
    const int xmin = xe[0], xn = xe[1];
    const float *wx = (const float *)(xe + 2);
    float acc = 0.f;
    for (int j = 0; j < xn; j++) acc += wx[j] * row[(xmin + j) * C + c];
    return acc;
}

// One workgroup per (frame, output row).
template <typename In>
__global__ __launch_bounds__(256) void crop_resize_aa_kernel(const ResizeKP p) {
    extern __shared__ float row[];      // cw * C
    const int t = blockIdx.x / p.oh, oy = blockIdx.x % p.oh;
    const int32_t *ye = p.ytab + (size_t)oy * (2 + p.ytaps);
    const In *base = (const In *)p.in + (((long)t * p.H + p.y0 + ye[0]) * p.W + p.x0) * p.C;
    aa_vpass<In>(base, (long)p.W * p.C, ye, p.cw * p.C, p.div, row);
    __syncthreads();
    for (int ox = threadIdx.x; ox < p.ow; ox += 256) {
        const int32_t *xe = p.xtab + (size_t)ox * (2 + p.xtaps);
        const int oxw = p.flip ? p.ow - 1 - ox : ox;
        for (int c = 0; c < p.C; c++) p.out[t * p.so_t + c * p.so_c + oy * p.so_h + oxw * p.so_w] = aa_hpass(row, xe, p.C, c);
    }
}

// The same resize written as the persistent stem's input records (csrc/conv_stem_pt.hip, tedspad_clip_to_tp's layout): X[n][tp][oh][b][ow/2][24] 16-bit, the
// record of pixel (oy, 2*wq + b) for output-frame pair tp holding value dt*3 + c = clip[n][c][4*tp - pt + dt][oy][2*wq + b], dt = 0..7 (temporal stride 2),
// zero outside the clip. Clip n = source frames first + n*clip_step + f*frame_step, f = 0 .. t_clip - 1 (dali_extraction.py:62-73: sequence_length 16,
// stride 2, step 32), frames past the end of the video are zero frames (pad_sequences). The fp32 clip (9.6 MB at 16 x 224 x 224) is never written.
// A workgroup owns one output row of one QUAD of frames 4*qd - pt .. 4*qd - pt + 3 of one clip: those four frames are the first half (dt 0..3, 24 bytes) of
// pair qd's records AND the second half (dt 4..7) of pair qd - 1's, so every frame is resized once, into one LDS tile of half records (5.3 KB at 224
// columns), which leaves twice as 8-byte pieces; the sibling quad's workgroup (the next blockIdx) fills the other halves of the same lines right behind it.
// One load -> sync -> resize -> sync -> store chain per workgroup, ~24 KB of LDS: six workgroups per CU. (One workgroup per (clip, row) with all four pairs'
// records in a 43 KB tile: two per CU, 9.0 ms per 375 clips; one per (clip, row, pair), every frame resized twice: 3.5 ms; byte loads instead of 4-byte
// loads cost 3.6 ms of an earlier 6.6: scripts/crop_tp_probe.py.)
struct ResizeTpKP {
    ResizeKP r;
    uint16_t *rec;
    int n_clips, first, clip_step, frame_step, t_clip, pt, stt, tp_n;
    const uint8_t *in_end;      // one past the last byte of the frame buffer (the 4-byte loads of the uint8 path stay inside it)
};

template <typename In, typename T>
__global__ __launch_bounds__(256) void crop_resize_tp_kernel(const ResizeTpKP q) {
    constexpr int FG = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_tp[];
    const ResizeKP &p = q.r;
    const int wq_n = p.ow >> 1;
    uint2 *tile = reinterpret_cast<uint2 *>(lds_tp);                 // [plane b][wq][3 x 8 B]: half records (4 frames x 3 channels)
    const int tile8 = 2 * wq_n * 3;
    const int span = p.cw * p.C;
    float *rows = reinterpret_cast<float *>(lds_tp + (size_t)tile8 * 8);        // [FG][span]
    float *lut = rows + FG * span;                                   // [256] (uint8 input, divisor != 255)
    int32_t *xt = reinterpret_cast<int32_t *>(lut + 256);            // this launch's column table [ow][2 + xtaps]: read 4 x 3 times per output pixel below
    const int nq = q.tp_n + 1;
    const int qd = blockIdx.x % nq, rowid = blockIdx.x / nq;
    const int n = rowid / p.oh, oy = rowid % p.oh;
    const int f0 = 4 * qd - q.pt;                                    // first clip frame of the quad
    const bool div255 = p.div == 255.0f;
    for (int i = threadIdx.x; i < tile8; i += 256) tile[i] = make_uint2(0u, 0u);               // frames outside the clip, channels >= C
    if (sizeof(In) == 1 && !div255) lut[threadIdx.x] = (float)threadIdx.x / p.div;
    for (int i = threadIdx.x; i < p.ow * (2 + p.xtaps); i += 256) xt[i] = p.xtab[i];
    const int32_t *ye = p.ytab + (size_t)oy * (2 + p.ytaps);
    const int yn = ye[1];
    const float *wy = (const float *)(ye + 2);
    const long rstride = (long)p.W * p.C;
    uint16_t *t16 = reinterpret_cast<uint16_t *>(lds_tp);
    const In *base[FG];
    bool live[FG];
#pragma unroll
    for (int ff = 0; ff < FG; ++ff) {
        const long fr = (long)q.first + (long)n * q.clip_step + (long)(f0 + ff) * q.frame_step;
        live[ff] = f0 + ff >= 0 && f0 + ff < q.t_clip && fr >= 0 && fr < p.T;    // else a zero frame: its slots stay zero (uniform)
        base[ff] = (const In *)p.in + (((live[ff] ? fr : 0) * p.H + p.y0 + ye[0]) * p.W + p.x0) * p.C;
    }
    if constexpr (sizeof(In) == 1) {
        if (!div255) __syncthreads();                                // the value table (uniform branch)
        // four consecutive bytes per lane and load (byte loads: 64 bytes per wave instruction)
        for (int g = threadIdx.x; 4 * g < span; g += 256) {
            float acc[FG][4];
#pragma unroll
            for (int ff = 0; ff < FG; ++ff)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[ff][k] = 0.f;
            for (int j = 0; j < yn; j++) {
                unsigned w4[FG];
#pragma unroll
                for (int ff = 0; ff < FG; ++ff) w4[ff] = live[ff] ? load_u8x4(reinterpret_cast<const uint8_t *>(base[ff]) + j * rstride + 4 * g, q.in_end) : 0u;
#pragma unroll
                for (int ff = 0; ff < FG; ++ff)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned bt = (w4[ff] >> (8 * k)) & 255u;
                        float px;
                        if (div255) {       // byte / 255.f without the division or a table: b * fl(1/255) corrected by one residual step is the correctly rounded
                                            // quotient for every byte value (checked exhaustively in exact arithmetic; tests compare with the dividing kernel bit for bit)
                            const float xb = (float)bt, r = 1.0f / 255.0f;
                            const float q0 = xb * r;
                            px = __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, xb), r, q0);
                        } else {
                            px = lut[bt];
                        }
                        acc[ff][k] += wy[j] * px;
                    }
            }
#pragma unroll
            for (int ff = 0; ff < FG; ++ff)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * g + k < span) rows[ff * span + 4 * g + k] = acc[ff][k];
        }
    } else {
        for (int e = threadIdx.x; e < span; e += 256) {
            float acc[FG];
#pragma unroll
            for (int ff = 0; ff < FG; ++ff) acc[ff] = 0.f;
            for (int j = 0; j < yn; j++) {
#pragma unroll
                for (int ff = 0; ff < FG; ++ff) acc[ff] += wy[j] * (live[ff] ? load_px<In>(base[ff] + j * rstride + e, p.div) : 0.f);
            }
#pragma unroll
            for (int ff = 0; ff < FG; ++ff) rows[ff * span + e] = acc[ff];
        }
    }
    __syncthreads();                                                 // rows, the zeroed tile and the column table are complete
    for (int ox = threadIdx.x; ox < p.ow; ox += 256) {
        const int32_t *xe = xt + ox * (2 + p.xtaps);
        const int xmin = xe[0], xn = xe[1];
        const float *wx = reinterpret_cast<const float *>(xe + 2);
        const int oxw = p.flip ? p.ow - 1 - ox : ox;
        // aa_hpass for the FG frames x 3 channels at once: a tap's weight is read once; every sum still runs over its taps in ascending order
        float acc[FG][3];
#pragma unroll
        for (int ff = 0; ff < FG; ++ff)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[ff][c] = 0.f;
        for (int j = 0; j < xn; j++) {
            const float wj = wx[j];
            const float *r = rows + (xmin + j) * p.C;
#pragma unroll
            for (int ff = 0; ff < FG; ++ff)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < p.C) acc[ff][c] += wj * r[ff * span + c];
        }
        uint16_t *rec = t16 + ((size_t)((oxw & 1) * wq_n + (oxw >> 1))) * 12;
#pragma unroll
        for (int ff = 0; ff < FG; ++ff) {
            if (!live[ff]) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) rec[ff * 3 + c] = c < p.C ? T::from_f32(acc[ff][c]) : (uint16_t)0;
        }
    }
    __syncthreads();
    // half records -> pair qd (first halves) and pair qd - 1 (second halves); both planes of a row are one contiguous run of records
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int tp = qd - half;
        if (tp < 0 || tp >= q.tp_n) continue;                        // uniform
        uint2 *dst = reinterpret_cast<uint2 *>(q.rec) + ((((long)n * q.tp_n + tp) * p.oh + oy) * 2 * (long)wq_n) * 6 + half * 3;
        for (int i = threadIdx.x; i < tile8; i += 256) {
            const int rc = i / 3, pc = i - rc * 3;
            dst[(long)rc * 6 + pc] = tile[i];
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

extern "C" int32_t tedspad_frames_crop_resize_tp(const void *frames, int32_t in_is_float, int32_t T, int32_t H, int32_t W, int32_t C, int32_t n_clips,
                                                 int32_t first, int32_t clip_step, int32_t frame_step, int32_t t_clip, int32_t y0, int32_t x0, int32_t ch,
                                                 int32_t cw, int32_t oh, int32_t ow, const int32_t *ytab, const int32_t *xtab, float divisor, int32_t flip,
                                                 void *records, int32_t pad_t, int32_t stride_t, int32_t t_pairs, int32_t dtype, void *stream) {
    TS_REQUIRE(frames && records && ytab && xtab, "tedspad_frames_crop_resize_tp: null pointer");
    TS_REQUIRE(T > 0 && H > 0 && W > 0 && C > 0 && C <= 3 && oh > 0 && ow > 0 && ow % 2 == 0, "tedspad_frames_crop_resize_tp: bad sizes (<= 3 channels, even output width)");
    TS_REQUIRE(n_clips > 0 && t_clip > 0 && frame_step > 0 && clip_step >= 0, "tedspad_frames_crop_resize_tp: bad clip sampling");
    TS_REQUIRE(y0 >= 0 && x0 >= 0 && ch > 0 && cw > 0 && y0 + ch <= H && x0 + cw <= W,
               "tedspad_frames_crop_resize_tp: crop box (%d,%d,%d,%d) outside the %dx%d frame", y0, x0, ch, cw, H, W);
    TS_REQUIRE(divisor != 0.f, "tedspad_frames_crop_resize_tp: divisor must be non-zero");
    TS_REQUIRE(stride_t == 2 && pad_t >= 0 && t_pairs > 0, "tedspad_frames_crop_resize_tp: temporal stride 2 (tedspad_clip_to_tp's record layout)");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_frames_crop_resize_tp: bad dtype");
    TS_REQUIRE((uintptr_t)records % 16 == 0 && (long)n_clips * oh * (t_pairs + 1) < (1L << 31), "tedspad_frames_crop_resize_tp: records must be 16-byte aligned; too many rows");
    const size_t lds = (size_t)ow * 24 + (size_t)4 * cw * C * sizeof(float) + 1024 + (size_t)ow * (2 + aa_taps(cw, ow)) * 4;      // a row of half records + four frames' row buffers + the value table + the column table
    TS_REQUIRE(lds <= 160 * 1024, "tedspad_frames_crop_resize_tp: %d columns of records + four %d x %d row buffers exceed the CU's 160 KB of LDS", ow, cw, C);
    ResizeTpKP q;
    ResizeKP &p = q.r;
    p.in = frames; p.out = nullptr; p.ytab = ytab; p.xtab = xtab;
    p.T = T; p.H = H; p.W = W; p.C = C; p.y0 = y0; p.x0 = x0; p.ch = ch; p.cw = cw; p.oh = oh; p.ow = ow;
    p.ytaps = aa_taps(ch, oh); p.xtaps = aa_taps(cw, ow); p.flip = flip; p.div = divisor;
    p.so_t = p.so_c = p.so_h = p.so_w = 0;
    q.rec = (uint16_t *)records; q.n_clips = n_clips; q.first = first; q.clip_step = clip_step; q.frame_step = frame_step; q.t_clip = t_clip;
    q.pt = pad_t; q.stt = stride_t; q.tp_n = t_pairs;
    q.in_end = (const uint8_t *)frames + (size_t)T * H * W * C * (in_is_float ? 4 : 1);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)((long)n_clips * oh * (t_pairs + 1)));
    if (lds > 64 * 1024) {                       // wide source frames (HD): raise the kernel's dynamic LDS limit once per variant
        static thread_local bool raised[4] = {false, false, false, false};
        const int vi = (in_is_float ? 2 : 0) + (dtype == TEDSPAD_F16 ? 0 : 1);
        if (!raised[vi]) {
            const void *fn = vi == 0 ? (const void *)crop_resize_tp_kernel<uint8_t, F16> : vi == 1 ? (const void *)crop_resize_tp_kernel<uint8_t, BF16> :
                             vi == 2 ? (const void *)crop_resize_tp_kernel<float, F16> : (const void *)crop_resize_tp_kernel<float, BF16>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("tedspad_frames_crop_resize_tp: cannot raise the dynamic LDS limit");
                return TEDSPAD_ELAUNCH;
            }
            raised[vi] = true;
        }
    }
    if (in_is_float) {
        if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((crop_resize_tp_kernel<float, F16>), g, dim3(256), lds, s, q);
        else hipLaunchKernelGGL((crop_resize_tp_kernel<float, BF16>), g, dim3(256), lds, s, q);
    } else {
        if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((crop_resize_tp_kernel<uint8_t, F16>), g, dim3(256), lds, s, q);
        else hipLaunchKernelGGL((crop_resize_tp_kernel<uint8_t, BF16>), g, dim3(256), lds, s, q);
    }
    return check_launch("tedspad_frames_crop_resize_tp");
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
