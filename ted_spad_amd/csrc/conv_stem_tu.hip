// Cin = 3 stem with the TEMPORAL taps folded into the channel dimension ("temporal-unfolded" form) for gfx950.
//
// The pixel-pair form of the stems (conv_igemm.hip, conv_stem_halo_kernel) multiplies K = kt*kh*ceil((kw+1)/2)*8 = 1120
// (padded to 1152) for 735 real taps x channels: the fourth channel of every pixel and the eighth tap of every row are
// zeros, so 36 % of the stem's MFMAs (a quarter of a forward) multiply padding. Here the clip is first laid out as
//
//   X'[n][to][h][b][wq][16]   value v = dt*3 + ci  (15 used):  X' = x[n][ci][to*st - pt + dt][h][2*wq + b]
//
// (tedspad_clip_to_tu: one 32-byte position holds the kt*3 values the output frame `to` needs at that pixel; even and odd
// columns in separate planes b), and the stem becomes a 2-D stride-2 convolution over (h, w) with cin = 16:
// K = kh*kw*16 = 784 for the 5x7x7 stem, one k16 MFMA sub-step per tap, 12.25 K steps instead of 18.
//
// Kernel structure = conv_stem_halo_kernel with split-K over 8 waves (tile_cfg 21): a workgroup owns a 1 x 8 x 32 output
// patch x 64 channels, its input halo (21 rows x 2 planes x 35 positions x 32 B = 46 KB) is DMA'd into LDS once, every MFMA
// pixel fragment is read straight from the halo: output pixel wo, tap dw reads column 2*wo + dw - pw = 2*(wo + a) + b, i.e.
// position (wo + a) of plane b -- consecutive lanes read consecutive 32-byte positions (the parity planes turn the stride-2
// walk into a contiguous one). The two 16-byte halves of position p are stored swapped when (p >> 3) & 1 (applied to the DMA
// source address and to the read address): every 16-lane group of a ds_read_b128 then covers all 64 banks. Only the
// [64 co][64 k] weight tile streams (2-slot ring, one DMA instruction per thread and K step).
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16t;

struct StemTU {
    const uint16_t *x;      // X'[n][To][Hi][2][Wq][16]
    const uint16_t *w;      // [>= 64][Kpad], k = (dh*kw + dw)*16 + v
    const float *scale, *shift;
    uint16_t *y;
    int N, To, Hi, Wq, Ho, Wo, ldy, Cout, relu;
    int kh, kw, ph, pw, ntaps, nk, Kpad;
    int amin, PP, HH;       // a_min = floor(-pw / 2); positions per plane row; halo rows
    int tiles_h, tiles_w;
};

constexpr int TU_TH = 8, TU_TW = 32;
constexpr int TU_WSTAGE = 64 * BK * 2;

template <typename T>
__global__ __launch_bounds__(512) void conv_stem_tu_kernel(const StemTU p) {
    constexpr int NT = 512, WS = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = b % p.tiles_w; b /= p.tiles_w;
    const int th = b % p.tiles_h; b /= p.tiles_h;
    const int to = b % p.To;
    const int n = b / p.To;
    const int ho0 = th * TU_TH, wo0 = tw * TU_TW;
    const int P = p.HH * 2 * p.PP;                      // halo positions (32 bytes each)
    const int S = 2 * P;                                // 16-byte slots
    const int Sr = (S + 63) / 64 * 64;
    const int halo_bytes = Sr * 16;
    unsigned char *wring = dsm + halo_bytes;            // [WS][64][64] 16-bit
    int2 *tapd = reinterpret_cast<int2 *>(wring + WS * TU_WSTAGE);   // per tap: {byte delta inside the halo, a - a_min}
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16t);

    for (int i = tid; i < p.nk * 4; i += NT) {
        int2 e = make_int2(0, 0);                        // K padding: zero weights, any in-range address
        if (i < p.ntaps) {
            const int dw = i % p.kw, dh = i / p.kw;
            const int bb = (dw - p.pw) & 1;
            const int a = (dw - p.pw - bb) / 2;          // exact: dw - pw - bb is even
            e.x = ((dh * 2 + bb) * p.PP + (a - p.amin)) * 32;
            e.y = a - p.amin;
        }
        tapd[i] = e;
    }
    __syncthreads();  // table complete (plain LDS stores) before any DMA is counted
    // ---- halo: slot s -> (row, plane, position, half); the stored half is swapped when (position >> 3) & 1 -------------
    const int ih0 = ho0 * 2 - p.ph, wq0 = wo0 + p.amin;
    const size_t img = ((size_t)n * p.To + to) * p.Hi;
    const int NH = (Sr + NT - 1) / NT;
    for (int i = 0; i < NH; ++i) {
        if (i * NT + wave * 64 >= Sr) break;             // wave-uniform
        const int s = i * NT + tid;
        const int hs = s & 1, pos = s >> 1;
        const int pp = pos % p.PP; const int r = pos / p.PP;
        const int bb = r & 1, row = r >> 1;
        const int ih = ih0 + row, wq = wq0 + pp;
        const bool ok = s < S && (unsigned)ih < (unsigned)p.Hi && (unsigned)wq < (unsigned)p.Wq;
        const int half = hs ^ ((pp >> 3) & 1);
        const uint16_t *src = ok ? p.x + (((img + ih) * 2 + bb) * p.Wq + wq) * 16 + half * 8 : zero;
        lds_dma16(src, lds0 + (i * NT + wave * 64) * 16);
    }
    // ---- weights: [64][64] tile per K step, swizzled on the source like the generic kernel --------------------------------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w + (size_t)rsub * p.Kpad + kc * 8;
    auto issue_w = [&](int kt, int slot) { lds_dma16(wsrc + kt * BK, lds0 + halo_bytes + slot * TU_WSTAGE + wave * 8 * (BK * 2)); };
    issue_w(0, 0);

    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    const int wq4 = wave & 3, kh2 = wave >> 2;       // output rows 2*wq4, 2*wq4+1; which two taps of every K step
    int pixb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) pixb[g] = ((2 * (2 * wq4 + g)) * 2 * p.PP + l31) * 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][g][r] = 0.f;

    for (int kt = 0; kt < p.nk; ++kt) {
        int2 dk[2];
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) dk[kq] = tapd[kt * 4 + kh2 * 2 + kq];
        wait_vmcnt<0>();                // stage kt (issued one step ago) and, on kt = 0, the halo
        __builtin_amdgcn_s_barrier();   // weight stage kt (+ halo on kt = 0) visible; the slot of stage kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + 1 < p.nk) issue_w(kt + 1, (kt + 1) & 1);
        const uint16_t *W = reinterpret_cast<const uint16_t *>(wring + (kt & 1) * TU_WSTAGE) + l31 * BK;
        const int tap0 = kt * 4 + kh2 * 2;
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            if (tap0 + kq >= p.ntaps) break;             // wave-uniform: K padding of the last step
            const int ks = kh2 * 2 + kq;
            const int hsel = ((lh ^ ((l31 + dk[kq].y) >> 3)) & 1) << 4;
            const int coff = (((ks << 1) | lh) ^ swz) << 3;
            uint4 fa[2], fw[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) fa[g] = *reinterpret_cast<const uint4 *>(dsm + pixb[g] + dk[kq].x + hsel);
#pragma unroll
            for (int a = 0; a < 2; ++a) fw[a] = *reinterpret_cast<const uint4 *>(W + a * 32 * BK + coff);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[a][g] = T::mfma(fw[a], fa[g], acc[a][g]);
        }
    }
    __syncthreads();

    // ---- epilogue: partial sums of the two wave groups meet in the fp32 staging tile [256 px][64 co] -----------------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
    if (kh2 == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ml = (2 * wq4 + g) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[a][g][4 * q], acc[a][g][4 * q + 1], acc[a][g][4 * q + 2], acc[a][g][4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + a * 32 + 8 * q + 4 * lh) = v;
                }
            }
    }
    __syncthreads();
    if (kh2 == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ml = (2 * wq4 + g) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 *ptr = reinterpret_cast<f32x4 *>(stg + ml * STG_LD + a * 32 + 8 * q + 4 * lh);
                    const f32x4 o = *ptr;
                    f32x4 v = {acc[a][g][4 * q] + o[0], acc[a][g][4 * q + 1] + o[1], acc[a][g][4 * q + 2] + o[2], acc[a][g][4 * q + 3] + o[3]};
                    *ptr = v;
                }
            }
    }
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;     // 8 chunks of 8 channels per pixel, 64 pixels per pass
    const int nch = cc * 8;
    if (nch >= p.Cout) return;
    float sc[8], sf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = r0 + it * 64;
        const int ho = ho0 + (r >> 5), wo = wo0 + (r & 31);
        if (ho >= p.Ho || wo >= p.Wo) continue;
        const size_t m = (((size_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8<T>(v);
    }
}

// fp32 NCTHW clip -> X'[n][to][h][b][wq][16]: a thread converts 8 consecutive columns of one row of one OUTPUT frame:
// kt x c x two 16-byte loads, then 4 positions x 32 bytes per parity plane (two contiguous 128-byte runs). Every input frame
// is read by kt / stride_t output frames (2.5x for the 5-tap stride-2 stem): measured 516 us per 75 clips against 266 us for
// the pixel-pair layout; walking the output frames inside one thread (re-reads served by L2) was slower (637 us: too few
// threads in flight).
template <typename T>
__global__ __launch_bounds__(256) void clip_to_tu_kernel(const float *x, uint16_t *y, int c, int t, int h, int w8, long sn, long sc, long st, long sh,
                                                         int kt, int stt, int pt, int to_n, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int iw8 = (int)(r % w8); r /= w8;
        const int ih = (int)(r % h); r /= h;
        const int to = (int)(r % to_n);
        const long n = r / to_n;
        float v[8][16];
#pragma unroll
        for (int px = 0; px < 8; ++px)
#pragma unroll
            for (int e = 0; e < 16; ++e) v[px][e] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            const int it = to * stt - pt + dt;
            if (dt < kt && (unsigned)it < (unsigned)t) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    if (ch < c) {
                        const float *px = x + n * sn + ch * sc + it * st + ih * sh + iw8 * 8;
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(px), bq = *reinterpret_cast<const f32x4 *>(px + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q][dt * 3 + ch] = a[q]; v[q + 4][dt * 3 + ch] = bq[q]; }
                    }
                }
            }
        }
        const int wq_n = w8 * 4;
        uint16_t *row = y + (((n * to_n + to) * h + ih) * 2) * (long)wq_n * 16;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint16_t *dst = row + ((long)bb * wq_n + iw8 * 4 + q) * 16;
                float lo[8], hi[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { lo[e] = v[2 * q + bb][e]; hi[e] = v[2 * q + bb][8 + e]; }
                *reinterpret_cast<uint4 *>(dst) = pack8<T>(lo);
                *reinterpret_cast<uint4 *>(dst + 8) = pack8<T>(hi);
            }
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_clip_to_tu(const float *x, void *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc,
                                      int64_t st, int64_t sh, int64_t sw, int32_t kt, int32_t stride_t, int32_t pad_t, int32_t to, int32_t dtype,
                                      void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && c <= 3 && t > 0 && h > 0 && w > 0 && kt > 0 && kt * 3 <= 16 && kt <= 5 && stride_t > 0 && pad_t >= 0 && to > 0,
               "tedspad_clip_to_tu: needs c <= 3 and kt <= 5 (kt*3 values per position fit 16)");
    TS_REQUIRE(sw == 1 && w % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && sn % 4 == 0 && sc % 4 == 0 && st % 4 == 0 && sh % 4 == 0,
               "tedspad_clip_to_tu: rows must be contiguous, W a multiple of 8, 16-byte aligned");
    TS_REQUIRE((to - 1) * stride_t - pad_t < t, "tedspad_clip_to_tu: output frames reach past the clip");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_clip_to_tu: bad dtype");
    const long total = (long)n * to * h * (w / 8);
    const long blocks = (total + 255) / 256;
    const dim3 g((unsigned)(blocks < 65536L * 16 ? blocks : 65536L * 16));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(clip_to_tu_kernel<F16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st, (long)sh, kt, stride_t, pad_t, to, total);
    else hipLaunchKernelGGL(clip_to_tu_kernel<BF16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st, (long)sh, kt, stride_t, pad_t, to, total);
    return check_launch("tedspad_clip_to_tu");
}

extern "C" int32_t tedspad_stem_tu_kpad(int32_t kh, int32_t kw) {
    if (kh <= 0 || kw <= 0 || kh > 7 || kw > 7) return TEDSPAD_EINVAL;
    return (kh * kw * 16 + BK - 1) / BK * BK;
}

extern "C" int32_t tedspad_stem_tu_fwd(const void *x_tu, const void *w_packed, const float *scale, const float *shift, void *y, int32_t n, int32_t to,
                                       int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t kh, int32_t kw, int32_t ph, int32_t pw, int32_t cout,
                                       int32_t ldy, int32_t relu, int32_t dtype, void *stream) {
    TS_REQUIRE(x_tu && w_packed && scale && shift && y && n > 0 && to > 0 && h > 0 && w > 0 && w % 2 == 0 && ho > 0 && wo > 0, "tedspad_stem_tu_fwd: bad arguments");
    TS_REQUIRE(kh > 0 && kw > 0 && kh <= 7 && kw <= 7 && ph >= 0 && pw >= 0 && ph < kh && pw < kw && cout > 0 && cout <= 64 && cout % 8 == 0 && ldy % 8 == 0 && ldy >= cout,
               "tedspad_stem_tu_fwd: kernel <= 7x7, cout <= 64 (multiple of 8)");
    TS_REQUIRE((ho - 1) * 2 - ph < h && (wo - 1) * 2 - pw < w, "tedspad_stem_tu_fwd: output extent reaches past the input");
    TS_REQUIRE(((uintptr_t)x_tu | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)shift) % 16 == 0, "tedspad_stem_tu_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_stem_tu_fwd: bad dtype");
    TS_REQUIRE((long)n * to * h * w * 16 < (1L << 40), "tedspad_stem_tu_fwd: tensor too large");
    StemTU p;
    p.x = (const uint16_t *)x_tu; p.w = (const uint16_t *)w_packed; p.scale = scale; p.shift = shift; p.y = (uint16_t *)y;
    p.N = n; p.To = to; p.Hi = h; p.Wq = w / 2; p.Ho = ho; p.Wo = wo; p.ldy = ldy; p.Cout = cout; p.relu = relu;
    p.kh = kh; p.kw = kw; p.ph = ph; p.pw = pw; p.ntaps = kh * kw; p.Kpad = tedspad_stem_tu_kpad(kh, kw); p.nk = p.Kpad / BK;
    // column 2*wo + dw - pw = 2*(wo + a) + b: a ranges over [floor(-pw / 2), floor((kw - 1 - pw) / 2)]
    const int amin = -((pw + 1) / 2), amax = (kw - 1 - pw) >= 0 ? (kw - 1 - pw) / 2 : -((pw - kw + 2) / 2);
    p.amin = amin; p.PP = TU_TW + (amax - amin); p.HH = (TU_TH - 1) * 2 + kh;
    p.tiles_h = (ho + TU_TH - 1) / TU_TH; p.tiles_w = (wo + TU_TW - 1) / TU_TW;
    const int S = 2 * p.HH * 2 * p.PP;
    const int main_bytes = (S + 63) / 64 * 64 * 16 + 2 * TU_WSTAGE + p.nk * 4 * 8;
    const int stage_bytes = 256 * (64 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    TS_REQUIRE(lds <= 160 * 1024, "tedspad_stem_tu_fwd: halo does not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    static thread_local int attr_set[2] = {0, 0};
    const int di = dtype == TEDSPAD_F16 ? 0 : 1;
    if (!attr_set[di]) {
        const void *fn = di == 0 ? (const void *)conv_stem_tu_kernel<F16> : (const void *)conv_stem_tu_kernel<BF16>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_stem_tu_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[di] = 1;
    }
    const dim3 grid((unsigned)((long)n * to * p.tiles_h * p.tiles_w));
    if (di == 0) hipLaunchKernelGGL(conv_stem_tu_kernel<F16>, grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL(conv_stem_tu_kernel<BF16>, grid, dim3(512), lds, s, p);
    return check_launch("tedspad_stem_tu_fwd");
}
