// HBM-bound helpers of the I3D / UNet forward path for gfx950: max-pool, global average
// pool and the layout conversions at the module boundary.  All of them move 16 bytes per
// lane (8 channels of one channels-last pixel) so every wave instruction is a run of
// full 128-byte lines; grids are sized >> 256 workgroups.
#include <stdarg.h>

#include "common.h"

namespace tedspad {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

struct PoolKP {
    const uint16_t *x;
    uint16_t *y;
    unsigned char *idx;   // optional: window-local index of the FIRST maximum per output element (training)
    int Ti, Hi, Wi, C8, ldx, ldy;
    int To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int pad_zero;
    long total;  // n*to*ho*wo*C8
};

// 8-lane 16-bit max without leaving the packed form where the ISA has it (v_pk_max_f16); bf16 goes through fp32.
template <typename T> __device__ __forceinline__ uint4 max8(uint4 a, uint4 b);
template <> __device__ __forceinline__ uint4 max8<F16>(uint4 a, uint4 b) {
    return __builtin_bit_cast(uint4, __builtin_elementwise_max(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b)));
}
template <> __device__ __forceinline__ uint4 max8<BF16>(uint4 a, uint4 b) {
    float fa[8], fb[8];
    unpack8<BF16>(a, fa);
    unpack8<BF16>(b, fb);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = __builtin_fmaxf(fa[i], fb[i]);
    return pack8<BF16>(fa);
}

// max over the window, 8 channels per thread. Padded taps contribute 0 when pad_zero
// (MaxPool3dSamePadding pads with zeros BEFORE pooling, i3d.py:41-45), else are skipped
// (nn.MaxPool3d semantics, large_i3d.py:138-139).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const PoolKP p) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % p.C8);
        long r = idx / p.C8;
        const int wo = (int)(r % p.Wo); r /= p.Wo;
        const int ho = (int)(r % p.Ho); r /= p.Ho;
        const int to = (int)(r % p.To);
        const int n = (int)(r / p.To);
        float m[8];
        int am[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { m[i] = -3.0e38f; am[i] = 0; }
        bool padded = false;
        for (int dt = 0; dt < p.kt; ++dt) {
            const int it = to * p.st - p.pt + dt;
            for (int dh = 0; dh < p.kh; ++dh) {
                const int ih = ho * p.sh - p.ph + dh;
                for (int dw = 0; dw < p.kw; ++dw) {
                    const int iw = wo * p.sw - p.pw + dw;
                    if ((unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) {
                        const size_t off = ((((size_t)n * p.Ti + it) * p.Hi + ih) * p.Wi + iw) * p.ldx + c8 * 8;
                        float v[8];
                        unpack8<T>(*reinterpret_cast<const uint4 *>(p.x + off), v);
                        const int li = (dt * p.kh + dh) * p.kw + dw;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (v[i] > m[i]) { m[i] = v[i]; am[i] = li; }   // strict: the first maximum wins (torch semantics)
                        }
                    } else {
                        padded = true;
                    }
                }
            }
        }
        if (padded && p.pad_zero) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (m[i] < 0.f) { m[i] = 0.f; am[i] = 255; }   // the zero padding wins: no input element gets the gradient
            }
        }
        const size_t opix = (((size_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
        *reinterpret_cast<uint4 *>(p.y + opix * p.ldy + c8 * 8) = pack8<T>(m);
        if (p.idx) {
            uint2 pk;
            pk.x = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
            pk.y = (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24);
            *reinterpret_cast<uint2 *>(p.idx + (opix * p.C8 + c8) * 8) = pk;
        }
    }
}

// Forward-only variant (no arg-max record): the maximum stays in packed 16-bit form (v_pk_max_f16), ~6x fewer vector
// instructions per tap than the fp32 compare-and-select above -- the inference pools are then bound by HBM, not by the VALU.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_noidx_kernel(const PoolKP p, const uint32_t ninf) {
    const uint4 lo = make_uint4(ninf, ninf, ninf, ninf), zero = make_uint4(0, 0, 0, 0);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % p.C8);
        long r = idx / p.C8;
        const int wo = (int)(r % p.Wo); r /= p.Wo;
        const int ho = (int)(r % p.Ho); r /= p.Ho;
        const int to = (int)(r % p.To);
        const int n = (int)(r / p.To);
        uint4 m = lo;
        bool padded = false;
        for (int dt = 0; dt < p.kt; ++dt) {
            const int it = to * p.st - p.pt + dt;
            for (int dh = 0; dh < p.kh; ++dh) {
                const int ih = ho * p.sh - p.ph + dh;
                const bool rowok = (unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi;
                const uint16_t *row = p.x + ((((size_t)n * p.Ti + it) * p.Hi + ih) * p.Wi) * p.ldx + c8 * 8;
                for (int dw = 0; dw < p.kw; ++dw) {
                    const int iw = wo * p.sw - p.pw + dw;
                    if (rowok && (unsigned)iw < (unsigned)p.Wi) m = max8<T>(m, *reinterpret_cast<const uint4 *>(row + (size_t)iw * p.ldx));
                    else padded = true;
                }
            }
        }
        if (padded && p.pad_zero) m = max8<T>(m, zero);
        const size_t opix = (((size_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
        *reinterpret_cast<uint4 *>(p.y + opix * p.ldy + c8 * 8) = m;
    }
}

// 3x3x3 / stride 1 / pad 1 max-pool (the pool branch of every InceptionModule, i3d.py:133-134,148: 9 of the 13 pools of
// InceptionI3d). The generic kernel above reads 27 taps per output; here one thread walks a W row for TWO output rows
// (h0, h0+1) of one 8-channel group: per column it loads the 3(t) x 4(h) inputs once, reduces them to the two column
// maxima, and an output is the max of three consecutive column maxima kept in registers: 6 loads per output instead of
// 27, all of them independent 16-byte loads that are contiguous across the lanes (channel-minor).
// `padv` is what an out-of-range tap contributes: 0 (MaxPool3dSamePadding pads with zeros) or -inf (nn.MaxPool3d).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_k3s1_kernel(const uint16_t *x, uint16_t *y, int N, int Tn, int H, int W, int C8, int ldx, int ldy,
                                                           uint32_t padw, int segs, int seglen, long total) {
    const uint4 padv = make_uint4(padw, padw, padw, padw);
    const int H2 = (H + 1) / 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        long r = idx / C8;
        const int seg = (int)(r % segs); r /= segs;          // a row is cut into `segs` runs of `seglen` columns (+1 halo column each side)
        const int h0 = (int)(r % H2) * 2; r /= H2;
        const int t = (int)(r % Tn);
        const int n = (int)(r / Tn);
        const int w0 = seg * seglen, w1 = min(W, w0 + seglen);
        if (w0 >= W) continue;
        const uint16_t *row[12];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int dh = 0; dh < 4; ++dh) {
                const int it = t + dt - 1, ih = h0 + dh - 1;
                row[dh * 3 + dt] = ((unsigned)it < (unsigned)Tn && (unsigned)ih < (unsigned)H)
                                       ? x + (((size_t)n * Tn + it) * H + ih) * (size_t)W * ldx + c8 * 8 : nullptr;
            }
        uint16_t *o0 = y + (((size_t)n * Tn + t) * H + h0) * (size_t)W * ldy + c8 * 8;
        uint16_t *o1 = o0 + (size_t)W * ldy;
        const bool two = h0 + 1 < H;
        uint4 cur[12], nxt[12];
        auto load_col = [&](int w, uint4 (&v)[12]) {
            const bool in = (unsigned)w < (unsigned)W;
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = (in && row[i]) ? *reinterpret_cast<const uint4 *>(row[i] + (size_t)w * ldx) : padv;
        };
        load_col(w0 - 1, cur);
        uint4 a1 = padv, a2 = padv, b1 = padv, b2 = padv;        // column maxima at w-1 and w-2 (rows h0 / h0+1)
        for (int w = w0 - 1; w <= w1; ++w) {
            if (w < w1) load_col(w + 1, nxt);                     // in flight while column w is reduced
            uint4 rm[4];
#pragma unroll
            for (int dh = 0; dh < 4; ++dh) rm[dh] = max8<T>(max8<T>(cur[dh * 3], cur[dh * 3 + 1]), cur[dh * 3 + 2]);
            const uint4 mid = max8<T>(rm[1], rm[2]);
            const uint4 a0 = max8<T>(rm[0], mid), b0 = max8<T>(mid, rm[3]);
            if (w >= w0 + 1) {
                *reinterpret_cast<uint4 *>(o0 + (size_t)(w - 1) * ldy) = max8<T>(max8<T>(a2, a1), a0);
                if (two) *reinterpret_cast<uint4 *>(o1 + (size_t)(w - 1) * ldy) = max8<T>(max8<T>(b2, b1), b0);
            }
            a2 = a1; a1 = a0; b2 = b1; b1 = b0;
#pragma unroll
            for (int i = 0; i < 12; ++i) cur[i] = nxt[i];
        }
    }
}

// The same pool with every input element fetched ONCE: the column-walking kernel above loads each frame for the threads of three output frames and each row for two row
// pairs -- 6 loads per output through L1 / L2 (2.7 TB/s of its bytes on the Inception module inputs). Here a workgroup owns a band of HB rows x W columns x CG
// 8-channel groups of ONE clip and walks its frames: frame t + 1 is fetched into registers (coalesced: CG x 16 contiguous bytes per pixel) while frame t's spatial
// 3 x 3 maxima are read from LDS (two frame buffers with a border of `padv`; one barrier per frame), and an output frame is the max of three consecutive spatial maxima
// kept in registers. HBM sees (HB + 2) / HB reads and one write per element. Maxima are exact: bit-identical to the kernels above.
constexpr int K3_NL = 10, K3_NO = 8;                     // per thread: input rows of a frame tile (HB + 2), output rows (HB)
struct K3S1Geo {
    int HB, CG, bands, cgroups;
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_k3s1_lds_kernel(const uint16_t *x, uint16_t *y, int Tn, int H, int W, int C8, int ldx, int ldy, uint32_t padw,
                                                               const K3S1Geo g) {
    extern __shared__ __attribute__((aligned(16))) uint4 k3buf[];
    const uint4 padv = make_uint4(padw, padw, padw, padw);
    int b = blockIdx.x;
    const int cgi = b % g.cgroups; b /= g.cgroups;
    const int band = b % g.bands;
    const int n = b / g.bands;
    const int c80 = cgi * g.CG, ncg = min(g.CG, C8 - c80);
    const int h0 = band * g.HB, nh = min(g.HB, H - h0);
    const int RW = W + 2;                                  // cells per tile row (a border column each side)
    const int cells = (g.HB + 2) * RW * g.CG;              // per frame buffer
    for (int i = threadIdx.x; i < 2 * cells; i += 256) k3buf[i] = padv;       // borders (and the rows outside the frame) stay `padv` for good
    // a thread owns one (column, channel group) and every rp-th row: its k-th piece is an affine function of k (no per-piece tables)
    const int tpr = W * ncg, rp = 256 / tpr;               // threads per row, rows per pass (the launcher made W * CG <= 256)
    const int lane_r = threadIdx.x / tpr, lc = threadIdx.x - lane_r * tpr;
    const int c = lc / ncg, q = lc - c * ncg;
    const bool live = lane_r < rp;
    const int r_lo = h0 == 0 ? 1 : 0, r_hi = min(nh + 1, H - h0);             // tile rows r (frame row h0 - 1 + r) that lie inside the frame
    const int src0 = ((h0 - 1 + lane_r) * W + c) * ldx + (c80 + q) * 8, srcs = rp * W * ldx;
    const int cel0 = (lane_r * RW + c + 1) * g.CG + q, cels = rp * RW * g.CG;
    const int dst0 = ((h0 + lane_r) * W + c) * ldy + (c80 + q) * 8, dsts = rp * W * ldy;
    const size_t fsx = (size_t)H * W * ldx, fsy = (size_t)H * W * ldy;
    const uint16_t *xn = x + (size_t)n * Tn * fsx;
    uint16_t *yn = y + (size_t)n * Tn * fsy;
    uint4 in[K3_NL];
#pragma unroll
    for (int k = 0; k < K3_NL; ++k) in[k] = padv;
    bool rowin[K3_NL];
#pragma unroll
    for (int k = 0; k < K3_NL; ++k) { const int r = lane_r + k * rp; rowin[k] = live && r >= r_lo && r <= r_hi; }
#define K3_FETCH(TT)                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < K3_NL; ++k)                                                                   \
        if (rowin[k]) in[k] = *reinterpret_cast<const uint4 *>(xn + (size_t)(TT) * fsx + src0 + k * srcs)
#define K3_PUT(FF)                                                                                                     \
    _Pragma("unroll") for (int k = 0; k < K3_NL; ++k)                                                                   \
        if (rowin[k]) (FF)[cel0 + k * cels] = in[k]
    K3_FETCH(0);
    __syncthreads();                                       // the fill is done
    K3_PUT(k3buf);
    __syncthreads();
    uint4 s1[K3_NO], s2[K3_NO];                            // spatial maxima of frames t - 1 and t - 2
#pragma unroll
    for (int k = 0; k < K3_NO; ++k) { s1[k] = padv; s2[k] = padv; }
    const int dR = RW * g.CG, dC = g.CG;
    for (int t = 0; t <= Tn; ++t) {
        if (t + 1 < Tn) { K3_FETCH(t + 1); }
        const uint4 *F = k3buf + (t & 1) * cells + cel0 + dR;          // the centre cell of output row lane_r
#pragma unroll
        for (int k = 0; k < K3_NO; ++k) {
            const int r = lane_r + k * rp;
            if (live && r < nh) {
                uint4 s0 = padv;                           // frame Tn: padding
                if (t < Tn) {
                    const uint4 *cp = F + k * cels;
                    const uint4 r0 = max8<T>(max8<T>(cp[-dR - dC], cp[-dR]), cp[-dR + dC]);
                    const uint4 r1 = max8<T>(max8<T>(cp[-dC], cp[0]), cp[dC]);
                    const uint4 r2 = max8<T>(max8<T>(cp[dR - dC], cp[dR]), cp[dR + dC]);
                    s0 = max8<T>(max8<T>(r0, r1), r2);
                }
                if (t >= 1) *reinterpret_cast<uint4 *>(yn + (size_t)(t - 1) * fsy + dst0 + k * dsts) = max8<T>(max8<T>(s2[k], s1[k]), s0);
                s2[k] = s1[k]; s1[k] = s0;
            }
        }
        if (t + 1 < Tn) { uint4 *Fn = k3buf + ((t + 1) & 1) * cells; K3_PUT(Fn); }       // that buffer was last read in iteration t - 1: everybody passed the barrier that ended it
        __syncthreads();
    }
#undef K3_FETCH
#undef K3_PUT
}

// mean over `spatial` pixels, fp32 accumulate + output. A workgroup owns 64 8-channel chunks of one sample; its four waves take every fourth
// pixel (four independent load streams per chunk: the one-thread-per-chunk form was latency-bound, 47 us for 90 MB at the bench size) and
// are summed through LDS.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const uint16_t *x, float *y, int n, int spatial, int c8n, int ldx) {
    __shared__ float red[3][64][8];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int cgroups = (c8n + 63) / 64;
    const int b = blockIdx.x / cgroups, c8 = (blockIdx.x % cgroups) * 64 + cl;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < c8n) {
        const uint16_t *px = x + (size_t)b * spatial * ldx + c8 * 8;
        for (int i = pl; i < spatial; i += 4) {
            float v[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(px + (size_t)i * ldx), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
    }
    if (pl) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pl - 1][cl][j] = s[j];
    }
    __syncthreads();
    if (pl == 0 && c8 < c8n) {
        const float inv = 1.f / (float)spatial;
        float *py = y + (size_t)b * c8n * 8 + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) py[j] = (s[j] + red[0][cl][j] + red[1][cl][j] + red[2][cl][j]) * inv;
    }
}

// nn.AvgPool3d(kernel (kt,kh,kw), stride 1, no padding) of InceptionI3d.extract_features on maps larger than the kernel
// (aux_code/models/i3d.py:293-295,336-340): x (n,t,h,w,c) 16-bit channels-last -> y fp32 (n,c,to,ho,wo) contiguous (the NCTHW tensor
// the reference returns). One thread per (output position, 8-channel chunk), fp32 accumulate.
template <typename T>
__global__ __launch_bounds__(256) void avgpool3d_s1_kernel(const uint16_t *x, float *y, int t, int h, int w, int c8n, int ldx, int kt, int kh, int kw,
                                                           int to, int ho, int wo, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int c8 = (int)(r % c8n); r /= c8n;
        const int ow = (int)(r % wo); r /= wo;
        const int oh = (int)(r % ho); r /= ho;
        const int ot = (int)(r % to);
        const long n = r / to;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < kt; ++a)
            for (int b = 0; b < kh; ++b)
                for (int d = 0; d < kw; ++d) {
                    float v[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(x + ((((n * t + ot + a) * h + oh + b) * w + ow + d) * (long)ldx + c8 * 8)), v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += v[j];
                }
        const float inv = 1.f / (float)(kt * kh * kw);
        const long plane = (long)to * ho * wo;
        float *py = y + ((n * c8n * 8 + c8 * 8) * plane + ((long)ot * ho + oh) * wo + ow);
#pragma unroll
        for (int j = 0; j < 8; ++j) py[j * plane] = s[j] * inv;
    }
}

// fp32 (n,c,t,h,w) with arbitrary element strides -> 16-bit (n,t,h,w,cpad).
// One thread per 8 output channels-last elements: cpad=4 -> two pixels, cpad=8 -> one.
template <typename T>
__global__ __launch_bounds__(256) void to_channels_last_kernel(const float *x, uint16_t *y, int c, int t, int h, int w, long sn,
                                                                long sc, long st, long sh, long sw, int cpad, long total8) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total8; idx += (long)gridDim.x * 256) {
        const int ppt = 8 / cpad;  // pixels per thread
        long pix = idx * ppt;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        for (int j = 0; j < ppt; ++j) {
            long r = pix + j;
            const int iw = (int)(r % w); r /= w;
            const int ih = (int)(r % h); r /= h;
            const int it = (int)(r % t);
            const long n = r / t;
            const float *px = x + n * sn + it * st + ih * sh + iw * sw;
            for (int ch = 0; ch < c; ++ch) v[j * cpad + ch] = px[ch * sc];
        }
        *reinterpret_cast<uint4 *>(y + idx * 8) = pack8<T>(v);
    }
}

// Fast path of the above for W-contiguous clips (sw == 1, w % 8 == 0, 16-byte aligned rows): a thread converts
// 8 consecutive pixels of one row: two 16-byte loads per channel plane, 64 (cpad 4) or 128 (cpad 8) bytes stored.
template <typename T, int CPAD>
__global__ __launch_bounds__(256) void to_channels_last_w8_kernel(const float *x, uint16_t *y, int c, int t, int h, int w8, long sn, long sc,
                                                                   long st, long sh, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int iw8 = (int)(r % w8); r /= w8;
        const int ih = (int)(r % h); r /= h;
        const int it = (int)(r % t);
        const long n = r / t;
        const float *px = x + n * sn + it * st + ih * sh + iw8 * 8;
        float v[8][CPAD];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int ch = 0; ch < CPAD; ++ch) v[p][ch] = 0.f;
        for (int ch = 0; ch < c; ++ch) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(px + ch * sc), b = *reinterpret_cast<const f32x4 *>(px + ch * sc + 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) { v[p][ch] = a[p]; v[p + 4][ch] = b[p]; }
        }
        uint16_t *py = y + idx * 8 * CPAD;
#pragma unroll
        for (int q = 0; q < CPAD; ++q) {   // CPAD chunks of 8 output elements
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[(q * 8 + e) / CPAD][(q * 8 + e) % CPAD];
            *reinterpret_cast<uint4 *>(py + q * 8) = pack8<T>(o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void to_nchw_kernel(const uint16_t *x, float *y, int c, long thw, int ldx, long total) {
    // idx over (n, c, thw): writes coalesced along thw, reads strided (tiny tensors only)
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx % thw;
        const long r = idx / thw;
        const int ch = (int)(r % c);
        const long n = r / c;
        y[idx] = T::to_f32(x[(n * thw + p) * ldx + ch]);
    }
}

// Bilinear x2 upsample, align_corners=True (nn.Upsample in unet_parts.py:50), written into a
// channel slice of the (larger, zero-padded) skip-concat buffer: unet_parts.py:56-67.
// fp32 index/lambda arithmetic follows torch's area_pixel_compute_source_index.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint16_t *x, uint16_t *y, int h, int w, int C8, int ldx, int ldy,
                                                          int Ho, int Wo, int py, int px, long total) {
    const int oh_sz = 2 * h, ow_sz = 2 * w;
    const float rh = oh_sz > 1 ? (float)(h - 1) / (float)(oh_sz - 1) : 0.f;
    const float rw = ow_sz > 1 ? (float)(w - 1) / (float)(ow_sz - 1) : 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        long r = idx / C8;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const long n = r / Ho;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
        const int uh = oh - py, uw = ow - px;
        if ((unsigned)uh < (unsigned)oh_sz && (unsigned)uw < (unsigned)ow_sz) {
            const float h1r = rh * uh, w1r = rw * uw;
            const int h1 = (int)h1r, w1 = (int)w1r;
            const int h1p = h1 < h - 1 ? 1 : 0, w1p = w1 < w - 1 ? 1 : 0;
            const float hl1 = h1r - h1, hl0 = 1.f - hl1, wl1 = w1r - w1, wl0 = 1.f - wl1;
            const uint16_t *p00 = x + ((n * h + h1) * w + w1) * (long)ldx + c8 * 8;
            float v00[8], v01[8], v10[8], v11[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(p00), v00);
            unpack8<T>(*reinterpret_cast<const uint4 *>(p00 + (long)w1p * ldx), v01);
            unpack8<T>(*reinterpret_cast<const uint4 *>(p00 + (long)h1p * w * ldx), v10);
            unpack8<T>(*reinterpret_cast<const uint4 *>(p00 + ((long)h1p * w + w1p) * ldx), v11);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = hl0 * (wl0 * v00[i] + wl1 * v01[i]) + hl1 * (wl0 * v10[i] + wl1 * v11[i]);
        }
        *reinterpret_cast<uint4 *>(y + ((n * Ho + oh) * Wo + ow) * (long)ldy + c8 * 8) = pack8<T>(o);
    }
}

inline int grid_for(long work_items) {
    long g = (work_items + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;  // grid-stride the rest
    return g < 1 ? 1 : (int)g;
}

// F.interpolate(scale_factor=2, mode="nearest") of smp's DecoderBlock (segmentation_models_pytorch 0.3.3, decoders/unetplusplus/decoder.py:
// DecoderBlock.forward), written into its channel slice of the block's concat buffer; also the plain channel-slice copy that places a
// tensor the UNet++ dense skip pathway concatenates twice. 16 bytes (8 channels) per thread, dtype-agnostic.
__global__ __launch_bounds__(256) void upsample_nearest2x_kernel(const uint4 *x, uint4 *y, int h, int w, int C8, int ldx8, int ldy8, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int c8 = (int)(r % C8); r /= C8;
        const int wo = (int)(r % (2 * w)); r /= 2 * w;
        const int ho = (int)(r % (2 * h));
        const long n = r / (2 * h);
        y[((n * 2 * h + ho) * 2 * w + wo) * ldy8 + c8] = x[((n * h + (ho >> 1)) * w + (wo >> 1)) * ldx8 + c8];
    }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const uint4 *x, uint4 *y, int C8, int ldx8, int ldy8, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long px = idx / C8;
        const int c8 = (int)(idx - px * C8);
        y[px * ldy8 + c8] = x[px * ldx8 + c8];
    }
}

// Backward of the nearest x2 upsample: dx[n][h][w] = (dx +) the sum of the 2 x 2 block of dy it was copied to (fp32 sum, one rounding).
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest2x_bwd_kernel(const uint4 *dy, uint4 *dx, int h, int w, int C8, int ldy8, int ldx8, int acc, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int c8 = (int)(r % C8); r /= C8;
        const int wi = (int)(r % w); r /= w;
        const int hi = (int)(r % h);
        const long n = r / h;
        float s[8], v[8];
        uint4 *dst = dx + ((n * h + hi) * w + wi) * ldx8 + c8;
        if (acc) unpack8<T>(*dst, s);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                unpack8<T>(dy[((n * 2 * h + 2 * hi + a) * 2 * w + 2 * wi + b) * ldy8 + c8], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += v[i];
            }
        *dst = pack8_lim<T>(s, __builtin_inff());      // gradients: never clamped (an overflow must stay visible)
    }
}

// y[:, :c] += x[:, :c] (a gradient that reaches a tensor through a second consumer)
template <typename T>
__global__ __launch_bounds__(256) void add_channels_kernel(const uint4 *x, uint4 *y, int C8, int ldx8, int ldy8, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long px = idx / C8;
        const int c8 = (int)(idx - px * C8);
        float a[8], b[8];
        unpack8<T>(x[px * ldx8 + c8], a);
        unpack8<T>(y[px * ldy8 + c8], b);
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] += a[i];
        y[px * ldy8 + c8] = pack8_lim<T>(b, __builtin_inff());
    }
}

// f16 head-room made observable (the inference stores saturate silently at +-65504, common.h): elements of a channels-last tensor that sit AT the largest finite
// value of their type (f16: 0x7bff -- what a clamped store writes; bf16 stores are never clamped: none) and elements that are not finite. One pass, 16 bytes per
// lane, a wave reduction and one atomic pair per wave.
template <typename T>
__global__ __launch_bounds__(256) void count_saturated_kernel(const uint4 *x, long rows, int c8, int ldx8, unsigned *out) {
    const long total = rows * c8;
    unsigned sat = 0, bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const uint4 v = x[r * ldx8 + (i - r * c8)];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned e = (w[k] >> (16 * h)) & 0x7fffu;
                if (T::kDtype == TEDSPAD_F16) { sat += e == 0x7bffu; bad += e >= 0x7c00u; }
                else bad += e >= 0x7f80u;
            }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sat += __shfl_xor(sat, d, 64); bad += __shfl_xor(bad, d, 64); }
    if ((threadIdx.x & 63) == 0 && (sat | bad)) {
        if (sat) atomicAdd(out, sat);
        if (bad) atomicAdd(out + 1, bad);
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_count_saturated(const void *x, int64_t rows, int32_t c, int32_t ldx, int32_t dtype, uint32_t *out2, void *stream) {
    TS_REQUIRE(x && out2 && rows >= 0 && c > 0 && c % 8 == 0 && ldx >= c && ldx % 8 == 0 && (uintptr_t)x % 16 == 0, "tedspad_count_saturated: x 16-byte aligned, c and ldx multiples of 8, ldx >= c");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_count_saturated: dtype");
    if (rows == 0) return TEDSPAD_OK;
    const long total = rows * (c / 8);
    const int grid = (int)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((count_saturated_kernel<F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x, (long)rows, c / 8, ldx / 8, out2);
    else hipLaunchKernelGGL((count_saturated_kernel<BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x, (long)rows, c / 8, ldx / 8, out2);
    return check_launch("tedspad_count_saturated");
}

extern "C" int32_t tedspad_abi_version(void) { return TEDSPAD_ABI_VERSION; }
extern "C" const char *tedspad_last_error(void) { return g_err; }

extern "C" int32_t tedspad_maxpool_fwd_idx(const tedspad_pool_desc *d, const void *x, void *y, uint8_t *idx, void *stream);

extern "C" int32_t tedspad_maxpool_fwd(const tedspad_pool_desc *d, const void *x, void *y, void *stream) {
    return tedspad_maxpool_fwd_idx(d, x, y, nullptr, stream);
}

extern "C" int32_t tedspad_maxpool_fwd_idx(const tedspad_pool_desc *d, const void *x, void *y, uint8_t *idx, void *stream) {
    TS_REQUIRE(d && x && y, "tedspad_maxpool_fwd: null pointer");
    TS_REQUIRE(!idx || (d->kt * d->kh * d->kw <= 255 && (uintptr_t)idx % 8 == 0), "tedspad_maxpool_fwd_idx: window too large for 8-bit indices");
    TS_REQUIRE(d->c > 0 && d->c % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldx >= d->c && d->ldy >= d->c,
               "tedspad_maxpool_fwd: c/ldx/ldy must be multiples of 8");
    TS_REQUIRE(d->n > 0 && d->to > 0 && d->ho > 0 && d->wo > 0 && d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0,
               "tedspad_maxpool_fwd: bad geometry");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "tedspad_maxpool_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(d->dtype == TEDSPAD_F16 || d->dtype == TEDSPAD_BF16, "tedspad_maxpool_fwd: bad dtype");
    TS_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0 && d->pt < d->kt && d->ph < d->kh && d->pw < d->kw,
               "tedspad_maxpool_fwd: front padding must be smaller than the window");
    PoolKP p;
    p.x = (const uint16_t *)x; p.y = (uint16_t *)y; p.idx = idx;
    p.Ti = d->t; p.Hi = d->h; p.Wi = d->w; p.C8 = d->c / 8; p.ldx = d->ldx; p.ldy = d->ldy;
    p.To = d->to; p.Ho = d->ho; p.Wo = d->wo;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw; p.st = d->st; p.sh = d->sh; p.sw = d->sw; p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
    p.pad_zero = d->pad_zero;
    p.total = (long)d->n * d->to * d->ho * d->wo * p.C8;
    hipStream_t s = (hipStream_t)stream;
    if (!idx && d->kt == 3 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 1 && d->ph == 1 && d->pw == 1 &&
        d->to == d->t && d->ho == d->h && d->wo == d->w) {
        {   // every element fetched once (maxpool_k3s1_lds_kernel) where a frame band fits its per-thread budgets; TEDSPAD_POOL_NO_LDS=1: the column walk (A/B knob)
            static const bool lds_ok = getenv("TEDSPAD_POOL_NO_LDS") == nullptr;
            K3S1Geo g;
            g.CG = p.C8 < 8 ? p.C8 : 8;                    // 128 contiguous bytes per pixel (4 and 16 measured: -0.5 % each on InceptionI3d)
            while (g.CG * 2 <= p.C8 && d->w * g.CG * 2 <= 128 && (long)d->h * d->w * g.CG * 2 <= 2048) g.CG *= 2;       // small frames: more channel groups per workgroup
            const int rp = d->w * g.CG <= 256 ? 256 / (d->w * g.CG) : 0;      // tile rows per pass of the 256 threads
            int hb = d->h;
            while (rp && hb > 1 && ((hb + rp - 1) / rp > K3_NO || (hb + 2 + rp - 1) / rp > K3_NL)) --hb;
            g.HB = hb; g.bands = (d->h + hb - 1) / hb; g.cgroups = (p.C8 + g.CG - 1) / g.CG;
            const long lds = 2L * (hb + 2) * (d->w + 2) * g.CG * 16;
            const long wgs = (long)d->n * g.bands * g.cgroups;
            const bool fits = rp >= 1 && (hb + rp - 1) / rp <= K3_NO && (hb + 2 + rp - 1) / rp <= K3_NL && lds <= 150 * 1024 && wgs < (1L << 30) &&
                              (long)d->h * d->w * (d->ldx > d->ldy ? d->ldx : d->ldy) < (1L << 31) && (hb >= 4 || hb == d->h);
            if (lds_ok && fits) {
                const uint32_t ninf2 = d->dtype == TEDSPAD_F16 ? 0xFC00FC00u : 0xFF80FF80u;
                const uint32_t padw2 = d->pad_zero ? 0u : ninf2;
                static thread_local int attr_k3[2] = {0, 0};
                const int ti = d->dtype == TEDSPAD_F16 ? 0 : 1;
                if (!attr_k3[ti]) {
                    const void *fn = ti == 0 ? (const void *)maxpool_k3s1_lds_kernel<F16> : (const void *)maxpool_k3s1_lds_kernel<BF16>;
                    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                        set_error("tedspad_maxpool_fwd: cannot raise the dynamic LDS limit");
                        return TEDSPAD_ELAUNCH;
                    }
                    attr_k3[ti] = 1;
                }
                if (ti == 0) hipLaunchKernelGGL(maxpool_k3s1_lds_kernel<F16>, dim3((unsigned)wgs), dim3(256), (size_t)lds, s, p.x, p.y, d->t, d->h, d->w, p.C8, p.ldx, p.ldy, padw2, g);
                else hipLaunchKernelGGL(maxpool_k3s1_lds_kernel<BF16>, dim3((unsigned)wgs), dim3(256), (size_t)lds, s, p.x, p.y, d->t, d->h, d->w, p.C8, p.ldx, p.ldy, padw2, g);
                return check_launch("tedspad_maxpool_fwd");
            }
        }
        long rows = (long)d->n * d->t * ((d->h + 1) / 2) * p.C8;
        int segs = 1;                                             // enough threads for ~8 waves per SIMD on 256 CUs
        while (rows * segs < 256L * 2048 && (d->w + segs - 1) / segs > 4) ++segs;
        const int seglen = (d->w + segs - 1) / segs;
        const long tot = rows * segs;
        // a row whose window is entirely padding cannot occur (pad 1 < 3), so with "skip" semantics -inf never survives
        const uint32_t ninf = d->dtype == TEDSPAD_F16 ? 0xFC00FC00u : 0xFF80FF80u;
        const uint32_t padw = d->pad_zero ? 0u : ninf;
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL(maxpool_k3s1_kernel<F16>, dim3(grid_for(tot)), dim3(256), 0, s, p.x, p.y, d->n, d->t, d->h, d->w, p.C8, p.ldx, p.ldy, padw, segs, seglen, tot);
        else hipLaunchKernelGGL(maxpool_k3s1_kernel<BF16>, dim3(grid_for(tot)), dim3(256), 0, s, p.x, p.y, d->n, d->t, d->h, d->w, p.C8, p.ldx, p.ldy, padw, segs, seglen, tot);
        return check_launch("tedspad_maxpool_fwd");
    }
    if (!idx) {
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL(maxpool_noidx_kernel<F16>, dim3(grid_for(p.total)), dim3(256), 0, s, p, 0xFC00FC00u);
        else hipLaunchKernelGGL(maxpool_noidx_kernel<BF16>, dim3(grid_for(p.total)), dim3(256), 0, s, p, 0xFF80FF80u);
        return check_launch("tedspad_maxpool_fwd");
    }
    if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL(maxpool_kernel<F16>, dim3(grid_for(p.total)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(maxpool_kernel<BF16>, dim3(grid_for(p.total)), dim3(256), 0, s, p);
    return check_launch("tedspad_maxpool_fwd");
}

extern "C" int32_t tedspad_global_avgpool_fwd(const void *x, float *y, int32_t n, int32_t spatial, int32_t c, int32_t ldx,
                                              int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && spatial > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldx >= c, "tedspad_global_avgpool_fwd: bad arguments");
    TS_REQUIRE(((uintptr_t)x) % 16 == 0, "tedspad_global_avgpool_fwd: x must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_global_avgpool_fwd: bad dtype");
    const int blocks = n * ((c / 8 + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(avgpool_kernel<F16>, dim3(blocks), dim3(256), 0, s, (const uint16_t *)x, y, n, spatial, c / 8, ldx);
    else hipLaunchKernelGGL(avgpool_kernel<BF16>, dim3(blocks), dim3(256), 0, s, (const uint16_t *)x, y, n, spatial, c / 8, ldx);
    return check_launch("tedspad_global_avgpool_fwd");
}

extern "C" int32_t tedspad_clip_to_channels_last(const float *x, void *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w,
                                                 int64_t sn, int64_t sc, int64_t st_, int64_t sh, int64_t sw, int32_t cpad,
                                                 int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && t > 0 && h > 0 && w > 0, "tedspad_clip_to_channels_last: bad arguments");
    TS_REQUIRE((cpad == 4 || cpad == 8) && c <= cpad, "tedspad_clip_to_channels_last: cpad must be 4 or 8 and >= c");
    TS_REQUIRE(cpad == 8 || w % 2 == 0, "tedspad_clip_to_channels_last: cpad=4 packs pixel pairs, w must be even");
    TS_REQUIRE(((uintptr_t)y) % 16 == 0, "tedspad_clip_to_channels_last: y must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_clip_to_channels_last: bad dtype");
    const long total8 = (long)n * t * h * w * cpad / 8;
    hipStream_t s = (hipStream_t)stream;
    if (sw == 1 && w % 8 == 0 && c <= 4 && ((uintptr_t)x % 16 == 0) && sn % 4 == 0 && sc % 4 == 0 && st_ % 4 == 0 && sh % 4 == 0) {
        const long tot = (long)n * t * h * (w / 8);
        const dim3 g(grid_for(tot));
        if (cpad == 4) {
            if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((to_channels_last_w8_kernel<F16, 4>), g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st_, (long)sh, tot);
            else hipLaunchKernelGGL((to_channels_last_w8_kernel<BF16, 4>), g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st_, (long)sh, tot);
        } else {
            if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((to_channels_last_w8_kernel<F16, 8>), g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st_, (long)sh, tot);
            else hipLaunchKernelGGL((to_channels_last_w8_kernel<BF16, 8>), g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w / 8, (long)sn, (long)sc, (long)st_, (long)sh, tot);
        }
        return check_launch("tedspad_clip_to_channels_last");
    }
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(to_channels_last_kernel<F16>, dim3(grid_for(total8)), dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st_, (long)sh, (long)sw, cpad, total8);
    else hipLaunchKernelGGL(to_channels_last_kernel<BF16>, dim3(grid_for(total8)), dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st_, (long)sh, (long)sw, cpad, total8);
    return check_launch("tedspad_clip_to_channels_last");
}

extern "C" int32_t tedspad_channels_last_to_nchw(const void *x, float *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w,
                                                 int32_t ldx, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && t > 0 && h > 0 && w > 0 && ldx >= c, "tedspad_channels_last_to_nchw: bad arguments");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_channels_last_to_nchw: bad dtype");
    const long thw = (long)t * h * w, total = (long)n * c * thw;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(to_nchw_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, y, c, thw, ldx, total);
    else hipLaunchKernelGGL(to_nchw_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, y, c, thw, ldx, total);
    return check_launch("tedspad_channels_last_to_nchw");
}

extern "C" int32_t tedspad_upsample_bilinear2x_fwd(const void *x, void *y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx,
                                                   int32_t ldy, int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left,
                                                   int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c,
               "tedspad_upsample_bilinear2x_fwd: bad arguments");
    TS_REQUIRE(pad_top >= 0 && pad_left >= 0 && ho >= 2 * h + pad_top && wo >= 2 * w + pad_left,
               "tedspad_upsample_bilinear2x_fwd: output smaller than the upsampled map");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "tedspad_upsample_bilinear2x_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_upsample_bilinear2x_fwd: bad dtype");
    const long total = (long)n * ho * wo * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(upsample2x_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, (uint16_t *)y, h, w, c / 8, ldx, ldy, ho, wo, pad_top, pad_left, total);
    else hipLaunchKernelGGL(upsample2x_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, (uint16_t *)y, h, w, c / 8, ldx, ldy, ho, wo, pad_top, pad_left, total);
    return check_launch("tedspad_upsample_bilinear2x_fwd");
}

extern "C" int32_t tedspad_upsample_nearest2x_fwd(const void *x, void *y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldx, int32_t ldy,
                                                  void *stream) {
    TS_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c,
               "tedspad_upsample_nearest2x_fwd: bad arguments");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "tedspad_upsample_nearest2x_fwd: pointers must be 16-byte aligned");
    const long total = (long)n * 2 * h * 2 * w * (c / 8);
    hipLaunchKernelGGL(upsample_nearest2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x, (uint4 *)y, h, w, c / 8,
                       ldx / 8, ldy / 8, total);
    return check_launch("tedspad_upsample_nearest2x_fwd");
}

extern "C" int32_t tedspad_copy_channels(const void *x, void *y, int64_t npix, int32_t c, int32_t ldx, int32_t ldy, void *stream) {
    TS_REQUIRE(x && y && npix > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c, "tedspad_copy_channels: bad arguments");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "tedspad_copy_channels: pointers must be 16-byte aligned");
    const long total = (long)npix * (c / 8);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x, (uint4 *)y, c / 8, ldx / 8, ldy / 8, total);
    return check_launch("tedspad_copy_channels");
}

extern "C" int32_t tedspad_upsample_nearest2x_bwd(const void *dy, void *dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ldy, int32_t ldx,
                                                  int32_t accumulate, int32_t dtype, void *stream) {
    TS_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c,
               "tedspad_upsample_nearest2x_bwd: bad arguments");
    TS_REQUIRE(((uintptr_t)dy | (uintptr_t)dx) % 16 == 0, "tedspad_upsample_nearest2x_bwd: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_upsample_nearest2x_bwd: bad dtype");
    const long total = (long)n * h * w * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(upsample_nearest2x_bwd_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint4 *)dy, (uint4 *)dx, h, w, c / 8, ldy / 8, ldx / 8, accumulate, total);
    else hipLaunchKernelGGL(upsample_nearest2x_bwd_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint4 *)dy, (uint4 *)dx, h, w, c / 8, ldy / 8, ldx / 8, accumulate, total);
    return check_launch("tedspad_upsample_nearest2x_bwd");
}

extern "C" int32_t tedspad_add_channels(const void *x, void *y, int64_t npix, int32_t c, int32_t ldx, int32_t ldy, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && npix > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= c && ldy >= c, "tedspad_add_channels: bad arguments");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "tedspad_add_channels: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_add_channels: bad dtype");
    const long total = (long)npix * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(add_channels_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint4 *)x, (uint4 *)y, c / 8, ldx / 8, ldy / 8, total);
    else hipLaunchKernelGGL(add_channels_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint4 *)x, (uint4 *)y, c / 8, ldx / 8, ldy / 8, total);
    return check_launch("tedspad_add_channels");
}

extern "C" int32_t tedspad_avgpool3d_s1_fwd(const void *x, float *y, int32_t n, int32_t t, int32_t h, int32_t w, int32_t c, int32_t ldx, int32_t kt,
                                            int32_t kh, int32_t kw, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && c % 8 == 0 && ldx % 8 == 0 && ldx >= c && kt > 0 && kh > 0 && kw > 0 && t >= kt && h >= kh && w >= kw,
               "tedspad_avgpool3d_s1_fwd: bad arguments (the map must be at least as large as the kernel)");
    TS_REQUIRE(((uintptr_t)x) % 16 == 0, "tedspad_avgpool3d_s1_fwd: x must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_avgpool3d_s1_fwd: bad dtype");
    const int to = t - kt + 1, ho = h - kh + 1, wo = w - kw + 1;
    const long total = (long)n * to * ho * wo * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(avgpool3d_s1_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, y, t, h, w, c / 8, ldx, kt, kh, kw, to, ho, wo, total);
    else hipLaunchKernelGGL(avgpool3d_s1_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)x, y, t, h, w, c / 8, ldx, kt, kh, kw, to, ho, wo, total);
    return check_launch("tedspad_avgpool3d_s1_fwd");
}
