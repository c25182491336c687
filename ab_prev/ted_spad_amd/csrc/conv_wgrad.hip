// Weight gradient of the channels-last convolution for gfx950:
//
//   dW[co][k] = sum_m dY[m][co] * X[m @ tap(k)][ci(k)],     k = (dt,dh,dw,ci), m = (n,to,ho,wo)
//
// (reference: autograd of nn.Conv3d / nn.Conv2d inside loss.backward(), train_anonymizer.py:122,190-191).
// The reduction index is the PIXEL, but both tensors are channels-last (channels contiguous), so
// the MFMA operands need a transpose: both tiles are DMA'd into LDS exactly as they sit in HBM
// ([64 pixels][64 channels], the X tile gathered per tap with the same table/zero-page scheme as the
// forward kernel) and the fragments are produced by the CDNA4 transposing LDS read
// `ds_read_b64_tr_b16` (4 pixels x 16 channels per 16-lane group, delivered pixel-major).
// A workgroup owns a [128 co] x [128|256 k] (or [64 co] x [256 k]) tile of dW and a slice of the pixels; partial tiles are
// summed into the fp32 dW buffer with no-return float atomics (128-byte row segments: the shape
// the memory-side atomic units run at full rate).  3-stage LDS-DMA ring, counted vmcnt, one
// barrier per 64 pixels -- the same pipeline as the forward kernel.
#include <stdlib.h>

#include "common.h"

namespace tedspad {
namespace {

struct WgradKP {
    const uint16_t *x;
    const uint16_t *dy;
    const int2 *ktab;
    float *dw;
    int M, Cout, Kpad, nkc;       // nkc = Kpad / 64
    int Ti, Hi, Wi, ldx;
    int To, Ho, Wo, ldy;
    int st, sh, sw, pt, ph, pw;
    int k_tiles, co_tiles, rows_per_split;
    unsigned long long mWo, mHo, mTo;   // ceil(2^40 / d): n / d == (n * m) >> 40 for n < 2^23, d < 2^12
};

__device__ uint4 g_zero16w;

typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned long long m) { return (unsigned)(((unsigned long long)n * m) >> 40); }

constexpr int WG_SUB = 64 * 64 * 2;      // one [64 px][64 ch] sub-tile
constexpr int WG_S = 3;

// NY x NX waves (4 or 8): the workgroup's dW tile is [64*NY co] x [64*NX k], one 64x64 block per wave. (2,2) is the
// square tile; (1,4) serves Cout <= 64 layers (the outer UNet levels, where most pixels are) without computing a zero
// half-tile; (2,4) with 8 waves moves 1.36x fewer L2->LDS bytes per FLOP (this kernel is bound by that stream).
template <typename T, int NY, int NX>
__global__ __launch_bounds__(64 * NY * NX) void conv_wgrad_kernel(const WgradKP p) {
    constexpr int NW = NY * NX;                      // waves
    constexpr int NP = 8 / NW;                       // passes of NW*8 rows that fill a 64-row sub-tile
    constexpr int WG_STAGE = (NY + NX) * WG_SUB;     // dY sub-tiles, then X sub-tiles
    constexpr int WG_L = NP * (NY + NX);             // DMA instructions per thread per stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_S * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int kt2 = b % p.k_tiles; b /= p.k_tiles;
    const int cot = b % p.co_tiles;
    const int ms = b / p.co_tiles;
    const int m_begin = ms * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nsteps = (m_end - m_begin + 63) / 64;

    // ---- DMA roles ---------------------------------------------------------------------------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);   // source chunk (swizzle on the source)
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16w);
    // this thread's chunk of the two X sub-tiles: tap offsets are fixed for the whole kernel
    int xoff[NX], xdt[NX], xdh[NX], xdw[NX];
    bool xok[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int kchunk = kt2 * NX + j;
        xok[j] = kchunk < p.nkc;
        const int2 e = xok[j] ? p.ktab[kchunk * 8 + kc] : make_int2(0, 31 | (31 << 8) | (31 << 16));
        xoff[j] = e.x;
        xdt[j] = e.y & 255; xdh[j] = ((e.y >> 8) & 255) - 8; xdw[j] = (e.y >> 16) - 16;
        xok[j] = xok[j] && xdt[j] < 8;
    }
    bool yok[NY];
#pragma unroll
    for (int j = 0; j < NY; ++j) yok[j] = cot * (64 * NY) + j * 64 + kc * 8 < p.Cout;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

    auto issue = [&](int step, int slot) {
        const unsigned stage = lds0 + slot * WG_STAGE + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int m = m_begin + step * 64 + i * (NW * 8) + rsub;
            const bool mok = m < m_end;
            const unsigned q1 = fdiv(m, p.mWo); const int wo = m - q1 * p.Wo;
            const unsigned q2 = fdiv(q1, p.mHo); const int ho = q1 - q2 * p.Ho;
            const unsigned n = fdiv(q2, p.mTo); const int to = q2 - n * p.To;
            const int t0 = to * p.st - p.pt, h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
            const int base = (((int)(n * p.Ti + t0) * p.Hi + h0) * p.Wi + w0) * p.ldx;
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                const uint16_t *src = (mok && yok[j]) ? p.dy + (size_t)m * p.ldy + cot * (64 * NY) + j * 64 + kc * 8 : zero;
                lds_dma16(src, stage + j * WG_SUB + i * (NW * 8) * 128);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const bool ok = mok && xok[j] && (unsigned)(t0 + xdt[j]) < (unsigned)p.Ti && (unsigned)(h0 + xdh[j]) < (unsigned)p.Hi &&
                                (unsigned)(w0 + xdw[j]) < (unsigned)p.Wi;
                const uint16_t *src = ok ? p.x + (ptrdiff_t)(base + xoff[j]) : zero;
                lds_dma16(src, stage + (NY + j) * WG_SUB + i * (NW * 8) * 128);
            }
        }
    };

    // ---- MFMA roles: wave (wm, wn) owns dW[64 co of dY sub-tile wm][64 k of X sub-tile wn] ------
    const int wm = wave % NY, wn = wave / NY;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
    // transposing read: lane 16g+4q+pp supplies row (8h + 4rd + q), columns 16(g&1)+4pp.. of a 32-column tile
    int off[2][2];   // [tile a|b][rd]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 8 * h + 4 * rd + q;
            const int c = 32 * a + 16 * (g & 1) + 4 * pp;
            off[a][rd] = row * 128 + ((((c >> 3) ^ ((row >> 1) & 7))) << 4) + (c & 7) * 2;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    auto tr = [&](const unsigned char *base) -> uint2 {
        const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(base));
        return __builtin_bit_cast(uint2, v);
    };

#pragma unroll
    for (int s = 0; s < WG_S - 1; ++s)
        if (s < nsteps) issue(s, s);
    int rdslot = 0, wrslot = WG_S - 1;
    for (int step = 0; step < nsteps; ++step) {
        if (nsteps - 1 - step >= 1) wait_vmcnt<WG_L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + WG_S - 1 < nsteps) issue(step + WG_S - 1, wrslot);
        const unsigned char *Y = smem + rdslot * WG_STAGE + wm * WG_SUB;
        const unsigned char *X = smem + rdslot * WG_STAGE + (NY + wn) * WG_SUB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const uint2 lo = tr(Y + ks * 16 * 128 + off[a][0]), hi = tr(Y + ks * 16 * 128 + off[a][1]);
                fa[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                const uint2 lo2 = tr(X + ks * 16 * 128 + off[a][0]), hi2 = tr(X + ks * 16 * 128 + off[a][1]);
                fb[a] = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][c] = T::mfma(fa[a], fb[c], acc[a][c]);
        }
        rdslot = rdslot + 1 == WG_S ? 0 : rdslot + 1;
        wrslot = wrslot + 1 == WG_S ? 0 : wrslot + 1;
    }

    // ---- partial tile -> fp32 dW with float atomics (lanes 0-31 / 32-63: two 128-byte row segments) ---
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = kt2 * (64 * NX) + wn * 64 + c * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot * (64 * NY) + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout && k < p.Kpad) atomicAdd(p.dw + (size_t)co * p.Kpad + k, acc[a][c][r]);
            }
        }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_conv_wgrad(const tedspad_conv_desc *d, const void *x, const void *dy, const int32_t *ktab, float *dw,
                                      void *stream) {
    TS_REQUIRE(d && x && dy && ktab && dw, "tedspad_conv_wgrad: null pointer");
    const int kpad = tedspad_conv_kpad(d);
    TS_REQUIRE(kpad > 0, "tedspad_conv_wgrad: bad descriptor");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)dy) % 16 == 0, "tedspad_conv_wgrad: pointers must be 16-byte aligned");
    const long M = (long)d->n * d->to * d->ho * d->wo;
    const long in_elems = (long)d->n * d->t * d->h * d->w * d->ldx;
    TS_REQUIRE(M < (1L << 23) && in_elems < (1L << 31), "tedspad_conv_wgrad: at most 2^23 output pixels per call; split the batch");
    TS_REQUIRE(d->wo < 4096 && d->ho < 4096 && d->to < 4096, "tedspad_conv_wgrad: output dims must be < 4096");
    TS_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0, "tedspad_conv_wgrad: negative padding");
    WgradKP p;
    p.x = (const uint16_t *)x; p.dy = (const uint16_t *)dy; p.ktab = (const int2 *)ktab; p.dw = dw;
    p.M = (int)M; p.Cout = d->cout; p.Kpad = kpad; p.nkc = kpad / 64;
    p.Ti = d->t; p.Hi = d->h; p.Wi = d->w; p.ldx = d->ldx;
    p.To = d->to; p.Ho = d->ho; p.Wo = d->wo; p.ldy = d->ldy;
    p.st = d->st; p.sh = d->sh; p.sw = d->sw; p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
    // dW tile shape: 128 co x 128 k, or 64 co x 256 k when a 128-wide co tile would be at least half padding
    const bool narrow = ((d->cout + 127) / 128) * 128 - d->cout >= 64;
    // 8 waves / 256-wide k tile where that pads K by <= 15 % (K = 64 * nkc)
    static const bool allow_wide = getenv("TEDSPAD_WGRAD_NO_WIDE") == nullptr;      // A/B knob
    const bool wide = allow_wide && !narrow && ((p.nkc + 3) / 4) * 4 * 100 <= p.nkc * 115;
    p.k_tiles = (narrow || wide) ? (p.nkc + 3) / 4 : (p.nkc + 1) / 2;
    p.co_tiles = narrow ? (d->cout + 63) / 64 : (d->cout + 127) / 128;
    auto magic = [](unsigned dv) { return ((1ULL << 40) + dv - 1) / dv; };
    p.mWo = magic(d->wo); p.mHo = magic(d->ho); p.mTo = magic(d->to);
    // split the pixels so the grid covers the chip ~4x, each split a whole number of 64-pixel steps (>= 8 steps)
    const long tiles = (long)p.k_tiles * p.co_tiles;
    long splits = (1024 + tiles - 1) / tiles;
    const long max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long rows = (M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    splits = (M + rows - 1) / rows;
    p.rows_per_split = (int)rows;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(tiles * splits));
    if (narrow) {
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL((conv_wgrad_kernel<F16, 1, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<BF16, 1, 4>), grid, dim3(256), 0, s, p);
    } else if (wide) {
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL((conv_wgrad_kernel<F16, 2, 4>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<BF16, 2, 4>), grid, dim3(512), 0, s, p);
    } else {
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL((conv_wgrad_kernel<F16, 2, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<BF16, 2, 2>), grid, dim3(256), 0, s, p);
    }
    return check_launch("tedspad_conv_wgrad");
}
