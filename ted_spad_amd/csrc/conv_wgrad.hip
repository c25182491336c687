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
#include "det_gate.h"

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
    int cc_tiles;                       // conv_wgrad3_kernel: Cin / 64
    unsigned long long mWo, mHo, mTo;   // ceil(2^40 / d): n / d == (n * m) >> 40 for n < 2^23, d < 2^12
};

__device__ uint4 g_zero16w;

typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned long long m) { return (unsigned)(((unsigned long long)n * m) >> 40); }

constexpr int WG_SUB = 64 * 64 * 2;      // one [64 px][64 ch] sub-tile

// LDS row r (one pixel, 64 channels = 8 chunks of 16 bytes) holds channel chunk c at chunk position c ^ wg_swz(r). (A swizzle derived from the
// bank table of `ds_read_b64_tr_b16` -- every second row pair moved to the other 64-byte half, ((r >> 1) & 1) << 2 -- measured the same
// alone and 0.6 ms per cfg3 iteration WORSE beside the other stream's kernels: these kernels are not bound by their fragment reads.)
__device__ __forceinline__ int wg_swz(int row) { return (row >> 1) & 7; }
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
    int b = xcd_remap(blockIdx.x, gridDim.x);      // the (k, co) tiles of one pixel split share its x / dY rows: one XCD (one L2) walks them
    const int kt2 = b % p.k_tiles; b /= p.k_tiles;
    const int cot = b % p.co_tiles;
    const int ms = b / p.co_tiles;
    const int m_begin = ms * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nsteps = (m_end - m_begin + 63) / 64;

    // ---- DMA roles ---------------------------------------------------------------------------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ wg_swz(wave * 8 + (lane >> 3));          // source chunk (swizzle on the source; the passes' row offsets are multiples of 16)
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
            off[a][rd] = row * 128 + ((((c >> 3) ^ wg_swz(row))) << 4) + (c & 7) * 2;
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
    const bool det = det_enter();                 // deterministic mode (det_gate.h): the workgroups flush one at a time, in blockIdx order
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
    det_exit(det);
}

// ---- 1 x 3 x 3, stride 1, pad 1, Cin % 64 == 0: the nine taps from THREE activation tiles -------------------------------------
// The kernel above gathers a [64 px][64 ch] activation tile per (tap, channel chunk): for a 3 x 3 conv that is nine tiles per 64 pixels,
// 120 KB of L2 -> LDS traffic with dY re-read by each K tile -- the stream this kernel family is bound by (the UNet's 64-channel
// full-resolution layers ran at ~330 TFLOP/s). Here a workgroup owns ALL nine taps of a [64 co] x [64 ci] block: per 64 consecutive
// output pixels it loads dY once and, per kernel row dh, ONE tile of 66 consecutive input pixels (m-1 .. m+64 of the row above / the same
// row / the row below in flat (n,h,w) order); the taps dw = -1, 0, +1 are that tile read at row offsets 0, 1, 2. 33 KB per 64 pixels.
//   * pixels of rows outside the image come from the zero page (decided per slot from its own (h + dh - 1));
//   * dw = +-1 at the left / right image edge would read the neighbouring ROW's end pixel: the waves that own those taps zero the
//     dY elements of the edge pixels in their A fragments instead (a 64-bit mask per step, built on the scalar unit);
//   * 6 waves = (dw) x (32-channel half of ci): a wave's A fragments (dY, masked for its dw) serve its three dh taps; accumulators
//     3 dh x 2 co halves x 16; two-slot LDS ring (70 KB: two workgroups per CU), one barrier per 64 pixels.
// What bounds it (measured: with its fragment reads and MFMAs removed the kernel takes as long as the whole one): the L2 -> LDS stream itself,
// 2.6 GB per full-resolution 64 -> 64 layer at ~4 TB/s over the chip (13-16 GB/s per CU; the guide's figure for LDS-DMA bursts that come
// from HBM is ~23 GB/s per CU). A 12-wave version with a 4-slot ring (tiles requested three steps ahead, row tracking on the scalar unit,
// 4x fewer address instructions), touching the next tiles' lines early and a swizzle built for the transposing reads' banks all left its
// ~600-700 TFLOP/s where they were -- and the 12-wave one, alone on its CU with 152 KB of LDS, lost what this one gains by sharing the CU with
// the other stream's kernels (ConvLayer.wgrad runs on a side stream). The next factor is bytes again: a ring of activation ROWS in LDS
// would fetch each input pixel once instead of once per dh (16 instead of 33 KB per 64 pixels).
constexpr int W3_XROWS = 72;                       // 66 used
constexpr int W3_STAGE = WG_SUB + 3 * W3_XROWS * 128;

template <typename T>
__global__ __launch_bounds__(384) void conv_wgrad3_kernel(const WgradKP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * W3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);      // the (k, co) tiles of one pixel split share its x / dY rows: one XCD (one L2) walks them
    const int cc = b % p.cc_tiles; b /= p.cc_tiles;
    const int cot = b % p.co_tiles;
    const int ms = b / p.co_tiles;
    const int m_begin = ms * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nsteps = (m_end - m_begin + 63) / 64;
    const int W = p.Wi, H = p.Hi;
    const unsigned long long mH = p.mHo;

    // ---- DMA roles: a wave moves 8 rows x 128 bytes per instruction; rows 48.. of a tile by the first waves ----------------------
    const int kc = (lane & 7) ^ wg_swz(wave * 8 + (lane >> 3));             // source chunk (swizzle on the source: LDS row r holds chunk c at c ^ wg_swz(r); 48-row passes keep it)
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16w);
    const bool yok = cot * 64 + kc * 8 < p.Cout;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const uint16_t *xc = p.x + cc * 64 + kc * 8;
    const uint16_t *yc = p.dy + cot * 64 + kc * 8;

    // Per-lane state of the two rows (ps = 0, 1) this lane moves per tile, advanced by 64 pixels per step: the address arithmetic of a step
    // (pixel -> (row, column), three validity tests, four 64-bit addresses) was as long as its MFMA work when redone from the pixel index.
    int qs[2], wq[2], hq[2];
    const uint16_t *px[2], *py[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int row = ps * 48 + wave * 8 + (lane >> 3);
        const int q = m_begin + row - 1;                                    // the flat pixel of X slot `row`; dY row `row` is pixel q + 1
        const int qc = q < 0 ? 0 : q;
        const unsigned qr = fdiv(qc, p.mWo);
        qs[ps] = q;
        wq[ps] = qc - (int)qr * W;
        hq[ps] = (int)(qr - fdiv(qr, mH) * H);
        px[ps] = xc + (ptrdiff_t)q * p.ldx;
        py[ps] = yc + (ptrdiff_t)(q + 1) * p.ldy;
    }
    const ptrdiff_t xrow = (ptrdiff_t)W * p.ldx;

    auto issue = [&](int slot) {                        // the tiles of the NEXT step not issued yet
        const unsigned stage = lds0 + slot * W3_STAGE;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r0 = ps * 48 + wave * 8;                              // wave-uniform
            const int row = r0 + (lane >> 3);
            const int q = qs[ps];
            if (r0 < 64) {
                const uint16_t *src = (q + 1 < m_end && yok) ? py[ps] : zero;
                lds_dma16(src, stage + r0 * 128);
            }
            if (r0 < W3_XROWS) {
                const bool qok = row < 66 && q >= 0 && q < p.M;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const bool ok = qok && (unsigned)(hq[ps] + dh - 1) < (unsigned)H;
                    const uint16_t *src = ok ? px[ps] + (dh - 1) * xrow : zero;
                    lds_dma16(src, stage + WG_SUB + dh * (W3_XROWS * 128) + r0 * 128);
                }
            }
            // advance to the next step
            qs[ps] = q + 64;
            px[ps] += (ptrdiff_t)64 * p.ldx;
            py[ps] += (ptrdiff_t)64 * p.ldy;
            int wn = wq[ps] + (q < 0 ? 63 : 64), hn = hq[ps];               // (slot 0 of the first split starts at pixel -1, set up as pixel 0)
            while (wn >= W) { wn -= W; hn = hn + 1 == H ? 0 : hn + 1; }
            wq[ps] = wn; hq[ps] = hn;
        }
    };

    // ---- MFMA roles ----------------------------------------------------------------------------------------------------------------
    const int dw = wave % 3, cih = wave / 3;            // tap column (shift dw: rows +0, +1, +2 of the X tiles), ci half
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
    int offa[2][2], offb[2];                            // dY: [co half][rd]; X: [rd]
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int row = 8 * h + 4 * rd + q4;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int c = 32 * a + 16 * (g & 1) + 4 * pp;
            offa[a][rd] = row * 128 + ((((c >> 3) ^ wg_swz(row))) << 4) + (c & 7) * 2;
        }
        const int rb = row + dw, cb = 32 * cih + 16 * (g & 1) + 4 * pp;
        offb[rd] = rb * 128 + ((((cb >> 3) ^ wg_swz(rb))) << 4) + (cb & 7) * 2;
    }
    f32x16 acc[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][a][r] = 0.f;
    auto tr = [&](const unsigned char *base) -> uint2 {
        const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(base));
        return __builtin_bit_cast(uint2, v);
    };

    if (nsteps > 0) issue(0);
    int wbase = m_begin % W;                            // column of the step's first pixel
    for (int step = 0; step < nsteps; ++step) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + 1 < nsteps) issue((step + 1) & 1);
        // edge pixels of this step for the wave's tap column: dw = 0 (shift -1) must not see w == 0, dw = 2 (shift +1) not w == W - 1
        unsigned long long edge = 0;
        if (dw != 1) {
            int p0 = dw == 0 ? W - wbase : W - 1 - wbase;
            if (p0 >= W) p0 -= W;
            for (; p0 < 64; p0 += W) edge |= 1ull << p0;
        }
        wbase = (wbase + 64) % W;
        const unsigned char *Y = smem + (step & 1) * W3_STAGE;
        const unsigned char *X = Y + WG_SUB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 fa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const uint2 lo = tr(Y + ks * 16 * 128 + offa[a][0]), hi = tr(Y + ks * 16 * 128 + offa[a][1]);
                fa[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            const unsigned bits = (unsigned)(edge >> (16 * ks + 8 * h)) & 0xffu;      // element j of the fragment = pixel 16 ks + 8 h + j
            if (bits) {
                unsigned mk[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) mk[d] = ((bits >> (2 * d)) & 1u ? 0u : 0xffffu) | ((bits >> (2 * d + 1)) & 1u ? 0u : 0xffff0000u);
#pragma unroll
                for (int a = 0; a < 2; ++a) { fa[a].x &= mk[0]; fa[a].y &= mk[1]; fa[a].z &= mk[2]; fa[a].w &= mk[3]; }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const unsigned char *Xd = X + d * (W3_XROWS * 128) + ks * 16 * 128;
                const uint2 lo = tr(Xd + offb[0]), hi = tr(Xd + offb[1]);
                const uint4 fb = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[d][a] = T::mfma(fa[a], fb, acc[d][a]);
            }
        }
    }

    // ---- partial tile -> fp32 dW with float atomics: k = ((dh * 3 + dw) * Cin + cc * 64 + cih * 32 + lane % 32) -----------------------
    const bool det = det_enter();
    const int l31 = lane & 31, lh = lane >> 5;
    const int Cin = p.cc_tiles * 64;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int k = (d * 3 + dw) * Cin + cc * 64 + cih * 32 + l31;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) atomicAdd(p.dw + (size_t)co * p.Kpad + k, acc[d][a][r]);
            }
    }
    det_exit(det);
}


// ---- the nine taps from ONE 6 x 18 halo of a 4 x 16 output patch -----------------------------------------------------------------------------------------
// conv_wgrad3_kernel walks flat pixels and takes three 66-row activation tiles per 64 outputs (33 KB with dY, each input pixel once per kernel row) plus per-lane
// row tracking and edge masks. Here a step is a PATCH of 4 rows x 16 columns of one image: its dY tile (64 rows, 8 KB) and ONE halo of 6 x 18 input positions
// (108 rows, 13.5 KB) that serves all nine taps -- tap (dh, dw) of patch row r is the 16 halo rows from (r + dh) * 18 + dw -- 21.5 KB per 64 outputs. Positions
// outside the image are zero rows, decided per DMA slot from the patch origin: no edge masks, no row tracking. The 22 KB stage leaves room for a THREE-slot ring
// at two workgroups per CU (the three-tile kernel: two slots, 70 KB), so a step's tiles are requested two steps ahead. MFMA roles, fragment layout and the flush
// are conv_wgrad3_kernel's. Frames whose width is not a multiple of 16 pay for the zero columns (W = 56: 12.5 %): the host picks per layer (launcher below).
constexpr int W3P_STAGE = WG_SUB + 112 * 128;      // 22 KB: 64 dY rows + 112 halo rows (108 used: 14 DMA groups of 8)
constexpr int W3P_STAGE0 = WG_SUB + 128 * 128;     // without loader waves: 16 halo groups (4 DMA instructions on each of the 6 waves)

struct Wgrad3pGeo {
    int tiles_h, tiles_w, npatch, patches_per_split;
};

// NLD > 0: NLD LOADER waves (6 ..) issue every LDS-DMA instruction of the step (22 groups: the halo's rows 112 .. 127 are never read) and do the counted waits; the six
// compute waves touch no global memory between the first barrier and the flush, so the ~300 cycles a wave sits behind each DMA instruction it issues (4 per step
// against 24 MFMAs) leave the MFMA pipes (tile 40's loader waves, conv_patch3.hip): 64 -> 64 at 384 x 112^2 545 -> 420 us, the other UNet shapes -13 ... -19 %.
// Where a step goes then (scripts/w3p_ablate.sh, 64 -> 64: 419 us whole): MFMAs + barriers alone 265 us (six compute waves on four SIMDs: two of them carry twice
// the MFMAs), DMA + barriers alone 247 us (1.23 GB: 5 TB/s, the HBM rate), fragment reads + MFMAs without DMA 333 us -- the layer sits AT the ridge (288 FLOP per byte),
// and the three streams overlap to 0.63 of their sum. Both forms: a halo row's B fragment is read ONCE and serves the up to three (patch row, kernel row) pairs on it
// -- 6 fragment pairs per step instead of 12 (on its own: no change). S: slots of the ring.
template <typename T, int NLD, int S>
__global__ __launch_bounds__(384 + 64 * NLD) void conv_wgrad3p_kernel(const WgradKP p, const Wgrad3pGeo g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);      // the (k, co) tiles of one pixel split share its x / dY rows: one XCD (one L2) walks them
    const int cc = b % p.cc_tiles; b /= p.cc_tiles;
    const int cot = b % p.co_tiles;
    const int ms = b / p.co_tiles;
    const int p_begin = ms * g.patches_per_split;
    const int p_end = min(g.npatch, p_begin + g.patches_per_split);
    const int nsteps = p_end - p_begin;
    const int W = p.Wi, H = p.Hi;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16w);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

    // ---- DMA roles: groups of 8 rows (8 of dY, then the halo's), a lane moves chunk (lane & 7) ^ swz(row) of its row. Without loaders: 24 groups, j * 6 + wave for
    // j = 0 .. 3; with: the 22 groups that hold rows a fragment reads, j * 2 + (wave - 6) for j = 0 .. 10 (dY for j < 4 on both) -------------------------------------
    constexpr bool LDR = NLD > 0;
    constexpr int STG = LDR ? W3P_STAGE : W3P_STAGE0;
    constexpr int NG = LDR ? 22 : 24, JW = LDR ? NLD : 6, NJ = (NG + JW - 1) / JW;
    const int w0 = LDR ? wave - 6 : wave;
    const int n_mine = (NG - w0 + JW - 1) / JW;          // groups of this wave (wave-uniform)
    auto role = [&](int j, int &r_r, int &r_c, int &kcs, bool &isy) {
        const int gi = j * JW + w0;                   // wave-uniform
        isy = gi < 8;
        const int row = (isy ? gi : gi - 8) * 8 + (lane >> 3);
        kcs = ((lane & 7) ^ wg_swz(row)) * 8;
        if (isy) { r_r = row >> 4; r_c = row & 15; }
        else { r_r = row < 108 ? row / 18 : 64; r_c = row < 108 ? row - (row / 18) * 18 : 0; }      // rows 108.. : never inside an image
    };
    auto issue = [&](int step, int slot) {
#ifdef TEDSPAD_W3P_ABLATE
        if (TEDSPAD_W3P_ABLATE & 4) return;
#endif
        const int pi = p_begin + step;
        const int tx = pi % g.tiles_w, t2 = pi / g.tiles_w;
        const int ox = tx * 16, oy = (t2 % g.tiles_h) * 4, f = t2 / g.tiles_h;
        const unsigned stage = lds0 + slot * STG;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gi = j * JW + w0;
            if (gi >= NG) break;                          // wave-uniform (the last round of an uneven split)
            int r_r, r_c, kcs; bool isy;
            role(j, r_r, r_c, kcs, isy);
            const uint16_t *src;
            if (isy) {
                const int y = oy + r_r, x = ox + r_c;
                const bool ok = y < H && x < W && cot * 64 + kcs < p.Cout;
                src = ok ? p.dy + ((size_t)(f * H + y) * W + x) * p.ldy + cot * 64 + kcs : zero;
            } else {
                const int y = oy - 1 + r_r, x = ox - 1 + r_c;
                const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                src = ok ? p.x + ((size_t)(f * H + y) * W + x) * p.ldx + cc * 64 + kcs : zero;
            }
            lds_dma16(src, stage + (isy ? gi * 1024 : WG_SUB + (gi - 8) * 1024));
        }
    };

#define W3_SEG() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    if (LDR && wave >= 6) {                            // ---- the loader waves: the ring's producer ----
        for (int i = 0; i < S - 1 && i < nsteps; ++i) issue(i, i);
        for (int step = 0; step < nsteps; ++step) {
            // step's pieces landed; those of the up to S - 2 steps after it may stay in flight (a count the switch does not know waits for everything: stricter, never wrong)
            const int ahead = min(S - 2, nsteps - 1 - step);
            switch (ahead * n_mine) {
                case 3: wait_vmcnt<3>(); break;
                case 4: wait_vmcnt<4>(); break;
                case 5: wait_vmcnt<5>(); break;
                case 6: wait_vmcnt<6>(); break;
                case 8: wait_vmcnt<8>(); break;
                case 10: wait_vmcnt<10>(); break;
                case 11: wait_vmcnt<11>(); break;
                case 12: wait_vmcnt<12>(); break;
                case 15: wait_vmcnt<15>(); break;
                case 16: wait_vmcnt<16>(); break;
                case 18: wait_vmcnt<18>(); break;
                case 20: wait_vmcnt<20>(); break;
                case 22: wait_vmcnt<22>(); break;
                case 24: wait_vmcnt<24>(); break;
                case 33: wait_vmcnt<33>(); break;
                default: wait_vmcnt<0>(); break;
            }
            W3_SEG();
            if (step + S - 1 < nsteps) issue(step + S - 1, (step + S - 1) % S);            // its slot held step - 1: every wave is past it
        }
        return;
    }

    // ---- MFMA roles (conv_wgrad3_kernel's) ------------------------------------------------------------------------------------------------------------------------
    const int dw = wave % 3, cih = wave / 3;
    const int gq = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3, h = gq >> 1;
    int offa[2][2], rowb[2];
    const int cb = 32 * cih + 16 * (gq & 1) + 4 * pp;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int row = 8 * h + 4 * rd + q4;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int c = 32 * a + 16 * (gq & 1) + 4 * pp;
            offa[a][rd] = row * 128 + ((((c >> 3) ^ wg_swz(row))) << 4) + (c & 7) * 2;       // + ks * 16 rows: 16 * 128 bytes keep wg_swz (16 rows = 2 periods)
        }
        rowb[rd] = row + dw;
    }
    f32x16 acc[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][a][r] = 0.f;
    auto tr = [&](const unsigned char *base) -> uint2 {
        const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(base));
        return __builtin_bit_cast(uint2, v);
    };

    uint4 fa[4][2], fb[6];                                                 // dY fragments of patch row ks (k-step), x fragments of halo row hr = ks + d
#ifdef TEDSPAD_W3P_ABLATE                                                  // timing experiments only (scripts/w3p_ablate.sh): 1 = no MFMAs, 2 = no fragment reads, 4 = no DMA
    if (TEDSPAD_W3P_ABLATE & 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i >> 1][i & 1] = make_uint4(lane, i, 0, 0); asm volatile("" : "+v"(fa[i >> 1][i & 1].x), "+v"(fa[i >> 1][i & 1].y), "+v"(fa[i >> 1][i & 1].z), "+v"(fa[i >> 1][i & 1].w)); }
#pragma unroll
        for (int i = 0; i < 6; ++i) { fb[i] = make_uint4(lane, i, 1, 0); asm volatile("" : "+v"(fb[i].x), "+v"(fb[i].y), "+v"(fb[i].z), "+v"(fb[i].w)); }
    }
#endif
    auto read_frags = [&](int step) {
#ifdef TEDSPAD_W3P_ABLATE
        if (TEDSPAD_W3P_ABLATE & 2) return;
#endif
        const unsigned char *Y = smem + (step % S) * STG;
        const unsigned char *X = Y + WG_SUB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const uint2 lo = tr(Y + ks * 16 * 128 + offa[a][0]), hi = tr(Y + ks * 16 * 128 + offa[a][1]);
                fa[ks][a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            uint2 v[2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int rb = hr * 18 + rowb[rd];
                v[rd] = tr(X + rb * 128 + ((((cb >> 3) ^ wg_swz(rb))) << 4) + (cb & 7) * 2);
            }
            fb[hr] = make_uint4(v[0].x, v[0].y, v[1].x, v[1].y);
        }
    };
    auto compute = [&]() {
#ifdef TEDSPAD_W3P_ABLATE
        if (TEDSPAD_W3P_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(fa[i >> 1][i & 1].x), "v"(fa[i >> 1][i & 1].y), "v"(fa[i >> 1][i & 1].z), "v"(fa[i >> 1][i & 1].w));
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(fb[i].x), "v"(fb[i].y), "v"(fb[i].z), "v"(fb[i].w));
            return;
        }
#endif
#pragma unroll
        for (int hr = 0; hr < 6; ++hr)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int ks = hr - d;
                if (ks < 0 || ks > 3) continue;
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[d][a] = T::mfma(fa[ks][a], fb[hr], acc[d][a]);
            }
    };
    if (LDR) {
        for (int step = 0; step < nsteps; ++step) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(step);
            compute();
        }
    } else {
        if (nsteps > 0) issue(0, 0);
        if (nsteps > 1) issue(1, 1);
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) wait_vmcnt<4>(); else wait_vmcnt<0>();      // this step's four pieces landed; the next step's may stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (step + 2 < nsteps) issue(step + 2, (step + 2) % 3);            // its slot held step - 1: every wave is past it
            read_frags(step);
            compute();
        }
    }
#undef W3_SEG

    const bool det = det_enter();
    const int l31 = lane & 31, lh = lane >> 5;
    const int Cin = p.cc_tiles * 64;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int k = (d * 3 + dw) * Cin + cc * 64 + cih * 32 + l31;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Cout) atomicAdd(p.dw + (size_t)co * p.Kpad + k, acc[d][a][r]);
            }
    }
    det_exit(det);
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
    hipStream_t s = (hipStream_t)stream;
    // 1 x 3 x 3 / stride 1 / pad 1 over whole 64-channel chunks: the three-tile kernel (frames are independent images: t folds into n)
    static const bool allow_rows = getenv("TEDSPAD_WGRAD_NO_ROWS") == nullptr;        // A/B knob
    if (allow_rows && d->kt == 1 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 1 && d->pw == 1 &&
        d->to == d->t && d->ho == d->h && d->wo == d->w && d->cin % 64 == 0 && 9 * d->cin <= kpad) {
        p.cc_tiles = d->cin / 64;
        p.co_tiles = (d->cout + 63) / 64;
        const long tiles3 = (long)p.cc_tiles * p.co_tiles;
        // workgroups to aim for: one per CU. The kernel runs on the training step's side stream: at 256 it leaves the main stream's kernels
        // the second workgroup slot of every CU (cfg3 phase 1: 36.4 ms at 1024 / 512, 35.2-36.3 at 256, 41.5 at 128 where it no longer finishes in time)
        static const long wgs3 = getenv("TEDSPAD_WGRAD3_WGS") ? atol(getenv("TEDSPAD_WGRAD3_WGS")) : 256;
        long splits3 = (wgs3 + tiles3 - 1) / tiles3;
        const long max_splits3 = (M + 511) / 512;
        if (splits3 > max_splits3) splits3 = max_splits3;
        if (splits3 < 1) splits3 = 1;
        long rows3 = (M + splits3 - 1) / splits3;
        rows3 = (rows3 + 63) / 64 * 64;
        splits3 = (M + rows3 - 1) / rows3;
        p.rows_per_split = (int)rows3;
        const dim3 grid3((unsigned)(tiles3 * splits3));
        // the patch form (4 x 16 outputs and their one 6 x 18 halo per step, three-slot ring): where the frames are wide enough that its zero columns cost less than
        // the three-tile kernel's row re-reads. TEDSPAD_WGRAD_PATCH = 0: never, 2: wherever it applies (A/B)
        static const int patch_mode = getenv("TEDSPAD_WGRAD_PATCH") ? atoi(getenv("TEDSPAD_WGRAD_PATCH")) : 1;
        const int tw3 = (d->w + 15) / 16, th3 = (d->h + 3) / 4;
        const bool patch_fits = (long)tw3 * 16 * 100 <= (long)d->w * 115 && d->w >= 14;       // <= 15 % zero columns (measured on the UNet's shapes, 384 frames: 112^2 -4 %, 56^2 +2 / -5 %, 28^2 -6 / -2 %, 14^2 -8 %, 7^2 +43 %)
        if (patch_mode == 2 || (patch_mode == 1 && patch_fits)) {
            Wgrad3pGeo g3;
            g3.tiles_h = th3; g3.tiles_w = tw3; g3.npatch = d->n * d->t * th3 * tw3;
            long sp = (wgs3 + tiles3 - 1) / tiles3;
            const long max_sp = (g3.npatch + 7) / 8;
            if (sp > max_sp) sp = max_sp;
            if (sp < 1) sp = 1;
            g3.patches_per_split = (int)((g3.npatch + sp - 1) / sp);
            sp = (g3.npatch + g3.patches_per_split - 1) / g3.patches_per_split;
            // TEDSPAD_WGRAD3P_LOADERS = 0: without the loader waves (A/B knob). Measured and not kept (profiles/r06_ab_experiments.md): 2 loaders (half the gain), 6 (= 4),
            // rings of 4 / 5 slots (= 3), the two compute groups a barrier segment apart (slower), an LDS-counter hand-over instead of the barrier (slower)
            static const bool ld3p = getenv("TEDSPAD_WGRAD3P_LOADERS") == nullptr || atoi(getenv("TEDSPAD_WGRAD3P_LOADERS")) != 0;
            const int form = ld3p ? 1 : 0;
            static thread_local int attr3p[4] = {0, 0, 0, 0};
            const bool h16 = d->dtype == TEDSPAD_F16;
            const int ti = form * 2 + (h16 ? 0 : 1);
#define W3P_FN(N) (h16 ? (const void *)conv_wgrad3p_kernel<F16, N, 3> : (const void *)conv_wgrad3p_kernel<BF16, N, 3>)
            const void *fn = form == 0 ? W3P_FN(0) : W3P_FN(4);
#undef W3P_FN
            const int ring = 3;
            if (!attr3p[ti]) {
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    set_error("tedspad_conv_wgrad: cannot raise the dynamic LDS limit");
                    return TEDSPAD_ELAUNCH;
                }
                attr3p[ti] = 1;
            }
            const dim3 gridp((unsigned)(tiles3 * sp));
            const int nthr = form == 0 ? 384 : 640;
            void *kargs[2] = {(void *)&p, (void *)&g3};
            if (hipLaunchKernel(fn, gridp, dim3(nthr), kargs, ring * (form == 0 ? W3P_STAGE0 : W3P_STAGE), s) != hipSuccess) {
                set_error("tedspad_conv_wgrad: launch failed");
                return TEDSPAD_ELAUNCH;
            }
            return check_launch("tedspad_conv_wgrad");
        }
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL((conv_wgrad3_kernel<F16>), grid3, dim3(384), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad3_kernel<BF16>), grid3, dim3(384), 0, s, p);
        return check_launch("tedspad_conv_wgrad");
    }
    // split the pixels so the grid covers the chip ~4x, each split a whole number of 64-pixel steps (>= 8 steps)
    const long tiles = (long)p.k_tiles * p.co_tiles;
    long splits = (1024 + tiles - 1) / tiles;              // (512 / 256 measured the same in the training step)
    const long max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long rows = (M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    splits = (M + rows - 1) / rows;
    p.rows_per_split = (int)rows;
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
namespace tedspad {
int32_t det_ctl_wgrad(int op, int on) { return det_ctl(op, on); }
}  // namespace tedspad
