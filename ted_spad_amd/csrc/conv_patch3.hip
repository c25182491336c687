// Persistent PING-PONG two-patch convolution for gfx950 (tile_cfg 40): the stride-1 'same' 1 x 3 x 3 convs with 32 < cout <= 64 of the anonymizers' wide levels
// (unet_parts.py:8-25 DoubleConv at 112 x 112 / 224 x 224; smp's DecoderBlock convs of the default unet++ `fa`, model_loaders.py:17-30, which
// dali_extraction.py:171-173 runs in front of every clip).
//
// What the one-idea-at-a-time kernels of rounds 4 / 5 (tiles 32, 38, the dropped resident-weight tile) -- and the first, lockstep form of this kernel (8 waves
// between the same barriers: 628 us on 400 x 112 x 112 x 64 -> 64, 427 us with the MFMAs ablated, i.e. the matrix time purely ADDITIVE) -- showed: nothing that is
// not an MFMA overlaps the MFMAs while the two waves of a SIMD run the same program in the same phase. So here, all together:
//   * ONE 8-wave workgroup per CU, persistent over a contiguous run of tiles; a tile = two consecutive 16 x 16 patches x 64 output channels. Waves 0-3 (group 0)
//     own patch 0, waves 4-7 (group 1) patch 1; a wave owns 4 rows x 64 channels (16 accumulator quads of v_mfma_f32_16x16x32).
//   * PING-PONG (the structure of conv_p8.hip): a PHASE is one tap (dw) of one kernel row (dh) of one 32-channel half chunk: a LOAD segment (the tap's 8 fragment
//     reads, at most one or two LDS-DMA instructions, one piece of the previous tile's epilogue) and a COMPUTE segment (16 MFMAs), each ended by a raw
//     s_barrier; group 1 runs one segment behind group 0, so while one wave of a SIMD multiplies, its partner reads, issues DMA and converts / stores.
//   * the halo of a half chunk lives in a ring of 2 (weights resident) or 3 (weights streamed) buffers of 41 KB and is fetched one / two half chunks ahead, ONE
//     DMA instruction per wave and phase (an LDS-DMA instruction holds its wave for its whole queueing time: a burst of 41 behind one barrier cost 340 us of the
//     2 377 us the lockstep form took on 320 -> 64);
//   * weights: cin <= 64 (at most six stages of [3 taps][64 co][32 k]): fetched ONCE per workgroup, resident for all of its tiles; otherwise a ring of nine
//     4 KB tap units, unit n + 8 issued in phase n by waves 0-3 (one instruction each);
//   * the epilogue straight from the accumulators, piece by piece in the load segments of the NEXT tile's first eight phases (two accumulator sets take turns):
//     v_permlane16_swap turns the 16 x 16 x 32 accumulator quads (4 channels = 8 bytes per lane) into 16-byte pieces, 64 contiguous bytes per pixel and store;
//   * the training extras from registers as well: batch statistics are summed per lane over ALL tiles of the workgroup and flushed once (2 atomic
//     instructions per wave instead of 128 atomics per patch), fp32 output, ReLU-backward mask, residual.
// K is walked (half chunk, dh, dw) exactly as tile 38 does: the sums are bit-identical to tile 38's.
//
// Ordering argument (segments are the barrier-delimited intervals; group 0 runs LOAD of phase n in segment 2n and COMPUTE in 2n + 1, group 1 one segment later; every
// LOAD segment ends with lgkmcnt(0) in front of its barrier, every COMPUTE segment with the counted vmcnt wait in front of its barrier):
//   RAW  bytes read in phase n were waited for -- by the wave that asked for them -- in phase <= n - 2, i.e. before the barrier that ends segment 2 (n - 2) + 2 = 2n - 2;
//        the earliest read of them is issued in segment 2n;
//   WAR  the fragment reads of phase k have returned by the end of segment 2k + 1; a DMA into their bytes is issued in phase >= k + 1, i.e. in segment >= 2k + 2.
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16p3;
__device__ uint4 g_sink_p3[64];          // where the masked-off lanes of the epilogue stores go (never read): every piece issues the same number of stores

constexpr int P3_S = 16;                                                            // patch side
constexpr int P3_WH = 18, P3_NP = P3_WH * P3_WH, P3_PSLOTS = P3_NP * 4;             // 1296 16-byte slots per patch and half chunk
constexpr int P3_HALO = (2 * P3_PSLOTS + 63) / 64 * 64 * 16;                        // 41 984: both patches' halos of a half chunk (41 wave instructions; the last one's upper half is padding)
constexpr int P3_WST = 3 * 64 * 64;                                                 // a weight stage [3 dw][64 co][32 k]: 12 288 bytes; a tap unit is a third of it
constexpr int P3_WTAP = 64 * 64;
constexpr int P3_LDS_RES = 2 * P3_HALO + 6 * P3_WST + 512;                          // 158 208: two halo buffers, six resident stages, scale / shift
constexpr int P3_LDS_STR = 3 * P3_HALO + 9 * P3_WTAP + 512;                         // 163 328: three halo buffers, nine tap units, scale / shift
constexpr int P3_NT = 512;
static_assert(P3_LDS_RES <= 160 * 1024 && P3_LDS_STR <= 160 * 1024, "one workgroup per CU");

struct Patch3Geo {
    int tiles_h, tiles_w, npatch, ntiles, nhc, dbg;
};

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {      // n: wave-uniform; a smaller count than asked for is always safe
    switch (n < 0 ? 0 : n) {
#define P3_W(N) case N: wait_vmcnt<N>(); break;
        P3_W(0) P3_W(1) P3_W(2) P3_W(3) P3_W(4) P3_W(5) P3_W(6) P3_W(7) P3_W(8) P3_W(9) P3_W(10) P3_W(11) P3_W(12) P3_W(13) P3_W(14) P3_W(15)
        P3_W(16) P3_W(17) P3_W(18) P3_W(19) P3_W(20) P3_W(21) P3_W(22) P3_W(23) P3_W(24) P3_W(25) P3_W(26) P3_W(27) P3_W(28) P3_W(29) P3_W(30)
#undef P3_W
        default: wait_vmcnt<31>(); break;
    }
}

__device__ __forceinline__ void swap16(uint32_t &x, uint32_t &y) {      // rows of 16 lanes: x of the odd rows <-> y of the even rows (its own inverse)
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0]; y = r[1];
}

template <typename T>
__device__ __forceinline__ uint32_t pack2_lim(float a, float b, float lim) {
    return (uint32_t)T::from_f32_lim(a, lim) | ((uint32_t)T::from_f32_lim(b, lim) << 16);
}

// a tile's two patches: first row / column, frame, presence
struct P3Tile {
    int pf[2], ph0[2], pw0[2];
    bool pon[2];
};

__device__ __forceinline__ P3Tile p3_decode(int tile, const Patch3Geo &g) {
    P3Tile t;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int pi = 2 * tile + q;
        t.pon[q] = pi < g.npatch;
        if (!t.pon[q]) pi = 2 * tile;
        const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
        t.pw0[q] = tw * P3_S; t.ph0[q] = (t2 % g.tiles_h) * P3_S; t.pf[q] = t2 / g.tiles_h;
    }
    return t;
}

#define P3_SEG_END()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0)

template <typename T, bool SRC, bool STATS, bool RES>
__global__ __launch_bounds__(P3_NT) void conv_patch3_kernel(const ConvKP p, const Patch3Geo g, const PatchSrc gs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, G);
    const int t_begin = (int)((long)wg * g.ntiles / G), t_end = (int)((long)(wg + 1) * g.ntiles / G);
    if (t_begin >= t_end) return;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16p3);
    const int nhc = g.nhc;
    // RES: at most six weight stages (nhc <= 2): fetched once, resident
    constexpr int NB = RES ? 2 : 3;                        // halo buffers
    constexpr int WBASE = NB * P3_HALO;
    constexpr int SCB = RES ? WBASE + 6 * P3_WST : WBASE + 9 * P3_WTAP;     // scale[64] | shift[64]
    const int ntl = t_end - t_begin;
    const int HS_total = ntl * nhc;                        // half chunks of this workgroup
    const int NP_total = HS_total * 9;                     // phases (= weight tap units) of this workgroup
    const int grp = wave >> 2, wr = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const int nst = (p.y ? 1 : 0) + (p.y32 ? 2 : 0);       // stores per epilogue piece
    const int dbg = g.dbg;                                 // timing ablations (wrong results; TEDSPAD_P3_ABLATE): 4 no epilogue pieces in the loop, 8 no halo DMA in the loop, 16 no weight DMA in the loop (folded into the existence flags: no cost when off)

    // ---- halo pieces: slot s of a half chunk -> patch s / 1296, position (s % 1296) >> 2, LDS piece s & 3. The geometry of a slot is recomputed per piece (~25
    // vector instructions in a load segment) instead of living in registers: the two accumulator sets need them ------------------------------------------------
    P3Tile ht = p3_decode(t_begin, g);                     // the tile whose halos are being issued
    auto halo_piece = [&](int s0, int hcx, int buf) __attribute__((always_inline))  {      // one wave instruction: slots s0 .. s0 + 63; s0 is wave-uniform
        int lz = lane;
        asm volatile("" : "+v"(lz));                       // the slot geometry is NOT loop-invariant for the register allocator: ~25 instructions per piece instead of registers
        const int s = s0 + lz;
        const int q = s >= P3_PSLOTS ? 1 : 0, r = s - q * P3_PSLOTS;
        const int pos = r >> 2, hr = (pos * 3641) >> 16, hcl = pos - hr * P3_WH;         // pos / 18 for pos < 1296
        const int c8 = ((r & 3) ^ ((pos >> 1) & 3)) << 3;
        const int ih = (q ? ht.ph0[1] : ht.ph0[0]) - 1 + hr, iw = (q ? ht.pw0[1] : ht.pw0[0]) - 1 + hcl;
        const int pf = q ? ht.pf[1] : ht.pf[0];
        const bool ok = s < 2 * P3_PSLOTS && (q ? ht.pon[1] : ht.pon[0]) && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        const uint16_t *sp = p.x + hcx * 32;
        long sl = p.ldx;
        int pix = (pf * p.Hi + ih) * p.Wi + iw;
        if (SRC) {
            const int ck = hcx >> 1;
            sp = gs.ptr[ck] + (hcx & 1) * 32; sl = gs.ld[ck];
            if ((gs.up >> ck) & 1) pix = (pf * (p.Hi >> 1) + (ih >> 1)) * (p.Wi >> 1) + (iw >> 1);
        }
        lds_dma16(ok ? sp + pix * sl + c8 : zero, lds0 + buf * P3_HALO + s0 * 16);
    };
    // all eight waves: piece i = slots i * 512 + tid (i = 0..4; i = 5: wave 0 alone); waves 4-7 alone: piece i = slots i * 256 + (tid - 256) (i = 0..9; i = 10: wave 4 alone)
    auto halo_piece8 = [&](int i, int hcx, int buf) __attribute__((always_inline))  { halo_piece(i * 512 + wave * 64, hcx, buf); };
    auto halo_piece4 = [&](int i, int hcx, int buf) __attribute__((always_inline))  { halo_piece(i * 256 + (wave - 4) * 64, hcx, buf); };

    // ---- weights: a tap unit [64 co][32 k], piece c of row co at c ^ ((co >> 1) & 3); a stage = the three taps dw of a kernel row -------------------------------
    const int wco = (wave & 3) * 16 + (lane >> 2);                                   // tap units are moved by four waves, 16 rows each
    const int wof4 = wco * p.Kpad + (((lane & 3) ^ ((wco >> 1) & 3)) << 3);
    auto issue_unit = [&](int hcx, int tap, int upos) __attribute__((always_inline))  {        // waves 0-3 (streamed weights): tap (dh * 3 + dw) of half chunk hcx -> ring position upos
        lds_dma16(p.w + tap * p.cin + hcx * 32 + wof4, lds0 + WBASE + upos * P3_WTAP + (wave & 3) * 1024);
    };

    // ---- MFMA roles --------------------------------------------------------------------------------------------------------------------------------------------
    const unsigned wrd = (unsigned)(l15 * 64 + ((kg ^ ((l15 >> 1) & 3)) << 4));       // this lane's piece of weight row (16 a + l15) of a tap unit
    const unsigned hq = (unsigned)(grp * (P3_PSLOTS * 16));
    float s1[STATS ? 4 : 1][4], s2[STATS ? 4 : 1][4];       // batch statistics of this lane's channels (16 a + 4 kg + j) over every tile of the workgroup
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }
    }
    long sgrp_cur = -1;
    auto flush_stats = [&]() __attribute__((always_inline))  {                             // wave-local: sums over the 16 pixel lanes, then ONE atomic instruction per statistic
        if constexpr (STATS) {
            float o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = s1[a][j], y = s2[a][j];
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) { x += __shfl_xor(x, d, 64); y += __shfl_xor(y, d, 64); }
                    if (l15 == a * 4 + j) { o1 = x; o2 = y; }
                    s1[a][j] = 0.f; s2[a][j] = 0.f;
                }
            const int ch = (l15 >> 2) * 16 + kg * 4 + (l15 & 3);
            if (sgrp_cur >= 0 && ch < p.Cout) {
                float *so = p.stats + sgrp_cur * 2 * p.stats_ld;
                atomicAdd(so + ch, o1);
                atomicAdd(so + p.stats_ld + ch, o2);
            }
        }
    };

    // ---- the epilogue of a finished tile, in eight pieces (accumulator row r = k >> 1, channel groups 2 pp, 2 pp + 1 with pp = k & 1) ----------------------------------
    int ep_pf = 0, ep_ho0 = 0, ep_wo0 = 0;                 // this wave's patch of the tile being drained: frame, first row of the wave, first column
    bool ep_pon = false;
    const bool ep_fast = !p.res && !p.mask && !p.y32 && p.y;
    const float ep_lo = p.relu ? 0.f : -p.sat;
    auto ep_set = [&](int tile) __attribute__((always_inline))  {
        int pi = 2 * tile + grp;
        ep_pon = pi < g.npatch;
        if (!ep_pon) pi = 2 * tile;
        const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
        ep_wo0 = tw * P3_S; ep_ho0 = (t2 % g.tiles_h) * P3_S + 4 * wr; ep_pf = t2 / g.tiles_h;
        if constexpr (STATS) {
            const long mfirst = ((long)ep_pf * p.Ho + (ep_ho0 - 4 * wr)) * p.Wo + ep_wo0;       // a patch lies inside one frame: inside one statistics group
            const long grpi = p.stats_rows ? mfirst / p.stats_rows : 0;
            if (grpi != sgrp_cur) {
                flush_stats();
                sgrp_cur = grpi;
            }
        }
    };
    auto ep_piece = [&](f32x4 (&acc)[4][4], const int k) {
        const int r = k >> 1, pp = k & 1;
        int lz = lane;
        asm volatile("" : "+v"(lz));                       // a piece's addresses are invariant over the tile's half chunks: computed here, not hoisted to the tile's top and kept (spilled) in registers
        const int l15 = lz & 15, kg = lz >> 4;
        const int ho = ep_ho0 + r, wo = ep_wo0 + l15;
        const bool valid = ep_pon && ho < p.Ho && wo < p.Wo;
        const size_t m = ((size_t)ep_pf * p.Ho + ho) * p.Wo + wo;
        const int poff = (2 * pp + (kg & 1)) * 16 + (kg >> 1) * 8;       // this lane's 16-byte piece (8 channels) of the pixel's row, after the swap
        const bool on = valid && poff < p.Cout;                          // 8-channel pieces beyond cout are neither read nor stored
        if (T::kDtype == TEDSPAD_F16 && ep_fast) {      // the inference epilogue (scale / shift, ReLU, saturation, 16-bit store): ~30 vector instructions per piece
            uint32_t pk[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int a = 2 * pp + h;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(dsm + SCB + (a * 16 + kg * 4) * 4);
                const f32x4 sf = *reinterpret_cast<const f32x4 *>(dsm + SCB + 256 + (a * 16 + kg * 4) * 4);
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = acc[r][a][j] * sc[j] + sf[j];
                if constexpr (STATS) {
                    if (valid) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s1[a][j] += w[j]; s2[a][j] += w[j] * w[j]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_fmed3f(w[j], ep_lo, p.sat);      // = min(max(relu(w), -sat), sat) for every w but a NaN
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[h][0]) : "v"(w[0]), "v"(w[1]));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[h][1]) : "v"(w[2]), "v"(w[3]));
            }
            swap16(pk[0][0], pk[1][0]); swap16(pk[0][1], pk[1][1]);
            uint4 *dst = on ? reinterpret_cast<uint4 *>(p.y + m * p.ldy + poff) : g_sink_p3 + lane;
            *dst = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
            return;
        }
        float v[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int a = 2 * pp + h;
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(dsm + SCB + (a * 16 + kg * 4) * 4);
            const f32x4 sf = *reinterpret_cast<const f32x4 *>(dsm + SCB + 256 + (a * 16 + kg * 4) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[h][j] = acc[r][a][j] * sc[j] + sf[j];
            if constexpr (STATS) {
                if (valid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s1[a][j] += v[h][j]; s2[a][j] += v[h][j] * v[h][j]; }
                }
            }
        }
        if (p.res) {
            uint4 rv = make_uint4(0u, 0u, 0u, 0u);
            if (on) rv = *reinterpret_cast<const uint4 *>(p.res + m * p.ldres + poff);
            swap16(rv.x, rv.z); swap16(rv.y, rv.w);                      // -> (x, y): this lane's 4 channels of group 2 pp, (z, w): of group 2 pp + 1
            const uint32_t rw[2][2] = {{rv.x, rv.y}, {rv.z, rv.w}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h][0] += T::to_f32((uint16_t)(rw[h][0] & 0xffffu)); v[h][1] += T::to_f32((uint16_t)(rw[h][0] >> 16));
                v[h][2] += T::to_f32((uint16_t)(rw[h][1] & 0xffffu)); v[h][3] += T::to_f32((uint16_t)(rw[h][1] >> 16));
            }
        }
        if (p.relu) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[h][j] = __builtin_fmaxf(v[h][j], 0.f);
        }
        if (p.mask) {           // ReLU backward fused into the data gradient that produces d(input)
            uint4 mv = make_uint4(0u, 0u, 0u, 0u);
            if (on) mv = *reinterpret_cast<const uint4 *>(p.mask + m * p.ldmask + poff);
            swap16(mv.x, mv.z); swap16(mv.y, mv.w);
            const uint32_t mw[2][2] = {{mv.x, mv.y}, {mv.z, mv.w}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h][0] = T::to_f32((uint16_t)(mw[h][0] & 0xffffu)) > 0.f ? v[h][0] : 0.f; v[h][1] = T::to_f32((uint16_t)(mw[h][0] >> 16)) > 0.f ? v[h][1] : 0.f;
                v[h][2] = T::to_f32((uint16_t)(mw[h][1] & 0xffffu)) > 0.f ? v[h][2] : 0.f; v[h][3] = T::to_f32((uint16_t)(mw[h][1] >> 16)) > 0.f ? v[h][3] : 0.f;
            }
        }
        if (p.y) {
            uint32_t x0 = pack2_lim<T>(v[0][0], v[0][1], p.sat), x1 = pack2_lim<T>(v[0][2], v[0][3], p.sat);
            uint32_t y0 = pack2_lim<T>(v[1][0], v[1][1], p.sat), y1 = pack2_lim<T>(v[1][2], v[1][3], p.sat);
            swap16(x0, y0); swap16(x1, y1);
            uint4 *dst = on ? reinterpret_cast<uint4 *>(p.y + m * p.ldy + poff) : g_sink_p3 + lane;
            *dst = make_uint4(x0, x1, y0, y1);
        }
        if (p.y32) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int nch = (2 * pp + h) * 16 + kg * 4;
                f32x4 *dst = (valid && nch < p.Cout) ? reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch) : reinterpret_cast<f32x4 *>(g_sink_p3 + lane);
                *dst = f32x4{v[h][0], v[h][1], v[h][2], v[h][3]};
            }
        }
    };

    // ---- prologue: scale / shift, the first halo(s), the weights -----------------------------------------------------------------------------------------------
    if (tid < 64) {
        *reinterpret_cast<float *>(dsm + SCB + tid * 4) = p.scale[tid];
        *reinterpret_cast<float *>(dsm + SCB + 256 + tid * 4) = p.shift[tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 6; ++i)
        if (i < 5 || wave == 0) halo_piece8(i, 0, 0);
    if (NB == 3 && HS_total > 1) {                          // nhc >= 3 here: half chunk 1 of the first tile
        for (int i = 0; i < 6; ++i)
            if (i < 5 || wave == 0) halo_piece8(i, 1, 1);
    }
    if (RES) {
        for (int st = 0; st < nhc * 3; ++st)
            for (int dw = 0; dw < 3; ++dw)
                if (((st * 3 + dw) & 1) == grp) {           // every group moves every other tap unit
                    const int hcx = st / 3, dh = st - hcx * 3;
                    lds_dma16(p.w + (dh * 3 + dw) * p.cin + hcx * 32 + wof4, lds0 + WBASE + st * P3_WST + dw * P3_WTAP + (wave & 3) * 1024);
                }
    } else if (wave < 4) {
        for (int u = 0; u < 8; ++u) issue_unit(u / 9, u % 9, u);      // units 0..7 (nhc >= 3: NP_total >= 27)
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);

    // ---- cursors (wave-uniform) --------------------------------------------------------------------------------------------------------------------------------
    int n = 0;                                             // phase of this workgroup (= weight tap unit read in it)
    int hb = 0;                                            // halo buffer of the half chunk being multiplied
    int hi_hs = NB - 1, hi_tile = t_begin, hi_hc = NB - 1, hi_buf = NB - 1;      // the half chunk whose halo is being issued: NB - 1 ahead
    while (hi_hc >= nhc) { hi_hc -= nhc; ++hi_tile; }
    int wu_hc = 0, wu_tap = 8, wu_pos = 8;                 // streamed weights: the unit issued in the current phase is n + 8
    int rd_pos = 0;                                        // ring position of the unit read in the current phase

    uint4 fw[4], fa[4];
    const bool no_h = (dbg & 8) != 0;
    const int NPI = (dbg & 16) ? 0 : NP_total;             // weight units that get issued in the loop
    // hc_pieces: the epilogue pieces of this half chunk's phases (the tile before is being drained): RES with two half chunks: 0..3 in phases 5..8 of the first, 4..7 in phases
    // 5..8 of the second (the phases that issue no halo DMA); otherwise 0..7 in phases 1..8 of the first half chunk.
    auto half_chunk = [&](f32x4 (&accC)[4][4], f32x4 (&accD)[4][4], const int hc, const bool draining) {
        const bool h_exists = hi_hs < HS_total && !no_h;   // is there a half chunk NB - 1 ahead whose halo this half chunk's phases issue?
        if (h_exists && hi_hc == 0 && hi_tile != t_begin) ht = p3_decode(hi_tile, g);      // its tile's patches (scalar arithmetic; no piece is issued before it)
        const bool split = RES && nhc == 2;
        const bool pieces_here = draining && (split || hc == 0);
        // ---- this half chunk's fragment addresses: Vw[dwi][e] + immediate. Position (4 wr + r + dh) * 18 + dw + l15 of the halo has its k-group kg at piece kg ^ t with
        // t = ((position >> 1) & 3) = (dh + r + ((dw + l15) >> 1)) & 3: dw = 0 / 2 share u0 = (l15 >> 1) & 3 (dw = 2: one more), dw = 1 has u1 = ((l15 + 1) >> 1) & 3 ----
        unsigned Vw[2][4];
        {
            int lz = lane;
            asm volatile("" : "+v"(lz));
            const unsigned l15z = lz & 15, kgz = lz >> 4;
            const unsigned B00 = (unsigned)(hb * P3_HALO) + hq + (72u * wr + l15z) * 64u + kgz * 16u;
            const unsigned u0 = (l15z >> 1) & 3u, u1 = ((l15z + 1u) >> 1) & 3u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { Vw[0][e] = B00 ^ (((u0 + e) & 3u) << 4); Vw[1][e] = B00 ^ (((u1 + e) & 3u) << 4); }
        }
        const unsigned wres = (unsigned)(WBASE + hc * 3 * P3_WST) + wrd;      // RES: this half chunk's three stages
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int dh = j / 3, dw = j - dh * 3;
            // ================= LOAD segment =================
            const unsigned wb = RES ? wres + (unsigned)(dh * P3_WST + dw * P3_WTAP) : (unsigned)(WBASE + rd_pos * P3_WTAP) + wrd;
#pragma unroll
            for (int a = 0; a < 4; ++a) fw[a] = *reinterpret_cast<const uint4 *>(dsm + wb + a * 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fa[r] = *reinterpret_cast<const uint4 *>(dsm + Vw[dw == 1 ? 1 : 0][(dh + r + (dw == 2 ? 1 : 0)) & 3] + (unsigned)(1152 * (r + dh) + 64 * dw));
            if (RES) {                                     // halo one half chunk ahead, all waves: pieces 0..4 in phases 0..4 (wave 0: its sixth beside the first)
                if (h_exists) {
                    if (j == 0 && wave == 0) halo_piece8(5, hi_hc, hi_buf);
                    if (j < 5) halo_piece8(j, hi_hc, hi_buf);
                }
            } else if (wave < 4) {                         // weights: unit n + 8
                if (n + 8 < NPI) issue_unit(wu_hc, wu_tap, wu_pos);
            } else if (h_exists) {                          // halo two half chunks ahead, waves 4-7: one piece per phase (two in phase 0, wave 4: two in phase 1)
                if (j == 0) halo_piece4(9, hi_hc, hi_buf);
                if (j == 1 && wave == 4) halo_piece4(10, hi_hc, hi_buf);
                halo_piece4(j, hi_hc, hi_buf);
            }
            if (pieces_here) {
                if (split) { if (j >= 5) { if (hc == 0) ep_piece(accD, j - 5); else ep_piece(accD, j - 1); } }      // (the piece index must be a constant: the accumulators are registers)
                else if (j >= 1) ep_piece(accD, j - 1);
            }
            if (j == 0 && hc == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) accC[r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P3_SEG_END();
            // ================= COMPUTE segment =================
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) accC[r][a] = T::mfma16(fw[a], fa[r], accC[r][a]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- counted waits. The epilogue's stores share the queue: those issued behind the oldest DMA that may stay in flight are allowed on top (every piece issues
            // exactly nst stores: masked lanes write to the sink); counting fewer than were issued only waits longer ----
            if (RES) {
                // the halo issued in phases 0..4 of this half chunk is read from phase 9 on: landed by the wait of phase 7; behind its last piece: the stores of phases 5..7 (4..7 when the
                // eight pieces run in phases 1..8)
                if (j == 7 && h_exists) wait_vmcnt_dyn(pieces_here ? (split ? 3 : 4) * nst : 0);
            } else if (wave < 4) {
                // every third phase: units up to n + 4 landed (they are read up to phase n + 4 + 2, the next wait comes in phase n + 3); n + 5 .. n + 8 may stay in flight, and the stores
                // of the four phases that issued them
                if (dw == 2) {
                    const int last = n + 8 < NPI ? n + 8 : NPI - 1;
                    int d = last - (n + 4);
                    d = d < 0 ? 0 : d;
                    const int ns = pieces_here ? (j < 4 ? j : 4) : 0;
                    wait_vmcnt_dyn(d == 4 ? 4 + ns * nst : d);
                }
            } else if (j == 7) {
                // the halo issued during the half chunk before this one is read from phase 9 on: behind its last piece lie this half chunk's pieces (2 + 1 + 6 in phases 0..7; wave 4 one
                // more, which it waits for too) and the stores of phases 1..7
                wait_vmcnt_dyn(h_exists ? 9 + (pieces_here ? 7 * nst : 0) : (pieces_here ? 7 * nst : 0));
            }
            P3_SEG_END();
            // ---- cursors ----
            ++n;
            if (!RES) {
                if (++rd_pos == 9) rd_pos = 0;
                if (++wu_pos == 9) wu_pos = 0;
                if (++wu_tap == 9) { wu_tap = 0; if (++wu_hc == nhc) wu_hc = 0; }
            }
        }
        if (++hb == NB) hb = 0;
        ++hi_hs;
        if (++hi_buf == NB) hi_buf = 0;
        if (++hi_hc == nhc) { hi_hc = 0; ++hi_tile; }
    };

    f32x4 accA[4][4], accB[4][4];
    auto tile_body = [&](f32x4 (&accC)[4][4], f32x4 (&accD)[4][4], const int tl) {
        const bool draining = tl > 0 && !(dbg & 4);
        if (draining) ep_set(t_begin + tl - 1);
        for (int hc = 0; hc < nhc; ++hc) half_chunk(accC, accD, hc, draining);
    };
    for (int tl = 0; tl < ntl; tl += 2) {
        tile_body(accA, accB, tl);
        if (tl + 1 < ntl) tile_body(accB, accA, tl + 1);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();             // pairs with the last in-loop barrier of waves 4-7
    asm volatile("" ::: "memory");
    // ---- the last tile's epilogue ------------------------------------------------------------------------------------------------------------------------------------
    ep_set(t_end - 1);
    if (!(dbg & 4)) {
        if (ntl & 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ep_piece(accA, k);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) ep_piece(accB, k);
        }
    }
    flush_stats();
}

int g_p3_cus = 0;
bool g_p3_det = false;       // deterministic mode: the persistent kernel's statistics flush is not gated (det_gate.h): it declines statistics then

template <typename T, bool SRC, bool STATS, bool RES>
int32_t launch_patch3_t(const ConvKP &p, int frames, int cin, hipStream_t s, const PatchSrc *src) {
    Patch3Geo g;
    g.tiles_h = (p.Ho + P3_S - 1) / P3_S; g.tiles_w = (p.Wo + P3_S - 1) / P3_S;
    g.npatch = frames * g.tiles_h * g.tiles_w; g.ntiles = (g.npatch + 1) / 2; g.nhc = cin / 32;
    const char *abl = getenv("TEDSPAD_P3_ABLATE");
    g.dbg = abl ? atoi(abl) : 0;
    if (!g_p3_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_p3_cus = n;
    }
    const char *nwg_s = getenv("TEDSPAD_P3_NWG");           // tests / experiments: workgroups of the persistent grid (read per launch)
    const int nwg_env = nwg_s ? atoi(nwg_s) : 0;
    int grid = nwg_env > 0 ? nwg_env : g_p3_cus;
    if (grid > g.ntiles) grid = g.ntiles;
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_patch3_kernel<T, SRC, STATS, RES>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    PatchSrc gsrc{};
    if (SRC) gsrc = *src;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(P3_NT), RES ? P3_LDS_RES : P3_LDS_STR, s, p, g, gsrc);
    return check_launch("tedspad_conv_fwd(persistent two-patch halo)");
}

}  // namespace

void patch3_set_det(int on) { g_p3_det = on != 0; }

int32_t launch_conv_patch3(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src) {
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi;
    if (cin % 32 != 0 || (src && cin % 64 != 0) || p.kt != 1 || p.kh != 3 || p.kw != 3 || p.pt != 0 || p.ph != 1 || p.pw != 1 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same ||
        p.Kpad < 9 * cin || p.ostrided || p.sigmoid || (!p.y && !p.y32) || (src && src->n != cin / 64) || (long)N * p.Ti * p.Hi * p.Wi >= (1L << 31) ||
        p.Cout > 64 || p.Cout <= 32 || (p.stats && g_p3_det)) {
        set_error("tedspad_conv_fwd: persistent two-patch config (tile_cfg 40) needs a stride-1 'same' 1 x 3 x 3 conv with cin %% 32 == 0 (gathered sources: %% 64) and 32 < cout <= 64 "
                  "(mask / stats / fp32 output allowed, no strided output map; no statistics in deterministic mode)");
        return TEDSPAD_EINVAL;
    }
    const int frames = N * p.Ti;
    const bool f16 = dtype == TEDSPAD_F16, st = p.stats != nullptr;
    if (cin <= 64) {      // weights resident (a gathered concatenation of one source is just a tensor: not instantiated)
        if (src) {
            set_error("tedspad_conv_fwd: tile_cfg 40 takes gathered sources with cin >= 128 only");
            return TEDSPAD_EINVAL;
        }
        if (st) return f16 ? launch_patch3_t<F16, false, true, true>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, true, true>(p, frames, cin, s, nullptr);
        return f16 ? launch_patch3_t<F16, false, false, true>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, false, true>(p, frames, cin, s, nullptr);
    }
    if (st) {
        if (src) return f16 ? launch_patch3_t<F16, true, true, false>(p, frames, cin, s, src) : launch_patch3_t<BF16, true, true, false>(p, frames, cin, s, src);
        return f16 ? launch_patch3_t<F16, false, true, false>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, true, false>(p, frames, cin, s, nullptr);
    }
    if (src) return f16 ? launch_patch3_t<F16, true, false, false>(p, frames, cin, s, src) : launch_patch3_t<BF16, true, false, false>(p, frames, cin, s, src);
    return f16 ? launch_patch3_t<F16, false, false, false>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, false, false>(p, frames, cin, s, nullptr);
}

}  // namespace tedspad
