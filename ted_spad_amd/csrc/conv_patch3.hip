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
//   * LOADER WAVES for the streamed-weight inference variants (template flag LDR): waves 8..11 issue every LDS-DMA instruction and do the counted waits, on the same phase
//     schedule; the compute waves' load segments are fragment reads only (400 x 112^2, 128 / 192 / 320 -> 64 channels: 945 / 1 313 / 2 020 -> 850 / 1 200 / 1 847 us).
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
#ifdef TEDSPAD_P3_STAMPS
__device__ unsigned long long *g_dbg_p3;   // stamp build (scripts/p3_cycles.py): [wave][64] s_memtime stamps of workgroup 0's third tile
#define P3_STAMP(k) do { if (stamp_on && lane == 0) g_dbg_p3[wave * 64 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define P3_STAMP(k) do { } while (0)
#endif
__device__ uint4 g_sink_p3[64];          // where the masked-off lanes of the epilogue stores go (never read): every piece issues the same number of stores

constexpr int P3_S = 16;                                                            // patch side
constexpr int P3_WH = 18, P3_NP = P3_WH * P3_WH, P3_PSLOTS = P3_NP * 4;             // 1296 16-byte slots per patch and half chunk
constexpr int P3_HG = 21 * 1024;                                                    // a group's halo of a half chunk: 21 wave instructions (the last one: 16 lanes of data)
constexpr int P3_HALO = 2 * P3_HG;                                                  // 43 008: both patches
constexpr int P3_WST = 3 * 64 * 64;                                                 // a weight stage [3 dw][64 co][32 k]: 12 288 bytes
constexpr int P3_WTAP = 64 * 64;
constexpr int P3_NWS = 6;                                                           // weight stages in LDS: all of them when cin <= 64, otherwise a ring filled five stages ahead
constexpr int P3_WBASE = 2 * P3_HALO;
constexpr int P3_SCB = P3_WBASE + P3_NWS * P3_WST;                                  // scale[64] | shift[64]
constexpr int P3_LDS = P3_SCB + 512;                                                // 160 256
constexpr int P3_NT = 512, P3_NTL = 768;                                            // threads: 8 compute waves (two groups of four); LDR: + 4 loader waves
static_assert(P3_LDS <= 160 * 1024, "one workgroup per CU");

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

// GEN: the general epilogue (residual / ReLU-backward mask / fp32 output / bf16); otherwise the inference epilogue (F16: scale / shift, ReLU, saturation, 16-bit store)
// LDR (the streamed-weight inference variants): four more waves (8..11, one per SIMD beside its two compute waves) issue every DMA and do the counted waits; the
// compute waves issue no vector-memory instruction but the epilogue's stores. A vector-memory instruction holds its wave for its queueing time (200-400 cycles behind the CU's
// other requests: profiles/r06_p3_stamps.md); in a compute wave's load segment that is time its SIMD partner's MFMAs wait for at the next barrier. 168 registers per wave.
template <typename T, bool SRC, bool STATS, bool RES, bool GEN, bool LDR>
__global__ __launch_bounds__(LDR ? P3_NTL : P3_NT) void conv_patch3_kernel(const ConvKP p, const Patch3Geo g, const PatchSrc gs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, G);
    const int t_begin = (int)((long)wg * g.ntiles / G), t_end = (int)((long)(wg + 1) * g.ntiles / G);
    if (t_begin >= t_end) return;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16p3);
    const int nhc = g.nhc;                                 // RES: nhc <= 2 (at most six weight stages: fetched once, resident)
    const int S = nhc * 3;                                 // stages (= phases) per tile
    const int ntl = t_end - t_begin;
    const int total = ntl * S;                             // phases of this workgroup
    const int grp = wave >> 2, wr = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const int nst = (p.y ? 1 : 0) + (p.y32 ? 2 : 0);       // stores per epilogue piece
    const int dbg = g.dbg;                                 // timing ablations (wrong results; TEDSPAD_P3_ABLATE): 4 no epilogue in the loop, 8 no halo DMA in the loop, 16 no weight DMA in the loop

    // ---- halo: slot s of a patch and half chunk -> position s >> 2 (halo row, column), LDS piece s & 3; four waves move a patch's 1296 slots: piece i = slot i * 256 + (tid & 255),
    // i = 0..4, and i = 5 for the first of the four (16 lanes of data). Without loader waves a group fetches ITS OWN patch's halo (what a group reads it has waited for itself:
    // one phase between wait and read is enough); the loader waves fetch both patches' (entries 0..5: patch 0, 6..11: patch 1) ----
    constexpr int NHP = LDR ? 12 : 6;
    const bool loader = LDR && wave >= 8;
    const int w4 = wave & 3;                               // this wave among the four that move a unit (= wr for a compute wave)
    const int lt = tid & 255;
    int hgeo[NHP];                                         // (c8 << 16) | (halo row << 8) | halo column, or -1: no such slot
#pragma unroll
    for (int ii = 0; ii < NHP; ++ii) {
        const int i = ii % 6;
        const int sl = i * 256 + lt;
        const int pos = sl >> 2, hr = pos / P3_WH, hcl = pos - hr * P3_WH;
        hgeo[ii] = (sl < P3_PSLOTS && (i < 5 || w4 == 0)) ? ((((sl & 3) ^ ((pos >> 1) & 3)) << 3) << 16) | (hr << 8) | hcl : -1;
    }
    int hpos[NHP], hposU[SRC ? NHP : 1];                   // pixel index of the slot in the frame (in a half-resolution source), -1: zero
    auto halo_tile = [&](int tile) {                       // the slots' sources: this group's patch of `tile` (loader waves: both patches)
#pragma unroll
        for (int q0 = 0; q0 < (LDR ? 2 : 1); ++q0) {
            const int q = LDR ? q0 : grp;
            int pi = 2 * tile + q;
            const bool pon = pi < g.npatch;
            if (!pon) pi = 2 * tile;
            const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
            const int pw0 = tw * P3_S, ph0 = (t2 % g.tiles_h) * P3_S, pf = t2 / g.tiles_h;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int ii = q0 * 6 + i;
                const int ih = ph0 - 1 + ((hgeo[ii] >> 8) & 0xff), iw = pw0 - 1 + (hgeo[ii] & 0xff);
                const bool ok = hgeo[ii] >= 0 && pon && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                hpos[ii] = ok ? (pf * p.Hi + ih) * p.Wi + iw : -1;
                if (SRC) hposU[ii] = ok ? (pf * (p.Hi >> 1) + (ih >> 1)) * (p.Wi >> 1) + (iw >> 1) : -1;
            }
        }
    };
    auto halo_piece = [&](int ii, int hcx, int buf) {
        const uint16_t *sp = p.x + hcx * 32;
        long sl = p.ldx;
        int pix = hpos[ii];
        if (SRC) {
            const int ck = hcx >> 1;
            sp = gs.ptr[ck] + (hcx & 1) * 32; sl = gs.ld[ck];
            if ((gs.up >> ck) & 1) pix = hposU[ii];
        }
        const int q = LDR ? ii / 6 : grp, i = ii % 6;
        lds_dma16(hpos[ii] >= 0 ? sp + pix * sl + (hgeo[ii] >> 16) : zero, lds0 + buf * P3_HALO + q * P3_HG + (i * 256 + w4 * 64) * 16);
    };

    // ---- weights: a stage [3 dw][64 co][32 k], piece c of row co at c ^ ((co >> 1) & 3); moved by the four waves of ONE group, a tap each instruction ---------------------
    const int wco = w4 * 16 + (lane >> 2);
    const int wof4 = wco * p.Kpad + (((lane & 3) ^ ((wco >> 1) & 3)) << 3);
    auto issue_stage = [&](int st, int slot) {             // stage st of a tile (every tile streams the same weights) -> ring slot
        const int hcx = st / 3, dh = st - hcx * 3;
        const uint16_t *src = p.w + dh * 3 * p.cin + hcx * 32 + wof4;
        const unsigned dst = lds0 + P3_WBASE + slot * P3_WST + w4 * 1024;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) lds_dma16(src + dw * p.cin, dst + dw * P3_WTAP);
    };

    // ---- MFMA roles --------------------------------------------------------------------------------------------------------------------------------------------
    const unsigned wrd = (unsigned)(l15 * 64 + ((kg ^ ((l15 >> 1) & 3)) << 4));       // this lane's piece of weight row (16 a + l15) of a tap
    const unsigned hq = (unsigned)(grp * P3_HG);
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
    auto ep_piece = [&](f32x4 (&acc)[4][4], const f32x4 (&scv)[4], const f32x4 (&sfv)[4], const int k) {
        const int r = k >> 1, pp = k & 1;
        int lz = lane;
        asm volatile("" : "+v"(lz));                       // keeps a piece's addresses out of the loops' preheaders (they would live -- spilled -- across the whole tile)
        const int l15 = lz & 15, kg = lz >> 4;
        const int ho = ep_ho0 + r, wo = ep_wo0 + l15;
        const bool valid = ep_pon && ho < p.Ho && wo < p.Wo;
        const size_t m = ((size_t)ep_pf * p.Ho + ho) * p.Wo + wo;
        const int poff = (2 * pp + (kg & 1)) * 16 + (kg >> 1) * 8;       // this lane's 16-byte piece (8 channels) of the pixel's row, after the swap
        const bool on = valid && poff < p.Cout;                          // 8-channel pieces beyond cout are neither read nor stored
        if constexpr (!GEN) {      // the inference epilogue (scale / shift, ReLU, saturation, 16-bit store): ~30 vector instructions per piece
            uint32_t pk[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int a = 2 * pp + h;
                const f32x4 sc = scv[a], sf = sfv[a];
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
        // (the general epilogue -- residual, ReLU-backward mask, fp32 output, bf16 -- runs as whole-tile passes: epilogue_general below)
    };

    auto epilogue = [&](f32x4 (&acc)[4][4]) {
        unsigned so = (unsigned)(P3_SCB + kg * 16);
        asm volatile("" : "+v"(so));                       // scale / shift of this lane's channels: eight reads per epilogue, here (not hoisted out of the tile loop into 32 registers)
        f32x4 scv[4], sfv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            scv[a] = *reinterpret_cast<const f32x4 *>(dsm + so + a * 64);
            sfv[a] = *reinterpret_cast<const f32x4 *>(dsm + so + 256 + a * 64);
        }
        if constexpr (!GEN) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ep_piece(acc, scv, sfv, k);
            return;
        } else {
        // ---- the general epilogue as passes over the whole tile, in the accumulators: scale / shift (+ statistics); the residual's eight 16-byte pieces requested TOGETHER (one
        // wait instead of eight: a load's wait also drains the DMA queue), added; ReLU; the mask's pieces likewise; rounding / stores. Per element the operations and their order are
        // those of tiles 32 / 38: same bits ----
        int lz = lane;
        asm volatile("" : "+v"(lz));
        const int l15z = lz & 15, kgz = lz >> 4;
        const int wo = ep_wo0 + l15z;
        size_t mrow[4];
        bool vrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vrow[r] = ep_pon && ep_ho0 + r < p.Ho && wo < p.Wo;
            mrow[r] = ((size_t)ep_pf * p.Ho + ep_ho0 + r) * p.Wo + wo;
        }
        const int poff0 = (kgz & 1) * 16 + (kgz >> 1) * 8;                 // + 32 pp: this lane's 16-byte piece of the pixel's row, after the swap
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][a][j] = acc[r][a][j] * scv[a][j] + sfv[a][j];
                if constexpr (STATS) {
                    if (vrow[r]) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s1[a][j] += acc[r][a][j]; s2[a][j] += acc[r][a][j] * acc[r][a][j]; }
                    }
                }
            }
        auto load8 = [&](const uint16_t *base, int ld, uint4 (&q)[8]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = k >> 1, poff = poff0 + 32 * (k & 1);
                // ONE load per piece from a per-lane address (the zero page where there is nothing to read): as `q = 0; if (valid) q = load` hipcc emitted every load under
                // its own exec mask with an s_waitcnt vmcnt(0) behind it -- eight round trips per pass where the comment above promises one wait
                const uint16_t *src = (vrow[r] && poff < p.Cout) ? base + mrow[r] * ld + poff : reinterpret_cast<const uint16_t *>(&g_zero16p3);
                q[k] = load16_opaque(src);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { swap16(q[k].x, q[k].z); swap16(q[k].y, q[k].w); }      // -> (x, y): this lane's 4 channels of group 2 pp, (z, w): of group 2 pp + 1
        };
        if (p.res) {
            uint4 q[8];
            load8(p.res, p.ldres, q);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = k >> 1, pp = k & 1;
                const uint32_t rw[2][2] = {{q[k].x, q[k].y}, {q[k].z, q[k].w}};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 &v = acc[r][2 * pp + h];
                    v[0] += T::to_f32((uint16_t)(rw[h][0] & 0xffffu)); v[1] += T::to_f32((uint16_t)(rw[h][0] >> 16));
                    v[2] += T::to_f32((uint16_t)(rw[h][1] & 0xffffu)); v[3] += T::to_f32((uint16_t)(rw[h][1] >> 16));
                }
            }
        }
        if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][a][j] = __builtin_fmaxf(acc[r][a][j], 0.f);
        }
        if (p.mask) {           // ReLU backward fused into the data gradient that produces d(input)
            uint4 q[8];
            load8(p.mask, p.ldmask, q);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = k >> 1, pp = k & 1;
                const uint32_t mw[2][2] = {{q[k].x, q[k].y}, {q[k].z, q[k].w}};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 &v = acc[r][2 * pp + h];
                    v[0] = T::to_f32((uint16_t)(mw[h][0] & 0xffffu)) > 0.f ? v[0] : 0.f; v[1] = T::to_f32((uint16_t)(mw[h][0] >> 16)) > 0.f ? v[1] : 0.f;
                    v[2] = T::to_f32((uint16_t)(mw[h][1] & 0xffffu)) > 0.f ? v[2] : 0.f; v[3] = T::to_f32((uint16_t)(mw[h][1] >> 16)) > 0.f ? v[3] : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = k >> 1, pp = k & 1, poff = poff0 + 32 * pp;
            const f32x4 &v0 = acc[r][2 * pp], &v1 = acc[r][2 * pp + 1];
            if (p.y) {
                uint32_t x0 = pack2_lim<T>(v0[0], v0[1], p.sat), x1 = pack2_lim<T>(v0[2], v0[3], p.sat);
                uint32_t y0 = pack2_lim<T>(v1[0], v1[1], p.sat), y1 = pack2_lim<T>(v1[2], v1[3], p.sat);
                swap16(x0, y0); swap16(x1, y1);
                uint4 *dst = (vrow[r] && poff < p.Cout) ? reinterpret_cast<uint4 *>(p.y + mrow[r] * p.ldy + poff) : g_sink_p3 + lane;
                *dst = make_uint4(x0, x1, y0, y1);
            }
            if (p.y32) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nch = (2 * pp + h) * 16 + kgz * 4;
                    f32x4 *dst = (vrow[r] && nch < p.Cout) ? reinterpret_cast<f32x4 *>(p.y32 + mrow[r] * p.ldy32 + nch) : reinterpret_cast<f32x4 *>(g_sink_p3 + lane);
                    *dst = h ? v1 : v0;
                }
            }
        }
        }
    };

    // ---- prologue: scale / shift, the first halo, the weights ------------------------------------------------------------------------------------------------------
    if (tid < 64) {
        *reinterpret_cast<float *>(dsm + P3_SCB + tid * 4) = p.scale[tid];
        *reinterpret_cast<float *>(dsm + P3_SCB + 256 + tid * 4) = p.shift[tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const bool no_h = (dbg & 8) != 0, no_w = (dbg & 16) != 0, no_ep = (dbg & 4) != 0;
    if constexpr (LDR) {
        if (loader) {
            // ================= the loader waves' whole program: the phase schedule and its barriers, a phase's weight stage in its first segment, the halo pieces in its second
            // (the buffer's last reads -- group 1's third tap of the phase before -- end with the first), then the counted wait =================
            halo_tile(t_begin);
            for (int ii = 0; ii < 12; ++ii)
                if (ii % 6 < 5 || w4 == 0) halo_piece(ii, 0, 0);
            {
                const int npre = total < 4 ? total : 4;              // stage n + 4 is issued in phase n
                for (int k = 0; k < npre; ++k) issue_stage(k % S, k);
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int n = 0, hb = 0, wislot = 4;
            for (int tl = 0; tl < ntl; ++tl) {
                for (int hc = 0; hc < nhc; ++hc) {
                    const bool h_next_tile = hc + 1 == nhc;
                    const bool h_exists = (!h_next_tile || tl + 1 < ntl) && !no_h;
                    const int h_hc = h_next_tile ? 0 : hc + 1;
#pragma unroll
                    for (int dh = 0; dh < 3; ++dh) {
                        if (n + 4 < total && !no_w) issue_stage((hc * 3 + dh + 4) % S, wislot);      // into the slot of stage n - 2
                        P3_SEG_END();
                        if (h_exists && dh < 2) {               // patch 0's pieces in the first phase, patch 1's in the second
                            if (dh == 0 && h_next_tile) halo_tile(t_begin + tl + 1);
#pragma unroll
                            for (int i = 0; i < 5; ++i) halo_piece(dh * 6 + i, h_hc, hb ^ 1);
                            if (w4 == 0) halo_piece(dh * 6 + 5, h_hc, hb ^ 1);
                        }
                        // What must have landed when this phase ends (both groups read it from the next phase on): weight stage n + 1 and, after a half chunk's third phase, the next
                        // half chunk's halo. This wave's queue per half chunk: [phase 0: W x 3, halo x 5 (6)] [phase 1: W x 3, halo x 5 (6)] [phase 2: W x 3]. Stage n + 1 was issued at the
                        // top of phase n - 3; behind it lie (phase 0) halo 5 | W 3, halo 5 | W 3 | W 3, halo 5 and (phase 1) halo 5 | W 3 | W 3, halo 5 | W 3, halo 5 -- this half chunk's
                        // halo only if there is one. Behind the halo's last piece lies the stage of phase 2. Counting fewer than were issued only waits longer.
                        if (n + 5 >= total || n < 3 || no_w || no_h) wait_vmcnt<0>();
                        else if (dh == 2) wait_vmcnt<3>();
                        else if (dh == 0) { if (h_exists) wait_vmcnt<24>(); else wait_vmcnt<19>(); }
                        else { if (h_exists) wait_vmcnt<24>(); else wait_vmcnt<14>(); }
                        P3_SEG_END();
                        ++n;
                        if (++wislot == P3_NWS) wislot = 0;
                    }
                    hb ^= 1;
                }
            }
            __builtin_amdgcn_s_barrier();                   // pairs with the last in-loop barrier of waves 4-7
            return;
        }
    } else {
        halo_tile(t_begin);
        for (int i = 0; i < 6; ++i)
            if (i < 5 || wr == 0) halo_piece(i, 0, 0);
        {
            const int npre = RES ? S : (total < 4 ? total : 4);      // streamed: stage n + 4 is issued in phase n
            for (int k = 0; k < npre; ++k)
                if ((k & 1) == grp) issue_stage(k % S, k);
        }
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();                           // the prologue's DMA has landed (every wave's, or the loaders')
    asm volatile("" ::: "memory");
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);

    int n = 0;                                             // phase (= weight stage) of this workgroup
    int hb = 0;                                            // halo buffer of the half chunk being multiplied
    int wslot = 0, wislot = 4;                             // ring slot of the stage read in this phase / issued in this phase (n + 4: its slot held stage n - 2, whose last reads -- the other group's third tap -- ended one segment ago)
    f32x4 acc[4][4];
    uint4 fw[2][4], fa[2][4];                              // two taps' fragments; the third tap's are read into the first set behind the first tap's MFMAs
    for (int tl = 0; tl < ntl; ++tl) {
        const int tile = t_begin + tl;
#ifdef TEDSPAD_P3_STAMPS
        const bool stamp_on = g_dbg_p3 && blockIdx.x == 0 && tl == 2;
#endif
        for (int hc = 0; hc < nhc; ++hc) {
            // the half chunk whose halo this one's first two phases fetch: the next of this tile, or the first of the next tile
            const bool h_next_tile = hc + 1 == nhc;
            const bool h_exists = (!h_next_tile || tl + 1 < ntl) && !no_h;
            const int h_hc = h_next_tile ? 0 : hc + 1;
            // ---- this half chunk's fragment addresses: Vw[dwi][e] + immediate. Position (4 wr + r + dh) * 18 + dw + l15 of the halo has its k-group kg at piece kg ^ t with
            // t = ((position >> 1) & 3) = (dh + r + ((dw + l15) >> 1)) & 3: dw = 0 / 2 share u0 = (l15 >> 1) & 3 (dw = 2: one more), dw = 1 has u1 = ((l15 + 1) >> 1) & 3 ----
            unsigned Vw[2][4];
            {
                const unsigned B00 = (unsigned)(hb * P3_HALO) + hq + (72u * wr + l15) * 64u + kg * 16u;
                const unsigned u0 = (l15 >> 1) & 3u, u1 = ((l15 + 1u) >> 1) & 3u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { Vw[0][e] = B00 ^ (((u0 + e) & 3u) << 4); Vw[1][e] = B00 ^ (((u1 + e) & 3u) << 4); }
            }
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                // ================= LOAD segment: the three taps of kernel row dh =================
                P3_STAMP((hc * 3 + dh) * 6 + 0);
                // DMA first (the longest way to go), then the finished tile's epilogue (the fragment registers are still free for it), then this phase's fragment reads
                if constexpr (!LDR) {
                if (h_exists && dh < 2) {                  // pieces 0, 1, 2 (and the group's sixth) in the first phase, 3, 4 in the second
                    if (dh == 0) {
                        if (h_next_tile) halo_tile(tile + 1);
                        halo_piece(0, h_hc, hb ^ 1); halo_piece(1, h_hc, hb ^ 1); halo_piece(2, h_hc, hb ^ 1);
                        if (wr == 0) halo_piece(5, h_hc, hb ^ 1);
                    } else {
                        halo_piece(3, h_hc, hb ^ 1); halo_piece(4, h_hc, hb ^ 1);
                    }
                }
                const bool w_own = !RES && ((n + 4) & 1) == grp && n + 4 < total && !no_w;       // this group moves stage n + 4
                if (w_own) issue_stage((hc * 3 + dh + 4) % S, wislot);
                }
                P3_STAMP((hc * 3 + dh) * 6 + 1);      // DMA issued
                if (dh == 0 && hc == 0) {
                    if (tl > 0 && !no_ep) {                // the tile before: its epilogue, straight from the accumulators, beside the other group's MFMAs
                        ep_set(tile - 1);
                        epilogue(acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int a = 0; a < 4; ++a) acc[r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_sched_barrier(0);
                }
                const unsigned wb = (unsigned)(P3_WBASE + (RES ? hc * 3 + dh : wslot) * P3_WST) + wrd;
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) fw[dw][a] = *reinterpret_cast<const uint4 *>(dsm + wb + dw * P3_WTAP + a * 1024);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        fa[dw][r] = *reinterpret_cast<const uint4 *>(dsm + Vw[dw == 1 ? 1 : 0][(dh + r) & 3] + (unsigned)(1152 * (r + dh) + 64 * dw));
                }
                P3_STAMP((hc * 3 + dh) * 6 + 2);      // weights DMA, epilogue done
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                P3_SEG_END();
                // ================= COMPUTE segment =================
                P3_STAMP((hc * 3 + dh) * 6 + 3);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[r][a] = T::mfma16(fw[0][a], fa[0][r], acc[r][a]);
                __builtin_amdgcn_sched_barrier(0);
                // the third tap's fragments, into the first tap's registers: they land under the second tap's MFMAs (a third fewer fragment registers, a third fewer reads in the
                // load segment, which is the longer one)
#pragma unroll
                for (int a = 0; a < 4; ++a) fw[0][a] = *reinterpret_cast<const uint4 *>(dsm + wb + 2 * P3_WTAP + a * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    fa[0][r] = *reinterpret_cast<const uint4 *>(dsm + Vw[0][(dh + r + 1) & 3] + (unsigned)(1152 * (r + dh) + 64 * 2));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[r][a] = T::mfma16(fw[1][a], fa[1][r], acc[r][a]);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[r][a] = T::mfma16(fw[0][a], fa[0][r], acc[r][a]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                P3_STAMP((hc * 3 + dh) * 6 + 4);      // MFMAs issued
                // ---- counted waits. This wave's queue, oldest first, per half chunk: [phase 0: halo x 3 (4), W?, the epilogue's stores] [phase 1: halo x 2, W?] [phase 2: W?]; a group
                // issues a stage (three instructions) in every other phase: stage n + 4 in phase n when (n + 4) & 1 is its number. What must have landed:
                //   * phase 2: this group's halo of the next half chunk (read from the next phase on, by this group only): everything but the youngest stage (issued in this phase or the one
                //     before); with it every stage of this group up to n + 2 has landed;
                //   * phases 0 and 1, when stage n + 2 is this group's (issued in phase n - 2; the other group reads it from phase n + 2 on, two phases after this wait): in phase 0 it was the
                //     youngest of the phase-2 wait before: behind it lie halo x 3 (4), W (n + 4) and the stores; in phase 1 it was issued in the phase-2 load before that wait: behind it lie
                //     [phase 0] halo x 3 (4), the stores, [phase 1] halo x 2, W (n + 4). (Counting a piece too few only waits longer.)
                if constexpr (LDR) {
                } else if (RES) {
                    if (dh == 2 && h_exists) wait_vmcnt<0>();
                } else if (n + 5 >= total || no_w) {
                    wait_vmcnt<0>();                        // the tail of the run: few stages left in flight, no counting
                } else if (dh == 2) {
                    wait_vmcnt<3>();
                } else if (((n + 2) & 1) == grp) {
                    const int st = hc == 0 && tl > 0 && !no_ep ? 8 * nst : 0;
                    wait_vmcnt_dyn((dh == 0 ? (h_exists ? 6 : 3) : (h_exists ? 8 : 3)) + st);
                }
                P3_STAMP((hc * 3 + dh) * 6 + 5);      // DMA waited for
                P3_SEG_END();
                ++n;
                if (!RES) {
                    if (++wslot == P3_NWS) wslot = 0;
                    if (++wislot == P3_NWS) wislot = 0;
                }
            }
            hb ^= 1;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();             // pairs with the last in-loop barrier of waves 4-7
    asm volatile("" ::: "memory");
    // ---- the last tile's epilogue ------------------------------------------------------------------------------------------------------------------------------------
    ep_set(t_end - 1);
    if (!no_ep) epilogue(acc);
    flush_stats();
}

int g_p3_cus = 0;
bool g_p3_det = false;       // deterministic mode: the persistent kernel's statistics flush is not gated (det_gate.h): it declines statistics then

template <typename T, bool SRC, bool STATS, bool RES, bool GEN, bool LDR = false>
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
    auto kfn = conv_patch3_kernel<T, SRC, STATS, RES, GEN, LDR>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    PatchSrc gsrc{};
    if (SRC) gsrc = *src;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(LDR ? P3_NTL : P3_NT), P3_LDS, s, p, g, gsrc);
    return check_launch("tedspad_conv_fwd(persistent two-patch halo)");
}

}  // namespace

void patch3_set_det(int on) { g_p3_det = on != 0; }

static int32_t launch_conv_patch3_64(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src);

// cout = 128: two launches, one per 64 output channels (weight rows, BatchNorm vectors, output / residual / mask / statistics columns shifted by 64): every output
// channel is summed in the order of the 64-channel launch, the input is read twice (the layers are MFMA-bound by intensity: cin >= 64 at 112^2 and larger)
int32_t launch_conv_patch3(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src) {
    static const bool split128 = getenv("TEDSPAD_P3_NO_128") == nullptr;          // A/B knob
    if (p.Cout != 128 || !split128) return launch_conv_patch3_64(dtype, p, N, cin, s, src);
    for (int half = 0; half < 2; ++half) {
        ConvKP q = p;
        const int o = 64 * half;
        q.Cout = 64;
        q.w = p.w + (size_t)o * p.Kpad;
        q.scale = p.scale + o; q.shift = p.shift + o;
        if (p.y) q.y = p.y + o;
        if (p.y32) q.y32 = p.y32 + o;
        if (p.res) q.res = p.res + o;
        if (p.mask) q.mask = p.mask + o;
        if (p.stats) q.stats = p.stats + o;
        const int32_t rc = launch_conv_patch3_64(dtype, q, N, cin, s, src);
        if (rc != TEDSPAD_OK) return rc;          // (the first launch declines before anything is written: the conditions do not depend on the half)
    }
    return TEDSPAD_OK;
}

static int32_t launch_conv_patch3_64(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src) {
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi;
    if (cin % 32 != 0 || (src && cin % 64 != 0) || p.kt != 1 || p.kh != 3 || p.kw != 3 || p.pt != 0 || p.ph != 1 || p.pw != 1 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same ||
        p.Kpad < 9 * cin || p.ostrided || p.sigmoid || (!p.y && !p.y32) || (src && src->n != cin / 64) || (long)N * p.Ti * p.Hi * p.Wi >= (1L << 31) ||
        p.Cout > 64 || p.Cout <= 32 || (p.stats && g_p3_det)) {
        set_error("tedspad_conv_fwd: persistent two-patch config (tile_cfg 40) needs a stride-1 'same' 1 x 3 x 3 conv with cin %% 32 == 0 (gathered sources: %% 64) and 32 < cout <= 64 or cout = 128 "
                  "(mask / stats / fp32 output allowed, no strided output map; no statistics in deterministic mode)");
        return TEDSPAD_EINVAL;
    }
    const int frames = N * p.Ti;
    const bool f16 = dtype == TEDSPAD_F16, st = p.stats != nullptr;
    const bool gen = !f16 || p.res || p.mask || p.y32 || !p.y;       // the general epilogue
    const bool res = cin <= 64;                                       // weights resident
    if (src && (res || st)) {      // (no caller: a gathered concatenation of one source is just a tensor; the gathered concatenation is the eval path of unet++)
        set_error("tedspad_conv_fwd: tile_cfg 40 takes gathered sources with cin >= 128 and without batch statistics only");
        return TEDSPAD_EINVAL;
    }
    static const bool ldr_ok = getenv("TEDSPAD_P3_LOADERS") == nullptr || atoi(getenv("TEDSPAD_P3_LOADERS")) != 0;      // A/B knob: 0 = the streamed variants without loader waves
#define P3_GO(TT, SRC_, ST_, RES_, GEN_)                                                                     \
    do {                                                                                                     \
        if (!(RES_) && !(ST_) && ldr_ok) return launch_patch3_t<TT, SRC_, false, false, GEN_, true>(p, frames, cin, s, src);   \
        return launch_patch3_t<TT, SRC_, ST_, RES_, GEN_>(p, frames, cin, s, src);                           \
    } while (0)
    if (f16) {
        if (res) { if (st) { if (gen) P3_GO(F16, false, true, true, true); P3_GO(F16, false, true, true, false); } if (gen) P3_GO(F16, false, false, true, true); P3_GO(F16, false, false, true, false); }
        if (src) { if (gen) P3_GO(F16, true, false, false, true); P3_GO(F16, true, false, false, false); }
        if (st) { if (gen) P3_GO(F16, false, true, false, true); P3_GO(F16, false, true, false, false); }
        if (gen) P3_GO(F16, false, false, false, true);
        P3_GO(F16, false, false, false, false);
    }
    if (res) { if (st) P3_GO(BF16, false, true, true, true); P3_GO(BF16, false, false, true, true); }
    if (src) P3_GO(BF16, true, false, false, true);
    if (st) P3_GO(BF16, false, true, false, true);
    P3_GO(BF16, false, false, false, true);
#undef P3_GO
}

}  // namespace tedspad
#ifdef TEDSPAD_P3_STAMPS
extern "C" int32_t tedspad_debug_set_p3_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tedspad::g_dbg_p3), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
