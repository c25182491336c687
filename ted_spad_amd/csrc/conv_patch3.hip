// Persistent two-patch convolution for gfx950 (tile_cfg 40): the stride-1 'same' 1 x 3 x 3 convs with cout <= 64 of the anonymizers' wide levels
// (unet_parts.py:8-25 DoubleConv at 112 x 112 / 224 x 224; smp's DecoderBlock convs of the default unet++ `fa`, model_loaders.py:17-30, which
// dali_extraction.py:171-173 runs in front of every clip) -- the combination the one-idea-at-a-time kernels of rounds 4 / 5 (tiles 32, 38, the dropped
// resident-weight tile) never had together:
//   * ONE 8-wave workgroup per CU, persistent over a contiguous run of tiles (a tile = two consecutive 16 x 16 patches x 64 output channels, as tile 38);
//   * the halo of a 32-channel half chunk DOUBLE-buffered (2 x 41 KB): the next half chunk's -- at the end of a tile the NEXT TILE's first -- halo is
//     issued three stages (one half chunk of MFMAs) before it is read, so no wave ever waits for a halo it has just asked for (tile 38: one buffer,
//     the fetch exposed after every half chunk);
//   * a ring of SIX weight stages ([3 taps dw][64 co][32 k] = 12 KB each) filled five stages ahead -- and when the conv has no more than six stages
//     (cin <= 64: the 64 -> 64 layers) the weights are fetched ONCE per workgroup and stay resident for all of its tiles;
//   * the epilogue straight from the accumulators: no LDS pass, no barrier -- a wave converts its own 64 px x 64 co, v_permlane16_swap turns the 16 x 16 x 32
//     accumulator quads (4 channels = 8 bytes per lane) into 16-byte pieces (64 contiguous bytes per pixel and store instruction) and stores them while the
//     prefetched halo / weight stages of the next tile are already in flight (they were issued before the epilogue started);
//   * the training extras from registers as well: batch statistics are summed per lane over ALL tiles of the workgroup and flushed once (2 atomic
//     instructions per wave instead of 128 atomics per patch), fp32 output, ReLU-backward mask, residual.
// A wave owns 4 rows of ONE patch x 64 channels (16 accumulator quads); K is walked (half chunk, dh, dw) exactly as tile 38 does: the sums are bit-identical to
// tile 38's. One barrier per stage (48 MFMAs per wave). LDS: 2 x 41 984 (halo) + 6 x 12 288 (weights) = 157 696 bytes.
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16p3;
__device__ uint4 g_sink_p3[64];          // where the masked-off lanes of the epilogue stores go (never read)

constexpr int P3_S = 16;                                                            // patch side
constexpr int P3_WH = 18, P3_NP = P3_WH * P3_WH, P3_PSLOTS = P3_NP * 4;             // 1296 16-byte slots per patch and half chunk
constexpr int P3_HALO = (2 * P3_PSLOTS + 63) / 64 * 64 * 16;                        // 41 984 (41 wave instructions; the last one's upper half is padding)
constexpr int P3_WST = 3 * 64 * 64, P3_NWS = 6, P3_AHEAD = 5;                       // weight stage bytes, ring slots, stages issued ahead
constexpr int P3_WBASE = 2 * P3_HALO;
constexpr int P3_LDS = P3_WBASE + P3_NWS * P3_WST;                                  // 157 696
constexpr int P3_NT = 512;
constexpr int P3_NHI = 5;                                                           // full halo DMA instructions per thread and half chunk (+ one more in wave 0)
static_assert(P3_LDS <= 160 * 1024, "one workgroup per CU");
static_assert(P3_NHI * P3_NT + 64 == P3_HALO / 16, "halo slots");

struct Patch3Geo {
    int tiles_h, tiles_w, npatch, ntiles, nhc, dbg;
};

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {      // n: wave-uniform; a smaller count than asked for is always safe
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        default: wait_vmcnt<14>(); break;
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

template <typename T, bool SRC, bool STATS>
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
    const int S = g.nhc * 3;                               // stages per tile
    const bool RES = S <= P3_NWS;                          // the whole weight tensor fits the ring: fetched once
    const int total = (t_end - t_begin) * S;               // stages of this workgroup
    const int WPS = wave < 4 ? 2 : 1;                      // weight DMA instructions of this wave per stage
    const int HPS = wave == 0 ? P3_NHI + 1 : P3_NHI;       // halo DMA instructions of this wave per half chunk

    // ---- halo slots of this thread (tile-independent part): slot s -> patch s / 1296, position (s % 1296) >> 2, LDS piece s & 3 ------------------
    int hgeo[P3_NHI + 1], hc8[P3_NHI + 1];                 // (patch << 16) | (halo row << 8) | halo column, or -1: padding slot
#pragma unroll
    for (int i = 0; i <= P3_NHI; ++i) {
        const int s = i * P3_NT + tid;
        const int q = s >= P3_PSLOTS ? 1 : 0, r = s - q * P3_PSLOTS;
        const int pos = r >> 2, hr = pos / P3_WH, hcl = pos - hr * P3_WH;
        hc8[i] = ((r & 3) ^ ((pos >> 1) & 3)) << 3;
        hgeo[i] = (s < 2 * P3_PSLOTS && (i < P3_NHI || tid < 64)) ? (q << 16) | (hr << 8) | hcl : -1;
    }
    int hpos[P3_NHI + 1], hposU[SRC ? P3_NHI + 1 : 1];
    auto halo_addr = [&](int tile) {                       // pixel indices of this thread's halo slots for `tile` (-1: outside the frame / no patch)
        int pf[2], ph0[2], pw0[2];
        bool pon[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int pi = 2 * tile + q;
            pon[q] = pi < g.npatch;
            if (!pon[q]) pi = 2 * tile;
            const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
            pw0[q] = tw * P3_S; ph0[q] = (t2 % g.tiles_h) * P3_S; pf[q] = t2 / g.tiles_h;
        }
#pragma unroll
        for (int i = 0; i <= P3_NHI; ++i) {
            const int q = (hgeo[i] >> 16) & 1;
            const int ih = ph0[q] - 1 + ((hgeo[i] >> 8) & 0xff), iw = pw0[q] - 1 + (hgeo[i] & 0xff);
            const bool ok = hgeo[i] >= 0 && pon[q] && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
            hpos[i] = ok ? (pf[q] * p.Hi + ih) * p.Wi + iw : -1;
            if (SRC) hposU[i] = ok ? (pf[q] * (p.Hi >> 1) + (ih >> 1)) * (p.Wi >> 1) + (iw >> 1) : -1;
        }
    };
    auto issue_halo = [&](int hcx, int buf) {              // half chunk hcx of the tile halo_addr() was last called for -> halo buffer buf
        const uint16_t *sp = p.x + hcx * 32;
        long sl = p.ldx;
        bool up = false;
        if (SRC) {
            const int ck = hcx >> 1;
            sp = gs.ptr[ck] + (hcx & 1) * 32; sl = gs.ld[ck]; up = (gs.up >> ck) & 1;
        }
        const unsigned dst = lds0 + buf * P3_HALO;
        if (wave == 0) {      // the 41st wave instruction goes FIRST: the counted waits see the same tail in every wave
            const int pi = SRC && up ? hposU[P3_NHI] : hpos[P3_NHI];
            lds_dma16(hpos[P3_NHI] >= 0 ? sp + pi * sl + hc8[P3_NHI] : zero, dst + P3_NHI * P3_NT * 16);
        }
#pragma unroll
        for (int i = 0; i < P3_NHI; ++i) {
            const int pi = SRC && up ? hposU[i] : hpos[i];
            lds_dma16(hpos[i] >= 0 ? sp + pi * sl + hc8[i] : zero, dst + (i * P3_NT + wave * 64) * 16);
        }
    };
    // ---- weight stage (hc, dh): [dw][co][32 k], piece c of row co at c ^ ((co >> 1) & 3) ----------------------------------------------------------
    int wof[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = j * P3_NT + tid, row = s >> 2, dw = row >> 6, co = row & 63;
        wof[j] = co * p.Kpad + dw * p.cin + (((s & 3) ^ ((co >> 1) & 3)) << 3);
    }
    auto issue_w = [&](int st, int slot) {                 // stage st of a tile (every tile streams the same weights) -> ring slot
        const int hcx = st / 3, dh = st - hcx * 3;
        const unsigned dst = lds0 + P3_WBASE + slot * P3_WST + wave * 1024;
        const uint16_t *src = p.w + dh * 3 * p.cin + hcx * 32;
        lds_dma16(src + wof[0], dst);
        if (wave < 4) lds_dma16(src + wof[1], dst + 8192);
    };

    // ---- MFMA roles: wave w owns rows 4 (w & 3) .. + 3 of patch w >> 2, all 64 channels ---------------------------------------------------------------
    const int l15 = lane & 15, kg = lane >> 4;
    const int wq = wave >> 2, wr = wave & 3;
    const unsigned wrd = (unsigned)(l15 * 64 + ((kg ^ ((l15 >> 1) & 3)) << 4));       // this lane's piece of weight row (16 a + l15) of a stage
    const unsigned hq = (unsigned)(wq * (P3_PSLOTS * 16));
    float s1[STATS ? 4 : 1][4], s2[STATS ? 4 : 1][4];       // batch statistics of this lane's channels (16 a + 4 kg + j) over every tile of the workgroup
    if (STATS) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }
    }
    float4 scv[4], sfv[4];                                  // scale / shift of this lane's channels (read before the first DMA: no vector-memory wait of the compiler's meets the DMA queue later)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        scv[a] = reinterpret_cast<const float4 *>(p.scale)[a * 4 + kg];
        sfv[a] = reinterpret_cast<const float4 *>(p.shift)[a * 4 + kg];
    }
    long sgrp_cur = -1;
    auto flush_stats = [&]() {                             // wave-local: sums over the 16 pixel lanes, then ONE atomic instruction per statistic
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

    // ---- prologue: the first tile's first halo, then the weight stages ----------------------------------------------------------------------------------
    halo_addr(t_begin);
    issue_halo(0, 0);
    const int npre = RES ? S : (total < P3_AHEAD ? total : P3_AHEAD);
    for (int k = 0; k < npre; ++k) issue_w(k % S, k % P3_NWS);
    wait_vmcnt_dyn(RES ? 0 : (npre - 1) * WPS);            // the halo and stage 0 landed

    int hb = 0, gsn = 0;                                   // halo buffer of the half chunk being multiplied; stage counter of this workgroup
    // timing ablations (wrong results; TEDSPAD_P3_ABLATE): 1 no MFMA, 2 no fragment reads, 4 no epilogue, 8 no halo DMA after the prologue, 16 no weight DMA after the prologue,
    // 32 no barriers, 64 no DMA waits
    const int dbg = g.dbg;
    uint4 fw[4], fa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fw[i] = make_uint4(lane, i, 0x3c003c00u, 0u); fa[i] = make_uint4(i, lane, 0x3c003c00u, 0u); }
    for (int tile = t_begin; tile < t_end; ++tile) {
        f32x4 acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < S; ++st, ++gsn) {
            const int hcx = st / 3, dh = st - hcx * 3;
            if (!(dbg & 32)) __builtin_amdgcn_s_barrier();                  // stage gsn (and on dh = 0 its halo) is visible to every wave; every wave is past stage gsn - 1
            asm volatile("" ::: "memory");
            bool h_issued = false;
            if (dh == 0 && !(dbg & 8)) {                                 // the other halo buffer was last read two barriers ago
                if (hcx + 1 < g.nhc) { issue_halo(hcx + 1, hb ^ 1); h_issued = true; }
                else if (tile + 1 < t_end) { halo_addr(tile + 1); issue_halo(0, hb ^ 1); h_issued = true; }
            }
            if (!RES && gsn + P3_AHEAD < total && !(dbg & 16)) issue_w((gsn + P3_AHEAD) % S, (gsn + P3_AHEAD) % P3_NWS);     // its slot held stage gsn - 1
            const unsigned wb = (unsigned)(P3_WBASE + (RES ? st : gsn % P3_NWS) * P3_WST) + wrd;
            const unsigned hbase = (unsigned)(hb * P3_HALO) + hq;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                if (!(dbg & 2)) {
#pragma unroll
                for (int a = 0; a < 4; ++a) fw[a] = *reinterpret_cast<const uint4 *>(dsm + wb + dw * 4096 + a * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pos = (4 * wr + r + dh) * P3_WH + dw + l15;
                    fa[r] = *reinterpret_cast<const uint4 *>(dsm + hbase + (unsigned)(pos * 64 + ((kg ^ ((pos >> 1) & 3)) << 4)));
                }
                }
                if (!(dbg & 1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[r][a] = T::mfma16(fw[a], fa[r], acc[r][a]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fw[i].x), "v"(fw[i].y), "v"(fw[i].z), "v"(fw[i].w), "v"(fa[i].x), "v"(fa[i].y), "v"(fa[i].z), "v"(fa[i].w));
                }
            }
            // ---- what the NEXT stage reads must have landed before the barrier that opens it (this wave's share; the barrier covers the others') ----------
            if (gsn + 1 < total && !(dbg & 64)) {
                // younger than what is needed, by issue order: weight stages gsn+2 .. gsn+5 (gsn+3 .. when the halo is waited for too: it was issued in front
                // of stage gsn+3's weights), and on dh = 0 / 1 the halo issued at the top of this half chunk. The epilogue's stores are not counted: a smaller count only waits longer.
                int allowed = 0;
                if (!RES) {
                    const int last = gsn + P3_AHEAD < total ? gsn + P3_AHEAD : total - 1;      // youngest weight stage issued so far
                    const int first = dh == 2 ? gsn + 3 : gsn + 2;
                    if (last >= first) allowed = (last - first + 1) * WPS;
                }
                if (dh == 2) wait_vmcnt_dyn(allowed);
                else if (!RES) {
                    // the halo issued in this half chunk (dh = 0: this stage, dh = 1: the stage before) may stay in flight
                    const bool hfl = dh == 0 ? h_issued : (hcx + 1 < g.nhc || tile + 1 < t_end);
                    wait_vmcnt_dyn(allowed + (hfl ? HPS : 0));
                }
            }
            if (dh == 2) hb ^= 1;
        }

        // ---- epilogue straight from the accumulators -------------------------------------------------------------------------------------------------------
        if (dbg & 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(acc[r][a][0]), "v"(acc[r][a][1]), "v"(acc[r][a][2]), "v"(acc[r][a][3]));
            continue;
        }
        int pi = 2 * tile + wq;
        const bool pon = pi < g.npatch;
        if (!pon) pi = 2 * tile;
        const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
        const int wo = tw * P3_S + l15, ho0 = (t2 % g.tiles_h) * P3_S + 4 * wr, pf = t2 / g.tiles_h;
        if (STATS) {
            const long mfirst = ((long)pf * p.Ho + (ho0 - 4 * wr)) * p.Wo + tw * P3_S;       // a patch lies inside one frame: inside one statistics group
            const long grp = p.stats_rows ? mfirst / p.stats_rows : 0;
            if (grp != sgrp_cur) {
                flush_stats();
                sgrp_cur = grp;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ho = ho0 + r;
            const bool valid = pon && ho < p.Ho && wo < p.Wo;
            const size_t m = ((size_t)pf * p.Ho + ho) * p.Wo + wo;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {               // channel groups a = 2 pp, 2 pp + 1 -> one 16-byte piece per lane
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int a = 2 * pp + h;
                    const float4 sc = scv[a], sf = sfv[a];
                    v[h][0] = acc[r][a][0] * sc.x + sf.x; v[h][1] = acc[r][a][1] * sc.y + sf.y;
                    v[h][2] = acc[r][a][2] * sc.z + sf.z; v[h][3] = acc[r][a][3] * sc.w + sf.w;
                    if constexpr (STATS) {
                        if (valid) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { s1[a][j] += v[h][j]; s2[a][j] += v[h][j] * v[h][j]; }
                        }
                    }
                }
                const int poff = (2 * pp + (kg & 1)) * 16 + (kg >> 1) * 8;       // this lane's 16-byte piece (8 channels) of the pixel's row, after the swap
                if (p.res) {
                    uint4 rv = make_uint4(0u, 0u, 0u, 0u);
                    if (valid && poff < p.Cout) rv = *reinterpret_cast<const uint4 *>(p.res + m * p.ldres + poff);
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
                    if (valid && poff < p.Cout) mv = *reinterpret_cast<const uint4 *>(p.mask + m * p.ldmask + poff);
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
                    const bool on = valid && poff < p.Cout;       // 8-channel pieces beyond cout are not stored
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
            }
        }
    }
    flush_stats();
}

int g_p3_cus = 0;
bool g_p3_det = false;       // deterministic mode: the persistent kernel's statistics flush is not gated (det_gate.h): it declines statistics then

template <typename T, bool SRC, bool STATS>
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
    auto kfn = conv_patch3_kernel<T, SRC, STATS>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    PatchSrc gsrc{};
    if (SRC) gsrc = *src;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(P3_NT), P3_LDS, s, p, g, gsrc);
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
    const bool f16 = dtype == TEDSPAD_F16;
    if (p.stats) {
        if (src) return f16 ? launch_patch3_t<F16, true, true>(p, frames, cin, s, src) : launch_patch3_t<BF16, true, true>(p, frames, cin, s, src);
        return f16 ? launch_patch3_t<F16, false, true>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, true>(p, frames, cin, s, nullptr);
    }
    if (src) return f16 ? launch_patch3_t<F16, true, false>(p, frames, cin, s, src) : launch_patch3_t<BF16, true, false>(p, frames, cin, s, src);
    return f16 ? launch_patch3_t<F16, false, false>(p, frames, cin, s, nullptr) : launch_patch3_t<BF16, false, false>(p, frames, cin, s, nullptr);
}

}  // namespace tedspad
