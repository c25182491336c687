// Persistent Cin = 3 stem for gfx950: conv1 (5x7x7, stride 2, pad (2,3,3)) + bn1 + ReLU of large_i3d.py:133-137,229-231, with the
// TEMPORAL half of maxpool1 (MaxPool3d((2,3,3), stride 2), large_i3d.py:138,232) fused into its epilogue.
//
// What the earlier stem kernels measured (DESIGN.md "Stem: what was measured"): K = 1120 for 735 real taps x channels in the
// pixel-pair form; a barrier every 8-16 MFMAs because the [64 co][64 k] weight tile streams through a ring; a 10 k-cycle halo
// prologue and a 7 k-cycle epilogue around a 19 k-cycle loop; 2.89 GB written per 225 clips that the max-pool shrinks 8x.
// Here:
//   * temporal-unfolded K (one k16 MFMA step per (dh, dw) tap: 16 values = 5 frames x 3 channels + 1 zero) -> K = 784;
//   * the clip is laid out ONCE per output-frame PAIR (tedspad_clip_to_tp): X[n][tp][h][b][w/2][24] 16-bit, a 48-byte record per
//     pixel (column 2*wq + b, the two column parities in separate planes) holding the 8 input frames 4*tp - 2 .. 4*tp + 5 x 3
//     channels that output frames 2*tp and 2*tp + 1 read: frame 2*tp takes the 32 bytes at offset 0, frame 2*tp + 1 the 32 bytes
//     at offset 12 (a 4-byte aligned LDS-DMA source). 9.6 MB per clip. (First version: one 128-byte record per pixel with all 16
//     frames, 6.4 MB per clip -- but then every 16-byte DMA piece comes from a different 128-byte line, 32 lines per 1 KB
//     wave-instruction, and the halo stream alone took 1180 us per 225 clips, the cache's line rate, not bytes; here consecutive
//     positions of a plane are consecutive records: 12 lines per instruction.)
//   * ALL weights (49 taps x [64 co][16] = 98 KB) stay resident in LDS: the K loop has no weight stream, its fragment
//     addresses are per-lane bases + compile-time immediates (fully unrolled, no address arithmetic);
//   * a workgroup is persistent (one per CU) and walks patches of 8 x 16 output pixels x 2 output frames; the input halo of a
//     patch (2 frames x 21 rows x 38 columns x 32 B = 50 KB) lives in two LDS regions by ROW PARITY: the taps with even dh read
//     only even halo rows, the odd ones only odd rows, so while the 28 even-dh taps of patch i are multiplied the odd rows of patch
//     i are landing, and while its 21 odd-dh taps run the even rows of patch i+1 land: one halo buffer, two barriers per patch.
//     (Four regions by (row, column) parity -- a DMA distance of 2/3 of a patch instead of 1/2 -- were measured: 10 % slower, the
//     two extra barriers and pipeline restarts per patch cost more than the distance buys; the halo stream is throughput-, not
//     latency-bound.)
//   * a wave owns output rows (r, r+4) x 16 columns of BOTH frames (64 px); the two frames are the temporal pooling window, so
//     relu(bn(.)) of both are max-ed in registers and only the pooled tensor Y[n][to/2][ho][wo][64] is written (half the bytes),
//     straight from the accumulators: v_permlane32_swap gives lane l the even and lane l + 32 the odd 8-channel group of its
//     pixel -> 16-byte stores, no LDS staging; the barrier after them waits with a counted vmcnt that leaves them in flight.
// LDS images are bank-conflict free for ds_read_b128: positions are 32 B (two 16-byte halves = the two k8 halves of the MFMA
// B operand); the halves of position p are stored swapped when (p >> 3) & 1, and the second pixel row of a fragment is 4
// output rows away (4 x 1216 B = 0 mod 256 B); weight rows (32 B per co) swap halves when (co >> 4) & 1.
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16s;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PT_TH = 8, PT_TW = 16;                  // output patch (rows x columns), x 2 output frames
constexpr int PT_PP = PT_TW + 3;                      // positions per parity-plane row (a = -2 .. +1 around 16 columns)
constexpr int PT_ROWB = 2 * PT_PP * 32;               // bytes per halo row: 2 column-parity planes
constexpr int PT_NTAP = 49;
constexpr int PT_W_BYTES = PT_NTAP * 64 * 32;         // 100352
constexpr int PT_REC = 48;                            // bytes per pixel record of the clip layout
__host__ __device__ constexpr int pt_rows(int par) { return par == 0 ? PT_TH + 3 : PT_TH + 2; }   // even halo rows 0..20: 11, odd: 10
__host__ __device__ constexpr int pt_frame(int par) { return pt_rows(par) * PT_ROWB; }            // bytes per frame of a region
__host__ __device__ constexpr int pt_jobs(int par) { return (2 * pt_frame(par) + 1023) / 1024; }  // 1 KB wave-instructions (27 / 24)
// the halo regions come first (their fragment addresses then are one per-lane base + a 16-bit immediate), the weights after them
__host__ __device__ constexpr int pt_off(int par) { return par == 0 ? 0 : pt_jobs(0) * 1024; }
constexpr int PT_W_OFF = pt_off(1) + pt_jobs(1) * 1024;
constexpr int PT_LDS = PT_W_OFF + PT_W_BYTES;
// MF = 16 (v_mfma_f32_16x16x32): one MFMA sums TWO taps (K = 32); the 49 taps are 25 pairs (the last one with a zero partner)
constexpr int PT_NPAIR = 25;
constexpr int PT_W16_BYTES = PT_NPAIR * 64 * 64;      // [pair][co][4 chunks of 8 values]: 102400
__host__ __device__ constexpr int pt_wbytes(int mf) { return mf == 16 ? PT_W16_BYTES : PT_W_BYTES; }
// POOL: the patch's column-pooled rows [8 rows][9 slots][64 channels] 16-bit (slots 0..6: complete 3-column windows, 7: columns
// 14, 15 of the window the next patch to the right completes, 8: column 0 alone, that patch's contribution to its left neighbour)
constexpr int PT_XB_ROW = 9 * 128, PT_XB_BYTES = PT_TH * PT_XB_ROW;
__host__ __device__ constexpr int pt_xb_off(int mf) { return PT_W_OFF + pt_wbytes(mf); }
__host__ __device__ constexpr int pt_lds_pool(int mf) { return pt_xb_off(mf) + PT_XB_BYTES; }
constexpr int PT_LDS_POOL = pt_lds_pool(32);
static_assert(pt_lds_pool(16) <= 160 * 1024 && pt_xb_off(16) % 128 == 0 && pt_xb_off(32) % 128 == 0, "MF = 16 image: exactly 160 KB");
constexpr int PT_POOL_PIECES = 4 * 9 * 8;            // 16-byte pieces a patch hands to the pooled tensor (4 pooled rows x 9 slots x 8)
static_assert(PT_LDS_POOL <= 160 * 1024, "weights + halo (+ pooled rows) must fit the CU's LDS");
static_assert(PT_W_BYTES % 1024 == 0 && (4 * PT_ROWB) % 256 == 0 && PT_ROWB % 32 == 0, "LDS image alignment");

struct StemPT {
    const unsigned char *x;       // X[n][tp][h][b][wq][24] 16-bit (tedspad_clip_to_tp)
    const unsigned char *wimg;    // [49][64][2][8] 16-bit, halves swizzled (tedspad_stem_pt_fwd)
    const float *scale, *shift;
    uint16_t *y;
    long sTp;                     // bytes per frame pair of x
    int sH, sP;                   // bytes per row (both planes) / plane row
    int N, Tp, H, Wq, Ho, Wo, ldy, relu;
    int tiles_h, tiles_w, total, chunk;     // POOL: total / chunk count column STRIPS (tiles_h patches each)
    uint16_t *side;               // POOL: S[n][tp][hp][tiles_w][64], column 0 of every patch pooled over rows
    int Hp, Wp;                   // POOL: pooled rows / columns
    int dbg;                      // timing ablations (wrong results): 1 = no halo DMA after the first patch, 2 = no stores, 4 = no MFMA phases
    // DIRECT: the kernel reads the fp32 (n, c, t, h, w) clip itself (no layout pass): element strides, W contiguous
    const float *xf;
    long sN;
    int sC, sT, sHf, C, T, W, pad_t;
};

__device__ __forceinline__ void gstore16(void *dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");   // s_nop 1: a store of > 64 bits reads its data registers late; hipcc pads nothing after an asm statement and may overwrite them
}

// The taps of one row parity, fully unrolled: the fragments of tap i+1 are requested BEFORE the MFMAs of tap i (two register sets),
// pinned with sched_group_barrier, so the matrix pipe stays fed across the LDS latency (hipcc's own order requests a tap's fragments
// only after the previous tap's MFMAs are issued: 2580 vs 2300 us per 225 clips).
template <typename T, int PAR, int NA>
__device__ __forceinline__ void stem_pt_phase(const unsigned char *dsm, const int (&pa)[4], const int (&wa)[NA], f32x16 (&acc)[NA][2]) {
    constexpr int NT = (PAR == 0 ? 4 : 3) * 7;
    constexpr int FR = pt_frame(PAR);
    uint4 fa[2][2], fw[2][NA];
    auto load = [&](int i, uint4 (&xa)[2], uint4 (&xw)[NA]) {
        const int dhh = i / 7, dw = i % 7;
        const int b = (dw + 1) & 1;                   // column 2*wo + dw - 3 = 2*(wo + a) + b
        const int ta = (dw - 3 - b) / 2 + 2;          // a + 2 (dw - 3 - b is even)
        const int tau = (2 * dhh + PAR) * 7 + dw;
#pragma unroll
        for (int g = 0; g < 2; ++g) xa[g] = *reinterpret_cast<const uint4 *>(dsm + pa[ta] + (pt_off(PAR) + g * FR + dhh * PT_ROWB + b * PT_PP * 32));
#pragma unroll
        for (int a = 0; a < NA; ++a) xw[a] = *reinterpret_cast<const uint4 *>(dsm + wa[a] + tau * 2048);
    };
    load(0, fa[0], fw[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 + NA, 0);      // the first tap's reads open the pipeline
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (i + 1 < NT) load(i + 1, fa[(i + 1) & 1], fw[(i + 1) & 1]);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[a][g] = T::mfma(fw[i & 1][a], fa[i & 1][g], acc[a][g]);
        if (i + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NA, 0);   // the LDS reads of tap i+1
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NA, 0);                   // the MFMAs of tap i
    }
}

// MF = 16: the same taps on v_mfma_f32_16x16x32 (K = 32 = two taps per instruction; under load the chip holds a ~12 % higher clock on this
// shape than on 32x32x16 at the same FLOP per cycle -- MI355X guide, DVFS (7); measured here: the same instruction mix runs the kernel in
// 1 887 instead of 2 137 us). Lane l: pixel column l & 15 (B operand) / output channel l & 15 (A operand), k-block q = l >> 4: tap q >> 1 of
// the pair, 8-value half q & 1 of its 16 values. The taps of a pair differ by a CONSTANT LDS offset (dw odd -> dw + 1: the other column-parity
// plane; dw = 0: the next row of the same parity), which sits in the per-lane base register (sel = q >> 1 times the offset): no per-read
// address arithmetic. A wave (8 per workgroup) owns 32 output channels x {2 rows x 2 frames} of 16 columns: 2 weight + 4 pixel fragments
// per 8 MFMAs. Pairs of phase PAR (dh = 2 dhh + PAR): i < 3*ND: (dhh = i / 3, dw = 1 + 2 (i % 3)) with (dhh, dw + 1); then dw = 0:
// (dhh = 2 k, 0) with (2 k + 1, 0); the last pair of phase 1 is (dhh = 2, dw = 0) with a zero-weight partner.
template <typename T, int PAR>
__device__ __forceinline__ void stem_pt_phase16(const unsigned char *dsm, const int (&pb)[5], const int wa, f32x4 (&acc)[2][2][2]) {
    constexpr int ND = PAR == 0 ? 4 : 3;
    constexpr int NP = PAR == 0 ? 14 : 11;
    constexpr int PBASE = PAR == 0 ? 0 : 14;
    constexpr int FR = pt_frame(PAR);
    uint4 fx[2][4], fw[2][2];
    auto load = [&](int i, uint4 (&xa)[4], uint4 (&xw)[2]) {
        int bi, off;
        if (i < 3 * ND) { bi = i % 3; off = (i / 3) * PT_ROWB; }
        else if (PAR == 0 || i == 3 * ND) { bi = 3; off = 2 * (i - 3 * ND) * PT_ROWB + PT_PP * 32; }
        else { bi = 4; off = 2 * PT_ROWB + PT_PP * 32; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int f = 0; f < 2; ++f) xa[r * 2 + f] = *reinterpret_cast<const uint4 *>(dsm + pb[bi] + (pt_off(PAR) + off + f * FR + r * 4 * PT_ROWB));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) xw[cb] = *reinterpret_cast<const uint4 *>(dsm + wa + (PBASE + i) * 4096 + cb * 1024);
    };
    load(0, fx[0], fw[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i + 1 < NP) load(i + 1, fx[(i + 1) & 1], fw[(i + 1) & 1]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[cb][r][f] = T::mfma16(fw[i & 1][cb], fx[i & 1][r * 2 + f], acc[cb][r][f]);
        // the next pair's six reads in two groups of three between the two halves of this pair's MFMAs (6 | 8 groups measured the same)
        if (i + 1 < NP) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (i + 1 < NP) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
}

// NW = 4: one wave per SIMD, every wave multiplies all 64 output channels of its 64 pixels (4 fragment reads per 4 MFMAs).
// NW = 8: two waves per SIMD; waves w and w + 4 own the same pixels and the output channels [0,32) / [32,64) (3 reads per 2
// MFMAs: 192 B/clk of the CU's 256 B/clk LDS at full MFMA rate): one wave's epilogue, DMA issue and patch arithmetic run
// under its partner's MFMAs.
//
// POOL: the SPATIAL half of maxpool1 (3 x 3, stride 2, no padding) as well. A workgroup walks column strips top to bottom. In the
// epilogue every wave pools its pixels over columns inside its 16-lane rows (DPP row shifts: lane = column) and leaves the patch's 8
// column-pooled rows in LDS; after the next barrier (the one every patch has anyway) 288 threads pool them over rows and store 16
// bytes each: pooled rows 4*th .. 4*th + 2 from this patch, row 4*th - 1 from rows 6, 7 of the patch above (carried in registers)
// and row 0 of this one. The window a patch shares with its right neighbour is written as two partial maxima (slot 7 into the
// pooled tensor, slot 8 into the side buffer) that stem_pool_fix_kernel joins. 1.44 GB per 225 clips no longer leave the chip.
//
// DIRECT: no layout pass in front (tedspad_clip_to_tp: 4.3 GB of traffic and 790 us per 225 clips for a pure copy). The halo image is built from the fp32
// NCTHW clip by the workgroup itself: thread = (half-position slot group): halo row, 4 consecutive columns (one 16-byte load per (frame, channel) plane: the
// halo's first column 2 wo0 - 4 is a multiple of 4), output frame f, k8-half hs -> 8 (7) loads of the planes (frame slot, channel) = value 6 f + 8 hs + e,
// converted and written as 4 x 16 bytes (one k8-half of 4 positions). 440 / 400 tasks per row-parity region: one per thread. The loads are ordinary vector
// loads into registers, so they need no LDS slot while in flight: the rows of patch i + 2 are requested a WHOLE patch before they are written (right after the
// rows of patch i + 1 left the same registers), which no LDS-DMA ring could afford here (160 KB are full).
//
// LW > 0 (DIRECT): LW LOADER waves beside the NW compute waves. The loaders own everything that touches global memory in the loop -- the fp32 loads of the halo rows,
// their conversion and LDS writes, the row pooling and its stores -- on the same two barriers per patch, with the same schedule (a region's rows are requested a whole
// patch before they are written); the compute waves run the MFMA phases and the epilogue into LDS and nothing else, and carry none of the 64 staging registers:
// 16 waves at 128 registers instead of 8 at 256 (LW = 8: one task per loader thread and region, as in the 8-wave form). On the record path (LW = 4) the loaders issue
// the halo LDS-DMA and wait for it, and pool the rows. Measured (375 clips, isolated): fp32 clip 4 045 -> 3 880 us, records 3 450 -> 3 160 us; cfg2 bench +1.2 %.
// What bounds the fp32 form either way is the VALU port the conversion (3 instructions per value pair) and the address arithmetic share with the MFMA issue.
template <typename T, int NW, bool POOL, int MF = 32, bool DIRECT = false, int LW = 0>
__global__ __launch_bounds__(64 * (NW + LW)) void conv_stem_pt_kernel(const StemPT p) {
    static_assert(MF == 32 || (MF == 16 && NW == 8 && POOL), "the 16x16x32 form is built for 8 waves with the pool fused");
    static_assert(!DIRECT || (MF == 16 && NW == 8 && POOL), "the fp32-clip loader is built for the production form");
    static_assert(LW == 0 || (MF == 16 && NW == 8 && POOL), "loader waves are built for the production forms");
    constexpr int NA = NW == 8 ? 1 : 2;
    constexpr int XB_OFF = pt_xb_off(MF);
    constexpr int NPT = 64 * (LW ? LW : NW);                 // threads that pool rows / (LW) own loader tasks
    constexpr int PIT = (PT_POOL_PIECES + NPT - 1) / NPT;
    constexpr int NTK = LW ? (4 * pt_rows(0) * 10 + NPT - 1) / NPT : 1;      // loader tasks per thread and region
    constexpr int NLW = LW ? LW : NW;                        // waves that issue the record path's halo DMA
    constexpr int ROUNDS = (pt_jobs(0) + NLW - 1) / NLW;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptid = LW ? tid - 64 * NW : tid;             // index among the pooling / loader threads (LW: negative on the compute waves, which do neither)
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const unsigned char *zero = reinterpret_cast<const unsigned char *>(&g_zero16s);

    // ---- this workgroup's patches: the 8 XCDs take contiguous chunks, an XCD's workgroups walk its chunk interleaved (neighbouring
    // patches, whose halos overlap, are in flight on one L2 at the same time) ------------------------------------------------------
    const int xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
    const int base = xcd * p.chunk;
    const int lim = min(p.chunk, p.total - base);
    int k = blockIdx.x >> 3;
    if (k >= lim) return;                                   // workgroup-uniform, before any barrier

    // ---- resident weights: a linear 98 KB copy --------------------------------------------------------------------------------
    if (wave < NW)
        for (int j = wave; j < pt_wbytes(MF) / 1024; j += NW) lds_dma16(p.wimg + j * 1024 + lane * 16, lds0 + PT_W_OFF + j * 1024);

    // ---- halo DMA slots of this lane (patch-invariant): LDS slot s of a region = (frame, row, plane, position, half) --------------
    int off[2][ROUNDS], rc[2][ROUNDS];
    const int dw0 = LW ? wave - NW : wave;                   // index among the DMA-issuing waves
    auto setup_dma = [&]() {
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int s = (i * NLW + dw0) * 64 + lane;
            const int hs = s & 1;
            int q = s >> 1;
            const int pos = q % PT_PP; q /= PT_PP;
            const int b = q & 1; q >>= 1;
            const int row = q % pt_rows(par), f = q / pt_rows(par);
            const int hr = 2 * row + par;                               // halo row 0..20 (input row ih0 + hr); plane column wo0 - 2 + pos
            const int hf = MF == 16 ? hs : hs ^ ((pos >> 3) & 1);   // MF = 16: unswizzled (see stem_pt_phase16)
            const bool ok = f < 2;
            off[par][i] = ok ? f * 12 + hf * 16 + hr * p.sH + b * p.sP + (pos - 2) * PT_REC : 0;
            rc[par][i] = ok ? (hr << 8) | pos : (1 << 28);              // a row far outside any clip: the slot reads the zero page
        }
    };
    if (!DIRECT && !LW) setup_dma();                         // (LW: inside the loader branch)

    // patch coordinates advance by a constant step (nx patches): mixed-radix addition with carries instead of three integer
    // divisions per patch
    struct Patch { const unsigned char *pb; const float *pf; int ih0, wqm2, n, tp, th, tw, ho0, wo0; };
    auto split = [&](int r, int &n_, int &tp_, int &th_, int &tw_) {
        tw_ = r % p.tiles_w; r /= p.tiles_w;
        if (POOL) th_ = 0;                                  // r counts strips
        else { th_ = r % p.tiles_h; r /= p.tiles_h; }
        tp_ = r % p.Tp; n_ = r / p.Tp;
    };
    auto locate = [&](Patch &q) {
        q.ho0 = q.th * PT_TH; q.wo0 = q.tw * PT_TW;
        q.ih0 = 2 * q.ho0 - 3; q.wqm2 = q.wo0 - 2;
        if (DIRECT) q.pf = p.xf + (long)q.n * p.sN + (long)(4 * q.tp - p.pad_t) * p.sT + (long)q.ih0 * p.sHf + (2 * q.wo0 - 4);   // frame slot 0, halo row 0, halo column 0
        else q.pb = p.x + ((long)q.n * p.Tp + q.tp) * p.sTp + (long)q.ih0 * p.sH + (long)q.wo0 * PT_REC;
    };
    int dn, dtp, dth, dtw;
    split(nx, dn, dtp, dth, dtw);
    auto advance = [&](const Patch &c) {
        Patch q = c;
        if (POOL && c.th + 1 < p.tiles_h) {                 // down the strip
            ++q.th;
            locate(q);
            return q;
        }
        if (POOL) q.th = 0;
        q.tw += dtw; if (q.tw >= p.tiles_w) { q.tw -= p.tiles_w; if (POOL) ++q.tp; else ++q.th; }
        if (!POOL) { q.th += dth; if (q.th >= p.tiles_h) { q.th -= p.tiles_h; ++q.tp; } }
        q.tp += dtp; if (q.tp >= p.Tp) { q.tp -= p.Tp; ++q.n; }
        q.n += dn;
        locate(q);
        return q;
    };
    auto issue = [&](int par, const Patch &q) {              // par is a literal at every call site
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int j = i * NLW + dw0;
            if (j >= pt_jobs(par)) break;                    // wave-uniform
            const bool ok = (unsigned)(q.ih0 + (rc[par][i] >> 8)) < (unsigned)p.H && (unsigned)(q.wqm2 + (rc[par][i] & 255)) < (unsigned)p.Wq;
            lds_dma16(ok ? q.pb + off[par][i] : zero, lds0 + pt_off(par) + j * 1024);
        }
    };

    // ---- DIRECT: this thread's task of either region: (frame f, half hs) = tid / (rows * 10), halo row, column quad q4c ------------------------------
    // so[par][e]: element offset of plane e's 4 columns from the patch base; fsch[par]: 3 bits frame slot + 2 bits channel per e; ldst[par]: LDS byte
    // address of the first of the 4 positions' halves (j -> + (j >> 1) * 32 + (j & 1) * plane); dcol / drow: halo column / row of the task
    int so[2][NTK][LW ? 1 : 8], fsch[2][NTK][2], ldst[2][NTK], dcol[2][NTK], drow[2][NTK];      // LW: so[.][.][0] = the task's (row, column) offset, the plane offsets are formed per load
    f32x4 stage[2][NTK][8];
    auto setup_tasks = [&]() {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int t = 0; t < NTK; ++t) {
                const int tk = ptid + t * NPT;
                const int per = pt_rows(par) * 10;
                const int fh = tk / per, rem = tk - fh * per;
                const int row = rem / 10, q4c = rem - row * 10;
                const int f = fh >> 1, hs = fh & 1, hr = 2 * row + par;
                const bool task = fh < 4;
                drow[par][t] = task ? hr : (1 << 20);          // no task: a row outside every clip -> zero page, and nothing is written (ldst < 0)
                dcol[par][t] = 4 * q4c;
                ldst[par][t] = task ? pt_off(par) + f * pt_frame(par) + row * PT_ROWB + 2 * q4c * 32 + 16 * hs : -1;
                if (LW) { drow[par][t] = (task ? hr : 4095) | (4 * q4c) << 12 | (fh & 3) << 20; dcol[par][t] = 0; so[par][t][0] = hr * p.sHf + 4 * q4c; }      // LW: row / column / (f, hs) in one word, one offset (the loader waves run at 128 registers)
                fsch[par][t][0] = fsch[par][t][1] = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int v = 6 * f + 8 * hs + e, fs = v / 3, ch = v - 3 * fs;
                    if (!LW) so[par][t][e] = ch * p.sC + fs * p.sT + hr * p.sHf + 4 * q4c;
                    fsch[par][t][e >> 2] |= (fs | (ch << 3)) << (8 * (e & 3));
                }
            }
    };
    if (DIRECT && !LW) setup_tasks();                       // (LW: inside the loader branch -- nothing of it may be live on the compute waves' path)
    auto ldst_of = [&](int par, int t) -> int {              // LDS byte address of the task's first half (-1: no task)
        return ldst[par][t];
    };
    auto issue_d = [&](int par, const Patch &q) {           // par is a literal at every call site
        const int tt0 = 4 * q.tp - p.pad_t;
#pragma unroll
        for (int t = 0; t < NTK; ++t) {
            int fc0 = fsch[par][t][0], fc1 = fsch[par][t][1], meta = drow[par][t], sb = so[par][t][0];
            if (LW) asm volatile("" : "+v"(meta), "+v"(sb), "+v"(fc0), "+v"(fc1));      // or hipcc hoists the 32 plane offsets out of the patch loop and spills them
            const int t_row = LW ? meta & 4095 : drow[par][t], t_col = LW ? (meta >> 12) & 255 : dcol[par][t];
            const bool rowok = (unsigned)(q.ih0 + t_row) < (unsigned)p.H && (unsigned)(2 * q.wo0 - 4 + t_col) < (unsigned)p.W;
            const bool hs1 = LW ? (t_row != 4095 && ((meta >> 20) & 1)) : (ldst[par][t] >= 0 && ((ldst[par][t] >> 4) & 1));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int code = ((e < 4 ? fc0 : fc1) >> (8 * (e & 3))) & 255;
                const bool ok = rowok && (unsigned)(tt0 + (code & 7)) < (unsigned)p.T && (code >> 3) < p.C && !(e == 7 && hs1);   // value 15 of a position meets a zero weight
                const int o = LW ? sb + (code >> 3) * p.sC + (code & 7) * p.sT : so[par][t][e];
                const float *src = ok ? q.pf + o : reinterpret_cast<const float *>(zero);
                // one load from a per-lane address: left to itself hipcc turns the select into two masked loads of the same registers (clip / zero page) with an
                // s_waitcnt vmcnt(0) between them, which serialises a thread's eight loads into four round trips
                unsigned long sa = reinterpret_cast<unsigned long>(src);
                asm volatile("" : "+v"(sa));
                stage[par][t][e] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(sa);      // (address space 1: a global_load, not a flat one)
            }
        }
        asm volatile("" ::: "memory");                       // the loads stay here (the registers are meant to be live across the next MFMA phase)
    };
    auto cvt2 = [&](float a, float b) -> unsigned {         // two values -> packed 16-bit pair, saturated like T::from_f32 (one v_med3_f32 per value)
        if constexpr (T::kDtype == TEDSPAD_F16) {
            unsigned pk;
            const float x = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), y = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(y));
            return pk;
        } else {
            return (unsigned)T::from_f32(a) | ((unsigned)T::from_f32(b) << 16);
        }
    };
    auto pack_d = [&](int buf, int t, int j) -> uint4 {
        return make_uint4(cvt2(stage[buf][t][0][j], stage[buf][t][1][j]), cvt2(stage[buf][t][2][j], stage[buf][t][3][j]),
                          cvt2(stage[buf][t][4][j], stage[buf][t][5][j]), cvt2(stage[buf][t][6][j], stage[buf][t][7][j]));
    };
    auto write_d = [&](int par, int t, int j, uint4 v) {    // one k8-half of position j of the task's four
        const bool last = (LW ? (drow[par][t] >> 12) & 255 : dcol[par][t]) == 36;      // columns 38, 39 are outside the halo (19 positions per plane)
        const int la = ldst_of(par, t);
        if (la >= 0 && (j < 2 || !last)) *reinterpret_cast<uint4 *>(dsm + la + (j >> 1) * 32 + (j & 1) * (PT_PP * 32)) = v;
    };
    auto commit_d = [&](int par) {                           // registers -> 4 x 16 bytes of the region's LDS image
#pragma unroll
        for (int t = 0; t < NTK; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) write_d(par, t, j, pack_d(par, t, j));
    };
    Patch cur;
    split(base + k, cur.n, cur.tp, cur.th, cur.tw);
    locate(cur);
    if (DIRECT && !LW) {
        issue_d(0, cur);
        issue_d(1, cur);
        commit_d(0);
        commit_d(1);
    } else if (!DIRECT && !LW) {
        issue(0, cur);
        issue(1, cur);
    }

    // the successor of patch (c, kc) in this workgroup's walk (and whether there is one)
    auto successor = [&](const Patch &c, int kc, Patch &out, int &kout) -> bool {
        const bool wrap = !POOL || c.th + 1 == p.tiles_h;
        kout = wrap ? kc + nx : kc;
        out = c;
        if (kout >= lim) return false;                      // workgroup-uniform
        out = advance(c);
        return true;
    };
    const bool dma = !(p.dbg & 1);
    // ---- POOL: this thread's pieces of the row pooling: u -> (pooled row i of the patch, slot, 8-channel group) --------------------
    int pxo[PIT], pi[PIT], psl[PIT];
    uint4 carry[PIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        const int u = ptid + it * NPT;
        psl[it] = (u >> 3) % 9;
        pi[it] = (u >= 0 && u < PT_POOL_PIECES) ? (u >> 3) / 9 : -1;
        pxo[it] = XB_OFF + psl[it] * 128 + (u & 7) * 16;
        carry[it] = make_uint4(0, 0, 0, 0);
    }
    auto pool_rows = [&](const Patch &q) {
        const size_t fr = (size_t)q.n * p.Tp + q.tp;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            if (LW) {                                       // the loader waves re-derive their piece per patch (three registers; they run at 128)
                int u = ptid + it * NPT;
                asm volatile("" : "+v"(u));
                psl[it] = (u >> 3) % 9;
                pi[it] = u < PT_POOL_PIECES ? (u >> 3) / 9 : -1;
                pxo[it] = XB_OFF + psl[it] * 128 + (u & 7) * 16;
            }
            if (pi[it] < 0) continue;
            const unsigned char *xb = dsm + pxo[it];
            uint4 m;
            int pr;
            if (pi[it] < 3) {
                const unsigned char *r = xb + 2 * pi[it] * PT_XB_ROW;
                m = pk_max8<T>(pk_max8<T>(*reinterpret_cast<const uint4 *>(r), *reinterpret_cast<const uint4 *>(r + PT_XB_ROW)),
                               *reinterpret_cast<const uint4 *>(r + 2 * PT_XB_ROW));
                pr = 4 * q.th + pi[it];
            } else {                                        // rows 6, 7 of the patch above + row 0 of this one
                m = pk_max8<T>(carry[it], *reinterpret_cast<const uint4 *>(xb));
                carry[it] = pk_max8<T>(*reinterpret_cast<const uint4 *>(xb + 6 * PT_XB_ROW), *reinterpret_cast<const uint4 *>(xb + 7 * PT_XB_ROW));
                pr = q.th > 0 ? 4 * q.th - 1 : p.Hp;
            }
            const int pc = 8 * q.tw + psl[it];
            uint16_t *dst = psl[it] < 8 ? p.y + ((fr * p.Hp + pr) * p.Wp + pc) * p.ldy : p.side + ((fr * p.Hp + pr) * p.tiles_w + q.tw) * 64;
            if (pr < p.Hp && (psl[it] == 8 || pc < p.Wp) && !(p.dbg & 2)) gstore16(dst + (pxo[it] & 127) / 2, __builtin_bit_cast(u32x4, m));
        }
    };


    // ---- LW: the loader waves ---------------------------------------------------------------------------------------------------------------------------------------
    // barriers: #0 (weights + both regions of the first patch), then per patch MID (even rows read, odd rows visible) and END (odd rows read, even rows of the next patch
    // visible; not after the last patch), then the final one before the last patch's row pooling -- the compute waves' sequence below.
    if (LW && !DIRECT && wave >= NW) {                       // ---- record path: the loaders issue the halo DMA (the schedule of the 8-wave form: a region is requested when the barrier frees it)
        setup_dma();
        Patch c = cur, n1 = cur;
        int kc = k, k1 = 0;
        issue(0, c);
        issue(1, c);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                         // #0
        asm volatile("" ::: "memory");
        while (true) {
            const bool more1 = successor(c, kc, n1, k1);
            wait_vmcnt<0>();                                  // the odd rows of patch c (and the row-pooling stores of the patch before) landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // MID of patch c
            asm volatile("" ::: "memory");
            if (more1 && dma) issue(0, n1);
            if (!more1) break;
            wait_vmcnt<0>();                                  // the even rows of the next patch landed
            __builtin_amdgcn_s_barrier();                     // END of patch c
            asm volatile("" ::: "memory");
            if (dma) issue(1, n1);
            pool_rows(c);
            c = n1; kc = k1;
        }
        asm volatile("; record loaders: last barrier" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pool_rows(c);
        return;
    }
    if (LW && DIRECT && wave >= NW) {
        setup_tasks();
        Patch c = cur, n1 = cur, n2 = cur;
        int kc = k, k1 = 0, k2 = 0;
        bool more1 = successor(c, kc, n1, k1);
        bool more2 = more1 && successor(n1, k1, n2, k2);
        issue_d(0, c);
        issue_d(1, c);
        wait_vmcnt<0>();
        commit_d(0);
        commit_d(1);
        bool l1 = more1 && dma;                               // the next patch's rows are to be loaded (dbg 1: none after the first patch)
        if (l1) { issue_d(0, n1); issue_d(1, n1); }           // the second patch's rows are in flight before the first one starts
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // #0
        asm volatile("" ::: "memory");
        while (true) {
            const bool l2 = more2 && dma;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // MID of patch c: its even rows are free
            asm volatile("" ::: "memory");
            if (l1) {
                wait_vmcnt<8 * NTK>();                        // the even rows of the next patch, requested a patch ago (behind them: its odd rows, at most one pooling store)
                commit_d(0);
                if (l2) issue_d(0, n2);                       // even rows two patches ahead take over the registers
            }
            if (!more1) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // END of patch c: its odd rows are free, its column-pooled rows are in LDS
            asm volatile("" ::: "memory");
            if (l1) {
                if (l2) wait_vmcnt<8 * NTK>(); else wait_vmcnt<0>();
                commit_d(1);                                  // odd rows of the next patch
                if (l2) issue_d(1, n2);
            }
            pool_rows(c);
            c = n1; kc = k1;
            more1 = more2; n1 = n2; k1 = k2; l1 = l2;
            more2 = more1 && successor(n1, k1, n2, k2);
        }
        asm volatile("; loader waves: last barrier" ::: "memory");   // (a distinct statement: hipcc otherwise merges this tail with the compute waves' and keeps the loaders' state live across THEIR loop: two spills)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pool_rows(c);
        return;
    }

    // ---- fragment bases ---------------------------------------------------------------------------------------------------------
    const int l15 = lane & 15, rsel = (lane >> 4) & 1, lh = lane >> 5, l31 = lane & 31;
    const int prow = (wave & 3) + 4 * rsel;                 // output row of this lane's pixel inside the patch
    const int a0 = NW == 8 ? wave >> 2 : 0;                 // first 32-channel group of this wave
    int pa[4], wa[NA];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
        const int pos = l15 + ta;
        pa[ta] = prow * PT_ROWB + pos * 32 + 16 * (lh ^ ((pos >> 3) & 1));
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) wa[a] = PT_W_OFF + ((a0 + a) * 32 + l31) * 32 + 16 * (lh ^ ((((a0 + a) * 32 + l31) >> 4) & 1));

    // MF = 16: per-lane bases of the pixel fragments ([0..2]: the dw pairs at plane position a = 1..3, [3]: the dw = 0 row pair, [4]: the
    // single dw = 0 tap) and of the weight fragments. Bank conflicts: a ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31} (+32): 8 lanes of one k-half on columns {0-3,12-15} and 8 lanes of the OTHER half on columns {4-11}. Sixteen
    // consecutive 32-byte positions with that half assignment cover all 64 banks when the halves are stored UNswizzled (columns c and c + 8
    // always sit in different halves); for the 64-byte weight rows the chunk index is XOR-ed with 2 * bit 3 of co
    const int q4 = lane >> 4;
    int pb16[5], wa16 = 0;
    if (MF == 16) {
        const int sel = q4 >> 1, hf = q4 & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = l15 + (j == 3 ? 0 : j + 1);
            const int base = (wave & 3) * PT_ROWB + pos * 32 + 16 * hf;
            pb16[j] = base + sel * (j == 3 ? PT_ROWB : PT_PP * 32);
            if (j == 3) pb16[4] = base;
        }
        const int co0 = a0 * 32 + l15;
        wa16 = PT_W_OFF + co0 * 64 + 16 * (q4 ^ (2 * ((co0 >> 3) & 1)));
    }
    // BatchNorm scale / shift of this lane's output channels: co = a*32 + (r & 3) + 8*(r >> 2) + 4*lh (MF = 16: a0*32 + 16*(r >> 2) + 4*q4 + (r & 3))
    float sc[NA][16], sf[NA][16];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = MF == 16 ? a0 * 32 + 16 * ((r >> 2) & 1) + 4 * q4 + (r & 3) : (a0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sc[a][r] = p.scale[co];
            sf[a][r] = p.shift[co];
        }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(sc[a][r]), "+v"(sf[a][r]));   // hipcc's wait for these loads happens here, not in the loop

    if (DIRECT && !LW && dma) {                             // the second patch's rows are in flight before the first one starts
        Patch n1; int k1;
        if (successor(cur, k, n1, k1)) { issue_d(0, n1); issue_d(1, n1); }
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // weights + both halo regions of the first patch visible
    asm volatile("" ::: "memory");

    bool pending = false;                                   // POOL: the column-pooled rows of `prev` wait in LDS
    Patch prev = cur;
    while (true) {
        Patch nxt, nn;
        int kn, kn2 = 0;
        const bool more = successor(cur, k, nxt, kn);       // workgroup-uniform
        bool more2 = false;
        nn = nxt;
        if (DIRECT && !LW && more) more2 = successor(nxt, kn, nn, kn2);
        if (POOL && !LW && pending) pool_rows(prev);        // before this patch's mid barrier; the epilogue after it rewrites the rows

        f32x16 acc[NA][2];
        f32x4 acc16[2][2][2];              // MF = 16: [16-channel block][row r / r + 4][frame]
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][g][r] = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc16[cb][r][f] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (MF == 16) { if (!(p.dbg & 4)) stem_pt_phase16<T, 0>(dsm, pb16, wa16, acc16); }
        else if (!(p.dbg & 4)) stem_pt_phase<T, 0, NA>(dsm, pa, wa, acc);             // taps dh = 0, 2, 4, 6 on the even halo rows
        if (!DIRECT && !LW) wait_vmcnt<0>();                // odd rows of this patch (the youngest operation of this wave) landed
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // DIRECT: this wave's writes of the odd rows (after the previous barrier) are in LDS
        __builtin_amdgcn_s_barrier();                       // ... for every wave; every wave is done reading the even rows
        asm volatile("" ::: "memory");
        if (DIRECT) {
            if (!LW && more && dma) {
                commit_d(0);                                // even rows of the NEXT patch: requested a patch ago, written under the odd taps
                if (more2) issue_d(0, nn);                  // ... and the ones after them take over the registers
            }
        } else if (!LW && more && dma) issue(0, nxt);       // even rows of the NEXT patch land under the odd taps + epilogue
        if (MF == 16) { if (!(p.dbg & 4)) stem_pt_phase16<T, 1>(dsm, pb16, wa16, acc16); }
        else if (!(p.dbg & 4)) stem_pt_phase<T, 1, NA>(dsm, pa, wa, acc);             // taps dh = 1, 3, 5 on the odd halo rows

        // ---- epilogue: relu(bn(.)) of both frames, max over the two frames (the temporal window of maxpool1), 16-byte stores --------
        bool stored = false;
        if (POOL && MF == 16) {
            // lane: column l15 (= its DPP row position), channels a0*32 + 16 cb + 4 q4 + {0..3} of rows (wave & 3) + 4 r
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float s_ = sc[0][4 * cb + i], b_ = sf[0][4 * cb + i];
                        v[i] = __builtin_fmaxf(__builtin_fmaxf(acc16[cb][r][0][i] * s_ + b_, acc16[cb][r][1][i] * s_ + b_), 0.f);
                    }
                    const unsigned d0 = (unsigned)T::from_f32(v[0]) | ((unsigned)T::from_f32(v[1]) << 16);
                    const unsigned d1 = (unsigned)T::from_f32(v[2]) | ((unsigned)T::from_f32(v[3]) << 16);
                    unsigned m[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned d = h ? d1 : d0;
                        const unsigned s1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x101, 0xf, 0xf, true);
                        const unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x102, 0xf, 0xf, true);
                        m[h] = T::pk_max(T::pk_max(d, s1), s2);
                    }
                    unsigned char *xb = dsm + XB_OFF + ((wave & 3) + 4 * r) * PT_XB_ROW + (a0 * 32 + cb * 16 + 4 * q4) * 2;
                    if (!(l15 & 1)) *reinterpret_cast<uint2 *>(xb + (l15 >> 1) * 128) = make_uint2(m[0], m[1]);
                    if (l15 == 0) *reinterpret_cast<uint2 *>(xb + 8 * 128) = make_uint2(d0, d1);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (POOL) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 4 * q + 2 * h + e;
                            const float v0 = acc[a][0][r] * sc[a][r] + sf[a][r], v1 = acc[a][1][r] * sc[a][r] + sf[a][r];
                            v[e] = __builtin_fmaxf(__builtin_fmaxf(v0, v1), 0.f);      // ReLU always: the pooling below pads with 0
                        }
                        d[q][h] = (unsigned)T::from_f32(v[0]) | ((unsigned)T::from_f32(v[1]) << 16);
                    }
#pragma unroll
                for (int q = 0; q < 4; q += 2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        auto sw = __builtin_amdgcn_permlane32_swap(d[q][h], d[q + 1][h], false, false);
                        d[q][h] = sw[0];
                        d[q + 1][h] = sw[1];
                    }
                // columns: lane = column inside a 16-lane DPP row; row_shl:n reads lane + n, lanes past the row read 0
                unsigned m[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned s1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d[q][h], 0x101, 0xf, 0xf, true);
                        const unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d[q][h], 0x102, 0xf, 0xf, true);
                        m[q][h] = T::pk_max(T::pk_max(d[q][h], s1), s2);
                    }
                unsigned char *xb = dsm + XB_OFF + prow * PT_XB_ROW + ((a0 + a) * 32 + 8 * lh) * 2;
                if (!(l15 & 1)) {
                    *reinterpret_cast<uint4 *>(xb + (l15 >> 1) * 128) = make_uint4(m[0][0], m[0][1], m[1][0], m[1][1]);
                    *reinterpret_cast<uint4 *>(xb + (l15 >> 1) * 128 + 32) = make_uint4(m[2][0], m[2][1], m[3][0], m[3][1]);
                }
                if (l15 == 0) {
                    *reinterpret_cast<uint4 *>(xb + 8 * 128) = make_uint4(d[0][0], d[0][1], d[1][0], d[1][1]);
                    *reinterpret_cast<uint4 *>(xb + 8 * 128 + 32) = make_uint4(d[2][0], d[2][1], d[3][0], d[3][1]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are in LDS before this wave reaches the next barrier
        } else {
            const int ho = cur.ho0 + prow, wo = cur.wo0 + l15;
            const bool inb = ho < p.Ho && wo < p.Wo && !(p.dbg & 2);
            stored = __builtin_amdgcn_ballot_w64(inb) != 0;  // the store instructions below are issued iff any lane is in bounds
            uint16_t *dst = p.y + ((((size_t)cur.n * p.Tp + cur.tp) * p.Ho + ho) * p.Wo + wo) * p.ldy + 8 * lh;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 4 * q + 2 * h + e;
                            const float v0 = acc[a][0][r] * sc[a][r] + sf[a][r], v1 = acc[a][1][r] * sc[a][r] + sf[a][r];
                            float m = __builtin_fmaxf(v0, v1);
                            if (p.relu) m = __builtin_fmaxf(m, 0.f);
                            v[e] = m;
                        }
                        d[q][h] = (unsigned)T::from_f32(v[0]) | ((unsigned)T::from_f32(v[1]) << 16);
                    }
                // lane l (lh = 0) ends with channels 8q .. 8q+7 for q = 0 / 2, lane l + 32 with those of q = 1 / 3
#pragma unroll
                for (int q = 0; q < 4; q += 2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        auto sw = __builtin_amdgcn_permlane32_swap(d[q][h], d[q + 1][h], false, false);
                        d[q][h] = sw[0];
                        d[q + 1][h] = sw[1];
                    }
                if (inb) {
                    gstore16(dst + (a0 + a) * 32, u32x4{d[0][0], d[0][1], d[1][0], d[1][1]});
                    gstore16(dst + (a0 + a) * 32 + 16, u32x4{d[2][0], d[2][1], d[3][0], d[3][1]});
                }
            }
        }
        if (!more) break;
        // even rows of the next patch landed; this patch's stores (issued after them; vector-memory operations retire in issue
        // order on gfx9) stay in flight
        if (DIRECT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the even rows this wave wrote after the mid barrier are in LDS
        else if (LW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (stored) wait_vmcnt<2 * NA>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                       // ... for every wave; every wave is done reading the odd rows
        asm volatile("" ::: "memory");
        if (DIRECT) {
            if (!LW && dma) {
                commit_d(1);
                if (more2) issue_d(1, nn);
            }
        } else if (!LW && dma) issue(1, nxt);               // odd rows of the next patch land under its even taps
        prev = cur;
        pending = POOL;
        cur = nxt;
        k = kn;
    }
    if (POOL) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!LW) pool_rows(cur);
    }
}

// joins the two halves of the pooled columns 8*tw - 1 (columns 14, 15 of patch tw - 1 | column 0 of patch tw)
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fix_kernel(uint16_t *y, const uint16_t *side, long rows, int wp, int tiles_w, int ldy) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(idx & 7);
    const long r = idx >> 3;
    const int tw = 1 + (int)(r % (tiles_w - 1));
    const long row = r / (tiles_w - 1);
    const int pc = 8 * tw - 1;
    if (row >= rows || pc >= wp) return;
    uint4 *d = reinterpret_cast<uint4 *>(y + (row * wp + pc) * ldy + c8 * 8);
    *d = pk_max8<T>(*d, *reinterpret_cast<const uint4 *>(side + (row * tiles_w + tw) * 64 + c8 * 8));
}

// fp32 NCTHW clip (any strides, W contiguous) -> X[n][tp][h][b][w/2][24] 16-bit: the record of pixel (h, 2*wq + b) for output-frame
// pair tp holds value dt*3 + c = x[n][c][2*st*tp - pt + dt][h][2*wq + b], dt = 0..7, zero outside the clip. A workgroup owns 64
// consecutive pixels of one row; wave g builds the records of pair g (g + 4, ...): 24 coalesced row loads per thread (the four
// frames two neighbouring pairs share are re-read through L1), three 16-byte LDS writes, then every (pair, plane) leaves as one
// contiguous 1.5 KB run.
template <typename T>
__global__ __launch_bounds__(256) void clip_to_tp_kernel(const float *x, uint16_t *y, int c, int t, int h, int w, long sn, long sc, long st, long sh,
                                                         int pt, int stt, int tp_n, int wtiles) {
    __shared__ __attribute__((aligned(16))) uint4 tile[4 * 2 * 32 * 3];     // [wave = pair][plane][32 records][3 x 16 B]
    const int tid = threadIdx.x, px = tid & 63, grp = tid >> 6;
    int b = blockIdx.x;
    const int wt = b % wtiles; b /= wtiles;
    const int ih = b % h;
    const long n = b / h;
    const int iw = wt * 64 + px;
    const int wq_n = w >> 1;
    const int nrec = min(64, w - wt * 64) >> 1;                             // records per plane in this tile
    for (int tp0 = 0; tp0 < tp_n; tp0 += 4) {
        const int tp = tp0 + grp;
        if (tp < tp_n) {
            float v[24];
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                const int tt = 2 * stt * tp - pt + e / 3, ch = e % 3;
                v[e] = 0.f;
                if (iw < w && tt >= 0 && tt < t && ch < c) v[e] = x[n * sn + ch * sc + tt * st + ih * sh + iw];
            }
            uint4 *rec = tile + ((grp * 2 + (px & 1)) * 32 + (px >> 1)) * 3;
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
                float f8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f8[e] = v[k3 * 8 + e];
                rec[k3] = pack8<T>(f8);
            }
        }
        __syncthreads();
        for (int i = tid; i < 8 * 96; i += 256) {
            const int run = i / 96, j = i - run * 96;                       // run = (wave, plane); j-th 16-byte piece of its 32 records
            const int g = run >> 1, bb = run & 1;
            if (tp0 + g < tp_n && j < nrec * 3) {
                uint4 *dst = reinterpret_cast<uint4 *>(y) + ((((n * tp_n + tp0 + g) * h + ih) * 2 + bb) * (long)wq_n + wt * 32) * 3;
                dst[j] = tile[run * 96 + j];
            }
        }
        __syncthreads();
    }
}

// The same layout pass with 16-byte loads (rows 16-byte aligned, W % 4 == 0): the kernel above issues 24 four-byte loads per thread and pair and re-reads the four
// frames two neighbouring pairs share -- 2 x the clip's bytes in 256-byte wave-instructions, which is what it spent its time on (the CU's address unit, not HBM).
// Here a workgroup still owns 64 consecutive pixels of a row; for a group of four pairs it loads each of the 20 frames x 3 channels ONCE as float4 (16 lanes per
// 64-pixel segment), converts, and scatters the values into an LDS tile [64 px][20 frames x 3 channels] (136-byte rows); the record of (pair g, pixel) is then the
// 48 bytes at offset 24 g of its row, and every (pair, plane) leaves as one contiguous 1.5 KB run as before.
template <typename T>
__global__ __launch_bounds__(256) void clip_to_tp4_kernel(const float *x, uint16_t *y, int c, int t, int h, int w, long sn, long sc, long st, long sh,
                                                          int pt, int stt, int tp_n, int wtiles) {
    constexpr int ROWB = 136;                                               // 20 x 3 x 2 = 120 bytes used; 34 dwords: a quarter-wave of the scatter hits 8 banks twice
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * ROWB];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int wt = b % wtiles; b /= wtiles;
    const int ih = b % h;
    const long n = b / h;
    const int wq_n = w >> 1;
    const int npx = min(64, w - wt * 64);
    const int nrec = npx >> 1;                                              // records per plane in this tile
    const float *xr = x + n * sn + ih * sh + wt * 64;
    for (int tp0 = 0; tp0 < tp_n; tp0 += 4) {
        const int f0 = 2 * stt * tp0 - pt;                                  // first frame slot of the group
        for (int i = tid; i < 60 * 16; i += 256) {
            const int seg = i >> 4, q = i & 15;                             // segment = (frame slot, channel), pixels 4q .. 4q + 3
            const int fs = seg / 3, ch = seg - fs * 3, tt = f0 + fs;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (4 * q < npx && tt >= 0 && tt < t && ch < c) v = *reinterpret_cast<const f32x4 *>(xr + ch * sc + tt * st + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint16_t *>(tile + (4 * q + j) * ROWB + seg * 2) = T::from_f32(v[j]);
        }
        __syncthreads();
        for (int i = tid; i < 8 * 96; i += 256) {
            const int run = i / 96, j = i - run * 96;                       // run = (pair of the group, plane); j-th 16-byte piece of its 32 records
            const int g = run >> 1, bb = run & 1;
            if (tp0 + g < tp_n && j < nrec * 3) {
                const int rec = j / 3, k3 = j - rec * 3;
                const unsigned char *src = tile + (2 * rec + bb) * ROWB + 24 * g + 16 * k3;        // 8-byte aligned
                const uint2 lo = *reinterpret_cast<const uint2 *>(src), hi = *reinterpret_cast<const uint2 *>(src + 8);
                uint4 *dst = reinterpret_cast<uint4 *>(y) + ((((n * tp_n + tp0 + g) * h + ih) * 2 + bb) * (long)wq_n + wt * 32) * 3;
                dst[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_clip_to_tp(const float *x, void *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc,
                                      int64_t st, int64_t sh, int64_t sw, int32_t pad_t, int32_t stride_t, int32_t t_pairs, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && c <= 3 && t > 0 && h > 0 && w > 0 && pad_t >= 0 && t_pairs > 0, "tedspad_clip_to_tp: bad arguments");
    TS_REQUIRE(stride_t == 2, "tedspad_clip_to_tp: temporal stride 2 (the second frame of a pair starts 12 bytes into the 48-byte record)");
    TS_REQUIRE(sw == 1 && w % 2 == 0 && (uintptr_t)y % 16 == 0, "tedspad_clip_to_tp: rows must be contiguous, W even, y 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_clip_to_tp: bad dtype");
    TS_REQUIRE((long)n * h * ((w + 63) / 64) < (1L << 31), "tedspad_clip_to_tp: too many tiles");
    const int wtiles = (w + 63) / 64;
    const dim3 g((unsigned)((long)n * h * wtiles));
    hipStream_t s = (hipStream_t)stream;
    if (w % 4 == 0 && ((uintptr_t)x | (uintptr_t)(sn * 4) | (uintptr_t)(sc * 4) | (uintptr_t)(st * 4) | (uintptr_t)(sh * 4)) % 16 == 0) {
        if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(clip_to_tp4_kernel<F16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, stride_t, t_pairs, wtiles);
        else hipLaunchKernelGGL(clip_to_tp4_kernel<BF16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, stride_t, t_pairs, wtiles);
        return check_launch("tedspad_clip_to_tp");
    }
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(clip_to_tp_kernel<F16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, stride_t, t_pairs, wtiles);
    else hipLaunchKernelGGL(clip_to_tp_kernel<BF16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, stride_t, t_pairs, wtiles);
    return check_launch("tedspad_clip_to_tp");
}

extern "C" int32_t tedspad_stem_pt_wimg_bytes(void) { return PT_W_BYTES; }
extern "C" int32_t tedspad_stem_pt_wimg16_bytes(void) { return PT_W16_BYTES; }

// shared launcher: pool = the spatial 3x3 / 2 max-pool fused as well (y is the pooled tensor then, side its scratch)
static int32_t stem_pt_launch(const char *who, const void *x_tp, const void *w_img, const float *scale, const float *shift, void *y, void *side, int32_t n,
                              int32_t t_pairs, int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t hp, int32_t wp, int32_t ldy, int32_t relu, int32_t nwg,
                              int32_t variant, int32_t dtype, bool pool, hipStream_t s, const StemPT *direct = nullptr) {
    StemPT p;
    p.xf = nullptr; p.sN = 0; p.sC = p.sT = p.sHf = 0; p.C = 3; p.T = 0; p.W = w; p.pad_t = 0;
    if (direct) { p.xf = direct->xf; p.sN = direct->sN; p.sC = direct->sC; p.sT = direct->sT; p.sHf = direct->sHf; p.C = direct->C; p.T = direct->T; p.pad_t = direct->pad_t; }
    p.x = (const unsigned char *)x_tp; p.wimg = (const unsigned char *)w_img; p.scale = scale; p.shift = shift; p.y = (uint16_t *)y;
    p.side = (uint16_t *)side; p.Hp = hp; p.Wp = wp;
    p.Wq = w / 2; p.sP = p.Wq * PT_REC; p.sH = 2 * p.sP; p.sTp = (long)h * p.sH;
    p.N = n; p.Tp = t_pairs; p.H = h; p.Ho = ho; p.Wo = wo; p.ldy = ldy; p.relu = relu;
    p.tiles_h = (ho + PT_TH - 1) / PT_TH; p.tiles_w = (wo + PT_TW - 1) / PT_TW;
    const long total = (long)n * t_pairs * p.tiles_w * (pool ? 1 : p.tiles_h);      // pool: column strips
    if (total >= (1L << 30) || (long)n * t_pairs * p.tiles_h * p.tiles_w >= (1L << 30)) {
        set_error("%s: too many patches", who);
        return TEDSPAD_EINVAL;
    }
    p.total = (int)total;
    p.chunk = (int)((total + 7) / 8);
    int grid = nwg > 0 ? nwg : 256;
    grid = (grid + 7) / 8 * 8;
    if ((long)grid > total + 7) grid = (int)((total + 7) / 8 * 8);
    static thread_local int attr_set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int w8 = (variant >> 1) & 1;
    p.dbg = (variant >> 8) & 15;
    if (direct) {                          // the same form reading the fp32 clip itself
        static const bool lw = getenv("TEDSPAD_STEM_LOADERS") == nullptr || atoi(getenv("TEDSPAD_STEM_LOADERS")) != 0;       // A/B knob: 0 = the 8-wave form
        static thread_local int attrd[4] = {0, 0, 0, 0};
        const int ti = (dtype == TEDSPAD_F16 ? 0 : 1) + (lw ? 2 : 0);
        const void *fnd = ti == 0 ? (const void *)conv_stem_pt_kernel<F16, 8, true, 16, true> : ti == 1 ? (const void *)conv_stem_pt_kernel<BF16, 8, true, 16, true>
                        : ti == 2 ? (const void *)conv_stem_pt_kernel<F16, 8, true, 16, true, 8> : (const void *)conv_stem_pt_kernel<BF16, 8, true, 16, true, 8>;
        if (!attrd[ti]) {
            if (hipFuncSetAttribute(fnd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("%s: cannot raise the dynamic LDS limit", who);
                return TEDSPAD_ELAUNCH;
            }
            attrd[ti] = 1;
        }
        if (ti == 0) hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, true, 16, true>), dim3(grid), dim3(512), pt_lds_pool(16), s, p);
        else if (ti == 1) hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, true, 16, true>), dim3(grid), dim3(512), pt_lds_pool(16), s, p);
        else if (ti == 2) hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, true, 16, true, 8>), dim3(grid), dim3(1024), pt_lds_pool(16), s, p);
        else hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, true, 16, true, 8>), dim3(grid), dim3(1024), pt_lds_pool(16), s, p);
    } else if (pool && (variant & 4)) {           // 16x16x32 MFMA form (8 waves; w_img in the tap-pair layout), by default with 4 loader waves issuing its halo DMA
        static const bool lw = getenv("TEDSPAD_STEM_LOADERS") == nullptr || atoi(getenv("TEDSPAD_STEM_LOADERS")) != 0;       // A/B knob: 0 = the 8-wave form (8 loader waves measured the same as 4)
        static thread_local int attr16[4] = {0, 0, 0, 0};
        const int ti = (dtype == TEDSPAD_F16 ? 0 : 1) + (lw ? 2 : 0);
        const void *fn16 = ti == 0 ? (const void *)conv_stem_pt_kernel<F16, 8, true, 16> : ti == 1 ? (const void *)conv_stem_pt_kernel<BF16, 8, true, 16>
                         : ti == 2 ? (const void *)conv_stem_pt_kernel<F16, 8, true, 16, false, 4> : (const void *)conv_stem_pt_kernel<BF16, 8, true, 16, false, 4>;
        if (!attr16[ti]) {
            if (hipFuncSetAttribute(fn16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("%s: cannot raise the dynamic LDS limit", who);
                return TEDSPAD_ELAUNCH;
            }
            attr16[ti] = 1;
        }
        if (ti == 0) hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, true, 16>), dim3(grid), dim3(512), pt_lds_pool(16), s, p);
        else if (ti == 1) hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, true, 16>), dim3(grid), dim3(512), pt_lds_pool(16), s, p);
        else if (ti == 2) hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, true, 16, false, 4>), dim3(grid), dim3(768), pt_lds_pool(16), s, p);
        else hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, true, 16, false, 4>), dim3(grid), dim3(768), pt_lds_pool(16), s, p);
    } else {
    const int di = ((dtype == TEDSPAD_F16 ? 0 : 1) * 2 + w8) * 2 + (pool ? 1 : 0);
    const void *fns[8] = {(const void *)conv_stem_pt_kernel<F16, 4, false>, (const void *)conv_stem_pt_kernel<F16, 4, true>,
                          (const void *)conv_stem_pt_kernel<F16, 8, false>, (const void *)conv_stem_pt_kernel<F16, 8, true>,
                          (const void *)conv_stem_pt_kernel<BF16, 4, false>, (const void *)conv_stem_pt_kernel<BF16, 4, true>,
                          (const void *)conv_stem_pt_kernel<BF16, 8, false>, (const void *)conv_stem_pt_kernel<BF16, 8, true>};
    if (!attr_set[di]) {
        if (hipFuncSetAttribute(fns[di], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("%s: cannot raise the dynamic LDS limit", who);
            return TEDSPAD_ELAUNCH;
        }
        attr_set[di] = 1;
    }
    const dim3 g(grid), b(w8 ? 512 : 256);
    const int lds = pool ? PT_LDS_POOL : PT_LDS;
    switch (di) {
        case 0: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 4, false>), g, b, lds, s, p); break;
        case 1: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 4, true>), g, b, lds, s, p); break;
        case 2: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, false>), g, b, lds, s, p); break;
        case 3: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 8, true>), g, b, lds, s, p); break;
        case 4: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 4, false>), g, b, lds, s, p); break;
        case 5: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 4, true>), g, b, lds, s, p); break;
        case 6: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, false>), g, b, lds, s, p); break;
        default: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 8, true>), g, b, lds, s, p); break;
    }
    }
    int32_t rc = check_launch(who);
    if (rc != TEDSPAD_OK || !pool || p.tiles_w < 2) return rc;
    const long rows = (long)n * t_pairs * hp;
    const long pieces = rows * (p.tiles_w - 1) * 8;
    const dim3 fg((unsigned)((pieces + 255) / 256));
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(stem_pool_fix_kernel<F16>, fg, dim3(256), 0, s, (uint16_t *)y, (const uint16_t *)side, rows, wp, p.tiles_w, ldy);
    else hipLaunchKernelGGL(stem_pool_fix_kernel<BF16>, fg, dim3(256), 0, s, (uint16_t *)y, (const uint16_t *)side, rows, wp, p.tiles_w, ldy);
    return check_launch(who);
}

extern "C" int32_t tedspad_stem_pt_fwd(const void *x_tp, const void *w_img, const float *scale, const float *shift, void *y, int32_t n, int32_t t_pairs,
                                       int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t ldy, int32_t relu, int32_t nwg,
                                       int32_t variant, int32_t dtype, void *stream) {
    TS_REQUIRE(x_tp && w_img && scale && shift && y && n > 0 && t_pairs > 0 && h > 0 && w > 0 && w % 2 == 0 && ho > 0 && wo > 0, "tedspad_stem_pt_fwd: bad arguments");
    TS_REQUIRE(ho == (h + 1) / 2 && wo == w / 2, "tedspad_stem_pt_fwd: 7x7 stride-2 pad-3 geometry (ho = ceil(h / 2))");
    TS_REQUIRE(ldy >= 64 && ldy % 8 == 0 && ((uintptr_t)x_tp | (uintptr_t)w_img | (uintptr_t)y) % 16 == 0, "tedspad_stem_pt_fwd: 64 output channels, 16-byte aligned pointers");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_stem_pt_fwd: bad dtype");
    TS_REQUIRE((long)h * w * PT_REC < (1L << 31), "tedspad_stem_pt_fwd: frame too large for 32-bit halo offsets");
    return stem_pt_launch("tedspad_stem_pt_fwd", x_tp, w_img, scale, shift, y, nullptr, n, t_pairs, h, w, ho, wo, 0, 0, ldy, relu, nwg, variant, dtype, false,
                          (hipStream_t)stream);
}

extern "C" int64_t tedspad_stem_pt_side_bytes(int32_t n, int32_t t_pairs, int32_t h, int32_t w) {
    const int ho = (h + 1) / 2, wo = w / 2;
    if (n <= 0 || t_pairs <= 0 || ho < 3 || wo < 3) return 0;
    return (int64_t)n * t_pairs * ((ho - 3) / 2 + 1) * ((wo + PT_TW - 1) / PT_TW) * 64 * 2;
}

// conv1 + bn1 + ReLU + the WHOLE maxpool1 (2 x 3 x 3 window, stride 2, no padding; large_i3d.py:133-138,229-232):
// y[n][t_pairs][hp][wp][ldy], hp = (ho - 3) / 2 + 1, wp = (wo - 3) / 2 + 1; side: tedspad_stem_pt_side_bytes() of scratch.
extern "C" int32_t tedspad_stem_pt_pool_fwd(const void *x_tp, const void *w_img, const float *scale, const float *shift, void *y, void *side, int32_t n,
                                            int32_t t_pairs, int32_t h, int32_t w, int32_t hp, int32_t wp, int32_t ldy, int32_t nwg, int32_t variant,
                                            int32_t dtype, void *stream) {
    TS_REQUIRE(x_tp && w_img && scale && shift && y && side && n > 0 && t_pairs > 0 && h > 0 && w > 0 && w % 2 == 0, "tedspad_stem_pt_pool_fwd: bad arguments");
    const int ho = (h + 1) / 2, wo = w / 2;
    TS_REQUIRE(ho >= 3 && wo >= 3 && hp == (ho - 3) / 2 + 1 && wp == (wo - 3) / 2 + 1, "tedspad_stem_pt_pool_fwd: 3x3 stride-2 unpadded pool of the ceil(h/2) x w/2 stem output");
    TS_REQUIRE(ldy >= 64 && ldy % 8 == 0 && ((uintptr_t)x_tp | (uintptr_t)w_img | (uintptr_t)y | (uintptr_t)side) % 16 == 0, "tedspad_stem_pt_pool_fwd: 64 output channels, 16-byte aligned pointers");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_stem_pt_pool_fwd: bad dtype");
    TS_REQUIRE((long)h * w * PT_REC < (1L << 31), "tedspad_stem_pt_pool_fwd: frame too large for 32-bit halo offsets");
    return stem_pt_launch("tedspad_stem_pt_pool_fwd", x_tp, w_img, scale, shift, y, side, n, t_pairs, h, w, ho, wo, hp, wp, ldy, 1, nwg, variant, dtype, true,
                          (hipStream_t)stream);
}

// The same launch WITHOUT the layout pass: x is the fp32 (n, c, t, h, w) clip batch the boundary hands over (element strides sn, sc, st, sh; sw = 1), read by the
// stem's own loader (conv1 -> bn1 -> relu -> maxpool1 on the NCTHW tensor: large_i3d.py:229-232; its producer is dali_extraction.py:38-50). w_img16: the
// tap-pair image of the 16x16x32 form. Rows must be 16-byte aligned (w % 4 == 0, every stride a multiple of 4 elements).
extern "C" int32_t tedspad_stem_pt_pool_clip_fwd(const float *x, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc, int64_t st, int64_t sh,
                                                 int64_t sw, int32_t pad_t, int32_t stride_t, int32_t t_pairs, const void *w_img16, const float *scale,
                                                 const float *shift, void *y, void *side, int32_t hp, int32_t wp, int32_t ldy, int32_t nwg, int32_t variant,
                                                 int32_t dtype, void *stream) {
    TS_REQUIRE(x && w_img16 && scale && shift && y && side && n > 0 && c > 0 && c <= 3 && t > 0 && h > 0 && w > 0 && t_pairs > 0 && pad_t >= 0,
               "tedspad_stem_pt_pool_clip_fwd: bad arguments");
    TS_REQUIRE(stride_t == 2, "tedspad_stem_pt_pool_clip_fwd: temporal stride 2");
    TS_REQUIRE(h < 4000 && w < 8000, "tedspad_stem_pt_pool_clip_fwd: frames up to 4000 x 8000 (the loader's packed task word)");
    TS_REQUIRE(sw == 1 && w % 4 == 0 && ((uintptr_t)x | (uintptr_t)(sn * 4) | (uintptr_t)(sc * 4) | (uintptr_t)(st * 4) | (uintptr_t)(sh * 4)) % 16 == 0,
               "tedspad_stem_pt_pool_clip_fwd: rows must be contiguous and 16-byte aligned (w %% 4 == 0, strides multiples of 4 elements)");
    TS_REQUIRE(sc >= 0 && st >= 0 && sh >= 0 && sn >= 0 && (c - 1) * sc + (long)(t + 8) * st + (long)(h + 32) * sh + w + 64 < (1L << 31),
               "tedspad_stem_pt_pool_clip_fwd: a sample must fit 32-bit element offsets");
    const int ho = (h + 1) / 2, wo = w / 2;
    TS_REQUIRE(ho >= 3 && wo >= 3 && hp == (ho - 3) / 2 + 1 && wp == (wo - 3) / 2 + 1, "tedspad_stem_pt_pool_clip_fwd: 3x3 stride-2 unpadded pool of the ceil(h/2) x w/2 stem output");
    TS_REQUIRE(ldy >= 64 && ldy % 8 == 0 && ((uintptr_t)w_img16 | (uintptr_t)y | (uintptr_t)side) % 16 == 0, "tedspad_stem_pt_pool_clip_fwd: 64 output channels, 16-byte aligned pointers");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_stem_pt_pool_clip_fwd: bad dtype");
    StemPT d;
    d.xf = x; d.sN = (long)sn; d.sC = (int)sc; d.sT = (int)st; d.sHf = (int)sh; d.C = c; d.T = t; d.pad_t = pad_t;
    return stem_pt_launch("tedspad_stem_pt_pool_clip_fwd", nullptr, w_img16, scale, shift, y, side, n, t_pairs, h, w, ho, wo, hp, wp, ldy, 1, nwg, variant | 6, dtype, true,
                          (hipStream_t)stream, &d);
}
