// Ping-pong implicit-GEMM convolution for gfx950 (tile_cfg 25): 256 pixels x 256 output channels per workgroup, 8 waves
// (two per SIMD), for convolutions with cin % 64 == 0 (no K table) and cout % 256 == 0.
//
// Why a second structure: every tile of conv_igemm.hip runs its steady-state loop at about half the MFMA rate
// ("Main loop of the generic kernel, measured in cycles") because the two waves of a SIMD run the SAME
// program between the same barriers: both issue their LDS fragment reads, both wait for them, both want the matrix
// pipe. Here the two waves of a SIMD (wave w and wave w + 4) are one barrier apart: a K tile is cut into four PHASES,
// a phase is a LOAD segment (fragment reads of one quadrant + two LDS-DMA instructions of the tile stream) followed by
// a COMPUTE segment (eight 32x32x16 MFMAs), every segment ends in a raw s_barrier, and waves 4-7 take one extra barrier
// before the loop -- so while one wave of a SIMD multiplies, its partner reads and issues DMA, and the matrix pipe
// always has a wave whose operands are already in registers.
//
//   wave (g, wn) = (wave >> 2, wave & 3) owns pixels {64g..64g+63} u {128+64g..128+64g+63} x channels {32wn..+31} u
//   {128+32wn..+31}: its operands of one K tile are two 64-row pixel fragments sets X0, X1 and two 32-row weight
//   fragment sets W0, W1, each living in its own 16 KB staging UNIT (128 rows x 64 k, the 128-byte-row XOR-swizzled
//   image of conv_igemm.hip):  Xh0 = pixel rows 0..127, Xh1 = 128..255, Wh0 = channel rows 0..127, Wh1 = 128..255.
//     phase 1: read X0 (8 x ds_read_b128)            acc[W0][X0] += ...
//     phase 2: read W1 (4)                           acc[W1][X0]
//     phase 3: read X1 (8)                           acc[W1][X1]
//     phase 4: read W0 of the NEXT tile (4)          acc[W0][X1]      (two W0 register sets; evens out the segments)
//   Every accumulator sums K in the same order as the generic tiles: results are bit-identical to them.
//
// Tile stream: 8 unit slots (two K tiles), one unit staged per phase by all 512 threads (2 LDS-DMA instructions each),
// in the order [phase 1: Wh1(t+1), phase 2: Xh1(t+1), phase 3: Wh0(t+2), phase 4: Xh0(t+2)] -- the gather arithmetic of
// the pixel units sits in the phases with few fragment reads. A unit is staged >= 5 phases before the phase that reads
// it and >= 2 phases after the last read of the slot's previous occupant, and every phase ends with a counted
// s_waitcnt vmcnt(6): all but the three youngest units of this wave have landed. Ordering argument (segments are the
// barrier-delimited intervals; group 0 runs LOAD of phase k in segment 2k, group 1 in segment 2k+1):
//   RAW  a unit read in phase n was waited for (by every wave) in phase <= n-2, i.e. before the barrier that ends
//        segment 2(n-2)+2 = 2n-2, and the earliest read of it is issued in segment 2n;
//   WAR  reads of phase k have returned (compiler-counted lgkmcnt before the MFMAs that use them) by the end of
//        segment 2k+2; DMA into their slot is issued in phase >= k+2, i.e. in segment >= 2k+4.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace tedspad {
namespace {

__device__ uint4 g_zero16p;   // zero page for padded taps / rows past M
#ifdef TEDSPAD_P8_ABLATIONS
__device__ unsigned long long *g_dbg_p8;   // debug builds: [workgroup][6] cycle stamps (entry, prologue done, loop done, pass 0 stored, pass 1 stored, stores retired)
#define P8_STAMP(i) do { if (g_dbg_p8 && threadIdx.x == 0) g_dbg_p8[(size_t)blockIdx.x * 6 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int fdiv_p(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

__device__ __forceinline__ void keep4(const uint4 &v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

// ABL: ablation bits for scripts/p8_check.py (TEDSPAD_P8_ABL; 0 in production): 1 no fragment reads, 2 no DMA in the loop,
// 4 no MFMAs, 8 no gather arithmetic (pixel units staged from linear addresses), 16 no swizzle on the DMA source addresses
// (lane-linear 128-byte rows: tests whether the permuted lanes cost address-coalescing) -- results are wrong with any bit set.
// MF: MFMA shape. 32: v_mfma_f32_32x32x16 (K summed in the order of the generic tiles: bit-identical to them); 16:
// v_mfma_f32_16x16x32 -- same fragment bytes, same MFMA cycles per K tile, but the chip holds a higher clock on this shape
// under load (MI355X guide, DVFS: ~1.13x), which speeds up every cycle of the kernel; fp32 sums are re-associated (tile_cfg 26).
// DMAC = 1 (experiment, not instantiated): the LDS-DMA instructions of a phase issued from the COMPUTE segment, between its MFMAs.
// Measured 10-25 % SLOWER on every layer shape (and 5-10x slower on two of them): a DMA instruction holds the wave for its whole
// queueing time at the CU's address unit, and the MFMAs behind it in program order wait with it. The load segment is where it belongs.
// DUAL: K-concatenated pair of pointwise convs with two sources (tedspad_conv_p8_dual_fwd): K tiles [0, nk1) gather from x (1x1x1,
// stride 1), K tiles [nk1, nk) from x2 (1x1x1 with spatial stride (sh2, sw2) on an (Hi2, Wi2) grid) -- conv3 and the strided downsample
// conv of the first bottleneck of layer2/3/4 (large_i3d.py:77-84) as ONE GEMM over [W3*s3 | Wd*sd]: the downsample tensor is never
// written or re-read and its launch disappears.
template <typename T, int ABL, int MF = 32, int DMAC = 0, bool DUAL = false>
__global__ __launch_bounds__(512) void conv_p8_kernel(const ConvKP p) {
    constexpr int BM = 256, BN = 256;
    constexpr int ROWB = BK * 2;                 // 128-byte rows
    constexpr int UNIT = 128 * ROWB;             // 16 KB
    constexpr int LDS_MAIN = 8 * UNIT;
    constexpr int LDS_STAGE = BM * (BN + 8) * 2;   // the epilogue's 16-bit staging tile
    constexpr int LDS_BYTES = LDS_MAIN > LDS_STAGE ? LDS_MAIN : LDS_STAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % p.tiles_n;
    const int tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    P8_STAMP(0);

    // ---- DMA roles: 8 consecutive lanes fetch the eight 16-byte chunks of one unit row; a wave instruction covers 8 rows,
    // the block's two instructions per unit cover rows [wave*8 + lane/8] and [64 + ...] --------------------------------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (ABL & 16) ? (lane & 7) : (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);   // SOURCE chunk (swizzle (row >> 1) & 7 on the source)
    int a_base[4], a_base2[DUAL ? 4 : 1];
    unsigned a_mask[4];   // bits 0..6 valid dt, 8..14 valid dh, 16..22 valid dw
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + i * 64 + rsub;       // i = 2*half + slot
        a_base[i] = 0;
        a_mask[i] = 0;
        if (DUAL) a_base2[i] = 0;
        if (DUAL && m < p.M) {
            const int r1 = fdiv_p(m, p.Wo, p.inv_wo), wo = m - r1 * p.Wo;
            const int r2 = fdiv_p(r1, p.Ho, p.inv_ho), ho = r1 - r2 * p.Ho;      // r2 = n * To + to (no temporal stride)
            a_base2[i] = ((r2 * p.Hi2 + ho * p.sh2) * p.Wi2 + wo * p.sw2) * p.ldx2;
        }
        if (m < p.M) {
            if (p.pointwise) {
                a_base[i] = m * p.ldx;
                a_mask[i] = 0x010101u;
            } else {
                int wo, ho, to, n;
                if (p.M < (1 << 24)) {
                    const int r1 = fdiv_p(m, p.Wo, p.inv_wo); wo = m - r1 * p.Wo;
                    const int r2 = fdiv_p(r1, p.Ho, p.inv_ho); ho = r1 - r2 * p.Ho;
                    n = fdiv_p(r2, p.To, p.inv_to); to = r2 - n * p.To;
                } else {
                    wo = m % p.Wo; const int r1 = m / p.Wo;
                    ho = r1 % p.Ho; const int r2 = r1 / p.Ho;
                    to = r2 % p.To; n = r2 / p.To;
                }
                const int t0 = to * p.st - p.pt, h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
                a_base[i] = (((n * p.Ti + t0) * p.Hi + h0) * p.Wi + w0) * p.ldx;
                unsigned mk = 0;
                for (int d = 0; d < p.kt; ++d) mk |= ((unsigned)(t0 + d) < (unsigned)p.Ti ? 1u : 0u) << d;
                for (int d = 0; d < p.kh; ++d) mk |= ((unsigned)(h0 + d) < (unsigned)p.Hi ? 1u : 0u) << (8 + d);
                for (int d = 0; d < p.kw; ++d) mk |= ((unsigned)(w0 + d) < (unsigned)p.Wi ? 1u : 0u) << (16 + d);
                a_mask[i] = mk;
            }
        }
    }
    const uint16_t *wsrc = p.w + (size_t)(n0 + rsub) * p.Kpad + kc * 8;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16p);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned ldsw = lds0 + wave * 8 * ROWB;      // this wave's 1 KiB piece inside a 64-row half unit

    // K tile -> tap: wave-uniform counters advancing with the staging order (cin % 64 == 0: a tile never straddles a tap)
    int u_dt = 0, u_dh = 0, u_dw = 0, u_c = 0;
    int u_tile = 0;
    auto next_entry = [&]() -> int2 {
        int2 e;
        if (DUAL) {      // both sources are pointwise: K tile -> channel offset inside its source; bit 30 of e.y marks the second source
            const bool second = u_tile >= p.nk1;
            e.x = (second ? u_tile - p.nk1 : u_tile) * BK + kc * 8;
            e.y = (8 << 8) | (16 << 16) | (second ? (1 << 30) : 0);
            ++u_tile;
            return e;
        }
        e.x = ((u_dt * p.Hi + u_dh) * p.Wi + u_dw) * p.ldx + u_c + kc * 8;
        e.y = u_dt | ((8 + u_dh) << 8) | ((16 + u_dw) << 16);
        u_c += BK;
        if (u_c == p.cin) { u_c = 0; if (++u_dw == p.kw) { u_dw = 0; if (++u_dh == p.kh) { u_dh = 0; ++u_dt; } } }
        return e;
    };
    // unit slots: slot = (tile & 1) * 4 + {0: Xh0, 1: Wh0, 2: Wh1, 3: Xh1}
    auto stage_x = [&](int2 e, int half, int slot) {
        const unsigned dst = ldsw + slot * UNIT;
        const unsigned s_t = e.y & 255, s_h = (e.y >> 8) & 255, s_w = ((unsigned)e.y >> 16) & 255;
        const bool second = DUAL && (e.y & (1 << 30));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned mk = a_mask[half * 2 + i];
            const unsigned ok = (mk >> s_t) & (mk >> s_h) & (mk >> s_w) & 1u;
            const uint16_t *src = (ABL & 8) ? p.x + (ptrdiff_t)(a_base[half * 2 + i] + (e.x & 1023)) : !ok ? zero :
                                  second ? p.x2 + (ptrdiff_t)(a_base2[DUAL ? half * 2 + i : 0] + e.x) : p.x + (ptrdiff_t)(a_base[half * 2 + i] + e.x);
            lds_dma16(src, dst + i * 64 * ROWB);
        }
    };
    auto stage_w = [&](int kt, int half, int slot) {
        const unsigned dst = ldsw + slot * UNIT;
        const uint16_t *src = wsrc + (size_t)(half * 128) * p.Kpad + kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(src + (size_t)(i * 64) * p.Kpad, dst + i * 64 * ROWB);
    };

    // ---- MFMA roles ------------------------------------------------------------------------------------------------
    const int grp = wave >> 2, wn = wave & 3;
    constexpr int FR = MF == 32 ? 32 : 16;            // rows per fragment
    constexpr int NKS = MF == 32 ? 4 : 2;             // K sub-steps per K tile (k16 | k32)
    constexpr int NXF = 64 / FR, NWF = 32 / FR;       // fragments per 64-pixel / 32-channel set
    const int l31 = lane & 31, lh = lane >> 5;        // MF 32: row, k half
    const int l15 = lane & 15, lq = lane >> 4;        // MF 16: row, k quarter
    const int frow = MF == 32 ? l31 : l15;
    const int swz = (frow >> 1) & 7;
    const unsigned xrow = (64 * grp + frow) * ROWB;      // byte offset of this lane's pixel row inside an X unit (+ FR rows per fragment)
    const unsigned wrow = (32 * wn + frow) * ROWB;       //                          weight row inside a W unit
    unsigned coff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) coff[ks] = MF == 32 ? ((((ks << 1) | lh) ^ swz) << 4) : ((((ks << 2) | lq) ^ swz) << 4);

    typedef typename std::conditional<MF == 32, f32x16, f32x4>::type acc_t;
    acc_t acc[2][2 * NXF][NWF];     // [W0 | W1][X0 fragments, X1 fragments][channel fragments]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2 * NXF; ++b)
#pragma unroll
            for (int c = 0; c < NWF; ++c)
#pragma unroll
                for (int r = 0; r < (MF == 32 ? 16 : 4); ++r) acc[a][b][c][r] = 0.f;

    // ---- prologue: six units (all of tile 0, Wh0 / Xh0 of tile 1) in the order the loop continues ---------------------
    int2 eA = next_entry();          // tile 0
    stage_w(0, 0, 1);
    stage_x(eA, 0, 0);
    stage_w(0, 1, 2);
    stage_x(eA, 1, 3);
    eA = next_entry();               // tile 1 (nk >= 2 is a launch requirement)
    stage_w(1, 0, 5);
    stage_x(eA, 0, 4);
    wait_vmcnt<6>();                 // Wh0(0), Xh0(0), Wh1(0) of this wave have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    P8_STAMP(1);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);

    uint4 fx[NXF][NKS], fwa[NWF][NKS], fwb[NWF][NKS], fw1[NWF][NKS];    // W0 lives in two register sets: tile t+1's is read in phase 4 of tile t
    if (ABL & 1) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int c = 0; c < NWF; ++c) { fwa[c][ks] = make_uint4(lane, ks, 3, 4); fwb[c][ks] = fwa[c][ks]; fw1[c][ks] = make_uint4(ks, lane, 1, 2); }
#pragma unroll
            for (int f = 0; f < NXF; ++f) fx[f][ks] = make_uint4(1, 2, lane, ks + f);
        }
    }
    const int nk = p.nk;
#define P8_SEG_END()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0)
#define P8_WAIT(N_STEADY, N_LAST2)                                   \
    if (t + 1 >= nk) wait_vmcnt<0>();                                \
    else if (t + 2 >= nk) wait_vmcnt<N_LAST2>();                     \
    else wait_vmcnt<N_STEADY>()
#define P8_READ_X(U)                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks)                                                                  \
        _Pragma("unroll") for (int f = 0; f < NXF; ++f)                                                                 \
            if (!(ABL & 1)) fx[f][ks] = *reinterpret_cast<const uint4 *>(cur + (U) * UNIT + xrow + f * FR * ROWB + coff[ks])
#define P8_READ_W(FW, BASE)                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks)                                                                  \
        _Pragma("unroll") for (int c = 0; c < NWF; ++c)                                                                 \
            if (!(ABL & 1)) FW[c][ks] = *reinterpret_cast<const uint4 *>((BASE) + wrow + c * FR * ROWB + coff[ks])
#define P8_MFMA(J, XB0, FW, STAGE)                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                                \
        if (DMAC && ks == 1) { __builtin_amdgcn_sched_barrier(0); STAGE; __builtin_amdgcn_sched_barrier(0); }           \
        _Pragma("unroll") for (int f = 0; f < NXF; ++f)                                                                 \
            _Pragma("unroll") for (int c = 0; c < NWF; ++c) {                                                           \
                if (ABL & 4) { keep4(FW[c][ks]); keep4(fx[f][ks]); }                                                    \
                else if constexpr (MF == 32) acc[J][XB0 + f][c] = T::mfma(FW[c][ks], fx[f][ks], acc[J][XB0 + f][c]);    \
                else acc[J][XB0 + f][c] = T::mfma16(FW[c][ks], fx[f][ks], acc[J][XB0 + f][c]);                          \
            }                                                                                                           \
    }                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                      \
    __builtin_amdgcn_sched_barrier(0)

    // W0 of tile 0 (Wh0(0) was waited for above and the barrier has published it)
    P8_READ_W(fwa, smem + 1 * UNIT);
    __builtin_amdgcn_sched_barrier(0);

    auto ktile = [&](const int t, uint4 (&fw0)[NWF][NKS], uint4 (&fw0n)[NWF][NKS]) {
        const unsigned char *cur = smem + (t & 1) * 4 * UNIT;
        const int nxt = ((t + 1) & 1) * 4, nn = (t & 1) * 4;     // slot bases of tile t+1 / tile t+2
        // ---- phase 1: X0 | Wh1(t+1) ----
        P8_READ_X(0);
        if (!DMAC && !(ABL & 2) && t + 1 < nk) stage_w(t + 1, 1, nxt + 2);
        P8_SEG_END();
        P8_MFMA(0, 0, fw0, if (!(ABL & 2) && t + 1 < nk) stage_w(t + 1, 1, nxt + 2));
        P8_WAIT(6, 6);
        P8_SEG_END();
        // ---- phase 2: W1 | Xh1(t+1) ----
        P8_READ_W(fw1, cur + 2 * UNIT);
        if (!DMAC && !(ABL & 2) && t + 1 < nk) stage_x(eA, 1, nxt + 3);
        P8_SEG_END();
        P8_MFMA(1, 0, fw1, if (!(ABL & 2) && t + 1 < nk) stage_x(eA, 1, nxt + 3));
        P8_WAIT(6, 6);
        P8_SEG_END();
        // ---- phase 3: X1 | Wh0(t+2) ----
        P8_READ_X(3);
        if (!DMAC && !(ABL & 2) && t + 2 < nk) stage_w(t + 2, 0, nn + 1);
        P8_SEG_END();
        P8_MFMA(1, NXF, fw1, if (!(ABL & 2) && t + 2 < nk) stage_w(t + 2, 0, nn + 1));
        P8_WAIT(6, 4);
        P8_SEG_END();
        // ---- phase 4: W0 of tile t+1 | Xh0(t+2) ----
        if (t + 1 < nk) { P8_READ_W(fw0n, smem + (nxt + 1) * UNIT); }
        if (!DMAC && !(ABL & 2) && t + 2 < nk) {
            eA = next_entry();
            stage_x(eA, 0, nn + 0);
        }
        P8_SEG_END();
        P8_MFMA(0, NXF, fw0, if (!(ABL & 2) && t + 2 < nk) { eA = next_entry(); stage_x(eA, 0, nn + 0); });
        P8_WAIT(6, 2);
        P8_SEG_END();
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(t, fwa, fwb);
        if (t + 1 < nk) ktile(t + 1, fwb, fwa);
    }
#undef P8_MFMA
#undef P8_READ_X
#undef P8_READ_W
#undef P8_SEG_END
#undef P8_WAIT
    if (grp == 0) __builtin_amdgcn_s_barrier();   // pairs with the last in-loop barrier of waves 4-7
    asm volatile("" ::: "memory");
    __syncthreads();   // ring free: reused as the staging tile
    P8_STAMP(2);

    // ---- epilogue: scale / shift (+ residual) / ReLU / rounding in the accumulator layout, ONE pass of 16-bit rows through LDS ([256 px][256 ch], 528-byte
    // rows), then whole 512-byte rows out. (The fp32 two-pass form of rounds 2-4 moved 2 x 128 KB in and out of LDS behind three barriers: 15 k cycles per workgroup
    // without a residual, 27 k with one -- as long as the whole K = 512 loop of layer4's conv3: profiles/r05_p8_stamps.md.) With a residual its rows are
    // requested first (coalesced 16-byte loads, the fragment registers are dead), land in the staging tile in the ROW layout, and every lane then reads back the
    // four residual values of each of its accumulator quads, forms acc * scale + shift + residual in fp32 -- the same operations in the same order as the
    // two-pass form: bit-identical results -- and overwrites them in place with the rounded result.
    {
        constexpr int S16 = 256 + 8;                        // 16-bit elements per staged row
        static_assert(BM * S16 * 2 <= LDS_BYTES, "16-bit staging tile");
        uint16_t *stg16 = reinterpret_cast<uint16_t *>(smem);
        const float lo = p.relu ? 0.f : (T::kDtype == TEDSPAD_F16 ? -p.sat : -3.3e38f);
        auto pack2 = [&](float a, float b) -> unsigned {
            if constexpr (T::kDtype == TEDSPAD_F16) {
                unsigned pk;
                const float x = __builtin_amdgcn_fmed3f(a, lo, p.sat), y = __builtin_amdgcn_fmed3f(b, lo, p.sat);
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(y));
                return pk;
            } else {
                return (unsigned)T::from_f32(__builtin_fmaxf(a, lo)) | ((unsigned)T::from_f32(__builtin_fmaxf(b, lo)) << 16);
            }
        };
        const bool has_res = p.res != nullptr;
        if (has_res) {
            const int cc = tid & 31, r0 = tid >> 5;         // 32 chunks of 8 channels per row, 16 rows per sweep
            uint4 rr[BM / 16];
#pragma unroll
            for (int it = 0; it < BM / 16; ++it) {
                const int m = m0 + r0 + it * 16;
                rr[it] = make_uint4(0u, 0u, 0u, 0u);
                if (m < p.M) rr[it] = *reinterpret_cast<const uint4 *>(p.res + (size_t)m * p.ldres + n0 + cc * 8);
            }
#pragma unroll
            for (int it = 0; it < BM / 16; ++it) *reinterpret_cast<uint4 *>(stg16 + (r0 + it * 16) * S16 + cc * 8) = rr[it];
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int nb = n0 + 128 * j;                          // folded output frames: scale / shift are indexed inside the frame
            if (p.fold_hw) nb -= (nb / p.fold_c) * p.fold_c;
            constexpr int NG = MF == 32 ? 4 : NWF;          // 4-channel groups of this lane per accumulator set
            f32x4 sc[NG], sh[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cb = nb + 32 * wn + (MF == 32 ? 8 * g + 4 * lh : 16 * g + 4 * lq);
                sc[g] = *reinterpret_cast<const f32x4 *>(p.scale + cb);
                sh[g] = *reinterpret_cast<const f32x4 *>(p.shift + cb);
            }
#pragma unroll
            for (int b = 0; b < 2 * NXF; ++b) {
                const int ml = (b / NXF) * 128 + 64 * grp + (b % NXF) * FR + frow;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int ch = 128 * j + 32 * wn + (MF == 32 ? 8 * g + 4 * lh : 16 * g + 4 * lq);
                    uint2 *slot = reinterpret_cast<uint2 *>(stg16 + ml * S16 + ch);
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float a;
                        if constexpr (MF == 32) a = acc[j][b][0][4 * g + i]; else a = acc[j][b][g][i];
                        v[i] = __builtin_fmaf(a, sc[g][i], sh[g][i]);
                    }
                    if (has_res) {
                        const uint2 rv = *slot;
                        v[0] += T::to_f32((uint16_t)(rv.x & 0xffffu)); v[1] += T::to_f32((uint16_t)(rv.x >> 16));
                        v[2] += T::to_f32((uint16_t)(rv.y & 0xffffu)); v[3] += T::to_f32((uint16_t)(rv.y >> 16));
                    }
                    *slot = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                }
            }
        }
        __syncthreads();
        P8_STAMP(3);
        const int cc = tid & 31, r0 = tid >> 5;             // 32 chunks of 8 channels per row, 16 rows per sweep
        int nsel = 0, nloc = n0 + cc * 8;
        if (p.fold_hw) { nsel = nloc / p.fold_c; nloc -= nsel * p.fold_c; }
#pragma unroll 4
        for (int it = 0; it < BM / 16; ++it) {
            const int r = r0 + it * 16;
            const int m = m0 + r;
            if (m < p.M) {
                const uint4 v = *reinterpret_cast<const uint4 *>(stg16 + r * S16 + cc * 8);
                size_t orow = (size_t)m;
                if (p.fold_hw) orow += (size_t)(m / p.fold_hw) * (size_t)((p.fold_f - 1) * p.fold_hw) + (size_t)nsel * p.fold_hw;
                *reinterpret_cast<uint4 *>(p.y + orow * p.ldy + nloc) = v;
            }
        }
        P8_STAMP(4);
#ifdef TEDSPAD_P8_ABLATIONS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P8_STAMP(5);
#endif
    }
}

template <typename T>
int32_t launch_p8(const ConvKP &p, hipStream_t s, int mf) {
    ConvKP q = p;
    q.tiles_n = p.Cout / 256;
    const int tiles_m = (p.M + 255) / 256;
    const dim3 grid(tiles_m * q.tiles_n), block(512);
    if (p.x2) {
        hipLaunchKernelGGL((conv_p8_kernel<T, 0, 32, 0, true>), grid, block, 0, s, q);
        return check_launch("tedspad_conv_p8_dual_fwd");
    }
    if (mf == 16) {
        hipLaunchKernelGGL((conv_p8_kernel<T, 0, 16>), grid, block, 0, s, q);
        return check_launch("tedspad_conv_fwd(p8/16)");
    }
#ifdef TEDSPAD_P8_ABLATIONS
    static const int abl = getenv("TEDSPAD_P8_ABL") ? atoi(getenv("TEDSPAD_P8_ABL")) : 0;
    switch (abl) {
        case 1: hipLaunchKernelGGL((conv_p8_kernel<T, 1>), grid, block, 0, s, q); break;
        case 2: hipLaunchKernelGGL((conv_p8_kernel<T, 2>), grid, block, 0, s, q); break;
        case 3: hipLaunchKernelGGL((conv_p8_kernel<T, 3>), grid, block, 0, s, q); break;
        case 4: hipLaunchKernelGGL((conv_p8_kernel<T, 4>), grid, block, 0, s, q); break;
        case 6: hipLaunchKernelGGL((conv_p8_kernel<T, 6>), grid, block, 0, s, q); break;
        case 7: hipLaunchKernelGGL((conv_p8_kernel<T, 7>), grid, block, 0, s, q); break;
        case 16: hipLaunchKernelGGL((conv_p8_kernel<T, 16>), grid, block, 0, s, q); break;
        case 20: hipLaunchKernelGGL((conv_p8_kernel<T, 20>), grid, block, 0, s, q); break;
        case 8: hipLaunchKernelGGL((conv_p8_kernel<T, 8>), grid, block, 0, s, q); break;
        default: hipLaunchKernelGGL((conv_p8_kernel<T, 0>), grid, block, 0, s, q); break;
    }
#else
    hipLaunchKernelGGL((conv_p8_kernel<T, 0>), grid, block, 0, s, q);
#endif
    return check_launch("tedspad_conv_fwd(p8)");
}

}  // namespace

}  // namespace tedspad
#ifdef TEDSPAD_P8_ABLATIONS
extern "C" int32_t tedspad_debug_set_p8_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tedspad::g_dbg_p8), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
namespace tedspad {

int32_t launch_conv_p8(int dtype, const ConvKP &p, hipStream_t s, int mf) {
    if (p.x2 && (!p.pointwise || p.res || p.nk1 < 1 || p.nk1 >= p.nk)) {
        set_error("tedspad_conv_p8_dual_fwd: two 1x1x1 convs (the first with stride 1), no residual");
        return TEDSPAD_EINVAL;
    }
    if (p.fold_hw && (p.res || p.x2 || p.fold_c % 128 != 0 || p.Cout % p.fold_c != 0)) {
        set_error("tedspad_conv_fwd_ex: folded output frames need fold_c %% 128 == 0, cout %% fold_c == 0, no residual");
        return TEDSPAD_EINVAL;
    }
    if (!p.utap || p.nk < 2 || p.Cout % 256 != 0 || p.sigmoid || p.mask || p.stats || p.ostrided || p.y32 || !p.y) {
        set_error("tedspad_conv_fwd: ping-pong config needs cin %% 64 == 0, K >= 128, cout %% 256 == 0 and the plain epilogue");
        return TEDSPAD_EINVAL;
    }
    return dtype == TEDSPAD_F16 ? launch_p8<F16>(p, s, mf) : launch_p8<BF16>(p, s, mf);
}

}  // namespace tedspad
