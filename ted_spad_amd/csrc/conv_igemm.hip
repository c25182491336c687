// Implicit-GEMM 3-D/2-D convolution for gfx950 (MI355X), channels-last, 16-bit operands,
// fp32 MFMA accumulation, fused BN scale/shift + residual + ReLU/sigmoid epilogue.
//
//   D[co][m] = sum_k W[co][k] * X[m][k],   m = (n,to,ho,wo),  k = (dt,dh,dw,ci)
//
// One kernel family serves every convolution of the hot path (reference call sites:
// Unit3D.forward aux_code/models/i3d.py:89-120, Bottleneck.forward
// aux_code/models/large_i3d.py:61-84, the two stems, DoubleConv unet_parts.py:8-25).
//
// Structure (round-1 measurements showed the first version was bound by the per-CU
// global->LDS rate, 16-24 GB/s/CU, not by MFMA): big tiles (up to 256 pixels x 128
// channels) for arithmetic intensity, and a 3-stage LDS ring filled by LDS-DMA
// (`global_load_lds_dwordx4`, 16 B/lane, no VGPR staging) that keeps one full K tile in
// flight across the single barrier per K step (counted s_waitcnt vmcnt, raw s_barrier):
//   * weights are the MFMA "A" operand (rows = output channels), activations the "B"
//     operand (columns = output pixels): each lane ends with 4 CONSECUTIVE channels of one
//     pixel per accumulator group -> 16-byte LDS writes in the epilogue;
//   * both operands live in LDS as [row][64] 16-bit tiles (128-byte rows); the DMA writes
//     lane-linear, so the XOR swizzle ((row>>1)&7 on the 16-byte chunk index) is applied to
//     the per-lane SOURCE address and again on the ds_read_b128 side -> conflict-free reads;
//   * the im2col gather is table driven: one {element offset, (dt,dh,dw) shift triple} per
//     8-channel K chunk, kept in LDS; every output pixel carries a packed per-dimension
//     validity mask, so any kernel/stride/asymmetric TF-SAME padding costs 3 shifts + 3 ands
//     per 16-byte load; padded taps (and rows past M) are redirected to a 16-byte zero page;
//   * the fp32 tile is staged through LDS so the final stores (and the residual loads)
//     are full 16-byte-per-lane coalesced rows of the NTHWC tensor.
#include "conv_common.h"
#include "det_gate.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16;
#ifdef TEDSPAD_DEBUG_TS
__device__ unsigned long long *g_dbg_ts_ig;   // debug builds only: [workgroup][4] = start, loop start, loop end, cycles waited in the loop (wave 0)
#endif              // zero page for padded taps (zero-initialised by the loader)

// n / d for 0 <= n < 2^24, d >= 1 via the fp32 reciprocal (both exactly representable), corrected to the exact quotient.
__device__ __forceinline__ int fdiv(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// KS = 2: split-K inside the workgroup (as in the stem, tile_cfg 21): a second set of WM*WN waves multiplies the k16
// sub-steps 2,3 of every 64-deep K step, the first set the sub-steps 0,1; partial sums meet in the staging tile. Twice the
// waves per SIMD for the same LDS: a single wave issues its MFMA groups at about half rate (dependent fragment reads).
template <typename T, int BM, int BN, int WM, int WN, int S, int KT, int KS = 1>
__global__ __launch_bounds__(WM *WN * 64 * KS) void conv_igemm_kernel(const ConvKP p) {
    constexpr int NT = WM * WN * 64 * KS;
    constexpr int RPS = NT / 8;           // tile rows filled by one DMA instruction slot of the block
    constexpr int SA = BM / RPS;          // DMA slots per thread per stage: activations
    constexpr int SW = BN / RPS;          //                                  weights
    constexpr int L = SA + SW;            // LDS-DMA instructions per thread per stage
    constexpr int TM = BM / WM / 32;      // 32-pixel tiles per wave
    constexpr int TN = BN / WN / 32;      // 32-channel tiles per wave
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr int STG_LD = BN + 4;        // fp32 staging row stride (floats): 16B-aligned, bank-skewed
    constexpr int LDS_MAIN = S * STAGE + KT;
    constexpr int LDS_STAGE = BM * STG_LD * 4;
    constexpr int LDS_BYTES = LDS_MAIN > LDS_STAGE ? LDS_MAIN : LDS_STAGE;
    static_assert(BM % RPS == 0 && BN % RPS == 0 && TM >= 1 && TN >= 1, "tile/wave geometry");
    static_assert(S >= 2 && (S - 2) * L <= 63, "vmcnt is a 6-bit counter");
    static_assert(2 * (NT / (BN / 8)) * BN * 4 <= LDS_BYTES, "batch-statistics reduction buffer must fit the staging area");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];  // ONE array: see cdna guide, glds traps
    int2 *ktab_lds = reinterpret_cast<int2 *>(smem + S * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % p.tiles_n;
    const int tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- DMA roles: 8 consecutive lanes fetch the eight 16-byte chunks of one tile row ------
    const int rsub = wave * 8 + (lane >> 3);                            // row inside a slot
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);  // SOURCE chunk (swizzle on the source)
    int a_base[SA];
    unsigned a_mask[SA];  // bits 0..6 valid dt, 8..14 valid dh, 16..22 valid dw
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int m = m0 + i * RPS + rsub;
        a_base[i] = 0;
        a_mask[i] = 0;
        if (m < p.M) {
            if (p.pointwise) {
                a_base[i] = m * p.ldx;
                a_mask[i] = 0x010101u;
            } else {
                int wo, ho, to, n;
                if (p.M < (1 << 24)) {   // exact in fp32: reciprocal estimate + one correction step instead of six integer divisions
                    const int r1 = fdiv(m, p.Wo, p.inv_wo); wo = m - r1 * p.Wo;
                    const int r2 = fdiv(r1, p.Ho, p.inv_ho); ho = r1 - r2 * p.Ho;
                    n = fdiv(r2, p.To, p.inv_to); to = r2 - n * p.To;
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
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;  // LDS byte address of the ring
    auto issue_a = [&](int2 e, int slot) {
        const unsigned stage = lds0 + slot * STAGE + wave * 8 * (BK * 2);
        const unsigned s_t = e.y & 255, s_h = (e.y >> 8) & 255, s_w = (unsigned)e.y >> 16;
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const unsigned ok = (a_mask[i] >> s_t) & (a_mask[i] >> s_h) & (a_mask[i] >> s_w) & 1u;
            const uint16_t *src = ok ? p.x + (ptrdiff_t)(a_base[i] + e.x) : zero;
            lds_dma16(src, stage + i * RPS * (BK * 2));
        }
    };
    auto issue_w = [&](int kt, int slot) {
        const unsigned stage = lds0 + slot * STAGE + wave * 8 * (BK * 2);
#pragma unroll
        for (int j = 0; j < SW; ++j)
            lds_dma16(wsrc + (size_t)(j * RPS) * p.Kpad + kt * BK, stage + BM * BK * 2 + j * RPS * (BK * 2));
    };
    // Uniform-tap layers (cin % 64 == 0: a 64-deep K tile never straddles a tap): the table entry of K tile kt is
    // plain arithmetic on wave-uniform counters that advance with the issue order -- no table loads in the prologue
    // (two dependent global-load latencies per workgroup, measured ~2-3 us of a ~4 us ramp).
    int u_dt = 0, u_dh = 0, u_dw = 0, u_c = 0;
    auto next_entry = [&]() -> int2 {
        int2 e;
        e.x = ((u_dt * p.Hi + u_dh) * p.Wi + u_dw) * p.ldx + u_c + kc * 8;
        e.y = u_dt | ((8 + u_dh) << 8) | ((16 + u_dw) << 16);
        u_c += BK;
        if (u_c == p.cin) { u_c = 0; if (++u_dw == p.kw) { u_dw = 0; if (++u_dh == p.kh) { u_dh = 0; ++u_dt; } } }
        return e;
    };
    auto issue = [&](int kt, int slot) {
        const unsigned stage = lds0 + slot * STAGE + wave * 8 * (BK * 2);
        const int2 e = p.utap ? next_entry() : ktab_lds[kt * 8 + kc];
        const unsigned s_t = e.y & 255, s_h = (e.y >> 8) & 255, s_w = (unsigned)e.y >> 16;
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const unsigned ok = (a_mask[i] >> s_t) & (a_mask[i] >> s_h) & (a_mask[i] >> s_w) & 1u;
            const uint16_t *src = ok ? p.x + (ptrdiff_t)(a_base[i] + e.x) : zero;
            lds_dma16(src, stage + i * RPS * (BK * 2));
        }
#pragma unroll
        for (int j = 0; j < SW; ++j)
            lds_dma16(wsrc + (size_t)(j * RPS) * p.Kpad + kt * BK, stage + BM * BK * 2 + j * RPS * (BK * 2));
    };

    // ---- MFMA roles ----------------------------------------------------------------------
    const int wsub = KS == 2 ? wave % (WM * WN) : wave;
    const int kh2 = KS == 2 ? wave / (WM * WN) : 0;     // which half of the k16 sub-steps this wave multiplies
    const int wm = wsub % WM, wn = wsub / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Epilogue constants, requested FIRST: loaded where they are used (after the staging barrier) their L2 round trip sat on
    // the critical path of every workgroup's epilogue; issued before any DMA they retire first (in order) and never disturb
    // the counted waits of the ring.
    constexpr int CPR = BN / 8;          // 16-byte output chunks per tile row
    constexpr int RPP = NT / CPR;        // rows per pass
    const int cc = tid % CPR, r0 = tid / CPR;
    const int n = n0 + cc * 8;
    const bool active = n < p.Cout;
    float sc[8], sf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = 0.f; sf[i] = 0.f; }
    if (active) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(p.scale + n), a1 = *reinterpret_cast<const f32x4 *>(p.scale + n + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(p.shift + n), h1 = *reinterpret_cast<const f32x4 *>(p.shift + n + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = a0[i]; sc[i + 4] = a1[i]; sf[i] = h0[i]; sf[i + 4] = h1[i]; }
    }

    // Prologue, ordered to shorten the per-workgroup ramp (measured: prologue + epilogue were 30-55 % of these kernels):
    // the weight DMAs of the first S-1 stages need no geometry and go out first, this thread's K-table entries for those
    // stages come straight from global memory, and the LDS copy of the table (used from the first in-loop issue on) is
    // written last; the first barrier of the loop publishes it.
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < p.nk) issue_w(s, s);
    if (p.utap) {
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < p.nk) issue_a(next_entry(), s);
    } else {
        int2 e0[S - 1];
#pragma unroll
        for (int s = 0; s < S - 1; ++s) e0[s] = p.ktab[(s < p.nk ? s : 0) * 8 + kc];
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < p.nk) issue_a(e0[s], s);
        for (int i = tid; i < p.nk * 8; i += NT) ktab_lds[i] = p.ktab[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // table in LDS (and, once, the whole prologue) before barrier 0
    }

#ifdef TEDSPAD_DEBUG_TS
    unsigned long long dbg_t0 = __builtin_readcyclecounter(), dbg_t1 = 0, dbg_wait = 0;
#endif
    int rd = 0, wr = S - 1;  // ring slots: stage kt is read from `rd`, stage kt+S-1 is written to `wr`
    for (int kt = 0; kt < p.nk; ++kt) {
#ifdef TEDSPAD_DEBUG_TS
        const unsigned long long dbg_w0 = __builtin_readcyclecounter();
#endif
        // stage kt must have landed; up to S-2 later stages stay in flight across the barrier
        const int later = p.nk - 1 - kt;
        if (S >= 4 && later >= 2) wait_vmcnt<(S >= 4 ? 2 : 0) * L>();
        else if (later >= 1) wait_vmcnt<(S >= 3 ? 1 : 0) * L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // everyone's DMA of stage kt is visible; the slot of stage kt-1 is free
        asm volatile("" ::: "memory");
#ifdef TEDSPAD_DEBUG_TS
        if (kt == 0) dbg_t1 = __builtin_readcyclecounter(); else dbg_wait += __builtin_readcyclecounter() - dbg_w0;
#endif
        if (kt + S - 1 < p.nk) issue(kt + S - 1, wr);
        const uint16_t *A = reinterpret_cast<const uint16_t *>(smem + rd * STAGE) + (wm * (BM / WM) + l31) * BK;
        const uint16_t *W = reinterpret_cast<const uint16_t *>(smem + rd * STAGE + BM * BK * 2) + (wn * (BN / WN) + l31) * BK;
#pragma unroll
        for (int kq = 0; kq < BK / 16 / KS; ++kq) {
            const int ks = kq + kh2 * (BK / 16 / KS);
            const int coff = (((ks << 1) | lh) ^ swz) << 3;
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) fa[b] = *reinterpret_cast<const uint4 *>(A + b * 32 * BK + coff);
#pragma unroll
            for (int a = 0; a < TN; ++a) fw[a] = *reinterpret_cast<const uint4 *>(W + a * 32 * BK + coff);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma(fw[a], fa[b], acc[a][b]);
        }
        rd = rd + 1 == S ? 0 : rd + 1;
        wr = wr + 1 == S ? 0 : wr + 1;
    }
    __syncthreads();  // all waves done with the ring before it is reused as the fp32 staging tile
#ifdef TEDSPAD_DEBUG_TS
    if (g_dbg_ts_ig && threadIdx.x == 0) {
        unsigned long long *dbg = g_dbg_ts_ig + (size_t)blockIdx.x * 4;
        dbg[0] = dbg_t0; dbg[1] = dbg_t1; dbg[2] = __builtin_readcyclecounter(); dbg[3] = dbg_wait;
    }
#endif

    // ---- epilogue: fp32 tile -> LDS -> coalesced 16-byte rows --------------------------
    // The residual rows this thread will add are requested BEFORE the staging writes (forward layers with the plain output map):
    // their L2 / HBM round trip runs under the staging traffic and the barriers instead of once per row inside the store loop.
    // Named variables, not an array: hipcc puts a conditionally filled array of this size into scratch.
    constexpr int NR = BM / RPP;          // output rows per thread
    static_assert(BM % RPP == 0 && NR <= 8, "epilogue row passes");
    // Without a residual the same registers prefetch the ReLU-backward MASK rows (data gradients of the training step: the deep layers' launches are 10-20 us and the
    // mask load inside the store loop was one dependent round trip per row, eight per tile).
    const bool res_pre = p.res && !p.ostrided && active;
    const bool mask_pre = !p.res && p.mask && !p.ostrided && active;
    const uint16_t *pre_ptr = res_pre ? p.res : p.mask;
    const int pre_ld = res_pre ? p.ldres : p.ldmask;
    auto res_row = [&](int it) -> uint4 {
        const int m = m0 + r0 + it * RPP;
        return ((res_pre || mask_pre) && it < NR && m < p.M) ? *reinterpret_cast<const uint4 *>(pre_ptr + (size_t)m * pre_ld + n) : make_uint4(0, 0, 0, 0);
    };
    const uint4 rres0 = res_row(0), rres1 = res_row(1), rres2 = res_row(2), rres3 = res_row(3);
    const uint4 rres4 = res_row(4), rres5 = res_row(5), rres6 = res_row(6), rres7 = res_row(7);
    float *stg = reinterpret_cast<float *>(smem);
    if (KS == 1 || kh2 == 1) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int ml = wm * (BM / WM) + b * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * (BN / WN) + a * 32 + 8 * g + 4 * lh;
                    f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl) = v;
                }
            }
    }
    __syncthreads();
    if (KS == 2) {
        if (kh2 == 0) {      // add the partner wave's partial sums (same lanes, same addresses)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int ml = wm * (BM / WM) + b * 32 + l31;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wn * (BN / WN) + a * 32 + 8 * g + 4 * lh;
                        f32x4 *ptr = reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl);
                        const f32x4 o = *ptr;
                        f32x4 v = {acc[a][b][4 * g] + o[0], acc[a][b][4 * g + 1] + o[1], acc[a][b][4 * g + 2] + o[2], acc[a][b][4 * g + 3] + o[3]};
                        *ptr = v;
                    }
                }
        }
        __syncthreads();
    }

    float s1[8], s2[8], t1[8], t2[8];      // t*: the rows of the tile that belong to the NEXT statistics group (grouped batch statistics)
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; t1[i] = 0.f; t2[i] = 0.f; }
    const int sgrp = p.stats_rows ? m0 / p.stats_rows : 0;
    const int smb = p.stats_rows ? (sgrp + 1) * p.stats_rows : 0x7fffffff;     // first row of the next group
    auto out_row = [&](const int it, const uint4 rpre) {
        const int r = r0 + it * RPP;
        const int m = m0 + r;
        if (m >= p.M) return;
        size_t op = (size_t)m;
        if (p.ostrided) {
            const int wo = m % p.Wo; const int q1 = m / p.Wo;
            const int ho = q1 % p.Ho; const int q2 = q1 / p.Ho;
            const int to = q2 % p.To; const int nb = q2 / p.To;
            op = (((size_t)nb * p.TF + to * p.ost + p.oot) * p.HF + ho * p.osh + p.ooh) * p.WF + wo * p.osw + p.oow;
        }
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8 + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
        if (p.stats) {
            if (m < smb) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { t1[i] += v[i]; t2[i] += v[i] * v[i]; }
            }
        }
        if (p.res) {
            float rr[8];
            uint4 rv = rpre;
            if (!res_pre) rv = *reinterpret_cast<const uint4 *>(p.res + op * p.ldres + n);
            unpack8<T>(rv, rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rr[i];
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        }
        if (p.sigmoid) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 1.f / (1.f + __expf(-v[i]));
        }
        if (p.mask) {
            float mk[8];
            uint4 mv = rpre;
            if (!mask_pre) mv = *reinterpret_cast<const uint4 *>(p.mask + op * p.ldmask + n);
            unpack8<T>(mv, mk);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = mk[i] > 0.f ? v[i] : 0.f;
        }
        if (p.y) *reinterpret_cast<uint4 *>(p.y + op * p.ldy + n) = pack8_lim<T>(v, p.sat);
        if (p.y32) {
            *reinterpret_cast<f32x4 *>(p.y32 + op * p.ldy32 + n) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(p.y32 + op * p.ldy32 + n + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    };
    if (active) {
        out_row(0, rres0);
        if constexpr (NR > 1) out_row(1, rres1);
        if constexpr (NR > 2) out_row(2, rres2);
        if constexpr (NR > 3) out_row(3, rres3);
        if constexpr (NR > 4) out_row(4, rres4);
        if constexpr (NR > 5) out_row(5, rres5);
        if constexpr (NR > 6) out_row(6, rres6);
        if constexpr (NR > 7) out_row(7, rres7);
    }
    if (p.stats) {   // block-level reduction of the batch statistics, then one atomic per channel
        __syncthreads();
        float *red = stg;   // [2][RPP][BN]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[r0 * BN + cc * 8 + i] = s1[i];
            red[(RPP + r0) * BN + cc * 8 + i] = s2[i];
        }
        __syncthreads();
        float *so = p.stats + (size_t)sgrp * 2 * p.stats_ld;
        const bool det = det_enter();            // deterministic mode: one workgroup at a time, in blockIdx order (det_gate.h)
        if (tid < BN && n0 + tid < p.Cout) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RPP; ++r) { a += red[r * BN + tid]; b += red[(RPP + r) * BN + tid]; }
            atomicAdd(so + n0 + tid, a);
            atomicAdd(so + p.stats_ld + n0 + tid, b);
        }
        if (smb < m0 + BM && smb < p.M) {       // the tile straddles a group boundary (workgroup-uniform)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[r0 * BN + cc * 8 + i] = t1[i];
                red[(RPP + r0) * BN + cc * 8 + i] = t2[i];
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Cout) {
                float a = 0.f, b = 0.f;
                for (int r = 0; r < RPP; ++r) { a += red[r * BN + tid]; b += red[(RPP + r) * BN + tid]; }
                atomicAdd(so + 2 * p.stats_ld + n0 + tid, a);
                atomicAdd(so + 3 * p.stats_ld + n0 + tid, b);
            }
        }
        det_exit(det);
    }
}

// ------------------------------------------------------------------------------------------
// Halo-direct kernel for the two Cin=3 stems in pixel-pair form (cin' = 8 -> one K chunk = one
// tap = 16 bytes): the im2col matrix of a stem tile is 10-35x larger than the input patch it is
// built from, and the generic kernel above is bound by exactly that L2->LDS traffic (N = 64 only).
// Here a workgroup owns a 1 x 8 x 32 output patch (256 pixels x all 64 channels): its input halo
// (kt x ((8-1)*sh+kh) x (31+kw) positions x 16 B ~ 59 KB for the 5x7x7 stem) is DMA'd into LDS
// ONCE, and every MFMA B fragment (8 channels of one tap for one pixel) is read straight from the
// halo at `pixel base + tap delta`: consecutive lanes = consecutive pixels along W = consecutive
// 16-byte slots, conflict-free. Only the 8 KB weight tile streams per K step (2-slot ring).
// 4 waves, each 64 pixels (2 output rows) x 64 channels; <= 80 KB LDS so two workgroups share a CU
// (one loads its halo while the other computes).
// ------------------------------------------------------------------------------------------
// output patch (rows x cols) = (256 / TW) x TW with TW = 32 (one MFMA pixel group per row) or 16 (a pixel group = two rows of 16: the
// 112 x 112 stem outputs of a 224 x 224 clip tile exactly with 16 x 16 patches, while 8 x 32 patches waste 12.5 % of their columns)
constexpr int ST_WSTAGE = 64 * BK * 2;        // 64 channels x 64 k x 2 B

// FR = output frames per patch (1: 4 waves, 256 pixels; 2: 8 waves, 512 pixels on frames to, to+1). The kernel is bound
// by the L2 -> LDS stream (per 256-pixel patch: 147 KB of weights + 61 KB of halo); with FR = 2 every weight stage
// serves twice the pixels and the two frames share most of their temporal halo (7 input frames instead of 2 x 5):
// 229 KB instead of 416 KB of DMA per 512 pixels.
// KS = 2: split-K inside the workgroup. Measured with s_memtime stamps (FR = 1, KS = 1): a workgroup lives 36 k cycles =
// prologue 10 k (halo DMA) + K loop 19 k + epilogue 7 k, and the loop takes 19 k whether or not the CU's other workgroup
// is also in its loop: ONE wave per SIMD issues its 4-MFMA groups at ~50 % duty (dependent LDS fragment reads), two fill
// the pipe. With 8 waves per patch -- waves 4-7 take the k16 sub-steps 2,3 of every K step, waves 0-3 the sub-steps 0,1,
// partial sums added through the staging tile -- a single workgroup saturates the matrix cores, so the other workgroup's
// prologue / epilogue no longer idles them. (fp32 partial sums are re-associated: results within one f16 rounding step
// of KS = 1, like the halo-direct trunk kernels.)
template <typename T, int FR, int KS, int TW = 32>
__global__ __launch_bounds__(256 * FR * KS) void conv_stem_halo_kernel(const ConvKP p, const int HH, const int WH, const int tiles_h, const int tiles_w) {
    constexpr int NT = 256 * FR * KS;
    constexpr int ST_TH = 256 / TW, ST_TW = TW;
    constexpr int WS = (FR == 1) ? 2 : 4;    // weight ring slots: FR = 2 has the LDS for four (its staging tile is larger anyway)
    constexpr int WL = NT == 256 ? 2 : 1;    // weight DMA instructions per thread and stage
    static_assert(FR * KS <= 2, "8 waves at most");
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tgroups = (p.To + FR - 1) / FR;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; b /= tiles_h;
    const int to = (b % tgroups) * FR;
    const int n = b / tgroups;
    const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
    const int HT = p.kt + (FR - 1) * p.st;              // input frames under the patch
    const int P = HT * HH * WH;                         // halo positions
    const int NH = (P + NT - 1) / NT;                   // DMA instructions per thread for the halo
    const int Pr = (P + 63) / 64 * 64;                  // positions rounded to whole wave instructions
    const int halo_bytes = Pr * 16;
    unsigned char *wring = dsm + halo_bytes;            // [WS][64][64] 16-bit
    int *tapd = reinterpret_cast<int *>(wring + WS * ST_WSTAGE);  // byte delta of every K chunk (= tap)
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16);

    for (int i = tid; i < p.nk * 8; i += NT) {
        const int y = p.ktab[i].y;
        const int dt = y & 255, dh = ((y >> 8) & 255) - 8, dw = (y >> 16) - 16;
        tapd[i] = dt < 8 ? ((dt * HH + dh) * WH + dw) * 16 : 0;   // K padding: zero weights, any in-range address
    }
    __syncthreads();  // table complete (and its global loads retired) before any DMA is counted
    // ---- halo: one 16-byte DMA per position, lane-linear in LDS --------------------------------
    const int t0 = to * p.st - p.pt, h0 = ho0 * p.sh - p.ph, w0 = wo0 * p.sw - p.pw;
    const float inv_wh = 1.0f / (float)WH, inv_hh = 1.0f / (float)HH;   // fp32 reciprocal + exactness fix-up instead of integer division:
    for (int i = 0; i < NH; ++i) {                                       // 2 divisions x up to 15 DMA slots per thread were ~4 k cycles of a 10 k prologue
        if (i * NT + wave * 64 >= Pr) break;             // wave-uniform: nothing of this instruction is inside the halo
        const int idx = i * NT + tid;
        const int r = fdiv(idx, WH, inv_wh); const int ww = idx - r * WH;
        const int dt = fdiv(r, HH, inv_hh); const int hh = r - dt * HH;
        const int it = t0 + dt, ih = h0 + hh, iw = w0 + ww;
        const bool ok = idx < P && (unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        const uint16_t *src = ok ? p.x + ((((size_t)n * p.Ti + it) * p.Hi + ih) * p.Wi + iw) * p.ldx : zero;
        lds_dma16(src, lds0 + (i * NT + wave * 64) * 16);
    }
    // ---- weights: [64][64] tile per K step, swizzled on the source like the generic kernel ------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w + (size_t)rsub * p.Kpad + kc * 8;
    auto issue_w = [&](int kt, int slot) {
        const unsigned dst = lds0 + halo_bytes + slot * ST_WSTAGE + wave * 8 * (BK * 2);
        lds_dma16(wsrc + kt * BK, dst);
        if (NT == 256) lds_dma16(wsrc + (size_t)32 * p.Kpad + kt * BK, dst + 32 * (BK * 2));   // 8 waves cover the 64 rows in one pass
    };
#pragma unroll
    for (int s0 = 0; s0 < WS - 1; ++s0)
        if (s0 < p.nk) issue_w(s0, s0);

    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    // wave (f, wq) = (wave / 4, wave % 4) owns output rows 2*wq, 2*wq+1 of frame to+f; B-fragment base of pixel (row, l31)
    const int wf = FR == 2 ? wave >> 2 : 0, wq = wave & 3;
    const int kh2 = KS == 2 ? wave >> 2 : 0;        // which half of the k16 sub-steps this wave multiplies
    int pixb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {      // fragment g of wave wq = patch pixels (2*wq + g)*32 .. +31 in row-major order of the TH x TW patch
        const int pr = ((2 * wq + g) * 32 + l31) / TW, pc = ((2 * wq + g) * 32 + l31) % TW;
        pixb[g] = ((wf * p.st * HH + pr * p.sh) * WH + pc * p.sw) * 16;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][g][r] = 0.f;

    for (int kt = 0; kt < p.nk; ++kt) {
        // this wave's tap deltas of the step (constant table): requested before the wait so they are not on the
        // fragment-read -> MFMA dependency chain
        int dk[BK / 16 / KS];
#pragma unroll
        for (int kq = 0; kq < BK / 16 / KS; ++kq) dk[kq] = tapd[kt * 8 + (kq + kh2 * (BK / 16 / KS)) * 2 + lh];
        // stage kt must have landed (the halo was issued before stage 0); up to WS-2 later stages stay in flight
        const int later = p.nk - 1 - kt;
        if (WS >= 4 && later >= 2) wait_vmcnt<(WS >= 4 ? 2 : 0) * WL>();
        else if (WS >= 3 && later >= 1) wait_vmcnt<(WS >= 3 ? 1 : 0) * WL>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // weight stage kt (+ halo, tap table on kt = 0) visible; the slot of stage kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + WS - 1 < p.nk) issue_w(kt + WS - 1, (kt + WS - 1) % WS);
        const uint16_t *W = reinterpret_cast<const uint16_t *>(wring + (kt % WS) * ST_WSTAGE) + l31 * BK;
#pragma unroll
        for (int kq = 0; kq < BK / 16 / KS; ++kq) {
            const int ks = kq + kh2 * (BK / 16 / KS);
            const int d = dk[kq];
            const int coff = (((ks << 1) | lh) ^ swz) << 3;
            uint4 fa[2], fw[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) fa[g] = *reinterpret_cast<const uint4 *>(dsm + pixb[g] + d);
#pragma unroll
            for (int a = 0; a < 2; ++a) fw[a] = *reinterpret_cast<const uint4 *>(W + a * 32 * BK + coff);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[a][g] = T::mfma(fw[a], fa[g], acc[a][g]);
        }
    }
    __syncthreads();

    // ---- epilogue: fp32 patch [256*FR pixels][64 channels] -> LDS -> coalesced rows -----------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
    const int wrow = KS == 2 ? (wave & 3) : wave;          // patch row pair of this wave
    if (KS == 1 || kh2 == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ml = (2 * wrow + g) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = a * 32 + 8 * q + 4 * lh;
                    f32x4 v = {acc[a][g][4 * q], acc[a][g][4 * q + 1], acc[a][g][4 * q + 2], acc[a][g][4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl) = v;
                }
            }
    }
    __syncthreads();
    if (KS == 2) {
        if (kh2 == 0) {      // first half: add the partner wave's partial sums (same lanes, same addresses)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int ml = (2 * wrow + g) * 32 + l31;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = a * 32 + 8 * q + 4 * lh;
                        f32x4 *ptr = reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl);
                        const f32x4 o = *ptr;
                        f32x4 v = {acc[a][g][4 * q] + o[0], acc[a][g][4 * q + 1] + o[1], acc[a][g][4 * q + 2] + o[2], acc[a][g][4 * q + 3] + o[3]};
                        *ptr = v;
                    }
                }
        }
        __syncthreads();
    }
    const int cc = tid & 7, r0 = tid >> 3;     // 8 chunks of 8 channels per pixel, NT/8 pixels per pass
    const int nch = cc * 8;
    const bool chan_ok = nch < p.Cout;
    float sc[8], sf[8], s1[8], s2[8];          // s1 / s2: batch statistics of the pre-activation (train-mode BatchNorm behind the stem: the UNet's first conv,
#pragma unroll                                 // I3Res50's stem), as the generic epilogue gathers them; a tile lies inside one sample, hence inside one statistics group
    for (int i = 0; i < 8; ++i) { sc[i] = chan_ok ? p.scale[nch + i] : 0.f; sf[i] = chan_ok ? p.shift[nch + i] : 0.f; s1[i] = 0.f; s2[i] = 0.f; }
    for (int r = r0; r < 256 * FR; r += NT / 8) {
        const int tf = to + (r >> 8), ho = ho0 + (r & 255) / TW, wo = wo0 + (r & 255) % TW;
        if (!chan_ok || tf >= p.To || ho >= p.Ho || wo >= p.Wo) continue;
        const size_t m = (((size_t)n * p.To + tf) * p.Ho + ho) * p.Wo + wo;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
        if (p.stats) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
        }
        if (p.res) {
            float rr[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(p.res + m * p.ldres + nch), rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rr[i];
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8_lim<T>(v, p.sat);
    }
    if (p.stats) {   // block-level reduction over the NT / 8 row groups, then one atomic per channel (workgroup-uniform branch)
        constexpr int RG = NT / 8;
        __syncthreads();
        float *red = stg;   // [2][RG][64]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[r0 * 64 + cc * 8 + i] = s1[i];
            red[(RG + r0) * 64 + cc * 8 + i] = s2[i];
        }
        __syncthreads();
        const size_t mfirst = ((size_t)n * p.To + to) * p.Ho * p.Wo;
        const size_t sgrp = p.stats_rows ? mfirst / (size_t)p.stats_rows : 0;
        float *so = p.stats + sgrp * 2 * p.stats_ld;
        const bool det = det_enter();
        if (tid < 64 && tid < p.Cout) {
            float sa = 0.f, sb = 0.f;
            for (int r = 0; r < RG; ++r) { sa += red[r * 64 + tid]; sb += red[(RG + r) * 64 + tid]; }
            atomicAdd(so + tid, sa);
            atomicAdd(so + p.stats_ld + tid, sb);
        }
        det_exit(det);
    }
}

template <typename T, int FR, int KS, int TW = 32>
int32_t launch_stem_halo(const ConvKP &p, int N, hipStream_t s) {
    constexpr int ST_TH = 256 / TW, ST_TW = TW;
    if (p.ldx < 8 || p.Cout > 64 || p.Kpad > 8 * 256 * 8 || p.sw != 1 || p.sigmoid) {
        set_error("tedspad_conv_fwd: halo-direct config needs cin == 8, cout <= 64, sw == 1");
        return TEDSPAD_EINVAL;
    }
    constexpr int NT = 256 * FR * KS;
    const int HH = (ST_TH - 1) * p.sh + p.kh, WH = (ST_TW - 1) * p.sw + p.kw;
    const int HT = p.kt + (FR - 1) * p.st;
    const int P = HT * HH * WH, NH = (P + NT - 1) / NT;
    const int main_bytes = (P + 63) / 64 * 64 * 16 + (FR == 1 ? 2 : 4) * ST_WSTAGE + p.nk * 8 * 4;
    const int stage_bytes = 256 * FR * (64 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    if (lds > 160 * 1024) {
        set_error("tedspad_conv_fwd: halo-direct config: halo does not fit LDS (%d bytes)", lds);
        return TEDSPAD_EINVAL;
    }
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_stem_halo_kernel<T, FR, KS, TW>;
    if (attr_set[T::kDtype] < lds) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 160 * 1024;
    }
    const int tiles_h = (p.Ho + ST_TH - 1) / ST_TH, tiles_w = (p.Wo + ST_TW - 1) / ST_TW;
    const int tgroups = (p.To + FR - 1) / FR;
    hipLaunchKernelGGL(kfn, dim3(N * tgroups * tiles_h * tiles_w), dim3(NT), lds, s, p, HH, WH, tiles_h, tiles_w);
    return check_launch("tedspad_conv_fwd(halo)");
}

template <typename T, int BM, int BN, int WM, int WN, int S, int KT, int KS = 1>
int32_t launch(const ConvKP &p, hipStream_t s) {
    if (KT == 0 ? !p.utap : p.Kpad > KT) {
        set_error("tedspad_conv_fwd: tile_cfg needs %s", KT == 0 ? "cin % 64 == 0" : "a shorter K");
        return TEDSPAD_EINVAL;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvKP q = p;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, S, KT, KS>), dim3(tiles_m * q.tiles_n), dim3(WM * WN * 64 * KS), 0, s, q);
    return check_launch("tedspad_conv_fwd");
}

inline long ntiles(const ConvKP &p, int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); }

// Tile configurations (tedspad_conv_desc.tile_cfg = index + 1; 0 = heuristic below).
//  id  pixels x channels  waves  ring  LDS      regime
//   1  256 x 128          4x2    3     154 KB   long K, wide N: 1 WG/CU, 8 waves
//   2  256 x  64          4x2    3     130 KB   long K, N <= 64
//   3  128 x 128          2x2    3     106 KB   fewer tiles than CUs at 256
//   4  128 x  64          2x2    3      82 KB
//   5   64 x  64          2x2    3      58 KB   tiny M
//   6  128 x 128          2x2    2      68 KB   short K (<= 1024): HBM-bound 1x1 convs, 2 WG/CU
//   7  128 x  64          2x2    2      50 KB   short K, N <= 64: 3 WG/CU
//   8   64 x 128          2x2    2      50 KB   short K, small M
//   9  1x8x32 patch, halo-direct (cin == 8 stems only): input patch resident in LDS, 2 WG/CU
//  10   64 x  64          2x2    2      33 KB   short K, streaming: 4 WG/CU
//  11  128 x 128          2x2    4     138 KB   long K, deeper ring
//  12  128 x  64          2x2    4     108 KB   long K, N <= 64, deeper ring
//  13  128 x 128          2x2    2      76 KB   long K, 2 WG/CU (one WG's prologue/epilogue under the other's MFMAs)
//  14  128 x  64          2x2    2      59 KB   long K, N <= 64, 2 WG/CU
//  17  256 x  64          4x2    2      80 KB   cin % 64 == 0 (no K table): TWO 8-wave WGs per CU
//  18  128 x 128          2x2    2      68 KB   cin % 64 == 0
// (table-free 128x64 / 64x64 / 64x128 tiles with 3-5 WG/CU were measured for the HBM-bound 1x1 layers: no faster than 17
//  -- every tile shape plateaus at ~3.1-3.3 TB/s of a 4.8 TB/s copy, the per-WG load -> MFMA -> store phases do not overlap)
//  20  2x8x32 patch (two output frames), halo-direct stem with 8 waves sharing every weight stage, 1 WG/CU
//  22  128 x 128, split-K over 8 waves (2 x 2 x 2), ring 2, 2 WG/CU: 4 waves per SIMD
//  23  the same, table-free (cin % 64 == 0)
//  24  256 x 128, split-K over 16 waves, ring 3, 1 WG/CU
//  21  1x8x32 patch, halo-direct stem with split-K over 8 waves (a single workgroup saturates the MFMA pipe), 2 WG/CU
//  19  128 x 64, PERSISTENT pointwise (conv_pw.hip): 1x1x1 convs with cin = 64 / 128, next tile prefetched under the stores
//  15, 16  retired (the round-1 8-wave halo-direct kernel: superseded by 32-34)
//  25  256 x 256, PING-PONG (conv_p8.hip): 8 waves, the two waves of a SIMD one barrier apart, 4 phases per K tile, 1 WG/CU
//  26  the same on v_mfma_f32_16x16x32 (higher sustained clock; fp32 sums re-associated)
// (256 x 64 with 4 waves of 64 px x 64 co, and 512 x 64 with 8 such waves, were measured on the 64-channel layers of layer1: 404 / 455 us
//  against 390 us for tile 17 -- every tile shape lands on the same ~515 TFLOP/s there, time proportional to K: the L2 -> LDS stream)
//  27  256-pixel flat halo (conv_flat.hip): stride-1 'same' 1 x kh x kw convs with cin = 64, cout <= 64: the tile's input halo is one contiguous
//      run of pixels fetched once, taps read from it; 4 waves, 2 WG/CU
//  28  temporal flat halo (conv_flat.hip): stride-1 'same' kt x 1 x 1 convs with cin % 64 == 0, cout <= 64, T <= 4: a workgroup owns 64 spatial
//      positions of all frames of a clip; each 64-channel chunk of the input is fetched once for all taps (K walked chunk-major)
//  29  stem, 16 x 16 patch (otherwise tile 9);  30  stem, 16 x 16 patch with split-K over 8 waves (otherwise tile 21)
//  31  stem, two output frames per workgroup (tile 20) on 16 x 16 patches
//  32  16 x 16 patch halo (conv_patch.hip): stride-1 'same' 1 x kh x kw convs with cin % 64 == 0, cout <= 128 (the UNet's wide outer levels)
//  33  the same with flat tiles (256 consecutive output pixels, halo = one contiguous run as tile 27): cin % 64 == 0, cout <= 128, narrow frames
//  34  the same for kt x 1 x 1 'same' convs: a tile is all T <= 4 frames of 256 / T spatial positions, taps outside the clip skipped (cout <= 512)
//  35  128 x 128 split-K over 8 waves, table-free, ring 3 (tile 23 waits 61 % of its wave-cycles with one K tile in flight on layer4's M = 22 050: profiles/r03_bench_cfg2_mfma_util.md)
//  36  the same with ring 4;  37  retired (256 x 128 split-K over 16 waves, table-free, ring 3: never picked)
//  38  TWO 16 x 16 patches per workgroup sharing every weight stage (conv_patch.hip, conv_patch2_kernel): 1 x 3 x 3 'same' convs, cin % 64 == 0, K in half chunks
//      of 32 channels, a kernel row of weights per stage: a third fewer bytes through the LDS fill path than tile 32
//  39  retired (the same on two flat tiles of 256 consecutive pixels: never picked)
//  40  PERSISTENT two-patch kernel (conv_patch3.hip, round 6): tile 38's tile on 8 waves, one workgroup per CU walking a run of tiles; halo double-buffered (the next
//      half chunk's / next tile's halo issued a half chunk ahead), weight ring of six stages filled five ahead -- resident when cin <= 64 --, epilogue straight from the
//      accumulators (v_permlane16_swap -> 16-byte pieces), batch statistics summed in registers over the whole run; 32 < cout <= 64; sums bit-identical to tile 38's
constexpr int NUM_CFGS = 40;

// 64-wide sibling of a 128-wide generic tile (same M tile and ring where there is one), 0: none
inline int narrow_sibling(int cfg) {
    switch (cfg) {
        case 1: case 24: return 2;
        case 3: return 4;
        case 6: return 7;
        case 11: return 12;
        case 13: case 22: return 14;
        case 18: case 23: case 35: case 36: return 17;
    }
    return 0;
}

template <typename T>
int32_t launch_cfg(int cfg, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src = nullptr) {
    // cout = 128 k + r with r <= 64 on a 128-wide tile (Inception's 3 x 3 x 3 convs 96 -> 208, 112 -> 224, 144 -> 288, 160 -> 320 and its fused 1 x 1 x 1 reduce
    // GEMMs): the last tile column would multiply up to 64 + 63 channels of zero weights. The first 128 k channels run on the chosen tile, the rest on its 64-wide
    // sibling with every per-channel pointer moved (conv_patch.hip does the same for the halo tiles); the sibling launches first: if it does not take the geometry
    // nothing has run yet and the conv goes unsplit (if the HEAD launch fails after it, the error is returned and y holds only the last channels). The KS = 1 tiles sum K in
    // one order: for them the split changes no bit. The split-K tiles (22 / 23 / 24 / 35 / 36, KS = 2) have KS = 1 siblings: their last r channels are summed in the plain K order,
    // the head in the split order -- the same one-rounding-step difference those tiles have against every other tile (test_every_tile_configuration_gives_the_same_result),
    // so TEDSPAD_IGEMM_NO_SPLIT=1 is bit-identical to the split only for KS = 1 tiles.
    static const bool split_ok = getenv("TEDSPAD_IGEMM_NO_SPLIT") == nullptr;      // A/B knob
    const int sib = narrow_sibling(cfg);
    if (split_ok && sib && !src && p.Cout > 128 && p.Cout % 128 != 0 && p.Cout % 128 <= 64 && !p.fold_hw && !p.x2) {
        const int head = p.Cout / 128 * 128;
        ConvKP a = p, b = p;
        a.Cout = head;
        b.Cout = p.Cout - head;
        b.w += (size_t)head * p.Kpad; b.scale += head; b.shift += head;
        if (b.res) b.res += head;
        if (b.y) b.y += head;
        if (b.mask) b.mask += head;
        if (b.stats) b.stats += head;
        if (b.y32) b.y32 += head;
        if (launch_cfg<T>(sib, b, N, cin, s, nullptr) == TEDSPAD_OK) return launch_cfg<T>(cfg, a, N, cin, s, nullptr);
    }
    switch (cfg) {
        case 9: return launch_stem_halo<T, 1, 1>(p, N, s);
        case 20: return launch_stem_halo<T, 2, 1>(p, N, s);
        case 21: return launch_stem_halo<T, 1, 2>(p, N, s);
        case 29: return launch_stem_halo<T, 1, 1, 16>(p, N, s);
        case 30: return launch_stem_halo<T, 1, 2, 16>(p, N, s);
        case 31: return launch_stem_halo<T, 2, 1, 16>(p, N, s);
        case 22: return launch<T, 128, 128, 2, 2, 2, KTAB_MAX_BYTES, 2>(p, s);
        case 23: return launch<T, 128, 128, 2, 2, 2, 0, 2>(p, s);
        case 24: return launch<T, 256, 128, 4, 2, 3, KTAB_MAX_BYTES, 2>(p, s);
        case 35: return launch<T, 128, 128, 2, 2, 3, 0, 2>(p, s);
        case 36: return launch<T, 128, 128, 2, 2, 4, 0, 2>(p, s);
        case 15:
        case 16:   // the 8-wave halo-direct kernel (round 1) never won the tuner once the chunk-major tiles 32-34 existed: retired, ids kept
        case 37:   // 256 x 128 split-K, table-free, ring 3 (round 4) and
        case 39:   // two flat tiles per workgroup (round 4): never picked by the tuner over a whole bench run (extraction, anonymised extraction, training: TEDSPAD_TILE_PICKS): retired in round 5
            set_error("tedspad_conv_fwd: tile_cfg 15 / 16 / 37 / 39 are retired");
            return TEDSPAD_EINVAL;
        case 1: return launch<T, 256, 128, 4, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 2: return launch<T, 256, 64, 4, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 3: return launch<T, 128, 128, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 4: return launch<T, 128, 64, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 5: return launch<T, 64, 64, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 6: return launch<T, 128, 128, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 7: return launch<T, 128, 64, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 8: return launch<T, 64, 128, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 10: return launch<T, 64, 64, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 11: return launch<T, 128, 128, 2, 2, 4, KTAB_MAX_BYTES>(p, s);
        case 12: return launch<T, 128, 64, 2, 2, 4, KTAB_MAX_BYTES>(p, s);
        case 13: return launch<T, 128, 128, 2, 2, 2, KTAB_MAX_BYTES>(p, s);
        case 14: return launch<T, 128, 64, 2, 2, 2, KTAB_MAX_BYTES>(p, s);
        case 17: return launch<T, 256, 64, 4, 2, 2, 0>(p, s);
        case 18: return launch<T, 128, 128, 2, 2, 2, 0>(p, s);
        case 19: return launch_conv_pw(T::kDtype, p, s);
        case 25: return launch_conv_p8(T::kDtype, p, s);
        case 27: return launch_conv_flat(T::kDtype, p, cin, s);
        case 28: return launch_conv_tflat(T::kDtype, p, N, cin, s);
        case 32: return launch_conv_patch(T::kDtype, p, N, cin, s, 0, src);
        case 33: return launch_conv_patch(T::kDtype, p, N, cin, s, 1, src);
        case 34: return launch_conv_patch(T::kDtype, p, N, cin, s, 2);
        case 38: return launch_conv_patch2(T::kDtype, p, N, cin, s, src);
        case 40: return launch_conv_patch3(T::kDtype, p, N, cin, s, src);
        case 26: return launch_conv_p8(T::kDtype, p, s, 16);
    }
    set_error("tedspad_conv_fwd: tile_cfg %d out of range 0..%d", cfg, NUM_CFGS);
    return TEDSPAD_EINVAL;
}

inline int heuristic_cfg(const ConvKP &p, int cin) {
    const bool narrow = p.Cout <= 64;
    if (cin == 8 && narrow && p.sw == 1 && p.kt * p.kh * p.kw >= 32 && !p.sigmoid) return 9;  // the stems
    if (p.Kpad <= 512) {  // short K: little to pipeline, bandwidth-bound -> more resident workgroups
        if (narrow) return 7;
        return ntiles(p, 128, 128) >= 512 ? 6 : 8;
    }
    // Largest tile that still gives every one of the 256 CUs a workgroup (1 resident WG/CU).
    if (narrow) {
        if (ntiles(p, 256, 64) >= 256) return 2;
        if (ntiles(p, 128, 64) >= 192) return 4;
        return 5;
    }
    if (ntiles(p, 256, 128) >= 256) return 1;
    if (ntiles(p, 128, 128) >= 192) return 3;
    if (ntiles(p, 128, 64) >= 192) return 4;
    return 5;
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

static bool desc_ok(const tedspad_conv_desc *d) {
    return d && d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cin % 8 == 0 && d->ldx % 8 == 0 &&
           d->ldx >= d->cin && d->cout > 0 && d->cout % 8 == 0 && d->ldy % 8 == 0 && d->ldy >= d->cout && d->kt > 0 &&
           d->kh > 0 && d->kw > 0 && d->kt <= 7 && d->kh <= 7 && d->kw <= 7 && d->st > 0 && d->sh > 0 && d->sw > 0 &&
           d->to > 0 && d->ho > 0 && d->wo > 0 && (d->dtype == TEDSPAD_F16 || d->dtype == TEDSPAD_BF16);
}

extern "C" int32_t tedspad_conv_kpad(const tedspad_conv_desc *d) {
    if (!desc_ok(d)) return TEDSPAD_EINVAL;
    const int k = d->kt * d->kh * d->kw * d->cin;
    return (k + BK - 1) / BK * BK;
}

extern "C" int32_t tedspad_conv_cout_pad(const tedspad_conv_desc *d) {
    if (!desc_ok(d)) return TEDSPAD_EINVAL;
    return (d->cout + 127) / 128 * 128;
}

extern "C" int32_t tedspad_conv_num_tile_cfgs(void) { return NUM_CFGS; }

extern "C" int32_t tedspad_conv_ktab_entries(const tedspad_conv_desc *d) {
    const int kp = tedspad_conv_kpad(d);
    return kp < 0 ? kp : kp / 8;
}

extern "C" int32_t tedspad_conv_build_ktab(const tedspad_conv_desc *d, int32_t *out) {
    TS_REQUIRE(desc_ok(d) && out, "tedspad_conv_build_ktab: bad descriptor (kernel dims must be <= 7, channels multiples of 8)");
    const int entries = tedspad_conv_ktab_entries(d);
    const int cpc = d->cin / 8;  // chunks per tap
    const int taps = d->kt * d->kh * d->kw;
    for (int e = 0; e < entries; ++e) {
        const int tap = e / cpc, c8 = e % cpc;
        if (tap >= taps) {  // K padding: shift amounts that hit no mask bit -> zero page
            out[2 * e] = 0;
            out[2 * e + 1] = 31 | (31 << 8) | (31 << 16);
            continue;
        }
        const int dw = tap % d->kw, dh = (tap / d->kw) % d->kh, dt = tap / (d->kw * d->kh);
        const long off = ((long)(dt * d->h + dh) * d->w + dw) * d->ldx + c8 * 8;
        TS_REQUIRE(off < (1L << 31), "tedspad_conv_build_ktab: tap offset overflows int32");
        out[2 * e] = (int32_t)off;
        out[2 * e + 1] = dt | ((8 + dh) << 8) | ((16 + dw) << 16);
    }
    return TEDSPAD_OK;
}

static int32_t conv_fwd_impl(const tedspad_conv_desc *d, const void *x, const void *w_packed, const int32_t *ktab,
                                       const float *scale, const float *shift, const void *residual, void *y, int32_t sigmoid,
                                       const tedspad_conv_extras *ex, void *stream, int pool_t, const ConvKP *dual = nullptr) {
    TS_REQUIRE(desc_ok(d), "tedspad_conv_fwd: bad descriptor (cin/cout/ld* multiples of 8, kernel dims <= 7)");
    const bool gathered = ex && ex->nchunk_src > 0;
    TS_REQUIRE((x || gathered) && w_packed && ktab && scale && shift && (y || (ex && ex->y32)), "tedspad_conv_fwd: null pointer");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)scale | (uintptr_t)shift) % 16 == 0,
               "tedspad_conv_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(!residual || (d->ldres % 8 == 0 && d->ldres >= d->cout), "tedspad_conv_fwd: bad ldres");
    // output geometry must be consistent with the input + padding (guards the gather's bounds)
    TS_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0 && (d->to - 1) * d->st - d->pt < d->t && (d->ho - 1) * d->sh - d->ph < d->h &&
                   (d->wo - 1) * d->sw - d->pw < d->w,
               "tedspad_conv_fwd: output extent reaches past the input");
    const long in_elems = (long)d->n * d->t * d->h * d->w * d->ldx;
    const long M = (long)d->n * d->to * d->ho * d->wo;
    TS_REQUIRE((gathered || in_elems < (1L << 31)) && M < (1L << 31), "tedspad_conv_fwd: tensor too large for 32-bit gather offsets; split the batch");
    ConvKP p;
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w_packed; p.ktab = (const int2 *)ktab;
    p.scale = scale; p.shift = shift; p.res = (const uint16_t *)residual; p.y = (uint16_t *)y;
    p.M = (int)M; p.Cout = d->cout; p.Kpad = tedspad_conv_kpad(d); p.nk = p.Kpad / BK;
    TS_REQUIRE(p.Kpad <= KTAB_MAX_BYTES, "tedspad_conv_fwd: K = kt*kh*kw*cin too large (max 10240)");
    p.Ti = d->t; p.Hi = d->h; p.Wi = d->w; p.ldx = d->ldx;
    p.To = d->to; p.Ho = d->ho; p.Wo = d->wo; p.ldy = d->ldy; p.ldres = d->ldres;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw;
    p.st = d->st; p.sh = d->sh; p.sw = d->sw; p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
    p.relu = d->relu; p.sigmoid = sigmoid;
    p.pointwise = (d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 &&
                   d->ph == 0 && d->pw == 0 && d->to == d->t && d->ho == d->h && d->wo == d->w);
    p.tiles_n = 0;
    p.cin = d->cin; p.utap = (d->cin % BK == 0) ? 1 : 0;
    p.inv_wo = 1.0f / (float)d->wo; p.inv_ho = 1.0f / (float)d->ho; p.inv_to = 1.0f / (float)d->to;
    p.mask = nullptr; p.stats = nullptr; p.ldmask = 0; p.stats_ld = 0; p.stats_rows = 0; p.ostrided = 0; p.y32 = nullptr; p.ldy32 = 0;
    p.fold_hw = 0; p.fold_c = 0; p.fold_f = 1;
    p.ost = p.osh = p.osw = 1; p.oot = p.ooh = p.oow = 0; p.TF = d->to; p.HF = d->ho; p.WF = d->wo;
    p.x2 = p.w2 = nullptr; p.scale2 = p.shift2 = nullptr; p.ldx2 = 0; p.nk1 = 0; p.Hi2 = p.Wi2 = 0; p.sh2 = p.sw2 = 1;
    p.sat = (ex && ex->nosat) ? __builtin_inff() : 65504.f;
    if (dual && dual->nk1 > 0) {      // K-concatenated pair on the ping-pong kernel (tedspad_conv_p8_dual_fwd)
        TS_REQUIRE(p.pointwise && d->cin % BK == 0 && !residual && !ex && !sigmoid && !pool_t && d->cout % 256 == 0,
                   "tedspad_conv_p8_dual_fwd: first conv 1x1x1 stride 1 with cin %% 64 == 0, cout %% 256 == 0");
        p.x2 = dual->x2; p.ldx2 = dual->ldx2; p.nk1 = d->cin / BK; p.Hi2 = dual->Hi2; p.Wi2 = dual->Wi2; p.sh2 = dual->sh2; p.sw2 = dual->sw2;
        p.Kpad = d->cin + dual->nk1 * BK; p.nk = p.Kpad / BK;      // dual->nk1 carries the K tiles of the SECOND source here
        p.utap = 1;
        return launch_conv_p8(d->dtype, p, (hipStream_t)stream);
    }
    if (dual) {
        TS_REQUIRE(p.pointwise && d->cin == 64 && !residual && !ex && !sigmoid && !pool_t, "tedspad_conv_pw_dual_fwd: two 1x1x1 stride-1 convs with cin = 64");
        p.x2 = dual->x2; p.w2 = dual->w2; p.scale2 = dual->scale2; p.shift2 = dual->shift2; p.ldx2 = dual->ldx2;
        return launch_conv_pw(d->dtype, p, (hipStream_t)stream, false);
    }
    bool extras = false;
    if (ex) {
        TS_REQUIRE(!ex->mask || (ex->ldmask % 8 == 0 && ex->ldmask >= d->cout && (uintptr_t)ex->mask % 16 == 0), "tedspad_conv_fwd_ex: bad mask");
        TS_REQUIRE(!ex->stats || ex->stats_ld >= d->cout, "tedspad_conv_fwd_ex: stats_ld must be >= cout");
        TS_REQUIRE(!ex->y32 || (ex->ldy32 % 4 == 0 && ex->ldy32 >= d->cout && (uintptr_t)ex->y32 % 16 == 0), "tedspad_conv_fwd_ex: bad y32");
        p.mask = (const uint16_t *)ex->mask; p.ldmask = ex->ldmask; p.stats = ex->stats; p.stats_ld = ex->stats_ld;
        p.stats_rows = ex->stats ? ex->stats_rows : 0;
        TS_REQUIRE(p.stats_rows >= 0 && (p.stats_rows == 0 || p.stats_rows >= 256), "tedspad_conv_fwd_ex: stats_rows must be 0 or >= 256 (a tile of up to 256 rows may straddle one group boundary)");
        p.y32 = ex->y32; p.ldy32 = ex->ldy32;
        if (ex->out_strided) {
            TS_REQUIRE(ex->ost > 0 && ex->osh > 0 && ex->osw > 0 && ex->oot >= 0 && ex->ooh >= 0 && ex->oow >= 0 &&
                           (d->to - 1) * ex->ost + ex->oot < ex->tf && (d->ho - 1) * ex->osh + ex->ooh < ex->hf &&
                           (d->wo - 1) * ex->osw + ex->oow < ex->wf,
                       "tedspad_conv_fwd_ex: strided output does not fit the (tf,hf,wf) tensor");
            p.ostrided = 1; p.ost = ex->ost; p.osh = ex->osh; p.osw = ex->osw; p.oot = ex->oot; p.ooh = ex->ooh; p.oow = ex->oow;
            p.TF = ex->tf; p.HF = ex->hf; p.WF = ex->wf;
        }
        extras = p.mask || p.stats || p.ostrided || p.y32;
        if (ex->fold_hw > 0) {
            TS_REQUIRE(!extras && !residual && !sigmoid && !pool_t && ex->fold_c > 0 && d->cout % ex->fold_c == 0 && ex->fold_c % 128 == 0 &&
                           ex->fold_ldy >= ex->fold_c && ex->fold_ldy % 8 == 0 && (long)d->to * d->ho * d->wo == ex->fold_hw &&
                           (long)d->n * ex->fold_hw * (d->cout / ex->fold_c) * ex->fold_ldy < (1L << 31),
                       "tedspad_conv_fwd_ex: folded output frames: plain epilogue, fold_hw = to*ho*wo, fold_c %% 128 == 0 dividing cout, fold_ldy >= fold_c");
            p.fold_hw = ex->fold_hw; p.fold_c = ex->fold_c; p.fold_f = d->cout / ex->fold_c;
            p.ldy = ex->fold_ldy;                  // row stride of the folded tensor (d->ldy only has to satisfy the descriptor check)
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (pool_t) {
        TS_REQUIRE(!extras && !sigmoid, "tedspad_conv_pool_t2_fwd: no mask / stats / fp32 / strided-output epilogue");
        return launch_conv_pw(d->dtype, p, s, true);
    }
    int cfg = d->tile_cfg > 0 ? d->tile_cfg : heuristic_cfg(p, extras ? 0 : d->cin);
    PatchSrc gsrc{};
    if (gathered) {       // the input is a gathered concatenation: patch / flat halo tiles only
        if (d->tile_cfg <= 0) cfg = (d->kh - 1) * d->w + d->kw - 1 + 256 <= 384 - 1 ? 33 : 32;
        if (cfg != 32 && cfg != 33 && cfg != 38 && cfg != 39 && cfg != 40) {
            set_error("tedspad_conv_fwd_ex: gathered sources (nchunk_src) run on tile_cfg 32 / 33 / 38 / 40 only");
            return TEDSPAD_EUNSUPPORTED;
        }
        TS_REQUIRE(!dual && !pool_t && !p.fold_hw && d->kt == 1 && d->cin % 64 == 0 && ex->nchunk_src == d->cin / 64 && ex->nchunk_src <= 8,
                   "tedspad_conv_fwd_ex: gathered sources: kt = 1, cin %% 64 == 0, nchunk_src = cin / 64 <= 8");
        gsrc.n = ex->nchunk_src; gsrc.up = ex->chunk_up & ((1 << gsrc.n) - 1);
        TS_REQUIRE(!gsrc.up || (d->h % 2 == 0 && d->w % 2 == 0), "tedspad_conv_fwd_ex: an upsampled source needs even h and w");
        for (int k = 0; k < gsrc.n; ++k) {
            const long px = (gsrc.up >> k) & 1 ? (long)d->n * d->t * (d->h / 2) * (d->w / 2) : (long)d->n * d->t * d->h * d->w;
            TS_REQUIRE(ex->chunk_src[k] && (uintptr_t)ex->chunk_src[k] % 16 == 0 && ex->chunk_ld[k] >= 64 && ex->chunk_ld[k] % 8 == 0 && px * ex->chunk_ld[k] < (1L << 31),
                       "tedspad_conv_fwd_ex: gathered sources: null / misaligned chunk_src, chunk_ld < 64 or not a multiple of 8, or a source beyond 2^31 elements");
            gsrc.ptr[k] = (const uint16_t *)ex->chunk_src[k]; gsrc.ld[k] = ex->chunk_ld[k];
        }
    }
    if (p.fold_hw) {
        if (d->tile_cfg <= 0) cfg = 25;
        TS_REQUIRE(cfg == 25 || cfg == 26, "tedspad_conv_fwd_ex: folded output frames run on the ping-pong kernel only (tile_cfg 25 / 26)");
    }
    if (cfg == 9 || cfg == 20 || cfg == 21 || (cfg >= 29 && cfg <= 31)) {
        TS_REQUIRE(d->cin == 8, "tedspad_conv_fwd: tile_cfg 9 (halo-direct) needs cin == 8");
        TS_REQUIRE(!(p.mask || p.ostrided || p.y32), "tedspad_conv_fwd_ex: tile_cfg 9 (halo-direct) has no mask / strided-output / fp32-output epilogue (batch statistics: yes)");
        TS_REQUIRE(!p.stats || !p.stats_rows || ((long)d->to * d->ho * d->wo > 0 && p.stats_rows % ((long)d->to * d->ho * d->wo) == 0),
                   "tedspad_conv_fwd_ex: halo-direct tiles need statistics groups of whole samples");
    }
    const PatchSrc *gp = gathered ? &gsrc : nullptr;
    return d->dtype == TEDSPAD_F16 ? launch_cfg<F16>(cfg, p, d->n, d->cin, s, gp) : launch_cfg<BF16>(cfg, p, d->n, d->cin, s, gp);
}

extern "C" int32_t tedspad_conv_fwd_ex(const tedspad_conv_desc *d, const void *x, const void *w_packed, const int32_t *ktab,
                                       const float *scale, const float *shift, const void *residual, void *y, int32_t sigmoid,
                                       const tedspad_conv_extras *ex, void *stream) {
    return conv_fwd_impl(d, x, w_packed, ktab, scale, shift, residual, y, sigmoid, ex, stream, 0);
}

extern "C" int32_t tedspad_conv_pool_t2_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                            const float *shift, const void *residual, void *y, void *stream) {
    static const int32_t dummy_ktab[2] = {0, 0};   // the pointwise path never reads the table
    return conv_fwd_impl(d, x, w_packed, dummy_ktab, scale, shift, residual, y, 0, nullptr, stream, 1);
}

extern "C" int32_t tedspad_conv_pw_dual_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                                            const float *shift, const void *x2, int32_t ldx2, const void *w2_packed, const float *scale2,
                                            const float *shift2, void *y, void *stream) {
    static const int32_t dummy_ktab[2] = {0, 0};
    TS_REQUIRE(x2 && w2_packed && scale2 && shift2 && ldx2 >= 64 && ldx2 % 8 == 0 &&
                   ((uintptr_t)x2 | (uintptr_t)w2_packed | (uintptr_t)scale2 | (uintptr_t)shift2) % 16 == 0,
               "tedspad_conv_pw_dual_fwd: second source: null / misaligned pointer or bad ldx2");
    ConvKP dual;
    dual.x2 = (const uint16_t *)x2; dual.w2 = (const uint16_t *)w2_packed; dual.scale2 = scale2; dual.shift2 = shift2; dual.ldx2 = ldx2;
    dual.nk1 = 0;
    return conv_fwd_impl(d, x, w_packed, dummy_ktab, scale, shift, nullptr, y, 0, nullptr, stream, 0, &dual);
}

extern "C" int32_t tedspad_conv_p8_dual_fwd(const tedspad_conv_desc *d, const void *x, const void *x2, int32_t cin2, int32_t ldx2, int32_t h2, int32_t w2,
                                            int32_t sh2, int32_t sw2, const void *w_packed, const float *scale, const float *shift, void *y,
                                            void *stream) {
    static const int32_t dummy_ktab[2] = {0, 0};
    TS_REQUIRE(d && x2 && cin2 > 0 && cin2 % BK == 0 && ldx2 >= cin2 && ldx2 % 8 == 0 && (uintptr_t)x2 % 16 == 0 && sh2 > 0 && sw2 > 0 &&
                   (d->ho - 1) * sh2 < h2 && (d->wo - 1) * sw2 < w2,
               "tedspad_conv_p8_dual_fwd: second source: cin2 %% 64 == 0, strided grid must cover the output");
    TS_REQUIRE((long)d->n * d->t * h2 * w2 * ldx2 < (1L << 31), "tedspad_conv_p8_dual_fwd: second source too large for 32-bit gather offsets; split the batch");
    ConvKP dual;
    dual.x2 = (const uint16_t *)x2; dual.ldx2 = ldx2; dual.nk1 = cin2 / BK; dual.Hi2 = h2; dual.Wi2 = w2; dual.sh2 = sh2; dual.sw2 = sw2;
    dual.w2 = nullptr; dual.scale2 = dual.shift2 = nullptr;
    return conv_fwd_impl(d, x, w_packed, dummy_ktab, scale, shift, nullptr, y, 0, nullptr, stream, 0, &dual);
}

extern "C" int32_t tedspad_conv_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const int32_t *ktab,
                                    const float *scale, const float *shift, const void *residual, void *y,
                                    int32_t sigmoid, void *stream) {
    return tedspad_conv_fwd_ex(d, x, w_packed, ktab, scale, shift, residual, y, sigmoid, nullptr, stream);
}

#ifdef TEDSPAD_DEBUG_TS
extern "C" int32_t tedspad_debug_set_igemm_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tedspad::g_dbg_ts_ig), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
namespace tedspad {
int32_t det_ctl_igemm(int op, int on) { return det_ctl(op, on); }
}  // namespace tedspad
