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
#include "common.h"

namespace tedspad {
namespace {

struct ConvKP {
    const uint16_t *x;
    const uint16_t *w;
    const int2 *ktab;
    const float *scale;
    const float *shift;
    const uint16_t *res;
    uint16_t *y;
    int M, Cout, Kpad, nk;
    int Ti, Hi, Wi, ldx;
    int To, Ho, Wo, ldy, ldres;
    int kt, kh, kw;
    int st, sh, sw, pt, ph, pw;
    int relu, sigmoid, pointwise;
    int tiles_n;
};

constexpr int BK = 64;                  // K elements per LDS tile row (8 chunks of 16 bytes)
constexpr int KTAB_MAX_BYTES = 10240;   // Kpad <= 10240 (one int2 per 8 K elements)
constexpr int KTAB_SMALL_BYTES = 1024;  // "short-K" configs: Kpad <= 1024, smaller LDS -> 2-3 workgroups per CU

__device__ uint4 g_zero16;              // zero page for padded taps (zero-initialised by the loader)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst + lane*16].
// Written as inline asm on purpose: hipcc treats the builtin form as an LDS store it must
// drain (s_waitcnt vmcnt(0)) before ANY later ds_read of the same array, which serialises the
// ring. In asm form the compiler does not count it; the kernel waits with counted vmcnt itself.
// M0 (the DMA's LDS base) is compiler-reserved: saved/restored inside the same statement.
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int S, int KT>
__global__ __launch_bounds__(WM *WN * 64) void conv_igemm_kernel(const ConvKP p) {
    constexpr int NT = WM * WN * 64;
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
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];  // ONE array: see cdna guide, glds traps
    int2 *ktab_lds = reinterpret_cast<int2 *>(smem + S * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    for (int i = tid; i < p.nk * 8; i += NT) ktab_lds[i] = p.ktab[i];

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
                const int wo = m % p.Wo; const int r1 = m / p.Wo;
                const int ho = r1 % p.Ho; const int r2 = r1 / p.Ho;
                const int to = r2 % p.To; const int n = r2 / p.To;
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
    __syncthreads();  // ktab_lds visible

    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;  // LDS byte address of the ring
    auto issue = [&](int kt, int slot) {
        const unsigned stage = lds0 + slot * STAGE + wave * 8 * (BK * 2);
        const int2 e = ktab_lds[kt * 8 + kc];
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
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < p.nk) issue(s, s);

    int rd = 0, wr = S - 1;  // ring slots: stage kt is read from `rd`, stage kt+S-1 is written to `wr`
    for (int kt = 0; kt < p.nk; ++kt) {
        // stage kt must have landed; up to S-2 later stages stay in flight across the barrier
        const int later = p.nk - 1 - kt;
        if (S >= 4 && later >= 2) wait_vmcnt<(S >= 4 ? 2 : 0) * L>();
        else if (later >= 1) wait_vmcnt<(S >= 3 ? 1 : 0) * L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // everyone's DMA of stage kt is visible; the slot of stage kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + S - 1 < p.nk) issue(kt + S - 1, wr);
        const uint16_t *A = reinterpret_cast<const uint16_t *>(smem + rd * STAGE) + (wm * (BM / WM) + l31) * BK;
        const uint16_t *W = reinterpret_cast<const uint16_t *>(smem + rd * STAGE + BM * BK * 2) + (wn * (BN / WN) + l31) * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
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

    // ---- epilogue: fp32 tile -> LDS -> coalesced 16-byte rows --------------------------
    float *stg = reinterpret_cast<float *>(smem);
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
    __syncthreads();

    constexpr int CPR = BN / 8;          // 16-byte output chunks per tile row
    constexpr int RPP = NT / CPR;        // rows per pass
    const int cc = tid % CPR, r0 = tid / CPR;
    const int n = n0 + cc * 8;
    if (n >= p.Cout) return;
    float sc[8], sf[8];
    {
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(p.scale + n), s1 = *reinterpret_cast<const f32x4 *>(p.scale + n + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(p.shift + n), h1 = *reinterpret_cast<const f32x4 *>(p.shift + n + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = s0[i]; sc[i + 4] = s1[i]; sf[i] = h0[i]; sf[i + 4] = h1[i]; }
    }
#pragma unroll 4
    for (int r = r0; r < BM; r += RPP) {
        const int m = m0 + r;
        if (m >= p.M) break;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8 + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
        if (p.res) {
            float rr[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(p.res + (size_t)m * p.ldres + n), rr);
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
        *reinterpret_cast<uint4 *>(p.y + (size_t)m * p.ldy + n) = pack8<T>(v);
    }
}

template <typename T, int BM, int BN, int WM, int WN, int S, int KT>
int32_t launch(const ConvKP &p, hipStream_t s) {
    if (p.Kpad > KT) {
        set_error("tedspad_conv_fwd: tile_cfg needs Kpad <= %d, got %d", KT, p.Kpad);
        return TEDSPAD_EINVAL;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvKP q = p;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, S, KT>), dim3(tiles_m * q.tiles_n), dim3(WM * WN * 64), 0, s, q);
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
constexpr int NUM_CFGS = 8;

template <typename T>
int32_t launch_cfg(int cfg, const ConvKP &p, hipStream_t s) {
    switch (cfg) {
        case 1: return launch<T, 256, 128, 4, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 2: return launch<T, 256, 64, 4, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 3: return launch<T, 128, 128, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 4: return launch<T, 128, 64, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 5: return launch<T, 64, 64, 2, 2, 3, KTAB_MAX_BYTES>(p, s);
        case 6: return launch<T, 128, 128, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 7: return launch<T, 128, 64, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
        case 8: return launch<T, 64, 128, 2, 2, 2, KTAB_SMALL_BYTES>(p, s);
    }
    set_error("tedspad_conv_fwd: tile_cfg %d out of range 0..%d", cfg, NUM_CFGS);
    return TEDSPAD_EINVAL;
}

inline int heuristic_cfg(const ConvKP &p) {
    const bool narrow = p.Cout <= 64;
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

extern "C" int32_t tedspad_conv_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const int32_t *ktab,
                                    const float *scale, const float *shift, const void *residual, void *y,
                                    int32_t sigmoid, void *stream) {
    TS_REQUIRE(desc_ok(d), "tedspad_conv_fwd: bad descriptor (cin/cout/ld* multiples of 8, kernel dims <= 7)");
    TS_REQUIRE(x && w_packed && ktab && scale && shift && y, "tedspad_conv_fwd: null pointer");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)scale | (uintptr_t)shift) % 16 == 0,
               "tedspad_conv_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(!residual || (d->ldres % 8 == 0 && d->ldres >= d->cout), "tedspad_conv_fwd: bad ldres");
    // output geometry must be consistent with the input + padding (guards the gather's bounds)
    TS_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0 && (d->to - 1) * d->st - d->pt < d->t && (d->ho - 1) * d->sh - d->ph < d->h &&
                   (d->wo - 1) * d->sw - d->pw < d->w,
               "tedspad_conv_fwd: output extent reaches past the input");
    const long in_elems = (long)d->n * d->t * d->h * d->w * d->ldx;
    const long M = (long)d->n * d->to * d->ho * d->wo;
    TS_REQUIRE(in_elems < (1L << 31) && M < (1L << 31), "tedspad_conv_fwd: tensor too large for 32-bit gather offsets; split the batch");
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
    hipStream_t s = (hipStream_t)stream;
    const int cfg = d->tile_cfg > 0 ? d->tile_cfg : heuristic_cfg(p);
    return d->dtype == TEDSPAD_F16 ? launch_cfg<F16>(cfg, p, s) : launch_cfg<BF16>(cfg, p, s);
}
