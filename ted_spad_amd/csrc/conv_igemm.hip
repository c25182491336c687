// Implicit-GEMM 3-D/2-D convolution for gfx950 (MI355X), channels-last, 16-bit operands,
// fp32 MFMA accumulation, fused BN scale/shift + residual + ReLU/sigmoid epilogue.
//
//   D[co][m] = sum_k W[co][k] * X[m][k],   m = (n,to,ho,wo),  k = (dt,dh,dw,ci)
//
// One kernel serves every convolution of the hot path (reference call sites:
// Unit3D.forward aux_code/models/i3d.py:89-120, Bottleneck.forward
// aux_code/models/large_i3d.py:61-84, the two stems, DoubleConv unet_parts.py:8-25):
//   * weights are the MFMA "A" operand (rows = output channels), activations the "B"
//     operand (columns = output pixels), so each lane ends with 4 CONSECUTIVE channels of
//     one pixel per accumulator group -> 16-byte LDS writes in the epilogue;
//   * both operands live in LDS as [row][64] 16-bit tiles (128-byte rows), 16-byte chunks
//     XOR-swizzled by (row>>1)&7 so every ds_read_b128 lane group covers a full 256-byte
//     bank row (conflict-free, MI355X LDS banking);
//   * the im2col gather is table driven: one {element offset, (dt,dh,dw)} entry per
//     8-channel K chunk, so any kernel/stride/asymmetric TF-SAME padding costs 3 adds and
//     3 unsigned compares per 16-byte load, and padded taps are predicated to zero;
//   * double-buffered LDS, next tile's global loads issued before the current tile's MFMAs;
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
    int st, sh, sw, pt, ph, pw;
    int relu, sigmoid, pointwise;
    int tiles_n;
};

constexpr int BK = 64;       // K elements per LDS tile (8 chunks of 16 bytes)
constexpr int NTHREADS = 256;

template <typename T, int BM, int BN>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_kernel(const ConvKP p) {
    constexpr int RA = BM / 32;          // activation rows staged per thread
    constexpr int RW = BN / 32;          // weight rows staged per thread
    constexpr int TM = BM / 64;          // 32-wide pixel tiles per wave
    constexpr int TN = BN / 64;          // 32-wide channel tiles per wave
    constexpr int LDS_MAIN = 2 * (BM + BN) * BK * 2;
    constexpr int STG_LD = BN + 4;       // fp32 staging row stride (floats): 16B-aligned, bank-skewed
    constexpr int LDS_STAGE = BM * STG_LD * 4;
    constexpr int LDS_BYTES = LDS_MAIN > LDS_STAGE ? LDS_MAIN : LDS_STAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    uint16_t *ldsA = reinterpret_cast<uint16_t *>(smem);                      // [2][BM][64] activations
    uint16_t *ldsW = reinterpret_cast<uint16_t *>(smem) + 2 * BM * BK;        // [2][BN][64] weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging roles: 8 consecutive lanes fetch one 128-byte K row segment ----------
    const int kc = tid & 7;
    const int srow = tid >> 3;           // 0..31
    int a_base[RA], a_t[RA], a_h[RA], a_w[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m >= p.M) {
            a_base[i] = 0; a_t[i] = -(1 << 20); a_h[i] = 0; a_w[i] = 0;
        } else if (p.pointwise) {
            a_base[i] = m * p.ldx; a_t[i] = 0; a_h[i] = 0; a_w[i] = 0;
        } else {
            const int wo = m % p.Wo; const int r1 = m / p.Wo;
            const int ho = r1 % p.Ho; const int r2 = r1 / p.Ho;
            const int to = r2 % p.To; const int n = r2 / p.To;
            a_t[i] = to * p.st - p.pt; a_h[i] = ho * p.sh - p.ph; a_w[i] = wo * p.sw - p.pw;
            a_base[i] = (((n * p.Ti + a_t[i]) * p.Hi + a_h[i]) * p.Wi + a_w[i]) * p.ldx;
        }
    }
    const uint16_t *wsrc = p.w + (size_t)(n0 + srow) * p.Kpad + kc * 8;

    uint4 ra[RA], rw[RW];
    auto load_tile = [&](int kt) {
        const int2 e = p.ktab[kt * 8 + kc];
        const int dt = (e.y << 24) >> 24, dh = (e.y << 16) >> 24, dw = (e.y << 8) >> 24;
        const bool kv = e.y >= 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool ok = kv && (unsigned)(a_t[i] + dt) < (unsigned)p.Ti && (unsigned)(a_h[i] + dh) < (unsigned)p.Hi &&
                            (unsigned)(a_w[i] + dw) < (unsigned)p.Wi;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (ok) ra[i] = *reinterpret_cast<const uint4 *>(p.x + (ptrdiff_t)(a_base[i] + e.x));
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) rw[j] = *reinterpret_cast<const uint4 *>(wsrc + (size_t)(32 * j) * p.Kpad + kt * BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = srow + 32 * i;
            *reinterpret_cast<uint4 *>(ldsA + buf * BM * BK + row * BK + ((kc ^ ((row >> 1) & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int row = srow + 32 * j;
            *reinterpret_cast<uint4 *>(ldsW + buf * BN * BK + row * BK + ((kc ^ ((row >> 1) & 7)) << 3)) = rw[j];
        }
    };

    // ---- MFMA roles: 4 waves as 2 (pixels) x 2 (channels) ---------------------------------
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < p.nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < p.nk) load_tile(kt + 1);
        const uint16_t *A = ldsA + buf * BM * BK + (wm * (BM / 2) + l31) * BK;
        const uint16_t *W = ldsW + buf * BN * BK + (wn * (BN / 2) + l31) * BK;
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
        if (kt + 1 < p.nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: fp32 tile -> LDS -> coalesced 16-byte rows --------------------------
    float *stg = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int ml = wm * (BM / 2) + b * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * (BN / 2) + a * 32 + 8 * g + 4 * lh;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl) = v;
            }
        }
    __syncthreads();

    constexpr int CPR = BN / 8;            // 16-byte output chunks per tile row
    constexpr int RPP = NTHREADS / CPR;    // rows per pass
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

template <typename T, int BM, int BN>
int32_t launch(const ConvKP &p, hipStream_t s) {
    const int tiles_m = (p.M + BM - 1) / BM;
    ConvKP q = p;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN>), dim3(tiles_m * q.tiles_n), dim3(NTHREADS), 0, s, q);
    return check_launch("tedspad_conv_fwd");
}

template <typename T>
int32_t dispatch(const ConvKP &p, hipStream_t s) {
    const bool narrow = p.Cout <= 64;
    // small problems: halve the pixel tile so the grid still covers the 256 CUs
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.Cout + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
    const bool small = tiles128 < 512;
    if (narrow) return small ? launch<T, 64, 64>(p, s) : launch<T, 128, 64>(p, s);
    return small ? launch<T, 64, 128>(p, s) : launch<T, 128, 128>(p, s);
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

static bool desc_ok(const tedspad_conv_desc *d) {
    return d && d->n > 0 && d->t > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cin % 8 == 0 && d->ldx % 8 == 0 &&
           d->ldx >= d->cin && d->cout > 0 && d->cout % 8 == 0 && d->ldy % 8 == 0 && d->ldy >= d->cout && d->kt > 0 &&
           d->kh > 0 && d->kw > 0 && d->kt < 128 && d->kh < 128 && d->kw < 128 && d->st > 0 && d->sh > 0 && d->sw > 0 &&
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

extern "C" int32_t tedspad_conv_ktab_entries(const tedspad_conv_desc *d) {
    const int kp = tedspad_conv_kpad(d);
    return kp < 0 ? kp : kp / 8;
}

extern "C" int32_t tedspad_conv_build_ktab(const tedspad_conv_desc *d, int32_t *out) {
    TS_REQUIRE(desc_ok(d) && out, "tedspad_conv_build_ktab: bad descriptor");
    const int entries = tedspad_conv_ktab_entries(d);
    const int cpc = d->cin / 8;  // chunks per tap
    const int taps = d->kt * d->kh * d->kw;
    for (int e = 0; e < entries; ++e) {
        const int tap = e / cpc, c8 = e % cpc;
        if (tap >= taps) {
            out[2 * e] = 0;
            out[2 * e + 1] = -1;
            continue;
        }
        const int dw = tap % d->kw, dh = (tap / d->kw) % d->kh, dt = tap / (d->kw * d->kh);
        const long off = ((long)(dt * d->h + dh) * d->w + dw) * d->ldx + c8 * 8;
        TS_REQUIRE(off < (1L << 31), "tedspad_conv_build_ktab: tap offset overflows int32");
        out[2 * e] = (int32_t)off;
        out[2 * e + 1] = dt | (dh << 8) | (dw << 16);
    }
    return TEDSPAD_OK;
}

extern "C" int32_t tedspad_conv_fwd(const tedspad_conv_desc *d, const void *x, const void *w_packed, const int32_t *ktab,
                                    const float *scale, const float *shift, const void *residual, void *y,
                                    int32_t sigmoid, void *stream) {
    TS_REQUIRE(desc_ok(d), "tedspad_conv_fwd: bad descriptor (cin/cout/ld* must be multiples of 8)");
    TS_REQUIRE(x && w_packed && ktab && scale && shift && y, "tedspad_conv_fwd: null pointer");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)scale | (uintptr_t)shift) % 16 == 0,
               "tedspad_conv_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(!residual || (d->ldres % 8 == 0 && d->ldres >= d->cout), "tedspad_conv_fwd: bad ldres");
    // output geometry must be consistent with the input + padding (guards the gather's bounds)
    TS_REQUIRE((d->to - 1) * d->st - d->pt < d->t && (d->ho - 1) * d->sh - d->ph < d->h && (d->wo - 1) * d->sw - d->pw < d->w,
               "tedspad_conv_fwd: output extent reaches past the input");
    const long in_elems = (long)d->n * d->t * d->h * d->w * d->ldx;
    const long M = (long)d->n * d->to * d->ho * d->wo;
    TS_REQUIRE(in_elems < (1L << 31) && M < (1L << 31), "tedspad_conv_fwd: tensor too large for 32-bit gather offsets; split the batch");
    ConvKP p;
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w_packed; p.ktab = (const int2 *)ktab;
    p.scale = scale; p.shift = shift; p.res = (const uint16_t *)residual; p.y = (uint16_t *)y;
    p.M = (int)M; p.Cout = d->cout; p.Kpad = tedspad_conv_kpad(d); p.nk = p.Kpad / BK;
    p.Ti = d->t; p.Hi = d->h; p.Wi = d->w; p.ldx = d->ldx;
    p.To = d->to; p.Ho = d->ho; p.Wo = d->wo; p.ldy = d->ldy; p.ldres = d->ldres;
    p.st = d->st; p.sh = d->sh; p.sw = d->sw; p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
    p.relu = d->relu; p.sigmoid = sigmoid;
    p.pointwise = (d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 &&
                   d->ph == 0 && d->pw == 0 && d->to == d->t && d->ho == d->h && d->wo == d->w);
    p.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == TEDSPAD_F16 ? dispatch<F16>(p, s) : dispatch<BF16>(p, s);
}
