// Persistent Cin = 3 stem for gfx950: conv1 (5x7x7, stride 2, pad (2,3,3)) + bn1 + ReLU of large_i3d.py:133-137,229-231, with the
// TEMPORAL half of maxpool1 (MaxPool3d((2,3,3), stride 2), large_i3d.py:138,232) fused into its epilogue.
//
// What the earlier stem kernels measured (DESIGN.md "Stem: what was measured"): K = 1120 for 735 real taps x channels in the
// pixel-pair form; a barrier every 8-16 MFMAs because the [64 co][64 k] weight tile streams through a ring; a 10 k-cycle halo
// prologue and a 7 k-cycle epilogue around a 19 k-cycle loop; 2.89 GB written per 225 clips that the max-pool shrinks 8x.
// Here:
//   * temporal-unfolded K (one k16 MFMA step per (dh, dw) tap: 16 values = 5 frames x 3 channels + 1 zero) -> K = 784;
//   * the clip is laid out ONCE as X[n][h][w][64] 16-bit ("time-channels-last", tedspad_clip_to_tc): value (t + pt)*3 + c of a
//     pixel's 128-byte record is x[n][c][t][h][w]; the 16 values output frame `to` needs at that pixel are the 32 bytes
//     starting at value st*3*to (byte 12*to for the stride-2 stem): no temporal duplication (6.4 MB per clip, what the
//     pixel-pair layout took), the halo DMA simply reads at a 4-byte-aligned offset inside the record;
//   * ALL weights (49 taps x [64 co][16] = 98 KB) stay resident in LDS: the K loop has no barrier and no weight stream, its
//     fragment addresses are a per-lane base + compile-time immediates (fully unrolled, no address arithmetic);
//   * a workgroup is persistent (one per CU, 4 waves) and walks patches of 8 x 16 output pixels x 2 output frames; the
//     input halo of a patch (2 frames x 21 rows x 38 columns x 32 B = 50 KB) lives in two regions by ROW PARITY: the taps
//     with even dh read only even halo rows, the odd ones only odd rows, so while the 28 even-dh taps of patch i are
//     multiplied the odd rows of patch i are landing, and while its 21 odd-dh taps run the even rows of patch i+1 land:
//     one halo buffer, two barriers per patch, the DMA always a phase ahead;
//   * a wave owns output rows (r, r+4) x 16 columns of BOTH frames (64 px x 64 co, four 32x32x16 MFMAs per tap); the two
//     frames are the temporal pooling window, so relu(bn(.)) of both are max-ed in registers and only the pooled tensor
//     Y[n][to/2][ho][wo][64] is written (half the bytes), straight from the accumulators: v_permlane32_swap gives lane l the even and
//     lane l + 32 the odd 8-channel group of its pixel -> 16-byte stores, no LDS staging, no barrier.
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
constexpr int PT_RE = PT_TH + 3, PT_RO = PT_TH + 2;   // even / odd halo rows under the patch (halo rows 0 .. 20)
constexpr int PT_ROWB = 2 * PT_PP * 32;               // bytes per halo row: 2 column-parity planes
constexpr int PT_FE = PT_RE * PT_ROWB, PT_FO = PT_RO * PT_ROWB;   // bytes per frame and region
constexpr int PT_NTAP = 49;
constexpr int PT_W_BYTES = PT_NTAP * 64 * 32;         // 100352
constexpr int PT_E_SLOTS = 2 * PT_FE / 16, PT_O_SLOTS = 2 * PT_FO / 16;
constexpr int PT_E_JOBS = (PT_E_SLOTS + 63) / 64, PT_O_JOBS = (PT_O_SLOTS + 63) / 64;   // 1 KB wave-instructions per region
constexpr int PT_E_OFF = PT_W_BYTES, PT_O_OFF = PT_E_OFF + PT_E_JOBS * 1024;
constexpr int PT_LDS = PT_O_OFF + PT_O_JOBS * 1024;
constexpr int PT_E_ROUNDS = (PT_E_JOBS + 3) / 4, PT_O_ROUNDS = (PT_O_JOBS + 3) / 4;
static_assert(PT_LDS <= 160 * 1024, "weights + halo must fit the CU's LDS");
static_assert(PT_W_BYTES % 1024 == 0 && (4 * PT_ROWB) % 256 == 0, "LDS image alignment");

struct StemPT {
    const unsigned char *x;       // time-channels-last clip
    const unsigned char *wimg;    // [49][64][2][8] 16-bit, halves swizzled (tedspad_stem_pt_fwd)
    const float *scale, *shift;
    uint16_t *y;
    long sN;                      // bytes per clip
    int sTo, sH, sW;              // bytes per output frame / row / pixel of x
    int N, Tp, H, W, Ho, Wo, ldy, relu;
    int tiles_h, tiles_w, total, chunk;
};

__device__ __forceinline__ void gstore16(void *dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
}

// The taps of one phase, fully unrolled. SCHED 0: the order hipcc picks (it requests the fragments of a tap only after the previous tap's
// MFMAs are issued); SCHED 1: the fragments of tap i+1 are requested BEFORE the MFMAs of tap i (two register sets), pinned with
// sched_group_barrier, so one wave per SIMD keeps its matrix pipe fed across the LDS latency.
template <typename T, int PAR, int SCHED>
__device__ __forceinline__ void stem_pt_phase(const unsigned char *dsm, const int (&pa)[4], const int (&wa)[2], f32x16 (&acc)[2][2]) {
    constexpr int FR = PAR == 0 ? PT_FE : PT_FO;
    constexpr int NT = (PAR == 0 ? 4 : 3) * 7;
    uint4 fa[2][2], fw[2][2];
    auto load = [&](int i, uint4 (&xa)[2], uint4 (&xw)[2]) {
        const int dhh = i / 7, dw = i % 7;
        const int b = (dw + 1) & 1;                   // column 2*wo + dw - 3 = 2*(wo + a) + b
        const int ta = (dw - 3 - b) / 2 + 2;          // a + 2 (dw - 3 - b is even)
        const int tau = (2 * dhh + PAR) * 7 + dw;
        const int imm = dhh * PT_ROWB + b * PT_PP * 32;
#pragma unroll
        for (int g = 0; g < 2; ++g) xa[g] = *reinterpret_cast<const uint4 *>(dsm + pa[ta] + g * FR + imm);
#pragma unroll
        for (int a = 0; a < 2; ++a) xw[a] = *reinterpret_cast<const uint4 *>(dsm + wa[a] + tau * 2048);
    };
    if (SCHED == 1) {
        load(0, fa[0], fw[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // the first tap's reads open the pipeline
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (SCHED == 1) {
            if (i + 1 < NT) load(i + 1, fa[(i + 1) & 1], fw[(i + 1) & 1]);
        } else {
            load(i, fa[i & 1], fw[i & 1]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[a][g] = T::mfma(fw[i & 1][a], fa[i & 1][g], acc[a][g]);
        if (SCHED == 1) {
            if (i + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 LDS reads (tap i+1)
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                   // 4 MFMAs (tap i)
        }
    }
}

template <typename T, int SCHED>
__global__ __launch_bounds__(256) void conv_stem_pt_kernel(const StemPT p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    for (int j = wave; j < PT_W_BYTES / 1024; j += 4) lds_dma16(p.wimg + j * 1024 + lane * 16, lds0 + j * 1024);

    // ---- halo DMA slots of this lane (patch-invariant): LDS slot s of a region = (frame, row, plane, position, half) --------------
    int offE[PT_E_ROUNDS], rcE[PT_E_ROUNDS], offO[PT_O_ROUNDS], rcO[PT_O_ROUNDS];
    auto decode = [&](int s, int rows, int par, int &off, int &rc) {
        const int hs = s & 1;
        int q = s >> 1;
        const int pos = q % PT_PP; q /= PT_PP;
        const int b = q & 1; q >>= 1;
        const int row = q % rows, f = q / rows;
        const int hr = 2 * row + par, col = 2 * (pos - 2) + b;     // halo row 0..20 (input row ih0 + hr), column relative to 2*wo0
        const int hf = hs ^ ((pos >> 3) & 1);
        const bool ok = f < 2;
        off = ok ? f * p.sTo + hr * p.sH + col * p.sW + hf * 16 : 0;
        rc = ok ? (hr << 8) | (col + 4) : (1 << 28);                // a row far outside any clip: the slot reads the zero page
    };
#pragma unroll
    for (int i = 0; i < PT_E_ROUNDS; ++i) decode((i * 4 + wave) * 64 + lane, PT_RE, 0, offE[i], rcE[i]);
#pragma unroll
    for (int i = 0; i < PT_O_ROUNDS; ++i) decode((i * 4 + wave) * 64 + lane, PT_RO, 1, offO[i], rcO[i]);

    struct Patch { const unsigned char *pb; int ih0, iwm4, n, tp, ho0, wo0; };
    auto patch_of = [&](int kk) {
        Patch q;
        int r = base + kk;
        const int tw = r % p.tiles_w; r /= p.tiles_w;
        const int th = r % p.tiles_h; r /= p.tiles_h;
        q.tp = r % p.Tp; q.n = r / p.Tp;
        q.ho0 = th * PT_TH; q.wo0 = tw * PT_TW;
        q.ih0 = 2 * q.ho0 - 3; q.iwm4 = 2 * q.wo0 - 4;
        q.pb = p.x + (long)q.n * p.sN + (long)(2 * q.tp) * p.sTo + (long)q.ih0 * p.sH + (long)(2 * q.wo0) * p.sW;
        return q;
    };
    auto issue_E = [&](const Patch &q) {
#pragma unroll
        for (int i = 0; i < PT_E_ROUNDS; ++i) {
            const int j = i * 4 + wave;
            if (j >= PT_E_JOBS) break;                      // wave-uniform
            const bool ok = (unsigned)(q.ih0 + (rcE[i] >> 8)) < (unsigned)p.H && (unsigned)(q.iwm4 + (rcE[i] & 255)) < (unsigned)p.W;
            lds_dma16(ok ? q.pb + offE[i] : zero, lds0 + PT_E_OFF + j * 1024);
        }
    };
    auto issue_O = [&](const Patch &q) {
#pragma unroll
        for (int i = 0; i < PT_O_ROUNDS; ++i) {
            const int j = i * 4 + wave;
            if (j >= PT_O_JOBS) break;
            const bool ok = (unsigned)(q.ih0 + (rcO[i] >> 8)) < (unsigned)p.H && (unsigned)(q.iwm4 + (rcO[i] & 255)) < (unsigned)p.W;
            lds_dma16(ok ? q.pb + offO[i] : zero, lds0 + PT_O_OFF + j * 1024);
        }
    };

    Patch cur = patch_of(k);
    issue_E(cur);
    issue_O(cur);

    // ---- fragment bases ---------------------------------------------------------------------------------------------------------
    const int l15 = lane & 15, rsel = (lane >> 4) & 1, lh = lane >> 5, l31 = lane & 31;
    const int prow = wave + 4 * rsel;                       // output row of this lane's pixel inside the patch
    int paE[4], paO[4], wa[2];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
        const int pos = l15 + ta;
        const int o = prow * PT_ROWB + pos * 32 + 16 * (lh ^ ((pos >> 3) & 1));
        paE[ta] = PT_E_OFF + o;
        paO[ta] = PT_O_OFF + o;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) wa[a] = (a * 32 + l31) * 32 + 16 * (lh ^ (((a * 32 + l31) >> 4) & 1));

    // BatchNorm scale / shift of this lane's 32 output channels: co = a*32 + (r & 3) + 8*(r >> 2) + 4*lh
    float sc[2][16], sf[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sc[a][r] = p.scale[co];
            sf[a][r] = p.shift[co];
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(sc[a][r]), "+v"(sf[a][r]));   // hipcc's wait for these loads happens here, not in the loop

    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();      // weights + both halo regions of the first patch visible
    asm volatile("" ::: "memory");

    while (true) {
        const int kn = k + nx;
        const bool more = kn < lim;                         // workgroup-uniform
        Patch nxt = cur;
        if (more) nxt = patch_of(kn);

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][g][r] = 0.f;

        stem_pt_phase<T, 0, SCHED>(dsm, paE, wa, acc);             // taps dh = 0, 2, 4, 6 on the even halo rows
        wait_vmcnt<0>();                                    // odd rows of this patch (issued a phase ago) landed
        __builtin_amdgcn_s_barrier();                       // ... for every wave; every wave is done reading the even rows
        asm volatile("" ::: "memory");
        if (more) issue_E(nxt);                             // even rows of the NEXT patch land under the odd taps + epilogue
        stem_pt_phase<T, 1, SCHED>(dsm, paO, wa, acc);             // taps dh = 1, 3, 5 on the odd halo rows

        // ---- epilogue: relu(bn(.)) of both frames, max over the two frames (the temporal window of maxpool1), 16-byte stores --------
        {
            const int ho = cur.ho0 + prow, wo = cur.wo0 + l15;
            const bool inb = ho < p.Ho && wo < p.Wo;
            uint16_t *dst = p.y + ((((size_t)cur.n * p.Tp + cur.tp) * p.Ho + ho) * p.Wo + wo) * p.ldy + 8 * lh;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
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
                    gstore16(dst + a * 32, u32x4{d[0][0], d[0][1], d[1][0], d[1][1]});
                    gstore16(dst + a * 32 + 16, u32x4{d[2][0], d[2][1], d[3][0], d[3][1]});
                }
            }
        }
        if (!more) break;
        wait_vmcnt<0>();                                    // even rows of the next patch landed (and this patch's stores retired)
        __builtin_amdgcn_s_barrier();                       // ... for every wave; every wave is done reading the odd rows
        asm volatile("" ::: "memory");
        issue_O(nxt);                                       // odd rows of the next patch land under its even taps
        cur = nxt;
        k = kn;
    }
}

// fp32 NCTHW clip (any strides, W contiguous) -> X[n][h][w][64] 16-bit: value (t + pt)*3 + c of pixel (h, w) = x[n][c][t][h][w], zero
// elsewhere (the temporal zero padding of the conv lives in the record). A thread owns 16 values of one pixel: 16 coalesced row
// loads (a wave covers 64 consecutive pixels of one (c, t) row), two 16-byte LDS writes, then the 64-pixel tile leaves as one
// contiguous 8 KB run.
template <typename T>
__global__ __launch_bounds__(256) void clip_to_tc_kernel(const float *x, uint16_t *y, int c, int t, int h, int w, long sn, long sc, long st, long sh,
                                                         int pt, int wtiles) {
    __shared__ __attribute__((aligned(16))) uint4 tile[64 * 8];
    const int tid = threadIdx.x, px = tid & 63, vq = tid >> 6;
    int b = blockIdx.x;
    const int wt = b % wtiles; b /= wtiles;
    const int ih = b % h;
    const long n = b / h;
    const int iw = wt * 64 + px;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int val = vq * 16 + e;
        const int tt = val / 3 - pt, ch = val % 3;
        v[e] = 0.f;
        if (iw < w && tt >= 0 && tt < t && ch < c) v[e] = x[n * sn + ch * sc + tt * st + ih * sh + iw];
    }
    float lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[8 + e]; }
    tile[px * 8 + vq * 2] = pack8<T>(lo);
    tile[px * 8 + vq * 2 + 1] = pack8<T>(hi);
    __syncthreads();
    const int npx = min(64, w - wt * 64);
    uint4 *dst = reinterpret_cast<uint4 *>(y + ((n * h + ih) * (long)w + wt * 64) * 64);
    for (int i = tid; i < npx * 8; i += 256) dst[i] = tile[i];
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_clip_to_tc(const float *x, void *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int64_t sn, int64_t sc,
                                      int64_t st, int64_t sh, int64_t sw, int32_t pad_t, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && c <= 3 && t > 0 && h > 0 && w > 0 && pad_t >= 0, "tedspad_clip_to_tc: bad arguments");
    TS_REQUIRE((t + pad_t) * 3 <= 64, "tedspad_clip_to_tc: (t + pad_t) * 3 values must fit the 64-value pixel record");
    TS_REQUIRE(sw == 1 && (uintptr_t)y % 16 == 0, "tedspad_clip_to_tc: rows must be contiguous, y 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_clip_to_tc: bad dtype");
    TS_REQUIRE((long)n * h * ((w + 63) / 64) < (1L << 31), "tedspad_clip_to_tc: too many tiles");
    const int wtiles = (w + 63) / 64;
    const dim3 g((unsigned)((long)n * h * wtiles));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(clip_to_tc_kernel<F16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, wtiles);
    else hipLaunchKernelGGL(clip_to_tc_kernel<BF16>, g, dim3(256), 0, s, x, (uint16_t *)y, c, t, h, w, (long)sn, (long)sc, (long)st, (long)sh, pad_t, wtiles);
    return check_launch("tedspad_clip_to_tc");
}

extern "C" int32_t tedspad_stem_pt_wimg_bytes(void) { return PT_W_BYTES; }

extern "C" int32_t tedspad_stem_pt_fwd(const void *x_tc, const void *w_img, const float *scale, const float *shift, void *y, int32_t n, int32_t t_pairs,
                                       int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t stride_t, int32_t ldy, int32_t relu, int32_t nwg,
                                       int32_t variant, int32_t dtype, void *stream) {
    TS_REQUIRE(x_tc && w_img && scale && shift && y && n > 0 && t_pairs > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "tedspad_stem_pt_fwd: bad arguments");
    TS_REQUIRE(stride_t > 0 && (stride_t * 6) % 4 == 0 && (2 * t_pairs - 1) * stride_t * 3 + 16 <= 64,
               "tedspad_stem_pt_fwd: the 16 values of every output frame must lie inside the 64-value record at a 4-byte offset");
    TS_REQUIRE(ho == (h + 1) / 2 && wo == (w + 1) / 2, "tedspad_stem_pt_fwd: 7x7 stride-2 pad-3 geometry (ho = ceil(h / 2))");
    TS_REQUIRE(ldy >= 64 && ldy % 8 == 0 && ((uintptr_t)x_tc | (uintptr_t)w_img | (uintptr_t)y) % 16 == 0, "tedspad_stem_pt_fwd: 64 output channels, 16-byte aligned pointers");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_stem_pt_fwd: bad dtype");
    TS_REQUIRE((long)h * w * 128 < (1L << 31), "tedspad_stem_pt_fwd: frame too large for 32-bit halo offsets");
    StemPT p;
    p.x = (const unsigned char *)x_tc; p.wimg = (const unsigned char *)w_img; p.scale = scale; p.shift = shift; p.y = (uint16_t *)y;
    p.sW = 128; p.sH = w * 128; p.sN = (long)h * w * 128; p.sTo = stride_t * 6;
    p.N = n; p.Tp = t_pairs; p.H = h; p.W = w; p.Ho = ho; p.Wo = wo; p.ldy = ldy; p.relu = relu;
    p.tiles_h = (ho + PT_TH - 1) / PT_TH; p.tiles_w = (wo + PT_TW - 1) / PT_TW;
    const long total = (long)n * t_pairs * p.tiles_h * p.tiles_w;
    TS_REQUIRE(total < (1L << 30), "tedspad_stem_pt_fwd: too many patches");
    p.total = (int)total;
    p.chunk = (int)((total + 7) / 8);
    int grid = nwg > 0 ? nwg : 256;
    grid = (grid + 7) / 8 * 8;
    if ((long)grid > total + 7) grid = (int)((total + 7) / 8 * 8);
    hipStream_t s = (hipStream_t)stream;
    static thread_local int attr_set[4] = {0, 0, 0, 0};
    const int sched = variant & 1;
    const int di = (dtype == TEDSPAD_F16 ? 0 : 1) * 2 + sched;
    const void *fns[4] = {(const void *)conv_stem_pt_kernel<F16, 0>, (const void *)conv_stem_pt_kernel<F16, 1>,
                          (const void *)conv_stem_pt_kernel<BF16, 0>, (const void *)conv_stem_pt_kernel<BF16, 1>};
    if (!attr_set[di]) {
        if (hipFuncSetAttribute(fns[di], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_stem_pt_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[di] = 1;
    }
    switch (di) {
        case 0: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 0>), dim3(grid), dim3(256), PT_LDS, s, p); break;
        case 1: hipLaunchKernelGGL((conv_stem_pt_kernel<F16, 1>), dim3(grid), dim3(256), PT_LDS, s, p); break;
        case 2: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 0>), dim3(grid), dim3(256), PT_LDS, s, p); break;
        default: hipLaunchKernelGGL((conv_stem_pt_kernel<BF16, 1>), dim3(grid), dim3(256), PT_LDS, s, p); break;
    }
    return check_launch("tedspad_stem_pt_fwd");
}
