// Halo-direct convolution for gfx950: stride-1 multi-tap convs (3x1x1, 1x3x3, 2-D 3x3, ...) with cin % 64 == 0.
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every input pixel once per tap, and round-1
// profiling showed it bound by the per-CU global->LDS (LDS-DMA) rate, ~25-36 GB/s/CU, not by MFMA. Here a
// workgroup owns a PT x PH x PW patch of <= 256 output pixels x BN output channels and walks K as
// (64-channel chunk of cin) x (tap): the input HALO of the patch for one channel chunk ([rows][64 ch], <= 352
// rows = 44 KB) is DMA'd into LDS ONCE and serves all taps -- every MFMA pixel-operand fragment is read at
// `halo row of the pixel + tap delta` -- so the activation traffic drops by ~the number of taps; only the
// [BN][64] weight tile streams per step. With BN = 256 the intensity is ~230 FLOP per DMA'd byte (85 before).
//
// Pipeline: what bounds these kernels is (bytes in flight per CU) / (DMA latency, ~2-3 us under load), so the
// weight tiles run THREE steps ahead in a 4-slot ring and the next chunk's halo is issued in full at the first
// step of the current chunk (2 halo slots). Per step (cc, tap): counted s_waitcnt vmcnt(N) with N = the exact
// number of this thread's DMA instructions issued after the weight tile of this step (kept in a 4-deep mark
// ring) -> one raw s_barrier -> issue [halo(cc+1) if tap == 0] + W(step+3) -> MFMAs. Same LDS image as conv_igemm (128-byte rows,
// XOR-swizzled 16-byte chunks, swizzle applied to the DMA source address), same epilogue (fp32 tile staged
// through LDS, coalesced 16-byte rows, scale/shift + residual + ReLU + mask + fp32 copy).
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16h;
#ifdef TEDSPAD_DEBUG_TS
__device__ unsigned long long *g_dbg_ts;   // debug builds only: [workgroup][4] = start, loop start, loop end, steps
#endif

struct HaloGeo {
    int PT, PH, PW;          // output patch
    int HT, HH, WH;          // halo extent = patch + kernel - 1
    int nH, NH;              // halo rows, DMA slots per thread per halo (ceil(nH / 64))
    int tiles_t, tiles_h, tiles_w, tiles_n;
    int ncc, ntaps, SL;      // cin / 64, taps, halo slices issued per step
    int halo_bytes;          // NH * 64 * 128
};

// halo rows per slot: 5 DMA slots (320 rows, 40 KB) next to the 4 x 16 KB weight ring of BN = 128; 6 (384 rows) with BN = 64

__device__ __forceinline__ void wait_vmcnt_n(int n) {
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

template <typename T, int BN>
__global__ __launch_bounds__(512) void conv_halo_kernel(const ConvKP p, const HaloGeo g) {
    constexpr int WN = BN / 64;            // waves along channels (64 channels each)
    constexpr int WM = 8 / WN;             // waves along pixels
    constexpr int TM = 256 / WM / 32;      // 32-pixel tiles per wave
    constexpr int TN = 2;                  // 32-channel tiles per wave
    constexpr int WSLOT = BN * BK * 2;
    constexpr int SW = BN / 64;            // weight DMA instructions per thread per step (64 rows x 8 chunks each)
    constexpr int STG_LD = 128 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char *hslot = dsm;                                   // [2][halo_bytes]
    unsigned char *wslot = dsm + 2 * g.halo_bytes;                // [4][WSLOT]
    int *dtab = reinterpret_cast<int *>(wslot + 4 * WSLOT);       // [ntaps] tap delta in halo rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = b % g.tiles_n; b /= g.tiles_n;
    const int tw = b % g.tiles_w; b /= g.tiles_w;
    const int th = b % g.tiles_h; b /= g.tiles_h;
    const int tt = b % g.tiles_t;
    const int n = b / g.tiles_t;
    const int to0 = tt * g.PT, ho0 = th * g.PH, wo0 = tw * g.PW, n0 = tile_n * BN;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16h);

    for (int i = tid; i < g.ntaps; i += 512) {
        const int dw = i % p.kw, dh = (i / p.kw) % p.kh, dt = i / (p.kw * p.kh);
        dtab[i] = (dt * g.HH + dh) * g.WH + dw;
    }
    // ---- halo DMA roles: 8 lanes per halo row; element offset of every row this thread fetches (or -1) ----------
    const int kc = (lane & 7) ^ ((tid >> 4) & 7);       // source chunk: swizzle ((row >> 1) & 7) on the source side
    int hoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        hoff[i] = -1;
        const int hr = i * 64 + (tid >> 3);
        if (i < g.NH && hr < g.nH) {
            const int hw = hr % g.WH; const int r = hr / g.WH;
            const int hh = r % g.HH; const int ht = r / g.HH;
            const int it = to0 - p.pt + ht, ih = ho0 - p.ph + hh, iw = wo0 - p.pw + hw;
            if ((unsigned)it < (unsigned)p.Ti && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi)
                hoff[i] = ((((n * p.Ti + it) * p.Hi + ih) * p.Wi + iw) * p.ldx) + kc * 8;
        }
    }
    auto issue_halo_slice = [&](int cc, int i) {   // slot i (0..NH-1) of the halo of channel chunk cc
        const uint16_t *src = hoff[i] >= 0 ? p.x + hoff[i] + cc * 64 : zero;
        lds_dma16(src, lds0 + (cc & 1) * g.halo_bytes + (i * 64 + wave * 8) * 128);
    };
    const uint16_t *wsrc = p.w + (size_t)(n0 + (tid >> 3)) * p.Kpad + kc * 8;
    const int Cin = g.ncc * 64;
    auto issue_w = [&](int step) {
        const int cc = step / g.ntaps, tap = step - cc * g.ntaps;
        const unsigned dst = lds0 + 2 * g.halo_bytes + (step & 3) * WSLOT + wave * 8 * 128;
        const uint16_t *src = wsrc + tap * Cin + cc * 64;
#pragma unroll
        for (int j = 0; j < SW; ++j) lds_dma16(src + (size_t)(j * 64) * p.Kpad, dst + j * 64 * 128);
    };

    // ---- MFMA roles ------------------------------------------------------------------------------------------------
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    const int npix = g.PT * g.PH * g.PW;
    int rowb[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int pl = wm * (256 / WM) + t * 32 + l31;
        rowb[t] = 0;
        if (pl < npix) {
            const int pw_ = pl % g.PW; const int r = pl / g.PW;
            const int ph_ = r % g.PH; const int pt_ = r / g.PH;
            rowb[t] = (pt_ * g.HH + ph_) * g.WH + pw_;
        }
    }
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
    __syncthreads();   // dtab written

    const int nsteps = g.ncc * g.ntaps;
#ifdef TEDSPAD_DEBUG_TS
    unsigned long long ts0 = __builtin_readcyclecounter(), ts1 = 0;
#endif
    // DMA instructions issued by this thread so far, and that count right after W(step) .. W(step+3) were issued
    int issued = 0, mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i < g.NH) { issue_halo_slice(0, i); ++issued; }
    issue_w(0); issued += SW; mk0 = issued;
    if (1 < nsteps) { issue_w(1); issued += SW; } mk1 = issued;
    if (2 < nsteps) { issue_w(2); issued += SW; } mk2 = issued;
    for (int step = 0; step < nsteps; ++step) {
        const int cc = step / g.ntaps, tap = step - cc * g.ntaps;
        // W(step) -- and everything older, i.e. the halo of this chunk -- must have landed; younger DMAs stay in flight
        wait_vmcnt_n(issued - mk0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef TEDSPAD_DEBUG_TS
        if (step == 0) ts1 = __builtin_readcyclecounter();
#endif
        if (tap == 0 && cc + 1 < g.ncc) {      // whole halo of the next chunk, BEFORE this step's weight issue (older => covered)
#pragma unroll
            for (int i = 0; i < 6; ++i)   // static indices: a runtime-indexed hoff[] would live in scratch
                if (i < g.NH) { issue_halo_slice(cc + 1, i); ++issued; }
        }
        if (step + 3 < nsteps) { issue_w(step + 3); issued += SW; }
        mk3 = issued;
        const unsigned char *H = hslot + (cc & 1) * g.halo_bytes;
        const uint16_t *W = reinterpret_cast<const uint16_t *>(wslot + (step & 3) * WSLOT) + (wn * 64 + l31) * BK;
        const int drow = dtab[tap];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int chunk = (ks << 1) | lh;
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int r = rowb[t] + drow;
                fa[t] = *reinterpret_cast<const uint4 *>(H + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) fw[a] = *reinterpret_cast<const uint4 *>(W + a * 32 * BK + ((chunk ^ swz) << 3));
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int t = 0; t < TM; ++t) acc[a][t] = T::mfma(fw[a], fa[t], acc[a][t]);
        }
        mk0 = mk1; mk1 = mk2; mk2 = mk3;
    }

#ifdef TEDSPAD_DEBUG_TS
    if (g_dbg_ts && tid == 0) { unsigned long long *dbg = g_dbg_ts + (size_t)blockIdx.x * 4; dbg[0] = ts0; dbg[1] = ts1; dbg[2] = __builtin_readcyclecounter(); dbg[3] = nsteps; }
#endif
    // ---- epilogue, 128 channels at a time: fp32 [256 px][128 ch] -> LDS -> coalesced rows ----------------------------
    float *stg = reinterpret_cast<float *>(dsm);
    const int cc16 = tid & 15, r0 = tid >> 4;     // 16 chunks of 8 channels per pixel row, 32 rows per pass
    for (int half = 0; half < (BN + 127) / 128; ++half) {
        __syncthreads();
        if ((wn >> 1) == half || BN <= 128) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    const int ml = wm * (256 / WM) + t * 32 + l31;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = (wn & 1) * 64 + a * 32 + 8 * q + 4 * lh;
                        f32x4 v = {acc[a][t][4 * q], acc[a][t][4 * q + 1], acc[a][t][4 * q + 2], acc[a][t][4 * q + 3]};
                        *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + nl) = v;
                    }
                }
        }
        __syncthreads();
        const int nch = n0 + half * 128 + cc16 * 8;
        if (nch < p.Cout && half * 128 + cc16 * 8 < BN) {
            float sc[8], sf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
            for (int r = r0; r < 256; r += 32) {
                if (r >= npix) break;
                const int pw_ = r % g.PW; const int q1 = r / g.PW;
                const int ph_ = q1 % g.PH; const int pt_ = q1 / g.PH;
                const int to = to0 + pt_, ho = ho0 + ph_, wo = wo0 + pw_;
                if (to >= p.To || ho >= p.Ho || wo >= p.Wo) continue;
                const size_t m = (((size_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc16 * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc16 * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
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
                if (p.mask) {
                    float mk[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(p.mask + m * p.ldmask + nch), mk);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = mk[i] > 0.f ? v[i] : 0.f;
                }
                if (p.y) *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8<T>(v);
                if (p.y32) {
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        }
    }
}

// Patch search: maximise the fraction of useful pixels per 256-pixel workgroup, halo <= HALO_MAX_ROWS.
bool choose_patch(const ConvKP &p, HaloGeo &g, int max_rows) {
    double best = 0.0;
    for (int pw = 1; pw <= 32 && pw <= p.Wo; ++pw) {
        if (pw != p.Wo && pw != 32 && pw != 16 && pw != 28 && pw != 14 && pw != 8 && (p.Wo % pw)) continue;
        for (int ph = 1; ph * pw <= 256 && ph <= p.Ho; ++ph)
            for (int pt = 1; pt * ph * pw <= 256 && pt <= p.To; ++pt) {
                const int ht = pt + p.kt - 1, hh = ph + p.kh - 1, wh = pw + p.kw - 1;
                const long nh = (long)ht * hh * wh;
                if (nh > max_rows) continue;
                const long tiles = (long)((p.To + pt - 1) / pt) * ((p.Ho + ph - 1) / ph) * ((p.Wo + pw - 1) / pw);
                const double eff = (double)p.To * p.Ho * p.Wo / (tiles * 256.0) - 1e-6 * nh / (pt * ph * pw);
                if (eff > best) {
                    best = eff;
                    g.PT = pt; g.PH = ph; g.PW = pw; g.HT = ht; g.HH = hh; g.WH = wh; g.nH = (int)nh;
                }
            }
    }
    return best > 0.0;
}

template <typename T, int BN>
int32_t launch_t(const ConvKP &p, int N, int cin, hipStream_t s) {
    HaloGeo g;
    if (!choose_patch(p, g, BN == 128 ? 320 : 384)) {
        set_error("tedspad_conv_fwd: halo-direct config: no patch fits");
        return TEDSPAD_EINVAL;
    }
    g.NH = (g.nH + 63) / 64;
    g.halo_bytes = g.NH * 64 * 128;
    g.ncc = cin / 64;
    g.ntaps = p.kt * p.kh * p.kw;
    g.SL = (g.NH + g.ntaps - 1) / g.ntaps;
    g.tiles_t = (p.To + g.PT - 1) / g.PT; g.tiles_h = (p.Ho + g.PH - 1) / g.PH; g.tiles_w = (p.Wo + g.PW - 1) / g.PW;
    g.tiles_n = (p.Cout + BN - 1) / BN;
    const int main_bytes = 2 * g.halo_bytes + 4 * BN * BK * 2 + g.ntaps * 4 + 16;
    const int stage_bytes = 256 * (128 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    if (p.kt * p.kh * p.kw < 3 || ((p.Cout + 127) / 128 * 128) % BN != 0) {
        set_error("tedspad_conv_fwd: halo-direct config: padded cout is not a multiple of %d", BN);
        return TEDSPAD_EINVAL;
    }
    if (lds > 160 * 1024 || g.NH > 6 || g.SL > 6) {
        set_error("tedspad_conv_fwd: halo-direct config does not fit LDS");
        return TEDSPAD_EINVAL;
    }
    auto kfn = conv_halo_kernel<T, BN>;
    static thread_local bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr = true;
    }
    const long grid = (long)N * g.tiles_t * g.tiles_h * g.tiles_w * g.tiles_n;
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), lds, s, p, g);
    return check_launch("tedspad_conv_fwd(halo)");
}

}  // namespace

int32_t launch_conv_halo(int dtype, const ConvKP &p, int N, int cin, int bn, hipStream_t s) {
    if (cin % 64 != 0 || p.st != 1 || p.sh != 1 || p.sw != 1 || p.kt * p.kh * p.kw < 2 || p.kt * p.kh * p.kw > 32 || p.stats || p.ostrided ||
        p.sigmoid || p.Kpad != p.kt * p.kh * p.kw * cin) {
        set_error("tedspad_conv_fwd: halo-direct config needs a stride-1 multi-tap conv with cin %% 64 == 0 and no stats/strided output");
        return TEDSPAD_EINVAL;
    }
    if (bn == 64) return dtype == TEDSPAD_F16 ? launch_t<F16, 64>(p, N, cin, s) : launch_t<BF16, 64>(p, N, cin, s);
    return dtype == TEDSPAD_F16 ? launch_t<F16, 128>(p, N, cin, s) : launch_t<BF16, 128>(p, N, cin, s);
}

}  // namespace tedspad

#ifdef TEDSPAD_DEBUG_TS
extern "C" int32_t tedspad_debug_set_halo_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tedspad::g_dbg_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
