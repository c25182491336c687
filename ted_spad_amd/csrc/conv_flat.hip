// Flat-halo convolution for gfx950 (tile_cfg 27): stride-1 "same" 2-D convs (1 x kh x kw) with cin = 64 and cout <= 64 -- conv2 of
// the layer1 bottlenecks (large_i3d.py:49), the 64 -> 64 convs of the UNet's outer level.
//
// Every generic tile lands on the same ~515 TFLOP/s on these layers, with a time proportional to K: the implicit GEMM re-fetches each
// input pixel once per tap and the L2 -> LDS DMA stream (~33 GB/s per CU beside running MFMAs) is what it waits for. With
// stride 1 and channels-last storage, the taps of output pixel q (flattened (n,t,h,w) index) are the pixels q + (dh-ph)*W + (dw-pw):
// the input HALO of a tile of 256 consecutive output pixels is ONE CONTIGUOUS run of 256 + (kh-1)*W + (kw-1) pixels -- a linear
// copy, no gather arithmetic -- fetched once (47 KB for W = 55 instead of 9 x 32 KB), and every MFMA pixel fragment is read from it at
// `pixel + tap delta`. Taps that fall outside the frame (which, flattened, alias pixels of the neighbouring row / frame / clip) are
// redirected per lane to a zero position by a 9-bit validity mask computed once per pixel. Only the [64 co][64 k] weight tile of
// a tap streams (3-slot ring, one stage in flight across every barrier). One K tile = one tap (64 channels, four k16 sub-steps).
//
// LDS image: positions of 128 bytes; chunk c of position p is stored at chunk c ^ ((p >> 1) & 7) (applied to the DMA source
// address and to the read address): 16 consecutive positions then cover all 64 banks in every 16-lane group of a ds_read_b128
// (the position's parity selects the half of the 256-byte bank row, the XOR spreads the eight positions of a parity).
// 4 waves, each 64 pixels x 64 channels; <= 80 KB for W <= 120, so two workgroups share a CU and one's halo load / epilogue
// runs under the other's MFMAs (the structure of conv_stem_halo_kernel, which this kernel follows).
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16f;

constexpr int FL_BM = 256;
constexpr int FL_WSTAGE = 64 * BK * 2;

struct FlatGeo {
    int W, H, R, NP, ntaps;     // R = ph*W + pw (halo positions in front of the tile); NP = 256 + (kh-1)*W + (kw-1)
};

template <typename T>
__global__ __launch_bounds__(256) void conv_flat_kernel(const ConvKP p, const FlatGeo g) {
    constexpr int NT = 256, WS = 3;      // weight ring: stage kt+2 is issued in step kt, one stage stays in flight across every barrier
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int q0 = tile * FL_BM;
    const int S = (g.NP + 1) * 8;                       // 16-byte slots: the halo + one zero position
    const int Sr = (S + 63) / 64 * 64;
    const int halo_bytes = Sr * 16;
    unsigned char *wring = dsm + halo_bytes;            // [WS][64][64] 16-bit
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16f);

    // ---- weights of tap 0, then the halo: a linear copy with the chunk swizzle on the source address -------------------------
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w + (size_t)rsub * p.Kpad + kc * 8;
    auto issue_w = [&](int kt, int slot) {
        const unsigned dst = lds0 + halo_bytes + slot * FL_WSTAGE + wave * 8 * (BK * 2);
        lds_dma16(wsrc + kt * BK, dst);
        lds_dma16(wsrc + (size_t)32 * p.Kpad + kt * BK, dst + 32 * (BK * 2));
    };
    issue_w(0, 0);                                       // issue order w(0), halo, w(1): the counted waits below rely on it
    const int NH = (Sr + NT - 1) / NT;
    for (int i = 0; i < NH; ++i) {
        if (i * NT + wave * 64 >= Sr) break;             // wave-uniform
        const int s = i * NT + tid;
        const int pos = s >> 3, cs = s & 7;
        const int q = q0 - g.R + pos;
        const bool ok = pos < g.NP && (unsigned)q < (unsigned)p.M;
        const uint16_t *src = ok ? p.x + (size_t)q * p.ldx + ((cs ^ ((pos >> 1) & 7)) << 3) : zero;
        lds_dma16(src, lds0 + (i * NT + wave * 64) * 16);
    }
    if (g.ntaps > 1) issue_w(1, 1);

    // ---- MFMA roles: wave w owns pixels 64w .. 64w+63 (two 32-pixel fragments) x 64 channels -----------------------------------
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int pj[2];
    unsigned vmask[2];        // bit (dh*kw + dw): the tap lies inside the frame
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = wave * 64 + b * 32 + l31;
        pj[b] = j;
        const int q = q0 + j;
        unsigned mk = 0;
        if (q < p.M) {
            const int r1 = q / g.W, w = q - r1 * g.W;
            const int h = r1 % g.H;
            for (int dh = 0; dh < p.kh; ++dh)
                for (int dw = 0; dw < p.kw; ++dw)
                    if ((unsigned)(h + dh - p.ph) < (unsigned)g.H && (unsigned)(w + dw - p.pw) < (unsigned)g.W) mk |= 1u << (dh * p.kw + dw);
        }
        vmask[b] = mk;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (g.ntaps > 1) wait_vmcnt<2>(); else wait_vmcnt<0>();   // halo + weight stage 0 of this wave have landed (stage 1 may still fly)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int dh = 0, dw = 0;
    for (int kt = 0; kt < g.ntaps; ++kt) {
        // this tap's pixel positions and its first pixel fragments: they depend on the resident halo only, not on the weight
        // stage, so they are requested BEFORE the wait / barrier of the step
        const int delta = dh * g.W + dw;
        unsigned xoff[2], xswz[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pos = ((vmask[b] >> kt) & 1u) ? pj[b] + delta : g.NP;
            xoff[b] = (unsigned)pos * 128u;
            xswz[b] = (unsigned)(pos >> 1) & 7u;
        }
        const uint4 fa00 = *reinterpret_cast<const uint4 *>(dsm + xoff[0] + (((unsigned)lh ^ xswz[0]) << 4));
        const uint4 fa01 = *reinterpret_cast<const uint4 *>(dsm + xoff[1] + (((unsigned)lh ^ xswz[1]) << 4));
        if (kt + 1 < g.ntaps) wait_vmcnt<2>(); else wait_vmcnt<0>();   // stage kt landed; stage kt+1 (2 instructions) may stay in flight
        __builtin_amdgcn_s_barrier();   // ... of every wave; the slot of stage kt-1 is free (on kt = 0 this repeats the barrier above)
        asm volatile("" ::: "memory");
        if (kt + 2 < g.ntaps) issue_w(kt + 2, (kt + 2) % WS);
        const uint16_t *Wt = reinterpret_cast<const uint16_t *>(wring + (kt % WS) * FL_WSTAGE) + l31 * BK;
        {   // k16 sub-step 0 on the fragments requested before the barrier
            const unsigned c = (unsigned)lh;
            const uint4 fw0 = *reinterpret_cast<const uint4 *>(Wt + ((c ^ swz) << 3)), fw1 = *reinterpret_cast<const uint4 *>(Wt + 32 * BK + ((c ^ swz) << 3));
            acc[0][0] = T::mfma(fw0, fa00, acc[0][0]);
            acc[0][1] = T::mfma(fw0, fa01, acc[0][1]);
            acc[1][0] = T::mfma(fw1, fa00, acc[1][0]);
            acc[1][1] = T::mfma(fw1, fa01, acc[1][1]);
        }
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
            const unsigned c = (unsigned)((ks << 1) | lh);
            uint4 fa[2], fw[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) fa[b] = *reinterpret_cast<const uint4 *>(dsm + xoff[b] + ((c ^ xswz[b]) << 4));
#pragma unroll
            for (int a = 0; a < 2; ++a) fw[a] = *reinterpret_cast<const uint4 *>(Wt + a * 32 * BK + ((c ^ swz) << 3));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = T::mfma(fw[a], fa[b], acc[a][b]);
        }
        if (++dw == p.kw) { dw = 0; ++dh; }
    }
    __syncthreads();

    // ---- epilogue: fp32 tile [256 px][64 co] -> LDS -> coalesced 16-byte rows (as conv_stem_halo_kernel) ----------------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ml = wave * 64 + b * 32 + l31;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                f32x4 v = {acc[a][b][4 * qd], acc[a][b][4 * qd + 1], acc[a][b][4 * qd + 2], acc[a][b][4 * qd + 3]};
                *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + a * 32 + 8 * qd + 4 * lh) = v;
            }
        }
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;     // 8 chunks of 8 channels per pixel, 32 pixels per pass
    const int nch = cc * 8;
    if (nch >= p.Cout) return;
    float sc[8], sf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
#pragma unroll
    for (int it = 0; it < FL_BM / 32; ++it) {
        const int r = r0 + it * 32;
        const size_t m = (size_t)q0 + r;
        if (m >= (size_t)p.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch + 4);
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
        *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8_lim<T>(v, p.sat);
    }
}

template <typename T>
int32_t launch_flat_t(const ConvKP &p, hipStream_t s) {
    FlatGeo g;
    g.W = p.Wi; g.H = p.Hi; g.R = p.ph * p.Wi + p.pw; g.ntaps = p.kh * p.kw;
    g.NP = FL_BM + (p.kh - 1) * p.Wi + (p.kw - 1);
    const int S = (g.NP + 1) * 8;
    const int main_bytes = (S + 63) / 64 * 64 * 16 + 3 * FL_WSTAGE;
    const int stage_bytes = FL_BM * (64 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    if (lds > 160 * 1024) {
        set_error("tedspad_conv_fwd: flat-halo config: halo does not fit LDS (%d bytes)", lds);
        return TEDSPAD_EINVAL;
    }
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_flat_kernel<T>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    const int tiles = (p.M + FL_BM - 1) / FL_BM;
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), lds, s, p, g);
    return check_launch("tedspad_conv_fwd(flat halo)");
}

// ------------------------------------------------------------------------------------------------------------------------------
// Temporal flat-halo kernel (tile_cfg 28): stride-1 "same" kt x 1 x 1 convs (conv1 of the bottlenecks with a temporal kernel,
// large_i3d.py:47) with cin % 64 == 0, cout <= 64 and T <= 4 frames. The generic tiles fetch every input pixel once per temporal
// tap; here a workgroup owns 64 consecutive spatial positions of ALL T frames of a clip (wave f = output frame f: 64 px x 64 co), so
// for each 64-channel chunk the T x 64 input positions (32 KB) are fetched once and serve every tap of every output frame: X bytes
// per 256 outputs drop from kt x 4 x 32 KB to 4 x 32 KB for cin = 256. Whether a tap exists (frame f + dt - pt inside the clip) is
// wave-uniform: missing taps are skipped, not multiplied by zeros. K is walked (chunk, tap) instead of (tap, chunk): fp32 sums are
// re-associated with respect to the generic tiles (like the halo-direct tiles 15 / 16).
// ------------------------------------------------------------------------------------------------------------------------------
struct TFlatGeo {
    int T, HW, tiles_s, nchunks, kt, pt;
};

template <typename T_>
__global__ __launch_bounds__(256) void conv_tflat_kernel(const ConvKP p, const TFlatGeo g) {
    constexpr int NT = 256;
    constexpr int NP = 256;                              // positions: frame f (< 4) x 64 pixels; position NP is the zero position
    constexpr int HALO = (NP + 8) * 128;                 // rounded to whole wave instructions (64 slots = 8 positions)
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int n = tile / g.tiles_s, s0 = (tile - n * g.tiles_s) * 64;
    unsigned char *wbuf = dsm + HALO;                    // [kt][64][64] 16-bit
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16f);

    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w + (size_t)rsub * p.Kpad + kc * 8;
    // halo slots of this thread: slot s = i*256 + tid -> position s >> 3 = frame (pos >> 6), pixel (pos & 63); chunk slot s & 7
    int hsrc[9];                                          // element offset of the slot's source (without the channel chunk), or -1
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int s = i * NT + tid;
        const int pos = s >> 3, cs = s & 7;
        const int f = pos >> 6, px = pos & 63;
        hsrc[i] = -1;
        if (pos < NP && f < g.T && s0 + px < g.HW) hsrc[i] = (int)((((long)n * g.T + f) * g.HW + s0 + px) * p.ldx) + ((cs ^ ((pos >> 1) & 7)) << 3);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int ch = 0; ch < g.nchunks; ++ch) {
        if (ch) __builtin_amdgcn_s_barrier();            // every wave has read the previous chunk's halo and weights
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i * NT + wave * 64 >= (NP + 8) * 8) break;     // wave-uniform
            lds_dma16(hsrc[i] >= 0 ? p.x + hsrc[i] + ch * 64 : zero, lds0 + (i * NT + wave * 64) * 16);
        }
        for (int dt = 0; dt < g.kt; ++dt) {
            const unsigned dst = lds0 + HALO + dt * FL_WSTAGE + wave * 8 * (BK * 2);
            const uint16_t *src = wsrc + dt * p.cin + ch * 64;
            lds_dma16(src, dst);
            lds_dma16(src + (size_t)32 * p.Kpad, dst + 32 * (BK * 2));
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int dt = 0; dt < g.kt; ++dt) {
            const int fin = wave + dt - g.pt;             // input frame of this tap for the wave's output frame: wave-uniform
            if (wave >= g.T || fin < 0 || fin >= g.T) continue;
            const uint16_t *Wt = reinterpret_cast<const uint16_t *>(wbuf + dt * FL_WSTAGE) + l31 * BK;
            unsigned xoff[2], xswz[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pos = fin * 64 + b * 32 + l31;
                xoff[b] = (unsigned)pos * 128u;
                xswz[b] = (unsigned)(pos >> 1) & 7u;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned c = (unsigned)((ks << 1) | lh);
                uint4 fa[2], fw[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) fa[b] = *reinterpret_cast<const uint4 *>(dsm + xoff[b] + ((c ^ xswz[b]) << 4));
#pragma unroll
                for (int a = 0; a < 2; ++a) fw[a] = *reinterpret_cast<const uint4 *>(Wt + a * 32 * BK + ((c ^ swz) << 3));
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = T_::mfma(fw[a], fa[b], acc[a][b]);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: staging row f*64 + px -> output pixel (n, f, s0 + px) ------------------------------------------------------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ml = wave * 64 + b * 32 + l31;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                f32x4 v = {acc[a][b][4 * qd], acc[a][b][4 * qd + 1], acc[a][b][4 * qd + 2], acc[a][b][4 * qd + 3]};
                *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + a * 32 + 8 * qd + 4 * lh) = v;
            }
        }
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;
    const int nch = cc * 8;
    if (nch >= p.Cout) return;
    float sc[8], sf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = r0 + it * 32;
        const int f = r >> 6, px = r & 63;
        if (f >= g.T || s0 + px >= g.HW) continue;
        const size_t m = ((size_t)n * g.T + f) * g.HW + s0 + px;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + nch + 4);
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
        if (p.res) {
            float rr[8];
            unpack8<T_>(*reinterpret_cast<const uint4 *>(p.res + m * p.ldres + nch), rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rr[i];
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8_lim<T_>(v, p.sat);
    }
}

template <typename T_>
int32_t launch_tflat_t(const ConvKP &p, int N, int cin, hipStream_t s) {
    TFlatGeo g;
    g.T = p.Ti; g.HW = p.Hi * p.Wi; g.tiles_s = (g.HW + 63) / 64; g.nchunks = cin / 64; g.kt = p.kt; g.pt = p.pt;
    const int main_bytes = (256 + 8) * 128 + p.kt * FL_WSTAGE;
    const int stage_bytes = FL_BM * (64 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_tflat_kernel<T_>;
    if (!attr_set[T_::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T_::kDtype] = 1;
    }
    hipLaunchKernelGGL(kfn, dim3(N * g.tiles_s), dim3(256), lds, s, p, g);
    return check_launch("tedspad_conv_fwd(temporal flat halo)");
}

}  // namespace

int32_t launch_conv_flat(int dtype, const ConvKP &p, int cin, hipStream_t s) {
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi && p.pt == 0 && p.ph < p.kh && p.pw < p.kw;
    if (cin != 64 || p.kt != 1 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same || p.kh * p.kw < 2 || p.kh * p.kw > 32 || p.Cout > 64 ||
        p.Kpad != p.kh * p.kw * 64 || p.mask || p.stats || p.ostrided || p.y32 || p.sigmoid || !p.y) {
        set_error("tedspad_conv_fwd: flat-halo config needs a stride-1 'same' 1 x kh x kw conv with cin = 64, cout <= 64 and the plain epilogue");
        return TEDSPAD_EINVAL;
    }
    return dtype == TEDSPAD_F16 ? launch_flat_t<F16>(p, s) : launch_flat_t<BF16>(p, s);
}

}  // namespace tedspad

namespace tedspad {

int32_t launch_conv_tflat(int dtype, const ConvKP &p, int N, int cin, hipStream_t s) {
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi && p.ph == 0 && p.pw == 0 && p.pt < p.kt;
    if (cin % 64 != 0 || p.kh != 1 || p.kw != 1 || p.kt < 2 || p.kt > 3 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same || p.Ti > 4 || p.Cout > 64 ||
        p.Kpad != p.kt * cin || p.mask || p.stats || p.ostrided || p.y32 || p.sigmoid || !p.y) {
        set_error("tedspad_conv_fwd: temporal flat-halo config needs a stride-1 'same' kt x 1 x 1 conv (kt 2..3) with cin %% 64 == 0, cout <= 64, T <= 4");
        return TEDSPAD_EINVAL;
    }
    return dtype == TEDSPAD_F16 ? launch_tflat_t<F16>(p, N, cin, s) : launch_tflat_t<BF16>(p, N, cin, s);
}

}  // namespace tedspad
