// The block that ends the default anonymizer's decoder, in one launch (arch = 'unet++', aux_code/model_loaders.py:17-30; smp 0.3.3
// decoders/unetplusplus/decoder.py: blocks['x_0_3'] = DecoderBlock(64, 0, 32), base/heads.py SegmentationHead(32, 3, kernel_size = 3)):
//     y = conv3x3_{32->3}( relu(bn(conv3x3_{32->32}( relu(bn(conv3x3_{64->32}( interpolate(x, 2, 'nearest') ))) ))) ) + bias            (fp32 NCHW out)
// at FULL resolution (16 x 224 x 224 pixels per clip). As separate launches these layers are latency-bound streams (profiles/r04_anon_extract_kernels_1stream.md:
// 64 -> 32 at 2.0 ms, 32 -> 32 at 1.4 ms, 32 -> 3 at 1.1-1.4 ms per 400 frames, each moving 1.3-3.9 GB for 0.2-0.7 TFLOP); fused, HBM sees the half-resolution
// 64-channel input (0.64 GB) and the 3-channel fp32 output (0.24 GB).
//
// One PERSISTENT 8-wave workgroup per CU; ALL weights (63 KB) are loaded into LDS once per launch; the workgroup walks 16 x 16 output patches:
//   X   the patch's input: 12 x 12 half-resolution positions x 64 channels as two half-chunk images (64-byte positions, piece c of position p at
//       c ^ ((p >> 1) & 3): the 16 x 16 x 32 MFMA's fragment read is conflict-free on it), double-buffered: the next patch's X lands while this one is computed;
//   M1  conv1's output on the 20 x 20 halo conv2 needs, 32 channels = one 64-byte position each; positions outside the image are ZERO (conv2 pads conv1's
//       output, not its input);  M2  conv2's output on 18 x 18, likewise;  the head writes y straight from the accumulators.
// The nearest x2 upsample is the fragment address: M1 position (r, c), tap (dh, dw) reads X position ((r + dh + 1) >> 1, (c + dw + 1) >> 1).
// Work split: 16-pixel groups of the stage's output (25 / 21 / 16 of them) dealt round-robin to the 8 waves, every wave computing all of a group's channels.
// Three barriers per patch; no weight stream, no per-tap barrier. conv1 re-computes a 20 x 20 halo per 16 x 16 patch (1.56 x), conv2 18 x 18 (1.27 x).
#include <type_traits>

#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16t;

constexpr int UT_W1 = 2 * 9 * 32 * 64, UT_W2 = 9 * 32 * 64, UT_W3 = 9 * 16 * 64;     // bytes: [hc][tap][co][32 ci], [tap][co][32], [tap][16 co][32]
constexpr int UT_WIMG = UT_W1 + UT_W2 + UT_W3;                                      // 64512
constexpr int UT_M1P = 20 * 20, UT_M2P = 18 * 18, UT_XP = 12 * 12;
constexpr int UT_OFF_M1 = UT_WIMG, UT_OFF_M2 = UT_OFF_M1 + UT_M1P * 64, UT_OFF_X = UT_OFF_M2 + UT_M2P * 64;
constexpr int UT_XBYTES = 2 * UT_XP * 64;                                            // 18432 per buffer
constexpr int UT_OFF_V = UT_OFF_X + 2 * UT_XBYTES;                                   // sc1, sh1, sc2, sh2 (32 floats each), b3 (4): read per stage epilogue (36 registers less in the loops)
constexpr int UT_LDS = UT_OFF_V + 4 * 128 + 16;                                      // 148240
static_assert(UT_LDS <= 160 * 1024 && UT_WIMG % 1024 == 0, "LDS budget");

struct UppTailKP {
    const uint16_t *x;
    const uint16_t *wimg;
    const float *sc1, *sh1, *sc2, *sh2, *b3;
    float *y;
    int ldx, N, H, W, tiles_h, tiles_w, npatch;
};

template <typename T>
__device__ __forceinline__ unsigned pack2(float a, float b);
template <>
__device__ __forceinline__ unsigned pack2<F16>(float a, float b) {
    return (unsigned)F16::from_f32_lim(a, 65504.f) | ((unsigned)F16::from_f32_lim(b, 65504.f) << 16);
}
template <>
__device__ __forceinline__ unsigned pack2<BF16>(float a, float b) {
    return (unsigned)BF16::from_f32_lim(a, 3.3e38f) | ((unsigned)BF16::from_f32_lim(b, 3.3e38f) << 16);
}

template <typename T>
__global__ __launch_bounds__(512) void unetpp_tail_kernel(const UppTailKP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16t);
    const int l15 = lane & 15, kg = lane >> 4;
    const int H2 = p.H >> 1, W2 = p.W >> 1;

    // ---- the weight image, once ---------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i * 512 + wave * 64 < UT_WIMG / 16) lds_dma16(p.wimg + (size_t)(i * 512 + tid) * 8, lds0 + (i * 512 + wave * 64) * 16);

    // ---- X slots of this thread: slot s -> half chunk s / 576, position (s % 576) >> 2, LDS piece s & 3 (the same for every patch) -------------------
    int xr[3], xc[3], xo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int s = i * 512 + tid, hcx = s >= 4 * UT_XP ? 1 : 0, r = s - hcx * 4 * UT_XP, pos = r >> 2;
        xr[i] = pos / 12; xc[i] = pos - xr[i] * 12;
        xo[i] = hcx * 32 + (((r & 3) ^ ((pos >> 1) & 3)) << 3);
    }
    auto patch_org = [&](int pi, int &f, int &oy, int &ox) {
        const int tx = pi % p.tiles_w, t2 = pi / p.tiles_w;
        ox = tx * 16; oy = (t2 % p.tiles_h) * 16; f = t2 / p.tiles_h;
    };
    auto issue_x = [&](int pi, int buf) {
        int f, oy, ox;
        patch_org(pi, f, oy, ox);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i * 512 + wave * 64 >= 8 * UT_XP) break;               // 1152 slots: the third instruction in waves 0 and 1 only
            const int y2 = (oy >> 1) - 2 + xr[i], x2 = (ox >> 1) - 2 + xc[i];
            const bool in = (unsigned)y2 < (unsigned)H2 && (unsigned)x2 < (unsigned)W2;
            lds_dma16(in ? p.x + ((size_t)(f * H2 + y2) * W2 + x2) * p.ldx + xo[i] : zero, lds0 + UT_OFF_X + buf * UT_XBYTES + (i * 512 + wave * 64) * 16);
        }
    };

    // ---- per-lane constants -----------------------------------------------------------------------------------------------------------------------------
    const unsigned wrd = (unsigned)(l15 * 64 + ((kg ^ ((l15 >> 1) & 3)) << 4));      // this lane's piece of weight row (16 a + l15)
    if (tid < 32) {
        float *vv = reinterpret_cast<float *>(dsm + UT_OFF_V);
        vv[tid] = p.sc1[tid]; vv[32 + tid] = p.sh1[tid]; vv[64 + tid] = p.sc2[tid]; vv[96 + tid] = p.sh2[tid];
        if (tid < 4) vv[128 + tid] = tid < 3 ? p.b3[tid] : 0.f;
    }
    auto vec4 = [&](int which, int a) -> f32x4 {            // this lane's channels 16 a + 4 kg .. + 3 of vector `which`
        return *reinterpret_cast<const f32x4 *>(dsm + UT_OFF_V + which * 128 + (16 * a + 4 * kg) * 4);
    };
    using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;

    // The patch loop exists four times, one per (conv1, conv2, head) group counts of a wave -- (4, 3, 0) wave 0, (3, 3, 3) waves 1-2, (3, 3, 2) waves 3-4, (3, 2, 2) waves 5-7 --,
    // chosen OUTSIDE it: the group counts are compile-time constants inside (no `if (j < ng)` cutting a tap into basic blocks), and each copy hoists only its own fragment
    // addresses out of its loop (chosen inside the loop, both bodies' addresses were hoisted: 74 spilled registers; re-derived per patch they cost more VALU than the pipelining gained).
    // TWO stages per patch: [conv1 of this patch + the HEAD of the patch before] | [conv2]. The head reads M2, conv1 reads X and writes M1: independent, so the head's
    // 16 row groups go to the waves conv1 leaves lighter (wave 0 carries conv1's 25th group and no head group): one barrier and the head's 18 MFMAs less on the critical wave.
    auto run = [&](auto ng1_c, auto ng2_c, auto ng3_c) {
    int pi = blockIdx.x;
    if (pi < p.npatch) issue_x(pi, 0);
    int buf = 0;
    int pf = -1, poy = 0, pox = 0;                         // the patch before (its head is still to do)
    // ---- head: 32 -> 3 (+ bias) on 16 x 16: output rows (wave - 1) + 7 j of waves 1-7; fp32 NCHW straight from the accumulators ----------------------------------
    auto head = [&](int f, int oy, int ox) {
        constexpr int NG = decltype(ng3_c)::value;
        if constexpr (NG > 0) {
            f32x4 acc[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            uint4 fw[2], fa[2][NG];
            auto load = [&](int t, uint4 &w, uint4 (&x)[NG]) {
                const int dh = t / 3, dw = t - 3 * dh;
                w = *reinterpret_cast<const uint4 *>(dsm + (unsigned)(UT_W1 + UT_W2 + t * 16 * 64) + wrd);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int mp = (wave - 1 + 7 * j + dh) * 18 + l15 + dw;
                    x[j] = *reinterpret_cast<const uint4 *>(dsm + UT_OFF_M2 + mp * 64 + ((kg ^ ((mp >> 1) & 3)) << 4));
                }
            };
            load(0, fw[0], fa[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + NG, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) load(t + 1, fw[(t + 1) & 1], fa[(t + 1) & 1]);
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[j] = T::mfma16(fw[t & 1], fa[t & 1][j], acc[j]);
                if (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NG, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
            }
            const f32x4 b3 = *reinterpret_cast<const f32x4 *>(dsm + UT_OFF_V + 4 * 128);
            if (kg == 0) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int yy = oy + wave - 1 + 7 * j, xx = ox + l15;
                    if (yy < p.H && xx < p.W) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) p.y[(((size_t)f * 3 + e) * p.H + yy) * p.W + xx] = acc[j][e] + b3[e];
                    }
                }
            }
        }
    };
    for (; pi < p.npatch; pi += gridDim.x, buf ^= 1) {
        int f, oy, ox;
        patch_org(pi, f, oy, ox);
        wait_vmcnt<0>();                                   // this patch's X (and, the first time, the weights) landed; the stores of the head before last are out
        __syncthreads();                                   // ... in every wave; and every wave is done with conv2 of the patch before (M2 complete, M1 free) and the other X buffer
        if (pi + (int)gridDim.x < p.npatch) issue_x(pi + gridDim.x, buf ^ 1);

        // ---- conv1: 64 (upsampled) -> 32 on the 20 x 20 halo: groups g = wave, wave + 8, ... of 16 consecutive M1 positions ---------------------------
        // The group count is a COMPILE-TIME constant of the body (wave 0: 4, the others 3; `if (j < ng)` inside the tap loop cut it into basic blocks: a tap's reads, a wait,
        // its MFMAs -- 39 % MFMA busy), and the fragments of tap t + 1 are requested before the MFMAs of tap t (two register sets, pinned with sched_group_barrier as in the stem).
        auto conv1 = [&](auto ng_c) {
            constexpr int NG = decltype(ng_c)::value;
            f32x4 acc[NG][2];
            int rr[NG], cq[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int i = (wave + 8 * j) * 16 + l15;
                rr[j] = i / 20; cq[j] = i - rr[j] * 20;
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const unsigned xb = (unsigned)(UT_OFF_X + buf * UT_XBYTES);
            uint4 fw[2][2], fa[2][NG];
#pragma unroll 1
            for (int hc = 0; hc < 2; ++hc) {
                auto load = [&](int t, uint4 (&w)[2], uint4 (&x)[NG]) {
                    const int dh = t / 3, dw = t - 3 * dh;
                    const unsigned wb = (unsigned)((hc * 9 + t) * 32 * 64) + wrd;
#pragma unroll
                    for (int a = 0; a < 2; ++a) w[a] = *reinterpret_cast<const uint4 *>(dsm + wb + a * 1024);
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        const int xp = ((rr[j] + dh + 1) >> 1) * 12 + ((cq[j] + dw + 1) >> 1);
                        x[j] = *reinterpret_cast<const uint4 *>(dsm + xb + hc * (UT_XP * 64) + xp * 64 + ((kg ^ ((xp >> 1) & 3)) << 4));
                    }
                };
                load(0, fw[0], fa[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + NG, 0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) load(t + 1, fw[(t + 1) & 1], fa[(t + 1) & 1]);
#pragma unroll
                    for (int j = 0; j < NG; ++j)
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[j][a] = T::mfma16(fw[t & 1][a], fa[t & 1][j], acc[j][a]);
                    if (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NG, 0);      // the LDS reads of tap t + 1
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NG, 0);                      // the MFMAs of tap t
                }
            }
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { sc[a] = vec4(0, a); sh[a] = vec4(1, a); }
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int i = (wave + 8 * j) * 16 + l15;
                const bool in = (unsigned)(oy - 2 + rr[j]) < (unsigned)p.H && (unsigned)(ox - 2 + cq[j]) < (unsigned)p.W;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = in ? __builtin_fmaxf(acc[j][a][e] * sc[a][e] + sh[a][e], 0.f) : 0.f;
                    uint2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    *reinterpret_cast<uint2 *>(dsm + UT_OFF_M1 + i * 64 + (((2 * a + (kg >> 1)) ^ ((i >> 1) & 3)) << 4) + (kg & 1) * 8) = pk;
                }
            }
        };
        conv1(ng1_c);                                       // 25 groups: 4 on wave 0, 3 on the others
        if (pf >= 0) head(pf, poy, pox);                    // the patch before: its M2 is complete since the barrier above
        __syncthreads();

        // ---- conv2: 32 -> 32 on 18 x 18: 21 groups of 16 consecutive M2 positions (the last one has 4) ------------------------------------------------
        auto conv2 = [&](auto ng_c) {
            constexpr int NG = decltype(ng_c)::value;
            f32x4 acc[NG][2];
            int rr[NG], cq[NG], idx[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int i = (wave + 8 * j) * 16 + l15;
                idx[j] = i;
                const int ic = i < UT_M2P ? i : UT_M2P - 1;
                rr[j] = ic / 18; cq[j] = ic - rr[j] * 18;
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            uint4 fw[2][2], fa[2][NG];
            auto load = [&](int t, uint4 (&w)[2], uint4 (&x)[NG]) {
                const int dh = t / 3, dw = t - 3 * dh;
                const unsigned wb = (unsigned)(UT_W1 + t * 32 * 64) + wrd;
#pragma unroll
                for (int a = 0; a < 2; ++a) w[a] = *reinterpret_cast<const uint4 *>(dsm + wb + a * 1024);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int mp = (rr[j] + dh) * 20 + cq[j] + dw;
                    x[j] = *reinterpret_cast<const uint4 *>(dsm + UT_OFF_M1 + mp * 64 + ((kg ^ ((mp >> 1) & 3)) << 4));
                }
            };
            load(0, fw[0], fa[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 + NG, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) load(t + 1, fw[(t + 1) & 1], fa[(t + 1) & 1]);
#pragma unroll
                for (int j = 0; j < NG; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a) acc[j][a] = T::mfma16(fw[t & 1][a], fa[t & 1][j], acc[j][a]);
                if (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NG, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NG, 0);
            }
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { sc[a] = vec4(2, a); sh[a] = vec4(3, a); }
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (idx[j] < UT_M2P) {
                    const int i = idx[j];
                    const bool in = (unsigned)(oy - 1 + rr[j]) < (unsigned)p.H && (unsigned)(ox - 1 + cq[j]) < (unsigned)p.W;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = in ? __builtin_fmaxf(acc[j][a][e] * sc[a][e] + sh[a][e], 0.f) : 0.f;
                        uint2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                        *reinterpret_cast<uint2 *>(dsm + UT_OFF_M2 + i * 64 + (((2 * a + (kg >> 1)) ^ ((i >> 1) & 3)) << 4) + (kg & 1) * 8) = pk;
                    }
                }
        };
        conv2(ng2_c);                                       // 21 groups: 3 on waves 0-4, 2 on waves 5-7
        pf = f; poy = oy; pox = ox;
    }
    __syncthreads();                                       // the last patch's M2
    if (pf >= 0) head(pf, poy, pox);
    };
    if (wave == 0) run(I4{}, I3{}, I0{});
    else if (wave < 3) run(I3{}, I3{}, I3{});               // head rows (wave - 1) + 7 j < 16: three on waves 1 and 2
    else if (wave < 5) run(I3{}, I3{}, I2{});
    else run(I3{}, I2{}, I2{});
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_unetpp_tail_wimg_bytes(void) { return UT_WIMG; }

extern "C" int32_t tedspad_unetpp_tail_fwd(const void *x, int32_t ldx, float *y, int32_t n, int32_t h, int32_t w, const void *w_img, const float *scale1,
                                           const float *shift1, const float *scale2, const float *shift2, const float *bias3, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && w_img && scale1 && shift1 && scale2 && shift2 && bias3, "tedspad_unetpp_tail_fwd: null pointer");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_img) % 16 == 0, "tedspad_unetpp_tail_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && ldx >= 64 && ldx % 8 == 0, "tedspad_unetpp_tail_fwd: h, w even; ldx >= 64, multiple of 8");
    TS_REQUIRE((long)n * (h / 2) * (w / 2) * ldx < (1L << 31) && (long)n * 3 * h * w < (1L << 31), "tedspad_unetpp_tail_fwd: tensor too large; split the batch");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_unetpp_tail_fwd: dtype");
    UppTailKP p;
    p.x = (const uint16_t *)x; p.wimg = (const uint16_t *)w_img; p.sc1 = scale1; p.sh1 = shift1; p.sc2 = scale2; p.sh2 = shift2; p.b3 = bias3; p.y = y;
    p.ldx = ldx; p.N = n; p.H = h; p.W = w; p.tiles_h = (h + 15) / 16; p.tiles_w = (w + 15) / 16; p.npatch = n * p.tiles_h * p.tiles_w;
    static thread_local int attr_set[2] = {0, 0};
    static thread_local int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("tedspad_unetpp_tail_fwd: cannot query the device");
            return TEDSPAD_ELAUNCH;
        }
        ncu = prop.multiProcessorCount;
    }
    const int grid = p.npatch < ncu ? p.npatch : ncu;
    if (dtype == TEDSPAD_F16) {
        auto kfn = unetpp_tail_kernel<F16>;
        if (!attr_set[0]) {
            if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("tedspad_unetpp_tail_fwd: cannot raise the dynamic LDS limit");
                return TEDSPAD_ELAUNCH;
            }
            attr_set[0] = 1;
        }
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), UT_LDS, (hipStream_t)stream, p);
    } else {
        auto kfn = unetpp_tail_kernel<BF16>;
        if (!attr_set[1]) {
            if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                set_error("tedspad_unetpp_tail_fwd: cannot raise the dynamic LDS limit");
                return TEDSPAD_ELAUNCH;
            }
            attr_set[1] = 1;
        }
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), UT_LDS, (hipStream_t)stream, p);
    }
    return check_launch("tedspad_unetpp_tail_fwd");
}
