// Whole layer1 bottleneck for gfx950: conv1 (3x1x1 temporal | 1x1x1, cin -> 64) + bn1 + ReLU -> conv2 (1 x 3 x 3, 64 -> 64) + bn2 + ReLU -> conv3 (1x1x1,
// 64 -> 256) + bn3 + residual + ReLU of a plain I3Res50 layer1 block (aux_code/models/large_i3d.py:61-84 without `downsample`, blocks layer1.1 / layer1.2 of
// :142) in ONE launch: x in, y out, both 64-channel tensors between the convolutions stay in LDS.
//
// Why (round-3 review, profiles/r03_bench_cfg2_kernels_1stream.md): layer1 was 27 % of the forward as two launches per block -- the temporal conv1
// (conv_tflat_kernel: reads 512 B, writes 128 B per pixel at 4.0 TB/s) and the fused tail (conv_bneck_tail_kernel: 128 + 512 in, 512 out at 4.6 TB/s) -- both bound by
// the bytes they move, 1 792 B per pixel and block. Fused, a block moves 1 024 B per pixel (+ the halo's share of x, which neighbouring tiles read through L2).
//
// Structure. ONE WORKGROUP = a spatial tile of 8 x 14 output pixels x ALL T <= 4 frames of a clip (55 = 4 x 14 - 1 columns, 7 x 8 - 1 rows: 3.7 % of the tiles'
// pixels lie outside the frame). conv2 needs conv1's output on the 10 x 16 halo of the tile, conv1 is pointwise in space: it is simply computed on the halo
// (1.43 x its FLOP, 23 % of the block's -> + 10 %), so no tile ever talks to a neighbour. 8 waves, v_mfma_f32_16x16x32 (M tiles of 16 rows):
//   * stage 1, conv1: rows = (frame, halo position): T x 160 = 40 M tiles of 16 -- a halo ROW is one tile --, 5 per wave, dealt round-robin so that every wave
//     holds tiles of several frames (a tile of frame t skips the temporal tap that leaves the clip: wave-uniform, no products on zero padding, and no wave idles).
//     K is walked in chunks of 32 input channels: the chunk of ALL frames (T x 160 positions x 64 B = 40 KB, fragment layout [frame][halo row][lane][16 B],
//     written by LDS-DMA with per-lane sources clamped into the frame) serves the three temporal taps, whose [64 co][32 k] weight images (4 KB each, packed by
//     the host in fragment order) arrive beside it; two slots, one barrier per chunk.
//   * relu(bn1(.)) -> M1[frame][halo position][64] in LDS (144-byte rows: conflict-free 16-byte fragment reads), ZERO at halo positions outside the frame (conv2's zero
//     padding pads conv1's OUTPUT);
//   * stage 2, conv2: rows = (frame, tile pixel): T x 112 = 28 M tiles, 4 | 3 per wave; the B fragment of tap (dh, dw) is M1 at the pixel's halo position + dh * 16 + dw:
//     an immediate offset, no tap masks; [64][64] weight images (8 KB) through a 3-slot ring, one barrier per tap;
//   * relu(bn2(.)) -> M2[frame][tile pixel][64] over M1;
//   * stage 3, conv3 + bn3 + residual + ReLU: operand roles swapped (pixels = A) so that a lane ends with 4 consecutive channels of 4 pixels (the layout of
//     conv_bneck_frame.hip): residual loads and stores are coalesced 8-byte accesses straight from / to the accumulator lanes; the whole 32 KB conv3 image is resident,
//     four passes of 64 output channels without a barrier.
#include "conv_common.h"
#include <stdlib.h>

namespace tedspad {
namespace {

struct BneckL1KP {
    const uint16_t *x;          // block input (n, t, h, w, cin), pixel stride ldx: conv1 operand and residual
    uint16_t *y;                // block output (n, t | t / 2, h, w, 256), pixel stride ldy
    const uint16_t *wimg;       // weight units of 4 KB in consumption order: stage 1 (cin / 32 chunks x kt taps), stage 2 (9 taps x 2), stage 3 (4 x 2)
    const float *scale1, *shift1, *scale2, *shift2, *scale3, *shift3;
    int N, T, H, W, ldx, ldy, cin, kt, tiles_h, tiles_w, relu;
    long long *stamps;          // diagnostic (TEDSPAD_L1_STAMPS = address of an int64 buffer, 8 per workgroup): s_memtime at the stage boundaries of wave 0
    int dbg;                    // timing ablations (wrong results; TEDSPAD_L1_ABLATE): 1 no stage 1, 2 no stage 2, 4 no stage 3, 8 no residual loads / stores, 16 no stage-1 x DMA, 32 no MFMAs in stage 1
};

constexpr int L1_TW = 14, L1_HW = 16;                     // tile / halo width
constexpr int L1_ROWB = 160;                              // bytes per M row: 10 sixteen-byte slots. A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19,
                                                          // 28-31} (+ 32): with lane = (row l15, chunk g) those 16 lanes hit 16 different slots mod 16 at strides of 10, 14, 18 slots --
                                                          // and two of them the same one at 9 (144 bytes, the first version: stage 2 ran at its LDS time, 15.6 k cycles for 8 k of MFMAs)
constexpr int L1_TMAX = 4;
constexpr int L1_UNIT = 4096;

// Geometry of a variant: TH output rows per tile, NW waves.
//   TH = 8, NW = 8: one workgroup per CU (156 KB), three x slots.
//   TH = 4, NW = 4: TWO workgroups per CU (78 KB each): while one streams its x chunks from HBM (a CU pulls ~23 GB/s of misses whatever it does) the other multiplies
//                   or stores -- the stages of ONE workgroup run one after the other. conv1 is recomputed on 6 x 16 halos (1.71 x instead of 1.43 x).
template <int TH, int NW>
struct L1Geo {
    static constexpr int HH = TH + 2, NPOS = HH * L1_HW, NPX = TH * L1_TW;
    static constexpr int NXS = TH == 8 ? 3 : 2;                                   // x slots
    static constexpr int XSLOT = L1_TMAX * NPOS * 64;
    static constexpr int M1B = L1_TMAX * NPOS * L1_ROWB;
    static constexpr int NT2 = (L1_TMAX * NPX + 15) / 16;                          // stage-2 / stage-3 M tiles of a full clip
    static constexpr int M2B = NT2 * 16 * L1_ROWB;
    static constexpr int EXB = 0;
    static constexpr int R0 = (NXS * XSLOT > M1B ? NXS * XSLOT : M1B) > M2B + EXB ? (NXS * XSLOT > M1B ? NXS * XSLOT : M1B) : M2B + EXB;
    static constexpr int RWU = TH == 8 ? 9 : 6;                                    // weight region in units
    static constexpr int ZOFF = R0 + RWU * L1_UNIT;                                // 1 KB of zeros: the pixel fragment of a temporal tap outside the clip
    static constexpr int LDS = ZOFF + 1024;
    static constexpr int NI1 = L1_TMAX * HH / NW;                                  // stage-1 tiles per wave
    static constexpr int NI2 = (NT2 + NW - 1) / NW;                                // stage-2 / 3 tiles per wave (at most)
    static_assert(L1_TMAX * HH % NW == 0 && LDS <= (NW == 8 ? 160 : 80) * 1024, "tiles deal evenly over the waves; the LDS image fits");
};

__device__ uint4 g_l1_zero;
__device__ uint2 g_l1_sink[64];      // where the results of pixels outside the frame go (never read)

#define L1_LDS16(off) (*reinterpret_cast<const uint4 *>(dsm + (off)))

// relu(a * s + b) of two neighbouring channels, saturated at hi (0: the pair is zero -- a halo position outside the frame), rounded to a packed 16-bit pair:
// fma, v_med3_f32 (ReLU and saturation together) and, for f16, one v_cvt_pk_f16_f32 for both: 2.5 vector instructions per value
template <typename T_>
__device__ __forceinline__ unsigned l1_pair(float a0, float a1, float s0, float s1, float b0, float b1, float hi) {
    const float v0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a0, s0, b0), 0.f, hi), v1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a1, s1, b1), 0.f, hi);
    if constexpr (T_::kDtype == TEDSPAD_F16) {
        unsigned pk;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v0), "v"(v1));
        return pk;
    } else {
        return (unsigned)T_::from_f32(v0) | ((unsigned)T_::from_f32(v1) << 16);
    }
}
// bn3 + residual + ReLU of two neighbouring channels (the epilogue of conv_bneck.hip's tails): v_fma_mix_f32 adds the f16 residual half to the shift without a
// conversion, one fma, one med3, one packed conversion: 3.5 instructions per value
template <typename T_>
__device__ __forceinline__ unsigned l1_pair_res(float a0, float a1, float s0, float s1, float b0, float b1, unsigned res, float lo) {
    if constexpr (T_::kDtype == TEDSPAD_F16) {
        unsigned pk;
        float t0, t1;
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(res), "v"(b0));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(res), "v"(b1));
        const float v0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a0, s0, t0), lo, 65504.f), v1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a1, s1, t1), lo, 65504.f);
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v0), "v"(v1));
        return pk;
    } else {
        const float o0 = __builtin_fmaf(a0, s0, b0) + T_::to_f32((uint16_t)(res & 0xffffu)), o1 = __builtin_fmaf(a1, s1, b1) + T_::to_f32((uint16_t)(res >> 16));
        return (unsigned)T_::from_f32(__builtin_fmaxf(o0, lo)) | ((unsigned)T_::from_f32(__builtin_fmaxf(o1, lo)) << 16);
    }
}


template <typename T_, int TH, int NW, bool POOL>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void bneck_l1_kernel(const BneckL1KP p) {
    typedef L1Geo<TH, NW> G;
    constexpr int HH = G::HH, NPX = G::NPX, NI1 = G::NI1, NI2 = G::NI2, RW = G::R0;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = tile % p.tiles_w; tile /= p.tiles_w;
    const int th = tile % p.tiles_h;
    const int n = tile / p.tiles_h;
    const int oh0 = th * TH, ow0 = tw * L1_TW;
    const int T = p.T, nc1 = (p.dbg & 1) ? 0 : p.cin >> 5, kt = p.kt, pt = kt >> 1;
    constexpr float HI = T_::kDtype == TEDSPAD_F16 ? 65504.f : 3.3e38f;

    // ---- stage 1 sources: piece pc = (frame f, halo row hr) = wave + NW i; lane (hc = l15, k group g) <- x[n, f, clamp(oh0 - 1 + hr), clamp(ow0 - 1 + hc), 32 c + 8 g ..]
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x + (size_t)n * T * p.H * p.W * p.ldx);
    const unsigned colo = ((unsigned)min(max(ow0 - 1 + l15, 0), p.W - 1) * (unsigned)p.ldx + (unsigned)g * 8u) * 2u;
    // Tiles of a wave: i < 4: (frame i, halo row `wave`) -- the frame, hence which temporal taps stay inside the clip, is a COMPILE-TIME property of i: no branch in the
    // K loop; i >= 4: the halo rows NW .. HH - 1 of all frames, dealt over the waves (their frame depends on the wave: a tap outside the clip multiplies a zero fragment)
    unsigned rowo[NI1];
    int t1[NI1], hr1[NI1], pc1[NI1];
#pragma unroll
    for (int i = 0; i < NI1; ++i) {
        int f = i, hr = wave;
        if (i >= 4) {
            const int idx = wave * (NI1 - 4) + (i - 4);
            f = idx / (HH - NW);
            hr = NW + idx - f * (HH - NW);
        }
        const bool has = f < T;
        t1[i] = has ? f : -8;                                   // no tile: every tap "leaves the clip"
        hr1[i] = hr;
        pc1[i] = has ? f * HH + hr : -1;
        rowo[i] = (unsigned)((min(f, T - 1) * p.H + min(max(oh0 - 1 + hr, 0), p.H - 1)) * p.W) * (unsigned)p.ldx * 2u;
    }
    auto issue_x = [&](int c, int slot) {
#pragma unroll
        for (int i = 0; i < NI1; ++i)
            if (pc1[i] >= 0 && !(p.dbg & 16)) lds_dma16(xb + rowo[i] + colo + c * 64, lds0 + slot * G::XSLOT + pc1[i] * 1024);
    };
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.wimg);
    auto issue_w = [&](int unit0, int nunits, unsigned dst) {   // units unit0 .. of the stream -> dst; 4 pieces per unit, dealt over the waves
        for (int i = wave; i < nunits * 4; i += NW) lds_dma16(wb + (size_t)unit0 * L1_UNIT + i * 1024 + lane * 16, lds0 + dst + i * 1024);
    };

    // bn1's scale / shift of this lane's 16 channels (16 g + 4 j + e), requested now: their round trip is long over when stage 1 ends
    float sc1[16], sh1[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4 *>(sc1 + 4 * q) = *reinterpret_cast<const f32x4 *>(p.scale1 + 16 * g + 4 * q);
        *reinterpret_cast<f32x4 *>(sh1 + 4 * q) = *reinterpret_cast<const f32x4 *>(p.shift1 + 16 * g + 4 * q);
    }
    if (tid < 64) *reinterpret_cast<uint4 *>(dsm + G::ZOFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);      // visible behind the first chunk's barrier
    long long st_[8];
#define L1_STAMP(k) { if (p.stamps) st_[k] = __builtin_amdgcn_s_memtime(); }
    L1_STAMP(0);
    f32x4 acc[4][NI1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NI1; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ================================ stage 1: conv1 on the halo of all frames ================================
    // NXS slots: the chunks c + 1 .. c + NXS - 1 are in flight while chunk c is multiplied. A wave issues npc pieces per chunk: its wait for chunk c leaves the
    // pieces of the younger chunks in flight.
    int npc = 0;
    for (int i = 0; i < NI1; ++i) npc += (pc1[i] >= 0 && !(p.dbg & 16)) ? 1 : 0;
    if (!(p.dbg & 64)) for (int i = wave; i < kt * 4; i += NW) ++npc;
    auto wait_n = [&](int k) {                                  // s_waitcnt vmcnt(k), k a run-time (wave-uniform) count
        switch (k) {
            case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;   case 3: wait_vmcnt<3>(); break;
            case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;   case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;
            case 8: wait_vmcnt<8>(); break;   case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
            case 12: wait_vmcnt<12>(); break; case 13: wait_vmcnt<13>(); break; case 14: wait_vmcnt<14>(); break; case 15: wait_vmcnt<15>(); break;
            case 16: wait_vmcnt<16>(); break; case 17: wait_vmcnt<17>(); break; case 18: wait_vmcnt<18>(); break;
            default: wait_vmcnt<0>(); break;
        }
    };
    auto issue_chunk = [&](int c) {
        issue_x(c, c % G::NXS);
        if (!(p.dbg & 64)) issue_w(c * kt, kt, RW + (c % G::NXS) * 3 * L1_UNIT);
    };
    const bool fast1 = T == 4 && kt == 3 && !(p.dbg & 32);
    for (int c = 0; c < G::NXS - 1; ++c)
        if (c < nc1) issue_chunk(c);
    for (int c = 0; c < nc1; ++c) {
        wait_n(npc * min(G::NXS - 2, nc1 - 1 - c));            // this wave's pieces of chunk c
        __builtin_amdgcn_s_barrier();                           // ... everybody's; everybody is done with chunk c - 1
        asm volatile("" ::: "memory");
        if (c == 0) L1_STAMP(1);
        if (c + G::NXS - 1 < nc1) issue_chunk(c + G::NXS - 1);  // into the slot of chunk c - 1
        const int xs = (c % G::NXS) * G::XSLOT, ws = RW + (c % G::NXS) * 3 * L1_UNIT;
        if (fast1) {
            // T == 4 frames, 3 taps: straight-line code (every fragment read of a tap is issued before its MFMAs; behind a branch per tile the LDS latency of each
            // read was exposed: 3.5 k cycles per chunk for 1.6 k of MFMA work)
            // The tiles i < 4 of a wave are the SAME halo row of frames 0..3: their pixel fragments are read once per chunk (4 reads) and serve all three taps
            // (tile i, tap dt multiplies frame i + dt - 1): 19 fragment reads per chunk and wave instead of 27 -- the CU's LDS read port, not the matrix pipe, bounded
            // this loop.
            uint4 bf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = L1_LDS16(xs + (f * HH + hr1[0]) * 1024 + lane * 16);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                uint4 a[4], bx[NI1 - 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = L1_LDS16(ws + dt * L1_UNIT + j * 1024 + lane * 16);
#pragma unroll
                for (int i = 4; i < NI1; ++i) {
                    const int f = t1[i] + dt - 1;
                    bx[i - 4] = L1_LDS16(((unsigned)f < 4u ? xs + (f * HH + hr1[i]) * 1024 : G::ZOFF) + lane * 16);
                }
                if (p.dbg & 128) continue;
#pragma unroll
                for (int i = 0; i < NI1; ++i)
                    if (i >= 4 || (i + dt - 1 >= 0 && i + dt - 1 <= 3)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][i] = T_::mfma16(a[j], i < 4 ? bf[i + dt - 1 < 0 ? 0 : i + dt - 1 > 3 ? 3 : i + dt - 1] : bx[i - 4], acc[j][i]);
                    }
            }
        } else {
            for (int dt = 0; dt < kt; ++dt) {
                uint4 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = L1_LDS16(ws + dt * L1_UNIT + j * 1024 + lane * 16);
                if (p.dbg & 128) continue;                              // no pixel fragment reads, no MFMAs
#pragma unroll
                for (int i = 0; i < NI1; ++i) {
                    const int f = t1[i] + dt - pt;                  // input frame of this tap (wave-uniform)
                    if ((unsigned)f < (unsigned)T && !(p.dbg & 32)) {
                        const uint4 b = L1_LDS16(xs + (f * HH + hr1[i]) * 1024 + lane * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][i] = T_::mfma16(a[j], b, acc[j][i]);
                    }
                }
            }
        }
    }
    L1_STAMP(2);
    __builtin_amdgcn_s_barrier();                               // every wave is done with the x slots and the stage-1 weight slots
    asm volatile("" ::: "memory");
    const int u2 = (p.cin >> 5) * kt;                           // first stage-2 unit
    issue_w(u2, 4, RW);                                         // taps 0, 1 -> ring slots 0, 1
    constexpr int NP2 = 8 / NW;                                 // pieces per wave and tap

    // stages 2 and 3: rows r = frame * NPX + tile pixel (flat), M tiles of 16 rows dealt over the waves: base + (wave < rem) each, consecutive
    const int NR = T * NPX, nt2 = (NR + 15) >> 4;
    const int cnt2 = nt2 / NW + (wave < nt2 % NW ? 1 : 0), tl0 = wave * (nt2 / NW) + min(wave, nt2 % NW);
    // byte offset of row (tile i, 4 g + e) in x (and y, ldx == ldy) at channel 4 l15; ~0u: outside the frame / past the clip. The residual rows of a stage-3 pass are
    // requested one pass ahead of their use, those of pass 0 here, a whole stage 2 ahead.
    // Stage 3's M tiles of this wave. Plain: the stage-2 tiles. POOL (MaxPool3d((2,1,1)) over frame pairs fused, large_i3d.py:139): wave = (pair q = wave >> 2, s = wave & 3)
    // takes tiles 2 s, 2 s + 1 of the frame's 7 (s = 3: tile 6) in BOTH frames 2 q, 2 q + 1 -- slots 0, 1 the even frame, 2, 3 the odd one: the maximum over the
    // pair is taken between accumulator registers of one lane (a first version pooled across waves through LDS: two more barriers per pass, 1 524 vs 1 181 us)
    int tile3[NI2];
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
        if (POOL) {
            const int q = wave >> 2, k = 2 * (wave & 3) + (i & 1), f = 2 * q + (i >> 1);
            tile3[i] = (f < 2 * (T / 2) && k < NPX / 16) ? f * (NPX / 16) + k : -1;
        } else {
            tile3[i] = i < cnt2 ? tl0 + i : -1;
        }
    }
    static_assert(!POOL || (NPX % 16 == 0 && NI2 == 4 && NW == 8), "the pooled wave mapping: whole tiles per frame, 2 + 2 tiles per wave");
    unsigned go[NI2][4];
#pragma unroll
    for (int i = 0; i < NI2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 16 * max(tile3[i], 0) + 4 * g + e;
            const int t = r / NPX, j2 = r - t * NPX;
            const int oh = j2 / L1_TW, ow = j2 - oh * L1_TW;
            const bool in = tile3[i] >= 0 && r < NR && oh0 + oh < p.H && ow0 + ow < p.W;
            go[i][e] = in ? ((unsigned)((t * p.H + oh0 + oh) * p.W + ow0 + ow) * (unsigned)p.ldx + 4u * (unsigned)l15) * 2u : ~0u;
        }
    const unsigned char *zsrc = reinterpret_cast<const unsigned char *>(&g_l1_zero) + (lane & 1) * 8;
    unsigned char *sink = reinterpret_cast<unsigned char *>(g_l1_sink) + lane * 8;
    uint2 res[2][NI2][4];
    auto load_res = [&](int cb) {
#pragma unroll
        for (int i = 0; i < NI2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // no branch per access (rows outside the frame read a zero line, their results go to a sink line): behind a conditional block hipcc waits with
                // vmcnt(0) before every access -- sixteen serialised round trips per pass, 8.5 k cycles
                const unsigned char *src = go[i][e] != ~0u ? xb + go[i][e] + cb * 128 : zsrc;
                if (!(p.dbg & 8)) res[cb & 1][i][e] = *reinterpret_cast<const uint2 *>(src);
            }
    };
    load_res(0);

    // ---- relu(bn1(.)) -> M1; lane: halo column l15 of the tile's row, channels 16 g + 4 j + e (the host orders the weight rows that way) ----------------
    {
        const float (&sc)[16] = sc1, (&sh)[16] = sh1;
        const bool colin = (unsigned)(ow0 - 1 + l15) < (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
            if (pc1[i] >= 0) {
                const float hi = colin && (unsigned)(oh0 - 1 + hr1[i]) < (unsigned)p.H ? HI : 0.f;      // outside the frame: zeros (conv2's padding)
                unsigned char *row = dsm + ((t1[i] * HH + hr1[i]) * L1_HW + l15) * L1_ROWB + g * 32;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    unsigned d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = 2 * hf + (k >> 1), e = 2 * (k & 1);
                        d[k] = l1_pair<T_>(acc[j][i][e], acc[j][i][e + 1], sc[4 * j + e], sc[4 * j + e + 1], sh[4 * j + e], sh[4 * j + e + 1], hi);
                    }
                    *reinterpret_cast<uint4 *>(row + hf * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                }
            }
        }
    }

    // ================================ stage 2: conv2 from M1 ================================
    f32x4 acc2[4][NI2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NI2; ++i) acc2[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int m1a[NI2];                                               // M1 byte address of the lane's row (16 (tl0 + i) + l15) at tap (0, 0), chunk g
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
        const int r = min(16 * (tl0 + i) + l15, NR - 1);        // rows past the clip (the last tile's padding): a valid row, results never stored
        const int t = r / NPX, j2 = r - t * NPX;
        const int oh = j2 / L1_TW, ow = j2 - oh * L1_TW;
        m1a[i] = ((t * HH + oh) * L1_HW + ow) * L1_ROWB + g * 16;
    }
    float sc2[16], sh2[16];                                     // bn2, requested a whole stage ahead
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4 *>(sc2 + 4 * q) = *reinterpret_cast<const f32x4 *>(p.scale2 + 16 * g + 4 * q);
        *reinterpret_cast<f32x4 *>(sh2 + 4 * q) = *reinterpret_cast<const f32x4 *>(p.shift2 + 16 * g + 4 * q);
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // M1 complete, taps 0 and 1 landed
    asm volatile("" ::: "memory");
    L1_STAMP(3);
    for (int tap = 0; tap < ((p.dbg & 2) ? 1 : 9); ++tap) {
        if (tap > 0) {
            if (tap < 8) wait_vmcnt<NP2>(); else wait_vmcnt<0>();  // this tap's pieces of this wave (issued two taps ago); the next tap's may fly
            __builtin_amdgcn_s_barrier();                       // ... everybody's; everybody is done with tap - 1
            asm volatile("" ::: "memory");
        }
        if (tap + 2 < 9) issue_w(u2 + 2 * (tap + 2), 2, RW + ((tap + 2) % 3) * 2 * L1_UNIT);
        const int ws = RW + (tap % 3) * 2 * L1_UNIT;
        const int dh = tap / 3, dw = tap - 3 * dh;
        const int toff = (dh * L1_HW + dw) * L1_ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = L1_LDS16(ws + ks * L1_UNIT + j * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < NI2; ++i)
                if (i < cnt2) {
                    const uint4 b = L1_LDS16(m1a[i] + toff + ks * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[j][i] = T_::mfma16(a[j], b, acc2[j][i]);
                }
        }
    }
    L1_STAMP(4);
    __builtin_amdgcn_s_barrier();                               // every wave is done with M1 and the ring
    asm volatile("" ::: "memory");
    const int u3 = u2 + 18;
    constexpr int NU3 = G::RWU >= 8 ? 8 : G::RWU;               // conv3 units resident at first (all 8, or 6: passes 0 .. 2; the last two follow behind pass 1)
    issue_w(u3, NU3, RW);

    // ---- relu(bn2(.)) -> M2[row][64] over M1 -------------------------------------------------------------------------------
    {
        const float (&sc)[16] = sc2, (&sh)[16] = sh2;
#pragma unroll
        for (int i = 0; i < NI2; ++i)
            if (i < cnt2) {
                unsigned char *row = dsm + (16 * (tl0 + i) + l15) * L1_ROWB + g * 32;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    unsigned d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = 2 * hf + (k >> 1), e = 2 * (k & 1);
                        d[k] = l1_pair<T_>(acc2[j][i][e], acc2[j][i][e + 1], sc[4 * j + e], sc[4 * j + e + 1], sh[4 * j + e], sh[4 * j + e + 1], HI);
                    }
                    *reinterpret_cast<uint4 *>(row + hf * 16) = make_uint4(d[0], d[1], d[2], d[3]);
                }
            }
    }

    // ================================ stage 3: conv3 + bn3 + residual + ReLU ================================
    // pixels = A: D[row 4 g + e of the tile][channel column l15]; the host orders the rows of tile jt so that column l15 is channel 4 l15 + jt of the pass's 64:
    // a lane holds 4 consecutive channels of 4 rows
    unsigned char *yb = reinterpret_cast<unsigned char *>(p.y + (size_t)n * (POOL ? T / 2 : T) * p.H * p.W * p.ldy);
    int m2a[NI2];
#pragma unroll
    for (int i = 0; i < NI2; ++i) m2a[i] = (16 * max(tile3[i], 0) + l15) * L1_ROWB + g * 16;
    const float lo = p.relu ? 0.f : -HI;
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // M2 complete, conv3 image landed
    asm volatile("" ::: "memory");
    L1_STAMP(5);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (p.dbg & 4) break;
        int wslot = cb * 2;                                     // unit of (cb, ks = 0) inside RW
        if (G::RWU < 8) {
            if (cb == 2) {                                      // units 6, 7 (pass 3) replace units 0, 1: every wave is done with passes 0, 1
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_w(u3 + 6, 2, RW);
            }
            if (cb == 3) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                wslot = 0;
            }
        }
        // (bn3 first: vector memory retires in order, a wait for these two leaves the younger residual loads of the next pass in flight)
        const f32x4 s3 = *reinterpret_cast<const f32x4 *>(p.scale3 + 64 * cb + 4 * l15), b3 = *reinterpret_cast<const f32x4 *>(p.shift3 + 64 * cb + 4 * l15);
        asm volatile("" ::: "memory");
        if (cb + 1 < 4) load_res(cb + 1);
        f32x4 acc3[4][NI2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NI2; ++i) acc3[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[j] = L1_LDS16(RW + (wslot + ks) * L1_UNIT + j * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < NI2; ++i)
                if (tile3[i] >= 0) {
                    const uint4 am = L1_LDS16(m2a[i] + ks * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc3[j][i] = T_::mfma16(am, bw[j], acc3[j][i]);
                }
        }
        if (!(p.dbg & 8)) {
            uint2 o[NI2][4];
#pragma unroll
            for (int i = 0; i < NI2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[i][e] = make_uint2(l1_pair_res<T_>(acc3[0][i][e], acc3[1][i][e], s3[0], s3[1], b3[0], b3[1], res[cb & 1][i][e].x, lo),
                                         l1_pair_res<T_>(acc3[2][i][e], acc3[3][i][e], s3[2], s3[3], b3[2], b3[3], res[cb & 1][i][e].y, lo));
            if (!POOL) {
#pragma unroll
                for (int i = 0; i < NI2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<uint2 *>(go[i][e] != ~0u ? yb + go[i][e] + cb * 128 : sink) = o[i][e];
            } else {
                // frame 2 q of x -> frame q of y: the even row's offset minus q frames
                const unsigned back = (unsigned)((wave >> 2) * p.H * p.W) * (unsigned)p.ldy * 2u;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint2 m = make_uint2(T_::pk_max(o[i][e].x, o[i + 2][e].x), T_::pk_max(o[i][e].y, o[i + 2][e].y));
                        *reinterpret_cast<uint2 *>((go[i][e] != ~0u && go[i + 2][e] != ~0u) ? yb + (go[i][e] - back) + cb * 128 : sink) = m;
                    }
            }
        }
    }
    if (p.stamps) {
        L1_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        L1_STAMP(7);
        if (tid == 0) for (int k = 0; k < 8; ++k) p.stamps[(size_t)blockIdx.x * 8 + k] = st_[k];
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_bneck_l1_lds_bytes(int32_t variant) { (void)variant; return L1Geo<8, 8>::LDS; }
extern "C" int32_t tedspad_bneck_l1_units(int32_t cin, int32_t kt) { return (cin / 32) * kt + 18 + 8; }

template <typename T_, int TH, int NW, bool POOL>
static int32_t l1_launch(const BneckL1KP &p, long grid, hipStream_t s) {
    static thread_local int attr_set = 0;
    const void *kfn = (const void *)bneck_l1_kernel<T_, TH, NW, POOL>;
    if (!attr_set) {
        if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, L1Geo<TH, NW>::LDS) != hipSuccess) {
            set_error("tedspad_bneck_l1_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set = 1;
    }
    constexpr int lds = L1Geo<TH, NW>::LDS;
    hipLaunchKernelGGL((bneck_l1_kernel<T_, TH, NW, POOL>), dim3((unsigned)grid), dim3(64 * NW), lds, s, p);
    return check_launch("tedspad_bneck_l1_fwd");
}

// A plain layer1 bottleneck of I3Res50 (large_i3d.py:61-84, no downsample branch: layer1.1, layer1.2) in one launch.
// x: (n, t, h, w, cin) 16-bit channels-last (pixel stride ldx); y: (n, t, h, w, 256), or (n, t / 2, h, w, 256) with pool_t2 (maxpool2, large_i3d.py:139, fused);
// pixel stride ldy == ldx: the block's input is its residual, cin == 256.
// w_img: tedspad_bneck_l1_units(cin, kt) units of 4 KB in the kernel's consumption order, packed by the host (engine.BneckL1.pack):
//   stage 1, unit c * kt + dt  : [j 0..3][lane = 16 kg + i][8] = W1[co = 16 (i >> 2) + 4 j + (i & 3)][dt][ci = 32 c + 8 kg ..]   (conv1, kt x 1 x 1, 64 outputs)
//   stage 2, unit tap * 2 + ks : [j][lane = 16 kg + i][8]      = W2[co = 16 (i >> 2) + 4 j + (i & 3)][tap][ci = 32 ks + 8 kg ..] (conv2, 1 x 3 x 3)
//   stage 3, unit cb * 2 + ks  : [j][lane = 16 kg + i][8]      = W3[co = 64 cb + 4 i + j][ci = 32 ks + 8 kg ..]                    (conv3, 256 outputs)
// scale / shift: the folded BatchNorms (fp32; 64, 64, 256 values). variant bit 0: tiles of 4 x 14 pixels, two 4-wave workgroups per CU (else 8 x 14, one of 8 waves).
extern "C" int32_t tedspad_bneck_l1_fwd(const void *x, int32_t ldx, void *y, int32_t ldy, int32_t n, int32_t t, int32_t h, int32_t w, int32_t cin, int32_t kt,
                                        const void *w_img, const float *scale1, const float *shift1, const float *scale2, const float *shift2,
                                        const float *scale3, const float *shift3, int32_t relu, int32_t pool_t2, int32_t variant, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && w_img && scale1 && shift1 && scale2 && shift2 && scale3 && shift3, "tedspad_bneck_l1_fwd: null pointer");
    TS_REQUIRE(n > 0 && t > 0 && t <= L1_TMAX && h > 0 && w > 0, "tedspad_bneck_l1_fwd: 1 <= t <= 4 frames per clip (all of them live in one workgroup)");
    TS_REQUIRE(cin == 256 && (kt == 1 || kt == 3), "tedspad_bneck_l1_fwd: 256 -> 64 -> 64 -> 256 channels, conv1 1x1x1 or 3x1x1");
    TS_REQUIRE(ldx >= cin && ldy == ldx && ldx % 8 == 0, "tedspad_bneck_l1_fwd: bad strides (ldy must equal ldx)");
    TS_REQUIRE(!pool_t2 || t >= 2, "tedspad_bneck_l1_fwd: the pooled output needs two frames");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_img | (uintptr_t)scale1 | (uintptr_t)shift1 | (uintptr_t)scale2 | (uintptr_t)shift2 | (uintptr_t)scale3 |
                (uintptr_t)shift3) % 16 == 0, "tedspad_bneck_l1_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_bneck_l1_fwd: bad dtype");
    TS_REQUIRE((long)t * h * w * ldx * 2 < (1L << 31), "tedspad_bneck_l1_fwd: a clip must fit 32-bit byte offsets");
    (void)variant;      // (bit 0 selected 4 x 14 tiles with two 4-wave workgroups per CU: measured no faster -- the CU's LDS read bandwidth bounds either form -- and removed)
    const int th = 8;
    BneckL1KP p;
    p.x = (const uint16_t *)x; p.y = (uint16_t *)y; p.wimg = (const uint16_t *)w_img;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
    p.N = n; p.T = t; p.H = h; p.W = w; p.ldx = ldx; p.ldy = ldy; p.cin = cin; p.kt = kt; p.relu = relu;
    p.tiles_h = (h + th - 1) / th; p.tiles_w = (w + L1_TW - 1) / L1_TW;
    { const char *e = getenv("TEDSPAD_L1_ABLATE"); p.dbg = e ? atoi(e) : 0; }
    { const char *e = getenv("TEDSPAD_L1_STAMPS"); p.stamps = e ? (long long *)strtoull(e, nullptr, 0) : nullptr; }
    const long grid = (long)n * p.tiles_h * p.tiles_w;
    TS_REQUIRE(grid < (1L << 31), "tedspad_bneck_l1_fwd: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    const int sel = (dtype == TEDSPAD_F16 ? 0 : 2) + (pool_t2 ? 1 : 0);
    switch (sel) {
        case 0: return l1_launch<F16, 8, 8, false>(p, grid, s);
        case 1: return l1_launch<F16, 8, 8, true>(p, grid, s);
        case 2: return l1_launch<BF16, 8, 8, false>(p, grid, s);
        default: return l1_launch<BF16, 8, 8, true>(p, grid, s);
    }
}
