// Whole bottleneck for gfx950: conv1 (1x1x1 | 3x1x1 on two frames) + bn1 + ReLU -> conv2 (1 x 3 x 3, stride 1) + bn2 + ReLU -> conv3 (1x1x1) + bn3 +
// residual + ReLU of a plain I3Res50 layer3 block (aux_code/models/large_i3d.py:42-84, blocks without `downsample`) in ONE launch. A frame of that
// stage is 14 x 14 pixels: ONE WORKGROUP OWNS A WHOLE FRAME, so the 3 x 3 conv needs no halo exchange, both 256-channel tensors between the
// convolutions live in LDS, and HBM sees the block input (once as the conv1 operand, once as the residual) and the block output only.
//
// Why (profiles/r02_bench_cfg2_kernels_1stream.md, round-2 review): as three launches a layer3 block takes 320-360 us per 225 clips against ~200 us of
// MFMA time at the rate the long-K kernels reach; the 256 -> 1024 conv3 + residual launch alone waits on its operands at 14 % MFMA utilisation, and
// the mid tensors cross HBM twice each.
//
// Structure. M = 196 pixels = 13 tiles of 16 (v_mfma_f32_16x16x32: 6 % padding instead of the 14 % of 32-pixel tiles), N = 256 channels per pass.
// 8 waves: wave = (channel group wc = wave & 3 of 64 channels, pixel half hp = wave >> 2: tiles 0..6 | 7..12); the two waves of a SIMD (w, w + 4)
// are one 7-tile and one 6-tile wave, so every SIMD has the same 52 MFMAs per K step. Every GEMM walks K in steps of 32 (one MFMA k-step):
//   * weights (the MFMA A operand) arrive as ONE linear stream of 16 KB slot images [wc][tile j][lane][8] -- packed on the host in exactly the
//     order the kernel consumes them (tedspad_bneck_frame_fwd), so a slot is two 1 KB LDS-DMA pieces per wave and every A fragment is a
//     lane-linear (conflict-free) ds_read_b128 -- through a 3-slot ring that runs across all three GEMMs without a bubble;
//   * stage 1 (conv1): the pixel operand streams from global memory into fragment-layout slots [tile][lane][8] (6-slot ring living in the region
//     that later holds the mid tensor); K = cin, or 2 * cin for the temporal conv on a two-frame clip in its folded form (engine.TPairConv:
//     frame f's output = Wa . x[f=0] + Wb . x[f=1] with (Wa, Wb) = (W1, W2) | (W0, W1); no products on the zero padding);
//   * relu(bn1(.)) is written to LDS as M[pixel][256] (rows of 512 + 32 bytes: every 16-lane group of a fragment read covers all 64 banks and the K step
//     and tile are immediate offsets), row 196 = zeros;
//   * stage 2 (conv2): the pixel fragments of tap (dh, dw) are read from M at row + (dh-1)*14 + (dw-1); lanes whose tap leaves the frame read the
//     zero row (one v_cndmask per fragment); relu(bn2(.)) replaces M after a barrier;
//   * stage 3 (conv3): four passes of 256 output channels, K = 256 from M, operand roles swapped (pixels = A) so that a lane ends with 4 consecutive channels
//     of 4 pixels and consecutive lanes with consecutive addresses: residual rows (requested two K steps ahead) and results move as fully coalesced
//     8-byte accesses straight from / to the accumulators' lanes.
// One s_barrier per K step; counted vmcnt waits keep the next two weight slots (and five pixel slots) in flight across it.
#include <type_traits>

#include "conv_common.h"

namespace tedspad {
namespace {


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct BneckFrameKP {
    const uint16_t *x;          // block input (n, t, 14, 14, cin), pixel stride ldx: conv1 operand and residual
    uint16_t *y;                // block output, pixel stride ldy
    const uint16_t *w1[2];      // stage-1 slot images by frame parity (the same image twice for a 1x1x1 conv1)
    const uint16_t *w23;        // stage-2 (72 slots) + stage-3 (4 x 8 slots) images
    const float *scale1, *shift1, *scale2, *shift2, *scale3, *shift3;
    int frames;                 // n * t
    int ldx, ldy;
    int S1;                     // K steps of stage 1: cin / 32 (1x1x1) or 2 * cin / 32 (temporal, t == 2)
    int temporal;
    int relu;
};

constexpr int BF_HW = 196, BF_W = 14, BF_H = 14, BF_CM = 256, BF_CIN = 1024;
constexpr int BF_ROWB = BF_CM * 2 + 32;                  // bytes per M row: 34 sixteen-byte slots, so that 16 consecutive rows at one chunk cover 16 different slots mod 16 --
                                                         // conflict-free fragment reads with the K step and the tile as IMMEDIATE offsets (an XOR swizzle cost two VALU per read)
constexpr int BF_ZR = BF_HW;                             // the zero row
constexpr int BF_MBYTES = (BF_HW + 1) * BF_ROWB;         // 107 168
constexpr int BF_SLOT = 16384;
constexpr int BF_NXS = 6, BF_NWS = 3;                    // pixel slots (rows 0..191 of the M region), weight slots
constexpr int BF_WOFF = BF_MBYTES;
constexpr int BF_LDS = BF_WOFF + BF_NWS * BF_SLOT;       // 156 320
constexpr int BF_RES_STEP = 5;                            // K step of a stage-3 pass in whose LOAD phase the residual rows are requested
constexpr int BF_S2 = 9 * (BF_CM / 32), BF_S3 = (BF_CIN / 256) * (BF_CM / 32);

// Two LDS-DMA pieces (2 x 1 KB) in ONE statement: wave-uniform source bases, one 32-bit per-lane byte offset, consecutive LDS destinations. M0 (the LDS base)
// is compiler-reserved and not preserved between statements, so it is written here and not saved (cdna_hip_programming.md, inline asm).
__device__ __forceinline__ void lds_dma16s_x2(const void *base0, const void *base1, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                 :
                 : "v"(voff), "s"(base0), "s"(base1), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void lds_dma16s_x2_nt(const void *base0, const void *base1, unsigned voff, unsigned lds_dst) {      // read-once stream: must not push the weights out of L2
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt"
                 :
                 : "v"(voff), "s"(base0), "s"(base1), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void gstore8f(const void *base, unsigned off, uint2 v) {      // base: wave-uniform; nt: streamed
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 d = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(d), "s"(base) : "memory");
}

#ifdef TEDSPAD_BF_ABLATE
// Diagnostic build only (scripts/bneck_frame_cycles.py, tedspad_debug_set_bf_ablate): 1 no weight DMA, 2 no pixel DMA, 4 no residual loads, 8 no stores,
// 16 no MFMAs, 32 no epilogue arithmetic, 64 no barriers, 128 no LDS fragment reads. Results are garbage; only the time is meaningful.
__device__ int g_bf_ablate;
#define BF_ABL(bit) (ablate & (bit))
#elif defined(TEDSPAD_BF_CT_ABLATE)
#define BF_ABL(bit) ((TEDSPAD_BF_CT_ABLATE) & (bit))      // the same switches decided at compile time: no extra branch in the loops (the runtime form costs 700 cycles per K step)
#else
#define BF_ABL(bit) 0
#endif
#ifdef TEDSPAD_BF_STAGE_STAMPS
// Diagnostic build only (scripts/bneck_frame_cycles.py): s_memtime / s_memrealtime at the stage boundaries of every workgroup and pixel half -- 5 stamps per workgroup,
// no measurable cost: cycles per stage and the shader clock the kernel really runs at.
__device__ long long *g_bf_stage_ts;
#define BF_STAGE_DECL() long long st_[5], sr_[2]; st_[0] = __builtin_amdgcn_s_memtime(); sr_[0] = __builtin_amdgcn_s_memrealtime()
#define BF_STAGE(i) st_[i] = __builtin_amdgcn_s_memtime()
#define BF_STAGE_FLUSH() { sr_[1] = __builtin_amdgcn_s_memrealtime(); if (lane == 0 && (wave & 3) == 0 && g_bf_stage_ts) { long long *o_ = g_bf_stage_ts + ((size_t)blockIdx.x * 2 + hp) * 8; for (int i_ = 0; i_ < 5; ++i_) o_[i_] = st_[i_]; o_[5] = sr_[0]; o_[6] = sr_[1]; } }
#else
#define BF_STAGE_DECL()
#define BF_STAGE(i)
#define BF_STAGE_FLUSH()
#endif
#ifdef TEDSPAD_BF_STAMPS
// Diagnostic build only: per workgroup and pixel half, cycles spent in the LOAD / COMPUTE phases of each stage split into work, counted-wait and barrier
// time. The stamps go to a buffer of their own; no output value depends on them. (They cost a third of the kernel's time: read them as proportions.)
__device__ long long *g_bf_ts;
#define BF_TS_DECL() long long ts_acc[18]; for (int i_ = 0; i_ < 18; ++i_) ts_acc[i_] = 0; long long ts_t = __builtin_amdgcn_s_memtime(); const long long ts_begin = ts_t
#define BF_TS(slot) { const long long n_ = __builtin_amdgcn_s_memtime(); ts_acc[slot] += n_ - ts_t; ts_t = n_; }
#define BF_TS_FLUSH() if (lane == 0 && (wave & 3) == 0 && g_bf_ts) { long long *o_ = g_bf_ts + ((size_t)blockIdx.x * 2 + hp) * 20; for (int i_ = 0; i_ < 18; ++i_) o_[i_] = ts_acc[i_]; o_[18] = ts_begin; o_[19] = __builtin_amdgcn_s_memtime(); }
#else
#define BF_TS_DECL()
#define BF_TS(slot)
#define BF_TS_FLUSH()
#endif
#define BF_LDS16(off) (*reinterpret_cast<const uint4 *>(dsm + (off)))
#ifdef TEDSPAD_BF_ABLATE
#define BF_BARRIER() { if (!BF_ABL(64)) asm volatile("s_barrier" ::: "memory"); }
#else
#define BF_BARRIER() asm volatile("s_barrier" ::: "memory")
#endif

// Wave program (h = pixel half = wave >> 2): every K step is a LOAD phase (issue the DMA of the slots two / five steps ahead, read this step's
// A and B fragments into registers) and a COMPUTE phase (28 | 24 MFMAs from registers), one s_barrier after each; waves 4-7 run one phase
// behind waves 0-3, so that on every SIMD one wave multiplies while its partner loads (the ping-pong of conv_p8.hip). A slot of step s is
// first read by waves 0-3; every wave therefore waits for ITS pieces of step s + 1 at the end of the phase in which waves 0-3 compute step s:
// waves 0-3 at the end of their COMPUTE(s), waves 4-7 at the end of their LOAD(s).
// (A version with the pixel half as a template parameter -- two copies of the body, no `if (hp)` / `if (t < ntl)` branch inside a K step -- was SLOWER: 57 KB of
// code for the two halves against 29 KB, stage 2 went from 1120 to 1500 cycles per step. The loops are instruction-fetch sensitive: keep the code small.)
template <typename T>
__global__ __launch_bounds__(512) void bneck_frame_kernel(const BneckFrameKP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, hp = wave >> 2;
    const int l15 = lane & 15, g = lane >> 4;
    const int ntl = hp ? 6 : 7;                              // pixel tiles of this wave
    const int hb = hp * 112;                                 // its first pixel
    const int frame = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;

#ifdef TEDSPAD_BF_ABLATE
    const int ablate = __builtin_amdgcn_readfirstlane(g_bf_ablate);
#endif
    const int par = p.temporal ? (frame & 1) : 0;
    const uint16_t *xres = p.x + (size_t)frame * BF_HW * p.ldx;                        // this frame: residual (and conv1 operand of the 1x1x1 form)
    const int S1 = p.S1, ST = S1 + BF_S2 + BF_S3;

    if (tid < BF_ROWB / 16) *reinterpret_cast<uint4 *>(dsm + BF_ZR * BF_ROWB + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

    // ---- weight stream: slot image of step wnext -> ring slot wfill; two 1 KB pieces per wave ------------------------------------------------
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(p.w1[par]);
    int wnext = 0;
    unsigned wfill = 0, wread = 0;                           // byte offsets of the slot to fill / to read inside the ring
    const unsigned wl0 = (unsigned)(wave * 2048 + lane * 16);
    auto issue_w = [&]() {                                   // (the host pads the stream by two images: the waits count two pieces per step to the end)
        if (!BF_ABL(1)) lds_dma16s_x2(wsrc, wsrc + 1024, wl0, lds0 + BF_WOFF + wfill + (unsigned)wave * 2048u);
        ++wnext;
        wsrc = wnext == S1 ? reinterpret_cast<const unsigned char *>(p.w23) : wsrc + BF_SLOT;
        wfill = wfill == (BF_NWS - 1) * BF_SLOT ? 0u : wfill + BF_SLOT;
    };
    // ---- pixel stream of stage 1: tiles 2 wave, 2 wave + 1 of the frame, 32 channels per step; rows past the frame re-read the last row
    // (their columns of the product are never stored) ------------------------------------------------------------------------------------------
    const unsigned char *xsrc = reinterpret_cast<const unsigned char *>(p.temporal ? p.x + (size_t)(frame - par) * BF_HW * p.ldx : xres);
    int xnext = 0;
    unsigned xfill = 0, xread = 0;
    const unsigned xo = ((unsigned)min(2 * wave * 16 + l15, BF_HW - 1) * (unsigned)p.ldx + (unsigned)g * 8u) * 2u;      // tile 2 wave; tile 2 wave + 1 is 16 rows further
    const bool x1in = (2 * wave + 1) * 16 + 15 < BF_HW;      // wave-uniform: the second tile lies inside the frame (tile 12 holds 4 pixels, tiles 13-15 none: they re-read the first tile's rows)
    const size_t x1step = x1in ? (size_t)32 * p.ldx : 0;
    auto issue_x = [&]() {
        if (!BF_ABL(2)) lds_dma16s_x2_nt(xsrc, xsrc + x1step, xo, lds0 + xfill + (unsigned)wave * 2048u);
        ++xnext;
        xsrc = xnext == BF_CIN / 32 ? xsrc + ((size_t)BF_HW * p.ldx - (BF_CIN - 32)) * 2 : xsrc + 64;      // the second frame of the clip (folded temporal form)
        xfill = xfill == (BF_NXS - 1) * BF_SLOT ? 0u : xfill + BF_SLOT;
    };
    // prologue in the steady-state order (weight pieces before pixel pieces inside a step): virtual steps -4 .. -1
    issue_x(); issue_x();
    issue_w(); issue_x();
    issue_w(); issue_x();

    f32x4 acc[4][7];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // LOAD reads the A fragments and the first NB pixel fragments BEFORE it issues the DMA pieces: a piece holds the wave for 100-185 cycles at the
    // address unit, which is when the reads return. COMPUTE reads pixel fragment t + NB after the MFMAs of tile t (three tiles = 190 cycles ahead of
    // its use) -- all seven in LOAD would cost 28 registers that stage 3 needs for the residual rows.
    uint4 a[4] = {}, b[7] = {};
    auto read_frags = [&](auto nb, auto baddr) {
        const unsigned wb = BF_WOFF + wread + (unsigned)(wc * 4096 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!BF_ABL(128)) a[j] = BF_LDS16(wb + j * 1024);
        wread = wread == (BF_NWS - 1) * BF_SLOT ? 0u : wread + BF_SLOT;
#pragma unroll
        for (int t = 0; t < decltype(nb)::value; ++t)
            if (t < ntl && !BF_ABL(128)) b[t] = BF_LDS16(baddr(t));
    };
    auto mma = [&](auto nb, auto baddr) {
        constexpr int NB = decltype(nb)::value;
#pragma unroll
        for (int t = 0; t < 7; ++t)
            if (t < ntl) {
                if (!BF_ABL(16)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][t] = T::mfma16(a[j], b[t], acc[j][t]);
                } else {
                    asm volatile("" ::"v"(a[0].x), "v"(a[1].y), "v"(a[2].z), "v"(a[3].w), "v"(b[t].x), "v"(b[t].w));
                }
                if (t + NB < 7 && t + NB < ntl && !BF_ABL(128)) b[t + NB] = BF_LDS16(baddr(t + NB));
            }
    };
#ifndef TEDSPAD_BF_NB12
#define TEDSPAD_BF_NB12 7
#endif
    typedef std::integral_constant<int, TEDSPAD_BF_NB12> NB7;      // stages 1 and 2: pixel fragments read in the LOAD phase
    constexpr int NB3V = 4;                                  // stage 3: pixel fragments read in LOAD (the rest three tiles ahead of their MFMAs in COMPUTE: registers)
    typedef std::integral_constant<int, NB3V> NB4;
    // relu(bn(.)) of the accumulators -> M rows (16 consecutive channels per lane and tile: two 16-byte chunks)
    auto write_mid = [&](const float *scale, const float *shift) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float sc[8], sh[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<f32x4 *>(sc + 4 * q) = *reinterpret_cast<const f32x4 *>(scale + 64 * wc + 16 * g + 8 * hf + 4 * q);
                *reinterpret_cast<f32x4 *>(sh + 4 * q) = *reinterpret_cast<const f32x4 *>(shift + 64 * wc + 16 * g + 8 * hf + 4 * q);
            }
            const int c = 8 * wc + 2 * g + hf;
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                if (t < ntl) {
                    const int px = hb + 16 * t + l15;
                    if (px < BF_HW) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(acc[2 * hf + (i >> 2)][t][i & 3] * sc[i] + sh[i], 0.f);
                        *reinterpret_cast<uint4 *>(dsm + px * BF_ROWB + (c << 4)) = pack8<T>(v);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // end of a LOAD phase: the A fragment reads have returned (the slot is refilled by the DMA issued in the partner's next LOAD phase)
#define BF_END_LOAD() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    wait_vmcnt<6>();                                         // step 0's pieces; younger: x(2), w(1), x(3)
    BF_BARRIER();
    if (hp) BF_BARRIER();                                    // waves 4-7: one phase behind
    BF_TS_DECL();
    BF_STAGE_DECL();

    // ================================ stage 1: conv1, both operands streamed ================================
    for (int s = 0; s < S1; ++s) {
        // ---- LOAD
        const unsigned xb = xread + (unsigned)(hp * 7168 + lane * 16);
        auto xaddr = [&](int t) { return xb + t * 1024; };
        read_frags(NB7{}, xaddr);
        xread = xread == (BF_NXS - 1) * BF_SLOT ? 0u : xread + BF_SLOT;
        issue_w();
        if (s + 4 < S1) issue_x();
        BF_END_LOAD();
        BF_TS(0);
        // the pieces of step s + 1; younger than w(s+1): x(s+3) [if s + 3 < S1], w(s+2), x(s+4) [if s + 4 < S1]
        if (hp) { if (s + 4 < S1) wait_vmcnt<6>(); else if (s + 3 < S1) wait_vmcnt<4>(); else wait_vmcnt<2>(); }
        BF_TS(1);
        BF_BARRIER();
        BF_TS(2);
        // ---- COMPUTE
        mma(NB7{}, xaddr);
        BF_TS(3);
        if (!hp) { if (s + 4 < S1) wait_vmcnt<6>(); else if (s + 3 < S1) wait_vmcnt<4>(); else wait_vmcnt<2>(); }
        BF_TS(4);
        BF_BARRIER();
        BF_TS(5);
    }
    BF_STAGE(1);
    // every wave is done with the pixel slots (waves 4-7 read their last one a phase ago): the conv1 tensor lands there
    write_mid(p.scale1, p.shift1);
    zero_acc();
    BF_BARRIER();
    BF_BARRIER();                                            // (waves 0-3 idle while waves 4-7 write, waves 4-7 idle while waves 0-3 load)

    // ================================ stage 2: conv2 from M ================================
    {
        unsigned vmask[7];                                   // bit dh*3 + dw: the tap of this lane's pixel lies inside the frame
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int px = hb + 16 * t + l15;
            unsigned mk = 0;
            if (px < BF_HW) {
                const int h = px / BF_W, w = px - h * BF_W;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
                        if ((unsigned)(h + dh - 1) < (unsigned)BF_H && (unsigned)(w + dw - 1) < (unsigned)BF_W) mk |= 1u << (dh * 3 + dw);
            }
            vmask[t] = mk;
        }
        const int zra = BF_ZR * BF_ROWB + g * 16;
        for (int tap = 0; tap < 9; ++tap) {
            const int dh = tap / 3, dw = tap - dh * 3;
            const int rr = (hb + l15 + (dh - 1) * BF_W + (dw - 1)) * BF_ROWB + g * 16;      // tile 0's pixel under this tap (tile t: + 16 t rows)
            int tb[7];                                       // this tap's fragment base per tile: the shifted row, or the zero row where the tap leaves the frame
#pragma unroll
            for (int t = 0; t < 7; ++t) tb[t] = ((vmask[t] >> tap) & 1u) ? rr + t * 16 * BF_ROWB : zra;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                auto maddr = [&](int t) { return tb[t] + kc * 64; };
                read_frags(NB7{}, maddr);
                issue_w();
                BF_END_LOAD();
                BF_TS(6);
                if (hp) wait_vmcnt<2>();
                BF_TS(7);
                BF_BARRIER();
                BF_TS(8);
                mma(NB7{}, maddr);
                BF_TS(9);
                if (!hp) wait_vmcnt<2>();
                BF_TS(10);
                BF_BARRIER();
                BF_TS(11);
            }
        }
    }
    BF_STAGE(2);
    write_mid(p.scale2, p.shift2);                           // every wave has read the conv1 tensor: the conv2 tensor replaces it
    zero_acc();
    BF_BARRIER();
    BF_BARRIER();

    BF_STAGE(3);
    // ================================ stage 3: conv3 + residual, four passes of 256 channels ================================
    // Operand roles SWAPPED: pixels are the MFMA A operand, weights the B operand, so D[pixel 4 g + e][channel column l15]: a lane ends with four
    // pixels x one column per channel tile, and the host orders the rows of the stage-3 images so that tile j's column l15 is channel 4 l15 + j of the wave's
    // 64-channel group -- FOUR CONSECUTIVE CHANNELS (8 bytes) per lane and pixel, consecutive lanes consecutive addresses: every residual load and every store
    // of a wave covers four pixel rows x 128 contiguous bytes. (With the stage-1/2 roles a lane owned one pixel x 16 channels: 16 rows per instruction, and the
    // address unit took ~250 cycles to issue each of them -- the ablation that removed them made the kernel 30 % faster.)
    // Fragment address of K step kc, tile t: (hb + l15 + 16 t) rows + chunk 4 kc + g, all immediates on one base. Tile 12 holds 4 pixels: its other lanes read
    // rows 196 .. 207 -- the zero row and the first bytes of the weight ring, valid LDS whose (finite) contents only reach product rows that are never stored.
    const int sb0 = (hb + l15) * BF_ROWB + g * 16;
    // global addresses: wave-uniform frame bases + 32-bit byte offsets; pixel row 4 g + e of tile t, channels 64 wc + 4 l15 .. + 3 of the pass
    const unsigned char *xres_b = reinterpret_cast<const unsigned char *>(xres);
    const unsigned char *y_b = reinterpret_cast<const unsigned char *>(p.y + (size_t)frame * BF_HW * p.ldy);
    const unsigned rlane = ((unsigned)(hb + 4 * g) * (unsigned)p.ldx + (unsigned)(64 * wc + 4 * l15)) * 2u, rrow = 2u * (unsigned)p.ldx;     // ldy == ldx
    for (int n = 0; n < BF_CIN / 256; ++n) {
        uint2 res[7][4];
        f32x4 sc3, sh3;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            auto maddr = [&](int t) { return sb0 + t * 16 * BF_ROWB + kc * 64; };
            read_frags(NB4{}, maddr);
            issue_w();
            BF_END_LOAD();
            BF_TS(12);
            // counted waits of waves 4-7 (24 residual loads at step 5, 2 bn loads at step 7, 24 stores): the slot of step kc + 1 and what this wave issued after it
            if (hp) { if (kc == 6 || (kc == 0 && n > 0)) wait_vmcnt<26>(); else wait_vmcnt<2>(); }
            auto load_bn3 = [&]() {                          // bn3 of the lane's four channels (L2 hits)
                unsigned bo = (unsigned)(64 * wc + 4 * l15) * 4u;
                asm volatile("" : "+v"(bo));                 // (a 32-bit lane offset built here: kept as a 64-bit pointer across the pass it spilled)
                bo += (unsigned)(256 * n) * 4u;
                sc3 = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(p.scale3) + bo);
                sh3 = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(p.shift3) + bo);
            };
            if (kc == 7) load_bn3();                         // a phase ahead of the epilogue
            BF_TS(13);
            BF_BARRIER();
            BF_TS(14);
            // (pixels, weights): the accumulator tile is D[pixel][channel]. At step BF_RES_STEP the residual rows of the pass are requested from inside this MFMA
            // stream, a tile's four rows behind that tile's MFMAs (a burst of 28 loads in the LOAD phase held the wave ~1.1 k cycles at the address unit, with the
            // partner waiting at the barrier): two K steps ahead of their use -- vector memory completes in order, a weight slot issued after them cannot be waited
            // for before they have arrived, so they cannot be requested earlier than the ring is deep.
            {
                unsigned rl = rlane, rr2 = rrow;
                if (kc == BF_RES_STEP) {
                    asm volatile("" : "+v"(rl), "+s"(rr2));  // offsets are built here (hoisted out of the loop they spilled)
                    rl += (unsigned)(256 * n) * 2u;
                }
#pragma unroll
                for (int t = 0; t < 7; ++t)
                    if (t < ntl) {
                        if (!BF_ABL(16)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j][t] = T::mfma16(b[t], a[j], acc[j][t]);
                        }
                        if (t + NB3V < 7 && t + NB3V < ntl && !BF_ABL(128)) b[t + NB3V] = BF_LDS16(maddr(t + NB3V));
                        if (kc == BF_RES_STEP) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (hb + 16 * t + 4 * g + e < BF_HW && !BF_ABL(4)) {
                                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                    const u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(xres_b + (rl + (unsigned)(16 * t + e) * rr2)));
                                    res[t][e] = make_uint2(r.x, r.y);
                                } else {
                                    res[t][e] = make_uint2(0u, 0u);
                                }
                            }
                        }
                    } else if (kc == BF_RES_STEP) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[t][e] = make_uint2(0u, 0u);
                    }
            }
            if (kc == 7) {
                // waves 0-3 run their epilogue one phase later, BESIDE the last compute + epilogue of waves 4-7 (which then idle one phase while waves 0-3 load the
                // next pass's first step): the two epilogues (~5 k cycles each, no MFMA) share one phase instead of stretching two
                if (!hp) BF_BARRIER();
                // epilogue in two sweeps: every result first (the packed 4 channels replace the residual registers), THEN the stores back to back -- hipcc
                // drains vmcnt(0) before the first use of a loaded register that follows an asm store (seen in the ISA: serialised store round trips)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    if (t < ntl && !BF_ABL(32)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned rw[2] = {res[t][e].x, res[t][e].y};
                            unsigned pk[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                float v[2];
                                if constexpr (T::kDtype == TEDSPAD_F16) {
                                    // 3.5 vector instructions per value instead of 5.5 (the epilogue is ~5 k cycles of VALU per pass with no MFMA beside it):
                                    // shift + residual in ONE v_fma_mix_f32 that reads the f16 half directly, one fma, one med3 (ReLU + saturation), and a
                                    // packed round-to-nearest conversion of both values (v_cvt_pk_f16_f32, gfx950)
                                    float t0, t1;
                                    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(rw[q]), "v"(sh3[2 * q]));
                                    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(rw[q]), "v"(sh3[2 * q + 1]));
                                    v[0] = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(acc[2 * q][t][e], sc3[2 * q], t0), 0.f), 65504.f);
                                    v[1] = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(acc[2 * q + 1][t][e], sc3[2 * q + 1], t1), 0.f), 65504.f);
                                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q]) : "v"(v[0]), "v"(v[1]));
                                } else {
#pragma unroll
                                    for (int i = 0; i < 2; ++i) {
                                        const int j = 2 * q + i;
                                        const float r = T::to_f32((uint16_t)(i ? rw[q] >> 16 : rw[q] & 0xffffu));
                                        v[i] = __builtin_fmaxf(acc[j][t][e] * sc3[j] + sh3[j] + r, 0.f);
                                    }
                                    pk[q] = (unsigned)T::from_f32(v[0]) | ((unsigned)T::from_f32(v[1]) << 16);
                                }
                            }
                            res[t][e] = make_uint2(pk[0], pk[1]);
                        }
                    }
                }
                // an ordered fence per tile that consumes its packed results: asm statements keep their order, so every result exists before the first store
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 r0 = {res[t][0].x, res[t][0].y, res[t][1].x, res[t][1].y}, r1 = {res[t][2].x, res[t][2].y, res[t][3].x, res[t][3].y};
                    asm volatile("" : "+v"(r0), "+v"(r1)::"memory");
                    res[t][0] = make_uint2(r0.x, r0.y); res[t][1] = make_uint2(r0.z, r0.w);
                    res[t][2] = make_uint2(r1.x, r1.y); res[t][3] = make_uint2(r1.z, r1.w);
                }
                unsigned yl = rlane, yr = rrow;
                asm volatile("" : "+v"(yl), "+s"(yr));
                yl += (unsigned)(256 * n) * 2u;
#pragma unroll
                for (int t = 0; t < 7; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t < ntl && hb + 16 * t + 4 * g + e < BF_HW && !BF_ABL(8)) gstore8f(y_b, yl + (unsigned)(16 * t + e) * yr, res[t][e]);
                zero_acc();
            }
            BF_TS(15);
            // counted waits of waves 0-3 (28 residual loads, 2 bn loads, 28 stores)
            if (!hp) { if (kc >= 5 || (kc == 0 && n > 0)) wait_vmcnt<30>(); else wait_vmcnt<2>(); }
            BF_TS(16);
            BF_BARRIER();
            if (kc == 7 && hp) BF_BARRIER();                 // (the idle phase of waves 4-7)
            BF_TS(17);
        }
    }
    wait_vmcnt<0>();                                         // (both halves leave stage 3 in the same phase: no closing barrier for waves 0-3)
    BF_STAGE(4);
    BF_STAGE_FLUSH();
    BF_TS_FLUSH();
}


}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_bneck_frame_lds_bytes(void) { return BF_LDS; }

#ifdef TEDSPAD_BF_STAMPS
extern "C" int32_t tedspad_debug_set_bf_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bf_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef TEDSPAD_BF_STAGE_STAMPS
extern "C" int32_t tedspad_debug_set_bf_stage_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bf_stage_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef TEDSPAD_BF_ABLATE
extern "C" int32_t tedspad_debug_set_bf_ablate(int32_t bits) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bf_ablate), &bits, sizeof(bits)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int32_t tedspad_bneck_frame_fwd(const void *x, int32_t ldx, void *y, int32_t ldy, int32_t n, int32_t t, int32_t h, int32_t w, int32_t cin,
                                           int32_t cmid, const void *w1_even, const void *w1_odd, int32_t steps1, const void *w23, const float *scale1,
                                           const float *shift1, const float *scale2, const float *shift2, const float *scale3, const float *shift3,
                                           int32_t relu, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && w1_even && w1_odd && w23 && scale1 && shift1 && scale2 && shift2 && scale3 && shift3, "tedspad_bneck_frame_fwd: null pointer");
    TS_REQUIRE(h == BF_H && w == BF_W && cin == BF_CIN && cmid == BF_CM, "tedspad_bneck_frame_fwd: built for 14 x 14 frames, 1024 -> 256 -> 1024 channels (I3Res50 layer3)");
    TS_REQUIRE(n > 0 && t > 0 && ldx >= cin && ldy == ldx && ldx % 8 == 0, "tedspad_bneck_frame_fwd: bad shape / strides (ldy must equal ldx)");
    TS_REQUIRE(steps1 == cin / 32 || (steps1 == 2 * cin / 32 && t == 2), "tedspad_bneck_frame_fwd: steps1 = cin / 32 (1x1x1 conv1) or 2 * cin / 32 (3x1x1 conv1 on two-frame clips)");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1_even | (uintptr_t)w1_odd | (uintptr_t)w23 | (uintptr_t)scale1 | (uintptr_t)shift1 | (uintptr_t)scale2 |
                (uintptr_t)shift2 | (uintptr_t)scale3 | (uintptr_t)shift3) % 16 == 0, "tedspad_bneck_frame_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_bneck_frame_fwd: bad dtype");
    TS_REQUIRE(relu != 0, "tedspad_bneck_frame_fwd: the block ends with a ReLU (large_i3d.py:84); relu must be non-zero");
    const long M = (long)n * t * h * w;
    TS_REQUIRE(M * (ldx > ldy ? ldx : ldy) < (1L << 31), "tedspad_bneck_frame_fwd: tensor too large for 32-bit offsets; split the batch");
    BneckFrameKP p;
    p.x = (const uint16_t *)x; p.y = (uint16_t *)y; p.w1[0] = (const uint16_t *)w1_even; p.w1[1] = (const uint16_t *)w1_odd; p.w23 = (const uint16_t *)w23;
    p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2; p.scale3 = scale3; p.shift3 = shift3;
    p.frames = n * t; p.ldx = ldx; p.ldy = ldy; p.S1 = steps1; p.temporal = steps1 != cin / 32; p.relu = relu;
    hipStream_t s = (hipStream_t)stream;
    static thread_local int attr_set[2] = {0, 0};
    const void *kfn = dtype == TEDSPAD_F16 ? (const void *)bneck_frame_kernel<F16> : (const void *)bneck_frame_kernel<BF16>;
    if (!attr_set[dtype]) {
        if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_bneck_frame_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[dtype] = 1;
    }
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(bneck_frame_kernel<F16>, dim3(p.frames), dim3(512), BF_LDS, s, p);
    else hipLaunchKernelGGL(bneck_frame_kernel<BF16>, dim3(p.frames), dim3(512), BF_LDS, s, p);
    return check_launch("tedspad_bneck_frame_fwd");
}
