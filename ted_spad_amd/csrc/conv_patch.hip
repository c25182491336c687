// Patch-halo convolution for gfx950 (tile_cfg 32): stride-1 'same' 1 x kh x kw convs with cin % 64 == 0 and cout <= 128 on WIDE frames --
// the DoubleConv layers of the UNet's outer levels (unet_parts.py:8-25: 64 / 128 channels at 224 x 224 and 112 x 112), where the
// flat-halo tile (conv_flat.hip) needs a halo of 256 + 2 W + 2 pixels and loses its second workgroup per CU.
//
// A workgroup owns a 16 x 16 output patch of one frame x all (<= 128) output channels and walks K chunk-major: for every 64-channel
// chunk of the input the (16 + kh - 1) x (16 + kw - 1) halo (18 x 18 positions x 128 B = 41 KB) is fetched ONCE -- positions
// outside the frame come from the zero page, so the taps need no masks -- and serves all kh x kw taps; the [BN][64] weight tile of
// a (chunk, tap) streams through a ring. Same LDS image as conv_flat.hip (128-byte positions, chunk XOR ((position >> 1) & 7) on
// the DMA source and on the read side), same 4-wave / two-workgroups-per-CU structure; a wave owns four rows of the patch (two
// MFMA pixel groups of two rows) x BN channels. cin = 64 walks K like the generic tiles (bit-identical); cin > 64 walks (chunk, tap):
// fp32 sums re-associated, like tiles 15 / 16 / 28. The epilogue also carries the training extras (batch statistics, fp32 output, ReLU-backward
// mask): the UNet's train-mode forward and its data gradients run on it too.
#include "conv_common.h"
#include "det_gate.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16q;

constexpr int PT_S = 16;                 // patch side

struct PatchGeo {
    int HH, WH, NP, ntaps, nchunks, tiles_h, tiles_w, tiles_n;
    long xelems;             // ... elements of the input tensor (M * ldx)
    int ncc, fstride;        // patch mode with kt > 1 (3 x 3 x 3 convs): channel chunks per temporal tap (nchunks = kt * ncc), elements per input frame
    int R;                   // FLAT: halo positions in front of the tile (ph * W + pw)
    int dbg;                 // timing ablations (wrong results; TEDSPAD_PATCH_ABLATE): 1 = weight stages only for the first tap of a tile, 2 = halo only for the first chunk, 4 = no per-tap barrier
    int T, HW, PXF;          // TEMPORAL: frames of a clip, pixels of a frame, pixels per frame in a tile (256 / T rounded to a power of two)
};

// FLAT: the tile is 256 CONSECUTIVE output pixels (flattened (n,t,h,w) index) instead of a 16 x 16 patch, its halo the contiguous run of
// 256 + (kh-1) W + (kw-1) pixels of conv_flat.hip, taps outside the frame redirected per lane to a zero position (tile_cfg 33): no tile
// quantisation on small frames (28 x 28, 14 x 14), for frames narrow enough that the run fits (W <= 60 for a 3 x 3 kernel).
// MODE 2 (TEMPORAL, tile_cfg 34): kt x 1 x 1 'same' convs, the generalisation of conv_tflat_kernel to cout <= 512: a tile is PXF spatial positions of
// ALL T <= 4 frames of a clip (position = frame * PXF + pixel), the chunk's T x PXF input positions are the halo, a tap moves a whole
// frame (delta = +-PXF positions) and taps that leave the clip are skipped wave-uniformly (T = 2: a third of a 3 x 1 x 1 conv's taps).
// SRC (patch / FLAT with kt = 1): the input is a gathered concatenation (tedspad_conv_extras.nchunk_src): chunk k of a halo position comes from its own
// tensor gs.ptr[k] (pixel stride gs.ld[k]), through a nearest x2 upsample when bit k of gs.up is set -- a slot keeps the PIXEL index of its position at full and
// at half resolution instead of an element offset; the LDS image, the K order and therefore the sums are those of the same conv on the materialised concat buffer.
template <typename T, int BN, int MODE = 0, bool SRC = false>
__global__ __launch_bounds__(256) void conv_patch_kernel(const ConvKP p, const PatchGeo g, const PatchSrc gs) {
    constexpr bool FLAT = MODE == 1, TEMP = MODE == 2;
    static_assert(!SRC || MODE != 2, "gathered sources: patch and flat modes only");
    constexpr int NT = 256;
    constexpr int WS = BN == 64 ? 3 : 2;                  // weight ring slots ([BN][64] 16-bit each)
    constexpr int WSTAGE = BN * BK * 2;
    constexpr int WL = BN / 32;                           // weight DMA instructions per thread and stage
    constexpr int NA = BN / 32;                           // 32-channel fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (b % g.tiles_n) * BN;                  // channel tile (fastest: consecutive workgroups share one halo through L2)
    b /= g.tiles_n;
    const int tile_lin = b;
    const int tw = b % g.tiles_w; b /= g.tiles_w;
    const int th = b % g.tiles_h; b /= g.tiles_h;          // b = n * T + t
    const int ho0 = th * PT_S, wo0 = tw * PT_S;
    const int q0 = tile_lin * 256;                         // FLAT: first output pixel of the tile
    const int S = (g.NP + (FLAT ? 1 : 0)) * 8;                         // FLAT: one more (zero) position
    const int Sr = (S + 63) / 64 * 64;
    const int halo_bytes = Sr * 16;
    unsigned char *wring = dsm + halo_bytes;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16q);

    // ---- halo slots of this thread: slot s -> position s >> 3 = (row, col) of the halo, stored chunk s & 7 -----------------------
    constexpr int NHMAX = 12;                              // 12 x 256 slots = 384 positions (e.g. 19 x 19 for a 4 x 4 kernel)
    int hsrc[NHMAX];                                       // element offset of the slot's source (without the channel chunk), or -1
    int hsrcU[SRC ? NHMAX : 1];                            // SRC: hsrc = pixel index of the position, hsrcU = pixel index in a half-resolution source
    const int swo = ((tid & 7) ^ ((tid >> 4) & 7)) << 3;   // SRC: the slot's swizzled 8-channel piece ((s & 7) ^ ((pos >> 1) & 7) does not depend on i)
    const int NH = (Sr + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NHMAX; ++i) {
        const int s = i * NT + tid;
        const int pos = s >> 3, cs = s & 7;
        const int hr = pos / g.WH, hc = pos - hr * g.WH;
        const int ih = ho0 - p.ph + hr, iw = wo0 - p.pw + hc;
        hsrc[i] = -1;
        if (TEMP) {
            const int f = pos / g.PXF, px = pos - f * g.PXF;
            const int nclip = tile_lin / g.tiles_w, s0 = (tile_lin - nclip * g.tiles_w) * g.PXF;
            if (i < NH && pos < g.NP && s0 + px < g.HW) hsrc[i] = (int)((((long)nclip * g.T + f) * g.HW + s0 + px) * p.ldx) + ((cs ^ ((pos >> 1) & 7)) << 3);
        } else if (FLAT) {
            const int q = q0 - g.R + pos;
            if (SRC) {
                hsrcU[i] = -1;
                if (i < NH && pos < g.NP && (unsigned)q < (unsigned)p.M) {
                    const int fr = q / g.HW, r = q - fr * g.HW, qh = r / p.Wi, qw = r - qh * p.Wi;
                    hsrc[i] = q;
                    hsrcU[i] = (fr * (p.Hi >> 1) + (qh >> 1)) * (p.Wi >> 1) + (qw >> 1);
                }
            } else if (i < NH && pos < g.NP && (unsigned)q < (unsigned)p.M) hsrc[i] = (int)((long)q * p.ldx) + ((cs ^ ((pos >> 1) & 7)) << 3);
        } else if (i < NH && pos < g.NP && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) {
            if (SRC) {
                hsrc[i] = (b * p.Hi + ih) * p.Wi + iw;
                hsrcU[i] = (b * (p.Hi >> 1) + (ih >> 1)) * (p.Wi >> 1) + (iw >> 1);
            } else hsrc[i] = (int)((((long)b * p.Hi + ih) * p.Wi + iw) * p.ldx) + ((cs ^ ((pos >> 1) & 7)) << 3);
        }
    }
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w + (size_t)(n0 + rsub) * p.Kpad + kc * 8;
    auto issue_w = [&](int ch, int tap, int slot) {          // ch: chunk index; patch mode with kt > 1: (dt, channel chunk) -> K offset ((dt * taps + tap) * cin + chunk * 64)
        const unsigned dst = lds0 + halo_bytes + slot * WSTAGE + wave * 8 * (BK * 2);
        const int dtw = TEMP ? 0 : ch / g.ncc;
        const uint16_t *src = wsrc + (dtw * g.ntaps + tap) * p.cin + (ch - dtw * g.ncc) * 64;
#pragma unroll
        for (int j = 0; j < WL; ++j) lds_dma16(src + (size_t)(j * 32) * p.Kpad, dst + j * 32 * (BK * 2));
    };

    // ---- MFMA roles: wave w owns patch rows 4w .. 4w+3; pixel group b = rows 4w+2b, 4w+2b+1 (lane l31: row l31 >> 4, col l31 & 15) ----
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int pbase[2];
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) pbase[bq] = (FLAT || TEMP) ? wave * 64 + bq * 32 + l31 : (4 * wave + 2 * bq + (l31 >> 4)) * g.WH + (l31 & 15);
    unsigned vmask[2] = {0u, 0u};      // FLAT: bit (dh*kw + dw): the tap lies inside the frame
    int tq[2] = {0, 0};                // FLAT with kt > 1: the frame index of the pixel inside its clip
    if (FLAT) {
#pragma unroll
        for (int bq = 0; bq < 2; ++bq) {
            const int q = q0 + pbase[bq];
            if (q < p.M) {
                const int r1 = q / p.Wi, w = q - r1 * p.Wi;
                const int h = r1 % p.Hi;
                tq[bq] = (r1 / p.Hi) % p.Ti;
                for (int dh = 0; dh < p.kh; ++dh)
                    for (int dw = 0; dw < p.kw; ++dw)
                        if ((unsigned)(h + dh - p.ph) < (unsigned)p.Hi && (unsigned)(w + dw - p.pw) < (unsigned)p.Wi) vmask[bq] |= 1u << (dh * p.kw + dw);
            }
        }
    }
    f32x16 acc[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int bq = 0; bq < 2; ++bq)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bq][r] = 0.f;

    bool started = false;
    for (int ch = 0; ch < g.nchunks; ++ch) {
        // patch mode with kt > 1 (the 3 x 3 x 3 convs of InceptionI3d, i3d.py Unit3D): chunk = (temporal tap dt, channel chunk); the halo of
        // input frame t + dt - pt is fetched like any other chunk's, a tap that leaves the clip is skipped by the whole workgroup
        long fshift = 0;
        int chc = ch, dtc = p.pt;
        if (!TEMP && g.ncc != g.nchunks) {
            dtc = ch / g.ncc;
            if (!FLAT && (unsigned)(b % p.Ti + dtc - p.pt) >= (unsigned)p.Ti) continue;      // the patch is one frame: the whole chunk leaves the clip
            fshift = (long)(dtc - p.pt) * g.fstride;
            chc = ch - dtc * g.ncc;
        }
        // FLAT with kt > 1: a tile may span frames and clips, so the temporal tap's validity is per pixel (like the spatial taps')
        bool fv[2] = {true, true};
        if (FLAT && g.ncc != g.nchunks) {
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) fv[bq] = (unsigned)(tq[bq] + dtc - p.pt) < (unsigned)p.Ti;
        }
        if (started) __builtin_amdgcn_s_barrier();         // every wave has read the previous chunk's halo and weight slots
        started = true;
        asm volatile("" ::: "memory");
        issue_w(ch, 0, 0);                                 // issue order w(0), halo, w(1): the counted waits below rely on it
        if (SRC) {
            const uint16_t *sp = gs.ptr[chc] + swo;
            const long sl = gs.ld[chc];
            const bool up = (gs.up >> chc) & 1;
#pragma unroll
            for (int i = 0; i < NHMAX; ++i) {
                if (i * NT + wave * 64 >= Sr) break;       // wave-uniform
                const int pi = up ? hsrcU[i] : hsrc[i];
                lds_dma16(hsrc[i] >= 0 ? sp + pi * sl : zero, lds0 + (i * NT + wave * 64) * 16);
            }
        } else {
#pragma unroll
        for (int i = 0; i < NHMAX; ++i) {
            if (i * NT + wave * 64 >= Sr) break;           // wave-uniform
            const long so = hsrc[i] + fshift;            // FLAT: the shifted run may leave the tensor at either end
            lds_dma16((hsrc[i] >= 0 && so >= 0 && so < g.xelems) ? p.x + so + chc * 64 : zero, lds0 + (i * NT + wave * 64) * 16);
        }
        }
        if (g.ntaps > 1) issue_w(ch, 1, 1);
        int dh = 0, dw = 0;
        const int f_out = TEMP ? (wave * 64) / g.PXF : 0;   // TEMPORAL: the wave's output frame (PXF >= 64: one frame per wave)
        for (int kt = 0; kt < g.ntaps; ++kt) {
            const int fin = f_out + kt - p.pt;                 // TEMPORAL: input frame of this tap
            const bool tap_on = !TEMP || (wave * 64 < g.NP && fin >= 0 && fin < g.T);
            const int delta = TEMP ? (kt - p.pt) * g.PXF : dh * g.WH + dw;                 // FLAT: g.WH = W
            unsigned xoff[2], xswz[2];
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int pos = (!FLAT || (((vmask[bq] >> kt) & 1u) && fv[bq])) ? pbase[bq] + delta : g.NP;
                xoff[bq] = (unsigned)pos * 128u;
                xswz[bq] = (unsigned)(pos >> 1) & 7u;
            }
            if (kt + 1 < g.ntaps) wait_vmcnt<WL>(); else wait_vmcnt<0>();   // stage kt (and, on kt = 0, the halo) landed; stage kt+1 may stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (WS == 3 && kt + 2 < g.ntaps) issue_w(ch, kt + 2, (kt + 2) % WS);
            const uint16_t *Wt = reinterpret_cast<const uint16_t *>(wring + (kt % WS) * WSTAGE) + l31 * BK;
            if (tap_on) {
            // (reading a tap's fragments ahead of its MFMAs -- the whole tap for BN = 64, two k-steps in flight for BN = 128, behind sched_barriers -- was measured on one
            // box against this loop: 2-5 % SLOWER on every shape; two waves per SIMD already cover the LDS latency of the compiler's read -> wait -> MFMA order)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned c = (unsigned)((ks << 1) | lh);
                uint4 fa[2], fw[NA];
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) fa[bq] = *reinterpret_cast<const uint4 *>(dsm + xoff[bq] + ((c ^ xswz[bq]) << 4));
#pragma unroll
                for (int a = 0; a < NA; ++a) fw[a] = *reinterpret_cast<const uint4 *>(Wt + a * 32 * BK + ((c ^ swz) << 3));
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int bq = 0; bq < 2; ++bq) acc[a][bq] = T::mfma(fw[a], fa[bq], acc[a][bq]);
            }
            }
            if (WS == 2 && kt + 1 < g.ntaps) {             // two slots: stage kt+1 can only be issued once every wave has read stage kt-1 ... and kt
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 2 < g.ntaps) issue_w(ch, kt + 2, kt & 1);
            }
            if (++dw == p.kw) { dw = 0; ++dh; }
        }
    }
    __syncthreads();

    // ---- epilogue: BN / 64 passes of [256 px][64 co] fp32 through LDS -> coalesced 16-byte rows -----------------------------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
    const int cc = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < BN / 64; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int ml = wave * 64 + bq * 32 + l31;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x16 &A = acc[pass * 2 + a2][bq];
                    f32x4 v = {A[4 * qd], A[4 * qd + 1], A[4 * qd + 2], A[4 * qd + 3]};
                    *reinterpret_cast<f32x4 *>(stg + ml * STG_LD + a2 * 32 + 8 * qd + 4 * lh) = v;
                }
            }
        __syncthreads();
        const int nch = n0 + pass * 64 + cc * 8;
        float s1[8], s2[8], t1[8], t2[8];       // t*: rows of the NEXT statistics group (grouped batch statistics; a tile straddles at most one boundary)
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; t1[i] = 0.f; t2[i] = 0.f; }
        size_t mfirst;                           // smallest output row of this tile
        if (TEMP) {
            const int nclip = tile_lin / g.tiles_w;
            mfirst = (size_t)nclip * g.T * g.HW + (size_t)(tile_lin - nclip * g.tiles_w) * g.PXF;
        } else mfirst = FLAT ? (size_t)q0 : ((size_t)b * p.Ho + ho0) * p.Wo + wo0;
        const size_t sgrp = p.stats_rows ? mfirst / (size_t)p.stats_rows : 0;
        const size_t smb = p.stats_rows ? (sgrp + 1) * (size_t)p.stats_rows : ~(size_t)0;
        if (nch < p.Cout) {
            float sc[8], sf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = r0 + it * 32;                 // staging row = wave*64 + bq*32 + l31  ->  patch row r >> 4, col r & 15
                const int ho = ho0 + (r >> 4), wo = wo0 + (r & 15);
                size_t m;
                if (TEMP) {
                    const int f = r / g.PXF, px = r - f * g.PXF;
                    const int nclip = tile_lin / g.tiles_w, s0 = (tile_lin - nclip * g.tiles_w) * g.PXF;
                    if (r >= g.NP || s0 + px >= g.HW) continue;
                    m = ((size_t)nclip * g.T + f) * g.HW + s0 + px;
                } else {
                    if (FLAT ? q0 + r >= p.M : (ho >= p.Ho || wo >= p.Wo)) continue;
                    m = FLAT ? (size_t)q0 + r : ((size_t)b * p.Ho + ho) * p.Wo + wo;
                }
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
                if (p.stats) {          // batch statistics of the pre-activation (train-mode BatchNorm), as the generic epilogue
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
                    unpack8<T>(*reinterpret_cast<const uint4 *>(p.res + m * p.ldres + nch), rr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += rr[i];
                }
                if (p.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
                }
                if (p.mask) {           // ReLU backward fused into the data gradient that produces d(input)
                    float mk[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(p.mask + m * p.ldmask + nch), mk);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = mk[i] > 0.f ? v[i] : 0.f;
                }
                if (p.y) *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8_lim<T>(v, p.sat);
                if (p.y32) {
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        }
        if (p.stats) {   // block-level reduction over the 32 row groups, then one atomic per channel
            __syncthreads();
            float *red = stg;   // [2][32][64]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[r0 * 64 + cc * 8 + i] = s1[i];
                red[(32 + r0) * 64 + cc * 8 + i] = s2[i];
            }
            __syncthreads();
            float *so = p.stats + sgrp * 2 * p.stats_ld;
            const bool det = det_enter(pass, BN / 64);       // deterministic mode (det_gate.h): passage `pass` of this workgroup
            if (tid < 64 && n0 + pass * 64 + tid < p.Cout) {
                float sa = 0.f, sb = 0.f;
                for (int r = 0; r < 32; ++r) { sa += red[r * 64 + tid]; sb += red[(32 + r) * 64 + tid]; }
                atomicAdd(so + n0 + pass * 64 + tid, sa);
                atomicAdd(so + p.stats_ld + n0 + pass * 64 + tid, sb);
            }
            if (FLAT && p.stats_rows && smb < (size_t)q0 + 256 && smb < (size_t)p.M) {     // only flat tiles cross samples (workgroup-uniform)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    red[r0 * 64 + cc * 8 + i] = t1[i];
                    red[(32 + r0) * 64 + cc * 8 + i] = t2[i];
                }
                __syncthreads();
                if (tid < 64 && n0 + pass * 64 + tid < p.Cout) {
                    float sa = 0.f, sb = 0.f;
                    for (int r = 0; r < 32; ++r) { sa += red[r * 64 + tid]; sb += red[(32 + r) * 64 + tid]; }
                    atomicAdd(so + 2 * p.stats_ld + n0 + pass * 64 + tid, sa);
                    atomicAdd(so + 3 * p.stats_ld + n0 + pass * 64 + tid, sb);
                }
            }
            det_exit(det, pass, BN / 64);
        }
    }
}

template <typename T, int BN, int MODE = 0, bool SRC = false>
int32_t launch_patch_t(const ConvKP &p, int NTf, int cin, hipStream_t s, const PatchSrc *src = nullptr) {
    constexpr bool FLAT = MODE == 1, TEMP = MODE == 2;
    PatchGeo g;
    g.HH = PT_S + p.kh - 1; g.WH = PT_S + p.kw - 1; g.NP = g.HH * g.WH; g.ntaps = p.kh * p.kw; g.nchunks = cin / 64;
    g.tiles_h = (p.Ho + PT_S - 1) / PT_S; g.tiles_w = (p.Wo + PT_S - 1) / PT_S; g.R = 0; g.tiles_n = (p.Cout + BN - 1) / BN;
    if (FLAT) { g.WH = p.Wi; g.NP = 256 + (p.kh - 1) * p.Wi + (p.kw - 1); g.R = p.ph * p.Wi + p.pw; g.tiles_h = 1; g.tiles_w = 1; }
    static const int dbg_env = getenv("TEDSPAD_PATCH_ABLATE") ? atoi(getenv("TEDSPAD_PATCH_ABLATE")) : 0;
    g.dbg = dbg_env;
    g.ncc = g.nchunks; g.fstride = p.Hi * p.Wi * p.ldx; g.xelems = (long)p.M * p.ldx;
    if (MODE != 2 && p.kt > 1) g.nchunks = p.kt * g.ncc;       // 3 x 3 x 3: (temporal tap, channel chunk)
    g.T = p.Ti; g.HW = p.Hi * p.Wi; g.PXF = p.Ti <= 1 ? 256 : p.Ti == 2 ? 128 : 64;
    if (TEMP) { g.ntaps = p.kt; g.NP = g.T * g.PXF; g.tiles_h = 1; g.tiles_w = (g.HW + g.PXF - 1) / g.PXF; }
    const int S = (g.NP + (FLAT ? 1 : 0)) * 8, Sr = (S + 63) / 64 * 64;
    if ((Sr + 255) / 256 > 12) {
        set_error("tedspad_conv_fwd: patch / flat halo config: halo larger than 384 positions (kernel too large, or frame too wide for the flat form)");
        return TEDSPAD_EINVAL;
    }
    const int main_bytes = Sr * 16 + (BN == 64 ? 3 : 2) * BN * BK * 2;
    const int stage_bytes = 256 * (64 + 4) * 4;
    const int lds = main_bytes > stage_bytes ? main_bytes : stage_bytes;
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_patch_kernel<T, BN, MODE, SRC>;
    PatchSrc gs{};
    if (SRC) gs = *src;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    hipLaunchKernelGGL(kfn, dim3((FLAT ? (p.M + 255) / 256 : TEMP ? (NTf / p.Ti) * g.tiles_w : NTf * g.tiles_h * g.tiles_w) * g.tiles_n), dim3(256), lds, s, p, g, gs);
    return check_launch("tedspad_conv_fwd(patch halo)");
}


// ---- two patches per workgroup (tile_cfg 38) ---------------------------------------------------------------------------------------------------------
// What bounds conv_patch_kernel<., 64> on the wide 3 x 3 layers (measured with TEDSPAD_PATCH_ABLATE on 400 x 112 x 112, cin 320 -> 64: 2353 us; without the
// weight stages 1943, without the halo 2000, without either 1692, without the per-tap barriers the same, without the fragment reads 1498; MFMA alone ~1045 at
// the clock the chip holds): the LDS-DMA fill. A (patch, 64-channel chunk) takes 41.5 KB of halo and 73.7 KB of weights into LDS for 18.9 MFLOP -- two resident
// workgroups ask for ~42 GB/s per CU, the rate this fill path gives -- and two thirds of those bytes are weights that the neighbouring patch streams again.
// Here a workgroup owns TWO 16 x 16 patches (any two consecutive ones of the patch list: they share nothing but the weights) and walks K in HALF chunks of 32
// channels, a 3-tap kernel row per weight stage:
//   * halo: [2 patches][18 x 18 positions][64 B], piece c of position p at (c ^ ((p >> 1) & 3)) -- conflict-free for the 16 x 16 x 32 MFMA's fragment read
//     (16 consecutive positions x the lane's k-group; checked over every base offset) -- 41.5 KB, the same as one patch's 64-channel halo;
//   * weights: ring of 3 stages [3 taps dw][64 co][32 k] = 12 KB; one barrier per stage = per 96 MFMAs (16 per barrier in the one-patch form);
//   * a wave owns rows 4w .. 4w+3 of both patches x 64 channels: 8 pixel groups x 4 channel groups of 16 x 16 x 32 MFMAs, 128 accumulator registers;
//   * per (two patches, 64 channels): 83 KB of halo + 73.7 KB of weights for 37.7 MFLOP -- 32 % fewer bytes through the fill path; 76.5 KB of LDS, two
//     workgroups per CU as before. Measured against tile 32 on one box (scripts/ab_probe.sh): 3-6 % faster on the 112 x 112 layers -- the bytes were not the whole
//     story: with every DMA ablated this kernel still takes 1 332 us where its MFMAs need ~1 045. K is walked (half chunk, dh, dw): fp32 sums re-associated like tiles 15 / 16 / 28 / 32.
// Stride-1 'same' 1 x 3 x 3 convs with cin % 32 == 0 (cout tiles of 64; 16-channel groups beyond cout are neither fetched nor multiplied: the 64 -> 32 -> 32 -> 3
// block at full resolution that ends the unet++ decoder), plain input or a gathered concatenation (cin % 64 == 0); the epilogue is the one-patch kernel's.
constexpr int P2_WH = 18, P2_NP = P2_WH * P2_WH, P2_PSLOTS = P2_NP * 4;            // 1296 16-byte slots per patch
constexpr int P2_HALO = (2 * P2_PSLOTS + 63) / 64 * 64 * 16;                       // 41984: the last wave-instruction's upper half is padding
constexpr int P2_WST = 3 * 64 * 64, P2_NWS = 3;
constexpr int P2_MAIN = P2_HALO + P2_NWS * P2_WST;
constexpr int P2_LDS = P2_MAIN > 256 * 68 * 4 ? P2_MAIN : 256 * 68 * 4;
static_assert(P2_LDS <= 80 * 1024, "two workgroups per CU");
constexpr int P2_NHI = 2 * P2_PSLOTS / 256;                                        // 10 full DMA instructions per thread (+ one more in wave 0)

struct Patch2Geo {
    int tiles_h, tiles_w, tiles_n, npatch, nhc, dbg;
    int HW, NPf;             // FLAT: pixels per frame; halo positions of a tile (512 + 2 W + 2)
};

// FLAT (tile_cfg 39): the two "patches" are the two halves of 512 CONSECUTIVE output pixels (flattened (n, h, w) index), the halo ONE contiguous run of
// 512 + 2 W + 2 positions -- no tile quantisation on the narrow frames (56 x 56, 28 x 28, 14 x 14; W <= 62 so that the run fits the same 40 KB) and the halves
// share their halo as well; a tap that leaves the frame is redirected per lane to a zero position (as tile 33 does). Everything else is the two-patch kernel.
constexpr int P2_ZP = 639;                                                          // FLAT: a position behind every run (never written with data: zero-filled)
template <typename T, bool SRC, bool FLAT = false>
__global__ __launch_bounds__(256) void conv_patch2_kernel(const ConvKP p, const Patch2Geo g, const PatchSrc gs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (b % g.tiles_n) * 64;                   // channel tile fastest: consecutive workgroups share their halos through L2
    b /= g.tiles_n;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16q);
    int pf[2], ph0[2], pw0[2];
    bool pon[2];
    const int q0 = b * 512;                                // FLAT: first output pixel of the tile
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int pi = 2 * b + q;
        pon[q] = FLAT ? q0 + q * 256 < p.M : pi < g.npatch;
        if (!pon[q]) pi = 2 * b;
        const int tw = pi % g.tiles_w, t2 = pi / g.tiles_w;
        pw0[q] = tw * PT_S; ph0[q] = (t2 % g.tiles_h) * PT_S; pf[q] = t2 / g.tiles_h;
    }
    // ---- halo slots of this thread: slot s -> patch s / 1296, position (s % 1296) >> 2, LDS piece s & 3 (FLAT: position s >> 2 of the run) ----------
    int hpos[P2_NHI + 1], hposU[SRC ? P2_NHI + 1 : 1], hc8[P2_NHI + 1];
#pragma unroll
    for (int i = 0; i <= P2_NHI; ++i) {
        const int s = i * 256 + tid;
        hpos[i] = -1;
        if (SRC) hposU[i] = -1;
        if (FLAT) {
            const int pos = s >> 2, qp = q0 - p.Wi - 1 + pos;
            hc8[i] = ((s & 3) ^ ((pos >> 1) & 3)) << 3;
            if (i < P2_NHI && pos < g.NPf && (unsigned)qp < (unsigned)p.M) {
                hpos[i] = qp;
                if (SRC) {
                    const int fr = qp / g.HW, r = qp - fr * g.HW, qh = r / p.Wi, qw = r - qh * p.Wi;
                    hposU[i] = (fr * (p.Hi >> 1) + (qh >> 1)) * (p.Wi >> 1) + (qw >> 1);
                }
            }
        } else {
            const int q = s >= P2_PSLOTS ? 1 : 0, r = s - q * P2_PSLOTS;
            const int pos = r >> 2, hr = pos / P2_WH, hcl = pos - hr * P2_WH;
            const int ih = ph0[q] - 1 + hr, iw = pw0[q] - 1 + hcl;
            hc8[i] = ((r & 3) ^ ((pos >> 1) & 3)) << 3;
            if (s < 2 * P2_PSLOTS && pon[q] && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) {
                hpos[i] = (pf[q] * p.Hi + ih) * p.Wi + iw;
                if (SRC) hposU[i] = (pf[q] * (p.Hi >> 1) + (ih >> 1)) * (p.Wi >> 1) + (iw >> 1);
            }
        }
    }
    auto issue_halo = [&](int hcx) {
        if ((g.dbg & 2) && hcx) return;
        const uint16_t *sp = p.x + hcx * 32;
        long sl = p.ldx;
        bool up = false;
        if (SRC) {
            const int ck = hcx >> 1;
            sp = gs.ptr[ck] + (hcx & 1) * 32; sl = gs.ld[ck]; up = (gs.up >> ck) & 1;
        }
        if (!FLAT && wave == 0) {      // the 41st wave-instruction (slots 2560 .. 2623, the upper 32 are padding) goes FIRST: the counted waits below see the same tail in every wave
            const int pi = SRC && up ? hposU[P2_NHI] : hpos[P2_NHI];
            lds_dma16(hpos[P2_NHI] >= 0 ? sp + pi * sl + hc8[P2_NHI] : zero, lds0 + P2_NHI * 256 * 16);
        }
#pragma unroll
        for (int i = 0; i < P2_NHI; ++i) {
            const int pi = SRC && up ? hposU[i] : hpos[i];
            lds_dma16(hpos[i] >= 0 ? sp + pi * sl + hc8[i] : zero, lds0 + (i * 256 + wave * 64) * 16);
        }
    };
    // ---- weight stage (hc, dh): [dw][co][32 k], piece c of row co at c ^ ((co >> 1) & 3) ----------------------------------------------------------
    int wof[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int s = j * 256 + tid, row = s >> 2, dw = row >> 6, co = row & 63;
        wof[j] = (n0 + co) * p.Kpad + dw * p.cin + (((s & 3) ^ ((co >> 1) & 3)) << 3);
    }
    const int na = (p.Cout - n0 + 15) >> 4 < 4 ? (p.Cout - n0 + 15) >> 4 : 4;      // 16-channel groups this tile really has (cout 32 / 8: the unet++ head block)
    auto issue_w = [&](int st) {           // wave w moves channel group w (16 rows x 3 taps): groups beyond cout are not fetched
        if (((g.dbg & 1) && st) || wave >= na) return;
        const int hcx = st / 3, dh = st - hcx * 3;
        const unsigned dst = lds0 + P2_HALO + (st % P2_NWS) * P2_WST + wave * 1024;
        const uint16_t *src = p.w + dh * 3 * p.cin + hcx * 32;
#pragma unroll
        for (int j = 0; j < 3; ++j) lds_dma16(src + wof[j], dst + j * 4096);
    };
    // ---- MFMA roles --------------------------------------------------------------------------------------------------------------------------------
    const int l15 = lane & 15, kg = lane >> 4;
    const unsigned wrd = (unsigned)(l15 * 64 + ((kg ^ ((l15 >> 1) & 3)) << 4));       // this lane's piece of weight row (16 a + l15) of a stage
    unsigned vmask[2][4];                                  // FLAT: bit (dh * 3 + dw) of [half][row group]: the tap of this lane's pixel lies inside its frame
    if (FLAT) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + q * 256 + (4 * wave + r) * 16 + l15;
                unsigned m = 0u;
                if (qq < p.M) {
                    const int r1 = qq / p.Wi, w = qq - r1 * p.Wi, h = r1 % p.Hi;
#pragma unroll
                    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                        for (int dw = 0; dw < 3; ++dw)
                            if ((unsigned)(h + dh - 1) < (unsigned)p.Hi && (unsigned)(w + dw - 1) < (unsigned)p.Wi) m |= 1u << (dh * 3 + dw);
                }
                vmask[q][r] = m;
            }
    }
    f32x4 acc[2][4][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[q][r][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int S = g.nhc * 3;
    issue_halo(0);
    issue_w(0);
    if (S > 1) issue_w(1);
    for (int st = 0; st < S; ++st) {
        const int hcx = st / 3, dh = st - hcx * 3;
        // stage st (and, on dh = 0, the half chunk's halo, issued after the previous stages' successors) landed; stage st + 1 may stay in flight
        if (dh == 0 || st + 1 >= S || wave >= na) wait_vmcnt<0>(); else wait_vmcnt<3>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 2 < S) issue_w(st + 2);                   // its slot held stage st - 1: every wave is past it
        const unsigned wb = (unsigned)(P2_HALO + (st % P2_NWS) * P2_WST) + wrd;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            uint4 fw[4], fa[2][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) if (a < na) fw[a] = *reinterpret_cast<const uint4 *>(dsm + wb + dw * 4096 + a * 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (FLAT) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int pos = (vmask[q][r] >> (dh * 3 + dw)) & 1u ? q * 256 + (4 * wave + r) * 16 + l15 + dh * p.Wi + dw : P2_ZP;
                        fa[q][r] = *reinterpret_cast<const uint4 *>(dsm + pos * 64 + ((kg ^ ((pos >> 1) & 3)) << 4));
                    }
                } else {
                    const int pos = (4 * wave + r + dh) * P2_WH + dw + l15;
                    const unsigned ao = (unsigned)(pos * 64 + ((kg ^ ((pos >> 1) & 3)) << 4));
#pragma unroll
                    for (int q = 0; q < 2; ++q) fa[q][r] = *reinterpret_cast<const uint4 *>(dsm + q * (P2_PSLOTS * 16) + ao);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 4; ++a) if (a < na) acc[q][r][a] = T::mfma16(fw[a], fa[q][r], acc[q][r][a]);
        }
        if (dh == 2 && hcx + 1 < g.nhc) {                  // the one halo buffer: the next half chunk can only follow once every wave has read this one
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_halo(hcx + 1);
        }
    }
    __syncthreads();

    // ---- epilogue, patch by patch: [256 px][64 co] fp32 through LDS -> coalesced 16-byte rows (as conv_patch_kernel) --------------------------------
    constexpr int STG_LD = 64 + 4;
    float *stg = reinterpret_cast<float *>(dsm);
    const int cc = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q) __syncthreads();
        if (pon[q]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    *reinterpret_cast<f32x4 *>(stg + ((4 * wave + r) * 16 + l15) * STG_LD + 16 * a + 4 * kg) = acc[q][r][a];
        }
        __syncthreads();
        const int nch = n0 + cc * 8;
        float s1[8], s2[8], t1[8], t2[8];       // t*: rows of the NEXT statistics group (FLAT: a half of 256 consecutive rows straddles at most one boundary)
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; t1[i] = 0.f; t2[i] = 0.f; }
        const int ho0 = ph0[q], wo0 = pw0[q];
        const size_t mfirst = FLAT ? (size_t)q0 + q * 256 : ((size_t)pf[q] * p.Ho + ho0) * p.Wo + wo0;       // a patch lies inside one frame: no straddling there
        const size_t sgrp = p.stats_rows ? mfirst / (size_t)p.stats_rows : 0;
        const size_t smb = p.stats_rows ? (sgrp + 1) * (size_t)p.stats_rows : ~(size_t)0;
        if (pon[q] && nch < p.Cout) {
            float sc[8], sf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = p.scale[nch + i]; sf[i] = p.shift[nch + i]; }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = r0 + it * 32;
                const int ho = ho0 + (r >> 4), wo = wo0 + (r & 15);
                if (FLAT ? mfirst + r >= (size_t)p.M : (ho >= p.Ho || wo >= p.Wo)) continue;
                const size_t m = FLAT ? mfirst + r : ((size_t)pf[q] * p.Ho + ho) * p.Wo + wo;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + r * STG_LD + cc * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
                if (p.stats) {
                    if (!FLAT || m < smb) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { t1[i] += v[i]; t2[i] += v[i] * v[i]; }
                    }
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
                if (p.mask) {
                    float mk[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(p.mask + m * p.ldmask + nch), mk);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = mk[i] > 0.f ? v[i] : 0.f;
                }
                if (p.y) *reinterpret_cast<uint4 *>(p.y + m * p.ldy + nch) = pack8_lim<T>(v, p.sat);
                if (p.y32) {
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(p.y32 + m * p.ldy32 + nch + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        }
        if (p.stats) {
            __syncthreads();
            float *red = stg;   // [2][32][64]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[r0 * 64 + cc * 8 + i] = s1[i];
                red[(32 + r0) * 64 + cc * 8 + i] = s2[i];
            }
            __syncthreads();
            float *so = p.stats + sgrp * 2 * p.stats_ld;
            const bool det = det_enter(q, 2);
            if (pon[q] && tid < 64 && n0 + tid < p.Cout) {
                float sa = 0.f, sb = 0.f;
                for (int r = 0; r < 32; ++r) { sa += red[r * 64 + tid]; sb += red[(32 + r) * 64 + tid]; }
                atomicAdd(so + n0 + tid, sa);
                atomicAdd(so + p.stats_ld + n0 + tid, sb);
            }
            if (FLAT && p.stats_rows && smb < mfirst + 256 && smb < (size_t)p.M) {     // only flat tiles cross samples (workgroup-uniform)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    red[r0 * 64 + cc * 8 + i] = t1[i];
                    red[(32 + r0) * 64 + cc * 8 + i] = t2[i];
                }
                __syncthreads();
                if (pon[q] && tid < 64 && n0 + tid < p.Cout) {
                    float sa = 0.f, sb = 0.f;
                    for (int r = 0; r < 32; ++r) { sa += red[r * 64 + tid]; sb += red[(32 + r) * 64 + tid]; }
                    atomicAdd(so + 2 * p.stats_ld + n0 + tid, sa);
                    atomicAdd(so + 3 * p.stats_ld + n0 + tid, sb);
                }
            }
            det_exit(det, q, 2);
        }
    }
}

template <typename T, bool SRC, bool FLAT = false>
int32_t launch_patch2_t(const ConvKP &p, int frames, int cin, hipStream_t s, const PatchSrc *src) {
    Patch2Geo g;
    g.HW = p.Hi * p.Wi; g.NPf = 512 + 2 * p.Wi + 2;
    g.tiles_h = (p.Ho + PT_S - 1) / PT_S; g.tiles_w = (p.Wo + PT_S - 1) / PT_S; g.tiles_n = (p.Cout + 63) / 64;
    g.npatch = frames * g.tiles_h * g.tiles_w; g.nhc = cin / 32;
    static const int dbg_env = getenv("TEDSPAD_PATCH_ABLATE") ? atoi(getenv("TEDSPAD_PATCH_ABLATE")) : 0;
    g.dbg = dbg_env;
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_patch2_kernel<T, SRC, FLAT>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_conv_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    PatchSrc gs{};
    if (SRC) gs = *src;
    hipLaunchKernelGGL(kfn, dim3((FLAT ? (p.M + 511) / 512 : (g.npatch + 1) / 2) * g.tiles_n), dim3(256), P2_LDS, s, p, g, gs);
    return check_launch("tedspad_conv_fwd(two-patch halo)");
}

}  // namespace

int32_t launch_conv_patch(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, int mode, const PatchSrc *src) {
    const bool flat = mode == 1;
    if (src && (mode == 2 || p.kt != 1 || src->n != cin / 64 || cin % 64 != 0)) {
        set_error("tedspad_conv_fwd_ex: gathered sources (nchunk_src) need tile_cfg 32 / 33, kt = 1 and one source per 64-channel chunk");
        return TEDSPAD_EINVAL;
    }
    if (mode == 2) {
        const bool same_t = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi && p.ph == 0 && p.pw == 0 && p.pt < p.kt;
        if (cin % 64 != 0 || p.kh != 1 || p.kw != 1 || p.kt < 2 || p.kt > 3 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same_t || p.Ti > 4 || p.Cout > 512 ||
            p.Kpad != p.kt * cin || p.ostrided || p.sigmoid || (!p.y && !p.y32)) {
            set_error("tedspad_conv_fwd: temporal halo config needs a stride-1 'same' kt x 1 x 1 conv (kt 2..3) with cin %% 64 == 0, cout <= 512, T <= 4");
            return TEDSPAD_EINVAL;
        }
        const int frames = N * p.Ti;
        if (p.Cout <= 64) return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 64, 2>(p, frames, cin, s) : launch_patch_t<BF16, 64, 2>(p, frames, cin, s);
        return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 128, 2>(p, frames, cin, s) : launch_patch_t<BF16, 128, 2>(p, frames, cin, s);
    }
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi && p.pt < p.kt && p.ph < p.kh && p.pw < p.kw;
    const bool kt_ok = p.kt == 1 ? p.pt == 0 : (p.kt <= 3 && (long)p.Ti * p.Hi * p.Wi * p.ldx * N < (1L << 31));
    if (cin % 64 != 0 || !kt_ok || p.st != 1 || p.sh != 1 || p.sw != 1 || !same || p.kh * p.kw < 2 || p.kh * p.kw > 16 || p.Cout > 512 ||
        p.Kpad != p.kt * p.kh * p.kw * cin || p.ostrided || p.sigmoid || (!p.y && !p.y32)) {
        set_error("tedspad_conv_fwd: patch-halo config needs a stride-1 'same' kt x kh x kw conv (kt <= 3) with cin %% 64 == 0, cout <= 512 (mask / stats / fp32 output allowed, no strided output map)");
        return TEDSPAD_EINVAL;
    }
    const int frames = N * p.Ti;
    // cout = 128 k + r with r <= 64 (Inception's 3 x 3 x 3 convs: 64 -> 192, 128 -> 192, 160 -> 320): the last 128-channel tile would multiply 64+ channels of
    // zero weights (a quarter of the whole layer at cout 192). Two launches instead: 128-channel tiles over the first 128 k channels, a 64-channel tile kernel over
    // the rest, every per-channel pointer moved to the rest's first channel. Same sums in the same order: bit-identical.
    static const bool split_ok = getenv("TEDSPAD_PATCH_NO_SPLIT") == nullptr;      // A/B knob
    if (split_ok && !src && p.Cout > 128 && p.Cout % 128 != 0 && p.Cout % 128 <= 64) {
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
        const int32_t rc = launch_conv_patch(dtype, a, N, cin, s, mode, nullptr);
        return rc != TEDSPAD_OK ? rc : launch_conv_patch(dtype, b, N, cin, s, mode, nullptr);
    }
    if (src) {
        const bool f16 = dtype == TEDSPAD_F16;
        if (flat) {
            if (p.Cout <= 64) return f16 ? launch_patch_t<F16, 64, 1, true>(p, frames, cin, s, src) : launch_patch_t<BF16, 64, 1, true>(p, frames, cin, s, src);
            return f16 ? launch_patch_t<F16, 128, 1, true>(p, frames, cin, s, src) : launch_patch_t<BF16, 128, 1, true>(p, frames, cin, s, src);
        }
        if (p.Cout <= 64) return f16 ? launch_patch_t<F16, 64, 0, true>(p, frames, cin, s, src) : launch_patch_t<BF16, 64, 0, true>(p, frames, cin, s, src);
        return f16 ? launch_patch_t<F16, 128, 0, true>(p, frames, cin, s, src) : launch_patch_t<BF16, 128, 0, true>(p, frames, cin, s, src);
    }
    if (flat) {
        if (p.Cout <= 64) return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 64, 1>(p, frames, cin, s) : launch_patch_t<BF16, 64, 1>(p, frames, cin, s);
        return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 128, 1>(p, frames, cin, s) : launch_patch_t<BF16, 128, 1>(p, frames, cin, s);
    }
    if (p.Cout <= 64) return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 64>(p, frames, cin, s) : launch_patch_t<BF16, 64>(p, frames, cin, s);
    return dtype == TEDSPAD_F16 ? launch_patch_t<F16, 128>(p, frames, cin, s) : launch_patch_t<BF16, 128>(p, frames, cin, s);
}


int32_t launch_conv_patch2(int dtype, const ConvKP &p, int N, int cin, hipStream_t s, const PatchSrc *src, int flat) {
    const bool same = p.To == p.Ti && p.Ho == p.Hi && p.Wo == p.Wi;
    if (cin % 32 != 0 || (src && cin % 64 != 0) || p.kt != 1 || p.kh != 3 || p.kw != 3 || p.pt != 0 || p.ph != 1 || p.pw != 1 || p.st != 1 || p.sh != 1 || p.sw != 1 || !same ||
        p.Kpad < 9 * cin || p.ostrided || p.sigmoid || (!p.y && !p.y32) || (src && src->n != cin / 64) || (long)N * p.Ti * p.Hi * p.Wi >= (1L << 31)) {
        set_error("tedspad_conv_fwd: two-patch halo config (tile_cfg 38) needs a stride-1 'same' 1 x 3 x 3 conv with cin %% 32 == 0 (gathered sources: %% 64; mask / stats / fp32 output allowed, no strided output map)");
        return TEDSPAD_EINVAL;
    }
    const int frames = N * p.Ti;
    const bool f16 = dtype == TEDSPAD_F16;
    (void)flat;          // tile_cfg 39 (FLAT = true: two flat tiles per workgroup) is retired: the tuner never picked it (round 5); the kernel's FLAT branches are no longer instantiated
    if (src) return f16 ? launch_patch2_t<F16, true>(p, frames, cin, s, src) : launch_patch2_t<BF16, true>(p, frames, cin, s, src);
    return f16 ? launch_patch2_t<F16, false>(p, frames, cin, s, nullptr) : launch_patch2_t<BF16, false>(p, frames, cin, s, nullptr);
}

}  // namespace tedspad
namespace tedspad {
int32_t det_ctl_patch(int op, int on) {
    if (op == 0) patch3_set_det(on);      // the persistent tile (40) declines batch statistics in deterministic mode: its flush is not gated
    return det_ctl(op, on);
}
}  // namespace tedspad
