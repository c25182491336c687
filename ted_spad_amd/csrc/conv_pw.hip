// Persistent pointwise convolution for gfx950 (tile_cfg 19): 1x1x1 / stride 1 convs with cin = 64 or 128 -- the conv3
// / downsample layers of the first two bottleneck stages (64 -> 256 on 900 k pixels, 128 -> 512), which write 4x the bytes
// they read and do almost no arithmetic.
//
// Measured on the generic kernel (scripts/conv_probe.py --sweep): every tile shape, at 2..5 workgroups per CU, needs the
// same ~185 us for 64 -> 256 at 907 k pixels = 2.5 TB/s of stores, while a copy kernel moves 4.8 TB/s: a workgroup's
// life there is dispatch -> DMA round trip -> 16 MFMAs -> staging -> stores, all serial, ~6.6 us per 32 KB of output.
// Here a workgroup is PERSISTENT: it owns one 64-channel slice of the output (its [64][cin] weight slice is DMA'd into
// LDS once) and walks 128-pixel tiles; the activations of tile i+1 (LDS-DMA, double buffered) are in flight while
// tile i is multiplied, staged and stored, and the row stores are never waited for inside the loop. Every wave owns 32 pixels x 64 channels end to end (MFMA -> its own fp32 staging rows -> 16-byte row stores):
// one barrier per tile. Same K order, same epilogue arithmetic as conv_igemm (results are bit-identical to it).
#include "conv_common.h"

namespace tedspad {
namespace {

__device__ uint4 g_zero16p;

constexpr int PW_BM = 128, PW_BN = 64, PW_STG_LD = PW_BN + 4;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gstore16(void *dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");   // s_nop 1: a store of > 64 bits reads its data registers late; hipcc pads nothing after an asm statement and may overwrite them
}

// POOLT: the conv output additionally goes through MaxPool3d((2,1,1), stride (2,1,1)) before it is written: the sequence
// of tiles is (frame 2k, frame 2k+1) of the same 128 pixels, the first result stays in registers, the second is max-ed
// with it and stored at the pooled position (tiles never straddle a frame: `hw` pixels per frame, `jt` tiles per frame).
// DUAL: two pointwise convs summed in one launch -- y = act(conv(x, w)*scale + shift + conv(x2, w2)*scale2 + shift2), both with
// cin = 64 (KB = 2: K block 0 comes from x / w, K block 1 from x2 / w2). The first bottleneck of layer1 (conv3 + bn3 and
// the downsample conv + bn of large_i3d.py:61-84) runs like this: the 256-channel downsample tensor is never written or
// re-read (1792 -> 768 bytes per pixel for the two launches it replaces). Each source keeps its own fp32 accumulator and its
// own fp32 BatchNorm scale (applied in registers before the staging), so nothing is rounded to 16 bits in between.
template <typename T, int KB, bool RES, bool POOLT, bool DUAL = false>   // KB = cin / 64; RES: fused residual input
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvKP p, const int tiles_m, const int nworkers, const int hw, const int jt) {
    constexpr int XSUB = PW_BM * BK * 2;        // one [128 px][64] sub-tile
    constexpr int WSUB = PW_BN * BK * 2;        // one [64 co][64] sub-tile
    constexpr int XBUF = KB * XSUB;
    constexpr int RBUF = RES ? PW_BM * PW_BN * 2 : 0;   // the tile's residual rows [128 px][64 co], plain row-major
    constexpr int OFF_X = KB * WSUB, OFF_RES = OFF_X + 2 * XBUF, OFF_STG = OFF_RES + 2 * RBUF;
    constexpr int LDS = OFF_STG + 4 * 32 * PW_STG_LD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % p.tiles_n, worker = lid / p.tiles_n;
    const int n0 = tile_n * PW_BN;

    // ---- DMA roles (as conv_igemm: 8 consecutive lanes fetch the eight 16-byte chunks of one row, swizzled source) ----
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16p);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned wrow = wave * 8 * (BK * 2);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            lds_dma16((DUAL && kb == 1 ? p.w2 + (size_t)(n0 + i * 32 + rsub) * p.Kpad : p.w + (size_t)(n0 + i * 32 + rsub) * p.Kpad + kb * BK) + kc * 8,
                      lds0 + kb * WSUB + i * 32 * (BK * 2) + wrow);
    // sequence index q -> first input row of the tile and the number of valid rows in it
    auto tile_rows = [&](int q, int &mb, int &valid) {
        if (POOLT) {            // q = 2 * (frame pair * jt + j) + f
            const int f = q & 1, u = q >> 1;
            const int j = u % jt, fp = u / jt;               // fp = n * (T/2) + tp
            const int tp = fp % (p.Ti >> 1), nb = fp / (p.Ti >> 1);
            mb = ((nb * p.Ti + 2 * tp + f) * hw) + j * PW_BM;
            valid = min(PW_BM, hw - j * PW_BM);
        } else {
            mb = q * PW_BM;
            valid = min(PW_BM, p.M - mb);
        }
    };
    auto issue_x = [&](int q, int buf) {
        int mb, valid;
        tile_rows(q, mb, valid);
        const int mend = mb + valid;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mb + i * 32 + rsub;
                const uint16_t *src = m >= mend ? zero : (DUAL && kb == 1) ? p.x2 + (size_t)m * p.ldx2 + kc * 8 : p.x + (size_t)m * p.ldx + kb * BK + kc * 8;
                lds_dma16(src, lds0 + OFF_X + buf * XBUF + kb * XSUB + i * 32 * (BK * 2) + wrow);
            }
        if (RES) {   // residual rows ride the same DMA stream (ordinary loads inside this loop would make hipcc drain it)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mb + i * 32 + rsub;
                const int nn = n0 + (lane & 7) * 8;
                const uint16_t *src = (m < mend && nn < p.Cout) ? p.res + (size_t)m * p.ldres + nn : zero;
                lds_dma16(src, lds0 + OFF_RES + buf * RBUF + i * 32 * (PW_BN * 2) + wave * 8 * (PW_BN * 2));
            }
        }
    };

    // ---- row roles of the epilogue: lane -> (row = j*8 + lane/8 of the wave's 32 pixels, 8 channels cc*8..) -----------
    const int cc = lane & 7, rrow = lane >> 3;
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
        if (DUAL) {
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(p.shift2 + n), g1 = *reinterpret_cast<const f32x4 *>(p.shift2 + n + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sc[i] = 1.f; sc[i + 4] = 1.f; sf[i] += g0[i]; sf[i + 4] += g1[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(sc[i]), "+v"(sf[i]));   // hipcc's wait for these loads happens HERE, not in the loop
    // ---- MFMA roles: the wave's 32 pixels x 64 channels (2 accumulator tiles) ---------------------------------------
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    float *stg = reinterpret_cast<float *>(smem + OFF_STG) + wave * 32 * PW_STG_LD;
    f32x4 dsc[DUAL ? 2 : 1][2][4];     // DUAL: per-source BatchNorm scales of the channels this lane accumulates (a*32 + 8g + 4lh + {0..3})
    if (DUAL) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dsc[0][a][g] = *reinterpret_cast<const f32x4 *>(p.scale + n0 + a * 32 + 8 * g + 4 * lh);
                dsc[DUAL ? 1 : 0][a][g] = *reinterpret_cast<const f32x4 *>(p.scale2 + n0 + a * 32 + 8 * g + 4 * lh);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(dsc[0][a][g][i]), "+v"(dsc[DUAL ? 1 : 0][a][g][i]));
    }

    // The row STORES are issued from inline asm, like the LDS-DMA: hipcc cannot count vector-memory operations across asm
    // statements and answers every store in such a loop with `s_waitcnt vmcnt(0)` (seen in the ISA of the first version:
    // four serialised store round trips per tile -- the same 2.5 TB/s as the generic kernel); an ordinary load in the
    // loop gets the same treatment (a full drain right after it), and asm-issued register loads are not an option
    // (hipcc copies their destination registers before the data has arrived) -- hence the residual goes through LDS.
    // The one wait on the DMA is counted by hand; vector-memory operations retire in issue order on gfx9 (one counter
    // for loads and stores; hipcc's own counted waits rely on it).
    // sequence of tiles of this worker: plain: q = worker, worker + nworkers, ...; POOLT: units u = worker, worker +
    // nworkers, ... each expanding to q = 2u (even frame) and 2u + 1 (odd frame)
    const int nseq = POOLT ? 2 * tiles_m : tiles_m;           // tiles_m counts UNITS when POOLT
    int q = POOLT ? 2 * worker : worker;
    auto next_q = [&](int c) { return POOLT ? ((c & 1) ? c + 2 * nworkers - 1 : c + 1) : c + nworkers; };
    if (q < nseq) issue_x(q, 0);
    int buf = 0;
    bool stores_pending = false;            // the previous tile issued exactly 4 store instructions per wave (a full tile)
    float keep[4][8];                        // POOLT: the even frame's rows
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) keep[j][i] = 0.f;
    for (; q < nseq; q = next_q(q)) {
        // this tile's activations were issued BEFORE the previous tile's row stores (if any): those may stay in flight
        if (stores_pending) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // ... of every wave; the other buffer is no longer read by anyone
        asm volatile("" ::: "memory");
        const int nxt = next_q(q);
        if (nxt < nseq) issue_x(nxt, buf ^ 1);
        int mb, valid;
        tile_rows(q, mb, valid);
        f32x16 acc[2], acc2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][r] = 0.f; acc2[a][r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const uint16_t *A = reinterpret_cast<const uint16_t *>(smem + OFF_X + buf * XBUF + kb * XSUB) + (wave * 32 + l31) * BK;
            const uint16_t *W = reinterpret_cast<const uint16_t *>(smem + kb * WSUB) + l31 * BK;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int coff = (((ks << 1) | lh) ^ swz) << 3;
                const uint4 fa = *reinterpret_cast<const uint4 *>(A + coff);
                const uint4 fw0 = *reinterpret_cast<const uint4 *>(W + coff), fw1 = *reinterpret_cast<const uint4 *>(W + 32 * BK + coff);
                if (DUAL && kb == 1) {
                    acc2[0] = T::mfma(fw0, fa, acc2[0]);
                    acc2[1] = T::mfma(fw1, fa, acc2[1]);
                } else {
                    acc[0] = T::mfma(fw0, fa, acc[0]);
                    acc[1] = T::mfma(fw1, fa, acc[1]);
                }
            }
        }
        if (DUAL) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[a][4 * g + i] = acc[a][4 * g + i] * dsc[0][a][g][i] + acc2[a][4 * g + i] * dsc[DUAL ? 1 : 0][a][g][i];
        }
        // wave-private staging: lane (pixel l31) holds channels a*32 + 8g + 4lh + {0..3}
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(stg + l31 * PW_STG_LD + a * 32 + 8 * g + 4 * lh) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool second = POOLT && (q & 1);
        size_t obase = (size_t)mb;                                  // first OUTPUT row of the tile
        if (POOLT) {
            const int u = q >> 1, j = u % jt, fp = u / jt;
            obase = (size_t)fp * hw + (size_t)j * PW_BM;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 32 + j * 8 + rrow;
            if (active && row < valid) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stg + (j * 8 + rrow) * PW_STG_LD + cc * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stg + (j * 8 + rrow) * PW_STG_LD + cc * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = v0[i] * sc[i] + sf[i]; v[i + 4] = v1[i] * sc[i + 4] + sf[i + 4]; }
                if (RES) {
                    float rr[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(smem + OFF_RES + buf * RBUF + row * (PW_BN * 2) + cc * 16), rr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += rr[i];
                }
                if (p.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
                }
                if (POOLT && !second) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) keep[j][i] = v[i];
                } else {
                    if (POOLT) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], keep[j][i]);
                    }
                    gstore16(p.y + (obase + row) * p.ldy + n, __builtin_bit_cast(u32x4, pack8_lim<T>(v, p.sat)));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();    // the wave's staging rows are rewritten by its next tile
        stores_pending = valid == PW_BM && (!POOLT || second);   // all 4 store instructions were issued by every wave
        buf ^= 1;
    }
}

template <typename T, int KB, bool RES, bool POOLT>
int32_t launch_pw2(const ConvKP &p, int tiles_m, int hw, int jt, hipStream_t s) {
    const int per_cu = (KB == 1 && !RES) ? 2 : 1;        // LDS per workgroup: 75 KB (cin 64, no residual) .. 147 KB
    int nworkers = (256 * per_cu + p.tiles_n - 1) / p.tiles_n;
    if (nworkers > tiles_m) nworkers = tiles_m;
    hipLaunchKernelGGL((conv_pw_kernel<T, KB, RES, POOLT>), dim3(p.tiles_n * nworkers), dim3(256), 0, s, p, tiles_m, nworkers, hw, jt);
    return check_launch("tedspad_conv_fwd(pointwise persistent)");
}

template <typename T, int KB>
int32_t launch_pw(const ConvKP &pin, bool pool_t, hipStream_t s) {
    ConvKP p = pin;
    p.tiles_n = (p.Cout + PW_BN - 1) / PW_BN;
    if (p.x2) {
        const int tiles = (p.M + PW_BM - 1) / PW_BM;
        int nworkers = (256 + p.tiles_n - 1) / p.tiles_n;
        if (nworkers > tiles) nworkers = tiles;
        hipLaunchKernelGGL((conv_pw_kernel<T, 2, false, false, true>), dim3(p.tiles_n * nworkers), dim3(256), 0, s, p, tiles, nworkers, 0, 1);
        return check_launch("tedspad_conv_pw_dual_fwd");
    }
    if (pool_t) {
        const int hw = p.Hi * p.Wi, jt = (hw + PW_BM - 1) / PW_BM;
        const int units = (p.M / (p.Ti * hw)) * (p.Ti / 2) * jt;      // (batch) x (frame pairs) x (tiles per frame)
        return p.res ? launch_pw2<T, KB, true, true>(p, units, hw, jt, s) : launch_pw2<T, KB, false, true>(p, units, hw, jt, s);
    }
    const int tiles_m = (p.M + PW_BM - 1) / PW_BM;
    return p.res ? launch_pw2<T, KB, true, false>(p, tiles_m, 0, 1, s) : launch_pw2<T, KB, false, false>(p, tiles_m, 0, 1, s);
}

}  // namespace

int32_t launch_conv_pw(int dtype, const ConvKP &p, hipStream_t s, bool pool_t) {
    if (p.x2 && (p.cin != 64 || p.res || pool_t)) {
        set_error("tedspad_conv_pw_dual_fwd: both convs need cin = 64, no residual");
        return TEDSPAD_EINVAL;
    }
    if (!p.pointwise || (p.cin != 64 && p.cin != 128) || p.Kpad != p.cin || p.mask || p.stats || p.y32 || p.ostrided || p.sigmoid || !p.y) {
        set_error("tedspad_conv_fwd: tile_cfg 19 (persistent pointwise) needs a 1x1x1 stride-1 conv with cin 64 or 128 and a plain epilogue");
        return TEDSPAD_EINVAL;
    }
    if (pool_t && p.Ti < 2) {
        set_error("tedspad_conv_pool_t2_fwd: needs at least two frames");
        return TEDSPAD_EINVAL;
    }
    if (p.x2) return dtype == TEDSPAD_F16 ? launch_pw<F16, 2>(p, false, s) : launch_pw<BF16, 2>(p, false, s);
    if (p.cin == 64) return dtype == TEDSPAD_F16 ? launch_pw<F16, 1>(p, pool_t, s) : launch_pw<BF16, 1>(p, pool_t, s);
    return dtype == TEDSPAD_F16 ? launch_pw<F16, 2>(p, pool_t, s) : launch_pw<BF16, 2>(p, pool_t, s);
}

}  // namespace tedspad
