// HBM-bound kernels of the training step for gfx950 (channels-last 16-bit activations, fp32
// statistics): train-mode BatchNorm forward/backward around the conv kernels, and the backward
// of max-pool, global average pool, bilinear x2 upsample and the UNet's sigmoid output.
// Reference: the autograd of the torch.nn modules of aux_code/models/{large_i3d,unet_parts}.py under
// `fa_model.train()` / `ft_model.train()` (anonymization_training/train_anonymizer.py:73-75,137-139).
// Every kernel moves 16 bytes per lane (8 channels of one pixel); per-channel reductions are
// register partials -> LDS tree -> one float atomic per channel per workgroup.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "det_gate.h"

namespace tedspad {
namespace {

inline int grid_for(long items) {
    long g = (items + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return g < 1 ? 1 : (int)g;
}

// grid for kernels that set up per-channel terms once per thread (bn_train_apply: 8 divisions, 8 rsqrt and 32 loads for 8 channels): `iters`
// items per thread where the tensor allows it, but never fewer than ~2 workgroups per CU
inline int grid_for_iters(long items, int iters) {
    long g = (items + 256L * iters - 1) / (256L * iters);
    const long full = (items + 255) / 256;
    if (g < 512) g = full < 512 ? full : 512;
    if (g > 256 * 16) g = 256 * 16;
    return g < 1 ? 1 : (int)g;
}

// ---- BatchNorm (train): finalize the batch statistics gathered by the conv epilogue -------------
__global__ void bn_finalize_kernel(const float *stats, int stats_ld, float count, const float *gamma, const float *beta,
                                   float eps, float momentum, float *running_mean, float *running_var, float *scale,
                                   float *shift, float *mean_out, float *invstd_out, int C, const float *conv_bias) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c] / count;
    float var = stats[stats_ld + c] / count - mean * mean;   // biased (normalisation)
    var = var < 0.f ? 0.f : var;
    const float invstd = rsqrtf(var + eps);
    const float s = gamma[c] * invstd;
    // the conv epilogue accumulated statistics of conv(x) + bias; y = (z - mean) * s + beta with z = conv + bias
    scale[c] = s;
    shift[c] = beta[c] - mean * s;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    if (running_mean) {   // nn.BatchNorm: running stats use the UNBIASED variance
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
    (void)conv_bias;
}

// y = act(z * scale + shift (+ res))
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *z, const float *scale, const float *shift, const uint16_t *res,
                                                        uint16_t *y, long pixels, int C8, int ldz, int ldres, int ldy, int relu) {
    const long total = pixels * C8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        const long px = idx / C8;
        float v[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(z + px * ldz + c8 * 8), b = *reinterpret_cast<const f32x4 *>(z + px * ldz + c8 * 8 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * scale[c8 * 8 + i] + shift[c8 * 8 + i];
        if (res) {
            float r[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(res + px * ldres + c8 * 8), r);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += r[i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        }
        *reinterpret_cast<uint4 *>(y + px * ldy + c8 * 8) = pack8_lim<T>(v, __builtin_inff());
    }
}

// bn_finalize + bn_apply in ONE launch (train-mode BatchNorm forward after the conv that accumulated the statistics): every thread
// derives scale / shift of its 8 channels from the batch sums itself (2 divisions and an rsqrt per channel: nothing beside the 32 + 16
// bytes it moves per channel), so the separate per-layer finalize launch (159 per phase-2 iteration of cfg3) disappears; workgroup 0
// also writes mean / invstd for the backward pass and updates the running statistics.
// z, the conv output in front of the BatchNorm: fp32, or 16-bit (z16: the dtype of y) -- what the reference's autocast region keeps
// (train_anonymizer.py:78,151: conv outputs are half tensors there); the batch statistics come from the conv's fp32 accumulators either way.
template <typename T>
__device__ __forceinline__ void load_z8(const void *z, bool z16, size_t off, float (&v)[8]) {
    if (z16) {
        unpack8<T>(*reinterpret_cast<const uint4 *>((const uint16_t *)z + off), v);
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4 *>((const float *)z + off), b = *reinterpret_cast<const f32x4 *>((const float *)z + off + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
    }
}

template <typename T, int UF>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const void *z, int z16, const float *stats, int stats_ld, float count, const float *gamma,
                                                              const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                                              float *mean_out, float *invstd_out, int C, const uint16_t *res, uint16_t *y, long pixels,
                                                              int C8, int ldz, int ldres, int ldy, int relu) {
    // statistics GROUPS (gridDim.y): group g = rows [g * pixels, (g + 1) * pixels) with its own sums, mean and invstd -- the three clips
    // of a training step normalised separately in one launch (train_anonymizer.py:169-175 calls ft_model three times)
    const int grp = blockIdx.y;
    if (grp == 0) {     // one channel per thread, spread over the workgroups (a 2048-channel layer in one workgroup was the launch's critical path)
        for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
            for (int g = 0; g < (int)gridDim.y; ++g) {                       // the running statistics take the groups' updates in order
                const float *st = stats + (size_t)g * 2 * stats_ld;
                const float mean = st[c] / count;
                float var = st[stats_ld + c] / count - mean * mean;
                var = var < 0.f ? 0.f : var;
                mean_out[(size_t)g * C8 * 8 + c] = mean;
                invstd_out[(size_t)g * C8 * 8 + c] = rsqrtf(var + eps);
                if (running_mean) {
                    const float unb = count > 1.f ? var * count / (count - 1.f) : var;
                    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
                }
            }
        }
    }
    stats += (size_t)grp * 2 * stats_ld;
    const size_t zbase = (size_t)grp * pixels * ldz;
    y += (size_t)grp * pixels * ldy;
    if (res) res += (size_t)grp * pixels * ldres;
    const long total = pixels * C8;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = stride % C8 == 0;                  // a thread then owns the same 8 channels in every iteration
    float sc[8], sf[8];
    int have = -1;
    // the per-channel terms of chunk c8: 16-byte loads where the chunk is whole and the vectors are aligned (element-wise guarded loads compile to a waited-for round each)
    const bool vec_ok = ((((uintptr_t)stats | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) && (stats_ld & 3) == 0;
    auto terms = [&](int c8) {
        if (vec_ok && c8 * 8 + 8 <= C) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(stats + c8 * 8), a1 = *reinterpret_cast<const f32x4 *>(stats + c8 * 8 + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4 *>(stats + stats_ld + c8 * 8), q1 = *reinterpret_cast<const f32x4 *>(stats + stats_ld + c8 * 8 + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gamma + c8 * 8), g1 = *reinterpret_cast<const f32x4 *>(gamma + c8 * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(beta + c8 * 8), b1 = *reinterpret_cast<const f32x4 *>(beta + c8 * 8 + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float mean = (i < 4 ? a0[i & 3] : a1[i & 3]) / count;
                float var = (i < 4 ? q0[i & 3] : q1[i & 3]) / count - mean * mean;
                var = var < 0.f ? 0.f : var;
                const float s_ = (i < 4 ? g0[i & 3] : g1[i & 3]) * rsqrtf(var + eps);
                sc[i] = s_; sf[i] = (i < 4 ? b0[i & 3] : b1[i & 3]) - mean * s_;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c8 * 8 + i;
                float s_ = 0.f, b_ = 0.f;
                if (c < C) {
                    const float mean = stats[c] / count;
                    float var = stats[stats_ld + c] / count - mean * mean;
                    var = var < 0.f ? 0.f : var;
                    s_ = gamma[c] * rsqrtf(var + eps);
                    b_ = beta[c] - mean * s_;
                }
                sc[i] = s_; sf[i] = b_;
            }
        }
        have = c8;
    };
    // UF items per round, their streaming loads requested before the first is used (a thread walks ~8 items: 8 dependent round trips in the rolled loop; the host
    // keeps UF = 1 on the full-resolution layers, where the unrolled form's registers cost occupancy)
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += UF * stride) {
        int c8s[UF]; long pxs[UF]; bool ok[UF];
        uint4 zq[UF], rq[UF]; f32x4 za[UF], zb[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const long id = idx + u * stride;
            ok[u] = id < total;
            if (u == 0 || !fixed) { c8s[u] = (int)(id % C8); pxs[u] = id / C8; }
            else { c8s[u] = c8s[0]; pxs[u] = pxs[0] + u * (stride / C8); }
            if (ok[u]) {
                const size_t zo = zbase + pxs[u] * ldz + c8s[u] * 8;
                if (z16) zq[u] = *reinterpret_cast<const uint4 *>((const uint16_t *)z + zo);
                else { za[u] = *reinterpret_cast<const f32x4 *>((const float *)z + zo); zb[u] = *reinterpret_cast<const f32x4 *>((const float *)z + zo + 4); }
                if (res) rq[u] = *reinterpret_cast<const uint4 *>(res + pxs[u] * ldres + c8s[u] * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            if (!ok[u]) continue;
            float v[8], r[8];
            if (z16) unpack8<T>(zq[u], v);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = za[u][i]; v[i + 4] = zb[u][i]; }
            }
            if (have != c8s[u]) terms(c8s[u]);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sf[i];
            if (res) {
                unpack8<T>(rq[u], r);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += r[i];
            }
            if (relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
            }
            *reinterpret_cast<uint4 *>(y + pxs[u] * ldy + c8s[u] * 8) = pack8_lim<T>(v, __builtin_inff());
        }
    }
}

// per-channel sums over pixels:  out[0][c] += sum g,  out[1][c] += sum g * xhat
//   g = dy * (y > 0 if relu),  xhat = (z - mean) * invstd  (xhat term skipped when z == nullptr)
// block = 256 threads = (256 / C8L) pixel lanes x C8L channel chunks, C8L = min(C8, 32)
template <typename T, int UF>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint16_t *dy, const uint16_t *y, const void *z, int z16, const float *mean,
                                                             const float *invstd, const float *gamma, const float *beta, float *out, int out_ld, long pixels,
                                                             int C8, int lddy, int ldy, int ldz, int relu) {
    __shared__ float red[2][256][8];
    const int C8L = C8 < 32 ? C8 : 32;
    const int PL = 256 / C8L;
    const int cl = threadIdx.x % C8L, pl = threadIdx.x / C8L;
    const int cgroups = (C8 + C8L - 1) / C8L;
    const int cg = blockIdx.x % cgroups;
    const int pb = blockIdx.x / cgroups, pblocks = gridDim.x / cgroups;
    const int c8 = cg * C8L + cl;
    {   // statistics group (gridDim.y): rows [grp * pixels, (grp + 1) * pixels), its own mean / invstd / sums
        const int grp = blockIdx.y;
        dy += (size_t)grp * pixels * lddy;
        if (y) y += (size_t)grp * pixels * ldy;
        if (z) { mean += (size_t)grp * C8 * 8; invstd += (size_t)grp * C8 * 8; }
        out += (size_t)grp * 2 * out_ld;
    }
    const size_t zbase = (size_t)blockIdx.y * pixels * ldz;
    float s0[8], s1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
    if (c8 < C8 && pl < PL) {
        // y == NULL with relu (units without a residual input): the ReLU mask is recomputed from z with the forward pass's own scale / shift
        // (tedspad_bn_train_apply: s = gamma * invstd, b = beta - mean * s, y = relu(z * s + b)) instead of re-reading the 16-bit output
        const bool remask = relu && !y;
        // the per-channel vectors as 16-byte loads, requested together (element-wise `z ? mean[c] : 0` compiled to 32 masked loads in eight waited-for rounds:
        // 6 us of a 30 us launch on the deep layers)
        float mu[8], is[8], ms[8], mb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mu[i] = is[i] = ms[i] = mb[i] = 0.f;
        if (z) {
            const f32x4 m0 = *reinterpret_cast<const f32x4 *>(mean + c8 * 8), m1 = *reinterpret_cast<const f32x4 *>(mean + c8 * 8 + 4);
            const f32x4 i0 = *reinterpret_cast<const f32x4 *>(invstd + c8 * 8), i1 = *reinterpret_cast<const f32x4 *>(invstd + c8 * 8 + 4);
            f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
            if (remask) {
                g0 = *reinterpret_cast<const f32x4 *>(gamma + c8 * 8); g1 = *reinterpret_cast<const f32x4 *>(gamma + c8 * 8 + 4);
                b0 = *reinterpret_cast<const f32x4 *>(beta + c8 * 8); b1 = *reinterpret_cast<const f32x4 *>(beta + c8 * 8 + 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mu[i] = m0[i]; mu[i + 4] = m1[i]; is[i] = i0[i]; is[i + 4] = i1[i];
                ms[i] = g0[i] * i0[i]; ms[i + 4] = g1[i] * i1[i];
                mb[i] = b0[i] - m0[i] * ms[i]; mb[i + 4] = b1[i] - m1[i] * ms[i + 4];
            }
        }
        // UF pixels per round, every load of the round requested before the first is used: a thread of a deep layer walks ~20 pixels, one dependent round trip
        // each in the rolled loop (UF = 4: -15 ... -30 % on the layers of 28^2 and below; on the full-resolution layers its 184 registers cost occupancy: +5 ... +35 %,
        // the host keeps UF = 1 where a thread walks 48 pixels or more). ZM: 0 no z, 1 z in 16 bits, 2 z in fp32; MM: 0 no mask, 1 mask recomputed from z, 2 mask from y
        const long p0 = (long)pb * PL + pl, pstep = (long)pblocks * PL;
        auto run = [&](auto zm_c, auto mm_c) {
            constexpr int ZM = decltype(zm_c)::value, MM = decltype(mm_c)::value;
            auto body = [&](const uint4 &gq, const uint4 &zq, const f32x4 &za, const f32x4 &zb, const uint4 &yq) {
                float g[8], zz[8], yy[8];
                unpack8<T>(gq, g);
                if (ZM == 1) unpack8<T>(zq, zz);
                if (ZM == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { zz[i] = za[i]; zz[i + 4] = zb[i]; }
                }
                if (MM == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = zz[i] * ms[i] + mb[i] > 0.f ? g[i] : 0.f;
                } else if (MM == 2) {
                    unpack8<T>(yq, yy);
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = yy[i] > 0.f ? g[i] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s0[i] += g[i];
                if (ZM) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) s1[i] += g[i] * (zz[i] - mu[i]) * is[i];
                }
            };
            long px = p0;
            for (; px + (UF - 1) * pstep < pixels; px += UF * pstep) {
                uint4 gq[UF], zq[UF], yq[UF];
                f32x4 za[UF], zb[UF];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const long q = px + u * pstep;
                    gq[u] = *reinterpret_cast<const uint4 *>(dy + q * lddy + c8 * 8);
                    if (ZM == 1) zq[u] = *reinterpret_cast<const uint4 *>((const uint16_t *)z + zbase + q * ldz + c8 * 8);
                    if (ZM == 2) { za[u] = *reinterpret_cast<const f32x4 *>((const float *)z + zbase + q * ldz + c8 * 8); zb[u] = *reinterpret_cast<const f32x4 *>((const float *)z + zbase + q * ldz + c8 * 8 + 4); }
                    if (MM == 2) yq[u] = *reinterpret_cast<const uint4 *>(y + q * ldy + c8 * 8);
                }
#pragma unroll
                for (int u = 0; u < UF; ++u) body(gq[u], zq[u], za[u], zb[u], yq[u]);     // pixels in the rolled loop's order: the sums keep their bits
            }
            for (; px < pixels; px += pstep) {
                uint4 gq, zq = gq, yq = gq;
                f32x4 za = {0.f, 0.f, 0.f, 0.f}, zb = za;
                gq = *reinterpret_cast<const uint4 *>(dy + px * lddy + c8 * 8);
                if (ZM == 1) zq = *reinterpret_cast<const uint4 *>((const uint16_t *)z + zbase + px * ldz + c8 * 8);
                if (ZM == 2) { za = *reinterpret_cast<const f32x4 *>((const float *)z + zbase + px * ldz + c8 * 8); zb = *reinterpret_cast<const f32x4 *>((const float *)z + zbase + px * ldz + c8 * 8 + 4); }
                if (MM == 2) yq = *reinterpret_cast<const uint4 *>(y + px * ldy + c8 * 8);
                body(gq, zq, za, zb, yq);
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        const int zm = !z ? 0 : z16 ? 1 : 2, mm = remask ? 1 : relu ? 2 : 0;      // (remask needs z: the host passes y otherwise)
        if (zm == 0) { if (mm == 2) run(I0{}, I2{}); else run(I0{}, I0{}); }
        else if (zm == 1) { if (mm == 1) run(I1{}, I1{}); else if (mm == 2) run(I1{}, I2{}); else run(I1{}, I0{}); }
        else { if (mm == 1) run(I2{}, I1{}); else if (mm == 2) run(I2{}, I2{}); else run(I2{}, I0{}); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s0[i]; red[1][threadIdx.x][i] = s1[i]; }
    __syncthreads();
    // threads 0 .. C8L*8-1 each own one channel of the group and sum over the pixel lanes
    const bool det = det_enter();                 // deterministic mode (det_gate.h): one workgroup at a time, in blockIdx order
    if (threadIdx.x < C8L * 8) {
        const int ch = threadIdx.x;   // channel within the group: chunk ch/8, element ch%8
        const int c = cg * C8L * 8 + ch;
        if (c < C8 * 8) {
            float a = 0.f, b = 0.f;
            for (int q = 0; q < PL; ++q) { a += red[0][q * C8L + ch / 8][ch % 8]; b += red[1][q * C8L + ch / 8][ch % 8]; }
            atomicAdd(out + c, a);
            if (z) atomicAdd(out + out_ld + c, b);
        }
    }
    det_exit(det);
}

// dz = k[c] * (g - a[c] - xhat * b[c]),  g = dy * (y > 0 if relu);  optionally dres = g
//   train BN: k = gamma*invstd, a = dbeta/M, b = dgamma/M
// dz = gamma*invstd*(g - S0/M - xhat*S1/M) with xhat = (z - mean)*invstd, written as dz = ks*g + A*z + B with three per-channel terms
//   ks = gamma*invstd,  A = -ks*invstd*S1/M,  B = -ks*(S0/M) - A*mean
// (+ the forward's shift mb = beta - mean*ks when the ReLU mask is recomputed from z), set up once per workgroup in LDS ([4][C] floats) and read
// back as 16-byte vectors per element: the version that re-read 6 global arrays x 8 channels per element was TA-bound (3.1 TB/s at 70 VGPRs), one
// that kept the terms in registers needed 88-100 VGPRs and ran at 2.4 TB/s (5 waves / SIMD).
// DB: also gathers d(conv bias) = sum over pixels of dz (zero up to rounding behind a BatchNorm): registers -> LDS -> one atomic per channel
// and workgroup, into row blockIdx.x % db_slots of the accumulator.
template <typename T, bool DB>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint16_t *dy, const uint16_t *y, const void *z, int z16, const float *mean,
                                                            const float *invstd, const float *gamma, const float *beta, const float *sums, int sums_ld,
                                                            float inv_count, uint16_t *dz, uint16_t *dres, float *dbias, int db_slots, long pixels, int C8,
                                                            int lddy, int ldy, int ldz, int lddz, int lddres, int relu) {
    extern __shared__ __attribute__((aligned(16))) float bn_terms[];      // [4][C]: ks, A, B, mb
    const int C = C8 * 8;
    if (DB) dbias += (size_t)(blockIdx.x % db_slots) * C;              // [db_slots][C]: 4096 workgroups adding into ONE row serialise in L2 (~150 us)
    {   // statistics group (gridDim.y)
        const int grp = blockIdx.y;
        dy += (size_t)grp * pixels * lddy;
        if (y) y += (size_t)grp * pixels * ldy;
        mean += (size_t)grp * C; invstd += (size_t)grp * C;
        sums += (size_t)grp * 2 * sums_ld;
        dz += (size_t)grp * pixels * lddz;
        if (dres) dres += (size_t)grp * pixels * lddres;
    }
    const bool remask = relu && !y;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float is = invstd[c], mu = mean[c];
        const float ks = gamma[c] * is;
        const float a = -ks * is * sums[sums_ld + c] * inv_count;
        bn_terms[c] = ks;
        bn_terms[C + c] = a;
        bn_terms[2 * C + c] = -ks * sums[c] * inv_count - a * mu;
        bn_terms[3 * C + c] = remask ? beta[c] - mu * ks : 0.f;        // the forward pass's own shift (its scale is ks; see bn_bwd_reduce_kernel)
    }
    __syncthreads();
    const long total = pixels * C8;
    const size_t zbase = (size_t)blockIdx.y * pixels * ldz;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = stride % C8 == 0;                  // a thread then owns the same 8 channels in every iteration (DB: sums kept in registers)
    float db[DB ? 8 : 1];
    int mine = -1;
    if (DB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) db[i] = 0.f;
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int c8 = (int)(idx % C8);
        const long px = idx / C8;
        float g[8], zz[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(dy + px * lddy + c8 * 8), g);
        load_z8<T>(z, z16, zbase + px * ldz + c8 * 8, zz);
        const float *kt = bn_terms + c8 * 8;
        float ks[8];
        *reinterpret_cast<f32x4 *>(ks) = *reinterpret_cast<const f32x4 *>(kt);
        *reinterpret_cast<f32x4 *>(ks + 4) = *reinterpret_cast<const f32x4 *>(kt + 4);
        if (remask) {
            float mb[8];
            *reinterpret_cast<f32x4 *>(mb) = *reinterpret_cast<const f32x4 *>(kt + 3 * C);
            *reinterpret_cast<f32x4 *>(mb + 4) = *reinterpret_cast<const f32x4 *>(kt + 3 * C + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = zz[i] * ks[i] + mb[i] > 0.f ? g[i] : 0.f;
        } else if (relu) {
            float yy[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(y + px * ldy + c8 * 8), yy);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = yy[i] > 0.f ? g[i] : 0.f;
        }
        if (dres) *reinterpret_cast<uint4 *>(dres + px * lddres + c8 * 8) = pack8_lim<T>(g, __builtin_inff());
        {
            float ka[8], kb[8];
            *reinterpret_cast<f32x4 *>(ka) = *reinterpret_cast<const f32x4 *>(kt + C);
            *reinterpret_cast<f32x4 *>(ka + 4) = *reinterpret_cast<const f32x4 *>(kt + C + 4);
            *reinterpret_cast<f32x4 *>(kb) = *reinterpret_cast<const f32x4 *>(kt + 2 * C);
            *reinterpret_cast<f32x4 *>(kb + 4) = *reinterpret_cast<const f32x4 *>(kt + 2 * C + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = ks[i] * g[i] + (ka[i] * zz[i] + kb[i]);
        }
        *reinterpret_cast<uint4 *>(dz + px * lddz + c8 * 8) = pack8_lim<T>(g, __builtin_inff());
        if (DB) {
            if (fixed) {
                mine = c8;
#pragma unroll
                for (int i = 0; i < 8; ++i) db[i] += g[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) atomicAdd(dbias + c8 * 8 + i, g[i]);
            }
        }
    }
    if (DB) {
        __shared__ float red[256][9];
        if (fixed && 256 % C8 == 0) {                     // thread t owns chunk t % C8 in every workgroup
#pragma unroll
            for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = db[i];
            __syncthreads();
            for (int ch = threadIdx.x; ch < C8 * 8; ch += 256) {      // C8 * 8 can exceed the 256 threads (C = 512 ... 2048): every channel gets its turn
                const int c8 = ch >> 3, i = ch & 7;
                float a = 0.f;
                for (int u = c8; u < 256; u += C8) a += red[u][i];
                atomicAdd(dbias + c8 * 8 + i, a);
            }
        } else if (fixed && mine >= 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(dbias + mine * 8 + i, db[i]);
        }
    }
}

// ---- max-pool backward (gather form): dx[i] = sum over windows o containing i of dy[o] * [argmax(o) == i] (+ add[i]),
//      argmax = the first-maximum index the forward recorded (tedspad_maxpool_fwd_idx): torch's tie rule ----
struct PoolBKP {
    const uint16_t *x, *dy, *add;
    const unsigned char *idx;
    uint16_t *dx;
    int Ti, Hi, Wi, C8, ldx, lddy, ldadd, lddx;
    int To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int relu_mask;
    long total;
};

template <typename T, typename IT = long>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolBKP p) {
    for (long idx0 = (long)blockIdx.x * 256 + threadIdx.x; idx0 < p.total; idx0 += (long)gridDim.x * 256) {
        const IT idx = (IT)idx0;                          // 32-bit element decode where the tensor allows it (see upsample2x_bwd_kernel)
        const int c8 = (int)(idx % (IT)p.C8);
        IT r = idx / (IT)p.C8;
        const int iw = (int)(r % (IT)p.Wi); r /= (IT)p.Wi;
        const int ih = (int)(r % (IT)p.Hi); r /= (IT)p.Hi;
        const int it = (int)(r % (IT)p.Ti);
        const long n = (long)(r / (IT)p.Ti);
        const size_t xi = (((size_t)n * p.Ti + it) * p.Hi + ih) * p.Wi + iw;
        float xv[8], acc[8];
        if (p.relu_mask) unpack8<T>(*reinterpret_cast<const uint4 *>(p.x + xi * p.ldx + c8 * 8), xv);    // only the mask needs the forward input
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        if (p.add) unpack8<T>(*reinterpret_cast<const uint4 *>(p.add + xi * p.ldadd + c8 * 8), acc);
        // output windows o with o*s - p <= i < o*s - p + k
        const int to_lo = max(0, (it + p.pt - p.kt + p.st) / p.st), to_hi = min(p.To - 1, (it + p.pt) / p.st);
        const int ho_lo = max(0, (ih + p.ph - p.kh + p.sh) / p.sh), ho_hi = min(p.Ho - 1, (ih + p.ph) / p.sh);
        const int wo_lo = max(0, (iw + p.pw - p.kw + p.sw) / p.sw), wo_hi = min(p.Wo - 1, (iw + p.pw) / p.sw);
        for (int to = to_lo; to <= to_hi; ++to)
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                    const size_t oi = (((size_t)n * p.To + to) * p.Ho + ho) * p.Wo + wo;
                    // window-local index of this input element in window o
                    const int li = ((it - (to * p.st - p.pt)) * p.kh + (ih - (ho * p.sh - p.ph))) * p.kw + (iw - (wo * p.sw - p.pw));
                    const uint2 pk = *reinterpret_cast<const uint2 *>(p.idx + (oi * p.C8 + c8) * 8);
                    float g[8];
                    unpack8<T>(*reinterpret_cast<const uint4 *>(p.dy + oi * p.lddy + c8 * 8), g);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int a = (int)(((i < 4 ? pk.x : pk.y) >> (8 * (i & 3))) & 255u);
                        acc[i] += (a == li) ? g[i] : 0.f;
                    }
                }
        if (p.relu_mask) {   // x is a ReLU output: fold that ReLU's backward in
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = xv[i] > 0.f ? acc[i] : 0.f;
        }
        *reinterpret_cast<uint4 *>(p.dx + xi * p.lddx + c8 * 8) = pack8_lim<T>(acc, __builtin_inff());
    }
}

// The UNet's pools (unet_parts.py:32: MaxPool2d(2)): 1 x 2 x 2 windows, stride 2, no padding, even frames -- every input element lies in exactly ONE window, at the
// window-local index (ih & 1) * 2 + (iw & 1). The general kernel above spends six runtime-divisor divisions on the window bounds and three on the element decode
// before its first load (2.95 TB/s of its bytes); here the decode is two 32-bit divisions and the loads are unconditional. Same selects, same sums: bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k2s2_kernel(const PoolBKP p) {
    const unsigned C8 = (unsigned)p.C8, Wi = (unsigned)p.Wi, Wo = (unsigned)p.Wo;
    for (long idx0 = (long)blockIdx.x * 256 + threadIdx.x; idx0 < p.total; idx0 += (long)gridDim.x * 256) {
        const unsigned idx = (unsigned)idx0;
        const unsigned pix = idx / C8, c8 = idx - pix * C8;          // pix = (n * Ti + it) * Hi * Wi + ih * Wi + iw
        const unsigned row = pix / Wi, iw = pix - row * Wi;          // row = (n * Ti + it) * Hi + ih; Hi even: row >> 1 = (n * Ti + it) * Ho + ho
        const size_t oi = (size_t)(row >> 1) * Wo + (iw >> 1);
        const int li = (int)((row & 1u) * 2u + (iw & 1u));
        const uint2 pk = *reinterpret_cast<const uint2 *>(p.idx + (oi * C8 + c8) * 8);
        const uint4 gq = *reinterpret_cast<const uint4 *>(p.dy + oi * p.lddy + c8 * 8);
        float xv[8], acc[8], g[8];
        if (p.relu_mask) unpack8<T>(*reinterpret_cast<const uint4 *>(p.x + (size_t)pix * p.ldx + c8 * 8), xv);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        if (p.add) unpack8<T>(*reinterpret_cast<const uint4 *>(p.add + (size_t)pix * p.ldadd + c8 * 8), acc);
        unpack8<T>(gq, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = (int)(((i < 4 ? pk.x : pk.y) >> (8 * (i & 3))) & 255u);
            acc[i] += (a == li) ? g[i] : 0.f;
        }
        if (p.relu_mask) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = xv[i] > 0.f ? acc[i] : 0.f;
        }
        *reinterpret_cast<uint4 *>(p.dx + (size_t)pix * p.lddx + c8 * 8) = pack8_lim<T>(acc, __builtin_inff());
    }
}

// global average pool backward: dx[n,p,c] = dfeat[n,c] / S
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float *dfeat, const uint16_t *mask, int ldmask, uint16_t *dx, int n, int spatial, int C8, int lddx) {
    const long total = (long)n * spatial * C8;
    const float inv = 1.f / (float)spatial;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % C8);
        const long px = idx / C8;
        const long b = px / spatial;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = dfeat[b * C8 * 8 + c8 * 8 + i] * inv;
        if (mask) {
            float mk[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(mask + px * ldmask + c8 * 8), mk);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = mk[i] > 0.f ? v[i] : 0.f;
        }
        *reinterpret_cast<uint4 *>(dx + px * lddx + c8 * 8) = pack8_lim<T>(v, __builtin_inff());
    }
}

// bilinear x2 (align_corners=True) backward, gather form over the INPUT pixels; dy lives in a
// (Ho,Wo) tensor (the concat buffer slice) at offset (py,px).
// IT: index type of the element decode -- 32-bit whenever the tensor allows it (a 64-bit division / modulo by a runtime divisor is ~100 instructions on this ISA,
// three of them per element)
template <typename T, typename IT = long>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const uint16_t *dy, uint16_t *dx, int h, int w, int C8, int lddy, int lddx,
                                                              int Ho, int Wo, int py, int px, long total) {
    const int oh_sz = 2 * h, ow_sz = 2 * w;
    const float rh = oh_sz > 1 ? (float)(h - 1) / (float)(oh_sz - 1) : 0.f;
    const float rw = ow_sz > 1 ? (float)(w - 1) / (float)(ow_sz - 1) : 0.f;
    for (long idx0 = (long)blockIdx.x * 256 + threadIdx.x; idx0 < total; idx0 += (long)gridDim.x * 256) {
        const IT idx = (IT)idx0;
        const int c8 = (int)(idx % (IT)C8);
        IT r = idx / (IT)C8;
        const int iw = (int)(r % (IT)w); r /= (IT)w;
        const int ih = (int)(r % (IT)h);
        const long n = (long)(r / (IT)h);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        // candidate output rows/cols: source index h1r = rh*uh lies in (ih-1, ih+1). The six candidates' row and column weights are computed ONCE per input pixel
        // (the first version recomputed the column weight inside the row loop: 36 weight evaluations per pixel, the kernel ran at 2.2 TB/s of its bytes on ALU);
        // same products, same summation order (uh, then uw, ascending): bit-identical
        const int uh_lo = max(0, 2 * ih - 2), uw_lo = max(0, 2 * iw - 2);
        float whv[6], wwv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int uh = uh_lo + k;
            const float h1r = rh * uh;
            const int h1 = (int)h1r;
            const int h1p = h1 < h - 1 ? 1 : 0;
            const float hl1 = h1r - h1, hl0 = 1.f - hl1;
            whv[k] = uh <= min(oh_sz - 1, 2 * ih + 3) ? (h1 == ih ? hl0 : 0.f) + (h1 + h1p == ih ? hl1 : 0.f) : 0.f;
            const int uw = uw_lo + k;
            const float w1r = rw * uw;
            const int w1 = (int)w1r;
            const int w1p = w1 < w - 1 ? 1 : 0;
            const float wl1 = w1r - w1, wl0 = 1.f - wl1;
            wwv[k] = uw <= min(ow_sz - 1, 2 * iw + 3) ? (w1 == iw ? wl0 : 0.f) + (w1 + w1p == iw ? wl1 : 0.f) : 0.f;
        }
        // A candidate row with weight: its six column candidates are loaded TOGETHER (clamped to the frame; the ones without weight are not added) -- the first form
        // tested each column's weight in front of its load, a chain of up to 16 dependent-in-order loads per thread (1.7 TB/s of its bytes).
        int uwc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) uwc[k] = min(uw_lo + k, ow_sz - 1);
#pragma unroll
        for (int kh_ = 0; kh_ < 6; ++kh_) {
            const float wh = whv[kh_];
            if (wh == 0.f) continue;
            const uint16_t *rowp = dy + ((n * Ho + uh_lo + kh_ + py) * Wo + px) * (long)lddy + c8 * 8;
            uint4 q[6];
#pragma unroll
            for (int kw_ = 0; kw_ < 6; ++kw_) q[kw_] = *reinterpret_cast<const uint4 *>(rowp + (long)uwc[kw_] * lddy);
#pragma unroll
            for (int kw_ = 0; kw_ < 6; ++kw_) {
                const float ww = wwv[kw_];
                if (ww == 0.f) continue;
                float g[8];
                unpack8<T>(q[kw_], g);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += wh * ww * g[i];
            }
        }
        *reinterpret_cast<uint4 *>(dx + ((n * h + ih) * w + iw) * (long)lddx + c8 * 8) = pack8_lim<T>(acc, __builtin_inff());
    }
}

// fp32 NC(T)HW gradient (+ optional sigmoid backward with the module output y) -> 16-bit channels-last, cpad 8
template <typename T>
__global__ __launch_bounds__(256) void nchw_grad_to_cl_kernel(const float *dy, const float *y, uint16_t *out, int c, long thw, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long px = idx % thw;
        const long n = idx / thw;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        for (int ch = 0; ch < c; ++ch) {
            const long o = (n * c + ch) * thw + px;
            float g = dy[o];
            if (y) { const float s = y[o]; g *= s * (1.f - s); }
            v[ch] = g;
        }
        *reinterpret_cast<uint4 *>(out + idx * 8) = pack8_lim<T>(v, __builtin_inff());
    }
}

// 16-bit channels-last (first c of ld channels) -> fp32 NC(T)HW, strided destination allowed
template <typename T>
__global__ __launch_bounds__(256) void cl_to_nchw_strided_kernel(const uint16_t *x, float *y, int c, int t, int h, int w, int ldx, long sn,
                                                                  long sc, long st, long sh, long sw, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long r = idx;
        const int iw = (int)(r % w); r /= w;
        const int ih = (int)(r % h); r /= h;
        const int it = (int)(r % t);
        const long n = r / t;
        for (int ch = 0; ch < c; ++ch) y[n * sn + ch * sc + it * st + ih * sh + iw * sw] = T::to_f32(x[idx * ldx + ch]);
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

#define TS_DT(code) ((code) == TEDSPAD_F16 || (code) == TEDSPAD_BF16)
#define TS_ZDT(zcode, code, ldz) ((zcode) == TEDSPAD_F32 || ((zcode) == (code) && (ldz) % 8 == 0))
#define LAUNCH_T(dtype, KERN, grid, ...)                                                               \
    do {                                                                                               \
        if ((dtype) == TEDSPAD_F16) hipLaunchKernelGGL(KERN<F16>, grid, dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERN<BF16>, grid, dim3(256), 0, s, __VA_ARGS__);                       \
    } while (0)

namespace tedspad {
namespace {
// eval-mode BatchNorm as y = x*scale + shift, folded in fp64 and rounded once (engine.fold_bn; one launch per BN
// instead of ~6 elementwise torch launches: the frozen network is re-folded after every optimizer step of the other phase)
__global__ void bn_fold_kernel(const float *gamma, const float *beta, const float *mean, const float *var, const float *conv_bias,
                               double eps, int C, float *scale, float *shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double inv = (double)gamma[c] / sqrt((double)var[c] + eps);
    double sh = (double)beta[c] - (double)mean[c] * inv;
    if (conv_bias) sh += (double)conv_bias[c] * inv;
    scale[c] = (float)inv;
    shift[c] = (float)sh;
}
}  // namespace
}  // namespace tedspad

extern "C" int32_t tedspad_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, const float *conv_bias,
                                   double eps, int32_t C, float *scale, float *shift, void *stream) {
    TS_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "tedspad_bn_fold: bad arguments");
    hipLaunchKernelGGL(tedspad::bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, conv_bias, eps, C,
                       scale, shift);
    return tedspad::check_launch("tedspad_bn_fold");
}

extern "C" int32_t tedspad_bn_finalize(const float *stats, int32_t stats_ld, int64_t count, const float *gamma, const float *beta,
                                       float eps, float momentum, float *running_mean, float *running_var, float *scale,
                                       float *shift, float *mean, float *invstd, int32_t C, void *stream) {
    TS_REQUIRE(stats && gamma && beta && scale && shift && mean && invstd && C > 0 && count > 0 && stats_ld >= C, "tedspad_bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, stats_ld, (float)count, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift, mean, invstd, C, (const float *)nullptr);
    return check_launch("tedspad_bn_finalize");
}

extern "C" int32_t tedspad_scale_shift_act(const float *z, const float *scale, const float *shift, const void *res, void *y, int64_t pixels,
                                           int32_t C, int32_t ldz, int32_t ldres, int32_t ldy, int32_t relu, int32_t dtype, void *stream) {
    TS_REQUIRE(z && scale && shift && y && pixels > 0 && C > 0 && C % 8 == 0 && ldz % 4 == 0 && ldy % 8 == 0 && TS_DT(dtype) && (uintptr_t)z % 16 == 0, "tedspad_scale_shift_act: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_T(dtype, bn_apply_kernel, dim3(grid_for(pixels * (C / 8))), z, scale, shift, (const uint16_t *)res, (uint16_t *)y, (long)pixels, C / 8, ldz, ldres, ldy, relu);
    return check_launch("tedspad_scale_shift_act");
}

extern "C" int32_t tedspad_bn_bwd_reduce(const void *dy, const void *y, const void *z, int32_t zdtype, const float *mean, const float *invstd, const float *gamma,
                                         const float *beta, float *sums,
                                         int32_t sums_ld, int64_t pixels, int32_t C, int32_t lddy, int32_t ldy, int32_t ldz, int32_t relu,
                                         int32_t groups, int32_t dtype, void *stream) {
    TS_REQUIRE(dy && sums && pixels > 0 && C > 0 && C % 8 == 0 && TS_DT(dtype) && (!relu || y || (z && gamma && beta)) && (!z || (mean && invstd && TS_ZDT(zdtype, dtype, ldz))) && sums_ld >= C && groups >= 1 && groups < 65536,
               "tedspad_bn_bwd_reduce: bad arguments (relu needs y, or z + gamma + beta to recompute the mask)");
    TS_REQUIRE((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "tedspad_bn_bwd_reduce: mean / invstd / gamma / beta must be 16-byte aligned (read as 16-byte vectors)");
    const int C8 = C / 8, C8L = C8 < 32 ? C8 : 32, cgroups = (C8 + C8L - 1) / C8L, PL = 256 / C8L;
    long pblocks = (pixels + (long)PL * 8 - 1) / ((long)PL * 8);   // >= 8 pixels per lane, but enough workgroups to cover the chip
    // Every workgroup ends in 2 x C float atomics on the same 2 x C addresses: past ~1000 workgroups per address the atomics, not the bytes, set the time, the sooner the
    // more channels there are (scripts/bn_probe.py, 384 frames: 112^2 x 64 237 us at 2048 workgroups, 217 at 1024, 254 at 512; 14^2 x 512 63 / 46 / 36 us).
    static const long wg_env = getenv("TEDSPAD_BNR_BLOCKS") ? atol(getenv("TEDSPAD_BNR_BLOCKS")) : 0;      // A/B knob
    const long wg_cap = wg_env > 0 ? wg_env : (C <= 128 ? 1024 : 512);
    if (pblocks > wg_cap / cgroups) pblocks = wg_cap / cgroups;
    if (pblocks < 1) pblocks = 1;
    hipStream_t s = (hipStream_t)stream;
    static const int uf_env = getenv("TEDSPAD_BNR_UNROLL") ? atoi(getenv("TEDSPAD_BNR_UNROLL")) : 0;      // A/B knob: 1 / 4 everywhere
    const bool unroll = uf_env ? uf_env > 1 : pixels / (pblocks * PL) < 48;
#define BNR_ARGS dim3((unsigned)(pblocks * cgroups), groups), dim3(256), 0, s, (const uint16_t *)dy, (const uint16_t *)y, z, (int)(zdtype != TEDSPAD_F32), mean, invstd, gamma, beta, \
                 sums, sums_ld, (long)pixels, C8, lddy, ldy, ldz, relu
    if (dtype == TEDSPAD_F16) { if (unroll) hipLaunchKernelGGL((bn_bwd_reduce_kernel<F16, 4>), BNR_ARGS); else hipLaunchKernelGGL((bn_bwd_reduce_kernel<F16, 1>), BNR_ARGS); }
    else { if (unroll) hipLaunchKernelGGL((bn_bwd_reduce_kernel<BF16, 4>), BNR_ARGS); else hipLaunchKernelGGL((bn_bwd_reduce_kernel<BF16, 1>), BNR_ARGS); }
#undef BNR_ARGS
    return check_launch("tedspad_bn_bwd_reduce");
}

extern "C" int32_t tedspad_bn_bwd_apply(const void *dy, const void *y, const void *z, int32_t zdtype, const float *mean, const float *invstd, const float *gamma,
                                        const float *beta, const float *sums, int32_t sums_ld, void *dz, void *dres, float *dbias, int32_t dbias_slots, int64_t pixels, int32_t C, int32_t lddy,
                                        int32_t ldy, int32_t ldz, int32_t lddz, int32_t lddres, int32_t relu, int32_t groups, int32_t dtype, void *stream) {
    TS_REQUIRE(dy && z && mean && invstd && gamma && sums && dz && pixels > 0 && C > 0 && C % 8 == 0 && TS_DT(dtype) && (!relu || y || beta) && groups >= 1 && groups < 65536 && TS_ZDT(zdtype, dtype, ldz) && (!dbias || dbias_slots >= 1) && (size_t)C * 16 + 9216 <= 65536,   // [4][C] fp32 terms + the 9 KB reduction image: the default 64 KB dynamic-LDS limit (C <= 3520)
               "tedspad_bn_bwd_apply: bad arguments (relu needs y, or beta to recompute the mask)");
    hipStream_t s = (hipStream_t)stream;
#define BWD_APPLY(TT, DD)                                                                                                                            \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, DD>), dim3(grid_for_iters(pixels * (C / 8), 8), groups), dim3(256), (size_t)C * 16, s, (const uint16_t *)dy, (const uint16_t *)y, z, \
                       (int)(zdtype != TEDSPAD_F32), mean, invstd, gamma, beta, sums, sums_ld, 1.f / (float)pixels, (uint16_t *)dz, (uint16_t *)dres, dbias,      \
                       dbias_slots, (long)pixels, C / 8, lddy, ldy, ldz, lddz, lddres, relu)
    if (dtype == TEDSPAD_F16) { if (dbias) BWD_APPLY(F16, true); else BWD_APPLY(F16, false); }
    else { if (dbias) BWD_APPLY(BF16, true); else BWD_APPLY(BF16, false); }
#undef BWD_APPLY
    return check_launch("tedspad_bn_bwd_apply");
}

extern "C" int32_t tedspad_maxpool_bwd(const tedspad_pool_desc *d, const void *x, const uint8_t *idx, const void *dy, int32_t lddy, const void *add,
                                       int32_t ldadd, void *dx, int32_t lddx, int32_t relu_mask, void *stream) {
    TS_REQUIRE(d && x && idx && dy && dx && d->c % 8 == 0 && TS_DT(d->dtype), "tedspad_maxpool_bwd: bad arguments");
    PoolBKP p;
    p.x = (const uint16_t *)x; p.idx = idx; p.dy = (const uint16_t *)dy; p.add = (const uint16_t *)add; p.dx = (uint16_t *)dx;
    p.Ti = d->t; p.Hi = d->h; p.Wi = d->w; p.C8 = d->c / 8; p.ldx = d->ldx; p.lddy = lddy; p.ldadd = ldadd; p.lddx = lddx;
    p.To = d->to; p.Ho = d->ho; p.Wo = d->wo;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw; p.st = d->st; p.sh = d->sh; p.sw = d->sw; p.pt = d->pt; p.ph = d->ph; p.pw = d->pw;
    p.relu_mask = relu_mask;
    p.total = (long)d->n * d->t * d->h * d->w * p.C8;
    hipStream_t s = (hipStream_t)stream;
    if (p.total < (1L << 31) && d->kt == 1 && d->kh == 2 && d->kw == 2 && d->st == 1 && d->sh == 2 && d->sw == 2 && d->pt == 0 && d->ph == 0 && d->pw == 0 &&
        d->h % 2 == 0 && d->w % 2 == 0 && d->to == d->t && d->ho == d->h / 2 && d->wo == d->w / 2) {
        LAUNCH_T(d->dtype, maxpool_bwd_k2s2_kernel, dim3(grid_for(p.total)), p);
        return check_launch("tedspad_maxpool_bwd");
    }
    if (p.total < (1L << 31)) {
        if (d->dtype == TEDSPAD_F16) hipLaunchKernelGGL((maxpool_bwd_kernel<F16, unsigned>), dim3(grid_for(p.total)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<BF16, unsigned>), dim3(grid_for(p.total)), dim3(256), 0, s, p);
    } else
    LAUNCH_T(d->dtype, maxpool_bwd_kernel, dim3(grid_for(p.total)), p);
    return check_launch("tedspad_maxpool_bwd");
}

extern "C" int32_t tedspad_global_avgpool_bwd(const float *dfeat, const void *mask, int32_t ldmask, void *dx, int32_t n, int32_t spatial, int32_t c,
                                              int32_t lddx, int32_t dtype, void *stream) {
    TS_REQUIRE(dfeat && dx && n > 0 && spatial > 0 && c % 8 == 0 && lddx >= c && TS_DT(dtype), "tedspad_global_avgpool_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_T(dtype, avgpool_bwd_kernel, dim3(grid_for((long)n * spatial * (c / 8))), dfeat, (const uint16_t *)mask, ldmask, (uint16_t *)dx, n, spatial, c / 8, lddx);
    return check_launch("tedspad_global_avgpool_bwd");
}

extern "C" int32_t tedspad_upsample_bilinear2x_bwd(const void *dy, void *dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t lddy,
                                                   int32_t lddx, int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, int32_t dtype,
                                                   void *stream) {
    TS_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c % 8 == 0 && lddy >= c && lddx >= c && TS_DT(dtype) && ho >= 2 * h + pad_top && wo >= 2 * w + pad_left,
               "tedspad_upsample_bilinear2x_bwd: bad arguments");
    const long total = (long)n * h * w * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    if (total < (1L << 31)) {
        if (dtype == TEDSPAD_F16) hipLaunchKernelGGL((upsample2x_bwd_kernel<F16, unsigned>), dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)dy, (uint16_t *)dx, h, w, c / 8, lddy, lddx, ho, wo, pad_top, pad_left, total);
        else hipLaunchKernelGGL((upsample2x_bwd_kernel<BF16, unsigned>), dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t *)dy, (uint16_t *)dx, h, w, c / 8, lddy, lddx, ho, wo, pad_top, pad_left, total);
    } else
    LAUNCH_T(dtype, upsample2x_bwd_kernel, dim3(grid_for(total)), (const uint16_t *)dy, (uint16_t *)dx, h, w, c / 8, lddy, lddx, ho, wo, pad_top, pad_left, total);
    return check_launch("tedspad_upsample_bilinear2x_bwd");
}

extern "C" int32_t tedspad_nchw_grad_to_channels_last(const float *dy, const float *y_sigmoid, void *out, int32_t n, int32_t c, int64_t thw,
                                                      int32_t dtype, void *stream) {
    TS_REQUIRE(dy && out && n > 0 && c > 0 && c <= 8 && thw > 0 && TS_DT(dtype), "tedspad_nchw_grad_to_channels_last: bad arguments");
    const long total = (long)n * thw;
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_T(dtype, nchw_grad_to_cl_kernel, dim3(grid_for(total)), dy, y_sigmoid, (uint16_t *)out, c, (long)thw, total);
    return check_launch("tedspad_nchw_grad_to_channels_last");
}

extern "C" int32_t tedspad_channels_last_to_nchw_strided(const void *x, float *y, int32_t n, int32_t c, int32_t t, int32_t h, int32_t w, int32_t ldx,
                                                         int64_t sn, int64_t sc, int64_t st_, int64_t sh, int64_t sw, int32_t dtype, void *stream) {
    TS_REQUIRE(x && y && n > 0 && c > 0 && ldx >= c && TS_DT(dtype), "tedspad_channels_last_to_nchw_strided: bad arguments");
    const long total = (long)n * t * h * w;
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_T(dtype, cl_to_nchw_strided_kernel, dim3(grid_for(total)), (const uint16_t *)x, y, c, t, h, w, ldx, (long)sn, (long)sc, (long)st_, (long)sh, (long)sw, total);
    return check_launch("tedspad_channels_last_to_nchw_strided");
}

extern "C" int32_t tedspad_bn_train_apply(const void *z, int32_t zdtype, const float *stats, int32_t stats_ld, int64_t count, const float *gamma, const float *beta,
                                          float eps, float momentum, float *running_mean, float *running_var, float *mean, float *invstd, int32_t C,
                                          const void *res, void *y, int64_t pixels, int32_t Cz, int32_t ldz, int32_t ldres, int32_t ldy, int32_t relu,
                                          int32_t groups, int32_t dtype, void *stream) {
    TS_REQUIRE(z && stats && gamma && beta && mean && invstd && y && count > 0 && pixels > 0 && C > 0 && Cz >= C && Cz % 8 == 0 && stats_ld >= C && ldz % 4 == 0 &&
                   ldy % 8 == 0 && TS_DT(dtype) && (uintptr_t)z % 16 == 0 && groups >= 1 && groups < 65536 && TS_ZDT(zdtype, dtype, ldz),
               "tedspad_bn_train_apply: bad arguments (z is fp32, or 16-bit of y's dtype with ldz % 8 == 0)");
    hipStream_t s = (hipStream_t)stream;
    const int gx = grid_for_iters(pixels * (Cz / 8), 8);
    static const int uf_env = getenv("TEDSPAD_BNA_UNROLL") ? atoi(getenv("TEDSPAD_BNA_UNROLL")) : 0;      // A/B knob: 1 / 4 everywhere
    // UF = 4 where a thread walks few items of a wide layer (measured: 28^2 x 256 and deeper gain 8 ... 45 %; 224^2 x 64 at the same item count loses 15 %;
    // the same unrolling of tedspad_bn_bwd_apply, whose terms live in LDS, lost 5 ... 20 % everywhere and was not kept)
    const bool unroll = uf_env ? uf_env > 1 : (Cz >= 256 && pixels * (Cz / 8) / ((long)gx * 256) <= 16);
#define BNA_ARGS dim3(gx, groups), dim3(256), 0, s, z, (int)(zdtype != TEDSPAD_F32), stats, stats_ld, (float)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, C, \
                 (const uint16_t *)res, (uint16_t *)y, (long)pixels, Cz / 8, ldz, ldres, ldy, relu
    if (dtype == TEDSPAD_F16) { if (unroll) hipLaunchKernelGGL((bn_train_apply_kernel<F16, 4>), BNA_ARGS); else hipLaunchKernelGGL((bn_train_apply_kernel<F16, 1>), BNA_ARGS); }
    else { if (unroll) hipLaunchKernelGGL((bn_train_apply_kernel<BF16, 4>), BNA_ARGS); else hipLaunchKernelGGL((bn_train_apply_kernel<BF16, 1>), BNA_ARGS); }
#undef BNA_ARGS
    return check_launch("tedspad_bn_train_apply");
}

namespace tedspad {
int32_t det_ctl_train_ops(int op, int on) { return det_ctl(op, on); }
}  // namespace tedspad

extern "C" int32_t tedspad_set_deterministic(int32_t on) {
    TS_REQUIRE(hipDeviceSynchronize() == hipSuccess, "tedspad_set_deterministic: device error pending");      // no launch may be in its gate while the state is rewritten
    const int o = on ? 1 : 0;
    TS_REQUIRE(tedspad::det_ctl_igemm(0, o) == 0 && tedspad::det_ctl_patch(0, o) == 0 && tedspad::det_ctl_wgrad(0, o) == 0 && tedspad::det_ctl_train_ops(0, o) == 0,
               "tedspad_set_deterministic: cannot write the device state");
    return TEDSPAD_OK;
}

extern "C" int32_t tedspad_deterministic_giveups(void) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const int32_t a = tedspad::det_ctl_igemm(1, 0), b = tedspad::det_ctl_patch(1, 0), c = tedspad::det_ctl_wgrad(1, 0), d = tedspad::det_ctl_train_ops(1, 0);
    return (a < 0 || b < 0 || c < 0 || d < 0) ? -1 : a + b + c + d;
}
