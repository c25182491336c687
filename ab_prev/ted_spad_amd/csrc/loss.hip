// Fused loss kernels of the anonymizer training step for gfx950: each computes the loss
// value AND the gradient w.r.t. its inputs in one launch (the reference builds ~15 tiny
// kernels + a host-side numpy mask per NT-Xent call: aux_code/nt_xent_original.py:26-32,49-70).
//
//  * NT-Xent (aux_code/nt_xent_original.py:49-70): R = cat[zjs, zis] (2N x C); S = R R^T / T on the
//    exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, one 32x32 tile of S per wave); the "same
//    representation" mask is the main diagonal (positives are the +-N diagonals and stay in the
//    softmax denominator exactly as in the reference's [pos | neg] logits); row log-sum-exp and the
//    loss reduction use wavefront shuffles; dR = (G + G^T) R / T with G = (softmax - onehot)/2N.
//  * TripletMarginLoss(margin, p=2, eps=1e-6)  (train_anonymizer.py:349-350,115): one wave per row.
//  * CrossEntropyLoss (mean)                    (train_anonymizer.py:347,107): one wave per row.
#include "common.h"

namespace tedspad {
namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

constexpr int NTX_MAX_ROWS = 64;   // 2N <= 64 (reference: 2N = 24)
constexpr int NTX_MAX_C = 256;

// single workgroup, 4 waves: wave w owns the 32x32 tile (w>>1, w&1) of S.
__global__ __launch_bounds__(256) void ntxent_kernel(const float *zis, const float *zjs, float *loss, float *dzis, float *dzjs,
                                                      int N, int C, float inv_t, int cosine) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int R2 = 2 * N;
    float *Rm = sm;                                  // [64][C+1] rows >= 2N are zero
    float *S = Rm + NTX_MAX_ROWS * (C + 1);          // [64][65]  logits, then G
    float *nrm = S + NTX_MAX_ROWS * 65;              // [64] row norms (cosine)
    float *red = nrm + NTX_MAX_ROWS;                 // [4]
    float *dR = red + 4;                             // [64][C+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ldr = C + 1;
    for (int i = tid; i < NTX_MAX_ROWS * C; i += 256) {
        const int r = i / C, c = i % C;
        float v = 0.f;
        if (r < N) v = zjs[r * C + c];              // representations = cat([zjs, zis])  (:50)
        else if (r < R2) v = zis[(r - N) * C + c];
        Rm[r * ldr + c] = v;
    }
    __syncthreads();
    if (cosine) {                                    // CosineSimilarity(dim=-1), eps 1e-8
        for (int r = wave; r < NTX_MAX_ROWS; r += 4) {
            float s = 0.f;
            for (int c = lane; c < C; c += 64) s += Rm[r * ldr + c] * Rm[r * ldr + c];
            s = wsum(s);
            const float nr = __builtin_fmaxf(sqrtf(s), 1e-8f);
            if (lane == 0) nrm[r] = nr;
            for (int c = lane; c < C; c += 64) Rm[r * ldr + c] /= nr;
        }
        __syncthreads();
    }
    // ---- S = R R^T on the f32 MFMA: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31] ------------------------
    {
        const int ti = wave >> 1, tj = wave & 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float *pa = Rm + (ti * 32 + (lane & 31)) * ldr + (lane >> 5);
        const float *pb = Rm + (tj * 32 + (lane & 31)) * ldr + (lane >> 5);
        for (int k = 0; k < C; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
        // C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = tj * 32 + (lane & 31);
            S[row * 65 + col] = acc[r] * inv_t;      // logits /= temperature (:64)
        }
    }
    __syncthreads();
    // ---- row-wise masked log-sum-exp; lane = column ----------------------------------------------
    float part = 0.f;
    for (int i = wave; i < R2; i += 4) {
        const int pos = (i + N) % R2;                // l_pos / r_pos diagonals (:56-60)
        const bool valid = lane < R2 && lane != i;   // mask removes only the main diagonal from [pos|neg]
        const float s = valid ? S[i * 65 + lane] : -3.0e38f;
        const float mx = wmax(s);
        const float e = valid ? __expf(s - mx) : 0.f;
        const float den = wsum(e);
        const float lse = mx + __logf(den);
        const float spos = S[i * 65 + pos];
        if (lane == 0) part += lse - spos;           // CE(sum) with label 0 (:66-67)
        // G_ij = (softmax_ij - [j == pos]) / 2N
        float g = valid ? e / den : 0.f;
        if (lane == pos) g -= 1.f;
        if (lane < NTX_MAX_ROWS) S[i * 65 + lane] = g / (float)R2;
    }
    for (int i = R2 + wave; i < NTX_MAX_ROWS; i += 4) S[i * 65 + lane] = 0.f;
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (tid == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)R2;   // loss / (2 * batch_size) (:70)
    // ---- dRhat = (G + G^T) Rhat / T ;  cosine: project through the normalisation -----------------
    if (!dzis) return;
    for (int i = wave; i < R2; i += 4) {
        float dot = 0.f;
        for (int c = lane; c < C; c += 64) {
            float d = 0.f;
            for (int j = 0; j < R2; ++j) d += (S[i * 65 + j] + S[j * 65 + i]) * Rm[j * ldr + c];
            d *= inv_t;
            dR[i * ldr + c] = d;
            dot += d * Rm[i * ldr + c];
        }
        dot = wsum(dot);
        for (int c = lane; c < C; c += 64) {
            float d = dR[i * ldr + c];
            if (cosine) d = (d - dot * Rm[i * ldr + c]) / nrm[i];   // (I - rhat rhat^T)/|r| applied to dRhat
            if (i < N) dzjs[i * C + c] = d;
            else dzis[(i - N) * C + c] = d;
        }
    }
}

// one wave per row: d(x,y) = ||x - y + eps||_2 ; loss_b = max(d(a,p) - d(a,n) + margin, 0)
__global__ __launch_bounds__(256) void triplet_kernel(const float *a, const float *p, const float *n, float *loss_rows,
                                                       float *da, float *dp, float *dn, int B, int C, float margin, float eps) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= B) return;
    float sp = 0.f, sn = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float x = a[row * C + c];
        const float u = x - p[row * C + c] + eps, v = x - n[row * C + c] + eps;
        sp += u * u; sn += v * v;
    }
    sp = sqrtf(wsum(sp)); sn = sqrtf(wsum(sn));
    const float l = sp - sn + margin;
    const bool active = l > 0.f;
    if (lane == 0) loss_rows[row] = active ? l : 0.f;
    if (!da) return;
    const float invb = active ? 1.f / (float)B : 0.f;
    for (int c = lane; c < C; c += 64) {
        const float x = a[row * C + c];
        const float gu = sp > 0.f ? (x - p[row * C + c] + eps) / sp * invb : 0.f;
        const float gv = sn > 0.f ? (x - n[row * C + c] + eps) / sn * invb : 0.f;
        da[row * C + c] = gu - gv;
        dp[row * C + c] = -gu;
        dn[row * C + c] = gv;
    }
}

__global__ __launch_bounds__(256) void ce_kernel(const float *logits, const long *labels, float *loss_rows, float *dlogits, int B, int C) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= B) return;
    float mx = -3.0e38f;
    for (int c = lane; c < C; c += 64) mx = __builtin_fmaxf(mx, logits[row * C + c]);
    mx = wmax(mx);
    float den = 0.f;
    for (int c = lane; c < C; c += 64) den += __expf(logits[row * C + c] - mx);
    den = wsum(den);
    const int lab = (int)labels[row];
    if (lane == 0) loss_rows[row] = mx + __logf(den) - logits[row * C + lab];
    if (!dlogits) return;
    const float invb = 1.f / (float)B;
    for (int c = lane; c < C; c += 64)
        dlogits[row * C + c] = (__expf(logits[row * C + c] - mx) / den - (c == lab ? 1.f : 0.f)) * invb;
}

// mean of B row losses -> scalar (B is tiny: one wave)
__global__ void mean_kernel(const float *rows, float *out, int B) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += rows[i];
    s = wsum(s);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_ntxent_fwd_bwd(const float *zis, const float *zjs, float *loss, float *dzis, float *dzjs, int32_t N,
                                          int32_t C, float temperature, int32_t use_cosine, void *stream) {
    TS_REQUIRE(zis && zjs && loss && N > 0 && C > 0 && temperature > 0.f, "tedspad_ntxent_fwd_bwd: bad arguments");
    TS_REQUIRE((dzis == nullptr) == (dzjs == nullptr), "tedspad_ntxent_fwd_bwd: pass both gradient buffers or neither");
    TS_REQUIRE(2 * N <= NTX_MAX_ROWS && C <= NTX_MAX_C && C % 2 == 0,
               "tedspad_ntxent_fwd_bwd: supports 2N <= 64 and even C <= 256 (reference: N=12, C=128)");
    const size_t lds = (size_t)(2 * NTX_MAX_ROWS * (C + 1) + NTX_MAX_ROWS * 65 + NTX_MAX_ROWS + 4) * sizeof(float);
    static thread_local bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)ntxent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_ntxent_fwd_bwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr = true;
    }
    hipLaunchKernelGGL(ntxent_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, zis, zjs, loss, dzis, dzjs, N, C, 1.f / temperature, use_cosine);
    return check_launch("tedspad_ntxent_fwd_bwd");
}

extern "C" int32_t tedspad_triplet_fwd_bwd(const float *a, const float *p, const float *n, float *loss, float *row_ws, float *da,
                                           float *dp, float *dn, int32_t B, int32_t C, float margin, float eps, void *stream) {
    TS_REQUIRE(a && p && n && loss && row_ws && B > 0 && C > 0, "tedspad_triplet_fwd_bwd: bad arguments");
    TS_REQUIRE((!da && !dp && !dn) || (da && dp && dn), "tedspad_triplet_fwd_bwd: pass all three gradient buffers or none");
    hipLaunchKernelGGL(triplet_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, p, n, row_ws, da, dp, dn, B, C, margin, eps);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, row_ws, loss, B);
    return check_launch("tedspad_triplet_fwd_bwd");
}

extern "C" int32_t tedspad_cross_entropy_fwd_bwd(const float *logits, const int64_t *labels, float *loss, float *row_ws,
                                                 float *dlogits, int32_t B, int32_t C, void *stream) {
    TS_REQUIRE(logits && labels && loss && row_ws && B > 0 && C > 0, "tedspad_cross_entropy_fwd_bwd: bad arguments");
    hipLaunchKernelGGL(ce_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, (const long *)labels, row_ws, dlogits, B, C);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, row_ws, loss, B);
    return check_launch("tedspad_cross_entropy_fwd_bwd");
}
