// Weight packing on the device: fp32 parameters in the reference layout (co, ci, kt, kh, kw) -> the 16-bit
// [rows_pad][kpad] matrices the conv kernels read, in ONE launch per matrix. The training step repacks every
// convolution it touches after each optimizer update (forward matrix + the data-gradient matrices); done with
// torch indexing ops that was ~5000 tiny launches per step.
//
//   FWD   out[n][((dt*KH + dh)*KWk + dw)*CINk + c]            = wk(n, c, dt, dh, dw) * scale[n]
//   DGRAD out[c][((et*Eh + eh)*Ew + ew)*CO8 + n]              = wk(n, c, ct + st*(Et-1-et), ch + sh*(Eh-1-eh), cw + sw*(Ew-1-ew)) * scale[n]
//         (the flipped, channel-transposed taps of one parity class of a stride-s convolution: train_engine.DgradPlan)
//   wk = the weight as the kernel sees it: channels zero-padded to CINk, or, for the Cin=3 stems, the pixel-pair form
//        wk(n, j*4 + c, dt, dh, d) = w(n, c, dt, dh, 2d + j - shift)   (engine.stem_pair_form)
#include "common.h"

namespace tedspad {
namespace {

struct PackKP {
    const float *w;       // (co, ci, kt, kh, kw) contiguous
    const float *scale;   // per co, or nullptr
    uint16_t *out;
    int co, ci, kt, kh, kw;
    int cink, kwk;        // kernel-form channels / width taps
    int pair_shift;       // >= 0: pixel-pair form with this shift; -1: plain
    int mode;             // 0 FWD, 1 DGRAD
    int rows, rows_pad, kpad;
    int Et, Eh, Ew, ct, ch, cw, st, sh, sw, co8;   // DGRAD
};

__device__ __forceinline__ float wk(const PackKP &p, int n, int c, int dt, int dh, int dw) {
    if (n >= p.co) return 0.f;
    int cs = c, ks = dw;
    if (p.pair_shift >= 0) {
        const int j = c >> 2;
        cs = c & 3;
        ks = 2 * dw + j - p.pair_shift;
        if (ks < 0 || ks >= p.kw) return 0.f;
    }
    if (cs >= p.ci) return 0.f;
    const float v = p.w[((((size_t)n * p.ci + cs) * p.kt + dt) * p.kh + dh) * p.kw + ks];
    return p.scale ? v * p.scale[n] : v;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackKP p) {
    const long chunks = (long)p.rows_pad * (p.kpad / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < chunks; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / (p.kpad / 8));
        const int k0 = (int)(idx % (p.kpad / 8)) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (p.mode == 0) {
            const int K = p.kt * p.kh * p.kwk * p.cink;
            if (row < p.rows && k0 < K) {
                const int c0 = k0 % p.cink; int tap = k0 / p.cink;
                const int dw = tap % p.kwk; tap /= p.kwk;
                const int dh = tap % p.kh; const int dt = tap / p.kh;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = wk(p, row, c0 + i, dt, dh, dw);
            }
        } else {
            const int K = p.Et * p.Eh * p.Ew * p.co8;
            if (row < p.rows && k0 < K) {
                const int n0 = k0 % p.co8; int e = k0 / p.co8;
                const int ew = e % p.Ew; e /= p.Ew;
                const int eh = e % p.Eh; const int et = e / p.Eh;
                const int dt = p.ct + p.st * (p.Et - 1 - et), dh = p.ch + p.sh * (p.Eh - 1 - eh), dw = p.cw + p.sw * (p.Ew - 1 - ew);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = wk(p, n0 + i, row, dt, dh, dw);
            }
        }
        *reinterpret_cast<uint4 *>(p.out + (size_t)row * p.kpad + k0) = pack8<T>(v);
    }
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_pack_conv_weights(const float *w, const float *scale, void *out, int32_t co, int32_t ci, int32_t kt, int32_t kh,
                                             int32_t kw, int32_t cink, int32_t kwk, int32_t pair_shift, int32_t mode, int32_t rows,
                                             int32_t rows_pad, int32_t kpad, const int32_t *dgrad_geo, int32_t dtype, void *stream) {
    TS_REQUIRE(w && out && co > 0 && ci > 0 && kt > 0 && kh > 0 && kw > 0 && cink % 8 == 0 && kpad % 8 == 0 && rows_pad >= rows && rows > 0,
               "tedspad_pack_conv_weights: bad arguments");
    TS_REQUIRE(dtype == TEDSPAD_F16 || dtype == TEDSPAD_BF16, "tedspad_pack_conv_weights: bad dtype");
    TS_REQUIRE(mode == 0 || (mode == 1 && dgrad_geo), "tedspad_pack_conv_weights: DGRAD mode needs its geometry");
    PackKP p;
    p.w = w; p.scale = scale; p.out = (uint16_t *)out;
    p.co = co; p.ci = ci; p.kt = kt; p.kh = kh; p.kw = kw; p.cink = cink; p.kwk = kwk; p.pair_shift = pair_shift; p.mode = mode;
    p.rows = rows; p.rows_pad = rows_pad; p.kpad = kpad;
    p.Et = p.Eh = p.Ew = 1; p.ct = p.ch = p.cw = 0; p.st = p.sh = p.sw = 1; p.co8 = (co + 7) / 8 * 8;
    if (mode == 1) {
        p.Et = dgrad_geo[0]; p.Eh = dgrad_geo[1]; p.Ew = dgrad_geo[2]; p.ct = dgrad_geo[3]; p.ch = dgrad_geo[4]; p.cw = dgrad_geo[5];
        p.st = dgrad_geo[6]; p.sh = dgrad_geo[7]; p.sw = dgrad_geo[8];
        TS_REQUIRE((long)p.Et * p.Eh * p.Ew * p.co8 <= kpad, "tedspad_pack_conv_weights: kpad too small for the data-gradient matrix");
    } else {
        TS_REQUIRE((long)kt * kh * kwk * cink <= kpad, "tedspad_pack_conv_weights: kpad too small");
    }
    const long chunks = (long)rows_pad * (kpad / 8);
    long g = (chunks + 255) / 256; if (g > 4096) g = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TEDSPAD_F16) hipLaunchKernelGGL(pack_kernel<F16>, dim3((unsigned)g), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(pack_kernel<BF16>, dim3((unsigned)g), dim3(256), 0, s, p);
    return check_launch("tedspad_pack_conv_weights");
}
