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
__device__ __forceinline__ void pack_body(const PackKP &p, long first, long stride) {
    const long chunks = (long)p.rows_pad * (p.kpad / 8);
    for (long idx = first; idx < chunks; idx += stride) {
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

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackKP p) {
    pack_body<T>(p, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---- the multi-job launches: a device table of jobs, workgroup -> job by binary search over the jobs' first workgroups ----------
// After an optimizer step EVERY 16-bit weight image of the updated network is stale (forward matrix + data-gradient matrices per
// convolution, plus the folded BatchNorm vectors of a frozen / eval-mode network): one launch per image was 80-270 launches of a
// few microseconds each per training iteration and the same number of host-side object rebuilds. The job tables are static
// (parameters, running statistics and the images all keep their addresses), so the host uploads them once.
static_assert(sizeof(tedspad_pack_job) == 120 && sizeof(tedspad_fold_job) == 96 && sizeof(tedspad_wgrad_unpack_job) == 64,
              "job structs are part of the ABI (ted_spad_amd/_lib.py mirrors them)");

// workgroup -> job: the last entry with block0 <= b. Every lane probes one entry per round (64-ary search: one round up to 64 jobs, two up to 4096) -- the
// workgroups of these launches live for a few microseconds, and a binary search was 7-8 DEPENDENT loads (~5 us) in front of the first useful one.
template <typename J>
__device__ __forceinline__ int find_job(const J *tab, int n, int b) {
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n;                                  // the answer is in [lo, hi); tab[lo].block0 <= b (tab[0].block0 == 0)
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int idx = lo + lane * step;
        const bool ok = idx < hi && tab[idx].block0 <= b;
        const unsigned long long m = __ballot(ok);       // lane 0 always votes
        const int last = 63 - __builtin_clzll(m);
        lo += last * step;
        hi = min(hi, lo + step);
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

__device__ __forceinline__ PackKP job_params(const tedspad_pack_job &j) {
    PackKP p;
    p.w = j.w; p.scale = j.scale; p.out = (uint16_t *)j.out;
    p.co = j.co; p.ci = j.ci; p.kt = j.kt; p.kh = j.kh; p.kw = j.kw; p.cink = j.cink; p.kwk = j.kwk; p.pair_shift = j.pair_shift; p.mode = j.mode;
    p.rows = j.rows; p.rows_pad = j.rows_pad; p.kpad = j.kpad;
    p.Et = p.Eh = p.Ew = 1; p.ct = p.ch = p.cw = 0; p.st = p.sh = p.sw = 1; p.co8 = (j.co + 7) / 8 * 8;
    if (j.mode == 1) {
        p.Et = j.geo[0]; p.Eh = j.geo[1]; p.Ew = j.geo[2]; p.ct = j.geo[3]; p.ch = j.geo[4]; p.cw = j.geo[5];
        p.st = j.geo[6]; p.sh = j.geo[7]; p.sw = j.geo[8];
    }
    return p;
}

// The weight images through LDS. pack_body gathers: a forward chunk is 8 reads kt*kh*kw floats apart, a data-gradient chunk 8 reads ci*kt*kh*kw floats apart (a transpose
// by gather, every read its own cache line: 6 GB fetched per training iteration for 0.5 GB of images, `profiles/r05_train_cfg3_kernels.md`). Here a workgroup stages a TILE
// of the parameter with coalesced reads, scaled as pack_body scales, and assembles the 16-byte chunks from LDS -- bit-identical to pack_body:
//   forward:        R whole output channels = ONE run of R * ci * taps floats;
//   data gradient:  C input channels x 64 output channels: per output channel one run of C * taps floats (C = 32 / 16 / 8 for 1 / <= 3 / more taps: runs of >= 128 B).
// Pixel-pair (stem) forms, data gradients with more than 27 taps and forward rows that do not fit the tile stay on pack_body.
constexpr int PK_FLOATS = 64 * (8 * 27 + 1);                     // 13 888 floats = 55.6 KB: the largest tile (data gradient of a 3 x 3 x 3 conv)
constexpr int PK_FWD_FLOATS = 4096;                              // forward tiles: as many whole rows as fit 16 KB (one row if it is longer)
struct PkPlan { int tiled, per_tile, tiles, floats; };           // floats: LDS the job's tiles need (the launch takes the maximum over its jobs)
__host__ __device__ inline PkPlan pk_plan(int mode, int pair_shift, int kw, int kwk, int ci, int taps, int rows_pad, int co8) {
    PkPlan q = {0, 0, 0, 0};
    if (pair_shift >= 0 || kwk != kw) return q;
    if (mode == 0) {
        const long run = (long)ci * taps;
        if (run > PK_FLOATS) return q;
        int R = (int)(PK_FWD_FLOATS / run);
        R = R < 1 ? 1 : R > 64 ? 64 : R;
        q.tiled = 1; q.per_tile = R; q.tiles = (rows_pad + R - 1) / R; q.floats = (int)(R * run);
    } else {
        if (taps > 27) return q;
        const int C = taps == 1 ? 32 : taps <= 3 ? 16 : 8;
        q.tiled = 1; q.per_tile = C; q.tiles = ((rows_pad + C - 1) / C) * ((co8 + 63) / 64); q.floats = 64 * (C * taps + 1);
    }
    return q;
}

template <typename T>
__device__ __forceinline__ void pack_fwd_tiles(const PackKP &p, float *tile, const PkPlan q, int first_tile, int tile_stride) {
    const int taps = p.kt * p.kh * p.kw, run = p.ci * taps, R = q.per_tile;
    const int K = taps * p.cink, kchunks = p.kpad / 8;
    for (int t = first_tile; t < q.tiles; t += tile_stride) {
        const int n0 = t * R, nr = min(R, p.rows_pad - n0);
        const int nload = max(0, min(nr, min(p.co, p.rows) - n0));     // rows that carry weights; the rest of the tile is zero rows
        __syncthreads();                                         // the previous tile's readers are done
        for (int i = threadIdx.x; i < nload * run; i += 256) {
            float v = p.w[(size_t)n0 * run + i];
            if (p.scale) v *= p.scale[n0 + i / run];
            tile[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * kchunks; i += 256) {
            const int r = i / kchunks, k0 = (i - r * kchunks) * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (r < nload && k0 < K) {
                const int c0 = k0 % p.cink, tap = k0 / p.cink;   // (dt, dh, dw) order on both sides (kwk == kw)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = c0 + e < p.ci ? tile[r * run + (c0 + e) * taps + tap] : 0.f;
            }
            *reinterpret_cast<uint4 *>(p.out + (size_t)(n0 + r) * p.kpad + k0) = pack8<T>(v);
        }
    }
}

template <typename T>
__device__ __forceinline__ void pack_dgrad_tiles(const PackKP &p, float *tile, const PkPlan q, int first_tile, int tile_stride) {
    const int taps = p.kt * p.kh * p.kw, E = p.Et * p.Eh * p.Ew, C = q.per_tile;
    const int S = C * taps + 1;                                  // floats per output channel in the tile (+1: odd stride)
    const int n_tiles = (p.co8 + 63) / 64;
    const int kchunks = p.kpad / 8, used = E * p.co8 / 8;        // chunks per row / chunks that carry the matrix (co8 % 8 == 0)
    for (int t = first_tile; t < q.tiles; t += tile_stride) {
        const int c0 = (t / n_tiles) * C, n0 = (t % n_tiles) * 64;
        __syncthreads();                                         // the previous tile's readers are done
        for (int i = threadIdx.x; i < 64 * C * taps; i += 256) {
            const int n = i / (C * taps), rem = i - n * (C * taps);
            const int c = c0 + rem / taps;
            float v = 0.f;
            if (n0 + n < p.co && c < p.ci && c < p.rows) {
                v = p.w[((size_t)(n0 + n) * p.ci + c0) * taps + rem];
                if (p.scale) v *= p.scale[n0 + n];
            }
            tile[n * S + rem] = v;
        }
        __syncthreads();
        const int nch = min(64, p.co8 - n0) / 8;                 // 8-channel chunks of this tile per (row, tap)
        for (int i = threadIdx.x; i < C * E * nch; i += 256) {
            const int jn = i % nch, e = (i / nch) % E, r = i / (nch * E);
            if (c0 + r >= p.rows_pad) continue;
            int ee = e;
            const int ew = ee % p.Ew; ee /= p.Ew;
            const int eh = ee % p.Eh; const int et = ee / p.Eh;
            const int dt = p.ct + p.st * (p.Et - 1 - et), dh = p.ch + p.sh * (p.Eh - 1 - eh), dw = p.cw + p.sw * (p.Ew - 1 - ew);
            const int tp = (dt * p.kh + dh) * p.kw + dw;
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = tile[(jn * 8 + u) * S + r * taps + tp];
            *reinterpret_cast<uint4 *>(p.out + (size_t)(c0 + r) * p.kpad + (size_t)e * p.co8 + n0 + jn * 8) = pack8<T>(v);
        }
        if (n0 == 0) {                                           // the rows' K padding [E * co8, kpad): zeros, written by the tile that holds channel 0
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (int i = threadIdx.x; i < C * (kchunks - used); i += 256) {
                const int r = i / (kchunks - used), k = used + i % (kchunks - used);
                if (c0 + r < p.rows_pad) *reinterpret_cast<uint4 *>(p.out + (size_t)(c0 + r) * p.kpad + (size_t)k * 8) = z;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const tedspad_pack_job *tab, int njobs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // the launch's largest tile (tedspad_pack_multi)
    const int ji = find_job(tab, njobs, (int)blockIdx.x);
    const tedspad_pack_job &j = tab[ji];
    const PackKP p = job_params(j);
    const PkPlan q = pk_plan(p.mode, p.pair_shift, p.kw, p.kwk, p.ci, p.kt * p.kh * p.kw, p.rows_pad, p.co8);      // job-uniform; the host planned nblocks with it
    if (q.tiled) {
        const int b = (int)blockIdx.x - j.block0;
        if (p.mode == 0) { if (j.dtype == TEDSPAD_F16) pack_fwd_tiles<F16>(p, tile, q, b, j.nblocks); else pack_fwd_tiles<BF16>(p, tile, q, b, j.nblocks); }
        else { if (j.dtype == TEDSPAD_F16) pack_dgrad_tiles<F16>(p, tile, q, b, j.nblocks); else pack_dgrad_tiles<BF16>(p, tile, q, b, j.nblocks); }
        return;
    }
    const long first = (long)((int)blockIdx.x - j.block0) * 256 + threadIdx.x, stride = (long)j.nblocks * 256;
    if (j.dtype == TEDSPAD_F16) pack_body<F16>(p, first, stride);
    else pack_body<BF16>(p, first, stride);
}

// one workgroup per job: eval-mode BatchNorm folded to (scale, shift) as bn_fold_kernel (train_ops.hip) does, written zero-padded to n
// floats and optionally to a second pair of vectors (a PackedConv's own padded copies); gamma == NULL: no BatchNorm, shift = conv_bias
__global__ __launch_bounds__(256) void fold_multi_kernel(const tedspad_fold_job *tab) {
    const tedspad_fold_job &j = tab[blockIdx.x];
    const int nmax = j.n > j.n2 ? j.n : j.n2;
    for (int c = threadIdx.x; c < nmax; c += 256) {
        float sc = 0.f, sh = 0.f;
        if (c < j.C) {
            if (j.gamma) {
                const double inv = (double)j.gamma[c] / sqrt((double)j.var[c] + j.eps);
                double b = (double)j.beta[c] - (double)j.mean[c] * inv;
                if (j.conv_bias) b += (double)j.conv_bias[c] * inv;
                sc = (float)inv; sh = (float)b;
            } else {
                sc = 1.f; sh = j.conv_bias ? j.conv_bias[c] : 0.f;
            }
        }
        if (c < j.n) { if (j.scale) j.scale[c] = sc; if (j.shift) j.shift[c] = sh; }
        if (c < j.n2) { if (j.scale2) j.scale2[c] = sc; if (j.shift2) j.shift2[c] = sh; }
    }
}

// packed fp32 weight-gradient accumulator [co_pad][kpad] (K ordered (dt,dh,dw,c) over cink kernel-form channels: tedspad_conv_wgrad)
// -> the parameter's gradient (co, ci, kt, kh, kw), written or accumulated, optionally scaled per output channel (frozen BatchNorm)
__global__ __launch_bounds__(256) void wgrad_unpack_multi_kernel(const tedspad_wgrad_unpack_job *tab, int njobs) {
    const int ji = find_job(tab, njobs, (int)blockIdx.x);
    const tedspad_wgrad_unpack_job &j = tab[ji];
    const int taps = j.kt * j.kh * j.kw;
    const long per_co = (long)j.ci * taps, total = (long)j.co * per_co;
    for (long e = (long)((int)blockIdx.x - j.block0) * 256 + threadIdx.x; e < total; e += (long)j.nblocks * 256) {
        const int n = (int)(e / per_co);
        const int r = (int)(e - (long)n * per_co);
        const int c = r / taps, tap = r - c * taps;
        float v = j.dw[(size_t)n * j.kpad + (size_t)tap * j.cink + c];
        if (j.row_scale) v *= j.row_scale[n];
        j.grad[e] = j.accumulate ? j.grad[e] + v : v;
    }
}

template <typename J>
int upload_and_plan(J *jobs, int njobs, void *table_dev, int upload, hipStream_t s, const char *who) {
    if (upload) {
        if (hipMemcpyAsync(table_dev, jobs, (size_t)njobs * sizeof(J), hipMemcpyHostToDevice, s) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: uploading the job table failed", who);
            return 0;
        }
    }
    return 1;
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

extern "C" int32_t tedspad_pack_multi(tedspad_pack_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream) {
    TS_REQUIRE(jobs && table_dev && njobs > 0, "tedspad_pack_multi: bad arguments");
    long blocks = 0;
    int lds_floats = 0;
    for (int i = 0; i < njobs; ++i) {
        tedspad_pack_job &j = jobs[i];
        TS_REQUIRE(j.w && j.out && j.co > 0 && j.ci > 0 && j.kt > 0 && j.kh > 0 && j.kw > 0 && j.cink % 8 == 0 && j.kpad % 8 == 0 && j.rows_pad >= j.rows &&
                       j.rows > 0 && (j.mode == 0 || j.mode == 1) && (j.dtype == TEDSPAD_F16 || j.dtype == TEDSPAD_BF16),
                   "tedspad_pack_multi: bad job");
        const long K = j.mode == 1 ? (long)j.geo[0] * j.geo[1] * j.geo[2] * ((j.co + 7) / 8 * 8) : (long)j.kt * j.kh * j.kwk * j.cink;
        TS_REQUIRE(K <= j.kpad && (j.mode == 0 || (j.geo[0] > 0 && j.geo[1] > 0 && j.geo[2] > 0)), "tedspad_pack_multi: kpad too small for a job's matrix");
        const long chunks = (long)j.rows_pad * (j.kpad / 8);
        long g = (chunks + 1023) / 1024;                       // pack_body: ~4 chunks of 8 weights per thread
        const PkPlan q = pk_plan(j.mode, j.pair_shift, j.kw, j.kwk, j.ci, j.kt * j.kh * j.kw, j.rows_pad, (j.co + 7) / 8 * 8);
        if (q.tiled) g = q.tiles;                              // one tile per workgroup (up to the cap)
        if (q.floats > lds_floats) lds_floats = q.floats;
        g = g < 1 ? 1 : (g > 256 ? 256 : g);
        j.block0 = (int32_t)blocks; j.nblocks = (int32_t)g;
        blocks += g;
    }
    TS_REQUIRE(blocks < (1L << 30), "tedspad_pack_multi: too many jobs");
    hipStream_t s = (hipStream_t)stream;
    if (!upload_and_plan(jobs, njobs, table_dev, upload, s, "tedspad_pack_multi")) return TEDSPAD_ELAUNCH;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds_floats * sizeof(float), s, (const tedspad_pack_job *)table_dev, njobs);
    return check_launch("tedspad_pack_multi");
}

extern "C" int32_t tedspad_fold_multi(tedspad_fold_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream) {
    TS_REQUIRE(jobs && table_dev && njobs > 0, "tedspad_fold_multi: bad arguments");
    for (int i = 0; i < njobs; ++i) {
        const tedspad_fold_job &j = jobs[i];
        TS_REQUIRE(j.C > 0 && j.n >= 0 && j.n2 >= 0 && (!j.gamma || (j.beta && j.mean && j.var)) && (j.n == 0 || j.n >= j.C) && (j.n2 == 0 || j.n2 >= j.C) &&
                       (j.n > 0 || j.n2 > 0),
                   "tedspad_fold_multi: bad job");
    }
    hipStream_t s = (hipStream_t)stream;
    if (!upload_and_plan(jobs, njobs, table_dev, upload, s, "tedspad_fold_multi")) return TEDSPAD_ELAUNCH;
    hipLaunchKernelGGL(fold_multi_kernel, dim3((unsigned)njobs), dim3(256), 0, s, (const tedspad_fold_job *)table_dev);
    return check_launch("tedspad_fold_multi");
}

extern "C" int32_t tedspad_wgrad_unpack_multi(tedspad_wgrad_unpack_job *jobs, int32_t njobs, void *table_dev, int32_t upload, void *stream) {
    TS_REQUIRE(jobs && table_dev && njobs > 0, "tedspad_wgrad_unpack_multi: bad arguments");
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        tedspad_wgrad_unpack_job &j = jobs[i];
        TS_REQUIRE(j.dw && j.grad && j.co > 0 && j.ci > 0 && j.kt > 0 && j.kh > 0 && j.kw > 0 && j.cink >= j.ci &&
                       (long)j.kt * j.kh * j.kw * j.cink <= j.kpad,
                   "tedspad_wgrad_unpack_multi: bad job");
        const long total = (long)j.co * j.ci * j.kt * j.kh * j.kw;
        long g = (total + 2047) / 2048;
        g = g < 1 ? 1 : (g > 256 ? 256 : g);
        j.block0 = (int32_t)blocks; j.nblocks = (int32_t)g;
        blocks += g;
    }
    TS_REQUIRE(blocks < (1L << 30), "tedspad_wgrad_unpack_multi: too many jobs");
    hipStream_t s = (hipStream_t)stream;
    if (!upload_and_plan(jobs, njobs, table_dev, upload, s, "tedspad_wgrad_unpack_multi")) return TEDSPAD_ELAUNCH;
    hipLaunchKernelGGL(wgrad_unpack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const tedspad_wgrad_unpack_job *)table_dev, njobs);
    return check_launch("tedspad_wgrad_unpack_multi");
}
