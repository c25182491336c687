// Bottleneck tail for gfx950: conv2 (1 x 3 x 3, stride 1, 64 -> 64) + bn2 + ReLU -> conv3 (1 x 1 x 1, 64 -> 256) + bn3 (+ residual | +
// the downsample branch) + ReLU of a layer1 bottleneck (aux_code/models/large_i3d.py:49-54,69-84) in ONE launch. The 64-channel tensor
// between the two convolutions never leaves the registers.
//
// Why: layer1 is 31 % of an I3Res50 forward and HBM-bound (profiles/r02_bench_cfg2_kernels_1stream.md: the 64 -> 256 pointwise convs
// run at 5.4 TB/s of their minimum bytes, conv2 at 32 % MFMA utilisation beside them). Separately the pair moves 1408 B per pixel
// (conv2 reads 128 + writes 128, conv3 reads 128 + 512 residual + writes 512); fused it moves 1152, and -- with two workgroups per CU --
// one workgroup's MFMA-bound conv2 loop runs under the other's memory-bound conv3 epilogue.
//
// Stage A = conv_flat_kernel (conv_flat.hip): a tile of 256 consecutive output pixels, its input halo one contiguous run in LDS, the
// [64][64] weight tile of a tap streaming through a 3-slot ring; a wave ends with D[co][px] (64 co x 64 px) in 64 accumulator registers.
// Stage B: `D = W2 . X` has its output channel on the accumulator ROWS, so the next product `Y3 = W3 . relu(bn2(D))` sums over rows and
// takes the tile as its MFMA B operand WITHOUT any lane movement (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's
// operand"): registers 8s .. 8s+7 of a 32 x 32 tile, packed to 16 bits, are the fragment of k-step s, in the k order
//     element j of lane half h  <->  row 16 s + 8 (j >> 2) + 4 h + (j & 3)
// which the host bakes into the column order of the conv3 weight image (`w3p`, tedspad_bneck_tail_fwd). The 256 x 64 (x 2 with the
// downsample branch) weight image is copied into the LDS that stage A has finished with; a wave then walks the 256 output channels
// in 4 steps of 64: 16 MFMAs (32 with the second source, whose pixel fragments come straight from global memory), bn3 scale / shift,
// the residual (whole rows by LDS-DMA into a wave-private image, read in the store layout and brought into the accumulator layout by
// v_permlane32_swap, which is its own inverse), ReLU, and the results leave through the same image as whole 128-byte rows.
#include "conv_common.h"
#include <type_traits>

namespace tedspad {
namespace {

__device__ uint4 g_zero16b;
__device__ uint4 g_sink16b[64];      // where the output rows past M go (never read)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BT_BM = 256;
constexpr int BT_WSTAGE = 64 * BK * 2;

__device__ __forceinline__ void gstore16(void *dst, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");   // s_nop 1: the store reads its data registers late, hipcc pads nothing after an asm statement; nt: written once, read by the next launch from HBM anyway
}

// bn3 (+ residual) + ReLU + saturation of two neighbouring output channels -> one packed pair. `b0`, `b1` carry the shift (and the second branch's
// product, when there is one); `res` = the residual pair (16-bit halves), `lo` = 0 with ReLU, else the most negative finite value.
// F16: 3.5 vector instructions per value -- v_fma_mix_f32 adds the f16 half to the shift without a conversion, one fma, one v_med3_f32 (ReLU and
// saturation together), v_cvt_pk_f16_f32 rounds both (gfx950) -- where the generic form below took 8.5 with `relu` / `res != NULL` as run-time selects
// (the tails issue vector instructions 45 % of their wave-cycles: profiles/r03_bench_cfg2_mfma_util.md).
template <typename T, bool RES>
__device__ __forceinline__ unsigned tail_pair(float a0, float a1, float s0, float s1, float b0, float b1, unsigned res, float lo) {
    if constexpr (T::kDtype == TEDSPAD_F16) {
        unsigned pk;
        if (RES) {
            float t0, t1;
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(res), "v"(b0));
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(res), "v"(b1));
            b0 = t0;
            b1 = t1;
        }
        const float v0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a0, s0, b0), lo, 65504.f);
        const float v1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(a1, s1, b1), lo, 65504.f);
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v0), "v"(v1));
        return pk;
    } else {
        float o0 = __builtin_fmaf(a0, s0, b0), o1 = __builtin_fmaf(a1, s1, b1);
        if (RES) {
            o0 += T::to_f32((uint16_t)(res & 0xffffu));
            o1 += T::to_f32((uint16_t)(res >> 16));
        }
        return (unsigned)T::from_f32(__builtin_fmaxf(o0, lo)) | ((unsigned)T::from_f32(__builtin_fmaxf(o1, lo)) << 16);
    }
}
template <typename T>
__device__ __forceinline__ float tail_lo(int relu) { return relu ? 0.f : (T::kDtype == TEDSPAD_F16 ? -65504.f : -3.3e38f); }

#ifdef TEDSPAD_BT_ABLATE      // diagnostic builds (wrong results): bit 1 = no residual loads, 2 = no result stores, 8 = every conv2 weight stage re-reads tap 0's 8 KB (always an L2 / L1 hit), 16 = no conv3 weight image load, 32 / 64 = halo source chunks unswizzled / swizzled by whole quads only
#define BT_ABL(bit) ((TEDSPAD_BT_ABLATE) & (bit))
#else
#define BT_ABL(bit) 0
#endif
#ifdef TEDSPAD_BT_STAGE_STAMPS
// Diagnostic build only (scripts/bneck_tail_cycles.py): s_memtime at the stage boundaries of every workgroup (wave 0) of the 64-channel tail.
__device__ long long *g_bt_stage_ts;
#define BT_STAGE_DECL() long long st_[13], sr_[2]; for (int i_ = 0; i_ < 13; ++i_) st_[i_] = 0; st_[0] = __builtin_amdgcn_s_memtime(); sr_[0] = __builtin_amdgcn_s_memrealtime()
#define BT_STAGE(i) st_[i] = __builtin_amdgcn_s_memtime()
#define BT_STAGE_FLUSH() { sr_[1] = __builtin_amdgcn_s_memrealtime(); if (tid == 0 && g_bt_stage_ts) { long long *o_ = g_bt_stage_ts + (size_t)blockIdx.x * 16; for (int i_ = 0; i_ < 13; ++i_) o_[i_] = st_[i_]; o_[14] = sr_[0]; o_[15] = sr_[1]; } }
#else
#define BT_STAGE_DECL()
#define BT_STAGE(i)
#define BT_STAGE_FLUSH()
#endif

struct BneckKP {
    const uint16_t *x;          // conv2 input (n,t,h,w,64+) 16-bit, pixel stride ldx
    const uint16_t *w2;         // conv2 weights, packed [>=64][Kpad] (K = (dh, dw, ci))
    const float *scale2, *shift2;
    const uint16_t *w3p;        // [cout3][KB*64]: columns 0..63 = conv3 weights in accumulator k order, 64..127 = downsample weights (natural order)
    const float *scale3, *shift3, *scaled;    // shift3 already holds b3 (+ bd)
    const uint16_t *res;        // residual (n,t,h,w,cout3) or NULL
    const uint16_t *x2;         // second source of the downsample branch (same pixel grid, 64 channels) or NULL
    uint16_t *y;
    int M, Kpad, W, H, kh, kw, ph, pw, ldx, ldres, ldx2, ldy, cout3, relu;
    int R, NP, ntaps;           // flat-halo geometry (conv_flat.hip)
    int HW, tpf;                // POOLT: pixels / tiles per frame
    unsigned winv, hinv;        // ceil(2^20 / W), ceil(2^20 / H): (t * inv) >> 20 == t / W for the small t the kernels divide (t < W + 256; t * W < 2^20)
};

// Residual rows in / result rows out go through wave-private LDS images (whole 128-byte lines per access; the 16-byte loads / stores straight in the
// accumulator layout of the first version touched 32 lines per instruction and were removed in round 5).
// POOLT (plain block): MaxPool3d((2,1,1), stride (2,1,1)) of the block's output fused (large_i3d.py:139 after layer1): a workgroup
// owns 256 pixels of an even frame AND the same pixels of the next frame: stage A runs twice (the second halo lands where the first
// was), both 64-channel tiles stay in registers as conv3 operands, stage B computes every 64-channel step for both frames and stores
// their maximum: the 256-channel tensor is written once, pooled (half the bytes of the unfused conv3 + pool pair's traffic again).
template <typename T, bool DUAL, bool POOLT>
__global__ __launch_bounds__(256, 2) void conv_bneck_tail_kernel(const BneckKP p) {
    static_assert(!POOLT || !DUAL, "the pooled variant is the plain block");
    constexpr int NT = 256, WS = 4, KB = DUAL ? 2 : 1, NF = POOLT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BT_STAGE_DECL();
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int q0f[NF];                                        // first pixel of the tile (in each of the two frames)
    int lim;                                            // valid pixels of the tile
    size_t obase;                                       // first output row
    if (POOLT) {
        const int fp = tile / p.tpf, j = tile - fp * p.tpf;       // frame pair (n, t / 2), tile of the frame
        q0f[0] = 2 * fp * p.HW + j * BT_BM;
        q0f[NF - 1] = q0f[0] + p.HW;
        lim = min(BT_BM, p.HW - j * BT_BM);
        obase = (size_t)fp * p.HW + j * BT_BM;
    } else {
        q0f[0] = tile * BT_BM;
        lim = min(BT_BM, p.M - q0f[0]);
        obase = (size_t)q0f[0];
    }
    const int S = (p.NP + 1) * 8;                       // 16-byte slots: the halo + one zero position
    const int Sr = (S + 63) / 64 * 64;
    const int halo_bytes = Sr * 16;
    unsigned char *wring = dsm + halo_bytes;            // [WS][64][64] 16-bit
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16b);

    // ================================ stage A: conv2 on the flat halo (conv_flat_kernel) ================================
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w2 + (size_t)rsub * p.Kpad + kc * 8;
    auto issue_w = [&](int kt, int slot) {
        const unsigned dst = lds0 + halo_bytes + slot * BT_WSTAGE + wave * 8 * (BK * 2);
        lds_dma16(wsrc + kt * BK, dst);
        lds_dma16(wsrc + (size_t)32 * p.Kpad + kt * BK, dst + 32 * (BK * 2));
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    const int NH = (Sr + NT - 1) / NT;
    // ---- stage B's LDS images and loads, set up here because they are ISSUED right after stage A's last barrier (under the bn2 / ReLU arithmetic) -------------
    // conv3 weight image: tiles of [64 co'][64 k] (the swizzled image of every weight tile of stage A), HG output-channel groups of 64 resident at a time: all of
    // them for the plain block (32 KB), two at a time with the second source (2 x 2 x 8 KB), so that weight image + BatchNorm vectors + the waves' row images
    // stay below the 80 KB that let two workgroups share a CU
    const int n3 = p.cout3 / 64;                           // 64-channel groups of the output
    const int HG = DUAL ? (n3 < 2 ? n3 : 2) : n3;
    auto load_w3 = [&](int g0) {                           // groups g0 .. g0 + HG - 1: slot kb * HG + (g - g0)
        for (int i = wave; i < KB * HG * 8; i += 4) {      // 8 wave-instructions (8 rows x 128 B) per tile
            const int tl = i >> 3, sub = i & 7;
            const int kb = tl / HG, g = g0 + tl - kb * HG;
            if (g >= n3) continue;
            const int row = sub * 8 + (lane >> 3);         // row of the tile
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            lds_dma16(p.w3p + (size_t)(g * 64 + row) * (KB * 64) + kb * 64 + ch * 8, lds0 + tl * BT_WSTAGE + sub * 1024);
        }
    };
    // BatchNorm vectors [3][CP] fp32 (scale3, shift3, scale of the second branch), CP = cout3 rounded up to 256: one 1 KB LDS-DMA instruction per 256 channels
    // (every load of this phase is an asm DMA, so the counted wait below can leave the residual rows in flight: hipcc's own wait for a plain load would drain them)
    const int CP = (p.cout3 + 255) & ~255;
    const int bnv_off = KB * HG * BT_WSTAGE;
    float *bnv = reinterpret_cast<float *>(dsm + bnv_off);
    auto load_bnv = [&]() {
        const int nck = CP >> 8;
        for (int i = wave; i < (DUAL ? 3 : 2) * nck; i += 4) {
            const int v = i / nck, ck = i - v * nck;
            const float *vec = v == 0 ? p.scale3 : v == 1 ? p.shift3 : p.scaled;
            const int c = min(ck * 256 + lane * 4, p.cout3 - 4);
            lds_dma16(vec + c, lds0 + bnv_off + (v * CP + ck * 256) * 4);
        }
    };
    const int wb_off = bnv_off + 3 * CP * 4;
    unsigned char *wbuf = dsm + wb_off + wave * 8192;
    const unsigned wbuf_lds = lds0 + wb_off + wave * 8192;
    const bool has_res = !DUAL && p.res != nullptr;
    const float lo = tail_lo<T>(p.relu);
    // row-layout role of this lane in instruction k: pixel row k*8 + (lane >> 3) of the wave, physical chunk lane & 7
    const int rrow = lane >> 3, rch = lane & 7;
    // row k*8 + rrow: chunk swizzle (row >> 1) & 7 = (rrow >> 1) ^ 4 (k & 1); the addresses are rebuilt from ONE base per call (an opaque
    // zero keeps hipcc from carrying 8 row pointers per frame across the loop -- the pooled variant spilled them, and every reload drained
    // the DMA queue)
    const int c0 = rch ^ (rrow >> 1);
    const int limw = lim - wave * 64;                   // valid rows of this wave
    auto issue_res = [&](int g, int f) {
        if (BT_ABL(1)) return;
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const uint16_t *base = p.res + (size_t)(q0f[f] + wave * 64 + rrow + opq) * p.ldres + 64 * g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint16_t *src = k * 8 + rrow < limw ? base + (size_t)(k * 8) * p.ldres + ((c0 ^ ((k & 1) << 2)) << 3) : zero;
            lds_dma16_nt(src, wbuf_lds + k * 1024);
        }
    };
    // bn2's scale / shift ([2][64] fp32) go through LDS too, past everything else: read as 4-channel vectors after stage A (64 scalar global loads per lane
    // sat between stage A and stage B before)
    const int main_end = halo_bytes + WS * BT_WSTAGE, tail_end = wb_off + 4 * 8192;
    float *bn2v = reinterpret_cast<float *>(dsm + (main_end > tail_end ? main_end : tail_end));
    f32x4 bn2reg = {0.f, 0.f, 0.f, 0.f};
    if (tid < 32) bn2reg = reinterpret_cast<const f32x4 *>(tid < 16 ? p.scale2 : p.shift2)[tid & 15];
    uint4 y2[NF][4][2];       // relu(bn2(conv2)) packed to 16 bits: fragment (a*2 + s) = rows 16 s .. 16 s + 15 of channel half a
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int q0 = q0f[f];
        if (f == 0) BT_STAGE(9);
        issue_w(0, 0);                                       // issue order w(0), halo, w(1): the counted waits below rely on it
        {
            // lane: 16-byte chunk cs of halo position pos0 + 32 i; its swizzle term ((pos >> 1) & 7) does not depend on i, the source advances 32 pixels per
            // instruction (one 64-bit add; the first version recomputed every address and took 4.4 k cycles to ISSUE the 12 instructions, scripts/bneck_tail_cycles.py)
            const int pos0 = tid >> 3, cs = tid & 7;
            const int qa = q0 - p.R + pos0;
            const uint16_t *src = p.x + (long)qa * p.ldx + ((BT_ABL(32) ? cs : BT_ABL(64) ? (cs ^ ((pos0 >> 1) & 4)) : (cs ^ ((pos0 >> 1) & 7))) << 3);
            const long step = (long)32 * p.ldx;
            for (int i = 0; i < NH; ++i) {
                if (i * NT + wave * 64 >= Sr) break;             // wave-uniform
                const bool ok = pos0 + 32 * i < p.NP && (unsigned)(qa + 32 * i) < (unsigned)p.M;
                if (BT_ABL(128)) {     // timing only: M0 left alone (every piece lands on the same 1 KB)
                    if (i == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + wave * 64 * 16) : "memory");
                    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(ok ? src : zero) : "memory");
                } else
                lds_dma16(ok ? src : zero, lds0 + (i * NT + wave * 64) * 16);
                src += step;
            }
        }
        if (p.ntaps > 1) issue_w(1, 1);
        if (p.ntaps > 2) issue_w(2, 2);
        if (f == 0) BT_STAGE(10);

        // Which taps of a pixel lie inside its frame: bit dh of rb / bit dw of cb (3 + 3 compares per pixel instead of a division, a modulo and kh x kw tests:
        // that prologue took 4.3 k cycles). The tile starts at row h0, column w0 of its frame (uniform); a lane's pixel is < 256 further on.
        int pj[2];
        unsigned rb[2], cb[2];
        {
            const int r0 = q0 / p.W, w0 = q0 - r0 * p.W, h0 = r0 % p.H;      // uniform
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int j = wave * 64 + b * 32 + l31;
                pj[b] = j;
                const unsigned t = (unsigned)(w0 + j);
                const unsigned dr = (t * p.winv) >> 20;
                const int w = (int)(t - dr * p.W);
                const unsigned hh = (unsigned)h0 + dr;
                const int h = (int)(hh - ((hh * p.hinv) >> 20) * p.H);
                unsigned r = 0, c = 0;
                for (int dh = 0; dh < p.kh; ++dh) r |= ((unsigned)(h + dh - p.ph) < (unsigned)p.H ? 1u : 0u) << dh;
                for (int dw = 0; dw < p.kw; ++dw) c |= ((unsigned)(w + dw - p.pw) < (unsigned)p.W ? 1u : 0u) << dw;
                rb[b] = q0 + j < p.M ? r : 0u;
                cb[b] = c;
            }
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        if (f == 0) BT_STAGE(11);
        if (f == 0 && tid < 32) *reinterpret_cast<f32x4 *>(bn2v + tid * 4) = bn2reg;
        if (p.ntaps > 2) wait_vmcnt<4>(); else if (p.ntaps > 1) wait_vmcnt<2>(); else wait_vmcnt<0>();   // halo + weight stage 0 of this wave have landed (stages 1, 2 may still fly)
        if (f == 0) BT_STAGE(12);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (f == 0) BT_STAGE(1);
        // The fragments of tap kt + 1 are REQUESTED during tap kt: k-step ks of the next tap right after the four MFMAs of k-step ks of this one, into the
        // registers those MFMAs have just read (a whole tap = 512 MFMA cycles between an LDS request and its use; the first version asked for a k-step's
        // fragments one k-step ahead, 128 cycles, and a tap took 1 500 - 1 800 cycles: scripts/bneck_tail_cycles.py). That needs weight stage kt + 1 landed when
        // tap kt starts, hence four ring slots with stages kt + 1, kt + 2 in flight; the DMA of stage kt + 3 is issued behind the first MFMAs of the tap, where
        // its issue cost hides.
        u32x4 fa[4][2], fw[4][2];
        auto tap_pos = [&](int dh_, int dw_, unsigned (&xo)[2], unsigned (&xs)[2]) {
            const int delta = dh_ * p.W + dw_;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pos = ((rb[b] >> dh_) & (cb[b] >> dw_) & 1u) ? pj[b] + delta : p.NP;
                xo[b] = (unsigned)pos * 128u;
                xs[b] = (unsigned)(pos >> 1) & 7u;
            }
        };
        // The requests are asm, not C++ loads: hipcc drains every outstanding LDS request at the loop header (s_waitcnt lgkmcnt(0)) when they are carried
        // around the loop, which would undo the pipeline; here a k-step waits for exactly its own four fragments (the 12 younger requests stay in flight).
        // The counted wait relies on LDS requests completing in order. A scalar load in flight (same counter, out of order) can only make the wait longer:
        // it counts as outstanding, so "at most 12 outstanding" still means that at most 12 LDS requests are, i.e. the four oldest have completed.
        auto read_frags = [&](int ks, int slot, const unsigned (&xo)[2], const unsigned (&xs)[2]) {
            const unsigned c = (unsigned)((ks << 1) | lh);
            const unsigned wt = lds0 + halo_bytes + slot * BT_WSTAGE + l31 * (BK * 2) + ((c ^ swz) << 4);
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][b]) : "v"(lds0 + xo[b] + ((c ^ xs[b]) << 4)));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fw[ks][0]) : "v"(wt));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fw[ks][1]) : "v"(wt));
        };
        static_assert(32 * BK * 2 == 4096, "second weight fragment: 32 rows further");
        int dh = 0, dw = 0;
        unsigned xoff[2], xswz[2];
        tap_pos(0, 0, xoff, xswz);
        asm volatile("; BT_COUNTED_LGKM_BEGIN" ::: "memory");      // (markers for the structure check of tests/test_host_logic.py)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) read_frags(ks, 0, xoff, xswz);
        // One tap; straight-line code (the counted DMA wait, the barrier, the DMA issue and the requests for the next tap are decided by the caller at compile time)
        auto tap = [&](int kt, auto waitn, auto issue, auto more) {
            if (++dw == p.kw) { dw = 0; ++dh; }          // (dh, dw) of tap kt + 1
            if constexpr (decltype(more)::value) tap_pos(dh, dw, xoff, xswz);
            if constexpr (decltype(waitn)::value >= 0) {
                wait_vmcnt<decltype(waitn)::value>();    // stage kt + 1 landed; stage kt + 2 (2 instructions) may stay in flight
                __builtin_amdgcn_s_barrier();            // ... of every wave; nobody reads the slot of stage kt - 1 any more
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // this k-step's four fragments have arrived: with the next tap's requests behind them 12 stay outstanding, on the last tap 4 (3 - ks)
                u32x4 x0 = fa[ks][0], x1 = fa[ks][1], w0 = fw[ks][0], w1 = fw[ks][1];      // (locals: the asm cannot tie elements of a captured array)
                if (decltype(more)::value || ks == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1));
                else if (ks == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1));
                else if (ks == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1));
                acc[0][0] = T::mfma(__builtin_bit_cast(uint4, w0), __builtin_bit_cast(uint4, x0), acc[0][0]);
                acc[0][1] = T::mfma(__builtin_bit_cast(uint4, w0), __builtin_bit_cast(uint4, x1), acc[0][1]);
                acc[1][0] = T::mfma(__builtin_bit_cast(uint4, w1), __builtin_bit_cast(uint4, x0), acc[1][0]);
                acc[1][1] = T::mfma(__builtin_bit_cast(uint4, w1), __builtin_bit_cast(uint4, x1), acc[1][1]);
                if constexpr (decltype(more)::value) read_frags(ks, (kt + 1) & 3, xoff, xswz);
                if constexpr (decltype(issue)::value) {
                    if (ks == 0) { if (!BT_ABL(8)) issue_w(kt + 3, (kt + 3) & 3); else issue_w(0, 0); }
                }
            }
        };
        using std::integral_constant;
        typedef integral_constant<bool, true> yes;
        typedef integral_constant<bool, false> no;
        int kt = 0;
        for (; kt + 3 < p.ntaps; ++kt) tap(kt, integral_constant<int, 2>{}, yes{}, yes{});
        if (kt + 2 < p.ntaps) { tap(kt, integral_constant<int, 2>{}, no{}, yes{}); ++kt; }
        if (kt + 1 < p.ntaps) { tap(kt, integral_constant<int, 0>{}, no{}, yes{}); ++kt; }
        tap(kt, integral_constant<int, -1>{}, no{}, no{});
        asm volatile("; BT_COUNTED_LGKM_END" ::: "memory");
        __syncthreads();          // every wave is done with the halo and the weight ring: the next frame's halo / the conv3 weight image lands there
        if (f == NF - 1) {
            // stage A's LDS is free: stage B's weight image, BatchNorm vectors and the first residual rows land under the arithmetic below
            BT_STAGE(2);
            if (!BT_ABL(16)) load_w3(0);
            load_bnv();
            if (has_res) issue_res(0, 0);
        }
        // ---- relu(bn2(.)) of the conv2 tile, packed to 16 bits ---------------------------------------------------------------------
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // one channel half at a time: its vectors are requested behind this statement and its fragments are complete at the one below (two volatile
            // statements keep their order); left alone hipcc requests both halves' vectors inside the last tap beside the full accumulator tile and spills
            if (!DUAL) asm volatile("" ::: "memory");          // (the two-source variant is spill-free as hipcc orders it, and not with this)
            float sc[16], sf[16];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int c = a * 32 + 8 * rq + 4 * lh;           // register 4 rq + i <-> channel c + i
                const f32x4 vs = *reinterpret_cast<const f32x4 *>(bn2v + c), vf = *reinterpret_cast<const f32x4 *>(bn2v + 64 + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) { sc[4 * rq + i] = vs[i]; sf[4 * rq + i] = vf[i]; }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaxf(acc[a][b][8 * s + j] * sc[8 * s + j] + sf[8 * s + j], 0.f);
                    y2[f][a * 2 + s][b] = pack8<T>(v);
                }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (!DUAL) asm volatile("" : "+v"(y2[f][a * 2 + s][b].x), "+v"(y2[f][a * 2 + s][b].y), "+v"(y2[f][a * 2 + s][b].z), "+v"(y2[f][a * 2 + s][b].w));
        }
    }

    // ================================ stage B: conv3 (+ downsample branch) on the register tile ================================
    // ---- this wave's pixels; the second source's pixel fragments (natural k order) straight from global memory --------------------------
    size_t mpx[2];
    bool inb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = wave * 64 + b * 32 + l31;
        inb[b] = j < lim;
        mpx[b] = (size_t)(inb[b] ? q0f[0] + j : 0);
    }
    uint4 xin[4][2];
    if (DUAL) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) xin[ks][b] = *reinterpret_cast<const uint4 *>(p.x2 + mpx[b] * p.ldx2 + (ks * 2 + lh) * 8);
    }
    if (has_res) wait_vmcnt<8>(); else wait_vmcnt<0>();      // weight image + BatchNorm vectors landed; the 8 residual rows of step 0 may still fly
    __syncthreads();           // ... of every wave
    BT_STAGE(3);

    // ---- 64 output channels per step. Every global access of the step moves whole 128-byte lines: the residual rows arrive by LDS-DMA
    // into a wave-private [64 px][64 ch] image (8 lanes per pixel row; the chunk swizzle of every other image here), the results leave
    // through the same image and are stored 8 lanes per row. (First version: 16-byte loads / stores straight from the accumulator layout,
    // one 32-byte piece per pixel and instruction -- 32 lines touched per instruction: 935 us for the plain block against 860 us unfused.)
    const int ng = p.cout3 / 64;
    for (int g = 0; g < ng; ++g) {
        if (g && g % HG == 0) {                  // the next HG groups of the weight image (second-source variant only)
            __syncthreads();
            load_w3(g);
            wait_vmcnt<0>();
            __syncthreads();
        }
        unsigned d[2][2][4][2];                  // [tile of the pair][pixel group][q][h]: packed results in the accumulator layout
        unsigned dk[POOLT ? 2 : 1][2][4][2];     // POOLT: the first frame's
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (has_res) {
                // the residual rows of this step landed; the previous 64-channel step's 8 stores (issued after the first frame's rows) stay in flight
                if (f == 0 && g > 0) wait_vmcnt<8>(); else wait_vmcnt<0>();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (POOLT) __builtin_amdgcn_sched_barrier(0);       // keep the two tiles' live ranges apart (the pooled variant is register-bound)
                const int t = 2 * g + tt;
                const unsigned char *Wt = dsm + (g % HG) * BT_WSTAGE + (tt * 32 + l31) * (BK * 2);
                f32x16 a3[2], ad[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { a3[b][r] = 0.f; ad[b][r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const unsigned c = (unsigned)((ks << 1) | lh);
                    const uint4 fw = *reinterpret_cast<const uint4 *>(Wt + ((c ^ swz) << 4));
#pragma unroll
                    for (int b = 0; b < 2; ++b) a3[b] = T::mfma(fw, y2[f][ks][b], a3[b]);
                    if (DUAL) {
                        const uint4 fd = *reinterpret_cast<const uint4 *>(Wt + HG * BT_WSTAGE + ((c ^ swz) << 4));
#pragma unroll
                        for (int b = 0; b < 2; ++b) ad[b] = T::mfma(fd, xin[ks][b], ad[b]);
                    }
                }
                // lane (l31, lh) holds channels 32 t + 8 q + 4 lh + {0..3}, q = 0..3
                f32x4 s3[4], b3[4], sd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q + 4 * lh;
                    s3[q] = *reinterpret_cast<const f32x4 *>(bnv + c);
                    b3[q] = *reinterpret_cast<const f32x4 *>(bnv + CP + c);
                    if (DUAL) sd[q] = *reinterpret_cast<const f32x4 *>(bnv + 2 * CP + c);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    unsigned rs[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
                    if (has_res) {
                        // residual chunks in the STORE layout (lane: channels 16 qq + 8 lh .. + 7 of the tile), swapped back into the accumulator layout
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const unsigned c8 = (unsigned)(4 * tt + 2 * qq + lh);
                            const uint4 L = *reinterpret_cast<const uint4 *>(wbuf + (b * 32 + l31) * 128 + ((c8 ^ swz) << 4));
                            auto s0 = __builtin_amdgcn_permlane32_swap(L.x, L.z, false, false);
                            auto s1 = __builtin_amdgcn_permlane32_swap(L.y, L.w, false, false);
                            rs[2 * qq][0] = s0[0]; rs[2 * qq + 1][0] = s0[1];
                            rs[2 * qq][1] = s1[0]; rs[2 * qq + 1][1] = s1[1];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int r = 4 * q + 2 * h;
                            if (DUAL) {
                                float v[2];
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const float o = a3[b][r + e] * s3[q][2 * h + e] + b3[q][2 * h + e] + ad[b][r + e] * sd[q][2 * h + e];
                                    v[e] = p.relu ? __builtin_fmaxf(o, 0.f) : o;
                                }
                                d[tt][b][q][h] = (unsigned)T::from_f32(v[0]) | ((unsigned)T::from_f32(v[1]) << 16);
                            } else {
                                // without a residual pointer rs is zero: the add stays (no select per value)
                                d[tt][b][q][h] = tail_pair<T, true>(a3[b][r], a3[b][r + 1], s3[q][2 * h], s3[q][2 * h + 1], b3[q][2 * h], b3[q][2 * h + 1], rs[q][h], lo);
                            }
                        }
#pragma unroll
                    for (int q = 0; q < 4; q += 2)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            auto sw = __builtin_amdgcn_permlane32_swap(d[tt][b][q][h], d[tt][b][q + 1][h], false, false);
                            d[tt][b][q][h] = sw[0];
                            d[tt][b][q + 1][h] = sw[1];
                        }
                }
            }
            if (POOLT && f == 0) {                   // keep the first frame's results; its residual rows are consumed: the second frame's may land
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int h = 0; h < 2; ++h) dk[tt][b][q][h] = d[tt][b][q][h];
                if (has_res) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_res(g, 1);
                }
            }
        }
        if (POOLT) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int h = 0; h < 2; ++h) d[tt][b][q][h] = T::pk_max(d[tt][b][q][h], dk[tt][b][q][h]);
        }
        // ---- results -> the wave's image (every residual read above is complete: its data was consumed) -> whole rows -> global ------------
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const unsigned c8 = (unsigned)(4 * tt + 2 * qq + lh);
                    *reinterpret_cast<uint4 *>(wbuf + (b * 32 + l31) * 128 + ((c8 ^ swz) << 4)) =
                        make_uint4(d[tt][b][2 * qq][0], d[tt][b][2 * qq][1], d[tt][b][2 * qq + 1][0], d[tt][b][2 * qq + 1][1]);
                }
        uint4 rowv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rowv[k] = *reinterpret_cast<const uint4 *>(wbuf + k * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the image is read back: the next step's residual rows may overwrite it
        if (has_res && g + 1 < ng) issue_res(g + 1, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = k * 8 + rrow;
            const int j = wave * 64 + row;
            uint16_t *dst = j < lim ? p.y + (obase + j) * p.ldy + 64 * g + ((rch ^ ((row >> 1) & 7)) << 3)
                                    : reinterpret_cast<uint16_t *>(g_sink16b) + lane * 8;      // rows past the tile: a scratch line, so that every wave issues 8 stores
            if (!BT_ABL(2)) gstore16(dst, u32x4{rowv[k].x, rowv[k].y, rowv[k].z, rowv[k].w});
        }
        if (g < 4) BT_STAGE(4 + g);
    }
    BT_STAGE(8);
    BT_STAGE_FLUSH();
}

// The same tail for 128 mid channels (layer2's plain bottlenecks: conv2 1 x 3 x 3 128 -> 128, conv3 128 -> 512 + residual; large_i3d.py:69-84).
// Stage A = the chunk-major flat tile of conv_patch.hip (tile_cfg 33): for each 64-channel chunk of the input the contiguous halo run is
// fetched once and serves all taps, the [128 co][64 k] weight tile of a (chunk, tap) streams through a two-slot ring; a wave ends with
// 128 co x 64 px in 128 accumulator registers. Stage B as above with eight k-steps (fragment a*2 + s = rows 16 s .. of channel quarter a) and
// the conv3 weight image ([64 co'][2 x 64 k] per output group) streamed two groups at a time. 72 KB of LDS in either stage: two workgroups
// per CU.
template <typename T>
__global__ __launch_bounds__(256, 2) void conv_bneck_tail128_kernel(const BneckKP p) {
    constexpr int NT = 256, WSTAGE = 128 * BK * 2, HG = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BT_STAGE_DECL();
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int q0 = tile * BT_BM;
    const int lim = min(BT_BM, p.M - q0);
    const int S = (p.NP + 1) * 8;
    const int Sr = (S + 63) / 64 * 64;
    const int halo_bytes = Sr * 16;
    unsigned char *wring = dsm + halo_bytes;            // [2][128][64] 16-bit
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)dsm;
    const uint16_t *zero = reinterpret_cast<const uint16_t *>(&g_zero16b);

    // ================================ stage A: conv2, chunk-major on the flat halo (conv_patch_kernel<T, 128, FLAT>) ================================
    const int rsub = wave * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const uint16_t *wsrc = p.w2 + (size_t)rsub * p.Kpad + kc * 8;
    auto issue_w = [&](int ch, int tap, int slot) {
        const unsigned dst = lds0 + halo_bytes + slot * WSTAGE + wave * 8 * (BK * 2);
        const uint16_t *src = wsrc + tap * 128 + ch * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16(src + (size_t)(j * 32) * p.Kpad, dst + j * 32 * (BK * 2));
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    const int NH = (Sr + NT - 1) / NT;
    // bn2's scale / shift -> LDS, past both stages' regions (read after stage A, several barriers later)
    float *bn2v;
    {
        const int main_end = halo_bytes + 2 * WSTAGE, tail_end = 2 * HG * BT_WSTAGE + 2 * p.cout3 * 4 + 4 * 8192;
        bn2v = reinterpret_cast<float *>(dsm + (main_end > tail_end ? main_end : tail_end));
        if (tid < 64) reinterpret_cast<f32x4 *>(bn2v)[tid] = reinterpret_cast<const f32x4 *>(tid < 32 ? p.scale2 : p.shift2)[tid & 31];
    }
    // which taps of a pixel lie inside its frame: bit dh of rb / bit dw of cb, from the tile's (uniform) first row / column -- no division per lane (see the
    // 64-channel kernel above)
    int pj[2];
    unsigned rb[2], cb[2];
    {
        const int r0 = q0 / p.W, w0 = q0 - r0 * p.W, h0 = r0 % p.H;      // uniform
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = wave * 64 + b * 32 + l31;
            pj[b] = j;
            const unsigned t = (unsigned)(w0 + j);
            const unsigned dr = (t * p.winv) >> 20;
            const int w = (int)(t - dr * p.W);
            const unsigned hh = (unsigned)h0 + dr;
            const int h = (int)(hh - ((hh * p.hinv) >> 20) * p.H);
            unsigned r = 0, c = 0;
            for (int dh = 0; dh < p.kh; ++dh) r |= ((unsigned)(h + dh - p.ph) < (unsigned)p.H ? 1u : 0u) << dh;
            for (int dw = 0; dw < p.kw; ++dw) c |= ((unsigned)(w + dw - p.pw) < (unsigned)p.W ? 1u : 0u) << dw;
            rb[b] = q0 + j < p.M ? r : 0u;
            cb[b] = c;
        }
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int ch = 0; ch < 2; ++ch) {
        if (ch) { BT_STAGE(9); __builtin_amdgcn_s_barrier(); }              // every wave has read the previous chunk's halo and weight slots
        asm volatile("" ::: "memory");
        issue_w(ch, 0, 0);                                 // issue order w(0), halo, w(1): the counted waits below rely on it
        for (int i = 0; i < NH; ++i) {
            if (i * NT + wave * 64 >= Sr) break;           // wave-uniform
            const int s = i * NT + tid;
            const int pos = s >> 3, cs = s & 7;
            const int q = q0 - p.R + pos;
            const bool ok = pos < p.NP && (unsigned)q < (unsigned)p.M;
            const uint16_t *src = ok ? p.x + (size_t)q * p.ldx + ch * 64 + ((cs ^ ((pos >> 1) & 7)) << 3) : zero;
            lds_dma16(src, lds0 + (i * NT + wave * 64) * 16);
        }
        if (p.ntaps > 1) issue_w(ch, 1, 1);
        int dh = 0, dw = 0;
        for (int kt = 0; kt < p.ntaps; ++kt) {
            const int delta = dh * p.W + dw;
            unsigned xoff[2], xswz[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pos = ((rb[b] >> dh) & (cb[b] >> dw) & 1u) ? pj[b] + delta : p.NP;
                xoff[b] = (unsigned)pos * 128u;
                xswz[b] = (unsigned)(pos >> 1) & 7u;
            }
            if (kt + 1 < p.ntaps) wait_vmcnt<4>(); else wait_vmcnt<0>();   // stage kt (and, on kt = 0, the halo) landed; stage kt+1 (4 instructions) may stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt == 0) { if (ch == 0) BT_STAGE(1); else BT_STAGE(10); }
            const uint16_t *Wt = reinterpret_cast<const uint16_t *>(wring + (kt & 1) * WSTAGE) + l31 * BK;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned c = (unsigned)((ks << 1) | lh);
                uint4 fa[2], fw[4];
#pragma unroll
                for (int b = 0; b < 2; ++b) fa[b] = *reinterpret_cast<const uint4 *>(dsm + xoff[b] + ((c ^ xswz[b]) << 4));
#pragma unroll
                for (int a = 0; a < 4; ++a) fw[a] = *reinterpret_cast<const uint4 *>(Wt + a * 32 * BK + ((c ^ swz) << 3));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = T::mfma(fw[a], fa[b], acc[a][b]);
            }
            if (kt + 1 < p.ntaps) {                        // two slots: stage kt+2 can only be issued once every wave has read stage kt
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 2 < p.ntaps) issue_w(ch, kt + 2, kt & 1);
            }
            if (++dw == p.kw) { dw = 0; ++dh; }
        }
    }
    __syncthreads();          // every wave is done with the halo and the weight ring
    BT_STAGE(2);

    // ================================ stage B: conv3 + residual on the register tile ================================
    const int n3 = p.cout3 / 64;
    auto load_w3 = [&](int g0) {                           // groups g0, g0 + 1: slot kb * HG + (g - g0), kb = the k half
        for (int i = wave; i < 2 * HG * 8; i += 4) {
            const int tl = i >> 3, sub = i & 7;
            const int kb = tl / HG, g = g0 + tl - kb * HG;
            if (g >= n3) continue;
            const int row = sub * 8 + (lane >> 3);
            const int chn = (lane & 7) ^ ((row >> 1) & 7);
            lds_dma16(p.w3p + (size_t)(g * 64 + row) * 128 + kb * 64 + chn * 8, lds0 + tl * BT_WSTAGE + sub * 1024);
        }
    };
    load_w3(0);
    float *bnv = reinterpret_cast<float *>(dsm + 2 * HG * BT_WSTAGE);     // [2][cout3]: scale3, shift3
    for (int i = tid; i < p.cout3; i += NT) {
        bnv[i] = p.scale3[i];
        bnv[p.cout3 + i] = p.shift3[i];
    }
    // relu(bn2(.)) of the conv2 tile, packed to 16 bits: fragment (a*2 + s) = rows 16 s .. 16 s + 15 of channel quarter a. bn2's scale / shift come from LDS
    // ([2][128] fp32 behind everything else, written before stage A) one channel quarter at a time: as 32 global loads per lane and quarter hipcc requested all
    // four quarters' vectors up front beside the full accumulator tile and spilled 24 registers here
    uint4 y2[8][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        asm volatile("" ::: "memory");
        float sc[16], sf[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int c = a * 32 + 8 * rq + 4 * lh;           // register 4 rq + i <-> channel c + i
            const f32x4 vs = *reinterpret_cast<const f32x4 *>(bn2v + c), vf = *reinterpret_cast<const f32x4 *>(bn2v + 128 + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sc[4 * rq + i] = vs[i]; sf[4 * rq + i] = vf[i]; }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaxf(acc[a][b][8 * s2 + j] * sc[8 * s2 + j] + sf[8 * s2 + j], 0.f);
                y2[a * 2 + s2][b] = pack8<T>(v);
            }
        // this quarter is packed before the next one's vectors are requested (two volatile statements keep their order; the reads sit behind the first)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                asm volatile("" : "+v"(y2[a * 2 + s2][b].x), "+v"(y2[a * 2 + s2][b].y), "+v"(y2[a * 2 + s2][b].z), "+v"(y2[a * 2 + s2][b].w));
    }
    wait_vmcnt<0>();
    __syncthreads();           // weight image + BN vectors visible
    BT_STAGE(3);

    unsigned char *wbuf = dsm + 2 * HG * BT_WSTAGE + 2 * p.cout3 * 4 + wave * 8192;
    const unsigned wbuf_lds = lds0 + 2 * HG * BT_WSTAGE + 2 * p.cout3 * 4 + wave * 8192;
    const bool has_res = p.res != nullptr;
    const float lo = tail_lo<T>(p.relu);
    const int rrow = lane >> 3, rch = lane & 7;
    const int c0 = rch ^ (rrow >> 1);
    const int limw = lim - wave * 64;
    auto issue_res = [&](int g) {
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const uint16_t *base = p.res + (size_t)(q0 + wave * 64 + rrow + opq) * p.ldres + 64 * g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint16_t *src = k * 8 + rrow < limw ? base + (size_t)(k * 8) * p.ldres + ((c0 ^ ((k & 1) << 2)) << 3) : zero;
            lds_dma16_nt(src, wbuf_lds + k * 1024);
        }
    };
    if (has_res) issue_res(0);
    for (int g = 0; g < n3; ++g) {
        if (g && g % HG == 0) {                  // the next two groups of the weight image
            __syncthreads();
            load_w3(g);
            wait_vmcnt<0>();
            __syncthreads();
        }
        unsigned d[2][2][4][2];
        if (has_res) {
            if (g == 0 || g % HG == 0) wait_vmcnt<0>(); else wait_vmcnt<8>();   // the residual rows of this step landed; the previous step's 8 stores stay in flight
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * g + tt;
            const unsigned char *Wt = dsm + (g % HG) * BT_WSTAGE + (tt * 32 + l31) * (BK * 2);
            f32x16 a3[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) a3[b][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const unsigned c = (unsigned)(((ks & 3) << 1) | lh);
                const uint4 fw = *reinterpret_cast<const uint4 *>(Wt + (ks >> 2) * HG * BT_WSTAGE + ((c ^ swz) << 4));
#pragma unroll
                for (int b = 0; b < 2; ++b) a3[b] = T::mfma(fw, y2[ks][b], a3[b]);
            }
            f32x4 s3[4], b3[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q + 4 * lh;
                s3[q] = *reinterpret_cast<const f32x4 *>(bnv + c);
                b3[q] = *reinterpret_cast<const f32x4 *>(bnv + p.cout3 + c);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                unsigned rs[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
                if (has_res) {
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const unsigned c8 = (unsigned)(4 * tt + 2 * qq + lh);
                        const uint4 L = *reinterpret_cast<const uint4 *>(wbuf + (b * 32 + l31) * 128 + ((c8 ^ swz) << 4));
                        auto s0 = __builtin_amdgcn_permlane32_swap(L.x, L.z, false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(L.y, L.w, false, false);
                        rs[2 * qq][0] = s0[0]; rs[2 * qq + 1][0] = s0[1];
                        rs[2 * qq][1] = s1[0]; rs[2 * qq + 1][1] = s1[1];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = 4 * q + 2 * h;
                        d[tt][b][q][h] = tail_pair<T, true>(a3[b][r], a3[b][r + 1], s3[q][2 * h], s3[q][2 * h + 1], b3[q][2 * h], b3[q][2 * h + 1], rs[q][h], lo);
                    }
#pragma unroll
                for (int q = 0; q < 4; q += 2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        auto sw = __builtin_amdgcn_permlane32_swap(d[tt][b][q][h], d[tt][b][q + 1][h], false, false);
                        d[tt][b][q][h] = sw[0];
                        d[tt][b][q + 1][h] = sw[1];
                    }
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const unsigned c8 = (unsigned)(4 * tt + 2 * qq + lh);
                    *reinterpret_cast<uint4 *>(wbuf + (b * 32 + l31) * 128 + ((c8 ^ swz) << 4)) =
                        make_uint4(d[tt][b][2 * qq][0], d[tt][b][2 * qq][1], d[tt][b][2 * qq + 1][0], d[tt][b][2 * qq + 1][1]);
                }
        uint4 rowv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rowv[k] = *reinterpret_cast<const uint4 *>(wbuf + k * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (has_res && g + 1 < n3) issue_res(g + 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = k * 8 + rrow;
            const int j = wave * 64 + row;
            uint16_t *dst = j < lim ? p.y + ((size_t)q0 + j) * p.ldy + 64 * g + ((rch ^ ((row >> 1) & 7)) << 3)
                                    : reinterpret_cast<uint16_t *>(g_sink16b) + lane * 8;
            gstore16(dst, u32x4{rowv[k].x, rowv[k].y, rowv[k].z, rowv[k].w});
        }
        if (g < 4) BT_STAGE(4 + g);
    }
    BT_STAGE(8);
    BT_STAGE_FLUSH();
}

template <typename T>
int32_t launch_bneck128(const BneckKP &p, hipStream_t s) {
    const int S = (p.NP + 1) * 8;
    const int main_bytes = (S + 63) / 64 * 64 * 16 + 2 * 128 * BK * 2;
    const int tail_bytes = 2 * 2 * BT_WSTAGE + 2 * p.cout3 * 4 + 4 * 8192;
    const int lds = (main_bytes > tail_bytes ? main_bytes : tail_bytes) + 1024;                          // + bn2's scale / shift
    if (lds > 160 * 1024) {
        set_error("tedspad_bneck_tail_fwd: halo does not fit LDS (%d bytes)", lds);
        return TEDSPAD_EINVAL;
    }
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_bneck_tail128_kernel<T>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_bneck_tail_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    hipLaunchKernelGGL(kfn, dim3((p.M + BT_BM - 1) / BT_BM), dim3(256), lds, s, p);
    return check_launch("tedspad_bneck_tail_fwd");
}

template <typename T, bool DUAL, bool POOLT = false>
int32_t launch_bneck(const BneckKP &p, hipStream_t s) {
    const int S = (p.NP + 1) * 8;
    const int main_bytes = (S + 63) / 64 * 64 * 16 + 4 * BT_WSTAGE;         // halo + the four-slot weight ring
    const int n3 = p.cout3 / 64, hg = DUAL ? (n3 < 2 ? n3 : 2) : n3;
    const int cp = (p.cout3 + 255) & ~255;
    const int tail_bytes = (DUAL ? 2 : 1) * hg * BT_WSTAGE + 3 * cp * 4 + 4 * 8192;     // weight image + BN vectors (+ one [64 px][64 ch] image per wave)
    const int lds = (main_bytes > tail_bytes ? main_bytes : tail_bytes) + 512;                          // + bn2's scale / shift
    if (lds > 160 * 1024) {
        set_error("tedspad_bneck_tail_fwd: halo / conv3 weight image does not fit LDS (%d bytes)", lds);
        return TEDSPAD_EINVAL;
    }
    static thread_local int attr_set[2] = {0, 0};
    auto kfn = conv_bneck_tail_kernel<T, DUAL, POOLT>;
    if (!attr_set[T::kDtype]) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("tedspad_bneck_tail_fwd: cannot raise the dynamic LDS limit");
            return TEDSPAD_ELAUNCH;
        }
        attr_set[T::kDtype] = 1;
    }
    const int tiles = POOLT ? (p.M / (2 * p.HW)) * p.tpf : (p.M + BT_BM - 1) / BT_BM;
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), lds, s, p);
    return check_launch("tedspad_bneck_tail_fwd");
}

}  // namespace
}  // namespace tedspad

using namespace tedspad;

extern "C" int32_t tedspad_bneck_tail_fwd(const tedspad_conv_desc *d2, const void *x, const void *w2_packed, const float *scale2, const float *shift2,
                                          const void *w3p, const float *scale3, const float *shift3, int32_t cout3, const void *residual,
                                          int32_t ldres, const void *x2, int32_t ldx2, const float *scale_d, void *y, int32_t ldy, int32_t relu,
                                          int32_t variant, void *stream) {
    TS_REQUIRE(d2 && x && w2_packed && scale2 && shift2 && w3p && scale3 && shift3 && y, "tedspad_bneck_tail_fwd: null pointer");
    const bool same = d2->to == d2->t && d2->ho == d2->h && d2->wo == d2->w && d2->pt == 0 && d2->ph < d2->kh && d2->pw < d2->kw;
    const bool c128 = d2->cin == 128 && d2->cout == 128;
    TS_REQUIRE(((d2->cin == 64 && d2->cout == 64) || c128) && d2->kt == 1 && d2->st == 1 && d2->sh == 1 && d2->sw == 1 && same && d2->kh * d2->kw >= 2 &&
                   d2->kh * d2->kw <= 32 && d2->ldx >= d2->cin && d2->ldx % 8 == 0,
               "tedspad_bneck_tail_fwd: conv2 must be a stride-1 'same' 1 x kh x kw conv with 64 -> 64 or 128 -> 128 channels");
    TS_REQUIRE(!c128 || (!x2 && !(variant & 4)), "tedspad_bneck_tail_fwd: the 128-channel form is the plain block (no second source, no temporal pool)");
    TS_REQUIRE(cout3 > 0 && cout3 % 64 == 0 && cout3 <= 512 && ldy >= cout3 && ldy % 8 == 0, "tedspad_bneck_tail_fwd: cout3 a multiple of 64 (<= 512), ldy >= cout3");
    TS_REQUIRE(!(residual && x2), "tedspad_bneck_tail_fwd: either a residual tensor or the second (downsample) source, not both");
    TS_REQUIRE(!residual || (ldres >= cout3 && ldres % 8 == 0), "tedspad_bneck_tail_fwd: bad ldres");
    TS_REQUIRE(!x2 || (scale_d && ldx2 >= 64 && ldx2 % 8 == 0), "tedspad_bneck_tail_fwd: second source needs its BatchNorm scale and ldx2 >= 64");
    TS_REQUIRE(((uintptr_t)x | (uintptr_t)w2_packed | (uintptr_t)w3p | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)x2 | (uintptr_t)scale3 | (uintptr_t)shift3 |
                (uintptr_t)scale2 | (uintptr_t)shift2 | (uintptr_t)scale_d) % 16 == 0,
               "tedspad_bneck_tail_fwd: pointers must be 16-byte aligned");
    TS_REQUIRE(d2->dtype == TEDSPAD_F16 || d2->dtype == TEDSPAD_BF16, "tedspad_bneck_tail_fwd: bad dtype");
    const long M = (long)d2->n * d2->t * d2->h * d2->w;
    TS_REQUIRE(M * (d2->ldx > ldy ? d2->ldx : ldy) < (1L << 31) && (!residual || M * ldres < (1L << 31)), "tedspad_bneck_tail_fwd: tensor too large for 32-bit offsets; split the batch");
    BneckKP p;
    p.x = (const uint16_t *)x; p.w2 = (const uint16_t *)w2_packed; p.scale2 = scale2; p.shift2 = shift2;
    p.w3p = (const uint16_t *)w3p; p.scale3 = scale3; p.shift3 = shift3; p.scaled = scale_d;
    p.res = (const uint16_t *)residual; p.x2 = (const uint16_t *)x2; p.y = (uint16_t *)y;
    p.M = (int)M; p.Kpad = tedspad_conv_kpad(d2); p.W = d2->w; p.H = d2->h; p.kh = d2->kh; p.kw = d2->kw; p.ph = d2->ph; p.pw = d2->pw;
    p.ldx = d2->ldx; p.ldres = ldres; p.ldx2 = ldx2; p.ldy = ldy; p.cout3 = cout3; p.relu = relu;
    // the prologue's (t * winv) >> 20 / (hh * hinv) >> 20 are exact divisions only while t * W < 2^20 (t < W + 256) and hh < 2 H < 2^11
    TS_REQUIRE((long)(d2->w + 256) * d2->w < (1L << 20) && d2->h < 1024, "tedspad_bneck_tail_fwd: frame too wide / tall for the prologue's multiply-shift divisions (w <= 903, h < 1024)");
    p.winv = ((1u << 20) + d2->w - 1) / d2->w; p.hinv = ((1u << 20) + d2->h - 1) / d2->h;
    p.R = d2->ph * d2->w + d2->pw; p.NP = BT_BM + (d2->kh - 1) * d2->w + (d2->kw - 1); p.ntaps = d2->kh * d2->kw;
    TS_REQUIRE(p.Kpad == p.ntaps * d2->cin, "tedspad_bneck_tail_fwd: unexpected K padding of the conv2 weights");
    if (c128) return d2->dtype == TEDSPAD_F16 ? launch_bneck128<F16>(p, (hipStream_t)stream) : launch_bneck128<BF16>(p, (hipStream_t)stream);
    p.HW = d2->h * d2->w; p.tpf = (p.HW + BT_BM - 1) / BT_BM;
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = d2->dtype == TEDSPAD_F16;
    if (variant & 4) {
        TS_REQUIRE(!x2 && d2->t % 2 == 0, "tedspad_bneck_tail_fwd: the temporal-pool variant takes the plain block (no second source) and an even frame count");
        return f16 ? launch_bneck<F16, false, true>(p, s) : launch_bneck<BF16, false, true>(p, s);
    }
    if (x2) return f16 ? launch_bneck<F16, true>(p, s) : launch_bneck<BF16, true>(p, s);
    return f16 ? launch_bneck<F16, false>(p, s) : launch_bneck<BF16, false>(p, s);
}

#ifdef TEDSPAD_BT_STAGE_STAMPS
extern "C" int32_t tedspad_debug_set_bt_stage_ts(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(tedspad::g_bt_stage_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
