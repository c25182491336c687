// Shared helpers for the gfx950 kernels of libtedspad_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "tedspad_hip.h"

namespace tedspad {

void set_error(const char *fmt, ...);

#define TS_REQUIRE(cond, ...)                     \
    do {                                          \
        if (!(cond)) {                            \
            tedspad::set_error(__VA_ARGS__);      \
            return TEDSPAD_EINVAL;                \
        }                                         \
    } while (0)

inline int32_t check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return TEDSPAD_ELAUNCH;
    }
    return TEDSPAD_OK;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-bit storage types. Both run the same MFMA rate on CDNA4; f16 carries 3 more mantissa
// bits, which is what the 1e-3 feature-parity gate needs (DESIGN.md "precision").
struct F16 {
    static constexpr int kDtype = TEDSPAD_F16;
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {   // 16x16x32: same FLOP per cycle, the chip holds a higher clock on it
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        f = __builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f);  // saturate instead of inf
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
    // the same with the clamp at +-lim: 65504 (inference: saturate) or +inf (training path: an overflow stays visible as inf / NaN in the
    // gradients, where the loss scale's non-finite check finds it, as in the reference's fp16-autocast run)
    static __device__ __forceinline__ uint16_t from_f32_lim(float f, float lim) {
        f = __builtin_fminf(__builtin_fmaxf(f, -lim), lim);
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
    static __device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {   // element-wise max of two packed pairs
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b)));
    }
};

struct BF16 {
    static constexpr int kDtype = TEDSPAD_BF16;
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
    static __device__ __forceinline__ uint16_t from_f32_lim(float f, float) { return from_f32(f); }   // bf16 has fp32's range: never clamped
    static __device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
        const float lo = __builtin_fmaxf(__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, b << 16));
        const float hi = __builtin_fmaxf(__builtin_bit_cast(float, a & 0xffff0000u), __builtin_bit_cast(float, b & 0xffff0000u));
        return (__builtin_bit_cast(uint32_t, lo) >> 16) | (__builtin_bit_cast(uint32_t, hi) & 0xffff0000u);
    }
};

// One 16-byte global load from a per-lane address the compiler cannot see through. A load from `valid ? tensor + offset : zero_page`, or `q = 0; if (valid) q = *ptr`, is otherwise
// emitted as masked loads under their own exec masks with an s_waitcnt vmcnt(0) behind each (two writers of one register): a thread's N loads become N round trips. With the address
// made opaque, N of these in a row are N back-to-back global_load_dwordx4 (round 6: the stem's clip loader, the residual / mask pieces of the conv epilogues).
__device__ __forceinline__ uint4 load16_opaque(const void *ptr) {
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    unsigned long a = reinterpret_cast<unsigned long>(ptr);
    asm volatile("" : "+v"(a));
    const u32x4g v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4g *>(a);       // address space 1: a global_load, not a flat one
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T>
__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = T::to_f32((uint16_t)(w[i] & 0xffffu));
        f[2 * i + 1] = T::to_f32((uint16_t)(w[i] >> 16));
    }
}

template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)T::from_f32(f[2 * i]) | ((uint32_t)T::from_f32(f[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T>
__device__ __forceinline__ uint4 pack8_lim(const float (&f)[8], float lim) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)T::from_f32_lim(f[2 * i], lim) | ((uint32_t)T::from_f32_lim(f[2 * i + 1], lim) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T>
__device__ __forceinline__ uint4 pk_max8(uint4 a, uint4 b) {
    return make_uint4(T::pk_max(a.x, b.x), T::pk_max(a.y, b.y), T::pk_max(a.z, b.z), T::pk_max(a.w, b.w));
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst + lane*16].
// Written as inline asm on purpose: hipcc treats the builtin form as an LDS store it must
// drain (s_waitcnt vmcnt(0)) before ANY later ds_read of the same array, which serialises the
// ring. In asm form the compiler does not count it; the kernel waits with counted vmcnt itself.
// M0 (the DMA's LDS base) is compiler-reserved: saved/restored inside the same statement.
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the same for a read-once stream (nt: the lines are not kept in L2, where the weights and halos every workgroup re-reads live)
__device__ __forceinline__ void lds_dma16_nt(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2). Remap the block id so that every
// XCD walks a CONTIGUOUS range of logical tiles: the N tiles of one pixel tile, and pixel tiles that share halo
// rows, then hit the same L2 instead of re-reading HBM (profiles/r01: 272 MB/clip of traffic vs 129 MB minimal).
// Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace tedspad
