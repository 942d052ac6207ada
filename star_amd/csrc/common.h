// Shared helpers for the gfx950 kernels of libstar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/star_hip.h"

int star_set_error(int code, const char* fmt, ...);
int star_num_cus();   // compute units of the current device (cached)
struct StarGemm;
int star_gemm_rows_try(const StarGemm* g, hipStream_t s);   // gemm_ln.hip: whole-row tiles for a plain narrow Linear; 0 = launched

#define STAR_CHECK(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return star_set_error(-1, __VA_ARGS__);   \
    } while (0)

#define STAR_LAUNCH_CHECK(name)                                                             \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) return star_set_error(-2, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) short star_short4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN stays NaN: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}

template <typename T> struct Act;
template <> struct Act<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct Act<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// 4-wide vector access in the activation type
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = pack_bf16(v[0], v[1]);
    t.y = pack_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}
// 16-wide runs (16-byte vector accesses)
__device__ __forceinline__ void ld16(const float* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(p)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}
__device__ __forceinline__ void ld16(const bf16_t* p, float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 t = reinterpret_cast<const uint4*>(p)[i];
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[8 * i + 2 * k] = __uint_as_float(w[k] << 16);
            v[8 * i + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ void st16(float* p, const float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
__device__ __forceinline__ void st16(bf16_t* p, const float (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        reinterpret_cast<uint4*>(p)[i] = make_uint4(pack_bf16(v[8 * i], v[8 * i + 1]), pack_bf16(v[8 * i + 2], v[8 * i + 3]),
                                                    pack_bf16(v[8 * i + 4], v[8 * i + 5]), pack_bf16(v[8 * i + 6], v[8 * i + 7]));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Transposed LDS reads as inline asm: the compiler treats the builtin as aliasing every LDS-DMA in flight and
// puts a full vmcnt(0) in front of it, which would serialise the ring.  The asm form is invisible to its wait
// insertion, so the matching lgkmcnt wait is explicit (tr_wait) and carries the registers as operands to pin
// the order.
__device__ __forceinline__ unsigned lds_u32(const unsigned char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
template <int OFF> __device__ __forceinline__ star_short4_t tr16(unsigned addr) {
    star_short4_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8, each with its own L2).  This maps
// the launch index to a work index such that every XCD walks a CONTIGUOUS range of work items (bijective).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the CURRENT device: a "done" flag has to be kept per device ordinal.
// Returns true the first time it sees the current device with this mask (a race between two threads sets the attribute twice: harmless).
static inline bool star_once_per_device(unsigned long long& mask) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (mask & bit) return false;
    mask |= bit;
    return true;
}
