// elem16.h -- the two 16-bit storage/operand types of the network kernels: bf16 (ADAS_PREC_BF16) and IEEE half
// (ADAS_PREC_FP16, the precision the reference ships: demo.py:18-29 `*_fp16.trt`, coreEngine.py:168).
// Both feed the same-rate CDNA4 MFMA (v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate); half carries 11 significant bits
// instead of 8.  Kernels are templated on one of these tags and touch element bits only through it, so the two precisions are
// the same code with a different operand type.  Device pointers stay `uint16_t*` (raw bits) on the host side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adas {

typedef __attribute__((ext_vector_type(8))) __bf16 e_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 e_bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 e_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 e_f16x2;
typedef __attribute__((ext_vector_type(4))) float e_f32x4;
typedef __attribute__((ext_vector_type(2))) float e_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t e_u32x4;

// Storage element types for code that overloads / specialises on the pointer type: `uint16_t` holds bf16 bits (the historical
// spelling throughout the kernels), `f16s` holds IEEE-half bits -- a distinct type so the two resolve differently.
struct f16s {
    uint16_t v;
};

struct Bf16 {
    typedef uint16_t storage;
    static constexpr bool kHalf = false;
    static constexpr uint32_t kNegInf2 = 0xff80ff80u;  // two -inf elements (max-pool padding)
    typedef e_bf16x8 vec8;
    // D = A(16x32) * B(32x16) + C, operands as 8 packed elements per lane
    static __device__ __forceinline__ e_f32x4 mfma(vec8 a, vec8 b, e_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ e_f32x4 mfma(e_u32x4 a, e_u32x4 b, e_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void enter() {}   // bf16 has fp32's exponent range: nothing to saturate
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {  // v_cvt_pk_bf16_f32, round to nearest even
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(e_f32x2{a, b}, e_bf16x2));
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ float to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return (uint16_t)(pack2(f, 0.f) & 0xffffu); }
    static __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) {  // per-element max (exact: element <-> f32 is lossless)
        return pack2(fmaxf(lo(a), lo(b)), fmaxf(hi(a), hi(b)));
    }
    // round to nearest even on the bits (finite inputs) -- what pack2 computes, spelled out for host-side weight packing
    static inline uint16_t host_from_f32(float f) {
        uint32_t u;
        __builtin_memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
};

struct Fp16 {
    typedef f16s storage;
    static constexpr bool kHalf = true;
    static constexpr uint32_t kNegInf2 = 0xfc00fc00u;
    typedef e_f16x8 vec8;
    static __device__ __forceinline__ e_f32x4 mfma(vec8 a, vec8 b, e_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ e_f32x4 mfma(e_u32x4 a, e_u32x4 b, e_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c, 0, 0, 0);
    }
    // v_cvt_f16_f32 does not saturate by default: a value past the half range would become inf and the next layer NaN.  Every
    // kernel that stores halves calls enter() first: MODE.FP16_OVFL = 1 makes an overflowed half result clamp to +-65504 (true
    // infinities, e.g. max-pool padding, stay infinite) -- one scalar instruction per wave instead of a v_med3 per stored element
    // (the per-element clamp measured -2 % end to end); results below 65504 are bit-identical.
    static __device__ __forceinline__ void enter() { __builtin_amdgcn_s_setreg((unsigned short)(1 | (23 << 6) | (0 << 11)), 1u); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {  // two v_cvt_f16_f32 (round to nearest even) + v_pack_b32_f16
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(e_f32x2{a, b}, e_f16x2));
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(e_f16x2, u)[0]; }
    static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(e_f16x2, u)[1]; }
    static __device__ __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) {
        return pack2(fmaxf(lo(a), lo(b)), fmaxf(hi(a), hi(b)));
    }
    static inline uint16_t host_from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
};

// Run `fn(tag)` with the element tag of a 16-bit engine precision (PREC_FP16 -> Fp16, otherwise Bf16).
#define ADAS_DISPATCH_E16(is_half, E, ...) \
    do {                                   \
        if (is_half) {                     \
            using E = ::adas::Fp16;        \
            __VA_ARGS__;                   \
        } else {                           \
            using E = ::adas::Bf16;        \
            __VA_ARGS__;                   \
        }                                  \
    } while (0)


// Contiguous global -> LDS copy of n16 16-byte words by THREADS threads with U loads in flight per thread.  The obvious
// `for (i = tid; i < n; i += THREADS) lds[i] = src[i];` compiles to load / s_waitcnt vmcnt(0) / ds_write per trip (hipcc neither unrolls
// nor pipelines a 16-byte copy loop with a runtime trip count): one exposed L2 round trip per 16 bytes per thread, 24 of them in front of
// the first MFMA of a 384 -> 256 pointwise conv.
template <int THREADS, int U = 8>
__device__ __forceinline__ void stage_lds16(void* lds_dst, const void* src, int n16, int tid) {
    typedef __attribute__((ext_vector_type(4))) uint32_t sl_u32x4;
    const sl_u32x4* s = reinterpret_cast<const sl_u32x4*>(src);
    sl_u32x4* d = reinterpret_cast<sl_u32x4*>(lds_dst);
    for (int i0 = 0; i0 < n16; i0 += THREADS * U) {
        sl_u32x4 t[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i = i0 + j * THREADS + tid;
            t[j] = s[i < n16 ? i : 0];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i = i0 + j * THREADS + tid;
            if (i < n16) d[i] = t[j];
        }
    }
}

}  // namespace adas
