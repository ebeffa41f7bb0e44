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

#if defined(__HIPCC__)
// Reciprocal of the 16-bit modes' activations (SiLU = v * rcp(1 + exp(-v)), sigmoid): the hardware reciprocal, v_rcp_f32 (1 ulp of fp32 --
// three orders of magnitude below the half / bf16 rounding of the stored result).  Until round 5 these epilogues called `__frcp_rn`, the
// CORRECTLY ROUNDED reciprocal, which hipcc expands into the IEEE division sequence (2 x v_div_scale, v_rcp, five fma / mul, v_div_fmas,
// v_div_fixup: 11 VALU instructions where one does) -- in every SiLU of every 16-bit conv kernel: the 59-98 VALU instructions per MFMA of
// the pointwise kernels (profiles/r04/pmc_halo_40x40.txt) and a third of conv_halo's tile time were mostly this.  The parity modes (fp32,
// fp16x3) keep fp32-class forms: expf and the IEEE division in the fp32 mode, x3_silu below (the same error class in 12 instructions) in fp16x3.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// SiLU for kernels templated on a storage type: the fast form where the result is stored in 16 bits, the exact form (expf, IEEE division) in
// the parity modes (float, x3s slots)
template <typename T>
__device__ __forceinline__ float silu_for(float v) {
    if constexpr (sizeof(T) == 2) return v * fast_rcp(1.0f + __expf(-v));
    else return v / (1.0f + expf(-v));
}
#endif

// ---- split precision (ADAS_PREC_FP16X3, kernels.h PREC_X3) ------------------------------------------------------------------
// A value x (f32) is carried as two halves: hi = half(x) and lo = half((x - hi) * 2^11), i.e. x = hi + lo * 2^-11 to 22 significant
// bits (x - hi is exact in f32; the scale keeps lo a NORMAL half whatever the magnitude of x).  A product of two such values is
//     a * w = a_hi * w_hi + 2^-11 (a_hi * w_lo + a_lo * w_hi) + O(2^-22),
// three f16 MFMAs into two fp32 accumulator sets (main, cross) that the epilogue combines: main + cross * 2^-11.
// Storage ("G8"): NHWC with 4 bytes per channel; every aligned group of 8 channels is 32 bytes = [8 hi halves][8 lo halves], so a
// lane's MFMA fragment (8 consecutive channels of one pixel) is one 16-byte load for hi and one for lo.  `x3s*` addresses channel
// SLOTS of 4 bytes (pointer arithmetic like float*); buffers are 256-byte aligned and pixel strides / view offsets multiples of 8
// channels, so the group a slot belongs to follows from its address.
struct x3s {
    uint32_t slot;
};
constexpr float kX3Up = 2048.0f, kX3Down = 1.0f / 2048.0f;

__host__ __device__ __forceinline__ void x3_split(float x, _Float16& h, _Float16& l) {
    _Float16 hh = (_Float16)x;
    if (!(x >= 6.103515625e-05f || x <= -6.103515625e-05f)) hh = (_Float16)0.0f;   // no subnormal hi: the value moves into lo
    h = hh;
    l = (_Float16)((x - (float)hh) * kX3Up);
}
__host__ __device__ __forceinline__ float x3_join(_Float16 h, _Float16 l) { return (float)h + (float)l * kX3Down; }

#if defined(__HIPCC__)
// SiLU of the split precision (round 6).  v / (1 + expf(-v)) as hipcc compiles it is 27 VALU instructions per value (ocml's expf with its
// range selects, then the IEEE division sequence: div_scale x2, rcp, five fma, div_fmas, div_fixup) -- 7,000 cycles per SIMD in the
// epilogue of every conv_h8x3 item of the detector, half the time of the fused C2f / stem launches.  The same quantity in 12:
//   e^x = exp2(t) (1 + r ln 2),  t = fl(x log2 e),  r = the product's rounding error (one fma) + x (log2 e - fl(log2 e)),
//   1 / d = v_rcp_f32 refined by one Newton step,
// x = min(-v, 88) so that e^x stays finite (v < -88: the result is -0 instead of -5e-37).  Against float64 over 3.5 M values (N(0, 3),
// U(-90, 90), N(0, 0.1)) with 1-ulp exp2 / rcp: max relative error 3.2e-7, mean 4.8e-8 -- NumPy's float32 evaluation of the
// reference expression (oracle/nets.py): 2.4e-7 / 3.4e-8 (tests/test_x3_silu_model.py restates this in NumPy).  -DADAS_X3_SILU_EXACT restores the 27-instruction form.
__device__ __forceinline__ float x3_silu(float v) {
#ifdef ADAS_X3_SILU_EXACT
    return v / (1.0f + expf(-v));
#else
    const float x = fminf(-v, 88.0f);
    const float t = x * 1.44269502162933349609375f;
    const float r = __builtin_fmaf(x, 1.44269502162933349609375f, -t) + x * 1.925963033500011e-08f;
    const float e = __builtin_amdgcn_exp2f(t);
    const float d = 1.0f + __builtin_fmaf(e * r, 0.693147180559945f, e);
    float q = __builtin_amdgcn_rcpf(d);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), q, q);
    return v * q;
#endif
}
// one slot (any channel): the group's base is the address rounded down to 32 bytes
__device__ __forceinline__ float x3_ld(const x3s* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const _Float16* g = reinterpret_cast<const _Float16*>(a & ~(uintptr_t)31);
    const int w = (int)((a & 31) >> 2);
    return x3_join(g[w], g[8 + w]);
}
__device__ __forceinline__ void x3_st(x3s* p, float v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    _Float16* g = reinterpret_cast<_Float16*>(a & ~(uintptr_t)31);
    const int w = (int)((a & 31) >> 2);
    _Float16 h, l;
    x3_split(v, h, l);
    g[w] = h;
    g[8 + w] = l;
}
// 4 consecutive channels starting at a multiple of 4 (the conv epilogues' lane ownership): 8 bytes of hi + 8 bytes of lo
__device__ __forceinline__ void x3_load4(const x3s* p, float v[4]) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const unsigned char* g = reinterpret_cast<const unsigned char*>(a & ~(uintptr_t)31) + ((a & 31) >> 1);
    const uint2 h = *reinterpret_cast<const uint2*>(g), l = *reinterpret_cast<const uint2*>(g + 16);
    const e_f16x2 h0 = __builtin_bit_cast(e_f16x2, h.x), h1 = __builtin_bit_cast(e_f16x2, h.y);
    const e_f16x2 l0 = __builtin_bit_cast(e_f16x2, l.x), l1 = __builtin_bit_cast(e_f16x2, l.y);
    v[0] = x3_join(h0[0], l0[0]); v[1] = x3_join(h0[1], l0[1]); v[2] = x3_join(h1[0], l1[0]); v[3] = x3_join(h1[1], l1[1]);
}
__device__ __forceinline__ void x3_store4(x3s* p, const float v[4]) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    unsigned char* g = reinterpret_cast<unsigned char*>(a & ~(uintptr_t)31) + ((a & 31) >> 1);
    e_f16x2 h0, h1, l0, l1;
    _Float16 h, l;
    x3_split(v[0], h, l); h0[0] = h; l0[0] = l;
    x3_split(v[1], h, l); h0[1] = h; l0[1] = l;
    x3_split(v[2], h, l); h1[0] = h; l1[0] = l;
    x3_split(v[3], h, l); h1[1] = h; l1[1] = l;
    *reinterpret_cast<uint2*>(g) = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
    *reinterpret_cast<uint2*>(g + 16) = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
}
// a whole group (p: 8-channel aligned slot)
__device__ __forceinline__ void x3_load8(const x3s* p, float v[8]) {
    const uint4 h = reinterpret_cast<const uint4*>(p)[0], l = reinterpret_cast<const uint4*>(p)[1];
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const e_f16x2 hh = __builtin_bit_cast(e_f16x2, hw[k]), ll = __builtin_bit_cast(e_f16x2, lw[k]);
        v[2 * k] = x3_join(hh[0], ll[0]);
        v[2 * k + 1] = x3_join(hh[1], ll[1]);
    }
}
__device__ __forceinline__ void x3_store8(x3s* p, const float v[8]) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e_f16x2 hh, ll;
        _Float16 h, l;
        x3_split(v[2 * k], h, l); hh[0] = h; ll[0] = l;
        x3_split(v[2 * k + 1], h, l); hh[1] = h; ll[1] = l;
        hw[k] = __builtin_bit_cast(uint32_t, hh);
        lw[k] = __builtin_bit_cast(uint32_t, ll);
    }
    reinterpret_cast<uint4*>(p)[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}
#endif

#if defined(__HIPCC__)
// 8 consecutive channels of one pixel as fp32, on any storage type (bf16 `uint16_t`, fp16 `f16s`, `float`, split `x3s`): the
// element-wise / depth-wise kernels run the same code in every precision
template <typename T> struct Vec8;
template <> struct Vec8<uint16_t> {
    static __device__ __forceinline__ void load(const uint16_t* p, float v[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = Bf16::lo(w[k]); v[2 * k + 1] = Bf16::hi(w[k]); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float v[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(Bf16::pack2(v[0], v[1]), Bf16::pack2(v[2], v[3]), Bf16::pack2(v[4], v[5]), Bf16::pack2(v[6], v[7]));
    }
};
template <> struct Vec8<f16s> {
    static __device__ __forceinline__ void load(const f16s* p, float v[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = Fp16::lo(w[k]); v[2 * k + 1] = Fp16::hi(w[k]); }
    }
    static __device__ __forceinline__ void store(f16s* p, const float v[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(Fp16::pack2(v[0], v[1]), Fp16::pack2(v[2], v[3]), Fp16::pack2(v[4], v[5]), Fp16::pack2(v[6], v[7]));
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float v[8]) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float v[8]) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};

template <> struct Vec8<x3s> {   // split precision: one G8 group (elem16.h)
    static __device__ __forceinline__ void load(const x3s* p, float v[8]) { x3_load8(p, v); }
    static __device__ __forceinline__ void store(x3s* p, const float v[8]) { x3_store8(p, v); }
};
#endif

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
