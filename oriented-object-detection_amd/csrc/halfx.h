// 16-bit storage types of the forward path.  Activations and weights are stored either as IEEE fp16 (default: what
// Ultralytics' own half=True inference uses; 11-bit significand keeps the result ~8x closer to the reference's fp32)
// or as bf16; accumulation, bias, SiLU, residual adds and softmax are always fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace obb {

typedef uint16_t half_bits_t;  // raw 16-bit pattern, interpretation given by the F16 template flag

template <bool F16>
struct HX;

template <>
struct HX<false> {  // bf16
    typedef __bf16 elem;
    typedef __attribute__((ext_vector_type(8))) __bf16 vec8;
    static __device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ float one(half_bits_t b) { return __uint_as_float((uint32_t)b << 16); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        __bf16 x = (__bf16)a, y = (__bf16)b;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        uint16_t ux, uy;
        __builtin_memcpy(&ux, &x, 2);
        __builtin_memcpy(&uy, &y, 2);
        return (uint32_t)ux | ((uint32_t)uy << 16);
    }
    static __device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma(vec8 a, vec8 b, __attribute__((ext_vector_type(4))) float c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <>
struct HX<true> {  // fp16
    typedef _Float16 elem;
    typedef __attribute__((ext_vector_type(8))) _Float16 vec8;
    static __device__ __forceinline__ float one(half_bits_t b) {
        _Float16 h;
        __builtin_memcpy(&h, &b, 2);
        return (float)h;
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return one((half_bits_t)(u & 0xffffu)); }
    static __device__ __forceinline__ float hi(uint32_t u) { return one((half_bits_t)(u >> 16)); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        _Float16 x = (_Float16)a, y = (_Float16)b;  // v_cvt_f16_f32: RNE (never the round-toward-zero pkrtz form)
        uint16_t ux, uy;
        __builtin_memcpy(&ux, &x, 2);
        __builtin_memcpy(&uy, &y, 2);
        return (uint32_t)ux | ((uint32_t)uy << 16);
    }
    static __device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma(vec8 a, vec8 b, __attribute__((ext_vector_type(4))) float c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <bool F16>
__device__ __forceinline__ void unpack8(const uint4 &v, float *f) {
    f[0] = HX<F16>::lo(v.x); f[1] = HX<F16>::hi(v.x); f[2] = HX<F16>::lo(v.y); f[3] = HX<F16>::hi(v.y);
    f[4] = HX<F16>::lo(v.z); f[5] = HX<F16>::hi(v.z); f[6] = HX<F16>::lo(v.w); f[7] = HX<F16>::hi(v.w);
}
template <bool F16>
__device__ __forceinline__ uint4 pack8(const float *f) {
    return make_uint4(HX<F16>::pack2(f[0], f[1]), HX<F16>::pack2(f[2], f[3]), HX<F16>::pack2(f[4], f[5]), HX<F16>::pack2(f[6], f[7]));
}

// acc[0..7] += x[0..7] * w[0..7] for two vectors of eight 16-bit values, fp32 accumulate.  fp16: v_fma_mix_f32 reads both halves of a
// dword directly (no conversion instructions; the compiler otherwise spends two v_cvt_f32_f16 per product); bf16: shifts / masks.
template <bool F16>
__device__ __forceinline__ void fma8_mixed(float (&acc)[8], const uint4 &x, const uint4 &w) {
    const uint32_t xd[4] = {x.x, x.y, x.z, x.w}, wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if constexpr (F16) {
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * d]) : "v"(xd[d]), "v"(wd[d]));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * d + 1]) : "v"(xd[d]), "v"(wd[d]));
        } else {
            acc[2 * d] = __builtin_fmaf(HX<false>::lo(xd[d]), HX<false>::lo(wd[d]), acc[2 * d]);
            acc[2 * d + 1] = __builtin_fmaf(HX<false>::hi(xd[d]), HX<false>::hi(wd[d]), acc[2 * d + 1]);
        }
    }
}

// SiLU with the hardware exp and reciprocal (v_exp_f32 / v_rcp_f32, ~1 ulp): its error is far below the 16-bit storage rounding
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// host-side conversions (RNE)
inline half_bits_t host_to_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (half_bits_t)((u >> 16) | 0x40);
    return (half_bits_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline half_bits_t host_to_f16(float f) {
    _Float16 h = (_Float16)f;
    half_bits_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
inline float host_from_bf16(half_bits_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
inline float host_from_f16(half_bits_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
inline half_bits_t host_to_half(float f, bool f16) { return f16 ? host_to_f16(f) : host_to_bf16(f); }
inline float host_from_half(half_bits_t h, bool f16) { return f16 ? host_from_f16(h) : host_from_bf16(h); }

}  // namespace obb
