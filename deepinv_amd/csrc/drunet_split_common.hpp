// Helpers shared by the bf16-split convolution kernels (drunet_split2d.hip, drunet_wsplit.hip): the two-part operand split
// x = xh + xl (xh = bf16(x), xl = bf16(x - xh), both round-to-nearest-even; x - xh is exact) and the bf16 MFMA wrapper.
#pragma once
#include "drunet_common.hpp"

namespace dinv_drunet {

__device__ __forceinline__ unsigned f2bf(float f) {   // round to nearest even, as v_cvt_pk_bf16_f32
#ifdef DINV_EMU
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
#else
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
#endif
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

// 8 fp32 -> 8 bf16 high parts + 8 bf16 low parts (each 16 bytes)
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = f2bf(v[e]);
        l[e] = f2bf(v[e] - bf2f(h[e]));
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

#ifdef DINV_EMU
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16& c) {
    emu_bf16x8 av, bv;
    std::memcpy(&av, &a, 16);
    std::memcpy(&bv, &b, 16);
    return emu_mfma_f32_32x32x16_bf16(av, bv, c);
}
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#endif

__device__ __forceinline__ uint4 ldu4(const float* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ float4 as_f4(const uint4& u) {
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}

}  // namespace dinv_drunet
