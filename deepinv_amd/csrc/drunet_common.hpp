// Shared helpers of the DRUNet convolution kernels (drunet.hip, drunet_wino.hip).
#pragma once
#include "common.hpp"

namespace dinv_drunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;          // pixels per workgroup (direct kernel)
constexpr int KC = 8;            // channels per block
constexpr int HALO = 1;          // staged halo pixels on each side of a row segment
constexpr int SEG = NT + 2 * HALO;
constexpr int LP = 12;           // LDS row pitch in floats (8 used + 4 pad): conflict-free b128 reads

struct Geom {
    int32_t batch, h, w, hp, wp;
    int64_t plane, np, sl, cs;
};

__host__ __device__ inline Geom make_geom(const dinv_act_geom& g) {
    Geom r;
    r.batch = g.batch; r.h = g.height; r.w = g.width; r.hp = g.hp; r.wp = g.wp;
    r.plane = g.plane; r.np = g.np; r.sl = g.sl; r.cs = g.cs;
    return r;
}

__device__ __forceinline__ bool interior(const Geom& g, int64_t p) {
    if (p >= g.np) return false;
    const int pi = (int)(p % g.plane);
    const int r = pi / g.wp, c = pi - r * g.wp;
    return r >= 1 && r <= g.h && c >= 1 && c <= g.w;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float comp(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

inline int check_geom(const dinv_act_geom* g) {
    DINV_REQUIRE(g != nullptr, "null geometry");
    DINV_REQUIRE(g->batch >= 1 && g->height >= 1 && g->width >= 1, "bad geometry %dx%dx%d", g->batch, g->height, g->width);
    DINV_REQUIRE(g->wp % 4 == 0 && g->wp >= g->width + 2 && g->hp == g->height + 2, "bad padded frame");
    DINV_REQUIRE(g->sl >= g->wp + HALO, "bad leading slack");
    DINV_REQUIRE(g->cs >= g->sl + dinv::ceil_div(g->np, NT) * NT + g->wp + HALO, "channel-block stride too small");
    return 0;
}


}  // namespace dinv_drunet
