// Shared helpers of the DRUNet convolution kernels (drunet.hip, drunet_wino.hip).
#pragma once
#include "common.hpp"
#include <atomic>

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

// 3-D volumes ride the 2-D kernels as stacks of slices: a volume of D slices occupies D + 2 consecutive "images" of the
// padded layout (one zero slice at each end), so a 3x3x3 convolution is three 3x3 launches on slice-shifted views.
// The stride-2 layers pair slice z of the half grid with slice 2 z + dz of the full grid:
struct DepthMap {
    int32_t dep_s, dep_l;   // images per volume (D + 2) on the half grid / on the full grid; 0: plain 2-D
    int32_t dz;             // depth tap (0 or 1)
};
// image of the full grid paired with image `b` of the half grid; -1 for a zero (padding) slice
__device__ __forceinline__ int64_t depth_pair(const DepthMap& d, int64_t b) {
    if (d.dep_s == 0) return b;
    const int64_t vol = b / d.dep_s;
    const int z = (int)(b - vol * d.dep_s);
    if (z < 1 || z > d.dep_s - 2) return -1;
    return vol * d.dep_l + 2 * (z - 1) + d.dz + 1;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load and
// store of the wave, which the software pipelines of the conv kernels want to keep in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
#ifdef DINV_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// hide a per-lane value from the optimizer (it can then neither re-materialise nor re-associate what it was computed from)
#ifdef DINV_EMU
#define DINV_OPAQUE(x) asm volatile("" : "+r"(x))
#else
#define DINV_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float comp(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// epilogue shared by the three conv kernels: D[i][j], j = lane&31 (pixel), i = (reg&3) + 8*(reg>>2) + 4*h (cout),
// i.e. register quad g = reg>>2 holds couts 8g+4h .. 8g+4h+3 = one float4 of channel block g.
template <int MREP, bool RELU, int NRES>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[MREP][2], int n, int64_t opix, bool in, int cb0,
                                           int cblocks_valid, int64_t cs, int lhi, float* __restrict__ y,
                                           const float* __restrict__ res1, const float* __restrict__ res2) {
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        float4 r1[4], r2[4];
        if (NRES >= 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (cb0 + 4 * m + g < cblocks_valid) r1[g] = ld4(res1 + ((int64_t)(cb0 + 4 * m + g) * cs + opix) * 8 + 4 * lhi);
        }
        if (NRES >= 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (cb0 + 4 * m + g < cblocks_valid) r2[g] = ld4(res2 + ((int64_t)(cb0 + 4 * m + g) * cs + opix) * 8 + 4 * lhi);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (cb0 + 4 * m + g >= cblocks_valid) continue;
            float4 v = make_float4(acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
            if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (NRES >= 1) v = add4(v, r1[g]);
            if (NRES >= 2) v = add4(v, r2[g]);
            if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
            st4(y + ((int64_t)(cb0 + 4 * m + g) * cs + opix) * 8 + 4 * lhi, v);
        }
    }
}

// exact unsigned division by a runtime constant (Granlund-Montgomery round-up multiplier): n / d for all 32-bit n
struct FastDiv {
    uint32_t m, s1, s2;
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
};
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}

// compute units per XCD of the current device (32 on MI355X: 256 CUs in 8 XCDs)
inline int cus_per_xcd(int dev) {
    static std::atomic<int> cache[64];
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        v = n / 8;
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

inline int check_geom(const dinv_act_geom* g) {
    DINV_REQUIRE(g != nullptr, "null geometry");
    DINV_REQUIRE(g->batch >= 1 && g->height >= 1 && g->width >= 1, "bad geometry %dx%dx%d", g->batch, g->height, g->width);
    DINV_REQUIRE(g->wp % 4 == 0 && g->wp >= g->width + 2 && g->hp == g->height + 2, "bad padded frame");
    DINV_REQUIRE(g->sl >= g->wp + HALO, "bad leading slack");
    DINV_REQUIRE(g->cs >= g->sl + dinv::ceil_div(g->np, NT) * NT + g->wp + HALO, "channel-block stride too small");
    return 0;
}


}  // namespace dinv_drunet
