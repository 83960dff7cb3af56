// DRUNet 3x3 convolution on the BF16 matrix cores with fp32-level accuracy (EXPERIMENTAL, opt-in: DINV_CONV_BF16X3=1).
//
// Every fp32 operand is split exactly into three bf16 parts, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), ...),
// and a product a*b is evaluated as the six leading terms a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 with fp32
// accumulation in the MFMA (dropped terms are below 2^-24 relative: same accuracy class as an fp32 FMA chain; a CPU
// emulation on conv-shaped sums gives 1.0e-7 relative error vs 2.2e-7 for plain fp32 accumulation).
// v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate and is a separate pipe (VALU co-issues, unlike the fp32
// MFMA: scripts/ubench/mfma_coissue.hip), so six products still leave 2.7x.
//
// Same operator, layout, tiling and epilogue as conv3x3_kernel (drunet.hip): workgroup = 256 pixels x 64 couts,
// 4 waves x (2x2 MFMA tiles).  K = 16 per MFMA = 8 channels x 2 taps: lane half h takes tap 2q + h (the tenth tap
// has zero weights).  Per 8-channel block the activations are split while they are staged (fp32 global -> three
// bf16 planes in LDS, 16 bytes per pixel and plane); the weights are pre-split on the host.
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WPLANE = 9 * 64;               // 16-byte units of one weight plane per block and cout tile
constexpr int WBLK = 3 * WPLANE;             // ... of the packed weights (always three planes)

struct Bf16Args {
    Geom g;
    const float* x;
    const uint4* w;    // [cout/64][cin/8][plane 3][tap 9][co 64] x (8 bf16)
    float* y;
    const float* res1;
    int32_t cin, cblocks_valid;
    int32_t ntiles, ytiles, tiles_per_xcd;
};

__device__ __forceinline__ void split3(const float4& v, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
    p1 = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    const float4 r = make_float4(v.x - (float)p1[0], v.y - (float)p1[1], v.z - (float)p1[2], v.w - (float)p1[3]);
    p2 = bf16x4{(__bf16)r.x, (__bf16)r.y, (__bf16)r.z, (__bf16)r.w};
    const float4 q = make_float4(r.x - (float)p2[0], r.y - (float)p2[1], r.z - (float)p2[2], r.w - (float)p2[3]);
    p3 = bf16x4{(__bf16)q.x, (__bf16)q.y, (__bf16)q.z, (__bf16)q.w};
}

// NPL = 3: six products (fp32-class accuracy, ~3e-7 per layer measured); NPL = 2: two-way split, three products
// (a1b1 + a1b2 + a2b1, ~5e-6 per layer), half the matrix work and a third less LDS.
template <int NPL, bool RELU, int NRES>
__global__ __launch_bounds__(256) void conv3x3_bf16x3_kernel(Bf16Args a) {
    constexpr int XS16 = NPL * 3 * SEG, WS16 = NPL * WPLANE;
    __shared__ __attribute__((aligned(16))) uint4 xs[XS16];   // [plane][row 3][SEG] x 8 bf16
    __shared__ __attribute__((aligned(16))) uint4 ws[WS16];   // [plane][tap 9][co 64] x 8 bf16
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int64_t p0 = (int64_t)tile * NT;
    const int nchunks = a.cin / KC;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int XSEG = SEG * 2;           // fp32 float4 per staged row segment (2 per pixel)
    constexpr int XV = 3 * XSEG;            // 1548 float4 of activations per block
    constexpr int XI = (XV + 255) / 256, WI = (WS16 + 255) / 256;
    static_assert(XI == 7 && WI <= 7, "prefetch slots");
    const uint4* wblk = a.w + (int64_t)ty * nchunks * WBLK;

    int xoff_lds[XI], xoff_g[XI];           // LDS offset in 8-byte units inside plane 0, global offset in floats
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = min(tid + it * 256, XV - 1);
        const int seg = idx / XSEG, r = idx - seg * XSEG;
        xoff_lds[it] = (seg * SEG + (r >> 1)) * 2 + (r & 1);
        xoff_g[it] = ((seg - 1) * a.g.wp - HALO) * 8 + r * 4;
    }
    const int64_t row0 = (a.g.sl + p0) * 8;

    // this lane's operand slots per tap pair q: tap t = 2q + lhi (t = 9: zero weights, any valid pixel)
    int aoff[5], boff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = min(2 * q + lhi, 8);
        aoff[q] = t * 64 + l31;
        boff[q] = (t / 3) * SEG + wv * 64 + l31 + HALO + (t % 3 - 1);
    }
    const bool atail = lhi != 0;            // lanes of the upper half have no tenth tap

    float4 xv0, xv1, xv2, xv3, xv4, xv5, xv6;
    uint4 w0, w1, w2, w3, w4, w5, w6;
    unsigned long long* xs8 = reinterpret_cast<unsigned long long*>(xs);
    for (int ch = -1; ch < nchunks; ++ch) {
        if (ch >= 0) {
            __syncthreads();  // previous block's MFMA phase has consumed LDS
#define DINV_XST(IT, REG)                                                                                        \
    if (IT < XI - 1 || tid + IT * 256 < XV) {                                                                    \
        bf16x4 s1_, s2_, s3_;                                                                                    \
        split3(REG, s1_, s2_, s3_);                                                                              \
        xs8[xoff_lds[IT]] = __builtin_bit_cast(unsigned long long, s1_);                                         \
        xs8[3 * SEG * 2 + xoff_lds[IT]] = __builtin_bit_cast(unsigned long long, s2_);                           \
        if (NPL > 2) xs8[2 * 3 * SEG * 2 + xoff_lds[IT]] = __builtin_bit_cast(unsigned long long, s3_);          \
    }
            DINV_XST(0, xv0) DINV_XST(1, xv1) DINV_XST(2, xv2) DINV_XST(3, xv3) DINV_XST(4, xv4) DINV_XST(5, xv5) DINV_XST(6, xv6)
#undef DINV_XST
#define DINV_WST(IT, REG) if (IT < WI && (IT < WI - 1 || tid + IT * 256 < WS16)) ws[tid + IT * 256] = REG;
            DINV_WST(0, w0) DINV_WST(1, w1) DINV_WST(2, w2) DINV_WST(3, w3) DINV_WST(4, w4) DINV_WST(5, w5) DINV_WST(6, w6)
#undef DINV_WST
            __syncthreads();
        }
        if (ch + 1 < nchunks) {
            const float* xb = a.x + (int64_t)(ch + 1) * a.g.cs * 8 + row0;
#define DINV_XLD(IT, REG) REG = ld4(xb + xoff_g[IT]);
            DINV_XLD(0, xv0) DINV_XLD(1, xv1) DINV_XLD(2, xv2) DINV_XLD(3, xv3) DINV_XLD(4, xv4) DINV_XLD(5, xv5) DINV_XLD(6, xv6)
#undef DINV_XLD
            const uint4* wsrc = wblk + (int64_t)(ch + 1) * WBLK;
#define DINV_WLD(IT, REG) if (IT < WI) REG = wsrc[min(tid + IT * 256, WS16 - 1)];
            DINV_WLD(0, w0) DINV_WLD(1, w1) DINV_WLD(2, w2) DINV_WLD(3, w3) DINV_WLD(4, w4) DINV_WLD(5, w5) DINV_WLD(6, w6)
#undef DINV_WLD
        }
        if (ch < 0) continue;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            bf16x8 A[2][NPL], B[2][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    uint4 v = ws[pl * 9 * 64 + aoff[q] + m * 32];
                    if (q == 4 && atail) v = make_uint4(0u, 0u, 0u, 0u);
                    A[m][pl] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) B[n][pl] = __builtin_bit_cast(bf16x8, xs[pl * 3 * SEG + boff[q] + n * 32]);
            }
            // leading terms of (a1 + a2 + a3)(b1 + b2 + b3), smallest first: six for NPL = 3, three for NPL = 2
            constexpr int NE = NPL == 3 ? 6 : 3;
            constexpr int PA[6] = {NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0, 0};
            constexpr int PB[6] = {0, 1, NPL == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][PA[e]], B[n][PB[e]], acc[m][n], 0, 0, 0);
        }
    }
    const int cb0 = ty * 8;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t p = p0 + wv * 64 + n * 32 + l31;
        if (p >= a.g.np) continue;
        store_tile<2, RELU, NRES>(acc, n, a.g.sl + p, interior(a.g, p), cb0, a.cblocks_valid, a.g.cs, lhi, a.y, a.res1,
                                  nullptr);
    }
}

}  // namespace

extern "C" int dinv_conv3x3_bf16x3(const dinv_act_geom* g, const float* x, const void* w_split, int32_t cin,
                                   int32_t cout, float* y, const float* res1, int32_t relu, int32_t planes,
                                   dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_split && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 8 && cin % 8 == 0 && cout >= 64 && cout % 64 == 0,
                 "bf16x3 conv needs cin %% 8 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(planes == 2 || planes == 3, "planes must be 2 (three products) or 3 (six products), got %d", planes);
    Bf16Args a{make_geom(*g), x, reinterpret_cast<const uint4*>(w_split), y, res1, cin, cout / 8, 0, 0, 0};
    a.ntiles = (int32_t)ceil_div(g->np, NT);
    a.ytiles = cout / 64;
    a.tiles_per_xcd = (int32_t)ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.tiles_per_xcd * a.ytiles * 8)), block(256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define DINV_LAUNCH(NPL_)                                                                                        \
    do {                                                                                                         \
        if (relu) hipLaunchKernelGGL((conv3x3_bf16x3_kernel<NPL_, true, 0>), grid, block, 0, st, a);             \
        else if (res1) hipLaunchKernelGGL((conv3x3_bf16x3_kernel<NPL_, false, 1>), grid, block, 0, st, a);      \
        else hipLaunchKernelGGL((conv3x3_bf16x3_kernel<NPL_, false, 0>), grid, block, 0, st, a);                 \
    } while (0)
    if (planes == 3) DINV_LAUNCH(3); else DINV_LAUNCH(2);
#undef DINV_LAUNCH
    DINV_CHECK_LAUNCH();
    return 0;
}
