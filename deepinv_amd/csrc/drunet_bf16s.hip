// DRUNet 2x2 stride-2 down / up convolutions on the BF16 matrix cores with the two-part exact operand split (gfx950):
// x = xh + xl (xh = bf16(x), xl = bf16(x - xh)), a product is ah*bl + al*bh + ah*bh with fp32 accumulation in
// v_mfma_f32_32x32x16_bf16 - the same arithmetic as the ResBlock 3x3 convolutions (drunet_split2d.hip).
// The down convolution also exists with a THREE-part split (NPL = 3: x = xh + xm + xl, 24 significand bits = all of an fp32
// operand; six products ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm, the dropped terms are below 2^-24 of |a||b|): the
// fp32-equivalent form that the `conv_precision = "fp32"` setting uses in place of the fp32-MFMA kernel of drunet.hip, which
// is bound by the fp32 matrix pipe (0.60 ms per launch at every level of the headline configuration).
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

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

// 8 fp32 -> three bf16 planes (high, middle, low parts: x = h + m + l to 24 bits)
__device__ __forceinline__ void split8x3(const float4& a, const float4& b, uint4& hi, uint4& mi, uint4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = f2bf(v[e]);
        const float r1 = v[e] - bf2f(h[e]);          // exact
        m[e] = f2bf(r1);
        l[e] = f2bf(r1 - bf2f(m[e]));                // exact difference, rounded once
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    mi = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
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

// ---------------------------------------------------------------------------------------------------------------------
// 2x2 stride-2 convolution (downsample_strideconv, deepinv/models/drunet.py:524-552) with the same operand split.
// K = 4 taps x Cin.  Every input pixel feeds exactly one output pixel and tap, so the B operand goes global -> registers
// -> split -> MFMA with no LDS stage (that stream IS the HBM traffic of the layer: the fp32 kernel of drunet.hip is
// bound by the fp32 matrix pipe, 0.66 ms at level 0; this one by HBM); the pre-split weights of a K step (one tap, 16
// channels, 64 couts: 4 KB) are shared by the four waves and double-buffered in LDS, one LDS-only barrier per K step.
struct DownSArgs {
    Geom gi, go;
    const float* x;    // [cin/8][gi.cs][8]
    const uint4* w;    // [tap 4][cin/16][plane NPL][cblk 2][cout] x (8 bf16)
    float* y;          // [cout/8][go.cs][8]
    int32_t cin, cout;
    int64_t ntiles, per_xcd;
    DepthMap dm;       // 3-D: output image (half grid) -> input image (full grid) for depth tap dm.dz
};

// ACC: y += conv (second depth tap of a 2x2x2 convolution).  NPL = 2: two-part split, three products; NPL = 3: three-part split,
// six products (fp32-equivalent)
template <bool ACC, int NPL>
__global__ __launch_bounds__(256) void down2x2_bf16s_kernel(DownSArgs a) {
    constexpr int WU = NPL * 128;                  // 16-byte weight units of a K step
    constexpr int WPT = (WU + 255) / 256;          // ... per thread
    __shared__ uint4 wl[2][WU];    // [stage][plane NPL][cblk 2][co 64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // cout tiles of one pixel tile next to each other on one XCD (they read the same input pixels)
    const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (logical >= a.ntiles) return;
    const int nct = a.cout / 64;
    const int64_t q0 = (logical / nct) * 256 + wv * 64;
    const int co0 = (int)(logical % nct) * 64;
    int64_t ioff[2];
    bool in[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + n * 32 + l31;
        in[n] = interior(a.go, q);
        ioff[n] = a.gi.sl;     // border / out-of-range lanes read a valid (zero frame) pixel; their result is not stored
        if (in[n]) {
            const int64_t b = q / a.go.plane;
            const int64_t bi = depth_pair(a.dm, b);
            if (bi < 0) in[n] = false;          // zero slice of a 3-D volume: stays zero
            else {
                const int qi = (int)(q - b * a.go.plane);
                const int R = qi / a.go.wp, C = qi - R * a.go.wp;
                ioff[n] = a.gi.sl + bi * a.gi.plane + (int64_t)(2 * (R - 1) + 1) * a.gi.wp + (2 * (C - 1) + 1);
            }
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int S = a.cin / 16, nsteps = 4 * S;
    // this thread's weight units of a K step: unit u = tid + 256 k = (plane * 2 + channel block) * 64 + cout
    float4 ba[2], bb[2];
    uint4 wreg[WPT];
    auto issue = [&](int g) {
        const int tap = g / S, s = g - tap * S;
        const int64_t toff = (int64_t)(tap >> 1) * a.gi.wp + (tap & 1);
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int u = tid + 256 * k;
            if (u < WU) wreg[k] = a.w[((int64_t)g * (2 * NPL) + (u >> 6)) * a.cout + co0 + (u & 63)];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* xb = a.x + ((int64_t)(2 * s + lhi) * a.gi.cs + ioff[n] + toff) * 8;
            ba[n] = ld4(xb);
            bb[n] = ld4(xb + 4);
        }
    };
    uint4 Bp[NPL][2];     // [plane: high, (middle,) low][n]
    auto commit = [&](int stage) {
#pragma unroll
        for (int k = 0; k < WPT; ++k)
            if (tid + 256 * k < WU) wl[stage][tid + 256 * k] = wreg[k];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if constexpr (NPL == 2) split8(ba[n], bb[n], Bp[0][n], Bp[1][n]);
            else split8x3(ba[n], bb[n], Bp[0][n], Bp[1][n], Bp[2][n]);
        }
    };
    issue(0);
    commit(0);
    if (nsteps > 1) issue(1);
    __syncthreads();
    // products (plane of a, plane of b), smallest terms first; product-major: consecutive MFMAs hit different accumulators
    constexpr int NPROD = NPL == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 1, 0, 0, 1, 0}, PB2[3] = {1, 0, 0};               // NPL = 2: ah*bl, al*bh, ah*bh
    constexpr int QA[6] = {1, 0, 2, 0, 1, 0}, QB[6] = {1, 2, 0, 1, 0, 0};       // NPL = 3: am*bm, ah*bl, al*bh, ah*bm, am*bh, ah*bh
    for (int g = 0; g < nsteps; ++g) {
        const uint4* ws = wl[g & 1];
        uint4 A[NPL][2];   // [plane][m]
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int m = 0; m < 2; ++m) A[pl][m] = ws[pl * 128 + lhi * 64 + m * 32 + l31];
#pragma unroll
        for (int e = 0; e < NPROD; ++e) {
            const int pa = NPL == 2 ? PA[e] : QA[e], pb = NPL == 2 ? PB2[e] : QB[e];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(A[pa][m], Bp[pb][n], acc[m][n]);
        }
        if (g + 1 < nsteps) {   // registers hold step g+1 (loaded one iteration ago)
            commit((g + 1) & 1);
            if (g + 2 < nsteps) issue(g + 2);
        }
        lds_barrier();
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + n * 32 + l31;
        if (q >= a.go.np) continue;
        store_tile<2, false, ACC ? 1 : 0>(acc, n, a.go.sl + q, in[n], co0 / 8, a.cout / 8, a.go.cs, lhi, a.y, ACC ? a.y : nullptr,
                                          nullptr);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2x2 stride-2 transposed convolution (upsample_convtranspose, deepinv/models/drunet.py:493-521), input optionally the
// sum of two tensors (U-Net skip add): four parity-class GEMMs with K = Cin.  Workgroup = 4 waves = 128 input pixels x
// 64 couts x 4 taps; wave (tp, pg) owns the tap pair dy = tp (dx = 0, 1) of pixel group pg (64 pixels): 8 accumulators.
// B operand global -> registers -> (skip add) -> split -> MFMA (each input pixel is read by the two waves of its pixel
// group, the second time from L1); the pre-split weights of a K step (4 taps x 16 channels x 64 couts: 16 KB) are
// double-buffered in LDS, one LDS-only barrier per K step.  Only interior output pixels are written.
// Measured (B = 32): 0.53 / 0.48 / 0.49 ms at the three levels against 0.55 / 0.49 / 0.49 ms for the fp32 kernel and HBM
// floors of 0.35 / 0.18 / 0.09 ms: neither matrix pipe is the limit here, the stride-2 scatter of the epilogue is (every
// store instruction writes 16-byte pieces 32 bytes apart; a full-line form needs a cross-lane transpose of the tile).
struct UpSArgs {
    Geom gi, go;
    const float* x;    // [cin/8][gi.cs][8]
    const float* x2;   // optional, added to x
    const uint4* w;    // [cin/16][tap 4][plane 2][cblk 2][cout] x (8 bf16)
    float* y;          // [cout/8][go.cs][8]
    int32_t cin, cout;
    DepthMap dm;       // 3-D: input image (half grid) -> output image (full grid) for depth tap dm.dz
};

#ifdef DINV_EMU
#define DINV_BF16S_UP_ATTR
#else
#define DINV_BF16S_UP_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <bool SKIP>
// (waves_per_eu(2, 2): with the default bounds the compiler keeps the 128 accumulator registers in AGPRs and spills 80 bytes per lane of
// MFMA temporaries to scratch; told that two waves per SIMD is all there will be, it allocates 222 unified registers and no scratch)
__global__ __launch_bounds__(256) DINV_BF16S_UP_ATTR void up2x2_bf16s_kernel(UpSArgs a) {
    __shared__ uint4 wl[2][1024];   // [stage][tap 4][plane 2][cblk 2][co 64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tp = wv & 1, pg = wv >> 1;
    const int co0 = blockIdx.y * 64;
    const int64_t p0 = (int64_t)blockIdx.x * 128 + pg * 64;
    int64_t ioff[2], ooff[2];
    bool in[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t p = p0 + n * 32 + l31;
        in[n] = interior(a.gi, p);
        ioff[n] = a.gi.sl + (in[n] ? p : 0);
        ooff[n] = 0;
        if (in[n]) {
            const int64_t b = p / a.gi.plane;
            const int64_t bo = depth_pair(a.dm, b);
            if (bo < 0) in[n] = false;          // zero slice of a 3-D volume: nothing to scatter
            else {
                const int pi = (int)(p - b * a.gi.plane);
                const int r = pi / a.gi.wp, c = pi - r * a.gi.wp;
                ooff[n] = a.go.sl + bo * a.go.plane + (int64_t)(2 * (r - 1) + 1 + tp) * a.go.wp + (2 * (c - 1) + 1);
            }
        }
    }
    f32x16 acc[2][2][2];   // [dx][m][n]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;
    const int S = a.cin / 16;
    // this thread's 4 weight units of a K step: unit u = tid + 256 k = ((tap*2 + plane)*2 + cblk)*64 + co
    float4 ba[2], bb[2];
    uint4 wreg[4];
    auto issue = [&](int s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = tid + 256 * k;
            wreg[k] = a.w[((int64_t)s * 16 + (u >> 6)) * a.cout + co0 + (u & 63)];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t o = ((int64_t)(2 * s + lhi) * a.gi.cs + ioff[n]) * 8;
            ba[n] = ld4(a.x + o);
            bb[n] = ld4(a.x + o + 4);
            if (SKIP) {
                ba[n] = add4(ba[n], ld4(a.x2 + o));
                bb[n] = add4(bb[n], ld4(a.x2 + o + 4));
            }
        }
    };
    auto commit = [&](int stage, uint4 (&Bh)[2], uint4 (&Bl)[2]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wl[stage][tid + 256 * k] = wreg[k];
        split8(ba[0], bb[0], Bh[0], Bl[0]);
        split8(ba[1], bb[1], Bh[1], Bl[1]);
    };
    uint4 Bh[2], Bl[2];
    issue(0);
    commit(0, Bh, Bl);
    if (S > 1) issue(1);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const uint4* ws = wl[s & 1] + (tp * 2) * 256;   // this wave's tap pair: taps 2 tp, 2 tp + 1
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint4 A[2][2];   // [plane][m]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int m = 0; m < 2; ++m) A[pl][m] = ws[t * 256 + pl * 128 + lhi * 64 + m * 32 + l31];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int pa = e == 1 ? 1 : 0;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[t][m][n] = mfma_bf16(A[pa][m], e == 0 ? Bl[n] : Bh[n], acc[t][m][n]);
            }
        }
        if (s + 1 < S) {
            commit((s + 1) & 1, Bh, Bl);
            if (s + 2 < S) issue(s + 2);
        }
        lds_barrier();
    }
    // lane holds co = co0 + 32 m + 8 rj + 4 lhi + (0..3) in registers 4 rj .. 4 rj + 3 of acc[t][m][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        if (!in[n]) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rj = 0; rj < 4; ++rj) {
                    const float4 v = make_float4(acc[t][m][n][4 * rj], acc[t][m][n][4 * rj + 1], acc[t][m][n][4 * rj + 2],
                                                 acc[t][m][n][4 * rj + 3]);
                    st4(a.y + ((int64_t)(co0 / 8 + m * 4 + rj) * a.go.cs + ooff[n] + t) * 8 + 4 * lhi, v);
                }
    }
}

}  // namespace

static int down2x2_bf16s_launch(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const void* w_split,
                                int32_t cin, int32_t cout, float* y, DepthMap dm, int accumulate, dinv_stream_t stream,
                                int planes = 2) {
    if (int e = check_geom(gin)) return e;
    if (int e = check_geom(gout)) return e;
    DINV_REQUIRE(x && w_split && y, "null tensor pointer");
    DINV_REQUIRE(gin->height == 2 * gout->height && gin->width == 2 * gout->width, "down2x2 geometry mismatch");
    if (dm.dep_s == 0) DINV_REQUIRE(gin->batch == gout->batch, "down2x2 geometry mismatch");
    else
        DINV_REQUIRE(dm.dep_s >= 3 && dm.dep_l == 2 * (dm.dep_s - 2) + 2 && gout->batch % dm.dep_s == 0 &&
                     gin->batch / dm.dep_l == gout->batch / dm.dep_s && gin->batch % dm.dep_l == 0 && (dm.dz == 0 || dm.dz == 1),
                     "down2x2: bad depth pairing (%d, %d, %d)", dm.dep_s, dm.dep_l, dm.dz);
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout % 64 == 0, "bf16-split down2x2 needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DownSArgs a{make_geom(*gin), make_geom(*gout), x, reinterpret_cast<const uint4*>(w_split), y, cin, cout, 0, 0, dm};
    a.ntiles = ceil_div(gout->np, 256) * (cout / 64);
    a.per_xcd = ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.per_xcd * 8));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (planes == 3) hipLaunchKernelGGL((down2x2_bf16s_kernel<false, 3>), grid, dim3(256), 0, st, a);
    else if (accumulate) hipLaunchKernelGGL((down2x2_bf16s_kernel<true, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((down2x2_bf16s_kernel<false, 2>), grid, dim3(256), 0, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_down2x2_bf16s(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                       const void* w_split, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    return down2x2_bf16s_launch(gin, gout, x, w_split, cin, cout, y, DepthMap{0, 0, 0}, 0, stream);
}

extern "C" int dinv_conv_down2x2_bf16x3(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                        const void* w_split3, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    return down2x2_bf16s_launch(gin, gout, x, w_split3, cin, cout, y, DepthMap{0, 0, 0}, 0, stream, 3);
}

extern "C" int dinv_conv_down2x2_bf16s_3d(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                          const void* w_split, int32_t cin, int32_t cout, float* y, int32_t depth_out,
                                          int32_t dz, int32_t accumulate, dinv_stream_t stream) {
    DINV_REQUIRE(depth_out >= 1, "bad depth %d", depth_out);
    return down2x2_bf16s_launch(gin, gout, x, w_split, cin, cout, y, DepthMap{depth_out + 2, 2 * depth_out + 2, dz}, accumulate, stream);
}

static int up2x2_bf16s_launch(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                              const void* w_split, int32_t cin, int32_t cout, float* y, DepthMap dm, dinv_stream_t stream) {
    if (int e = check_geom(gin)) return e;
    if (int e = check_geom(gout)) return e;
    DINV_REQUIRE(x && w_split && y, "null tensor pointer");
    DINV_REQUIRE(gout->height == 2 * gin->height && gout->width == 2 * gin->width, "up2x2 geometry mismatch");
    if (dm.dep_s == 0) DINV_REQUIRE(gin->batch == gout->batch, "up2x2 geometry mismatch");
    else
        DINV_REQUIRE(dm.dep_s >= 3 && dm.dep_l == 2 * (dm.dep_s - 2) + 2 && gin->batch % dm.dep_s == 0 &&
                     gout->batch % dm.dep_l == 0 && gout->batch / dm.dep_l == gin->batch / dm.dep_s && (dm.dz == 0 || dm.dz == 1),
                     "up2x2: bad depth pairing (%d, %d, %d)", dm.dep_s, dm.dep_l, dm.dz);
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout % 64 == 0, "bf16-split up2x2 needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    UpSArgs a{make_geom(*gin), make_geom(*gout), x, x2, reinterpret_cast<const uint4*>(w_split), y, cin, cout, dm};
    const dim3 grid((unsigned)ceil_div(gin->np, 128), (unsigned)(cout / 64));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x2) hipLaunchKernelGGL(up2x2_bf16s_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(up2x2_bf16s_kernel<false>, grid, dim3(256), 0, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_up2x2_bf16s(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                                     const void* w_split, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    return up2x2_bf16s_launch(gin, gout, x, x2, w_split, cin, cout, y, DepthMap{0, 0, 0}, stream);
}

extern "C" int dinv_conv_up2x2_bf16s_3d(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                                        const void* w_split, int32_t cin, int32_t cout, float* y, int32_t depth_in, int32_t dz,
                                        dinv_stream_t stream) {
    DINV_REQUIRE(depth_in >= 1, "bad depth %d", depth_in);
    return up2x2_bf16s_launch(gin, gout, x, x2, w_split, cin, cout, y, DepthMap{depth_in + 2, 2 * depth_in + 2, dz}, stream);
}
