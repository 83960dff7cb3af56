// Weight gradients of the DRUNet convolutions (training through deepinv.unfolded, deepinv/unfolded/unfolded.py:116-226
// with a DRUNet prior, deepinv/models/drunet.py:39-263) on the padded channel-blocked activation layout of drunet.hip.
//
// The DATA gradients need no new kernels: the gradient of a 3x3 convolution with respect to its input is the 3x3
// convolution with the transposed, spatially flipped weights; that of the 2x2 stride-2 convolution is the 2x2 stride-2
// transposed convolution with the same weights and vice versa (deepinv_amd/models/drunet_train.py re-packs the weights
// and calls the forward kernels).  What is new here:
//
//   dW[m][n][t] = sum_p  S[m][p] * L[n][map(p) + off_t]
//
//   3x3 conv   (y = conv(x, w), w [Cout,Cin,3,3]):      S = dL/dy, L = x, map(p) = p, off_t = (dy-1)*wp + (dx-1), 9 taps
//   2x2 down   (w [Cout,Cin,2,2], y on the half grid):   S = dL/dy (half grid), L = x (full grid), map = (2r-1, 2c-1)
//   2x2 up     (w [Cin,Cout,2,2], y on the double grid): S = x (half grid),  L = dL/dy (full grid), same map,   4 taps
//
// i.e. one GEMM with K = all padded pixels (the zero frame of S makes border pixels contribute nothing), M x N = the two
// channel counts, done on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, K = 2 pixels per
// instruction, the lane halves take the even / odd pixel), operands straight from global memory (a lane reads the 4
// bytes of its channel; 32 lanes cover 4 adjacent channel blocks of one pixel).  One wave owns a 32 x 32 (m, n) tile for
// all taps over a slice of the pixels; the slices' partial sums go to a workspace and a second kernel adds them in a
// fixed order (deterministic, no atomics).
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

struct WgradArgs {
    Geom gs, gl;          // geometry of S and of L (equal for the 3x3 case)
    const float* s;       // [cs_alloc/8][gs.cs][8]
    const float* l;       // [cl_alloc/8][gl.cs][8]
    float* part;          // [nsplit][M][N][T]
    int32_t M, N;         // logical channel counts (rows / columns of dW)
    int32_t cs_alloc, cl_alloc;   // allocated channels (multiples of 8) of S and L
    int32_t mt, nt;       // 32-wide tiles
    int32_t nsplit;
    int64_t per_split;    // pixels per slice (even)
    DepthMap dm;          // 3-D stride-2 layers: image of S (half grid) -> image of L (full grid)
};

template <int T, bool STRIDE2>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wv;           // (m tile, n tile, slice), slice fastest
    const int64_t ntile = (int64_t)a.mt * a.nt * a.nsplit;
    if (unit >= ntile) return;
    const int split = (int)(unit % a.nsplit);
    const int tile = (int)(unit / a.nsplit);
    const int m0 = (tile / a.nt) * 32, n0 = (tile % a.nt) * 32;
    const int cm = m0 + l31, cn = n0 + l31;
    const bool mv = cm < a.cs_alloc, nv = cn < a.cl_alloc;        // lanes beyond the allocated channels feed zeros
    const float* sp = a.s + ((int64_t)(mv ? cm / 8 : 0) * a.gs.cs + a.gs.sl) * 8 + (mv ? cm % 8 : 0);
    const float* lp = a.l + ((int64_t)(nv ? cn / 8 : 0) * a.gl.cs + a.gl.sl) * 8 + (nv ? cn % 8 : 0);
    int64_t off[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        off[t] = STRIDE2 ? ((int64_t)(t >> 1) * a.gl.wp + (t & 1)) * 8 : ((int64_t)(t / 3 - 1) * a.gl.wp + (t % 3 - 1)) * 8;
    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int64_t pbeg = (int64_t)split * a.per_split;
    const int64_t pend = min(pbeg + a.per_split, a.gs.np);
    // U k-steps (2 pixels each) per iteration: all operand loads of the iteration are issued before its first MFMA, so
    // one memory latency is paid per U * T MFMAs instead of per T (the loop is latency-, not bandwidth-bound)
    constexpr int U = 4;
    for (int64_t p2 = pbeg; p2 < pend; p2 += 2 * U) {
        float sv[U], lv[U][T];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p2 + 2 * u + lhi;
            const bool pv = p < pend;
            int64_t q = pv ? p : 0;                          // pixel of L that tap (0,0) of pixel p reads
            if (STRIDE2) {
                q = 0;
                if (pv) {
                    const int64_t b = p / a.gs.plane;
                    const int pi = (int)(p - b * a.gs.plane);
                    const int r = pi / a.gs.wp, c = pi - r * a.gs.wp;
                    // frame pixels (and zero slices) of S are zero: send them to a valid address
                    const int64_t bl = depth_pair(a.dm, b);
                    if (bl >= 0 && r >= 1 && r <= a.gs.h && c >= 1 && c <= a.gs.w)
                        q = bl * a.gl.plane + (int64_t)(2 * (r - 1) + 1) * a.gl.wp + (2 * (c - 1) + 1);
                }
            }
            sv[u] = (pv && mv) ? sp[p * 8] : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) lv[u][t] = (pv && nv) ? lp[q * 8 + off[t]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], lv[u][t], acc[t], 0, 0, 0);
    }
    // D[i][j]: j = l31 (column n), i = (reg & 3) + 8 (reg >> 2) + 4 lhi (row m)
    float* out = a.part + (int64_t)split * a.M * a.N * T;
    const int n = n0 + l31;
    if (n < a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m >= a.M) continue;
#pragma unroll
            for (int t = 0; t < T; ++t) out[((int64_t)m * a.N + n) * T + t] = acc[t][r];
        }
    }
}

// The same reduction for THIN layers (M <= 16 and N <= 16: the 16-channel level of BASELINE config 4's 3-D DRUNet, which holds
// most of its voxels): v_mfma_f32_16x16x4_f32 - a 16 x 16 (m, n) tile, K = 4 pixels per instruction (lane quarter k takes
// pixel k), 32 cycles instead of the 64 a 32 x 32 x 2 instruction spends on a tile that would be three quarters padding:
// 4x fewer matrix-pipe cycles per pixel (the 32 x 32 form was pipe-bound there: 9 taps x 64 cycles per 2 pixels).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int T, bool STRIDE2>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wv;           // one (m, n) tile: the unit is the pixel slice
    if (unit >= a.nsplit) return;
    const int split = (int)unit;
    const bool mv = l15 < a.cs_alloc, nv = l15 < a.cl_alloc;     // lanes beyond the allocated channels feed zeros
    const float* sp = a.s + ((int64_t)(mv ? l15 / 8 : 0) * a.gs.cs + a.gs.sl) * 8 + (mv ? l15 % 8 : 0);
    const float* lp = a.l + ((int64_t)(nv ? l15 / 8 : 0) * a.gl.cs + a.gl.sl) * 8 + (nv ? l15 % 8 : 0);
    int64_t off[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        off[t] = STRIDE2 ? ((int64_t)(t >> 1) * a.gl.wp + (t & 1)) * 8 : ((int64_t)(t / 3 - 1) * a.gl.wp + (t % 3 - 1)) * 8;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t pbeg = (int64_t)split * a.per_split;
    const int64_t pend = min(pbeg + a.per_split, a.gs.np);
    constexpr int U = 4;      // k-steps (4 pixels each) whose loads are issued before the first MFMA of the iteration
    for (int64_t p4 = pbeg; p4 < pend; p4 += 4 * U) {
        float sv[U], lv[U][T];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p4 + 4 * u + lq;
            const bool pv = p < pend;
            int64_t q = pv ? p : 0;
            if (STRIDE2) {
                q = 0;
                if (pv) {
                    const int64_t b = p / a.gs.plane;
                    const int pi = (int)(p - b * a.gs.plane);
                    const int r = pi / a.gs.wp, c = pi - r * a.gs.wp;
                    const int64_t bl = depth_pair(a.dm, b);
                    if (bl >= 0 && r >= 1 && r <= a.gs.h && c >= 1 && c <= a.gs.w)
                        q = bl * a.gl.plane + (int64_t)(2 * (r - 1) + 1) * a.gl.wp + (2 * (c - 1) + 1);
                }
            }
            sv[u] = (pv && mv) ? sp[p * 8] : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) lv[u][t] = (pv && nv) ? lp[q * 8 + off[t]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[u], lv[u][t], acc[t], 0, 0, 0);
    }
    // D[i][j]: j = lane & 15 (column n), i = 4 (lane >> 4) + reg (row m)
    float* out = a.part + (int64_t)split * a.M * a.N * T;
    if (l15 < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * lq + r;
            if (m >= a.M) continue;
#pragma unroll
            for (int t = 0; t < T; ++t) out[((int64_t)m * a.N + l15) * T + t] = acc[t][r];
        }
    }
}

// dw[e] (+)= sum over slices in a fixed order: 16 lanes share an element (lane j adds slices j, j + 16, ... in order,
// then the 16 partial sums are added in lane order), 16 elements per workgroup
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t n, int nsplit, int accumulate,
                                                           float* __restrict__ dw) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + el;
    float v = 0.f;
    if (e < n)
        for (int s = grp; s < nsplit; s += 16) v += part[(int64_t)s * n + e];
    red[grp][el] = v;
    __syncthreads();
    if (grp == 0 && e < n) {
        float t = accumulate ? dw[e] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][el];
        dw[e] = t;
    }
}

// g <- g where a > 0, else 0   (backward of the ReLU between the two convolutions of a ResBlock, drunet.py:403-434)
__global__ __launch_bounds__(256) void relu_backward_kernel(int64_t n4, const float4* __restrict__ act, float4* __restrict__ g) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = act[i];
        float4 v = g[i];
        v.x = a.x > 0.f ? v.x : 0.f;
        v.y = a.y > 0.f ? v.y : 0.f;
        v.z = a.z > 0.f ? v.z : 0.f;
        v.w = a.w > 0.f ? v.w : 0.f;
        g[i] = v;
    }
}

__global__ __launch_bounds__(256) void relu_kernel(int64_t n4, float4* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        x[i] = v;
    }
}

int split_count(const dinv_act_geom* gs, int mt, int nt) {
    // enough waves for ~8 per CU, slices of at least 512 pixels, an even number of pixels per slice
    const int64_t want = (int64_t)256 * 8 / std::max(1, mt * nt);
    const int64_t maxs = std::max<int64_t>(1, gs->np / 512);
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, maxs), 1024));
}

}  // namespace

extern "C" size_t dinv_conv_wgrad_workspace_bytes(const dinv_act_geom* gs, int32_t m, int32_t n, int32_t taps) {
    if (!gs || m < 1 || n < 1 || (taps != 9 && taps != 4)) return 0;
    const int mt = (m + 31) / 32, nt = (n + 31) / 32;
    return (size_t)split_count(gs, mt, nt) * m * n * taps * sizeof(float);
}

static int wgrad_launch(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m, const float* l, int32_t n,
                        int32_t taps, float* dw, int32_t accumulate, void* ws, size_t ws_bytes, DepthMap dm,
                        dinv_stream_t stream) {
    if (int e = check_geom(gs)) return e;
    if (int e = check_geom(gl)) return e;
    DINV_REQUIRE(s && l && dw && ws, "null pointer");
    DINV_REQUIRE(m >= 1 && n >= 1 && (taps == 9 || taps == 4), "bad wgrad shape %d x %d x %d", m, n, taps);
    if (taps == 9)
        DINV_REQUIRE(gs->height == gl->height && gs->width == gl->width && gs->batch == gl->batch && gs->cs == gl->cs,
                     "3x3 weight gradient needs both tensors on one grid");
    else {
        DINV_REQUIRE(gl->height == 2 * gs->height && gl->width == 2 * gs->width, "2x2 weight gradient: the second tensor lives on the doubled grid");
        if (dm.dep_s == 0) DINV_REQUIRE(gs->batch == gl->batch, "2x2 weight gradient: batch mismatch");
        else
            DINV_REQUIRE(dm.dep_s >= 3 && dm.dep_l == 2 * (dm.dep_s - 2) + 2 && gs->batch % dm.dep_s == 0 && gl->batch % dm.dep_l == 0 &&
                         gs->batch / dm.dep_s == gl->batch / dm.dep_l && (dm.dz == 0 || dm.dz == 1), "2x2x2 weight gradient: bad depth pairing");
    }
    DINV_REQUIRE(ws_bytes >= dinv_conv_wgrad_workspace_bytes(gs, m, n, taps), "workspace too small");
    WgradArgs a{};
    a.gs = make_geom(*gs); a.gl = make_geom(*gl);
    a.s = s; a.l = l; a.part = reinterpret_cast<float*>(ws);
    a.M = m; a.N = n;
    a.cs_alloc = (m + 7) / 8 * 8; a.cl_alloc = (n + 7) / 8 * 8;
    a.mt = (m + 31) / 32; a.nt = (n + 31) / 32;
    a.nsplit = split_count(gs, a.mt, a.nt);
    a.per_split = (ceil_div(gs->np, a.nsplit) + 15) / 16 * 16;   // whole iterations of 4 k-steps (of 2 or 4 pixels)
    a.dm = dm;
    const int64_t units = (int64_t)a.mt * a.nt * a.nsplit;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(units, 4)), block(256);
    if (m <= 16 && n <= 16) {      // thin layer: 16 x 16 x 4 instruction (mt = nt = 1: one unit per pixel slice)
        if (taps == 9) hipLaunchKernelGGL((wgrad_thin_kernel<9, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_thin_kernel<4, true>), grid, block, 0, st, a);
    } else if (taps == 9) hipLaunchKernelGGL((wgrad_kernel<9, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<4, true>), grid, block, 0, st, a);
    DINV_CHECK_LAUNCH();
    const int64_t ne = (int64_t)m * n * taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(ne, 16)), dim3(256), 0, st, a.part, ne, a.nsplit,
                       accumulate, dw);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_wgrad(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m,
                               const float* l, int32_t n, int32_t taps, float* dw, int32_t accumulate, void* ws,
                               size_t ws_bytes, dinv_stream_t stream) {
    return wgrad_launch(gs, gl, s, m, l, n, taps, dw, accumulate, ws, ws_bytes, DepthMap{0, 0, 0}, stream);
}

extern "C" int dinv_conv_wgrad_3d(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m,
                                  const float* l, int32_t n, float* dw, int32_t accumulate, void* ws, size_t ws_bytes,
                                  int32_t depth_s, int32_t dz, dinv_stream_t stream) {
    DINV_REQUIRE(depth_s >= 1, "bad depth %d", depth_s);
    return wgrad_launch(gs, gl, s, m, l, n, 4, dw, accumulate, ws, ws_bytes, DepthMap{depth_s + 2, 2 * depth_s + 2, dz}, stream);
}

extern "C" int dinv_relu_backward(int64_t n, const float* act, float* grad, dinv_stream_t stream) {
    DINV_REQUIRE(n >= 0 && n % 4 == 0, "length must be a multiple of 4");
    if (n == 0) return 0;
    DINV_REQUIRE(act && grad, "null pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n / 4, 256), 8192);
    hipLaunchKernelGGL(relu_backward_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n / 4,
                       reinterpret_cast<const float4*>(act), reinterpret_cast<float4*>(grad));
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_relu_inplace(int64_t n, float* x, dinv_stream_t stream) {
    DINV_REQUIRE(n >= 0 && n % 4 == 0, "length must be a multiple of 4");
    if (n == 0) return 0;
    DINV_REQUIRE(x != nullptr, "null pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n / 4, 256), 8192);
    hipLaunchKernelGGL(relu_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n / 4,
                       reinterpret_cast<float4*>(x));
    DINV_CHECK_LAUNCH();
    return 0;
}
