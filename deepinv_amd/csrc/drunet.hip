// DRUNet convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the 64 Conv2d / ConvTranspose2d launches of deepinv/models/drunet.py:200-210
// (3x3 s1 p1 no-bias convs, ReLU inside ResBlocks, residual adds, 2x2 s2 strided-conv down,
// 2x2 s2 transposed-conv up, skip adds) -- reference layers built at drunet.py:39-101,
// 323-434, 524-602.  fp32 in / fp32 accumulate: the f32 MFMA is bit-for-bit an fmaf chain, so
// parity with the reference's fp32 CPU path is summation-order only.
//
// Activation layout ("padded pixel rows, channels blocked by 8"):
//     act[c/8][SL + b*PLANE + r*WP + col][c%8]            (one block row = cs pixels x 8 floats)
// with (r,col) in a zero-bordered (H+2) x WP frame, WP = roundup(W+2,4).  A 3x3 tap is a constant
// shift (dy*WP + dx) of the flattened pixel index, so the convolution is a GEMM
//     Y[co][p] = sum_{tap,ci} Wt[tap][co][ci] * X[ci][p + shift(tap)]
// with M = Cout (A operand = weights), N = flattened padded pixels (B operand = activations), K = 9*Cin.
// The 8-channel blocking is chosen for the 32x32x2 MFMA: lane half h = lane>>5 supplies k = h, so a lane
// reads channels 4h..4h+3 of its pixel (B) / of its cout row (A) as ONE 16-byte LDS read that feeds four
// k-steps, and the D fragment (4 consecutive couts per lane per register quad) is ONE 16-byte global store.
// Border outputs are computed and then overwritten with exact zeros (select) to keep the zero-frame invariant.
//
// Workgroup tile: 256 pixels x 64 couts, 4 waves, each wave 64 px x 64 co = 2x2 MFMA tiles (64 accumulator
// registers).  Per 8-channel block the WG stages 3 row segments x 258 px x 8 ch (each activation loaded once,
// reused by the 3 dx taps out of LDS) and the 9x64x8 weight block into LDS rows padded to 12 floats
// (conflict-free ds_read_b128): 64.8 KB -> 2 workgroups / CU.  The global loads of block k+1 are in flight
// while block k is on the matrix cores (register software pipeline).  Per tap a wave issues 4 ds_read_b128
// and 16 MFMAs (1024 matrix-pipe cycles).  Roofline: 157.3 TFLOP/s fp32 MFMA.
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

struct Conv3Args {
    Geom g;
    const float* x;    // [cin/8][cs][8]
    const float* x2;   // optional second input added on load (U-Net skip), or null
    const float* w;    // packed [cout/MT][cin/8][9][MT][8]
    float* y;          // [cblocks_valid][cs][8]
    const float* res1; // optional residuals added in the epilogue
    const float* res2;
    int32_t cin, cblocks_valid, relu;
    int32_t ntiles, ytiles, tiles_per_xcd;  // pixel tiles, cout tiles, ceil(ntiles / 8)
    // 3x3x3 mode (volumes as stacks of depth_s = D + 2 slices, a zero slice at each end): the K loop also runs over the
    // ndz = 3 depth taps, tap dz reading the input shifted by dz - 1 slices (dz_stride floats per slice); ndz = 1: plain 2-D
    int32_t ndz, depth_s;
    int64_t dz_stride;
};

// frame pixels, pixels past the end and (3-D) the two padding slices of every volume are written as exact zeros
__device__ __forceinline__ bool writes_value(const Conv3Args& a, int64_t p) {
    if (!interior(a.g, p)) return false;
    if (a.depth_s > 0) {
        const int z = (int)((p / a.g.plane) % a.depth_s);
        return z >= 1 && z <= a.depth_s - 2;
    }
    return true;
}

template <int MREP, bool RELU, int NRES>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Args a) {
    constexpr int MT = 32 * MREP;
    __shared__ __attribute__((aligned(16))) float xs[3 * SEG * LP];
    __shared__ __attribute__((aligned(16))) float ws[9 * MT * LP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware block -> (pixel tile, cout tile) map.  Blocks b, b+8, b+16, ... run on the same XCD (observed
    // placement; used for speed only).  Each XCD walks a contiguous range of pixel tiles, all cout tiles of a
    // pixel tile back to back: neighbouring tiles share 2 of their 3 staged row segments and the cout tiles
    // share all of them, so those re-reads hit that XCD's L2 instead of the fabric (PMC: FETCH_SIZE was 3x the
    // activation bytes at grid.y = 1 and 11x at grid.y = 8 with the plain row-major block order).
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int64_t p0 = (int64_t)tile * NT;
    const int ncin = a.cin / KC;
    const int nchunks = a.ndz * ncin;        // K steps: depth taps x 8-channel blocks
    f32x16 acc[MREP][2];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int XSEG = SEG * 2;           // float4 per staged row segment (2 per pixel)
    constexpr int XV = 3 * XSEG;            // 1548 float4 of activations per block
    constexpr int WV = 9 * MT * 2;          // float4 of weights per block
    constexpr int XI = (XV + 255) / 256, WI = (WV + 255) / 256;
    static_assert(XI == 7 && WI <= 5, "prefetch slots");
    const float4* wblk = reinterpret_cast<const float4*>(a.w) + (int64_t)ty * nchunks * WV;

    // per-thread staging slots are fixed across blocks: (segment, float4-within-segment) decomposition once
    int xoff_lds[XI], xoff_g[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = min(tid + it * 256, XV - 1);  // surplus slots re-load the last element (never stored)
        const int seg = idx / XSEG, r = idx - seg * XSEG;
        xoff_lds[it] = (seg * SEG + (r >> 1)) * LP + (r & 1) * 4;
        xoff_g[it] = ((seg - 1) * a.g.wp - HALO) * 8 + r * 4;  // floats, relative to pixel p0 of the block row
    }
    const int64_t row0 = (a.g.sl + p0) * 8;  // float offset of pixel p0 inside a block row

    // software pipeline: the global loads of block ch+1 are in flight while block ch is on the matrix cores;
    // unpredicated loads + scalar prefetch slots keep everything in registers (hipcc leaves arrays of
    // loop-carried prefetch data in scratch, and predicated loads force early vmcnt waits)
    float4 xv0, xv1, xv2, xv3, xv4, xv5, xv6;
    float4 w0, w1, w2, w3, w4;
    for (int ch = -1; ch < nchunks; ++ch) {
        if (ch >= 0) {
            __syncthreads();  // previous block's MFMA phase has consumed LDS
#define DINV_XST(IT, REG) if (IT < XI - 1 || tid + IT * 256 < XV) st4(xs + xoff_lds[IT], REG);
            DINV_XST(0, xv0) DINV_XST(1, xv1) DINV_XST(2, xv2) DINV_XST(3, xv3) DINV_XST(4, xv4) DINV_XST(5, xv5) DINV_XST(6, xv6)
#undef DINV_XST
#define DINV_WST(IT, REG) if (IT < WI && (IT < WI - 1 || tid + IT * 256 < WV)) { const int i_ = tid + IT * 256; st4(ws + (i_ >> 1) * LP + (i_ & 1) * 4, REG); }
            DINV_WST(0, w0) DINV_WST(1, w1) DINV_WST(2, w2) DINV_WST(3, w3) DINV_WST(4, w4)
#undef DINV_WST
            __syncthreads();
        }
        if (ch + 1 < nchunks) {
            const int dz = (ch + 1) / ncin, cb = (ch + 1) - dz * ncin;
            const int64_t koff = (int64_t)cb * a.g.cs * 8 + row0 + (int64_t)(dz - (a.ndz >> 1)) * a.dz_stride;
            const float* xb = a.x + koff;
#define DINV_XLD(IT, REG) REG = ld4(xb + xoff_g[IT]);
            DINV_XLD(0, xv0) DINV_XLD(1, xv1) DINV_XLD(2, xv2) DINV_XLD(3, xv3) DINV_XLD(4, xv4) DINV_XLD(5, xv5) DINV_XLD(6, xv6)
#undef DINV_XLD
            if (a.x2) {
                const float* xb2 = a.x2 + koff;
#define DINV_XLD2(IT, REG) REG = add4(REG, ld4(xb2 + xoff_g[IT]));
                DINV_XLD2(0, xv0) DINV_XLD2(1, xv1) DINV_XLD2(2, xv2) DINV_XLD2(3, xv3) DINV_XLD2(4, xv4) DINV_XLD2(5, xv5) DINV_XLD2(6, xv6)
#undef DINV_XLD2
            }
            const float4* wsrc = wblk + (int64_t)(ch + 1) * WV;
#define DINV_WLD(IT, REG) if (IT < WI) REG = wsrc[min(tid + IT * 256, WV - 1)];
            DINV_WLD(0, w0) DINV_WLD(1, w1) DINV_WLD(2, w2) DINV_WLD(3, w3) DINV_WLD(4, w4)
#undef DINV_WLD
        }
        if (ch < 0) continue;
        // one tap = 4 ds_read_b128 (channels 4h..4h+3 of this lane's cout rows / pixels) + 16 MFMAs
        const float* xrow = xs + (wv * 64 + l31 + HALO) * LP + 4 * lhi;
        const float* wrow = ws + l31 * LP + 4 * lhi;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3 - 1;
            float4 av[MREP], bv[2];
#pragma unroll
            for (int m = 0; m < MREP; ++m) av[m] = ld4(wrow + (tap * MT + m * 32) * LP);
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[n] = ld4(xrow + (dy * SEG + n * 32 + dx) * LP);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MREP; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(av[m], s), comp(bv[n], s), acc[m][n], 0, 0, 0);
        }
    }
    const int cb0 = ty * (MT / 8);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t p = p0 + wv * 64 + n * 32 + l31;
        if (p >= a.g.np) continue;
        store_tile<MREP, RELU, NRES>(acc, n, a.g.sl + p, writes_value(a, p), cb0, a.cblocks_valid, a.g.cs, lhi, a.y, a.res1,
                                     a.res2);
    }
}

// THIN layers (cout <= 16: the 16-channel level of BASELINE config 4's 3-D DRUNet, which holds most of its voxels, and
// its 2-channel tail) on v_mfma_f32_16x16x4_f32: a 16-cout x 16-pixel tile per instruction instead of a 32-cout tile that
// would be half (or more) zero padding - same matrix-pipe rate, no wasted rows.  Same staging as conv3x3_kernel (3 row
// segments of 258 pixels x 8 channels + the 9 x 16 x 8 weight block per K step, register prefetch of the next step);
// a wave owns 64 pixels = 4 tiles; lane (i = lane & 15, q = lane >> 4) holds channels 2q, 2q+1 of cout row i / pixel i
// (one 8-byte LDS read feeds the two k-steps of an 8-channel block; rows are 12 floats apart: conflict-free).
// D: pixel = lane & 15, couts 4q .. 4q+3 = one 16-byte store into channel block q >> 1.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool RELU, int NRES>
__global__ __launch_bounds__(256) void conv3_thin_kernel(Conv3Args a) {
    constexpr int MT = 16;
    __shared__ __attribute__((aligned(16))) float xs[3 * SEG * LP];
    __shared__ __attribute__((aligned(16))) float ws[9 * MT * LP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, tl = blockIdx.x >> 3;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int64_t p0 = (int64_t)tile * NT;
    const int ncin = a.cin / KC;
    const int nchunks = a.ndz * ncin;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int XSEG = SEG * 2, XV = 3 * XSEG, WV = 9 * MT * 2;
    constexpr int XI = (XV + 255) / 256, WI = (WV + 255) / 256;
    static_assert(XI == 7 && WI == 2, "prefetch slots");
    const float4* wblk = reinterpret_cast<const float4*>(a.w);
    int xoff_lds[XI], xoff_g[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = min(tid + it * 256, XV - 1);
        const int seg = idx / XSEG, r = idx - seg * XSEG;
        xoff_lds[it] = (seg * SEG + (r >> 1)) * LP + (r & 1) * 4;
        xoff_g[it] = ((seg - 1) * a.g.wp - HALO) * 8 + r * 4;
    }
    const int64_t row0 = (a.g.sl + p0) * 8;
    float4 xv0, xv1, xv2, xv3, xv4, xv5, xv6, w0, w1;
    for (int ch = -1; ch < nchunks; ++ch) {
        if (ch >= 0) {
            __syncthreads();
#define DINV_XST(IT, REG) if (IT < XI - 1 || tid + IT * 256 < XV) st4(xs + xoff_lds[IT], REG);
            DINV_XST(0, xv0) DINV_XST(1, xv1) DINV_XST(2, xv2) DINV_XST(3, xv3) DINV_XST(4, xv4) DINV_XST(5, xv5) DINV_XST(6, xv6)
#undef DINV_XST
            st4(ws + (tid >> 1) * LP + (tid & 1) * 4, w0);
            if (tid + 256 < WV) st4(ws + ((tid + 256) >> 1) * LP + (tid & 1) * 4, w1);
            __syncthreads();
        }
        if (ch + 1 < nchunks) {
            const int dz = (ch + 1) / ncin, cb = (ch + 1) - dz * ncin;
            const float* xb = a.x + (int64_t)cb * a.g.cs * 8 + row0 + (int64_t)(dz - (a.ndz >> 1)) * a.dz_stride;
#define DINV_XLD(IT, REG) REG = ld4(xb + xoff_g[IT]);
            DINV_XLD(0, xv0) DINV_XLD(1, xv1) DINV_XLD(2, xv2) DINV_XLD(3, xv3) DINV_XLD(4, xv4) DINV_XLD(5, xv5) DINV_XLD(6, xv6)
#undef DINV_XLD
            const float4* wsrc = wblk + (int64_t)(ch + 1) * WV;
            w0 = wsrc[tid];
            w1 = wsrc[min(tid + 256, WV - 1)];
        }
        if (ch < 0) continue;
        const float* xrow = xs + (wv * 64 + l15 + HALO) * LP + 2 * lq;
        const float* wrow = ws + l15 * LP + 2 * lq;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3 - 1;
            const float2 av = *reinterpret_cast<const float2*>(wrow + tap * MT * LP);
            float2 bv[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv[n] = *reinterpret_cast<const float2*>(xrow + (dy * SEG + n * 16 + dx) * LP);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[n].x, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[n].y, acc[n], 0, 0, 0);
        }
    }
    const int cb = lq >> 1;
    if (cb >= a.cblocks_valid) return;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int64_t p = p0 + wv * 64 + n * 16 + l15;
        if (p >= a.g.np) continue;
        const int64_t o = ((int64_t)cb * a.g.cs + a.g.sl + p) * 8 + 4 * (lq & 1);
        float4 v = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
        if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (NRES == 1) v = add4(v, ld4(a.res1 + o));
        if (NRES == 2) {     // res1 is a GATE (the forward pass's ReLU output): ReLU backward in the data-gradient convolution's epilogue
            const float4 gt = ld4(a.res1 + o);
            v = make_float4(gt.x > 0.f ? v.x : 0.f, gt.y > 0.f ? v.y : 0.f, gt.z > 0.f ? v.z : 0.f, gt.w > 0.f ? v.w : 0.f);
        }
        if (!writes_value(a, p)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        st4(a.y + o, v);
    }
}

// ---------------------------------------------------------------------------------------
// 2x2 stride-2 convolution (downsample_strideconv, drunet.py:524-552): K = 4 taps x Cin; B operand gathered
// straight from global/L2 (one 16-byte load = 4 k-steps); 2.3 % of DRUNet's FLOPs together with the up-conv.
struct DownArgs {
    Geom gi, go;
    const float* x;  // [cin/8][gi.cs][8]
    const float* w;  // [4][cin/8][cout][8]
    float* y;        // [cout/8][go.cs][8]
    int32_t cin, cout;
    int64_t ntiles, per_xcd;   // workgroups, and workgroups per XCD of the XCD-aware order
};

__global__ __launch_bounds__(256) void down2x2_kernel(DownArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Consecutive block ids go to different XCDs.  Give each XCD a contiguous range of the (pixel tile, cout tile)
    // order with the cout tile fastest, so the cout/64 workgroups that read the same input pixels run next to each
    // other on one L2 instead of fetching them cout/64 times from HBM.
    const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (logical >= a.ntiles) return;
    const int nct = a.cout / 64;
    const int64_t q0 = (logical / nct) * NT + wv * 64;
    const int co0 = (int)(logical % nct) * 64;
    int64_t ioff[2];
    bool in[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + n * 32 + l31;
        in[n] = interior(a.go, q);
        ioff[n] = a.gi.sl;
        if (in[n]) {
            const int64_t b = q / a.go.plane;
            const int qi = (int)(q - b * a.go.plane);
            const int R = qi / a.go.wp, C = qi - R * a.go.wp;
            ioff[n] = a.gi.sl + b * a.gi.plane + (int64_t)(2 * (R - 1) + 1) * a.gi.wp + (2 * (C - 1) + 1);
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int ncb = a.cin / 8, ngroups = 4 * ncb;
    // group g = (tap, channel block): 4 operand loads feed 16 MFMAs; loads of group g+1 issued before the MFMAs of g
    float4 a00, a01, b00, b01, a10, a11, b10, b11;
#define DINV_LOAD_GROUP(GIDX, A0, A1, B0, B1)                                                                   \
    {                                                                                                           \
        const int tap_ = (GIDX) / ncb, cb_ = (GIDX) - tap_ * ncb;                                               \
        const int64_t toff_ = (int64_t)(tap_ >> 1) * a.gi.wp + (tap_ & 1);                                      \
        const float* wt_ = a.w + (((int64_t)tap_ * ncb + cb_) * a.cout + co0 + l31) * 8 + 4 * lhi;              \
        A0 = ld4(wt_); A1 = ld4(wt_ + 32 * 8);                                                                  \
        B0 = ld4(a.x + ((int64_t)cb_ * a.gi.cs + ioff[0] + toff_) * 8 + 4 * lhi);                               \
        B1 = ld4(a.x + ((int64_t)cb_ * a.gi.cs + ioff[1] + toff_) * 8 + 4 * lhi);                               \
    }
#define DINV_MMA_GROUP(A0, A1, B0, B1)                                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                             \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A0, s), comp(B0, s), acc[0][0], 0, 0, 0);         \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A0, s), comp(B1, s), acc[0][1], 0, 0, 0);         \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A1, s), comp(B0, s), acc[1][0], 0, 0, 0);         \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A1, s), comp(B1, s), acc[1][1], 0, 0, 0);         \
    }
    DINV_LOAD_GROUP(0, a00, a01, b00, b01)
    for (int gidx = 0; gidx < ngroups; gidx += 2) {
        if (gidx + 1 < ngroups) DINV_LOAD_GROUP(gidx + 1, a10, a11, b10, b11)
        DINV_MMA_GROUP(a00, a01, b00, b01)
        if (gidx + 2 < ngroups) DINV_LOAD_GROUP(gidx + 2, a00, a01, b00, b01)
        if (gidx + 1 < ngroups) DINV_MMA_GROUP(a10, a11, b10, b11)
    }
#undef DINV_LOAD_GROUP
#undef DINV_MMA_GROUP
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + n * 32 + l31;
        if (q >= a.go.np) continue;
        store_tile<2, false, 0>(acc, n, a.go.sl + q, in[n], co0 / 8, a.cout / 8, a.go.cs, lhi, a.y, nullptr, nullptr);
    }
}

// 2x2 stride-2 transposed convolution (upsample_convtranspose, drunet.py:493-521): four parity-class GEMMs with
// K = Cin; input optionally the sum of two tensors (U-Net skip add).  Only interior pixels are written; the zero
// frame of the output buffer is never touched.
struct UpArgs {
    Geom gi, go;
    const float* x;   // [cin/8][gi.cs][8]
    const float* x2;  // optional, added to x
    const float* w;   // [4][cin/8][cout][8]
    float* y;         // [cout/8][go.cs][8]
    int32_t cin, cout;
};

// Workgroup = 4 waves x 32 input pixels; every wave computes all four output parities (taps) and 64 couts of its
// pixels, so each input pixel (and its skip partner) is read from HBM exactly once: 8 accumulators (128
// registers), operands double buffered in registers straight from global memory (weights are L1/L2 resident).
__global__ __launch_bounds__(256) void up2x2_kernel(UpArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int co0 = blockIdx.y * 64;
    const int64_t p = (int64_t)blockIdx.x * 128 + wv * 32 + l31;
    const bool in = interior(a.gi, p);
    const int ncb = a.cin / 8;
    // per-lane byte offsets inside one channel block (buffer addressing: SGPR base + 32-bit lane offset)
    const uint32_t xoff = (uint32_t)((a.gi.sl + (in ? p : 0)) * 32 + 16 * lhi);
    const uint32_t woff = (uint32_t)((co0 + l31) * 32 + 16 * lhi);
    uint32_t yoff = 0xffffffffu;   // out of range: the stores of border lanes are dropped
    if (in) {
        const int64_t b = p / a.gi.plane;
        const int pi = (int)(p - b * a.gi.plane);
        const int r = pi / a.gi.wp, c = pi - r * a.gi.wp;
        yoff = (uint32_t)((a.go.sl + b * a.go.plane + (int64_t)(2 * (r - 1) + 1) * a.go.wp + (2 * (c - 1) + 1)) * 32 + 16 * lhi);
    }
    auto bld = [](const float* sbase, uint32_t off) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;
    float4 A0[4][2], A1[4][2], B0, B1;
#define DINV_LOAD_GROUP(CB, A, B)                                                                               \
    {                                                                                                           \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                         \
            const float* wt_ = a.w + ((int64_t)t * ncb + (CB)) * a.cout * 8;                                    \
            A[t][0] = bld(wt_, woff); A[t][1] = bld(wt_, woff + 32 * 32);                                       \
        }                                                                                                       \
        B = bld(a.x + (int64_t)(CB) * a.gi.cs * 8, xoff);                                                       \
        if (a.x2) B = add4(B, bld(a.x2 + (int64_t)(CB) * a.gi.cs * 8, xoff));                                   \
    }
#define DINV_MMA_GROUP(A, B)                                                                                    \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                               \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                             \
        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A[t][0], s), comp(B, s), acc[t][0], 0, 0, 0);     \
        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A[t][1], s), comp(B, s), acc[t][1], 0, 0, 0);     \
    }
    DINV_LOAD_GROUP(0, A0, B0)
    for (int cb = 0; cb < ncb; cb += 2) {
        if (cb + 1 < ncb) DINV_LOAD_GROUP(cb + 1, A1, B1)
        DINV_MMA_GROUP(A0, B0)
        if (cb + 2 < ncb) DINV_LOAD_GROUP(cb + 2, A0, B0)
        if (cb + 1 < ncb) DINV_MMA_GROUP(A1, B1)
    }
#undef DINV_LOAD_GROUP
#undef DINV_MMA_GROUP
    // lane holds co = co0 + 32 m + 8 rj + 4 lhi + (0..3) in registers 4 rj .. 4 rj + 3 of acc[t][m]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rj = 0; rj < 4; ++rj) {
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                a.y + (int64_t)(co0 / 8 + m * 4 + rj) * a.go.cs * 8, 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t o = yoff == 0xffffffffu ? yoff : yoff + (uint32_t)(((t >> 1) * a.go.wp + (t & 1)) * 32);
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const float4 v = make_float4(acc[t][m][4 * rj], acc[t][m][4 * rj + 1], acc[t][m][4 * rj + 2], acc[t][m][4 * rj + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yr, o, 0, 0);
            }
        }
}

// ---------------------------------------------------------------------------------------
// NCHW <-> blocked padded rows.  pack also writes the noise-level map channel (drunet.py:238-251:
// x = cat(x, sigma map)); one thread per pixel writes the whole first channel block (cin+1 <= 8).
__global__ void pack_kernel(Geom g, const float* __restrict__ x, int cin, const float* __restrict__ sigma,
                            int sigma_mode, float sigma_scalar, float* __restrict__ act) {
    // grid: (ceil(w/64), h, batch)
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int row = blockIdx.y;
    const int b = blockIdx.z;
    if (col >= g.w) return;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < cin ? x[(((int64_t)b * cin + c) * g.h + row) * g.w + col] : 0.f;
    // sigma_mode 0: scalar, 1: per-sample [B], 2: map [B,1,H,W]
    const float sg = sigma_mode == 0 ? sigma_scalar : sigma_mode == 1 ? sigma[b] : sigma[((int64_t)b * g.h + row) * g.w + col];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c == cin) v[c] = sg;
    float* o = act + (g.sl + (int64_t)b * g.plane + (int64_t)(row + 1) * g.wp + col + 1) * 8;
    st4(o, make_float4(v[0], v[1], v[2], v[3]));
    st4(o + 4, make_float4(v[4], v[5], v[6], v[7]));
}

__global__ void unpack_kernel(Geom g, const float* __restrict__ act, int cout, float* __restrict__ y) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int row = blockIdx.y;
    const int b = blockIdx.z;
    if (col >= g.w) return;
    const float* o = act + (g.sl + (int64_t)b * g.plane + (int64_t)(row + 1) * g.wp + col + 1) * 8;
    for (int c = 0; c < cout; ++c) y[(((int64_t)b * cout + c) * g.h + row) * g.w + col] = o[c];
}

}  // namespace

extern "C" int dinv_act_geom_init(int32_t batch, int32_t height, int32_t width, dinv_act_geom* g) {
    DINV_REQUIRE(g != nullptr, "null geometry");
    DINV_REQUIRE(batch >= 1 && height >= 1 && width >= 1, "bad geometry %dx%dx%d", batch, height, width);
    g->batch = batch; g->height = height; g->width = width;
    g->hp = height + 2;
    g->wp = (width + 2 + 3) / 4 * 4;
    g->plane = (int64_t)g->hp * g->wp;
    g->np = g->plane * batch;
    g->sl = g->wp + 4;
    // trailing slack: whole 512-pixel tiles (flattened 1-D kernels) and, for the 2-D tiles of drunet_split2d.hip, the halo
    // rows of the last row tile (up to 33 rows past the last frame) plus a partial column tile's overhang
    g->cs = g->sl + std::max<int64_t>(ceil_div(g->np, 2 * NT) * 2 * NT + g->wp + 4, g->np + (int64_t)34 * g->wp + 64);
    g->cs = (g->cs + 3) / 4 * 4;
    return 0;
}

static int conv3_launch(const dinv_act_geom* g, const float* x, const float* x2, const float* w_packed, int32_t cin,
                        int32_t cout, int32_t cout_valid, int32_t cout_tile, float* y, const float* res1, const float* res2,
                        int32_t relu, int32_t depth, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_packed && y, "null tensor pointer");
    DINV_REQUIRE(cin % KC == 0 && cin >= KC, "cin=%d must be a positive multiple of %d (pad with zero channels)", cin, KC);
    const bool thin = cout_tile == 16, gate = (relu & 2) != 0;
    if (thin) {
        DINV_REQUIRE(cout == 16 && cout_valid >= 1 && cout_valid <= 16 && !x2 && !res2, "thin kernel: cout padded to 16, no x2 / res2");
    } else {
        DINV_REQUIRE(cout % 32 == 0 && cout_valid >= 1 && cout_valid <= cout, "bad cout=%d/valid=%d", cout, cout_valid);
        DINV_REQUIRE((cout_tile == 32 || cout_tile == 64) && cout % cout_tile == 0, "cout_tile=%d must be 16, 32 or 64 and divide cout=%d", cout_tile, cout);
    }
    if (depth > 0) DINV_REQUIRE(g->batch % (depth + 2) == 0, "batch %d is not a whole number of (depth + 2)-slice volumes", g->batch);
    const int cbv = (cout_valid + 7) / 8;  // channel blocks that exist in the output buffer
    const int ntiles = (int)ceil_div(g->np, NT);
    const int tpx = (ntiles + 7) / 8;
    const int ytiles = cout / cout_tile;
    Conv3Args a{make_geom(*g), x, x2, w_packed, y, res1, res2, cin, cbv, relu, ntiles, ytiles, tpx,
                depth > 0 ? 3 : 1, depth > 0 ? depth + 2 : 0, g->plane * 8};
    const unsigned gx = (unsigned)(8 * tpx * ytiles);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nres = (res1 ? 1 : 0) + (res2 ? 1 : 0);
    DINV_REQUIRE(res1 || !res2, "res2 given without res1");
    const dim3 grid(gx);
    if (gate) {
        DINV_REQUIRE(thin && res1 && !(relu & 1), "gate (relu bit 1): thin kernel only, with the gating activation as res1, without ReLU");
        hipLaunchKernelGGL((conv3_thin_kernel<false, 2>), grid, dim3(256), 0, s, a);
    } else if (thin) {
        if (relu) { if (nres) hipLaunchKernelGGL((conv3_thin_kernel<true, 1>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((conv3_thin_kernel<true, 0>), grid, dim3(256), 0, s, a); }
        else      { if (nres) hipLaunchKernelGGL((conv3_thin_kernel<false, 1>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((conv3_thin_kernel<false, 0>), grid, dim3(256), 0, s, a); }
    } else if (cout_tile == 64) {
#define DINV_LAUNCH_C3(RELU, NRES) hipLaunchKernelGGL((conv3x3_kernel<2, RELU, NRES>), grid, dim3(256), 0, s, a)
        if (relu) { if (nres == 0) DINV_LAUNCH_C3(true, 0); else if (nres == 1) DINV_LAUNCH_C3(true, 1); else DINV_LAUNCH_C3(true, 2); }
        else      { if (nres == 0) DINV_LAUNCH_C3(false, 0); else if (nres == 1) DINV_LAUNCH_C3(false, 1); else DINV_LAUNCH_C3(false, 2); }
#undef DINV_LAUNCH_C3
    } else {
#define DINV_LAUNCH_C3(RELU, NRES) hipLaunchKernelGGL((conv3x3_kernel<1, RELU, NRES>), grid, dim3(256), 0, s, a)
        if (relu) { if (nres == 0) DINV_LAUNCH_C3(true, 0); else if (nres == 1) DINV_LAUNCH_C3(true, 1); else DINV_LAUNCH_C3(true, 2); }
        else      { if (nres == 0) DINV_LAUNCH_C3(false, 0); else if (nres == 1) DINV_LAUNCH_C3(false, 1); else DINV_LAUNCH_C3(false, 2); }
#undef DINV_LAUNCH_C3
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3x3(const dinv_act_geom* g, const float* x, const float* x2, const float* w_packed,
                            int32_t cin, int32_t cout, int32_t cout_valid, int32_t cout_tile, float* y,
                            const float* res1, const float* res2, int32_t relu, dinv_stream_t stream) {
    return conv3_launch(g, x, x2, w_packed, cin, cout, cout_valid, cout_tile, y, res1, res2, relu, 0, stream);
}

extern "C" int dinv_conv3x3x3(const dinv_act_geom* g, const float* x, const float* w_packed, int32_t cin, int32_t cout,
                              int32_t cout_valid, int32_t cout_tile, float* y, const float* res1, int32_t relu,
                              int32_t depth, dinv_stream_t stream) {
    DINV_REQUIRE(depth >= 1, "bad depth %d", depth);
    DINV_REQUIRE(g && g->cs >= g->sl + g->np + 2 * g->plane, "3-D views need a guard slice on each side of the buffer");
    return conv3_launch(g, x, nullptr, w_packed, cin, cout, cout_valid, cout_tile, y, res1, nullptr, relu, depth, stream);
}

extern "C" int dinv_conv_down2x2(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                 const float* w, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(gin)) return e;
    if (int e = check_geom(gout)) return e;
    DINV_REQUIRE(x && w && y, "null tensor pointer");
    DINV_REQUIRE(gin->height == 2 * gout->height && gin->width == 2 * gout->width && gin->batch == gout->batch,
                 "down2x2 geometry mismatch");
    DINV_REQUIRE(cin % 8 == 0 && cout % 64 == 0, "down2x2 needs cin %% 8 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DownArgs a{make_geom(*gin), make_geom(*gout), x, w, y, cin, cout, 0, 0};
    a.ntiles = ceil_div(gout->np, NT) * (cout / 64);
    a.per_xcd = ceil_div(a.ntiles, 8);
    hipLaunchKernelGGL(down2x2_kernel, dim3((unsigned)(a.per_xcd * 8)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_up2x2(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                               const float* w, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(gin)) return e;
    if (int e = check_geom(gout)) return e;
    DINV_REQUIRE(x && w && y, "null tensor pointer");
    DINV_REQUIRE(gout->height == 2 * gin->height && gout->width == 2 * gin->width && gin->batch == gout->batch,
                 "up2x2 geometry mismatch");
    DINV_REQUIRE(cin % 8 == 0 && cout % 64 == 0, "up2x2 needs cin %% 8 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(gout->cs * 32 < (1ll << 32) && (int64_t)cout * 32 < (1ll << 31),
                 "up2x2: one channel block must stay below 4 GB (32-bit buffer offsets)");
    UpArgs a{make_geom(*gin), make_geom(*gout), x, x2, w, y, cin, cout};
    hipLaunchKernelGGL(up2x2_kernel, dim3((unsigned)ceil_div(gin->np, 128), cout / 64), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_act_pack(const dinv_act_geom* g, const float* x, int32_t cin, const float* sigma,
                             int32_t sigma_mode, float sigma_scalar, float* act, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && act, "null tensor pointer");
    DINV_REQUIRE(cin >= 1 && cin + 1 <= 8, "pack supports up to 7 image channels (+ noise map) in the first block");
    DINV_REQUIRE(sigma_mode == 0 || sigma != nullptr, "sigma tensor missing");
    DINV_REQUIRE(g->batch <= 65535 && g->height <= 65535, "pack grid too large");
    hipLaunchKernelGGL(pack_kernel, dim3((g->width + 63) / 64, g->height, g->batch), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), make_geom(*g), x, cin, sigma, sigma_mode, sigma_scalar, act);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_act_unpack(const dinv_act_geom* g, const float* act, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(y && act, "null tensor pointer");
    DINV_REQUIRE(cout >= 1 && cout <= 8, "unpack reads the first channel block (<= 8 channels)");
    DINV_REQUIRE(g->batch <= 65535 && g->height <= 65535, "unpack grid too large");
    hipLaunchKernelGGL(unpack_kernel, dim3((g->width + 63) / 64, g->height, g->batch), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), make_geom(*g), act, cout, y);
    DINV_CHECK_LAUNCH();
    return 0;
}
