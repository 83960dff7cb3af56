// DRUNet convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the 64 Conv2d / ConvTranspose2d launches of deepinv/models/drunet.py:200-210
// (3x3 s1 p1 no-bias convs, ReLU inside ResBlocks, residual adds, 2x2 s2 strided-conv down,
// 2x2 s2 transposed-conv up, skip adds) -- reference layers built at drunet.py:39-101,
// 323-434, 524-602.  fp32 in / fp32 accumulate: the f32 MFMA is bit-for-bit an fmaf chain, so
// parity with the reference's fp32 CPU path is summation-order only.
//
// Activation layout ("padded channel planes", CNHW): act[c][SL + b*PLANE + r*WP + col] with
// (r,col) in a zero-bordered (H+2) x WP frame, WP = roundup(W+2,4).  A 3x3 tap is then a
// constant shift (dy*WP + dx) of the flattened pixel index, so the convolution is a GEMM
//     Y[co][p] = sum_{tap,ci} Wt[tap][ci][co] * X[ci][p + shift(tap)]
// with M = Cout (A operand = weights), N = flattened padded pixels (B operand = activations,
// pixel-contiguous -> every global / LDS access is lane-linear), K = 9*Cin.  Border outputs are
// computed and then overwritten with exact zeros (select, not multiply) to keep the invariant.
//
// Workgroup tile: 256 pixels x 64 couts, 4 waves, each wave 64 px x 64 co = 2x2 MFMA tiles
// (64 accumulator VGPRs).  Per 8-channel chunk the WG stages 3 row segments x 8 ch x 264 px
// of activations (each loaded once, reused by the 3 dx taps straight from LDS) and the
// 9x8x64 weight block: 43.8 KB LDS -> 3 workgroups / CU, whose MFMA phases cover each other's
// staging.  144 MFMAs (9216 matrix-pipe cycles) per chunk per wave vs ~11 16-byte loads per
// thread: the kernel is matrix-pipe bound (roofline: 157.3 TFLOP/s fp32 MFMA).
#include "common.hpp"

using namespace dinv;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;       // pixels per workgroup
constexpr int KC = 8;         // input channels per staged chunk
constexpr int HALO = 4;       // staged halo (floats) on each side of a row segment, keeps 16 B alignment
constexpr int SEG = NT + 2 * HALO;

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

struct Conv3Args {
    Geom g;
    const float* x;    // [cin][cs]
    const float* x2;   // optional second input added on load (U-Net skip), or null
    const float* w;    // packed [cout/MT][cin/KC][9][KC][MT]
    float* y;          // [cout_valid][cs]
    const float* res1; // optional residuals added in the epilogue
    const float* res2;
    int32_t cin, cout_valid, relu;
};

template <int MREP, bool RELU, int NRES>
__global__ __launch_bounds__(256) void conv3x3_kernel(Conv3Args a) {
    constexpr int MT = 32 * MREP;
    __shared__ __attribute__((aligned(16))) float xs[3][KC][SEG];
    __shared__ __attribute__((aligned(16))) float ws[9][KC][MT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * NT;
    const int nchunks = a.cin / KC;
    f32x16 acc[MREP][2];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int XV = 3 * KC * (SEG / 4);  // float4 loads of activations per chunk (1584)
    constexpr int WV = 9 * KC * MT / 4;     // float4 loads of weights per chunk
    constexpr int XI = (XV + 255) / 256, WI = (WV + 255) / 256;
    const float4* wblk = reinterpret_cast<const float4*>(a.w) + (int64_t)blockIdx.y * nchunks * WV;

    // per-thread staging slots are fixed across chunks: precompute the (row, col) decomposition once
    int xoff_lds[XI];
    int64_t xoff_g[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = min(tid + it * 256, XV - 1);  // surplus slots re-load the last element (never stored)
        const int row = idx / (SEG / 4);
        const int c4 = idx - row * (SEG / 4);
        const int seg = row / KC, kc = row - seg * KC;
        xoff_lds[it] = (seg * KC + kc) * SEG + c4 * 4;
        xoff_g[it] = (int64_t)kc * a.g.cs + a.g.sl + p0 + (int64_t)(seg - 1) * a.g.wp - HALO + c4 * 4;
    }
    float* xs_flat = &xs[0][0][0];
    float4* ws_flat = reinterpret_cast<float4*>(&ws[0][0][0]);

    // software pipeline: the global loads of chunk ch+1 are in flight while chunk ch is on the matrix cores;
    // every load of a chunk is issued back to back (one exposed latency per chunk at most, not one per load).
    // One code location for the loads (ch = -1 is the prologue trip) keeps xv/wv4 in registers.
    float4 xv[XI];
    float4 w0, w1, w2, w3, w4;  // weight prefetch slots as scalars (an array here is left in scratch by hipcc)
    static_assert(WI <= 5, "weight prefetch slots");
    for (int ch = -1; ch < nchunks; ++ch) {
        if (ch >= 0) {
            __syncthreads();  // previous chunk's MFMA phase has consumed LDS
#pragma unroll
            for (int it = 0; it < XI; ++it)
                if (it < XI - 1 || tid + it * 256 < XV) *reinterpret_cast<float4*>(xs_flat + xoff_lds[it]) = xv[it];
#define DINV_WST(IT, REG) if (IT < WI && (IT < WI - 1 || tid + IT * 256 < WV)) ws_flat[tid + IT * 256] = REG;
            DINV_WST(0, w0) DINV_WST(1, w1) DINV_WST(2, w2) DINV_WST(3, w3) DINV_WST(4, w4)
#undef DINV_WST
            __syncthreads();
        }
        if (ch + 1 < nchunks) {
            // unpredicated loads (surplus slots re-load the last element and are never stored): a predicated
            // load into a loop-carried register forces an early vmcnt wait
            const float* xbase = a.x + (int64_t)(ch + 1) * KC * a.g.cs;
#pragma unroll
            for (int it = 0; it < XI; ++it) xv[it] = *reinterpret_cast<const float4*>(xbase + xoff_g[it]);
            if (a.x2) {
                const float* x2base = a.x2 + (int64_t)(ch + 1) * KC * a.g.cs;
#pragma unroll
                for (int it = 0; it < XI; ++it) {
                    const float4 u = *reinterpret_cast<const float4*>(x2base + xoff_g[it]);
                    xv[it].x += u.x; xv[it].y += u.y; xv[it].z += u.z; xv[it].w += u.w;
                }
            }
            const float4* wsrc = wblk + (int64_t)(ch + 1) * WV;
#define DINV_WLD(IT, REG) if (IT < WI) REG = wsrc[min(tid + IT * 256, WV - 1)];
            DINV_WLD(0, w0) DINV_WLD(1, w1) DINV_WLD(2, w2) DINV_WLD(3, w3) DINV_WLD(4, w4)
#undef DINV_WLD
        }
        if (ch < 0) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3 - 1;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const int k = kk * 2 + lhi;
                float av[MREP], bv[2];
#pragma unroll
                for (int m = 0; m < MREP; ++m) av[m] = ws[tap][k][m * 32 + l31];
#pragma unroll
                for (int n = 0; n < 2; ++n) bv[n] = xs[dy][k][wv * 64 + n * 32 + l31 + HALO + dx];
#pragma unroll
                for (int m = 0; m < MREP; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    // epilogue: D[i][j], j = lane&31 (pixel), i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
    const int co0 = blockIdx.y * MT;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t p = p0 + wv * 64 + n * 32 + l31;
        if (p >= a.g.np) continue;
        const bool in = interior(a.g, p);
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int cb = co0 + m * 32 + 4 * lhi;
            if (cb >= a.cout_valid) continue;  // (tail conv: only the first cout_valid planes exist)
            const int64_t ob = (int64_t)cb * a.g.cs + a.g.sl + p;
            float r1[16], r2[16];
            // residual planes have zero borders, so border lanes may load them too: 16 independent loads in flight
            if (NRES >= 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) r1[r] = a.res1[ob + (int64_t)((r & 3) + 8 * (r >> 2)) * a.g.cs];
            }
            if (NRES >= 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) r2[r] = a.res2[ob + (int64_t)((r & 3) + 8 * (r >> 2)) * a.g.cs];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb + (r & 3) + 8 * (r >> 2);
                if (MREP == 1 && co >= a.cout_valid) continue;
                float v = acc[m][n][r];
                if (RELU) v = fmaxf(v, 0.f);
                if (NRES >= 1) v += r1[r];
                if (NRES >= 2) v += r2[r];
                a.y[ob + (int64_t)((r & 3) + 8 * (r >> 2)) * a.g.cs] = in ? v : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 2x2 stride-2 convolution (downsample_strideconv, drunet.py:524-552): K = 4*Cin, B operand
// gathered straight from global/L2 (stride-2 pixels); 2.3 % of DRUNet's FLOPs.
struct DownArgs {
    Geom gi, go;
    const float* x;  // [cin][gi.cs]
    const float* w;  // [4][cin][cout]
    float* y;        // [cout][go.cs]
    int32_t cin, cout;
};

__global__ __launch_bounds__(256) void down2x2_kernel(DownArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * NT + wv * 64;
    const int co0 = blockIdx.y * 64;
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
    // K loop = 4 taps x cin, walked in groups of 4 k-steps (8 channels); the 16 operand loads of group g+1
    // are issued before the 16 MFMAs of group g (register double buffer)
    const int ngroups = 4 * (a.cin / 8);
    float av[2][4][2], bv[2][4][2];
    auto load_group = [&](int gidx, float (&A)[4][2], float (&B)[4][2]) {
        const int tap = gidx / (a.cin / 8), c0 = (gidx - tap * (a.cin / 8)) * 8;
        const int64_t toff = (int64_t)(tap >> 1) * a.gi.wp + (tap & 1);
        const float* wt = a.w + (int64_t)tap * a.cin * a.cout + co0 + l31;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int ci = c0 + 2 * sidx + lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) A[sidx][m] = wt[(int64_t)ci * a.cout + m * 32];
#pragma unroll
            for (int n = 0; n < 2; ++n) B[sidx][n] = a.x[(int64_t)ci * a.gi.cs + ioff[n] + toff];
        }
    };
    auto mma_group = [&](float (&A)[4][2], float (&B)[4][2]) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sidx][m], B[sidx][n], acc[m][n], 0, 0, 0);
    };
    load_group(0, av[0], bv[0]);
    for (int gidx = 0; gidx < ngroups; gidx += 2) {
        if (gidx + 1 < ngroups) load_group(gidx + 1, av[1], bv[1]);
        mma_group(av[0], bv[0]);
        if (gidx + 2 < ngroups) load_group(gidx + 2, av[0], bv[0]);
        if (gidx + 1 < ngroups) mma_group(av[1], bv[1]);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + n * 32 + l31;
        if (q >= a.go.np) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                a.y[(int64_t)co * a.go.cs + a.go.sl + q] = in[n] ? acc[m][n][r] : 0.f;
            }
    }
}

// 2x2 stride-2 transposed convolution (upsample_convtranspose, drunet.py:493-521): four
// parity-class GEMMs with K = Cin; input optionally the sum of two tensors (U-Net skip add).
struct UpArgs {
    Geom gi, go;
    const float* x;   // [cin][gi.cs]
    const float* x2;  // optional, added to x
    const float* w;   // [4][cin][cout]
    float* y;         // [cout][go.cs]
    int32_t cin, cout;
};

__global__ __launch_bounds__(256) void up2x2_kernel(UpArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * NT + wv * 64;
    const int co0 = blockIdx.y * 64;
    const int tap = blockIdx.z;
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
            const int pi = (int)(p - b * a.gi.plane);
            const int r = pi / a.gi.wp, c = pi - r * a.gi.wp;
            ooff[n] = a.go.sl + b * a.go.plane + (int64_t)(2 * (r - 1) + (tap >> 1) + 1) * a.go.wp + (2 * (c - 1) + (tap & 1) + 1);
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const float* wt = a.w + (int64_t)tap * a.cin * a.cout + co0 + l31;
    const int ngroups = a.cin / 8;
    float av[2][4][2], bv[2][4][2];
    auto load_group = [&](int gidx, float (&A)[4][2], float (&B)[4][2]) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int ci = gidx * 8 + 2 * sidx + lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) A[sidx][m] = wt[(int64_t)ci * a.cout + m * 32];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                B[sidx][n] = a.x[(int64_t)ci * a.gi.cs + ioff[n]];
                if (a.x2) B[sidx][n] += a.x2[(int64_t)ci * a.gi.cs + ioff[n]];
            }
        }
    };
    auto mma_group = [&](float (&A)[4][2], float (&B)[4][2]) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sidx][m], B[sidx][n], acc[m][n], 0, 0, 0);
    };
    load_group(0, av[0], bv[0]);
    for (int gidx = 0; gidx < ngroups; gidx += 2) {
        if (gidx + 1 < ngroups) load_group(gidx + 1, av[1], bv[1]);
        mma_group(av[0], bv[0]);
        if (gidx + 2 < ngroups) load_group(gidx + 2, av[0], bv[0]);
        if (gidx + 1 < ngroups) mma_group(av[1], bv[1]);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        if (!in[n]) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                a.y[(int64_t)co * a.go.cs + ooff[n]] = acc[m][n][r];
            }
    }
}

// ---------------------------------------------------------------------------------------
// NCHW <-> padded channel planes.  pack also writes the noise-level map channel
// (drunet.py:238-251: x = cat(x, sigma map)).
__global__ void pack_kernel(Geom g, const float* __restrict__ x, int cin, const float* __restrict__ sigma,
                            int sigma_mode, float sigma_scalar, float* __restrict__ act) {
    // grid: (ceil(w/64), h, (cin+1)*batch)
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int row = blockIdx.y;
    const int c = blockIdx.z / g.batch, b = blockIdx.z % g.batch;
    if (col >= g.w) return;
    float v;
    if (c < cin) {
        v = x[(((int64_t)b * cin + c) * g.h + row) * g.w + col];
    } else {
        // sigma_mode 0: scalar, 1: per-sample [B], 2: map [B,1,H,W]
        v = sigma_mode == 0 ? sigma_scalar : sigma_mode == 1 ? sigma[b] : sigma[((int64_t)b * g.h + row) * g.w + col];
    }
    act[(int64_t)c * g.cs + g.sl + (int64_t)b * g.plane + (int64_t)(row + 1) * g.wp + col + 1] = v;
}

__global__ void unpack_kernel(Geom g, const float* __restrict__ act, int cout, float* __restrict__ y) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int row = blockIdx.y;
    const int c = blockIdx.z / g.batch, b = blockIdx.z % g.batch;
    if (col >= g.w) return;
    y[(((int64_t)b * cout + c) * g.h + row) * g.w + col] =
        act[(int64_t)c * g.cs + g.sl + (int64_t)b * g.plane + (int64_t)(row + 1) * g.wp + col + 1];
}

int check_geom(const dinv_act_geom* g) {
    DINV_REQUIRE(g != nullptr, "null geometry");
    DINV_REQUIRE(g->batch >= 1 && g->height >= 1 && g->width >= 1, "bad geometry %dx%dx%d", g->batch, g->height, g->width);
    DINV_REQUIRE(g->wp % 4 == 0 && g->wp >= g->width + 2 && g->hp == g->height + 2, "bad padded frame");
    DINV_REQUIRE(g->sl % 4 == 0 && g->sl >= g->wp + HALO && g->cs % 4 == 0, "bad slack/stride");
    DINV_REQUIRE(g->cs >= g->sl + ceil_div(g->np, NT) * NT + g->wp + HALO, "channel stride too small");
    return 0;
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
    g->sl = g->wp + HALO;  // multiple of 4
    g->cs = g->sl + ceil_div(g->np, NT) * NT + g->wp + HALO;
    g->cs = (g->cs + 3) / 4 * 4;
    return 0;
}

extern "C" int dinv_conv3x3(const dinv_act_geom* g, const float* x, const float* x2, const float* w_packed,
                            int32_t cin, int32_t cout, int32_t cout_valid, float* y, const float* res1,
                            const float* res2, int32_t relu, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_packed && y, "null tensor pointer");
    DINV_REQUIRE(cin % KC == 0 && cin >= KC, "cin=%d must be a positive multiple of %d (pad with zero planes)", cin, KC);
    DINV_REQUIRE(cout % 32 == 0 && cout_valid >= 1 && cout_valid <= cout, "bad cout=%d/valid=%d", cout, cout_valid);
    Conv3Args a{make_geom(*g), x, x2, w_packed, y, res1, res2, cin, cout_valid, relu};
    const unsigned gx = (unsigned)ceil_div(g->np, NT);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nres = (res1 ? 1 : 0) + (res2 ? 1 : 0);
    DINV_REQUIRE(res1 || !res2, "res2 given without res1");
    if (cout % 64 == 0) {
        DINV_REQUIRE(cout_valid == cout, "partial output planes are only supported for 32-wide cout tiles");
        const dim3 grid(gx, cout / 64);
#define DINV_LAUNCH_C3(RELU, NRES) hipLaunchKernelGGL((conv3x3_kernel<2, RELU, NRES>), grid, dim3(256), 0, s, a)
        if (relu) { if (nres == 0) DINV_LAUNCH_C3(true, 0); else if (nres == 1) DINV_LAUNCH_C3(true, 1); else DINV_LAUNCH_C3(true, 2); }
        else      { if (nres == 0) DINV_LAUNCH_C3(false, 0); else if (nres == 1) DINV_LAUNCH_C3(false, 1); else DINV_LAUNCH_C3(false, 2); }
#undef DINV_LAUNCH_C3
    } else {
        const dim3 grid(gx, cout / 32);
#define DINV_LAUNCH_C3(RELU, NRES) hipLaunchKernelGGL((conv3x3_kernel<1, RELU, NRES>), grid, dim3(256), 0, s, a)
        if (relu) { if (nres == 0) DINV_LAUNCH_C3(true, 0); else if (nres == 1) DINV_LAUNCH_C3(true, 1); else DINV_LAUNCH_C3(true, 2); }
        else      { if (nres == 0) DINV_LAUNCH_C3(false, 0); else if (nres == 1) DINV_LAUNCH_C3(false, 1); else DINV_LAUNCH_C3(false, 2); }
#undef DINV_LAUNCH_C3
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_down2x2(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                 const float* w, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(gin)) return e;
    if (int e = check_geom(gout)) return e;
    DINV_REQUIRE(x && w && y, "null tensor pointer");
    DINV_REQUIRE(gin->height == 2 * gout->height && gin->width == 2 * gout->width && gin->batch == gout->batch,
                 "down2x2 geometry mismatch");
    DINV_REQUIRE(cin % 8 == 0 && cout % 64 == 0, "down2x2 needs cin %% 8 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DownArgs a{make_geom(*gin), make_geom(*gout), x, w, y, cin, cout};
    hipLaunchKernelGGL(down2x2_kernel, dim3((unsigned)ceil_div(gout->np, NT), cout / 64), dim3(256), 0,
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
    UpArgs a{make_geom(*gin), make_geom(*gout), x, x2, w, y, cin, cout};
    hipLaunchKernelGGL(up2x2_kernel, dim3((unsigned)ceil_div(gin->np, NT), cout / 64, 4), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_act_pack(const dinv_act_geom* g, const float* x, int32_t cin, const float* sigma,
                             int32_t sigma_mode, float sigma_scalar, float* act, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && act, "null tensor pointer");
    DINV_REQUIRE(sigma_mode == 0 || sigma != nullptr, "sigma tensor missing");
    DINV_REQUIRE((int64_t)(cin + 1) * g->batch <= 65535 && g->height <= 65535, "pack grid too large");
    hipLaunchKernelGGL(pack_kernel, dim3((g->width + 63) / 64, g->height, (cin + 1) * g->batch), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), make_geom(*g), x, cin, sigma, sigma_mode, sigma_scalar, act);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_act_unpack(const dinv_act_geom* g, const float* act, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(y && act, "null tensor pointer");
    DINV_REQUIRE((int64_t)cout * g->batch <= 65535 && g->height <= 65535, "unpack grid too large");
    hipLaunchKernelGGL(unpack_kernel, dim3((g->width + 63) / 64, g->height, cout * g->batch), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), make_geom(*g), act, cout, y);
    DINV_CHECK_LAUNCH();
    return 0;
}
