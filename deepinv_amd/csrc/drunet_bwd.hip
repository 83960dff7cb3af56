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
    float* part;          // [nparts][M][N][T]
    int32_t M, N;         // logical channel counts (rows / columns of dW)
    int32_t cs_alloc, cl_alloc;   // allocated channels (multiples of 8) of S and L
    int32_t mt, nt;       // tiles (32 wide; 16 wide for the thin-layer kernel, where mt = nt = 1)
    int32_t nparts;       // pixel slices per tile = workgroups per tile (each of the 4 waves takes a quarter of the slice)
    int64_t per_wave;     // pixels per wave (a multiple of 16)
    DepthMap dm;          // 3-D stride-2 layers: image of S (half grid) -> image of L (full grid)
    int64_t l_step, part_step;   // blockIdx.y = depth tap of a 3x3x3 layer: L shifted by l_step floats per tap, its own partials
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int TILE> struct WgAcc { typedef f32x16 type; };
template <> struct WgAcc<16> { typedef f32x4 type; };

constexpr int WG_U = 4;   // k-steps per loop iteration: all their operand loads are in flight together

// operands of WG_U k-steps starting at pixel p0 (PX pixels per k-step, lane part lk takes pixel lk of each).  The loads
// are unconditional and nothing looks at a loaded value before the MFMAs (a predicated load, or a select on its result,
// makes the compiler wait for it in the middle of the batch): pixels past the end of the slice read pixel 0 - the corner
// of the first frame, where S is zero like on every frame pixel.  Lanes of channels beyond the allocation read channel
// 0; their rows / columns of the tile are never written.
template <int T, bool STRIDE2, int PX>
__device__ __forceinline__ void wg_load(const WgradArgs& a, const float* sp, const float* lp, const int64_t (&off)[T],
                                        int64_t p0, int64_t pend, int lk, float (&sv)[WG_U], float (&lv)[WG_U][T]) {
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
        const int64_t pp = p0 + PX * u + lk;
        const int64_t p = pp < pend ? pp : 0;
        int64_t q = p;                                   // pixel of L that tap (0,0) of pixel p reads
        if (STRIDE2) {                                   // 32-bit index arithmetic (np < 2^31 is checked at launch)
            const unsigned pu = (unsigned)p, plane = (unsigned)a.gs.plane, wp = (unsigned)a.gs.wp;
            const unsigned b = pu / plane, pi = pu - b * plane;
            const int r = (int)(pi / wp), c = (int)(pi - (unsigned)r * wp);
            // frame pixels (and zero slices) of S are zero: send them to a valid address
            int bl = (int)b;
            if (a.dm.dep_s != 0) {
                const unsigned vol = b / (unsigned)a.dm.dep_s;
                const int z = (int)(b - vol * (unsigned)a.dm.dep_s);
                bl = (z < 1 || z > a.dm.dep_s - 2) ? -1 : (int)vol * a.dm.dep_l + 2 * (z - 1) + a.dm.dz + 1;
            }
            const bool in = bl >= 0 && r >= 1 && r <= a.gs.h && c >= 1 && c <= a.gs.w;
            q = in ? (int64_t)bl * a.gl.plane + (int64_t)(2 * (r - 1) + 1) * a.gl.wp + (2 * (c - 1) + 1) : 0;
        }
        sv[u] = sp[p * 8];
#pragma unroll
        for (int t = 0; t < T; ++t) lv[u][t] = lp[q * 8 + off[t]];
    }
}

// TILE = 32: v_mfma_f32_32x32x2_f32 (64 cycles, 2 pixels per instruction), a 32 x 32 (m, n) tile per wave.
// TILE = 16: v_mfma_f32_16x16x4_f32 (32 cycles, 4 pixels per instruction) for THIN layers (M, N <= 16: the 16-channel level
//            of BASELINE config 4's 3-D DRUNet, which holds most of its voxels) - the 32 x 32 instruction would spend
//            4x the matrix-pipe cycles per pixel on a tile that is three quarters padding, and was pipe-bound there.
// A workgroup = 4 waves on ONE tile and 4 consecutive pixel slices; the operands of iteration i + 1 are loaded while the
// MFMAs of iteration i run (register double buffer: one wave per SIMD already overlaps memory latency and matrix pipe);
// the four waves' tiles are added through LDS (fixed order) and written as one partial sum per workgroup.
template <int T, bool STRIDE2, int TILE>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr int PX = 64 / TILE, NACC = TILE == 32 ? 16 : 4;
    constexpr int TG = TILE == 32 ? 3 : T;               // taps per reduction round (48 KB / 36 KB of LDS)
    typedef typename WgAcc<TILE>::type Acc;
    __shared__ float red[4 * 3 * 16 * 64];
    static_assert(TG * NACC <= 48, "reduction round");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: slice bounds and loop control in SGPRs
    const int lc = lane & (TILE - 1), lk = lane / TILE;
    const int part = (int)(blockIdx.x % a.nparts), tile = (int)(blockIdx.x / a.nparts);
    const int m0 = (tile / a.nt) * TILE, n0 = (tile % a.nt) * TILE;
    const int cm = m0 + lc, cn = n0 + lc;
    const bool mv = cm < a.cs_alloc, nv = cn < a.cl_alloc;
    const float* sp = a.s + ((int64_t)(mv ? cm / 8 : 0) * a.gs.cs + a.gs.sl) * 8 + (mv ? cm % 8 : 0);
    const float* lp = a.l + (int64_t)blockIdx.y * a.l_step + ((int64_t)(nv ? cn / 8 : 0) * a.gl.cs + a.gl.sl) * 8 + (nv ? cn % 8 : 0);
    int64_t off[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        off[t] = STRIDE2 ? ((int64_t)(t >> 1) * a.gl.wp + (t & 1)) * 8 : ((int64_t)(t / 3 - 1) * a.gl.wp + (t % 3 - 1)) * 8;
    Acc acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[t][r] = 0.f;
    const int64_t pbeg = min(((int64_t)part * 4 + wv) * a.per_wave, a.gs.np);
    const int64_t pend = min(pbeg + a.per_wave, a.gs.np);
    float sv[WG_U], lv[WG_U][T], sn[WG_U], ln[WG_U][T];
    if (pbeg < pend) wg_load<T, STRIDE2, PX>(a, sp, lp, off, pbeg, pend, lk, sv, lv);
    // vmcnt(0): the loop is entered with nothing in flight, like every later iteration (the copy at its end waits for the
    // prefetch) - otherwise the compiler's wait-count bookkeeping makes the first MFMAs of each iteration wait for the
    // prefetch issued just before them
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int64_t p = pbeg; p < pend; p += PX * WG_U) {
        const bool more = p + PX * WG_U < pend;
        if (more) wg_load<T, STRIDE2, PX>(a, sp, lp, off, p + PX * WG_U, pend, lk, sn, ln);
#pragma unroll
        for (int u = 0; u < WG_U; ++u)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if constexpr (TILE == 32) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], lv[u][t], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[u], lv[u][t], acc[t], 0, 0, 0);
            }
        if (more) {
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                sv[u] = sn[u];
#pragma unroll
                for (int t = 0; t < T; ++t) lv[u][t] = ln[u][t];
            }
        }
    }
    // workgroup sum (waves in order) of TG taps at a time, then one partial tile per workgroup.
    // D[i][j] of lane l, register r:  32 x 32: j = l & 31, i = (r & 3) + 8 (r >> 2) + 4 (l >> 5);  16 x 16: j = l & 15, i = 4 (l >> 4) + r
    float* out = a.part + (int64_t)blockIdx.y * a.part_step + (int64_t)part * a.M * a.N * T;
    constexpr int RG = TG * NACC;                        // registers per lane per round
    for (int t0 = 0; t0 < T; t0 += TG) {
        if (t0) __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t < t0 || t >= t0 + TG) continue;
#pragma unroll
            for (int r = 0; r < NACC; ++r) red[(wv * RG + (t - t0) * NACC + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        for (int e = tid; e < RG * 64; e += 256) {
            const int j = e >> 6, l = e & 63;
            const int t = t0 + j / NACC, r = j % NACC;
            if (t >= T) continue;
            const float v = ((red[e] + red[RG * 64 + e]) + red[2 * RG * 64 + e]) + red[3 * RG * 64 + e];
            const int n = n0 + (l & (TILE - 1));
            const int m = m0 + (TILE == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + r);
            if (m < a.M && n < a.N) out[((int64_t)m * a.N + n) * T + t] = v;
        }
    }
}

// ---- 3x3 weight gradient with the operands staged through LDS (the form the 3x3 / 3x3x3 layers use; the stride-2 layers keep
// the register-gather kernel above).  A workgroup = 4 waves on one (m, n) tile and one pixel slice, walked in chunks of P
// pixels: the chunk of S (P pixels x TILE channels) and the three row segments of L it touches ((P + 2) pixels each, rows
// -1 / 0 / +1) are copied global -> registers -> LDS with 16-byte accesses in the global layout's own order (a pixel's
// TILE channels are contiguous: no transposition), double buffered - the loads of chunk c + 1 are in flight while chunk
// c is on the matrix cores, one barrier per chunk.  Each wave takes a quarter of the chunk's pixels for all 9 taps and reads
// its MFMA operands with conflict-free 4-byte LDS reads (TILE lanes = TILE consecutive floats, the lane parts = consecutive
// pixels).  Against the gather form: the 36 strided 4-byte global loads per 36 MFMAs (2-4 of 16 lanes per cache line) become
// 9 coalesced 16-byte loads per thread per chunk, and each L value is fetched once per row role instead of once per tap.
template <int TILE>
__global__ __launch_bounds__(256) void wgrad_lds_kernel(WgradArgs a) {
    constexpr int T = 9;
    constexpr int PX = 64 / TILE, NACC = TILE == 32 ? 16 : 4;
    constexpr int P = TILE == 32 ? 64 : 128;             // pixels per chunk
    constexpr int SEG = P + 2;                           // pixels per staged row segment of L
    constexpr int Q = TILE / 4;                          // float4 per staged pixel
    constexpr int STAGE = (P + 3 * SEG) * TILE;          // floats per stage: 33.5 KB / 33.2 KB
    constexpr int NS = P * Q, NL = 3 * SEG * Q;          // float4 of S / of L per chunk
    constexpr int SI = (NS + 255) / 256, LI = (NL + 255) / 256;
    constexpr int TG = TILE == 32 ? 3 : T;               // taps per reduction round
    constexpr int RG = TG * NACC;
    typedef typename WgAcc<TILE>::type Acc;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    static_assert(4 * RG * 64 <= 2 * STAGE, "reduction scratch aliases the stages");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & (TILE - 1), lk = lane / TILE;
    const int part = (int)(blockIdx.x % a.nparts), tile = (int)(blockIdx.x / a.nparts);
    const int m0 = (tile / a.nt) * TILE, n0 = (tile % a.nt) * TILE;
    const int64_t pbeg = min((int64_t)part * 4 * a.per_wave, a.gs.np);
    const int64_t pend = min(pbeg + 4 * a.per_wave, a.gs.np);
    const int nchunks = (int)((pend - pbeg + P - 1) / P);
    // staging slots of this thread (fixed across chunks): float offsets relative to the chunk's first pixel / LDS offsets
    int s_g[SI], s_l[SI], l_g[LI], l_l[LI];
    bool s_ok[SI], l_ok[LI];
#pragma unroll
    for (int k = 0; k < SI; ++k) {
        const int f = min(tid + 256 * k, NS - 1), pix = f / Q, q = f - pix * Q;
        const int ch = m0 + 4 * q, cb = ch < a.cs_alloc ? ch / 8 : 0;      // channel blocks beyond the allocation: block 0 (the
        s_ok[k] = tid + 256 * k < NS;                                       // tile rows they feed are never written)
        s_g[k] = (int)((cb * a.gs.cs + pix) * 8 + (ch & 4));
        s_l[k] = pix * TILE + 4 * q;
    }
#pragma unroll
    for (int k = 0; k < LI; ++k) {
        const int f = min(tid + 256 * k, NL - 1), rp = f / Q, q = f - rp * Q;
        const int rr = rp / SEG, sp = rp - rr * SEG;
        const int ch = n0 + 4 * q, cb = ch < a.cl_alloc ? ch / 8 : 0;
        l_ok[k] = tid + 256 * k < NL;
        l_g[k] = (int)((cb * a.gl.cs + (int64_t)(rr - 1) * a.gl.wp + sp - 1) * 8 + (ch & 4));
        l_l[k] = (P + rp) * TILE + 4 * q;
    }
    const float* sbase = a.s + a.gs.sl * 8;
    const float* lbase = a.l + (int64_t)blockIdx.y * a.l_step + a.gl.sl * 8;
    float4 sr[SI], lr[LI];
    // chunk c -> registers.  S pixels past the end of the slice are redirected to pixel 0 of the buffer (the corner of the
    // first frame: zero in S like every frame pixel); L needs no guard (finite data in the slack, multiplied by S = 0)
    auto fetch = [&](int c) {
        const int64_t p0 = pbeg + (int64_t)c * P;
#pragma unroll
        for (int k = 0; k < SI; ++k) {
            const int pix = s_l[k] / TILE;
            const int64_t po = p0 + pix < pend ? p0 * 8 + s_g[k] : (int64_t)s_g[k] - (int64_t)pix * 8;
            sr[k] = ld4(sbase + po);
        }
#pragma unroll
        for (int k = 0; k < LI; ++k) lr[k] = ld4(lbase + p0 * 8 + l_g[k]);
    };
    auto commit = [&](int stage) {
        float* st = lds + stage * STAGE;
#pragma unroll
        for (int k = 0; k < SI; ++k)
            if (s_ok[k]) st4(st + s_l[k], sr[k]);
#pragma unroll
        for (int k = 0; k < LI; ++k)
            if (l_ok[k]) st4(st + l_l[k], lr[k]);
    };
    Acc acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[t][r] = 0.f;
    if (nchunks > 0) {
        fetch(0);
        commit(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) fetch(c + 1);
        const float* st = lds + (c & 1) * STAGE;
        const float* sp = st + (wv * (P / 4) + lk) * TILE + lc;                  // S[pixel][channel]
        const float* lp = st + (P + wv * (P / 4) + lk + 1) * TILE + lc;          // L row segment 0, tap dx = 0
#pragma unroll
        for (int u = 0; u < P / 4 / PX; ++u) {
            const float av = sp[u * PX * TILE];
            float bv[T];
#pragma unroll
            for (int t = 0; t < T; ++t) bv[t] = lp[((t / 3) * SEG + u * PX + (t % 3 - 1)) * TILE];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if constexpr (TILE == 32) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) commit((c + 1) & 1);
        __syncthreads();
    }
    // workgroup sum (waves in order) of TG taps at a time, then one partial tile per workgroup (as in wgrad_kernel)
    float* red = lds;
    float* out = a.part + (int64_t)blockIdx.y * a.part_step + (int64_t)part * a.M * a.N * T;
    for (int t0 = 0; t0 < T; t0 += TG) {
        if (t0) __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t < t0 || t >= t0 + TG) continue;
#pragma unroll
            for (int r = 0; r < NACC; ++r) red[(wv * RG + (t - t0) * NACC + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        for (int e = tid; e < RG * 64; e += 256) {
            const int j = e >> 6, l = e & 63;
            const int t = t0 + j / NACC, r = j % NACC;
            if (t >= T) continue;
            const float v = ((red[e] + red[RG * 64 + e]) + red[2 * RG * 64 + e]) + red[3 * RG * 64 + e];
            const int n = n0 + (l & (TILE - 1));
            const int m = m0 + (TILE == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + r);
            if (m < a.M && n < a.N) out[((int64_t)m * a.N + n) * T + t] = v;
        }
    }
}

// dw[e] (+)= sum over slices in a fixed order: 16 lanes share an element (lane j adds slices j, j + 16, ... in order,
// then the 16 partial sums are added in lane order), 16 elements per workgroup.  blockIdx.y = depth tap z of nz: its own
// partials, element (mn, t) lands at dw[mn][z][t]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t n, int nsplit, int accumulate,
                                                           float* __restrict__ dw, int64_t part_step, int taps) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + el;
    part += (int64_t)blockIdx.y * part_step;
    float v = 0.f;
    if (e < n)
        for (int s = grp; s < nsplit; s += 16) v += part[(int64_t)s * n + e];
    red[grp][el] = v;
    __syncthreads();
    if (grp == 0 && e < n) {
        const int64_t d = gridDim.y == 1 ? e : ((e / taps) * gridDim.y + blockIdx.y) * taps + e % taps;
        float t = accumulate ? dw[d] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][el];
        dw[d] = t;
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

int part_count(const dinv_act_geom* gs, int m, int n) {
    // workgroups (4 waves each) per tile: ~2 waves per SIMD over the chip, wave slices of at least 128 pixels
    const bool thin = m <= 16 && n <= 16;
    const int tiles = thin ? 1 : ((m + 31) / 32) * ((n + 31) / 32);
    const int64_t want = std::max<int64_t>(1, 512 / tiles);
    const int64_t maxs = std::max<int64_t>(1, gs->np / 512);
    return (int)std::min<int64_t>(want, maxs);
}

}  // namespace

extern "C" size_t dinv_conv_wgrad_workspace_bytes(const dinv_act_geom* gs, int32_t m, int32_t n, int32_t taps) {
    if (!gs || m < 1 || n < 1 || (taps != 9 && taps != 4)) return 0;
    return (size_t)part_count(gs, m, n) * m * n * taps * sizeof(float);
}

static int wgrad_launch(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m, const float* l, int32_t n,
                        int32_t taps, float* dw, int32_t accumulate, void* ws, size_t ws_bytes, DepthMap dm,
                        dinv_stream_t stream, int ndepth = 1, int64_t l_step = 0) {
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
    DINV_REQUIRE(ws_bytes >= ndepth * dinv_conv_wgrad_workspace_bytes(gs, m, n, taps), "workspace too small");
    DINV_REQUIRE(gs->np < (int64_t)1 << 31 && gl->np < (int64_t)1 << 31, "more than 2^31 padded pixels");
    WgradArgs a{};
    a.gs = make_geom(*gs); a.gl = make_geom(*gl);
    a.s = s; a.l = l; a.part = reinterpret_cast<float*>(ws);
    a.M = m; a.N = n;
    a.cs_alloc = (m + 7) / 8 * 8; a.cl_alloc = (n + 7) / 8 * 8;
    const bool thin = m <= 16 && n <= 16;          // 16 x 16 x 4 instruction, one tile
    a.mt = thin ? 1 : (m + 31) / 32; a.nt = thin ? 1 : (n + 31) / 32;
    a.nparts = part_count(gs, m, n);
    a.per_wave = (ceil_div(gs->np, (int64_t)4 * a.nparts) + 15) / 16 * 16;   // whole iterations of 4 k-steps (of 2 or 4 pixels)
    a.dm = dm;
    a.l_step = l_step; a.part_step = (int64_t)a.nparts * m * n * taps;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(a.mt * a.nt * a.nparts), (unsigned)ndepth), block(256);
    if (taps == 9) {
        DINV_REQUIRE((int64_t)(a.cs_alloc / 8 + 1) * gs->cs * 8 < ((int64_t)1 << 31) && (int64_t)(a.cl_alloc / 8 + 1) * gl->cs * 8 < ((int64_t)1 << 31),
                     "activation buffers too large for 32-bit staging offsets");
        if (thin) hipLaunchKernelGGL((wgrad_lds_kernel<16>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_lds_kernel<32>), grid, block, 0, st, a);
    } else if (thin) hipLaunchKernelGGL((wgrad_kernel<4, true, 16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<4, true, 32>), grid, block, 0, st, a);
    DINV_CHECK_LAUNCH();
    const int64_t ne = (int64_t)m * n * taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(ne, 16), (unsigned)ndepth), dim3(256), 0, st, a.part, ne, a.nparts,
                       accumulate, dw, a.part_step, taps);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_wgrad(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m,
                               const float* l, int32_t n, int32_t taps, float* dw, int32_t accumulate, void* ws,
                               size_t ws_bytes, dinv_stream_t stream) {
    return wgrad_launch(gs, gl, s, m, l, n, taps, dw, accumulate, ws, ws_bytes, DepthMap{0, 0, 0}, stream);
}

extern "C" int dinv_conv_wgrad_3x3x3(const dinv_act_geom* g, const float* s, int32_t m, const float* l, int32_t n,
                                     int64_t depth_stride, float* dw, int32_t accumulate, void* ws, size_t ws_bytes,
                                     dinv_stream_t stream) {
    DINV_REQUIRE(depth_stride > 0, "bad depth stride %lld", (long long)depth_stride);
    return wgrad_launch(g, g, s, m, l, n, 9, dw, accumulate, ws, ws_bytes, DepthMap{0, 0, 0}, stream, 3, depth_stride);
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
