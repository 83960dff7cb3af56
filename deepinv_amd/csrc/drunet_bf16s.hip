// DRUNet ResBlock 3x3 convolution on the BF16 matrix cores with a two-part exact operand split (gfx950).
//
// Operator: y = [relu](conv3x3(x)) (+ res1), stride 1, zero padding 1, no bias (deepinv/models/drunet.py:403-434),
// on the padded channel-blocked activation layout of drunet.hip.
//
// Arithmetic: every fp32 operand is written as x = xh + xl with bf16 parts (xh = bf16(x), xl = bf16(x - xh), both
// round-to-nearest-even; x - xh is exact), and a product a*b is evaluated as ah*bl + al*bh + ah*bh with fp32
// accumulation in v_mfma_f32_32x32x16_bf16: the dropped al*bl term and the rounding of the low parts are ~2^-17
// relative per operand, 2-4e-6 per layer against an fp64 convolution and 1.7e-6 on the DRUNet output with O(1)-gain
// ResBlock weights (tests/test_drunet_gpu.py, tests/test_golden_gpu.py).  The bf16 matrix pipe is 16x the fp32 one and
// co-issues with the vector ALU, so three products leave a 5.3x higher ceiling than the fp32 MFMA path.
//
// Structure (what differs from the first bf16 kernel, drunet_bf16.hip):
//   * K = 16 per MFMA = 2 channel blocks of 8 x ONE tap (lane half h supplies channel block 2s + h): no padding tap,
//     9 MFMAs per tile and 16 channels instead of 10;
//   * workgroup = 8 waves = 512 consecutive padded pixels x 64 couts, each wave a 64 x 64 tile (2 x 2 accumulators);
//   * the K loop runs over sub-steps (16 channels, one kernel row dy): a sub-step needs one 514-pixel row segment of
//     the input (split into hi / lo bf16 planes while it is staged) and 3 taps x 16 channels x 64 couts of pre-split
//     weights = 45 KB of LDS; two such stages alternate: the global loads of sub-step t+2 (into registers) and the
//     split + LDS write of sub-step t+1 run underneath the 36 MFMAs per wave of sub-step t; ONE barrier per sub-step.
//     Waves 4-7 stage before their MFMAs, waves 0-3 after them, so that on a SIMD (which holds waves w and w+4) one
//     wave's vector work overlaps the other wave's matrix work.
//
// Measured on MI355X (level 1: 128 channels, 160x160, B = 32; random data, so the matrix pipe runs at its power-limited
// ~2.0 GHz): this kernel 0.73-0.85 ms = 1.0 PFLOP/s executed (the best plain-HIP bf16 GEMMs reach 1.25-1.34 PFLOP/s on
// random operands, cdna_hip_programming.md 5); the same loop without staging 0.63 ms, staging without MFMAs 0.40 ms.
// Variants that were built and measured slower: loads two sub-steps ahead in a second register set (0.89 ms), a
// three-slot ring with the next sub-step's first operands read before the barrier and sched_barrier-pinned phases
// (0.89 ms: pinning keeps the split's vector work out of the MFMA blocks of the same wave); persistent workgroups that
// carry the staging pipeline across (pixel tile, cout tile) items, so that only the accumulator write-back separates
// two items (47.6 ms per DRUNet forward against 45.3 ms: the bookkeeping in the hot loop costs more than the ~9 us of
// pipeline fill + write-back per 512-pixel tile that a fit of t = tiles x (F + nsub x 2.0 us) shows).
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

constexpr int TPMAX = 512;              // pixels per workgroup of the widest variant (the geometry is padded for it)
constexpr int WUNITS = 2 * 3 * 2 * 64;  // ... of its weights: [plane][dx][cblk][co 64]
constexpr int NSTAGE = 2;
// NW waves per workgroup: 8 -> 512 pixels, one workgroup per CU (2 x 45 KB of LDS); 4 -> 256 pixels, TWO workgroups per
// CU (2 x 29 KB each) whose phases drift apart, so one's pipeline fill / epilogue runs under the other's MFMAs
template <int NW> struct Tile {
    static constexpr int TP = NW * 64;
    static constexpr int SEGX = TP + 2;            // staged row segment (one halo pixel on each side)
    static constexpr int XUNITS = 2 * 2 * SEGX;    // 16-byte units of a stage's activations: [plane][cblk][SEGX]
    static constexpr int STAGE = XUNITS + WUNITS;  // NW = 8: 2824 units = 45,184 bytes
};

struct SArgs {
    Geom g;
    const float* x;
    const uint4* w;    // [cout/64][cin/16][dy 3][plane 2][dx 3][cblk 2][co 64] x (8 bf16)
    float* y;
    const float* res1;
    int32_t cin, cblocks_valid;
    int32_t ntiles, ytiles, tiles_per_xcd;
};

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

struct Staged {   // one sub-step's share of a thread, between its global loads and its LDS writes
    float4 x0a, x0b, x1a, x1b, x2a, x2b;
    uint4 w0, w1, w2;
};

template <bool RELU, int NRES, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_bf16s_kernel(SArgs a) {
    constexpr int TP = Tile<NW>::TP, SEGX = Tile<NW>::SEGX, XUNITS = Tile<NW>::XUNITS, STAGE = Tile<NW>::STAGE;
    DINV_DYN_LDS(uint4, lds);   // [NSTAGE][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware order: consecutive pixel tiles (which share their halo rows) and the cout tiles of one pixel tile stay
    // on one XCD (observed placement: block b runs on XCD b % 8; speed only)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int64_t p0 = (int64_t)tile * TP;
    const int nsub = 3 * (a.cin / 16);

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- this thread's staging slots (the same in every sub-step).  Every thread moves two full chunks (its pixel
    // tid of both channel blocks) and one weight unit; the 4 chunks of the two tail pixels and the remaining 256
    // weight units are LOADED by every thread (clamped, redundant addresses) and only WRITTEN by the threads that own
    // them, so the loads are straight-line code without divergent branches.
    const int tq = tid & 3;                                  // tail chunk: channel block tq>>1, pixel TP + (tq&1)
    const int xg0 = tid * 8, xg1 = (int)(a.g.cs * 8) + tid * 8;                      // cs * 16 < 2^31 (launcher)
    const int xg2 = (int)((int64_t)(tq >> 1) * a.g.cs * 8) + (TP + (tq & 1)) * 8;
    const int xl0 = tid, xl1 = SEGX + tid, xl2 = (tq >> 1) * SEGX + TP + (tq & 1);    // + plane * 2 * SEGX
    const int wu1 = NW == 8 ? 512 + (tid & 255) : 256 + tid;    // NW = 4: three weight units per thread
    const uint4* wsrc0 = a.w + (int64_t)ty * nsub * WUNITS;
    const float* xsrc0 = a.x + (a.g.sl + p0 - 1) * 8;

    Staged rg;
    auto issue = [&](int t) {   // global loads of sub-step t into registers
        const int s = t / 3, dyi = t - 3 * s;
        const float* xs = xsrc0 + ((int64_t)(2 * s) * a.g.cs + (int64_t)(dyi - 1) * a.g.wp) * 8;
        rg.x0a = ld4(xs + xg0); rg.x0b = ld4(xs + xg0 + 4);
        rg.x1a = ld4(xs + xg1); rg.x1b = ld4(xs + xg1 + 4);
        rg.x2a = ld4(xs + xg2); rg.x2b = ld4(xs + xg2 + 4);
        const uint4* ws = wsrc0 + (int64_t)t * WUNITS;
        rg.w0 = ws[tid];
        rg.w1 = ws[wu1];
        if constexpr (NW == 4) rg.w2 = ws[512 + tid];
    };
    auto commit = [&](int t) {   // split + write the registers of sub-step t into its ring slot
        uint4* st = lds + (t % NSTAGE) * STAGE;
        uint4 hi, lo;
        split8(rg.x0a, rg.x0b, hi, lo);
        st[xl0] = hi; st[2 * SEGX + xl0] = lo;
        split8(rg.x1a, rg.x1b, hi, lo);
        st[xl1] = hi; st[2 * SEGX + xl1] = lo;
        split8(rg.x2a, rg.x2b, hi, lo);
        if (tid < 4) { st[xl2] = hi; st[2 * SEGX + xl2] = lo; }
        st[XUNITS + tid] = rg.w0;
        if (NW == 4 || tid < 256) st[XUNITS + wu1] = rg.w1;
        if constexpr (NW == 4) st[XUNITS + 512 + tid] = rg.w2;
    };

    // operand slots of this lane: A = weights (row = cout l31 of m-tile, k half = channel block lhi),
    //                             B = pixels  (col = pixel l31 of n-tile, k half = channel block lhi)
    const int aslot = lhi * 64 + l31;                       // + (plane*3 + dx)*128 + m*32
    const int bslot = lhi * SEGX + wv * 64 + l31;           // + plane*2*SEGX + n*32 + dx

    issue(0);
    commit(0);
    if (nsub > 1) issue(1);
    __syncthreads();
    for (int t = 0; t < nsub; ++t) {
        // registers hold sub-step t+1 (loaded one iteration ago); its slot was last read in iteration t-1, before
        // the barrier that ended that iteration
        // waves w and w+4 share a SIMD (dispatch order 0,2,1,3): one of each kind per SIMD; with four waves per
        // workgroup the SIMD partner belongs to the other resident workgroup, whose phase is unrelated
        const bool early = NW == 8 && (wv & 4) != 0;
        if (early) {
            if (t + 1 < nsub) commit(t + 1);
            if (t + 2 < nsub) issue(t + 2);
        }
        const uint4* st = lds + (t % NSTAGE) * STAGE;
        const uint4* xs = st;
        const uint4* ws = st + XUNITS;
        // operands of tap dx+1 are read while the 12 MFMAs of tap dx run; the three products go product-major, so that
        // consecutive MFMAs write different accumulators (a dependent 32x32 MFMA would wait for its predecessor)
        uint4 A[2][2][2], B[2][2][2];   // [buffer][tile][plane]
        auto rd = [&](int buf, int dx) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int m = 0; m < 2; ++m) A[buf][m][pl] = ws[(pl * 3 + dx) * 128 + aslot + m * 32];
#pragma unroll
                for (int n = 0; n < 2; ++n) B[buf][n][pl] = xs[pl * 2 * SEGX + bslot + n * 32 + dx];
            }
        };
        rd(0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int cur = dx & 1;
            if (dx < 2) rd(cur ^ 1, dx + 1);
            // smallest terms first: ah*bl, al*bh, ah*bh
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int pa = e == 1 ? 1 : 0, pb = e == 0 ? 1 : 0;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(A[cur][m][pa], B[cur][n][pb], acc[m][n]);
            }
        }
        if (!early) {
            if (t + 1 < nsub) commit(t + 1);
            if (t + 2 < nsub) issue(t + 2);
        }
        lds_barrier();   // slot t is consumed; slot t+1 is complete (global loads of t+2 stay in flight)
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

// ---------------------------------------------------------------------------------------------------------------------
// 2x2 stride-2 convolution (downsample_strideconv, deepinv/models/drunet.py:524-552) with the same operand split.
// K = 4 taps x Cin.  Every input pixel feeds exactly one output pixel and tap, so the B operand goes global -> registers
// -> split -> MFMA with no LDS stage (that stream IS the HBM traffic of the layer: the fp32 kernel of drunet.hip is
// bound by the fp32 matrix pipe, 0.66 ms at level 0; this one by HBM); the pre-split weights of a K step (one tap, 16
// channels, 64 couts: 4 KB) are shared by the four waves and double-buffered in LDS, one LDS-only barrier per K step.
struct DownSArgs {
    Geom gi, go;
    const float* x;    // [cin/8][gi.cs][8]
    const uint4* w;    // [tap 4][cin/16][plane 2][cblk 2][cout] x (8 bf16)
    float* y;          // [cout/8][go.cs][8]
    int32_t cin, cout;
    int64_t ntiles, per_xcd;
    DepthMap dm;       // 3-D: output image (half grid) -> input image (full grid) for depth tap dm.dz
};

template <bool ACC>    // ACC: y += conv (second depth tap of a 2x2x2 convolution)
__global__ __launch_bounds__(256) void down2x2_bf16s_kernel(DownSArgs a) {
    __shared__ uint4 wl[2][256];   // [stage][plane 2][cblk 2][co 64]
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
    // this thread's weight unit of a K step: plane (tid >> 7), channel block ((tid >> 6) & 1), cout tid & 63
    const int64_t wunit = (int64_t)(tid >> 6) * a.cout + co0 + (tid & 63);
    float4 ba[2], bb[2];
    uint4 wreg;
    auto issue = [&](int g) {
        const int tap = g / S, s = g - tap * S;
        const int64_t toff = (int64_t)(tap >> 1) * a.gi.wp + (tap & 1);
        wreg = a.w[(int64_t)g * 4 * a.cout + wunit];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* xb = a.x + ((int64_t)(2 * s + lhi) * a.gi.cs + ioff[n] + toff) * 8;
            ba[n] = ld4(xb);
            bb[n] = ld4(xb + 4);
        }
    };
    issue(0);
    wl[0][tid] = wreg;
    uint4 Bh[2], Bl[2];
    split8(ba[0], bb[0], Bh[0], Bl[0]);
    split8(ba[1], bb[1], Bh[1], Bl[1]);
    if (nsteps > 1) issue(1);
    __syncthreads();
    for (int g = 0; g < nsteps; ++g) {
        const uint4* ws = wl[g & 1];
        uint4 A[2][2];   // [plane][m]
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int m = 0; m < 2; ++m) A[pl][m] = ws[pl * 128 + lhi * 64 + m * 32 + l31];
        // smallest terms first (ah*bl, al*bh, ah*bh), product-major: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int pa = e == 1 ? 1 : 0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(A[pa][m], e == 0 ? Bl[n] : Bh[n], acc[m][n]);
        }
        if (g + 1 < nsteps) {   // registers hold step g+1 (loaded one iteration ago)
            wl[(g + 1) & 1][tid] = wreg;
            split8(ba[0], bb[0], Bh[0], Bl[0]);
            split8(ba[1], bb[1], Bh[1], Bl[1]);
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

template <bool SKIP>
__global__ __launch_bounds__(256) void up2x2_bf16s_kernel(UpSArgs a) {
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

extern "C" int dinv_conv3x3_bf16s(const dinv_act_geom* g, const float* x, const void* w_split, int32_t cin,
                                  int32_t cout, float* y, const float* res1, int32_t relu, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_split && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "bf16-split conv needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(g->cs >= g->sl + ceil_div(g->np, TPMAX) * TPMAX + g->wp + HALO, "channel-block stride too small for 512-pixel tiles");
    DINV_REQUIRE(g->cs * 16 < ((int64_t)1 << 31), "activation row too long for 32-bit staging offsets");
    SArgs a{make_geom(*g), x, reinterpret_cast<const uint4*>(w_split), y, res1, cin, cout / 8, 0, 0, 0};
    // 256-pixel workgroups (two per CU) by default: measured per DRUNet forward (56 ResBlock convs) 44.1 vs 45.3 ms at
    // B = 32 and 6.5 vs 7.4 ms at B = 4 (the per-GPU batch of the 8-GPU run); DINV_BF16S_WAVES=8 selects the 512-pixel form
    const char* env_nw = getenv("DINV_BF16S_WAVES");   // read per call: the emulated CPU tests switch it
    const int nw = env_nw && atoi(env_nw) == 8 ? 8 : 4;
    const int tp = nw * 64;
    a.ntiles = (int32_t)ceil_div(g->np, tp);
    a.ytiles = cout / 64;
    a.tiles_per_xcd = (int32_t)ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.tiles_per_xcd * a.ytiles * 8)), block((unsigned)tp);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static_assert((size_t)NSTAGE * Tile<8>::STAGE * sizeof(uint4) <= 160 * 1024, "ring does not fit the LDS");
    static_assert((size_t)2 * NSTAGE * Tile<4>::STAGE * sizeof(uint4) <= 160 * 1024, "two 4-wave workgroups must fit one CU");
#define DINV_S_LAUNCH(R, N, W)                                                                                    \
    do {                                                                                                          \
        constexpr size_t lds = (size_t)NSTAGE * Tile<W>::STAGE * sizeof(uint4);                                   \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16s_kernel<R, N, W>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        if (e_ != hipSuccess) return fail(100 + (int)e_, "hipFuncSetAttribute: %s", hipGetErrorString(e_));       \
        hipLaunchKernelGGL((conv3x3_bf16s_kernel<R, N, W>), grid, block, lds, st, a);                             \
    } while (0)
#define DINV_S_LAUNCH_W(R, N)          \
    do {                               \
        if (nw == 4) DINV_S_LAUNCH(R, N, 4); \
        else DINV_S_LAUNCH(R, N, 8);   \
    } while (0)
    if (relu) DINV_S_LAUNCH_W(true, 0);
    else if (res1) DINV_S_LAUNCH_W(false, 1);
    else DINV_S_LAUNCH_W(false, 0);
#undef DINV_S_LAUNCH_W
#undef DINV_S_LAUNCH
    DINV_CHECK_LAUNCH();
    return 0;
}

static int down2x2_bf16s_launch(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const void* w_split,
                                int32_t cin, int32_t cout, float* y, DepthMap dm, int accumulate, dinv_stream_t stream) {
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
    if (accumulate) hipLaunchKernelGGL(down2x2_bf16s_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(down2x2_bf16s_kernel<false>, grid, dim3(256), 0, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv_down2x2_bf16s(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                                       const void* w_split, int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    return down2x2_bf16s_launch(gin, gout, x, w_split, cin, cout, y, DepthMap{0, 0, 0}, 0, stream);
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
