// DRUNet ResBlock 3x3 convolution on the BF16 matrix cores, two-part exact operand split, 2-D pixel tiles (gfx950).
//
// Operator: y = [relu](conv3x3(x)) (+ res1), stride 1, zero padding 1, no bias (deepinv/models/drunet.py:403-434), on the
// padded channel-blocked activation layout of drunet.hip.  Arithmetic as in round 2: every fp32 operand is x = xh + xl with
// xh = bf16(x), xl = bf16(x - xh) (both RNE, x - xh exact), a product is ah*bl + al*bh + ah*bh with fp32 accumulation in
// v_mfma_f32_32x32x16_bf16; the dropped al*bl term and the rounding of xl are <= 2^-16 |x| per operand (worst case,
// tests/test_emu_drunet.py::test_split_worst_case), 2-4e-6 per layer against an fp64 convolution on random data.
//
// What differs from the round-2 kernel (which staged one 258-pixel ROW segment per kernel row dy, i.e. loaded and split
// every input element 3 x Cout/64 times: 4.4-5 VALU instructions per MFMA, matrix pipe 55 % busy, profiles/pmc/r03_*):
//   * a workgroup owns a 2-D tile of TR x TC interior pixels (8x32, 16x16, 32x8 = 256 pixels, or half of that) of the image
//     stack (rows of all images are one flattened axis: the zero frame rows between images are ordinary rows whose results
//     are not stored), so ONE staged (TR+2) x (TC+2) halo region serves all nine taps of a 16-channel step: 1.3 staged
//     pixels per output pixel instead of 3.0, and no padded-frame columns are computed (level 3: 5 % wasted MFMAs, was 15 %);
//   * activations may arrive PRE-SPLIT (each pixel's 8-channel block as 8 bf16 high parts + 8 bf16 low parts in the same
//     32 bytes): the producer's epilogue splits once per element, the consumer stages by plain copies.  ResBlock conv1
//     writes its ReLU output t that way (t is consumed only by conv2, which would split it into exactly these parts: same
//     bits as before); conv2 reads it, adds the fp32 residual and writes fp32;
//   * the weight rows of a 32-cout MFMA tile are permuted (host pack) so that a lane's 16 accumulator registers are two
//     complete 8-channel blocks of one pixel: every epilogue access is 32 contiguous bytes per lane.
// Pipeline: the K loop runs over 16-channel steps.  ONE activation stage (the halo region, pre-split: 21-26 KB) and ONE weight
// stage (all nine taps x 16 channels x 64 couts, pre-split: 36 KB) live in LDS; the 108 MFMAs of a step run per wave without a
// barrier while the next step's operands (3 activation chunks + 9 weight units per thread, loaded at the top of the step)
// wait in registers; at the step boundary: barrier, 15 ds_write_b128 per thread, barrier.  Two workgroups per CU (59-63 KB).
// Measured (MI355X, B = 32, profiles/r03_split2d_*.jsonl): 0.73 / 0.63 / 0.58 / 0.60 ms per launch at the four DRUNet levels
// against 0.90 / 0.79 / 0.72 / 0.75 ms for the round-2 kernel.  Two other pipelines were built on the same tile code and
// measured within 3 % of this one (double-buffered activation stage + per-kernel-row weight stages + a barrier per row, 2
// workgroups per CU; single activation stage + per-row weight stages, 3 workgroups per CU): the kernel is no longer limited
// by its structure but by the chip's power budget - on zero-filled activations the same launch runs 1.6-1.7 PFLOP/s
// executed (0.45 ms at level 1), on random data the clock drops to ~1.9 GHz and it stops at 1.24-1.31 PFLOP/s, which is
// what plain-HIP bf16 GEMMs reach on this chip (cdna_hip_programming.md 5).  What is left is fewer MFMAs, not better-fed ones.
// A lesson recorded here: staging registers must be plain scalars - small arrays indexed inside lambdas were left in
// scratch memory by hipcc, which turned every staged load into load -> wait -> scratch store (58 % of wave time parked).
#include "drunet_split_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

constexpr int WUNITS = 2 * 3 * 2 * 64;   // 16-byte units of a sub-step's weights: [plane][dx][cblk][row 64]

template <int TC_, int NREP_> struct Tile2 {
    static constexpr int TC = TC_, NREP = NREP_;
    static constexpr int TP = 4 * 32 * NREP;             // pixels per workgroup (4 waves x NREP n-tiles of 32)
    static constexpr int TR = TP / TC;
    static constexpr int AR = TR + 2, CW = TC + 2;        // staged rows / columns (one halo pixel on each side)
    static constexpr int AW = TC == 8 ? 12 : TC + 2;      // LDS row pitch in units: conflict-free ds_read_b128 (see rot())
    static constexpr int APL = AR * AW;                   // units per (plane, channel block)
    static constexpr int ASTAGE = 4 * APL;                // [plane 2][cblk 2][AR][AW]
    static constexpr int NCH = 2 * AR * CW;               // 32-byte chunks (pixel x channel block) of a stage
    static constexpr int CPS = (NCH + 3 * 256 - 1) / (3 * 256);   // chunks per thread per sub-step
    static constexpr int LDS_UNITS = ASTAGE + 3 * WUNITS;       // one activation stage + the nine taps of a step
    static_assert(32 % TC == 0 && TP % TC == 0, "tile width must divide an MFMA n-tile");
    // lane -> pixel inside a 32-pixel n-tile: row l/TC, column (l%TC + rot) % TC.  The rotation makes the 16 lanes that
    // one ds_read_b128 cycle serves ({0-3,12-15,20-27}, {4-11,16-19,28-31}) hit 16 distinct 16-byte bank slots for the
    // pitches above (searched exhaustively: 32 -> none needed, 16 -> (0,14), 8 -> (0,4,4,0) with pitch 12).
    static __device__ __forceinline__ int rot(int row_in_ntile) {
        if constexpr (TC == 16) return row_in_ntile ? 14 : 0;
        else if constexpr (TC == 8) return (row_in_ntile == 1 || row_in_ntile == 2) ? 4 : 0;
        else return 0;
    }
};

struct S2Args {
    Geom g;
    const float* x;      // fp32 [cin/8][cs][8]  or pre-split [cin/8][cs][hi 8 bf16 | lo 8 bf16]
    const uint4* w;      // [cout/64][cin/16][dy 3][plane 2][dx 3][cblk 2][row 64] x (8 bf16), rows permuted (pack_split2d_weight)
    float* y;
    const float* res1;
    int32_t cin;
    int32_t ntc, ytiles, ntiles, tiles_per_xcd;
    int32_t rows;        // batch * hp flattened image rows
    // 3x3x3 mode (volumes as stacks of depth_s = D + 2 slices, a zero slice at each end): the K loop also runs over the
    // ndz = 3 depth taps, tap dz reading the input shifted by dz - 1 slices (dz_stride floats per slice); ndz = 1: plain 2-D
    int32_t ndz, depth_s;
    int64_t dz_stride;
};

template <bool IN_SPLIT, bool OUT_SPLIT, bool RELU, int NRES, int TC, int NREP>
__global__ __launch_bounds__(256, 2) void conv3x3_split2d_kernel(S2Args a) {
    using T = Tile2<TC, NREP>;
    constexpr int TR = T::TR, AR = T::AR, CW = T::CW, AW = T::AW, APL = T::APL, ASTAGE = T::ASTAGE, NCH = T::NCH;
    static_assert(T::CPS == 1, "one activation chunk per thread per third of the stage");
    DINV_DYN_LDS(uint4, lds);   // [ASTAGE] activations, then [3][WUNITS] weights
    uint4* const lds_w = lds + ASTAGE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware order: consecutive pixel tiles (shared halos) and the cout tiles of one pixel tile stay on one XCD
    // (observed placement: block b runs on XCD b % 8; speed only)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int tr_i = tile / a.ntc, tc_i = tile - tr_i * a.ntc;
    const int r0 = tr_i * TR, c0 = 1 + tc_i * TC;      // first interior row (flattened over images) / column of the tile
    const int nstep = a.cin / 16, nsub = 3 * nstep;
    const int nk = a.ndz * nstep;                       // K steps: depth taps x 16-channel steps

    f32x16 acc[2][NREP];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- staging slots of this thread (the same in every step): chunk q = (cblk, row, col) of the halo region
    int xoff[3], loff[3];     // global offset in floats (channel step 0), LDS unit of the high part
    bool own[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int q = k * 256 + tid;
        own[k] = q < NCH;
        if (!own[k]) q = NCH - 1;         // clamped redundant load, no write: straight-line code
        const int cb = q / (AR * CW), rem = q - cb * (AR * CW);
        const int row = rem / CW, col = rem - row * CW;
        xoff[k] = (int)(((int64_t)cb * a.g.cs + a.g.sl + (int64_t)(r0 - 1 + row) * a.g.wp + (c0 - 1 + col)) * 8);
        loff[k] = cb * APL + row * AW + col;
    }
    const uint4* wsrc0 = a.w + (int64_t)ty * a.ndz * nsub * WUNITS;
    const int64_t step_stride = (int64_t)2 * a.g.cs * 8;      // floats per 16-channel step

    // operand slots of this lane: A = weights (row l31 of m-tile, k half = channel block lhi),
    //                             B = pixels (pixel l31 of n-tile, k half = channel block lhi)
    const int aslot = lhi * 64 + l31;                       // + (plane*3 + dx)*128 + m*32
    int bslot[NREP], prow[NREP], pcol[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int q = (wv * NREP + n) * 32 + l31;
        const int tr = q / TC, j = q - tr * TC;
        int tc = j + T::rot(l31 / TC);
        if (tc >= TC) tc -= TC;
        prow[n] = tr; pcol[n] = tc;
        bslot[n] = lhi * APL + tr * AW + tc;                // + plane*2*APL + dy*AW + dx
    }

    // one sub-step's 36 MFMAs per wave: operands of tap dx+1 are read while the 12 MFMAs of tap dx run; smallest terms first
    // (ah*bl, al*bh, ah*bh), product-major, so that consecutive MFMAs hit different accumulators
    auto mma = [&](const uint4* xs, const uint4* ws) {
        uint4 A[2][2][2], B[2][NREP][2];   // [buffer][tile][plane]
        auto rd = [&](int buf, int dx) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int m = 0; m < 2; ++m) A[buf][m][pl] = ws[(pl * 3 + dx) * 128 + aslot + m * 32];
#pragma unroll
                for (int n = 0; n < NREP; ++n) B[buf][n][pl] = xs[pl * 2 * APL + bslot[n] + dx];
            }
        };
        rd(0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int cur = dx & 1;
            if (dx < 2) rd(cur ^ 1, dx + 1);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int pa = e == 1 ? 1 : 0, pb = e == 0 ? 1 : 0;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NREP; ++n) acc[m][n] = mfma_bf16(A[cur][m][pa], B[cur][n][pb], acc[m][n]);
            }
        }
    };

    // ---- staging registers of this thread: the next step's operands (plain scalars: see the header)
    uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;       // weight units tid + 256 k of the step: [dy 3][plane 2][dx 3][cblk 2][row 64]
    uint4 n0a, n0b, n1a, n1b, n2a, n2b;             // three activation chunks (32 bytes each; hi / lo after the split)
    auto ldw = [&](int s) {
        const uint4* ws = wsrc0 + (int64_t)(3 * s) * WUNITS;
        w0 = ws[tid]; w1 = ws[256 + tid]; w2 = ws[512 + tid];
        w3 = ws[768 + tid]; w4 = ws[1024 + tid]; w5 = ws[1280 + tid];
        w6 = ws[1536 + tid]; w7 = ws[1792 + tid]; w8 = ws[2048 + tid];
    };
    auto putw = [&]() {
        lds_w[tid] = w0; lds_w[256 + tid] = w1; lds_w[512 + tid] = w2;
        lds_w[768 + tid] = w3; lds_w[1024 + tid] = w4; lds_w[1280 + tid] = w5;
        lds_w[1536 + tid] = w6; lds_w[1792 + tid] = w7; lds_w[2048 + tid] = w8;
    };
    auto ld = [&](int k, int rd, uint4& ua, uint4& ub) {   // K step k = depth tap k / nstep, channel step k % nstep
        const int dz = k / nstep, s = k - dz * nstep;
        const float* p = a.x + (int64_t)s * step_stride + (int64_t)(dz - (a.ndz >> 1)) * a.dz_stride + xoff[rd];
        ua = ldu4(p);
        ub = ldu4(p + 4);
    };
    auto sp = [&](uint4& ua, uint4& ub) {
        if constexpr (!IN_SPLIT) {
            uint4 hi, lo;
            split8(as_f4(ua), as_f4(ub), hi, lo);
            ua = hi; ub = lo;
        }
    };
    auto put = [&](int rd, const uint4& hi, const uint4& lo) {
        if (own[rd]) {
            lds[loff[rd]] = hi;
            lds[2 * APL + loff[rd]] = lo;
        }
    };

    // ---- prologue: both stages of step 0 (one HBM round trip)
    ld(0, 0, n0a, n0b); ld(0, 1, n1a, n1b); ld(0, 2, n2a, n2b);
    ldw(0);
    sp(n0a, n0b); sp(n1a, n1b); sp(n2a, n2b);
    put(0, n0a, n0b); put(1, n1a, n1b); put(2, n2a, n2b);
    putw();
    __syncthreads();

    for (int s = 0; s < nk; ++s) {
        const bool more = s + 1 < nk;
        if (more) {    // every load of the next step up front: a whole step of MFMAs to land in
            ldw(s + 1);
            ld(s + 1, 0, n0a, n0b); ld(s + 1, 1, n1a, n1b); ld(s + 1, 2, n2a, n2b);
        }
        mma(lds, lds_w);
        mma(lds + AW, lds_w + WUNITS);
        if (more) { sp(n0a, n0b); sp(n1a, n1b); }
        mma(lds + 2 * AW, lds_w + 2 * WUNITS);
        if (more) {
            sp(n2a, n2b);
            lds_barrier();   // every read of both stages is done
            put(0, n0a, n0b); put(1, n1a, n1b); put(2, n2a, n2b);
            putw();
            lds_barrier();
        }
    }

    // ---- epilogue: register quads 2k, 2k+1 of acc[m][n] are channels 0..7 of block cb0 + 4m + 2k + lhi (row permutation of
    // pack_split2d_weight): 32 contiguous bytes per lane
    const int cb0 = ty * 8;
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int RR = r0 + prow[n], cc = c0 + pcol[n];
        if (RR >= a.rows || cc > a.g.w) continue;
        const int img = RR / a.g.hp, rr = RR - img * a.g.hp;
        bool in = rr >= 1 && rr <= a.g.h;           // frame rows between images stay zero
        if (a.depth_s > 0) {                        // ... and so do the two padding slices of every volume
            const int z = img % a.depth_s;
            in = in && z >= 1 && z <= a.depth_s - 2;
        }
        const int64_t opix = a.g.sl + (int64_t)RR * a.g.wp + cc;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float4 rs[NRES >= 1 ? 4 : 1];
            if (NRES >= 1) {      // the four residual loads of this m-tile first: independent 16-byte loads in flight
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float* rp = a.res1 + ((int64_t)(cb0 + 4 * m + 2 * k + lhi) * a.g.cs + opix) * 8;
                    rs[2 * k] = ld4(rp);
                    rs[2 * k + 1] = ld4(rp + 4);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t o = ((int64_t)(cb0 + 4 * m + 2 * k + lhi) * a.g.cs + opix) * 8;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[m][n][8 * k + e];
                if (RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (NRES == 1) {
                    const float4 ra = rs[2 * k], rb = rs[2 * k + 1];
                    v[0] += ra.x; v[1] += ra.y; v[2] += ra.z; v[3] += ra.w;
                    v[4] += rb.x; v[5] += rb.y; v[6] += rb.z; v[7] += rb.w;
                }
                if (NRES == 2) {      // gate: res1 is the ReLU output of the forward pass, its sign masks this data gradient
                    const float4 ra = rs[2 * k], rb = rs[2 * k + 1];
                    v[0] = ra.x > 0.f ? v[0] : 0.f; v[1] = ra.y > 0.f ? v[1] : 0.f; v[2] = ra.z > 0.f ? v[2] : 0.f; v[3] = ra.w > 0.f ? v[3] : 0.f;
                    v[4] = rb.x > 0.f ? v[4] : 0.f; v[5] = rb.y > 0.f ? v[5] : 0.f; v[6] = rb.z > 0.f ? v[6] : 0.f; v[7] = rb.w > 0.f ? v[7] : 0.f;
                }
                if (!in) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                if constexpr (OUT_SPLIT) {
                    uint4 hi, lo;
                    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
                    *reinterpret_cast<uint4*>(a.y + o) = hi;
                    *reinterpret_cast<uint4*>(a.y + o + 4) = lo;
                } else {
                    st4(a.y + o, make_float4(v[0], v[1], v[2], v[3]));
                    st4(a.y + o + 4, make_float4(v[4], v[5], v[6], v[7]));
                }
            }
        }
    }
}

template <bool IN_SPLIT, bool OUT_SPLIT, bool RELU, int NRES, int TC, int NREP>
int launch_s2(S2Args a, hipStream_t st) {
    using T = Tile2<TC, NREP>;
    constexpr size_t lds = (size_t)T::LDS_UNITS * sizeof(uint4);
    static_assert(2 * lds <= 160 * 1024, "two workgroups must fit one CU");
    const int ntr = (int)ceil_div(a.rows, T::TR);
    a.ntc = (int)ceil_div(a.g.w, TC);
    a.ntiles = ntr * a.ntc;
    a.tiles_per_xcd = (int32_t)ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.tiles_per_xcd * a.ytiles * 8)), block(256);
    auto kern = conv3x3_split2d_kernel<IN_SPLIT, OUT_SPLIT, RELU, NRES, TC, NREP>;
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e_ != hipSuccess) return fail(100 + (int)e_, "hipFuncSetAttribute: %s", hipGetErrorString(e_));
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

template <bool IN_SPLIT, bool OUT_SPLIT, bool RELU, int NRES>
int dispatch_tile(const S2Args& a, int tc, int nrep, hipStream_t st) {
#define DINV_S2(TCV, NR) return launch_s2<IN_SPLIT, OUT_SPLIT, RELU, NRES, TCV, NR>(a, st)
    if (nrep == 2) {
        if (tc == 32) DINV_S2(32, 2);
        if (tc == 16) DINV_S2(16, 2);
        DINV_S2(8, 2);
    }
    if (tc == 32) DINV_S2(32, 1);
    if (tc == 16) DINV_S2(16, 1);
    DINV_S2(8, 1);
#undef DINV_S2
}

}  // namespace

static int split_launch(const dinv_act_geom* g, const void* x, const void* w_split, int32_t cin, int32_t cout, void* y,
                        const float* res1, int32_t flags, int32_t depth, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_split && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "bf16-split conv needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    const bool in_split = flags & 1, out_split = flags & 2, relu = flags & 4, gate = flags & 8;
    DINV_REQUIRE((flags & ~0x30f) == 0 && ((flags >> 8) & 3) != 3, "unknown flags %d", flags);
    DINV_REQUIRE(!gate || (res1 && !relu && !in_split && !out_split), "gate mode: fp32 in / out, no relu, res1 = the activation whose sign gates the output");
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(!(out_split && res1), "a pre-split output carries no residual");
    DINV_REQUIRE(depth == 0 || (depth >= 1 && g->batch % (depth + 2) == 0), "batch %d is not a stack of volumes of %d + 2 slices",
                 g->batch, depth);
    // halo rows of the last row tile and the column overhang of a partial column tile must stay inside a channel block
    DINV_REQUIRE(g->cs >= g->sl + g->np + (int64_t)34 * g->wp + 64 + (depth ? 2 * g->plane : 0),
                 "channel-block stride too small for 2-D tiles (rebuild the geometry%s)", depth ? "; 3-D: one guard plane on each side" : "");
    DINV_REQUIRE(g->cs * 16 < ((int64_t)1 << 31), "activation row too long for 32-bit staging offsets");
    S2Args a{make_geom(*g), reinterpret_cast<const float*>(x), reinterpret_cast<const uint4*>(w_split),
             reinterpret_cast<float*>(y), res1, cin, 0, cout / 64, 0, 0, g->batch * g->hp,
             depth ? 3 : 1, depth ? depth + 2 : 0, g->plane * 8};
    // tile width: the widest of 32 / 16 / 8 that divides the image width (DRUNet levels: 320 -> 32, 160 -> 32, 80 -> 16,
    // 40 -> 8); tiles of 128 pixels only when 256-pixel tiles would fill less than three quarters of the 512 resident
    // workgroup slots (measured per level at B = 4 / 8, profiles/r03_tile_size_ab.jsonl: 240 workgroups -> 128 px wins by
    // 13 %, 420 / 440 -> 256 px wins by 16-18 %)
    const int tc = g->width % 32 == 0 ? 32 : (g->width % 16 == 0 ? 16 : 8);
    const int64_t n256 = ceil_div((int64_t)g->batch * g->hp, 256 / tc) * ceil_div(g->width, tc) * (cout / 64);
    const int nrep = ((flags >> 8) & 3) ? ((flags >> 8) & 3) : (n256 >= 384 ? 2 : 1);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (in_split) {
        if (relu) return dispatch_tile<true, false, true, 0>(a, tc, nrep, st);
        if (res1) return dispatch_tile<true, false, false, 1>(a, tc, nrep, st);
        return dispatch_tile<true, false, false, 0>(a, tc, nrep, st);
    }
    if (out_split) {
        if (relu) return dispatch_tile<false, true, true, 0>(a, tc, nrep, st);
        return dispatch_tile<false, true, false, 0>(a, tc, nrep, st);
    }
    if (relu) return dispatch_tile<false, false, true, 0>(a, tc, nrep, st);
    if (gate) return dispatch_tile<false, false, false, 2>(a, tc, nrep, st);
    if (res1) return dispatch_tile<false, false, false, 1>(a, tc, nrep, st);
    return dispatch_tile<false, false, false, 0>(a, tc, nrep, st);
}

// flags: bit 0 = x is pre-split, bit 1 = write y pre-split, bit 2 = relu; bits 8-9 = pixels per workgroup (0: chosen from the
// grid size, 1: 128, 2: 256)
extern "C" int dinv_conv3x3_split(const dinv_act_geom* g, const void* x, const void* w_split, int32_t cin, int32_t cout,
                                  void* y, const float* res1, int32_t flags, dinv_stream_t stream) {
    return split_launch(g, x, w_split, cin, cout, y, res1, flags, 0, stream);
}

// 3x3x3 convolution of volumes stored as stacks of depth + 2 slices (g->batch = volumes x (depth + 2), a zero slice at each
// end of every volume): ONE launch, the K loop runs over the three depth taps as well (tap dz reads the slices shifted by
// dz - 1), the accumulators never leave the registers; the two padding slices of the output are written as zeros.  x must
// have one plane of readable memory in front of its first slice and behind its last one (the tap views reach there).
// w_split: [cout/64][dz 3][cin/16][dy 3][plane 2][dx 3][cblk 2][row 64][8] bf16 (pack_split3d_weight).
extern "C" int dinv_conv3x3x3_split(const dinv_act_geom* g, const void* x, const void* w_split, int32_t cin, int32_t cout,
                                    void* y, const float* res1, int32_t flags, int32_t depth, dinv_stream_t stream) {
    DINV_REQUIRE(depth >= 1, "bad depth %d", depth);
    return split_launch(g, x, w_split, cin, cout, y, res1, flags, depth, stream);
}
