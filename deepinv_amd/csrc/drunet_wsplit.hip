// DRUNet ResBlock 3x3 convolution as Winograd F(2,3) along image rows on the BF16 matrix cores, two-part exact operand
// split (gfx950).
//
// Operator: y = [relu](conv3x3(x)) (+ res1), stride 1, zero padding 1, no bias (deepinv/models/drunet.py:403-434), fp32 in and
// out on the padded channel-blocked activation layout of drunet.hip.  Why: drunet_split2d.hip evaluates every fp32 product
// as three bf16 products and sits on the chip's power limit (1.1 PFLOP/s executed, profiles/r03_zero_data_test.jsonl) - what
// is left is issuing fewer matrix instructions.  F(2,3) along the columns of an output pixel PAIR (2j, 2j+1) of one row:
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3            (d0..d3 = input columns 2j-1 .. 2j+2)
//     U0 = g0,  U1 = (g0 + g1 + g2)/2,  U2 = (g0 - g1 + g2)/2,  U3 = g2     (g = the three dx taps of kernel row dy)
//     M_k = sum over (ci, dy) of U_k[dy] * V_k[row + dy],     y(2j) = M0 + M1 + M2,   y(2j+1) = M1 - M2 - M3
// = 12 multiplies per pair, channel pair and kernel row instead of 18: 1.5x fewer MFMAs.  U is formed in fp64 on the host
// and split (hi = bf16, lo = bf16 of the rest) when packed; V is formed in fp32 from the staged pixels and split on the
// fly; every product is Uh*Vl + Ul*Vh + Uh*Vh with fp32 accumulation in v_mfma_f32_32x32x16_bf16, as in the direct kernel.
// Error: the same 2^-16 per operand, relative to |U| (x) |V| instead of |g| (x) |d|: 4.0e-6 per layer against an fp64
// convolution on random data on the hardware (direct form: 3.1e-6; profiles/r03_wsplit_variants.jsonl), 2e-5 asserted in
// tests/test_emu_drunet.py and tests/test_drunet_gpu.py.
//
// Work decomposition (one workgroup = 4 waves = 256 output pixels = 128 pairs x 64 couts, two workgroups per CU):
//   * wave k owns Winograd point k for the whole tile: accumulators M_k[2 cout tiles of 32][4 pair tiles of 32] = 128
//     registers.  Its A operands (U_k) are used by no other wave, so they never pass through LDS: each lane loads its
//     16 bytes of the MFMA fragment straight from the packed weights (L2-resident), one kernel row ahead of its use.
//   * V (all four points, high and low parts) of a 16-channel step is built by the whole workgroup from the halo region
//     ((TR+2) rows x (TC+2) columns: thread = one (channel block, row, pair)), written to LDS once (double buffered: the
//     next step's V is written while this step's is read; one barrier per step) and read by wave k only for point k:
//     0.33 ds_read_b128 per MFMA (direct kernel: 0.67) and no weight stage at all - LDS is nearly idle.
//   * epilogue: the four waves exchange M_k through LDS (one 32-cout tile at a time, lane-linear 16-byte accesses), wave w
//     finishes pair tile w: both output pixels of a pair are one lane's 64 contiguous bytes per channel block.
// Measured (MI355X, B = 32, conv1 / conv2 of a ResBlock, direct kernel in brackets): 0.66 / 0.73 (0.83 / 0.73), 0.54 / 0.58
// (0.70 / 0.64), 0.48 / 0.49 (0.65 / 0.59), 0.45 / 0.47 (0.65 / 0.60) ms per launch at the four DRUNet levels; the matrix pipe is
// busy 50 % of the cycles (profiles/pmc/r03_conv_wsplit_sq.csv).  Diagnostic builds: a quarter of the time is the
// price of feeding the V stage, a sixth the fragment loads - as issue cost inside the in-order waves (forcing the loads to hit
// L1, a longer lead, one load per MFMA gap, halving the activation requests through DPP neighbour exchange: all within 0-5 %);
// the epilogue exchange and the barrier are free; four accumulators in rotation instead of two change nothing.
#include <type_traits>

#include "drunet_split_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

constexpr int WSUB = 4 * 2 * 2 * 64;   // 16-byte units of one (16-channel step, kernel row) of packed weights: [point 4][m 2][plane 2][lane 64]

template <int TC_> struct TileW {
    static constexpr int TC = TC_, PC = TC / 2;          // tile columns, column pairs per row
    static constexpr int TP = 256, TR = TP / TC, AR = TR + 2;
    static constexpr int VPL = AR * PC;                   // units per (plane, channel block, point)
    static constexpr int VSTAGE = 16 * VPL;               // [plane 2][cblk 2][point 4][AR][PC]
    static constexpr int NTASK = 2 * AR * PC;             // (cblk, row, pair) transform tasks per step; 256 < NTASK <= 320
    static constexpr int XUNITS = 4 * 4 * 4 * 64;         // epilogue exchange of one cout tile: [point][pair tile][quad][lane]
    static constexpr int LDS_UNITS = 2 * VSTAGE > XUNITS ? 2 * VSTAGE : XUNITS;
    static_assert(32 % PC == 0 && NTASK > 256 && NTASK <= 320, "tile shape");
};

struct WsArgs {
    Geom g;
    const float* x;      // fp32 [cin/8][cs][8]
    const uint4* w;      // [cout/64][cin/16][dy 3][point 4][m 2][plane 2][lane 64] x (8 bf16) (pack_wsplit_weight)
    float* y;
    const float* res1;
    int32_t cin;
    int32_t ntc, ytiles, ntiles, tiles_per_xcd;
    int32_t rows;        // batch * hp flattened image rows
};

__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ uint4 as_u4(const f32x16& v, int q) {
    return make_uint4(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3]));
}

// two fp32 -> two bf16 (round to nearest even) in one v_cvt_pk_bf16_f32, first value in the low half
__device__ __forceinline__ unsigned cvt2(float v0, float v1) {
#ifdef DINV_EMU
    return f2bf(v0) | (f2bf(v1) << 16);
#else
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {v0, v1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
#endif
}
// split8 of drunet_split_common.hpp with the conversions paired by hand (this file is compiled without SLP vectorisation)
__device__ __forceinline__ void split8p(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = cvt2(v[2 * e], v[2 * e + 1]);
        l[e] = cvt2(v[2 * e] - __uint_as_float(h[e] << 16), v[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// keeps the instruction scheduler from sinking a group of global loads down to their first use (it does, to save
// registers - and the loads then pay their whole latency inside the MFMA stream)
__device__ __forceinline__ void sched_fence() {
#ifndef DINV_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

template <bool RELU, int NRES, int TC>
__global__ __launch_bounds__(256, 2) void conv3x3_wsplit_kernel(WsArgs a) {
    using T = TileW<TC>;
    constexpr int PC = T::PC, TR = T::TR, VPL = T::VPL, VSTAGE = T::VSTAGE, NTASK = T::NTASK;
    DINV_DYN_LDS(uint4, lds);   // two V stages; reused by the epilogue exchange
    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;     // wave = Winograd point
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware order as in drunet_split2d.hip (speed only)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int tr_i = tile / a.ntc, tc_i = tile - tr_i * a.ntc;
    const int r0 = tr_i * TR, c0 = 1 + tc_i * TC;      // first interior row (flattened over images) / column of the tile
    const int nstep = a.cin / 16, nsub = 3 * nstep;

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- transform tasks: task q = (cblk, row, pair) of the halo region reads the four pixels d0..d3 = columns 2 pair - 1 ..
    // 2 pair + 2 (tile coordinates) of one row and produces the four points.  Thread tid owns task tid (all four points); the
    // NTASK - 256 tasks of the last halo rows are cut by POINT: wave k computes V_k of task 256 + lane from the two pixels
    // that point needs - every wave carries the same vector work (a whole extra task per thread of one wave made that
    // wave the slowest one of every step)
    int xoff0, voff0, xoffa, xoffb, voffe;
    {
        const int q = tid;
        const int cb = q / VPL, rem = q - cb * VPL, row = rem / PC, pr = rem - row * PC;
        xoff0 = (int)(((int64_t)cb * a.g.cs + a.g.sl + (int64_t)(r0 - 1 + row) * a.g.wp + (c0 - 1 + 2 * pr)) * 8);
        voff0 = cb * 4 * VPL + row * PC + pr;
    }
    const bool has1 = 256 + lane < NTASK;
    const float esgn = k == 1 ? 1.f : -1.f;              // V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
    {
        const int q = has1 ? 256 + lane : NTASK - 1;     // clamped: lanes without a task load a valid address and write nothing
        const int cb = q / VPL, rem = q - cb * VPL, row = rem / PC, pr = rem - row * PC;
        const int base = (int)(((int64_t)cb * a.g.cs + a.g.sl + (int64_t)(r0 - 1 + row) * a.g.wp + (c0 - 1 + 2 * pr)) * 8);
        xoffa = base + 8 * (k == 0 ? 0 : k == 2 ? 2 : 1);
        xoffb = base + 8 * (k == 2 ? 1 : k == 3 ? 3 : 2);
        voffe = (cb * 4 + k) * VPL + row * PC + pr;
    }
    const int64_t step_stride = (int64_t)2 * a.g.cs * 8;      // floats per 16-channel step

    // B operand slots (pair l31 of pair tile n, k half = channel block lhi) of this wave's point
    int bslot[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int q = n * 32 + l31;
        const int tr = q / PC, pr = q - tr * PC;
        bslot[n] = (lhi * 4 + k) * VPL + tr * PC + pr;       // + plane * 8 VPL + dy * PC
    }
    // A operand fragments of this wave's point, straight from the packed weights
    const uint4* const wsrc = a.w + (int64_t)ty * nsub * WSUB + k * 256 + lane;

    uint4 d0a, d0b, d1a, d1b, d2a, d2b, d3a, d3b;     // the four pixels of this thread's task (8 channels each)
    uint4 eaa, eab, eba, ebb;                         // the two pixels of this wave's point of an extra task
    auto ldd = [&](int s) {
        const float* p = a.x + (int64_t)s * step_stride + xoff0;
        d0a = ldu4(p);      d0b = ldu4(p + 4);
        d1a = ldu4(p + 8);  d1b = ldu4(p + 12);
        d2a = ldu4(p + 16); d2b = ldu4(p + 20);
        d3a = ldu4(p + 24); d3b = ldu4(p + 28);
    };
    auto lde = [&](int s) {
        const float* p = a.x + (int64_t)s * step_stride;
        eaa = ldu4(p + xoffa); eab = ldu4(p + xoffa + 4);
        eba = ldu4(p + xoffb); ebb = ldu4(p + xoffb + 4);
    };
    auto putv01 = [&](uint4* vb) {      // points 0 and 1 of the thread's task
        const float4 f0a = as_f4(d0a), f0b = as_f4(d0b), f1a = as_f4(d1a), f1b = as_f4(d1b), f2a = as_f4(d2a), f2b = as_f4(d2b);
        uint4 h0, l0, h1, l1;
        split8p(sub4(f0a, f2a), sub4(f0b, f2b), h0, l0);
        split8p(add4(f1a, f2a), add4(f1b, f2b), h1, l1);
        vb[voff0] = h0;           vb[voff0 + 8 * VPL] = l0;
        vb[voff0 + VPL] = h1;     vb[voff0 + 9 * VPL] = l1;
    };
    auto putv23 = [&](uint4* vb) {      // points 2 and 3
        const float4 f1a = as_f4(d1a), f1b = as_f4(d1b), f2a = as_f4(d2a), f2b = as_f4(d2b), f3a = as_f4(d3a), f3b = as_f4(d3b);
        uint4 h2, l2, h3, l3;
        split8p(sub4(f2a, f1a), sub4(f2b, f1b), h2, l2);
        split8p(sub4(f1a, f3a), sub4(f1b, f3b), h3, l3);
        vb[voff0 + 2 * VPL] = h2; vb[voff0 + 10 * VPL] = l2;
        vb[voff0 + 3 * VPL] = h3; vb[voff0 + 11 * VPL] = l3;
    };
    auto pute = [&](uint4* vb) {        // this wave's point of the extra task: A + sgn B (exact: sgn = +-1)
        const float4 fa0 = as_f4(eaa), fa1 = as_f4(eab), fb0 = as_f4(eba), fb1 = as_f4(ebb);
        uint4 h, l;
        split8p(make_float4(fmaf(esgn, fb0.x, fa0.x), fmaf(esgn, fb0.y, fa0.y), fmaf(esgn, fb0.z, fa0.z), fmaf(esgn, fb0.w, fa0.w)),
                make_float4(fmaf(esgn, fb1.x, fa1.x), fmaf(esgn, fb1.y, fa1.y), fmaf(esgn, fb1.z, fa1.z), fmaf(esgn, fb1.w, fa1.w)), h, l);
        if (has1) {
            vb[voffe] = h;
            vb[voffe + 8 * VPL] = l;
        }
    };
    auto lda = [&](int j, uint4& a00, uint4& a01, uint4& a10, uint4& a11) {     // [m][plane] of sub-step j = 3 step + dy
        const uint4* p = wsrc + (int64_t)j * WSUB;
        a00 = p[0]; a01 = p[64]; a10 = p[128]; a11 = p[192];
    };

    // one kernel row of a step: 24 MFMAs; the B fragments of pair tile n + 1 are read while the six MFMAs of tile n run;
    // smallest terms first (Uh*Vl, Ul*Vh, Uh*Vh).  (Rotating over four accumulators instead of two measured the same.)
    auto mma = [&](const uint4* vs, const uint4& a00, const uint4& a01, const uint4& a10, const uint4& a11) {
        uint4 B[2][2];
        B[0][0] = vs[bslot[0]]; B[0][1] = vs[8 * VPL + bslot[0]];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int cur = n & 1;
            if (n < 3) { B[cur ^ 1][0] = vs[bslot[n + 1]]; B[cur ^ 1][1] = vs[8 * VPL + bslot[n + 1]]; }
            acc[0][n] = mfma_bf16(a00, B[cur][1], acc[0][n]);
            acc[1][n] = mfma_bf16(a10, B[cur][1], acc[1][n]);
            acc[0][n] = mfma_bf16(a01, B[cur][0], acc[0][n]);
            acc[1][n] = mfma_bf16(a11, B[cur][0], acc[1][n]);
            acc[0][n] = mfma_bf16(a00, B[cur][0], acc[0][n]);
            acc[1][n] = mfma_bf16(a10, B[cur][0], acc[1][n]);
        }
    };

    // ---- prologue: V of step 0, the weights of its first two kernel rows, the extra-task pixels of step 1
    // A fragments [m][plane] of kernel rows 0 / 1 / 2: row dy of a step uses set dy and, as its first act, requests the row two
    // sub-steps ahead into the set the previous row has just finished with (the 512-channel layers' weights do not stay in
    // the 4 MB L2 next to the streaming activations: one row of lead, ~0.9 us, did not cover those misses)
    uint4 a0_00, a0_01, a0_10, a0_11, a1_00, a1_01, a1_10, a1_11, a2_00, a2_01, a2_10, a2_11;
    lda(0, a0_00, a0_01, a0_10, a0_11);
    lda(1, a1_00, a1_01, a1_10, a1_11);
    ldd(0);
    lde(0);
    putv01(lds);
    putv23(lds);
    pute(lds);
    lde(nstep > 1 ? 1 : 0);
    __syncthreads();

    // one 16-channel step.  While its 72 MFMAs run, the V stage of the next step is built: the transform is spread over the
    // three kernel rows (32 / 64 / 64 vector instructions beside 24 MFMAs each; bunched into one row it cost 10 % of the
    // kernel).  The LAST step of a tile has nothing to stage and requests no weights beyond its own rows (a global load costs
    // the issuing wave far more than its issue slot, so at 64 input channels - four steps - re-staging a dummy step and
    // clamped dummy weight loads were a fifth of all loads)
    auto step = [&](int s, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const uint4* const vcur = lds + (s & 1) * VSTAGE;
        uint4* const vnext = lds + ((s + 1) & 1) * VSTAGE;
        const int j = 3 * s;
        // kernel row 0; the thread's four pixels of the next step are requested; extra-task point (its pixels came in a row ago)
        lda(j + 2, a2_00, a2_01, a2_10, a2_11);
        if constexpr (!LAST) ldd(s + 1);
        sched_fence();
        mma(vcur, a0_00, a0_01, a0_10, a0_11);
        if constexpr (!LAST) pute(vnext);
        sched_fence();
        // kernel row 1; points 0, 1
        if constexpr (!LAST) {
            lda(j + 3, a0_00, a0_01, a0_10, a0_11);
            sched_fence();
        }
        mma(vcur + PC, a1_00, a1_01, a1_10, a1_11);
        if constexpr (!LAST) putv01(vnext);
        sched_fence();
        // kernel row 2; points 2, 3; the extra-task pixels of the step after the next are requested
        if constexpr (!LAST) {
            lda(j + 4, a1_00, a1_01, a1_10, a1_11);
            lde(s + 2 < nstep ? s + 2 : s + 1);      // (second-to-last step: unused, clamped instead of a branch - a
            sched_fence();                           // branch here costs registers, and the kernel has none to spare)
        }
        mma(vcur + 2 * PC, a2_00, a2_01, a2_10, a2_11);
        if constexpr (!LAST) putv23(vnext);
        sched_fence();
        lds_barrier();     // the next V stage is complete, every read of this one is done (last step: the stages become the
                           // exchange buffer of the epilogue)
    };
    for (int s = 0; s + 1 < nstep; ++s) step(s, std::false_type{});
    step(nstep - 1, std::true_type{});

    // ---- epilogue: y(2j) = M0 + M1 + M2, y(2j+1) = M1 - M2 - M3.  Wave w finishes pair tile w; register quads 2q, 2q+1 of a
    // 32-cout tile m are channels 0..7 of block cb0 + 4m + 2q + lhi (row permutation of pack_wsplit_weight)
    const int cb0 = ty * 8;
    const int qp = k * 32 + l31;
    const int ptr_ = qp / PC, ppr = qp - ptr_ * PC;
    const int RR = r0 + ptr_, cc = c0 + 2 * ppr;
    const bool store = RR < a.rows && cc < a.g.w + 1;      // even width: both pixels of a pair are inside or outside together
    const int img = RR / a.g.hp, rr = RR - img * a.g.hp;
    const bool in = rr >= 1 && rr <= a.g.h;                // frame rows between images stay zero
    const int64_t opix = a.g.sl + (int64_t)RR * a.g.wp + cc;
    // (the barrier that ended the last step: every V read is done, the stages become the exchange buffer)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float4 rs[NRES >= 1 ? 8 : 1];
        if (NRES >= 1 && store) {      // residual loads first: in flight during the exchange
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* rp = a.res1 + ((int64_t)(cb0 + 4 * m + 2 * q + lhi) * a.g.cs + opix) * 8;
                rs[4 * q] = ld4(rp); rs[4 * q + 1] = ld4(rp + 4); rs[4 * q + 2] = ld4(rp + 8); rs[4 * q + 3] = ld4(rp + 12);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) lds[((k * 4 + n) * 4 + q) * 64 + lane] = as_u4(acc[m][n], q);
        lds_barrier();
        float y0[16], y1[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 m0 = as_f4(lds[((0 * 4 + k) * 4 + q) * 64 + lane]), m1 = as_f4(lds[((1 * 4 + k) * 4 + q) * 64 + lane]);
            const float4 m2 = as_f4(lds[((2 * 4 + k) * 4 + q) * 64 + lane]), m3 = as_f4(lds[((3 * 4 + k) * 4 + q) * 64 + lane]);
            y0[4 * q] = (m0.x + m1.x) + m2.x; y0[4 * q + 1] = (m0.y + m1.y) + m2.y;
            y0[4 * q + 2] = (m0.z + m1.z) + m2.z; y0[4 * q + 3] = (m0.w + m1.w) + m2.w;
            y1[4 * q] = (m1.x - m2.x) - m3.x; y1[4 * q + 1] = (m1.y - m2.y) - m3.y;
            y1[4 * q + 2] = (m1.z - m2.z) - m3.z; y1[4 * q + 3] = (m1.w - m2.w) - m3.w;
        }
        if (m == 0) lds_barrier();     // the exchange buffer is free for the second cout tile
        if (!store) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t o = ((int64_t)(cb0 + 4 * m + 2 * q + lhi) * a.g.cs + opix) * 8;
            float v[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = y0[8 * q + e]; v[8 + e] = y1[8 * q + e]; }
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (NRES == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 r = rs[4 * q + e];
                    v[4 * e] += r.x; v[4 * e + 1] += r.y; v[4 * e + 2] += r.z; v[4 * e + 3] += r.w;
                }
            }
            if (!in) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) st4(a.y + o + 4 * e, make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]));
        }
    }
}

template <bool RELU, int NRES, int TC>
int launch_ws(WsArgs a, hipStream_t st) {
    using T = TileW<TC>;
    constexpr size_t lds = (size_t)T::LDS_UNITS * sizeof(uint4);
    static_assert(2 * lds <= 160 * 1024, "two workgroups must fit one CU");
    const int ntr = (int)ceil_div(a.rows, T::TR);
    a.ntc = (int)ceil_div(a.g.w, TC);
    a.ntiles = ntr * a.ntc;
    a.tiles_per_xcd = (int32_t)ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.tiles_per_xcd * a.ytiles * 8)), block(256);
    auto kern = conv3x3_wsplit_kernel<RELU, NRES, TC>;
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e_ != hipSuccess) return fail(100 + (int)e_, "hipFuncSetAttribute: %s", hipGetErrorString(e_));
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

template <bool RELU, int NRES>
int dispatch_ws(const WsArgs& a, int tc, hipStream_t st) {
    if (tc == 32) return launch_ws<RELU, NRES, 32>(a, st);
    if (tc == 16) return launch_ws<RELU, NRES, 16>(a, st);
    return launch_ws<RELU, NRES, 8>(a, st);
}

}  // namespace

// y = [relu](conv3x3(x)) (+ res1): Winograd F(2,3) along rows on the bf16 matrix cores, operands split in two bf16 parts.
// w_wsplit from pack_wsplit_weight (deepinv_amd/hip/drunet.py): [cout/64][cin/16][dy 3][point 4][m 2][plane 2][lane 64][8] bf16.
// Needs an even image width, cin % 16 == 0, cout % 64 == 0.  flags: bit 2 = relu (the bit of dinv_conv3x3_split).
extern "C" int dinv_conv3x3_wsplit(const dinv_act_geom* g, const void* x, const void* w_wsplit, int32_t cin, int32_t cout,
                                   void* y, const float* res1, int32_t flags, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_wsplit && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "Winograd bf16-split conv needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(g->width % 2 == 0, "Winograd F(2,3) along rows needs an even image width (got %d)", g->width);
    DINV_REQUIRE((flags & ~4) == 0, "unknown flags %d", flags);
    const bool relu = flags & 4;
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    // halo rows of the last row tile and the column overhang of a partial column tile must stay inside a channel block
    DINV_REQUIRE(g->cs >= g->sl + g->np + (int64_t)34 * g->wp + 64, "channel-block stride too small for 2-D tiles (rebuild the geometry)");
    DINV_REQUIRE(g->cs * 16 < ((int64_t)1 << 31), "activation row too long for 32-bit staging offsets");
    WsArgs a{make_geom(*g), reinterpret_cast<const float*>(x), reinterpret_cast<const uint4*>(w_wsplit),
             reinterpret_cast<float*>(y), res1, cin, 0, cout / 64, 0, 0, g->batch * g->hp};
    const int tc = g->width % 32 == 0 ? 32 : (g->width % 16 == 0 ? 16 : 8);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (relu) return dispatch_ws<true, 0>(a, tc, st);
    if (res1) return dispatch_ws<false, 1>(a, tc, st);
    return dispatch_ws<false, 0>(a, tc, st);
}
