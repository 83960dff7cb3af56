// DRUNet's last layer (m_tail, deepinv/models/drunet.py:39-101: nc[0] -> image channels, input = x + skip) on the vector ALU, gfx950.
// With 1-4 output channels the 32-row MFMA tile of drunet.hip computes 8-16x more than needed (measured 1.12 ms at 32 x 320 x 320);
// here a lane produces pixels x COUT channels with wave-uniform weights (scalar loads).
//
// Built without SLP vectorisation like the whole library (csrc/Makefile).  This kernel is where the reason was found: with hipcc's
// SLP-packed fp32 ops, tail3x3_shift_kernel was not reproducible while a bf16-split convolution (drunet_wsplit.hip, bf16 MFMA) ran on
// another stream of the same device - the batch lanes of models/drunet.py do exactly that.  In lanes 48..63 of a wave single terms of
// the sum were dropped: the LOW result of v_pk_mul_f32 / v_pk_fma_f32 forms that read the HIGH half of src1 (op_sel[1] = 1) is wrong
// there while another kernel's wave runs bf16 MFMAs on the SIMD (scripts/r06/probe/pk_forms_probe.hip).  The
// scalar-FMA build is bit-reproducible in every setting and runs at the same 0.39 ms (the kernel streams its two input tensors);
// tests/test_drunet_gpu.py and tests/test_loops_gpu.py hold the regression tests.
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

// One lane produces one pixel x COUT channels; the 9 taps of neighbouring lanes hit L1 (launches beyond 65535 slices only).
template <int COUT>
__global__ __launch_bounds__(256) void tail3x3_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ x2,
                                                      const float* __restrict__ w, float* __restrict__ y, int ncb) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one pixel per lane: neighbours share cache lines
    if (p >= g.np) return;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int cb = 0; cb < ncb; ++cb) {
        const int64_t base = ((int64_t)cb * g.cs + g.sl + p) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int64_t o = base + ((int64_t)(t / 3 - 1) * g.wp + (t % 3 - 1)) * 8;
            float4 a = ld4(x + o), b = ld4(x + o + 4);
            if (x2) { a = add4(a, ld4(x2 + o)); b = add4(b, ld4(x2 + o + 4)); }
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float* wv = w + (((int64_t)cb * 9 + t) * COUT + co) * 8;   // uniform: scalar loads
                acc[co] += a.x * wv[0] + a.y * wv[1] + a.z * wv[2] + a.w * wv[3] + b.x * wv[4] + b.y * wv[5] +
                           b.z * wv[6] + b.w * wv[7];
            }
        }
    }
    if (!interior(g, p)) return;   // the zero frame is never written
    float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int co = 0; co < COUT; ++co) o4[co] = acc[co];
    st4(y + (g.sl + p) * 8, make_float4(o4[0], o4[1], o4[2], o4[3]));
}

// Lane-shift variant (the one the launcher uses): one lane owns one padded column and RB = 8 output rows; every input pixel
// (32 bytes per channel block and tensor) is loaded ONCE by its own lane and reaches the two neighbouring columns through a
// wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1), so the vector cache sees 10 rows x 1 column of loads per 8 output pixels
// instead of 6 x 3 per 4: the kernel then runs at the rate the two input tensors stream from HBM.  A wave covers `sw` <= 62
// output columns (lanes 1 .. sw; lanes 0 and sw + 1 only feed their neighbours).
__device__ __forceinline__ float lane_prev(float v) {     // value of lane - 1
#if defined(DINV_EMU)
    return __shfl_up(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float lane_next(float v) {     // value of lane + 1
#if defined(DINV_EMU)
    return __shfl_down(v, 1);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
#endif
}

// RB = output rows per wave: 8 where the launch has waves to spare (10 input rows per 8 output rows), 4 or 2 for small launches
// (a 4-slice batch at RB = 8 is 960 waves for 1024 SIMDs - each walking 80 dependent row loads: 100 us where the two tensors
// stream in 20)
template <int COUT, int RB>
__global__ __launch_bounds__(256) void tail3x3_shift_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ x2,
                                                            const float* __restrict__ w, float* __restrict__ y, int ncb, int sw) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * sw + lane;                                 // padded column of this lane
    const int rg = blockIdx.y * 4 + (threadIdx.x >> 6);                   // group of RB image rows (wave-uniform)
    const int b = blockIdx.z;
    const int r0 = 1 + rg * RB;                                           // first padded row of the group
    if (r0 > g.h) return;                                                 // whole wave
    const bool feeds = lane <= sw + 1 && c <= g.w + 1;                    // columns 0 and w + 1 are the zero frame
    float acc[RB][COUT];
#pragma unroll
    for (int k = 0; k < RB; ++k)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[k][co] = 0.f;
    const int64_t p0 = (int64_t)b * g.plane + (int64_t)r0 * g.wp + (feeds ? c : 0);
    for (int cb = 0; cb < ncb; ++cb) {
        const int64_t base = ((int64_t)cb * g.cs + g.sl + p0) * 8;
#pragma unroll
        for (int rr = -1; rr <= RB; ++rr) {                               // input rows r0 - 1 .. r0 + RB
            if (r0 + rr > g.h + 1) continue;                              // below the zero frame: nothing to read (wave-uniform)
            const int64_t o = base + (int64_t)rr * g.wp * 8;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bq = a;
            if (feeds) {
                a = ld4(x + o); bq = ld4(x + o + 4);
                if (x2) { a = add4(a, ld4(x2 + o)); bq = add4(bq, ld4(x2 + o + 4)); }
            }
            const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                float u[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] = dc == 0 ? v[i] : dc < 0 ? lane_prev(v[i]) : lane_next(v[i]);
#pragma unroll
                for (int k = 0; k < RB; ++k) {                            // output row r0 + k uses it as tap dy = rr - k + 1
                    const int dy = rr - k + 1;
                    if (dy < 0 || dy > 2) continue;
                    const int t = dy * 3 + dc + 1;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float* wv = w + (((int64_t)cb * 9 + t) * COUT + co) * 8;   // uniform: scalar loads
                        acc[k][co] += u[0] * wv[0] + u[1] * wv[1] + u[2] * wv[2] + u[3] * wv[3] + u[4] * wv[4] + u[5] * wv[5] +
                                      u[6] * wv[6] + u[7] * wv[7];
                    }
                }
            }
        }
    }
    if (lane < 1 || lane > sw || c > g.w) return;
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        if (r0 + k > g.h) break;
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < COUT; ++co) o4[co] = acc[k][co];
        st4(y + (g.sl + p0 + (int64_t)k * g.wp) * 8, make_float4(o4[0], o4[1], o4[2], o4[3]));
    }
}

}  // namespace

extern "C" int dinv_conv3x3_tail(const dinv_act_geom* g, const float* x, const float* x2, const float* w_tail,
                                 int32_t cin, int32_t cout, float* y, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && w_tail && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 8 && cin % 8 == 0 && cout >= 1 && cout <= 4, "tail conv needs cin %% 8 == 0 and 1 <= cout <= 4 (got %d,%d)", cin, cout);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Geom gg = make_geom(*g);
    if (g->batch <= 65535) {
        const int nstrip = (int)ceil_div(g->width, 62), sw = (int)ceil_div(g->width, nstrip);     // balanced strips of <= 62 columns
        // rows per wave: the largest of 8, 4, 2 that still gives every SIMD of the chip about two waves
        int rb = 8;
        while (rb > 2 && (int64_t)nstrip * ceil_div(g->height, rb) * g->batch < 2048) rb >>= 1;
        const dim3 rgrid((unsigned)nstrip, (unsigned)ceil_div(ceil_div(g->height, rb), 4), (unsigned)g->batch);
#define DINV_TAIL(CO, RBV) hipLaunchKernelGGL((tail3x3_shift_kernel<CO, RBV>), rgrid, dim3(256), 0, st, gg, x, x2, w_tail, y, cin / 8, sw)
#define DINV_TAIL_RB(CO) do { if (rb == 8) DINV_TAIL(CO, 8); else if (rb == 4) DINV_TAIL(CO, 4); else DINV_TAIL(CO, 2); } while (0)
        switch (cout) {
            case 1: DINV_TAIL_RB(1); break;
            case 2: DINV_TAIL_RB(2); break;
            case 3: DINV_TAIL_RB(3); break;
            default: DINV_TAIL_RB(4); break;
        }
#undef DINV_TAIL_RB
#undef DINV_TAIL
        DINV_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)ceil_div(g->np, 256)), block(256);
    switch (cout) {
        case 1: hipLaunchKernelGGL(tail3x3_kernel<1>, grid, block, 0, st, gg, x, x2, w_tail, y, cin / 8); break;
        case 2: hipLaunchKernelGGL(tail3x3_kernel<2>, grid, block, 0, st, gg, x, x2, w_tail, y, cin / 8); break;
        case 3: hipLaunchKernelGGL(tail3x3_kernel<3>, grid, block, 0, st, gg, x, x2, w_tail, y, cin / 8); break;
        default: hipLaunchKernelGGL(tail3x3_kernel<4>, grid, block, 0, st, gg, x, x2, w_tail, y, cin / 8); break;
    }
    DINV_CHECK_LAUNCH();
    return 0;
}
