// DRUNet ResBlock 3x3 convolution: Winograd F(2x2, 3x3) on the BF16 matrix cores with the exact two-part operand split
// (gfx950).  OPT-IN (DINV_DRUNET_CONV=wbf16): written at the end of round 2 and validated on the host emulation; three
// short hardware runs (scripts/bench_wbf16.py, profiles/r02_wbf16_first.jsonl; B = 32, ms per conv at the four DRUNet
// levels, direct bf16-split kernel in brackets): correct at first launch (6e-6 relative to the direct kernel);
//   first form                                              0.99 / 0.77 / 0.67 / 0.79   (0.94 / 0.73 / 0.73 / 0.76)
//   + packed transform, waves w / w+4 staggered             0.90 / 0.69 / 0.60 / 0.70   (0.92 / 0.73 / 0.71 / 0.74)
//   + epilogue in two chunks of 32 couts                    0.90 / 0.70 / 0.60 / 0.68   (no change)
// A fit over the levels gives ~2.4 us per 16-channel block and ~8.5 us fixed per workgroup against ~0.8 us of MFMA issue
// per block: with 2.25x fewer MFMAs the kernel is bound by everything else (two barriers per block with all 8 waves, one
// workgroup per CU because of its 157 KB of LDS, so every fill / write-back is exposed).  Not profiled with counters
// (the GPU budget of the round was spent); the default stays the direct bf16-split kernel (drunet_bf16s.hip).
//
// Operator: y = [relu](conv3x3(x)) (+ res1), stride 1, zero padding 1, no bias (deepinv/models/drunet.py:403-434), on the
// padded channel-blocked activation layout of drunet.hip.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36: 2.25x fewer MFMAs than the direct kernel, which runs
// at the sustained (power-limited) rate of the bf16 pipe for its three products.  U = G g G^T is computed on the host in
// fp64, rounded to fp32 and split (uh = bf16(u), ul = bf16(u - uh)); V = B^T d B is formed in fp32 (additions only) and
// split on the fly; M = uh*vl + ul*vh + uh*vh accumulates in fp32 (v_mfma_f32_32x32x16_bf16).
//
// Workgroup = 8 waves = 64 tile positions (8 x 8 tiles = 16 x 16 output pixels of one image) x 64 couts x 16 Winograd
// points; wave w owns points 2w, 2w+1 (2 x 2 x 2 accumulators).  Per block of 16 input channels:
//   * the 18 x 18 x 16 raw input region goes global -> registers -> LDS (coalesced; the 4x4 patches overlap 4x);
//   * every thread transforms one (position, channel pair): 16 ds_read_b64, 32 packed adds, 16 splits, and writes its 4
//     bytes of the 16 bf16 operand units V[xi][plane][cblk][pos] (four adjacent lanes complete a 16-byte unit);
//   * the MFMAs of block s read V[s & 1] while block s + 1 is transformed into V[(s + 1) & 1]; the pre-split U of a wave's
//     two points comes straight from global / L2 into registers (8 x 16 bytes per lane, re-loaded under the transform).
// Epilogue: the accumulators are exchanged through LDS in two chunks of 32 couts ([xi][pos][32 co], padded), a thread
// then owns (position, 4 couts), applies A^T . A, ReLU / residual, and stores float4 to the interior pixels only.
// Budget per block and CU at full MFMA issue (1536 cycles): LDS 212 KB (raw 20 + 64, V 64 + 64) = 138 B/clk - above the
// 128 B/clk of the LDS, but the bf16 pipe sustains only ~55 % of its issue rate on this chip (see drunet_bf16s.hip), at
// which LDS sits near 60 %, the vector cache (U) near 25 % and the vector ALU near 30 %.
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

constexpr int NTHR = 512;
constexpr int TPS = 8;                         // tile positions per side of a workgroup tile
constexpr int RS = 2 * TPS + 2;                // raw region side (18 pixels)
constexpr int RPITCH = 20;                     // floats per raw pixel in LDS: 16 channels + 4 pad (16-byte aligned rows)
constexpr int RAWF = RS * RS * RPITCH;         // 6480 floats
constexpr int VUNITS = 16 * 2 * 2 * 64;        // 16-byte units of one V stage: [xi][plane][cblk][pos]
constexpr int MPITCH = 36;                     // floats per (xi, pos) row of the epilogue exchange: 32 couts + 4 pad
constexpr size_t LDS_BYTES = (size_t)RAWF * 4 + (size_t)2 * VUNITS * 16;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((size_t)16 * 64 * MPITCH * 4 <= LDS_BYTES, "epilogue exchange must fit the raw tile + the V stages");

struct WArgs {
    Geom g;
    const float* x;
    const uint4* u;    // [cout/64][cin/16][xi 16][plane 2][cblk 2][co 64] x (8 bf16)
    float* y;
    const float* res1;
    int32_t cin, cblocks_valid;
    int32_t tby, tbx;              // workgroup tiles per image along rows / columns
    int32_t ntiles, ytiles, tiles_per_xcd;
};

__device__ __forceinline__ unsigned f2bf(float f) {   // round to nearest even, as v_cvt_pk_bf16_f32
#ifdef DINV_EMU
    unsigned v = __float_as_uint(f);
    if ((v & 0x7fffffffu) > 0x7f800000u) return (v >> 16) | 0x40u;
    v += 0x7fffu + ((v >> 16) & 1u);
    return v >> 16;
#else
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
#endif
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

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

typedef float f2 __attribute__((ext_vector_type(2)));

// two fp32 values -> packed bf16 high parts and packed bf16 low parts (value 0 in the low half-word)
__device__ __forceinline__ void split2(const f2& v, unsigned& hi, unsigned& lo) {
#ifdef DINV_EMU
    const unsigned h0 = f2bf(v.x), h1 = f2bf(v.y);
    hi = h0 | (h1 << 16);
    lo = f2bf(v.x - bf2f(h0)) | (f2bf(v.y - bf2f(h1)) << 16);
#else
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 h = __builtin_convertvector(v, bf2);                 // v_cvt_pk_bf16_f32
    hi = __builtin_bit_cast(unsigned, h);
    f2 hf;
    hf.x = __uint_as_float(hi << 16);
    hf.y = __uint_as_float(hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf2));
#endif
}

template <bool RELU, int NRES>
__global__ __launch_bounds__(NTHR) void conv3x3_wbf16_kernel(WArgs a) {
    DINV_DYN_LDS(float, lds);
    float* raw = lds;                                              // [18][18][RPITCH]
    uint4* vst = reinterpret_cast<uint4*>(lds + RAWF);             // [2][VUNITS]
    float* mex = lds;                                              // epilogue exchange, over the raw tile and the V stages
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware order as in drunet_bf16s.hip: the cout tiles of one pixel tile and neighbouring pixel tiles share an L2
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int ty = jx % a.ytiles, tl = jx / a.ytiles;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tl >= a.tiles_per_xcd || tile >= a.ntiles) return;
    const int per_img = a.tby * a.tbx;
    const int b = tile / per_img, tin = tile - b * per_img;
    const int tyb = tin / a.tbx, txb = tin - tyb * a.tbx;
    // top-left pixel of the raw region in padded coordinates: row 2 * (8 tyb), column 2 * (8 txb)
    const int64_t org = (int64_t)b * a.g.plane + (int64_t)(2 * TPS * tyb) * a.g.wp + 2 * TPS * txb;
    const int64_t pix_max = a.g.cs - a.g.sl - 1;                   // last addressable pixel of a channel block
    const int S = a.cin / 16;

    // ---- raw staging slots: unit q = tid + 512 k < 1296: pixel q / 4 of the region, 16-byte piece q % 4 of its 16 channels
    constexpr int NRAW = (RS * RS * 4 + NTHR - 1) / NTHR;          // 3
    int64_t rg[NRAW];                                              // global float offset (without the channel-block-pair term)
    int rl[NRAW];                                                  // LDS float offset, -1: no unit
#pragma unroll
    for (int k = 0; k < NRAW; ++k) {
        const int q = tid + NTHR * k;
        rl[k] = -1;
        rg[k] = 0;
        if (q < RS * RS * 4) {
            const int px = q >> 2, sub = q & 3;
            const int rr = px / RS, cc = px - rr * RS;
            int64_t pix = org + (int64_t)rr * a.g.wp + cc;
            pix = pix > pix_max ? pix_max : pix;                   // partial tiles at the bottom of the last image
            rg[k] = ((int64_t)(sub >> 1) * a.g.cs + a.g.sl + pix) * 8 + 4 * (sub & 1);
            rl[k] = px * RPITCH + 4 * sub;
        }
    }
    float4 rreg[NRAW];
    auto load_raw = [&](int s) {
        const float* xs = a.x + (int64_t)(2 * s) * a.g.cs * 8;
#pragma unroll
        for (int k = 0; k < NRAW; ++k) rreg[k] = rl[k] >= 0 ? ld4(xs + rg[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto write_raw = [&]() {
#pragma unroll
        for (int k = 0; k < NRAW; ++k)
            if (rl[k] >= 0) *reinterpret_cast<float4*>(raw + rl[k]) = rreg[k];
    };

    // ---- transform role: lane = (position within a group of 16, channel pair), wave = (position group, channel block)
    const int cp = lane & 3, tcb = wv & 1;
    const int tpos = (wv >> 1) * 16 + (lane >> 2);
    const int traw = ((2 * (tpos >> 3)) * RS + 2 * (tpos & 7)) * RPITCH + tcb * 8 + 2 * cp;    // patch element (0, 0)
    auto transform = [&](int stage) {
        f2 t[4][4];     // B^T d, one patch column at a time (keeps 8 instead of 32 raw values live); packed fp32 adds
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const f2*>(raw + traw + (i * RS + j) * RPITCH);
            t[0][j] = d[0] - d[2];
            t[1][j] = d[1] + d[2];
            t[2][j] = d[2] - d[1];
            t[3][j] = d[1] - d[3];
        }
        unsigned* vw = reinterpret_cast<unsigned*>(vst + (size_t)stage * VUNITS);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f2 v[4];    // (B^T d) B
            v[0] = t[r][0] - t[r][2];
            v[1] = t[r][1] + t[r][2];
            v[2] = t[r][2] - t[r][1];
            v[3] = t[r][1] - t[r][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xi = 4 * r + c;
                unsigned hi, lo;
                split2(v[c], hi, lo);
                // unit (xi, plane, cblk, pos), dword cp of its 8 channels
                vw[((((xi * 2 + 0) * 2 + tcb) * 64 + tpos) << 2) + cp] = hi;
                vw[((((xi * 2 + 1) * 2 + tcb) * 64 + tpos) << 2) + cp] = lo;
            }
        }
    };

    // ---- MFMA role: wave wv owns the Winograd points 2 wv, 2 wv + 1
    const uint4* ubase = a.u + (int64_t)ty * S * (16 * 2 * 2 * 64);
    uint4 U[2][2][2];   // [point][plane][m]; re-loaded right after the MFMAs of a block, under the next transform
    auto load_u = [&](int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    U[p][pl][m] = ubase[(((int64_t)s * 16 + 2 * wv + p) * 2 + pl) * 128 + lhi * 64 + m * 32 + l31];
    };
    f32x16 acc[2][2][2];   // [point][m][n]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][m][n][r] = 0.f;
    auto mma = [&](int stage) {
        const uint4* vs = vst + (size_t)stage * VUNITS;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int xi = 2 * wv + p;
            uint4 B[2][2];   // [plane][n]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int n = 0; n < 2; ++n) B[pl][n] = vs[((xi * 2 + pl) * 2 + lhi) * 64 + n * 32 + l31];
            // smallest terms first: uh*vl, ul*vh, uh*vh; product-major so that consecutive MFMAs hit different accumulators
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int pa = e == 1 ? 1 : 0, pb = e == 0 ? 1 : 0;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[p][m][n] = mfma_bf16(U[p][pa][m], B[pb][n], acc[p][m][n]);
            }
        }
    };

    // ---- pipeline over the blocks of 16 input channels: the MFMAs of block s read V[s & 1] while block s + 1 is
    // transformed into V[(s + 1) & 1]; the raw tile of block s + 2 and the U of block s + 1 are in flight meanwhile
    load_raw(0);
    load_u(0);
    write_raw();
    __syncthreads();
    transform(0);
    if (S > 1) load_raw(1);
    __syncthreads();                 // V[0] complete; the raw tile may be overwritten
    // waves w and w + 4 share a SIMD (drunet_bf16s.hip): one of them transforms before its MFMAs, the other after, so that
    // on every SIMD vector work (transform, split) and matrix work overlap instead of alternating in lockstep
    const bool early = (wv & 4) != 0;
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) write_raw();  // raw tile of block s + 1
        lds_barrier();
        if (s + 2 < S) load_raw(s + 2);
        if (early && s + 1 < S) transform((s + 1) & 1);
        mma(s & 1);
        if (s + 1 < S) {
            load_u(s + 1);
            if (!early) transform((s + 1) & 1);
        }
        lds_barrier();               // V[(s+1)&1] complete, V[s&1] and the raw tile consumed
    }

    // ---- epilogue: two chunks of 32 couts through LDS ([xi][pos][32 co], padded), then (position, 4 couts) per thread
    const int epos = tid & 63, eq = tid >> 6;                              // eq: cout quad 0..7 of the chunk
    const int tyt = TPS * tyb + (epos >> 3), txt = TPS * txb + (epos & 7);     // tile coordinates in the image
    const int R0 = 1 + 2 * tyt, C0 = 1 + 2 * txt;                          // padded coordinates of output (0, 0)
    const int64_t p00 = (int64_t)b * a.g.plane + (int64_t)R0 * a.g.wp + C0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {                                          // chunk m = accumulator row tile m (32 couts)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {                           // couts 8 rq + 4 lhi .. + 3 of the chunk
                    const float4 v = make_float4(acc[p][m][n][4 * rq], acc[p][m][n][4 * rq + 1], acc[p][m][n][4 * rq + 2],
                                                 acc[p][m][n][4 * rq + 3]);
                    *reinterpret_cast<float4*>(mex + (((2 * wv + p) * 64 + n * 32 + l31) * MPITCH + 8 * rq + 4 * lhi)) = v;
                }
        __syncthreads();
        {
            float4 M[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) M[r][c] = *reinterpret_cast<const float4*>(mex + (((4 * r + c) * 64 + epos) * MPITCH + 4 * eq));
            const int cbo = ty * 8 + m * 4 + (eq >> 1);                    // output channel block
            if (cbo < a.cblocks_valid) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float4 tr[4];    // row i of A^T M
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 &m0 = M[0][c], &m1 = M[1][c], &m2 = M[2][c], &m3 = M[3][c];
                        tr[c] = i == 0 ? make_float4(m0.x + m1.x + m2.x, m0.y + m1.y + m2.y, m0.z + m1.z + m2.z, m0.w + m1.w + m2.w)
                                       : make_float4(m1.x - m2.x - m3.x, m1.y - m2.y - m3.y, m1.z - m2.z - m3.z, m1.w - m2.w - m3.w);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float4 o = j == 0 ? make_float4(tr[0].x + tr[1].x + tr[2].x, tr[0].y + tr[1].y + tr[2].y,
                                                        tr[0].z + tr[1].z + tr[2].z, tr[0].w + tr[1].w + tr[2].w)
                                          : make_float4(tr[1].x - tr[2].x - tr[3].x, tr[1].y - tr[2].y - tr[3].y,
                                                        tr[1].z - tr[2].z - tr[3].z, tr[1].w - tr[2].w - tr[3].w);
                        if (R0 + i > a.g.h || C0 + j > a.g.w) continue;    // outside the image: interior pixels only
                        const int64_t off = ((int64_t)cbo * a.g.cs + a.g.sl + p00 + (int64_t)i * a.g.wp + j) * 8 + 4 * (eq & 1);
                        if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                        if (NRES >= 1) o = add4(o, ld4(a.res1 + off));
                        st4(a.y + off, o);
                    }
                }
            }
        }
        if (m == 0) __syncthreads();
    }
}

}  // namespace

extern "C" int dinv_conv3x3_wbf16(const dinv_act_geom* g, const float* x, const void* u_split, int32_t cin, int32_t cout,
                                  float* y, const float* res1, int32_t relu, dinv_stream_t stream) {
    if (int e = check_geom(g)) return e;
    DINV_REQUIRE(x && u_split && y, "null tensor pointer");
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "Winograd bf16-split conv needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(x != y, "in-place convolution is not supported");
    WArgs a{make_geom(*g), x, reinterpret_cast<const uint4*>(u_split), y, res1, cin, cout / 8, 0, 0, 0, 0, 0};
    a.tby = (int32_t)ceil_div((g->height + 1) / 2, TPS);
    a.tbx = (int32_t)ceil_div((g->width + 1) / 2, TPS);
    a.ntiles = g->batch * a.tby * a.tbx;
    a.ytiles = cout / 64;
    a.tiles_per_xcd = (int32_t)ceil_div(a.ntiles, 8);
    const dim3 grid((unsigned)(a.tiles_per_xcd * a.ytiles * 8)), block(NTHR);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define DINV_W_LAUNCH(R, N)                                                                                       \
    do {                                                                                                          \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wbf16_kernel<R, N>),            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);          \
        if (e_ != hipSuccess) return fail(100 + (int)e_, "hipFuncSetAttribute: %s", hipGetErrorString(e_));       \
        hipLaunchKernelGGL((conv3x3_wbf16_kernel<R, N>), grid, block, LDS_BYTES, st, a);                          \
    } while (0)
    if (relu) DINV_W_LAUNCH(true, 0);
    else if (res1) DINV_W_LAUNCH(false, 1);
    else DINV_W_LAUNCH(false, 0);
#undef DINV_W_LAUNCH
    DINV_CHECK_LAUNCH();
    return 0;
}
