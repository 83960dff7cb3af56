// DRUNet 3x3 convolution, Winograd F(4x4, 3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Same operator as conv3x3_kernel (drunet.hip) / conv3x3_wino_kernel (drunet_wino.hip): nn.Conv2d 3x3 s1 p1 no bias inside
// the ResBlocks (reference deepinv/models/drunet.py:403-434), same "padded pixel rows, channels blocked by 8" activations,
// fp32 multiplies, fp32 accumulation - but 36 multiplies per 4x4 output tile and (ci, co) instead of 144: the MFMA work is
// 4x below the direct form and 1.78x below F(2x2, 3x3).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs   (Lavin & Gray)
//
// What shapes the kernel: the fp32 MFMA shares the vector ALU (every VALU instruction of a SIMD costs matrix time), and 36
// points x 64 couts x 32 tile positions of accumulators already fill the register file of a CU (8 waves x 144 registers).
//
// * Workgroup = 8 waves = 64 couts x 32 tile positions (512 output pixels) x 36 Winograd points; wave (c2, q) owns the
//   point slots 9q .. 9q+8 for cout half c2: 9 accumulators.  The points are stored (packed U, V stage) in an order in which
//   every wave's nine slots are a full row of the 6x6 transform followed by half a row, so that the same six accumulators
//   retire first in every wave's epilogue.  One workgroup per CU, two waves per SIMD, persistent (one per CU, walking a
//   contiguous range of the tile order of its XCD; the cout tiles of one position group are neighbours).
// * U = G g G^T (fp64 on the host, rounded once) is packed as the MFMA A fragments of exactly the wave that uses them
//   ([cout/64][cin/8][wave][point 9][lane][channel 4]): every U value is needed by ONE wave, so it never passes through
//   LDS - each lane loads 16 bytes (4 K steps of one point) straight from L2, three points ahead of their use.
// * V = B^T d B is computed ONCE per workgroup and 8-channel block (not per cout half): the raw 6x6-patch region of the
//   32 positions is staged global -> registers -> LDS ([channel half][pixel][4]); thread (position, channel, row half)
//   reads its patch with conflict-free 4-byte LDS reads, transforms it (72 vector instructions: the two row halves
//   {0,1,2} / {5,3,4} share one instruction stream through per-lane coefficients) and writes its 18 points to the V stage
//   [lane half][position][point][4 K steps] (43 16-byte slots per row: conflict-free 16-byte reads AND 4-byte writes, see VP),
//   which the MFMA waves read as B operands.  Double buffered: during block b the workgroup multiplies V[b], transforms raw[b+1] into V[b+1] and stages
//   raw[b+2]; ONE LDS-only barrier per block.  Every filler instruction is pinned between two MFMAs (sched_barrier).
// * Epilogue: each wave reduces its points to s = M A (a full Winograd row and half a row: 8 values), the four waves of
//   a cout half exchange s through LDS, all 512 threads finish A^T s (two output columns of four channels each), fused
//   ReLU / residual (requested after the first exchange round's writes, when six accumulators are dead), 16-byte stores that
//   cover 1 KB per wave instruction.  The next tile's first two raw blocks are requested as soon as all accumulators are dead
//   and land in registers behind the second exchange round.  The first block of a tile accumulates onto the constant 0.
#include "drunet_common.hpp"
#include "drunet_split_common.hpp"
#include <cstdlib>
#include <type_traits>
#include <utility>

using namespace dinv;
using namespace dinv_drunet;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 512;
// One (lane half, position) row of V = 36 points x 4 K steps in 43 16-byte slots: 3 / 1 / 3 empty slots behind the points 8 / 17 / 26
// and 4 empty slots between the two lane halves.  Reads (16 bytes per lane, 32 rows): any odd slot pitch is conflict-free.  Writes
// (4 bytes; a 32-lane group = 8 channels x 2 neighbouring positions x the 2 row halves): the slot of a lane is
// 43 position + d row_half + 4 lane_half (mod 8) with d = 34, 22, 10 for the three write rows - eight different slots, no conflicts
// (the dense 37-slot row made every V write 2- to 4-way conflicted)
constexpr int VP = 172;
constexpr int VHALF = 32 * VP + 16;
constexpr int VBUF = 2 * VHALF;          // floats per V stage (44.2 KB)
__host__ __device__ constexpr int vslot(int k) { return k + (k >= 9 ? 3 : 0) + (k >= 18 ? 1 : 0) + (k >= 27 ? 3 : 0); }   // 16-byte slot of point k
constexpr int EXCH = 4 * 8 * 8 * 32 * 4; // floats of the epilogue exchange: [q 4][value 8][cout quad 8][position 32][4]
constexpr int UBLK = 8 * 9 * 64 * 4;     // floats of U per (cout tile, channel block): [wave 8][point 9][lane 64][4]

struct W4Args {
    Geom g;
    const float* x;
    const float* w;
    float* y;
    const float* res;
    int32_t ncb, nct, nty, ntx;
    int64_t nsr, nwg, per_xcd;
    int32_t slots;                 // resident workgroups per XCD
    int32_t ct_major, npw;         // tile order: 1 = cout tile outermost (logical = ct * npw + pw), 0 = position group outermost
    // tail split: the per_xcd - full_x tiles of an XCD's last, incomplete round are cut into split_f parts along the input
    // channels (one part per workgroup); parts write partial OUTPUTS (the output transform is linear) to `part_buf`, the part
    // that arrives last (ticket in `tickets`) adds them in part order, applies ReLU / residual and stores
    int32_t full_x, split_f, ntail;
    float* part_buf;               // [XCD 8][tail tile][part][8192 float4]
    int32_t* tickets;              // [XCD 8][tail tile], zero between launches (the last arriver resets its counter)
    FastDiv d_img, d_ntx, d_nct, d_npw;   // / (nty*ntx), / ntx, / nct, / npw
#ifdef DINV_W4_TIMING
    long long* dbg;                // phase timestamps (s_memtime) of wave 0: 16 per tile, first 4 tiles of every workgroup
    int32_t stagger;               // experiment: workgroup j of an XCD starts (j % 4) * stagger / 4 cycles late (DINV_W4_STAGGER)
#endif
};

// the 32 positions of a workgroup are NSUB rectangles of TH x TW output tiles (4x4 pixels each)
template <int TH, int TW>
struct Shape4 {
    static constexpr int PT = TH * TW;
    static constexpr int NSUB = 32 / PT;
    static constexpr int RH = 4 * TH + 2;            // staged rows per rectangle
    static constexpr int RW = 4 * TW + 2;            // staged columns per rectangle
    static constexpr int NPIX = NSUB * RH * RW;
    static constexpr int NPIXP = NPIX + ((9 - NPIX % 8) % 8);   // = 1 mod 8: the two channel halves sit 4 banks apart
    static constexpr int RAWF = 2 * NPIXP * 4;       // floats per raw stage: [half 2][pixel][4]
    static constexpr int RAW4 = 2 * NPIX;            // 16-byte loads that stage one block
    static constexpr int NLD = (RAW4 + NTHR - 1) / NTHR;
    static constexpr int VOFF = 2 * RAWF;            // float offset of the V stages
    static constexpr int MAINF = 2 * VBUF + 2 * RAWF;
    static constexpr int LDSF = MAINF > EXCH ? MAINF : EXCH;
    static_assert(PT <= 32 && 32 % PT == 0, "rectangle must divide the 32 positions");
    static_assert(NPIXP % 8 == 1 && NLD <= 4, "raw stage layout");
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}

// two fp32 -> one register of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
#ifdef DINV_EMU
    return f2bf(a) | (f2bf(b) << 16);
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#endif
}
// x = xh + xm + xl, three bf16 parts (round to nearest even each; the two differences are exact), as packed pairs of ADJACENT
// channels: h.x = (h0 | h1 << 16), h.y = (h2 | h3 << 16).  11 vector instructions per two values.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hp, unsigned& mp, unsigned& lp) {
    hp = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
    mp = cvt_pk_bf16(ra, rb);
    lp = cvt_pk_bf16(ra - __uint_as_float(mp << 16), rb - __uint_as_float(mp & 0xffff0000u));
}
__device__ __forceinline__ void split3(const float4& x, uint2& h, uint2& m, uint2& l) {
    split3_pair(x.x, x.y, h.x, m.x, l.x);
    split3_pair(x.z, x.w, h.y, m.y, l.y);
}
// the 8 K slots of a lane in one 16-deep bf16 step: [X of its channels 0..3 | Y of its channels 0..3] - the packed pairs as they
// are, no byte shuffles (A and B use the same slot order, any order is as good as another)
__device__ __forceinline__ uint4 pair(const uint2& X, const uint2& Y) { return make_uint4(X.x, X.y, Y.x, Y.y); }

#ifdef DINV_EMU
#define DINV_PIN(x) ((void)0)
#define DINV_PIN2(x, y) ((void)0)
#else
#define DINV_PIN(x) asm volatile("" : "+v"(x))
#define DINV_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#endif
#ifdef DINV_EMU
#define DINV_W4_ATTR
#else
#define DINV_W4_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

// SPLIT = false: the whole tiles jt = j, j + slots, ... < full_x of the workgroup's XCD range.  SPLIT = true (a second launch): one
// part (split_f-th of the input channels) of one tail tile per workgroup; it publishes partial outputs and the last part combines
// BF3 = true (dinv_conv3x3_winograd4_bf16x3): the same kernel with every fp32 multiply evaluated on the BF16 matrix cores as a
// THREE-part operand split x = xh + xm + xl (round to nearest even each; exact to 2^-24) and SIX products
//   um vm + uh vh,   um vh + uh vl,   uh vm + ul vh
// (dropped: um vl, ul vm, ul vl - <= 2^-23 |u||v| when both operands sit at the half-ulp extremes of their bf16 parts, ~2^-26
// rms; tests/test_emu_drunet.py::test_winograd4_bf16x3_worst_case), fp32 accumulation: three v_mfma_f32_32x32x16_bf16 per point
// and 8-channel block instead of four v_mfma_f32_32x32x2_f32, 3/8 of their matrix-pipe time.  A 16-deep bf16 step holds 8
// channels x 2 parts (the 8 K slots of a lane half = part X of its 4 channels, then part Y), and with the parts kept in the
// order (m, h, l) the three operand pairs are overlapping four-register WINDOWS of six registers - nothing is duplicated:
//   A = (um, uh) x B = (vm, vh),   A = (um, uh) x B = (vh, vl),   A = (uh, ul) x B = (vm, vh).
// U arrives already split (hip/drunet.py: pack_winograd4_bf16x3_weight; 24 bytes per lane and point, loaded as 16 + 8 three points
// ahead); V is split in the registers of the wave that multiplies it, one point ahead of its use.  V stage, staging, transform
// and epilogue are the fp32 kernel's.  Measured (round 5, scripts/r05/wino4_bench.cpp, profiles/r05_wino4_*): per-layer error
// 1.1 / 1.5 / 2.0 / 2.9e-6 at the four DRUNet levels (fp32 form: 1.2 / 1.6 / 2.4 / 3.2e-6); main loop 4500 cycles per block
// against 5500 (matrix pipe alone: 1750 against 4600) - and the SAME wall time at 32 slices, because the package sits at its
// 1400 W cap under either form: the clock settles at 1.95 GHz under this one and at 2.32 GHz under the fp32 one
// (profiles/r05_wino4_power_cap_smi.log).  Bursts of 20-40 launches run 5-20 % faster in this form at every batch; SUSTAINED (the
// whole DRUNet for seconds, scripts/r05/bf16x3_e2e.py) it is 1-3 % slower at 4, 8, 16 and 32 slices: energy per convolution, not
// pipe time, is what the package limit prices.  Opt-in.
template <int TH, int TW, bool RELU, int NRES, bool SPLIT, bool BF3 = false>
__global__ __launch_bounds__(NTHR) DINV_W4_ATTR
void conv3x3_wino4_kernel(W4Args a) {
    using S = Shape4<TH, TW>;
    DINV_DYN_LDS(float, lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int l31 = lane & 31, h = lane >> 5;
    const int q = wave & 3;

    const int64_t bid = blockIdx.x;
    const int64_t per_img = (int64_t)a.nty * a.ntx;
    const int64_t xcs = a.g.cs * 8;

    // ---- MFMA role: this lane's B-operand row of the V stage and its A fragments inside one (cout tile, block) slab of U
    int vrd = S::VOFF + h * VHALF + l31 * VP + (9 * q + 3 * (q & 1) + 4 * (q >> 1)) * 4;   // (float index into lds: the V stages sit behind the raw stages; this wave reads points 9q..)
    // (BF3: a point of a wave = [lane 64][um 4 | uh 4] then [lane 64][ul 4] bf16: 1536 bytes)
    constexpr int UB = BF3 ? UBLK * 3 / 2 : UBLK;               // floats of U per (cout tile, channel block)
    constexpr int UPT = BF3 ? 1536 : 1024;                      // bytes per (wave, point)
    const uint32_t uoff = (uint32_t)(wave * 9 * UPT + lane * 16);
    const uint32_t uoff_l = (uint32_t)(wave * 9 * UPT + 1024 + lane * 8);

    // ---- transform role: thread = (position, channel of the block, row half)
    const int tch = tid & 7, trh = (tid >> 3) & 1, tpos = tid >> 4;
    const int tsub = tpos / S::PT, tty = (tpos % S::PT) / TW, ttx = tpos % TW;
    const int rbase = (tch >> 2) * (S::NPIXP * 4) + ((tsub * S::RH + 4 * tty) * S::RW + 4 * ttx) * 4 + (tch & 3);
    constexpr int RROW = S::RW * 4;                                  // floats between two rows of the raw stage
    // rows of the patch that enter xa, xb, xc:  row half 0 -> (0, 2, 4), row half 1 -> (1, 3, 5)
    const int rx = rbase + trh * RROW;
    // per-lane coefficients that let both row halves run ONE instruction stream (B^T rows {0,1,2} / {5,4,3}):
    //   tA = 4 xa - 5 xb + xc;  p = d4 - al d2;  qq = d3 - al d1;  tB = p + be qq;  tC = p - be qq
    const float al = trh ? 1.f : 4.f, be = trh ? -2.f : 1.f, nbe = -be;
    // Point slots of the V stage (and of the packed U): every wave's 9 consecutive slots are (a full Winograd row, half a row),
    //   0-5 row 0 | 6-8 row 1 cols 0-2 | 9-14 row 2 | 15-17 row 1 cols 3-5 | 18-23 row 3 | 24-26 row 4 cols 0-2 | 27-32 row 5 | 33-35 row 4 cols 3-5
    // so that the SAME six accumulators die first in every wave's epilogue.  V write rows (A, B, C) = (0, 1, 2) / (5, 4, 3):
    // B is the row that is stored in two pieces (its columns 3-5 sit 9 slots behind its columns 0-2)
    const int vwbase = (tch >> 2) * VHALF + tpos * VP + (tch & 3);
    int vwa = S::VOFF + vwbase + (trh ? vslot(27) : vslot(0)) * 4, vwb = S::VOFF + vwbase + (trh ? vslot(24) : vslot(6)) * 4,
        vwc = S::VOFF + vwbase + (trh ? vslot(18) : vslot(9)) * 4;
    static_assert(vslot(9) == 12 && vslot(18) == 22 && vslot(27) == 34 && vslot(15) - vslot(6) == vslot(33) - vslot(24), "V row layout");
    // LDS map: [raw stage 0][raw stage 1][V stage 0][V stage 1].  The V bases are beyond the 64 KB reach of a ds immediate
    // offset: they are folded into the per-lane offsets above, which are made opaque so that the compiler addresses every
    // access as (one base register + immediate) instead of hoisting one address register per distinct constant
    DINV_OPAQUE(vrd); DINV_OPAQUE(vwa); DINV_OPAQUE(vwb); DINV_OPAQUE(vwc);
    float* const V0 = lds;          // (+ S::VOFF through vrd / vwa / vwb / vwc)
    float* const R0 = lds;

    // ---- tile descriptors
    uint32_t goff[S::NLD];       // byte offsets of this thread's staging loads inside one channel block of x
    const float* wsrc;           // packed U of the tile's cout block
    int ct;
    uint32_t pw;
    auto describe = [&](int64_t logical) {
        // Tile order.  Position group outermost: the cout tiles of one position group are neighbours and share the input
        // through the XCD's L2 - right while all of U (cout tiles x 73.7 KB per channel block) stays L2-resident.  Beyond
        // that (256 / 512 channels: 9.4 / 37.7 MB of U against 4 MB of L2) every tile would stream its U slab from the
        // Infinity Cache again; with the cout tile outermost an XCD works on ONE cout tile at a time, its compute units walk
        // the channel blocks of that slab together, and the input (which fits the Infinity Cache) is what gets re-read
        if (a.ct_major) {
            ct = (int)a.d_npw.div((uint32_t)logical);
            pw = (uint32_t)logical - (uint32_t)ct * (uint32_t)a.npw;
        } else {
            pw = a.d_nct.div((uint32_t)logical);
            ct = (int)((uint32_t)logical - pw * (uint32_t)a.nct);
        }
        wsrc = a.w + (int64_t)ct * a.ncb * UB;
        int t = tid;
        DINV_OPAQUE(t);   // recompute the per-lane constants per tile instead of keeping them live
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) {
            const int e = t + NTHR * i;
            const int half = e & 1, px = e >> 1;      // lanes = consecutive 16-byte halves of consecutive pixels: 1 KB per wave load
            const int c = px % S::RW;
            const int r = (px / S::RW) % S::RH;
            const int sb = px / (S::RW * S::RH);
            const uint32_t s = pw * S::NSUB + sb;
            const uint32_t b = a.d_img.div(s);
            const uint32_t rem = s - b * (uint32_t)per_img;
            const uint32_t tyb = a.d_ntx.div(rem), txb = rem - tyb * a.ntx;
            const int gr = 4 * tyb * TH + r, gc = 4 * txb * TW + c;
            const bool ok = e < S::RAW4 && s < (uint32_t)a.nsr && gr < a.g.hp && gc < a.g.wp;
            // out-of-frame pixels read the (always zero) top-left border pixel of image 0 instead
            goff[i] = 4u * (ok ? (uint32_t)((a.g.sl + (int64_t)b * a.g.plane + (int64_t)gr * a.g.wp + gc) * 8 + half * 4)
                             : (uint32_t)(a.g.sl * 8));
        }
    };
    // LDS float offset of staging load i inside a raw stage
    auto loff = [&](int i) {
        const int e = tid + NTHR * i;
        return ((e & 1) * S::NPIXP + (e >> 1)) * 4;
    };
    auto stage_ok = [&](int i) { return S::RAW4 % NTHR == 0 || i + 1 < S::NLD || tid + NTHR * i < S::RAW4; };

    // buffer loads: uniform base in an SGPR resource + 32-bit per-lane byte offset (no address arithmetic on the vector ALU)
    auto ld4_so = [](const float* sbase, uint32_t byte_off, uint32_t soff) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, (int)soff, 0));
    };
    auto ld_u = [&](const float* wt, int cb, int k) { return ld4_so(wt, uoff, (uint32_t)(cb * (UB * 4) + k * UPT)); };
    auto ld_ul = [&](const float* wt, int cb, int k) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, uoff_l, (int)(cb * (UB * 4) + k * UPT), 0));
    };
    auto ld_x = [&](int cb, int i) { return ld4_so(a.x + cb * xcs, goff[i], 0u); };

    f32x16 acc[9];
    float4 u[3], v[2];
    uint2 ulo[3];                     // (BF3) u[] holds (um, uh) of a point as four registers of bf16 pairs, ulo[] its ul
    uint2 vm[2], vh[2], vl[2];        // (BF3) the three bf16 parts of the current and of the next point's V
    float t[3][6];
    float e7[7];
    float4 pr[S::NLD], pr2[S::NLD];

    // ---- the transform, in pieces that the main loop spreads over its MFMA slots
    // column j of the patch: 7 reads (xa, xb, xc from this row half's rows, d1..d4), then row half's three rows of B^T d
    auto tr_read = [&](const float* raw, int j) {
        e7[0] = raw[rx + j * 4];
        e7[1] = raw[rx + 2 * RROW + j * 4];
        e7[2] = raw[rx + 4 * RROW + j * 4];
        e7[3] = raw[rbase + 1 * RROW + j * 4];
        e7[4] = raw[rbase + 2 * RROW + j * 4];
        e7[5] = raw[rbase + 3 * RROW + j * 4];
        e7[6] = raw[rbase + 4 * RROW + j * 4];
    };
    auto tr_rows = [&](int j) {
        t[0][j] = fmaf(-5.f, e7[1], fmaf(4.f, e7[0], e7[2]));
        const float p = fmaf(-al, e7[4], e7[6]);
        const float qq = fmaf(-al, e7[3], e7[5]);
        t[1][j] = fmaf(be, qq, p);
        t[2][j] = fmaf(nbe, qq, p);
    };
    // row i of (B^T d) B: outputs 0..2 (part 0) or 3..5 (part 1), written to the V stage
    auto tr_cols = [&](float* vst, int i, int part) {
        const int wr = i == 0 ? vwa : i == 1 ? vwb : vwc;
        const int hi = i == 1 ? (vslot(15) - vslot(6)) * 4 : 12;          // float offset of columns 3-5 behind columns 0-2
        const float* tt = t[i];
        if (part == 0) {
            const float o0 = fmaf(-5.f, tt[2], fmaf(4.f, tt[0], tt[4]));
            const float p = fmaf(-4.f, tt[2], tt[4]);
            const float qq = fmaf(-4.f, tt[1], tt[3]);
            vst[wr] = o0;
            vst[wr + 4] = p + qq;
            vst[wr + 8] = p - qq;
        } else {
            const float p = tt[4] - tt[2];
            const float r = tt[3] - tt[1];
            const float o5 = fmaf(-5.f, tt[3], fmaf(4.f, tt[1], tt[5]));
            vst[wr + hi] = fmaf(2.f, r, p);
            vst[wr + hi + 4] = fmaf(-2.f, r, p);
            vst[wr + hi + 8] = o5;
        }
    };

    // ---- this workgroup's work items.  item(): logical tile / channel-block range / part of the item at position jq, false when none
    const int64_t xbase = (bid & 7) * a.per_xcd;
    const int nbp = a.ncb / a.split_f;                      // channel blocks per part
    int cb0 = 0, cb1 = a.ncb, part = -1, tail_idx = 0;      // current item: block range; part < 0: a whole tile
    int64_t logical = 0;
    struct Item { bool ok; int64_t logical; int cb0, cb1, part, tail_idx; };
    auto item = [&](int64_t jq) {
        Item r;
        if (!SPLIT) {
            r.logical = xbase + jq; r.cb0 = 0; r.cb1 = a.ncb; r.part = -1; r.tail_idx = 0;
            r.ok = jq < a.full_x && r.logical < a.nwg;
        } else {
            r.tail_idx = (int)(jq / a.split_f);
            r.part = (int)(jq - (int64_t)r.tail_idx * a.split_f);
            r.logical = xbase + a.full_x + r.tail_idx;
            r.cb0 = r.part * nbp;
            r.cb1 = r.cb0 + nbp;
            r.ok = jq < a.slots && r.tail_idx < a.ntail && r.logical < a.nwg;
        }
        return r;
    };
    int64_t jt = bid >> 3;
    {
        const Item it0 = item(jt);
        if (!it0.ok) return;
        logical = it0.logical; cb0 = it0.cb0; cb1 = it0.cb1; part = it0.part; tail_idx = it0.tail_idx;
    }
    describe(logical);
#pragma unroll
    for (int i = 0; i < S::NLD; ++i) { pr[i] = ld_x(cb0, i); pr2[i] = ld_x(cb0 + 1, i); }

#ifdef DINV_W4_TIMING
    if (a.stagger > 0) {
        const long long t0 = (long long)__builtin_readcyclecounter();
        const long long wait = (long long)((bid >> 3) & 3) * a.stagger / 4;
        while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    int tile_k = 0;
#define DINV_STAMP(i) do { if (a.dbg && tid == 0 && tile_k < 4) a.dbg[(bid * 4 + tile_k) * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DINV_STAMP(i) do { } while (0)
#endif
    for (;;) {
        DINV_STAMP(0);
        const Item nxt = item(jt + a.slots);
        const bool more = !SPLIT && nxt.ok;
        const float* const wt = wsrc;
        const int ect = ct;
        const uint32_t epw = pw;
        const int ecb0 = cb0, ecb1 = cb1, epart = part, etail = tail_idx;
        // ---- tile prologue: the first two raw blocks (requested during the previous tile's epilogue) go to LDS, block 0 is
        // transformed with nothing to hide behind
#pragma unroll
        for (int i = 0; i < S::NLD; ++i)
            if (stage_ok(i)) { st4(R0 + loff(i), pr[i]); st4(R0 + S::RAWF + loff(i), pr2[i]); }
        u[0] = ld_u(wt, ecb0, 0);
        u[1] = ld_u(wt, ecb0, 1);
        if constexpr (BF3) { ulo[0] = ld_ul(wt, ecb0, 0); ulo[1] = ld_ul(wt, ecb0, 1); u[2] = ld_u(wt, ecb0, 2); ulo[2] = ld_ul(wt, ecb0, 2); }
        lds_barrier();
        DINV_STAMP(1);
#pragma unroll
        for (int j = 0; j < 6; ++j) { tr_read(R0, j); tr_rows(j); }
#pragma unroll
        for (int i = 0; i < 3; ++i) { tr_cols(V0, i, 0); tr_cols(V0, i, 1); }
        lds_barrier();
        DINV_STAMP(2);
        v[0] = ld4(V0 + vrd);
        if constexpr (BF3) {
            v[1] = ld4(V0 + vrd + 4);
            split3(v[0], vh[0], vm[0], vl[0]);
        }

        // ---- one 8-channel block = 36 slots of one MFMA + its share of: the U ring (3 points ahead, straight from L2), the
        // V ring (1 point ahead), staging block cb+2 (loads in slots 0.., LDS writes in slots 24..), transforming block cb+1
        // (columns in slots 0-17, rows and V writes in slots 18-29), the barrier in slot 30
        // (FIRST: the tile's first block starts every accumulator from the constant 0 - no registers are zeroed)
        auto block = [&](int cb, auto par, auto first_) {
            constexpr int P = decltype(par)::value;
            constexpr bool FIRST = decltype(first_)::value;
            const float* vcur = V0 + P * VBUF;
            float* vnxt = V0 + (1 - P) * VBUF;
            const float* rnxt = R0 + (1 - P) * S::RAWF;
            float* rst = R0 + P * S::RAWF;
            const int cbu = cb + 1 < ecb1 ? cb + 1 : ecb1 - 1;   // past the end: re-read the last block (unused data)
            const int cbs = cb + 2 < ecb1 ? cb + 2 : ecb1 - 1;
            if constexpr (BF3) {
                // 27 slots = 9 points x 3 bf16 MFMAs.  The 8 K slots of a lane = [part X of its 4 channels | part Y of its 4 channels];
                // with U = (um, uh, ul) and V = (vm, vh, vl) as six registers each, the three operand pairs are register WINDOWS:
                //   A = (um, uh) x B = (vm, vh)  ->  um vm + uh vh
                //   A = (um, uh) x B = (vh, vl)  ->  um vh + uh vl
                //   A = (uh, ul) x B = (vm, vh)  ->  uh vm + ul vh          = the six products, nothing duplicated
                // U arrives split (packed on the host), two points ahead; V: raw fp32 from the stage two points ahead, split one
                // point ahead (point pt's slots split point pt + 1), so that no MFMA waits for the conversion of its own operand
                static_for<27>([&](auto s_) {
                    constexpr int SL = decltype(s_)::value, pt = SL / 3, j = SL % 3;
                    constexpr int cur = (pt + P) % 2, nx = (pt + 1 + P) % 2, ur = pt % 3;
                    if constexpr (SL == 24) lds_barrier();
                    uint4 A, B;
                    if constexpr (j == 0) { A = __builtin_bit_cast(uint4, u[ur]); B = pair(vm[cur], vh[cur]); }
                    if constexpr (j == 1) { A = __builtin_bit_cast(uint4, u[ur]); B = pair(vh[cur], vl[cur]); }
                    if constexpr (j == 2) {
                        A = make_uint4(__float_as_uint(u[ur].z), __float_as_uint(u[ur].w), ulo[ur].x, ulo[ur].y);
                        B = pair(vm[cur], vh[cur]);
                    }
                    // (the matrix instruction is a pure value to the compiler and would drift across the slot boundaries: its
                    // accumulator is made opaque on both sides, which ties it to this slot)
                    if constexpr (FIRST && j == 0) acc[pt] = mfma_bf16(A, B, (f32x16)(0.f));
                    else { DINV_PIN(acc[pt]); acc[pt] = mfma_bf16(A, B, acc[pt]); }
                    DINV_PIN(acc[pt]);
                    if constexpr (j == 2) {           // U ring: three points ahead, straight from L2 (this point's entry is free now)
                        constexpr int k = pt + 3;
                        if constexpr (k < 9) { u[k % 3] = ld_u(wt, cb, k); ulo[k % 3] = ld_ul(wt, cb, k); }
                        else { u[k % 3] = ld_u(wt, cbu, k - 9); ulo[k % 3] = ld_ul(wt, cbu, k - 9); }
                    }
                    // V ring: raw(pt + 2) replaces raw(pt) (split during point pt - 1); the next block's first two points are
                    // behind the barrier
                    if constexpr (pt < 7 && j == 0) v[cur] = ld4(vcur + vrd + (pt + 2) * 4);
                    if constexpr (pt == 8 && j == 0) { v[nx] = ld4(vnxt + vrd); v[cur] = ld4(vnxt + vrd + 4); }
                    // (the inputs of every piece of vector work are made opaque in its slot: pure arithmetic would drift to the
                    // slot of its operands' loads and wait for them there)
                    if constexpr (pt < 8 && j == 0) { DINV_PIN2(v[nx].x, v[nx].y); split3_pair(v[nx].x, v[nx].y, vh[nx].x, vm[nx].x, vl[nx].x); }
                    if constexpr (pt < 8 && j == 1) { DINV_PIN2(v[nx].z, v[nx].w); split3_pair(v[nx].z, v[nx].w, vh[nx].y, vm[nx].y, vl[nx].y); }
                    if constexpr (pt == 8 && j == 2) { DINV_PIN2(v[nx].x, v[nx].z); split3(v[nx], vh[nx], vm[nx], vl[nx]); }
                    if constexpr (SL < S::NLD) pr[SL] = ld_x(cbs, SL);
                    if constexpr (SL < 18 && SL % 3 == 0) tr_read(rnxt, SL / 3);
                    if constexpr (SL < 18 && SL % 3 == 2) { DINV_PIN2(e7[0], e7[3]); DINV_PIN2(e7[4], e7[5]); tr_rows(SL / 3); }
                    if constexpr (SL >= 18 && SL < 24) {
                        constexpr int ti = (SL - 18) / 2;
                        DINV_PIN2(t[ti][1], t[ti][2]); DINV_PIN2(t[ti][3], t[ti][4]);
                        tr_cols(vnxt, ti, (SL - 18) % 2);
                    }
                    if constexpr (SL >= 20 && SL < 20 + S::NLD)
                        if (stage_ok(SL - 20)) st4(rst + loff(SL - 20), pr[SL - 20]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                return;
            }
            static_for<36>([&](auto s_) {
                constexpr int SL = decltype(s_)::value, pt = SL / 4, m = SL % 4;
                if constexpr (SL == 30) lds_barrier();
                if constexpr (FIRST && m == 0)
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(u[pt % 3], m), comp(v[(pt + P) % 2], m), (f32x16)(0.f), 0, 0, 0);
                else
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(u[pt % 3], m), comp(v[(pt + P) % 2], m), acc[pt], 0, 0, 0);
                if constexpr (m == 0) {
                    constexpr int k = pt + 2;
                    if constexpr (k < 9) u[k % 3] = ld_u(wt, cb, k);
                    else u[k % 3] = ld_u(wt, cbu, k - 9);
                }
                if constexpr (m == 1) {
                    constexpr int k = pt + 1;
                    if constexpr (k < 9) v[(k + P) % 2] = ld4(vcur + vrd + k * 4);
                    else v[(k + P) % 2] = ld4(vnxt + vrd);            // next block's point 0: behind the barrier
                }
                if constexpr (SL < S::NLD) pr[SL] = ld_x(cbs, SL);
                if constexpr (SL < 18 && SL % 3 == 0) tr_read(rnxt, SL / 3);
                if constexpr (SL < 18 && SL % 3 == 2) tr_rows(SL / 3);
                if constexpr (SL >= 18 && SL < 30 && (SL - 18) % 2 == 0) tr_cols(vnxt, (SL - 18) / 4, ((SL - 18) / 2) % 2);
                if constexpr (SL >= 24 && SL < 24 + S::NLD)
                    if (stage_ok(SL - 24)) st4(rst + loff(SL - 24), pr[SL - 24]);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        {
            using P0 = std::integral_constant<int, 0>;
            using P1 = std::integral_constant<int, 1>;
            block(ecb0, P0{}, std::true_type{});         // even number of channel blocks per item: checked on the host
            block(ecb0 + 1, P1{}, std::false_type{});
#pragma unroll 1
            for (int cb = ecb0 + 2; cb < ecb1; cb += 2) {
                block(cb, P0{}, std::false_type{});
                block(cb + 1, P1{}, std::false_type{});
            }
        }

        DINV_STAMP(3);
        // ---- epilogue.  Wave (c2, q) holds the point slots 9q .. 9q+8 of cout half c2: a full Winograd row, then half a row,
        //   q = 0: row 0, row 1 cols 0-2     q = 1: row 2, row 1 cols 3-5     q = 2: row 3, row 4 cols 0-2     q = 3: row 5, row 4 cols 3-5
        // and reduces them along the row: s = (row of M) A, four values per (part of a) row.  Two exchange rounds through LDS
        // in which EVERY wave writes and every thread finishes: round A = the full rows (0, 2, 3, 5), round B = the half
        // rows (the two halves of rows 1 and 4 are added by the reader).  The finishing thread (wave = 8-channel block of the
        // 64 couts, lane = (tile t8 of 8, output column j, channel half h)) owns one output column of 4 tiles x 4 rows, so a
        // store instruction covers 8 tiles x 4 columns x 32 bytes = 1 KB of consecutive memory; its 16 outputs live in
        // registers across the rounds, initialised with the residual (requested before the first barrier) or with zero.
        // E (float4 units): [writer wave 8][quad g 4][it 4][64], slot of (t8, j, h) = 8 t8 + ((2 j + h + t8) & 7): the
        // rotation makes the writers' groups of 8 lanes (t8 = 0..7 at fixed j, h) hit 8 different 16-byte bank groups
        int te = tid;
        DINV_OPAQUE(te);
        const int fl = te & 63, ft8 = fl >> 3, fj = (fl >> 1) & 3, fh = fl & 1;
        const int cblk_r = wave, c2r = wave >> 2, gr = wave & 3;
        const int64_t cbo = ((int64_t)ect * 8 + cblk_r) * xcs;           // channel block this wave finishes
        const int erd = (((c2r * 4) * 4 + gr) * 4 * 64 + ft8 * 8 + ((2 * fj + fh + ft8) & 7)) * 4;   // + (qq * 16 + it) * 256 floats
        const int wt8 = l31 & 7;
        const int ewr = ((wave * 16 + (l31 >> 3)) * 64 + wt8 * 8) * 4;    // + g * 1024 + ((2 j + h + t8) & 7) * 4 floats
        // byte offsets (inside one channel block) of the thread's output pixel (row 0 of tile it, column j); tiles of partial
        // rectangles / beyond the batch get an out-of-range offset: buffer loads return 0, buffer stores are dropped
        uint32_t obase[4];
        bool ovalid[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int fp = it * 8 + ft8;
            const int fsub = fp / S::PT, fty = (fp % S::PT) / TW, ftx = fp % TW;
            const uint32_t fs = epw * S::NSUB + fsub;
            const uint32_t fb = a.d_img.div(fs);
            const uint32_t frem = fs - fb * (uint32_t)per_img;
            const uint32_t ftyb = a.d_ntx.div(frem), ftxb = frem - ftyb * a.ntx;
            const int oy = 4 * (ftyb * TH + fty), ox = 4 * (ftxb * TW + ftx) + fj;
            ovalid[it] = fs < (uint32_t)a.nsr && oy < a.g.h && ox < a.g.w;
            obase[it] = (uint32_t)((a.g.sl + (int64_t)fb * a.g.plane + (int64_t)(oy + 1) * a.g.wp + ox + 1) * 32 + 16 * fh);
        }
        const uint32_t orow = (uint32_t)a.g.wp * 32u;
        auto out_off = [&](int it, int i) { return ovalid[it] ? obase[it] + (uint32_t)i * orow : 0xffffffffu; };
        float4 yo[4][4], rv[4][4];       // [tile it][output row i]: partial outputs, residual values

        // one accumulator register quad (= 4 couts of this lane half) of point k as a float4
        auto Q = [&](int k, int g) { return make_float4(acc[k][4 * g], acc[k][4 * g + 1], acc[k][4 * g + 2], acc[k][4 * g + 3]); };
        auto est = [&](int g, int j, float4 val) { st4(lds + ewr + g * 1024 + ((2 * j + h + wt8) & 7) * 4, val); };
        auto write_full = [&]() {       // accumulators 0-5
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 F0 = Q(0, g), F1 = Q(1, g), F2 = Q(2, g), F3 = Q(3, g), F4 = Q(4, g), F5 = Q(5, g);
                const float4 sa = add4(F1, F2), sb = sub4(F1, F2), sc = add4(F3, F4), sd = sub4(F3, F4);
                est(g, 0, add4(add4(F0, sa), sc));
                est(g, 1, fma4(2.f, sd, sb));
                est(g, 2, fma4(4.f, sc, sa));
                est(g, 3, add4(fma4(8.f, sd, sb), F5));
            }
        };
        auto write_half = [&]() {       // accumulators 6-8 (the branch on the wave's parity is outside the loops)
            if (q & 1) {          // row cols 3-5:  s = (H3 + H4, 2 (H3 - H4), 4 (H3 + H4), 8 (H3 - H4) + H5)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 H0 = Q(6, g), H1 = Q(7, g), H2 = Q(8, g);
                    const float4 c = add4(H0, H1), d = sub4(H0, H1);
                    est(g, 0, c);
                    est(g, 1, add4(d, d));
                    est(g, 2, make_float4(4.f * c.x, 4.f * c.y, 4.f * c.z, 4.f * c.w));
                    est(g, 3, fma4(8.f, d, H2));
                }
            } else {              // row cols 0-2:  s = (L0 + L1 + L2, L1 - L2, L1 + L2, L1 - L2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 H0 = Q(6, g), H1 = Q(7, g), H2 = Q(8, g);
                    const float4 c = add4(H1, H2), d = sub4(H1, H2);
                    est(g, 0, add4(H0, c));
                    est(g, 1, d);
                    est(g, 2, c);
                    est(g, 3, d);
                }
            }
        };
        auto erd4 = [&](int qq, int it) { return ld4(lds + erd + (qq * 16 + it) * 256); };

        lds_barrier();                  // every wave is done with the V / raw stages: the exchange buffer overlays them
        DINV_STAMP(4);
        write_full();
        if (NRES && !SPLIT) {     // the six full-row accumulators are dead: room for the 16 residual values, which travel while
                                    // the first round is finished and the second one written
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[it][i] = ld4_so(a.res + cbo, out_off(it, i), 0u);
        }
        lds_barrier();
        DINV_STAMP(5);
#pragma unroll
        for (int it = 0; it < 4; ++it) {    // rows 0, 2, 3, 5:  Y0 = S0 + S2 + S3, Y1 = 2 S3 - S2, Y2 = S2 + 4 S3, Y3 = 8 S3 - S2 + S5
            const float4 S0 = erd4(0, it), S2 = erd4(1, it), S3 = erd4(2, it), S5 = erd4(3, it);
            yo[it][0] = add4(add4(S0, S2), S3);
            yo[it][1] = sub4(add4(S3, S3), S2);
            yo[it][2] = fma4(4.f, S3, S2);
            yo[it][3] = add4(fma4(8.f, S3, make_float4(-S2.x, -S2.y, -S2.z, -S2.w)), S5);
            __builtin_amdgcn_sched_barrier(0);      // one tile's reads in flight at a time: three accumulators and the residuals are live
        }
        DINV_STAMP(6);
        jt += a.slots;
        if (more) {
            logical = nxt.logical; cb0 = nxt.cb0; cb1 = nxt.cb1; part = nxt.part; tail_idx = nxt.tail_idx;
            describe(logical);
        }
        lds_barrier();
        DINV_STAMP(7);
        write_half();
        __builtin_amdgcn_sched_barrier(0);
        // every accumulator is dead: the next tile's first two raw blocks are requested now and land while this tile's outputs
        // are finished and stored
        if (more) {
#pragma unroll
            for (int i = 0; i < S::NLD; ++i) { pr[i] = ld_x(cb0, i); pr2[i] = ld_x(cb0 + 1, i); }
        }
        lds_barrier();
        DINV_STAMP(8);
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + cbo, 0, 0xffffffff, 0x00020000);
        auto emit = [&](int it, int i, float4 val) {      // ReLU / residual / store of output row i of tile it
            uint32_t ob = obase[it];
            DINV_OPAQUE(ob);        // recompute the row offsets here instead of keeping the residual loads' 16 alive
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yrsrc,
                                                   ovalid[it] ? ob + (uint32_t)i * orow : 0xffffffffu, 0, 0);
        };
        // rows 1, 4 (two halves each):  Y0 += S1 + S4, Y1 += S1 - 2 S4, Y2 += S1 + 4 S4, Y3 += S1 - 8 S4
        if constexpr (!SPLIT) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 S1 = add4(erd4(0, it), erd4(1, it)), S4 = add4(erd4(2, it), erd4(3, it));
                float4 o[4];
                o[0] = add4(yo[it][0], add4(S1, S4));
                o[1] = add4(yo[it][1], fma4(-2.f, S4, S1));
                o[2] = add4(yo[it][2], fma4(4.f, S4, S1));
                o[3] = add4(yo[it][3], fma4(-8.f, S4, S1));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 val = o[i];
                    if (RELU) val = make_float4(fmaxf(val.x, 0.f), fmaxf(val.y, 0.f), fmaxf(val.z, 0.f), fmaxf(val.w, 0.f));
                    if (NRES) val = add4(val, rv[it][i]);
                    emit(it, i, val);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ---- one part of a tail tile: publish the partial outputs, take a ticket; the last part to arrive adds all parts in
            // part order (deterministic), applies ReLU / residual and stores (cdna_hip_programming.md Guideline 16: plain stores ->
            // workgroup barrier -> one lane's agent-scope release -> ticket; last arriver: agent-scope acquire -> plain loads)
            const int64_t tslot = (bid & 7) * (int64_t)a.ntail + etail;
            float4* const pb = reinterpret_cast<float4*>(a.part_buf) + (tslot * a.split_f) * 8192 + (int64_t)wave * 1024 + lane;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 S1 = add4(erd4(0, it), erd4(1, it)), S4 = add4(erd4(2, it), erd4(3, it));
                float4* const pp = pb + (int64_t)epart * 8192 + it * 256;
                pp[0] = add4(yo[it][0], add4(S1, S4));
                pp[64] = add4(yo[it][1], fma4(-2.f, S4, S1));
                pp[128] = add4(yo[it][2], fma4(4.f, S4, S1));
                pp[192] = add4(yo[it][3], fma4(-8.f, S4, S1));
                __builtin_amdgcn_sched_barrier(0);
            }
            int* const flag = reinterpret_cast<int*>(lds + S::LDSF);
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef DINV_EMU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                *flag = atomicAdd(a.tickets + tslot, 1);
            }
            __syncthreads();
            if (*flag == a.split_f - 1) {
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    a.tickets[tslot] = 0;       // ready for the next launch
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int i = 0; i < 4; ++i) yo[it][i] = pb[it * 256 + i * 64];
                for (int pp = 1; pp < a.split_f; ++pp)
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int i = 0; i < 4; ++i) yo[it][i] = add4(yo[it][i], pb[(int64_t)pp * 8192 + it * 256 + i * 64]);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float4 val = yo[it][i];
                        if (RELU) val = make_float4(fmaxf(val.x, 0.f), fmaxf(val.y, 0.f), fmaxf(val.z, 0.f), fmaxf(val.w, 0.f));
                        if (NRES) val = add4(val, ld4_so(a.res + cbo, out_off(it, i), 0u));     // (tail tiles only: latency not hidden)
                        emit(it, i, val);
                    }
            }
        }
        DINV_STAMP(9);
        lds_barrier();
        DINV_STAMP(10);
#ifdef DINV_W4_TIMING
        ++tile_k;
#endif
        if (!more) break;
    }
#undef DINV_STAMP
}

// what the calling thread's last launch did with its tail (dinv_conv3x3_winograd4_last_split: tests, diagnostics)
struct LastSplit { int32_t split_f, ntail; };
LastSplit& last_split() {
    static thread_local LastSplit v{1, 0};
    return v;
}

template <int TH, int TW, bool RELU, int NRES, bool BF3>
int launch_shape(W4Args a, hipStream_t st) {
    using S = Shape4<TH, TW>;
    a.nty = (int32_t)ceil_div(a.g.h / 4, TH);
    a.ntx = (int32_t)ceil_div(a.g.w / 4, TW);
    a.nsr = (int64_t)a.g.batch * a.nty * a.ntx;
    a.npw = (int32_t)ceil_div(a.nsr, S::NSUB);
    a.nwg = (int64_t)a.npw * a.nct;
    a.ct_major = (int64_t)a.nct * a.ncb * UBLK * 4 > (3ll << 20) ? 1 : 0;
    a.d_npw = make_fastdiv((uint32_t)a.npw);
    a.per_xcd = ceil_div(a.nwg, 8);
    DINV_REQUIRE(a.nsr + 64 < (1ll << 31) && a.nwg < (1ll << 31), "winograd F(4,3) conv: too many tiles for 32-bit indexing");
    a.d_img = make_fastdiv((uint32_t)(a.nty * a.ntx));
    a.d_ntx = make_fastdiv((uint32_t)a.ntx);
    a.d_nct = make_fastdiv((uint32_t)a.nct);
    const size_t shm = S::LDSF * sizeof(float) + 16;   // + the ticket word of the tail split
    static std::atomic<uint64_t> configured{0};   // per instantiation: bit d = attribute set on device d
    auto kern = conv3x3_wino4_kernel<TH, TW, RELU, NRES, false, BF3>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(3, "hipGetDevice failed");
    const uint64_t bit = 1ull << (dev & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return fail(3, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    const int cpx = cus_per_xcd(dev);
    DINV_REQUIRE(cpx <= 64, "winograd F(4,3) conv: the ticket area of the tail split holds 64 tiles per XCD (device has %d CUs per XCD)", cpx);
    // whole rounds of cpx tiles per XCD, then the tail: cut along the input channels when a workspace was given
    const int64_t ntail = a.per_xcd % cpx;
    a.split_f = 1;
    if (a.part_buf && ntail > 0)
        for (int f = 8; f >= 2; f /= 2)
            if (ntail * f <= cpx && a.ncb % f == 0 && (a.ncb / f) % 2 == 0) { a.split_f = f; break; }
    a.ntail = a.split_f > 1 ? (int32_t)ntail : 0;
    a.full_x = (int32_t)(a.per_xcd - a.ntail);
    last_split() = {a.split_f, a.ntail};
    if (a.full_x > 0) {
        a.slots = (int32_t)(a.full_x < cpx ? a.full_x : cpx);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.slots * 8)), dim3(NTHR), shm, st, a);
        DINV_CHECK_LAUNCH();
    }
    if (a.split_f > 1) {
        auto kern_s = conv3x3_wino4_kernel<TH, TW, RELU, NRES, true, BF3>;
        static std::atomic<uint64_t> configured_s{0};
        if (!(configured_s.load(std::memory_order_relaxed) & bit)) {
            if (hipFuncSetAttribute((const void*)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                return fail(3, "hipFuncSetAttribute(max dynamic LDS) failed");
            configured_s.fetch_or(bit, std::memory_order_relaxed);
        }
        a.slots = (int32_t)(a.ntail * a.split_f);
        hipLaunchKernelGGL(kern_s, dim3((unsigned)(a.slots * 8)), dim3(NTHR), shm, st, a);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

template <bool RELU, int NRES, bool BF3>
int launch_any(const W4Args& a, hipStream_t st) {
    const int tyn = a.g.h / 4, txn = a.g.w / 4;
    // rectangle shape with the least padded-tile waste; ties go to the largest rectangle (fewest halo loads)
    const int shapes[3][2] = {{4, 8}, {4, 4}, {2, 2}};
    int best = 0;
    double bw = 1e30;
    for (int i = 0; i < 3; ++i) {
        const double w = (double)ceil_div(tyn, shapes[i][0]) * shapes[i][0] * ceil_div(txn, shapes[i][1]) * shapes[i][1];
        if (w < bw * 0.999) { bw = w; best = i; }
    }
    switch (best) {
        case 0: return launch_shape<4, 8, RELU, NRES, BF3>(a, st);
        case 1: return launch_shape<4, 4, RELU, NRES, BF3>(a, st);
        default: return launch_shape<2, 2, RELU, NRES, BF3>(a, st);
    }
}

}  // namespace

#ifdef DINV_W4_TIMING
static long long* g_w4_dbg = nullptr;
extern "C" void dinv_debug_wino4_timing(long long* p) { g_w4_dbg = p; }
#endif

// workspace of the tail split: per XCD one ticket word and up to (compute units per XCD) part buffers of 64 couts x 512 pixels
static size_t w4_ws_tickets() { return 8 * 64 * sizeof(int32_t); }
extern "C" size_t dinv_conv3x3_winograd4_workspace_bytes(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return w4_ws_tickets() + (size_t)8 * cus_per_xcd(dev) * 8192 * 16;
}

extern "C" int dinv_conv3x3_winograd4_last_split(int32_t* split_f, int32_t* n_tail_tiles) {
    if (split_f) *split_f = last_split().split_f;
    if (n_tail_tiles) *n_tail_tiles = last_split().ntail;
    return 0;
}

static int winograd4_any(bool bf3, const dinv_act_geom* g, const float* x, const float* w_wino4, int32_t cin,
                                      int32_t cout, float* y, const float* res1, int32_t relu, void* workspace,
                                      size_t workspace_bytes, dinv_stream_t stream) {
    if (check_geom(g)) return 1;
    DINV_REQUIRE(x && w_wino4 && y, "null pointer");
    DINV_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "winograd F(4,3) conv needs cin %% 16 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(g->height % 4 == 0 && g->width % 4 == 0, "winograd F(4,3) conv needs height and width to be multiples of 4 (got %dx%d)",
                 g->height, g->width);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(g->cs * 32 < (1ll << 32), "winograd F(4,3) conv: one channel block must stay below 4 GB (32-bit buffer offsets)");
    W4Args a{};
    a.g = make_geom(*g);
    a.x = x; a.w = w_wino4; a.y = y; a.res = res1;
    a.ncb = cin / 8; a.nct = cout / 64;
    if (workspace) {
        DINV_REQUIRE(workspace_bytes >= dinv_conv3x3_winograd4_workspace_bytes(), "winograd F(4,3) conv: workspace too small");
        a.tickets = reinterpret_cast<int32_t*>(workspace);
        a.part_buf = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + w4_ws_tickets());
    }
#ifdef DINV_W4_TIMING
    a.dbg = g_w4_dbg;
    a.stagger = getenv("DINV_W4_STAGGER") ? atoi(getenv("DINV_W4_STAGGER")) : 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (bf3) {
        if (relu) return launch_any<true, 0, true>(a, st);
        if (res1) return launch_any<false, 1, true>(a, st);
        return launch_any<false, 0, true>(a, st);
    }
    if (relu) return launch_any<true, 0, false>(a, st);
    if (res1) return launch_any<false, 1, false>(a, st);
    return launch_any<false, 0, false>(a, st);
}

extern "C" int dinv_conv3x3_winograd4(const dinv_act_geom* g, const float* x, const float* w_wino4, int32_t cin,
                                      int32_t cout, float* y, const float* res1, int32_t relu, void* workspace,
                                      size_t workspace_bytes, dinv_stream_t stream) {
    return winograd4_any(false, g, x, w_wino4, cin, cout, y, res1, relu, workspace, workspace_bytes, stream);
}

extern "C" int dinv_conv3x3_winograd4_bf16x3(const dinv_act_geom* g, const float* x, const void* w_wino4x3, int32_t cin,
                                             int32_t cout, float* y, const float* res1, int32_t relu, void* workspace,
                                             size_t workspace_bytes, dinv_stream_t stream) {
    return winograd4_any(true, g, x, static_cast<const float*>(w_wino4x3), cin, cout, y, res1, relu, workspace, workspace_bytes, stream);
}
