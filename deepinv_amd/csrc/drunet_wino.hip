// DRUNet 3x3 convolution, Winograd F(2x2, 3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Same operator as conv3x3_kernel in drunet.hip (nn.Conv2d 3x3 s1 p1 no bias inside the ResBlocks, reference
// deepinv/models/drunet.py:403-434), same "padded pixel rows, channels blocked by 8" activation layout, but the
// MFMA work is cut 2.25x: every 2x2 output tile needs 16 multiplies per (ci, co) instead of 36.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// Measured facts that shape the kernel (scripts/ubench/mfma_coissue.hip): the fp32 MFMA shares the vector ALU,
// so every VALU instruction a wave issues costs MFMA time (4.5 cycles with one wave per SIMD, 2.2 with two;
// LDS reads are free).  The design therefore minimises VALU instructions per MFMA and runs two waves per SIMD:
//
// * U = G g G^T is precomputed on the host (fp64 -> fp32), packed [cout/64][cin/8][ci 8][co 64][xi 16].
// * Workgroup = 8 waves = 64 couts x 64 tile positions x 16 Winograd points.  Wave (xr, wq) owns Winograd ROW
//   xr (4 points) for all 64 couts and 32 positions: 8 accumulators (128 AGPRs), two waves per SIMD.  Row xr
//   of V = B^T d B needs only two rows of the patch: 8 ds_read_b128 + 16 FMAs + 16 adds feed 32 MFMAs per
//   8-channel block (1 VALU per MFMA; the all-points-per-wave layout needs 4).
// * Per 8-channel block the raw input region (zero border included, so no edge cases) and the 32 KB slice of U
//   are staged global -> registers -> LDS, double buffered, one LDS-only barrier per block; the 6 ds_write_b128
//   per wave are spread over the block (U parts 0-2 run one block ahead, U part 3 and the pixels two blocks
//   ahead), every staging register holds a load issued a full block earlier; buffer loads with SGPR bases (no
//   address VALU); invalid pixels are redirected to a zero border pixel (no selects).
// * Persistent workgroups (one per CU, 32 per XCD walking a contiguous range of the tile order); the next tile's
//   first loads are issued as soon as the accumulators are dead.
// * The 64 positions of a workgroup are NSUB rectangles of TH x TW tiles enumerated over (batch, tile rows,
//   tile cols), so the 40x40 and 80x80 levels fill the MFMA columns as well as 320x320 does.
// * Epilogue: each wave reduces its row to s = M A (2 values), the four row-waves exchange s through LDS and
//   wave xr finishes A^T s for channel sub-block xr: fused ReLU / residual add, float4 stores of interior
//   pixels only.
#include "drunet_common.hpp"
#include <atomic>
#include <type_traits>

using namespace dinv;
using namespace dinv_drunet;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int WP = 20;                 // LDS pitch of one (ci, co) row of U: 16 xi + 4 pad (conflict-free b128)
constexpr int WLDS = 8 * 64 * WP;      // floats of U per channel block in LDS
constexpr int RP = 12;                 // LDS pitch of one staged pixel: 8 channels + 4 pad
constexpr int NTHR = 512;
constexpr int EXCH = 8 * 4 * 2 * 2 * 256;   // floats of the epilogue exchange buffer (128 KB)

struct WinoArgs {
    Geom g;
    const float* x;
    const float* w;
    float* y;
    const float* res;
    int32_t ncb, nct, nty, ntx;
    int64_t nsr, nwg, per_xcd;
    int32_t slots;   // resident workgroups per XCD
    FastDiv d_img, d_ntx, d_nct;   // / (nty*ntx), / ntx, / nct
#ifdef DINV_WINO_TIMING
    long long* dbg;  // phase timestamps (s_memtime) of wave 0, 8 per tile, first 4 tiles of every workgroup
#endif
};

template <int TH, int TW>
struct Shape {
    static constexpr int PT = TH * TW;             // tile positions per rectangle
    static constexpr int NSUB = 64 / PT;           // rectangles per workgroup
    static constexpr int RH = 2 * TH + 2;          // staged rows per rectangle
    static constexpr int RW2 = TW + 1;             // staged columns per parity
    static constexpr int RAWF = NSUB * RH * 2 * RW2 * RP;
    static constexpr int RAW4 = NSUB * RH * 2 * RW2 * 2;   // float4 loads to stage one block
    static constexpr int NLD = (RAW4 + NTHR - 1) / NTHR;
    static constexpr int BUF = WLDS + RAWF;        // floats per LDS buffer
    static constexpr int LDSF = 2 * BUF > EXCH ? 2 * BUF : EXCH;
    static_assert(PT <= 64 && 64 % PT == 0, "rectangle must divide the 64 positions");
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also fences global memory, i.e. it would wait for
// the in-flight staging / residual loads and the output stores, which are meant to overlap the next phase.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int TH, int TW, bool RELU, int NRES>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_wino_kernel(WinoArgs a) {
    using S = Shape<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int l31 = lane & 31, h = lane >> 5;
    const int xr = wave & 3, wq = wave >> 2;

    // Persistent workgroups (one per CU: LDS and registers allow no second one): slot j of XCD x walks tiles
    // x*per_xcd + j + k*slots of the logical order, so each XCD owns a contiguous range of it and the cout
    // tiles of one position tile (and neighbouring position tiles) share an L2.
    const int64_t bid = blockIdx.x;
    const int64_t per_img = (int64_t)a.nty * a.ntx;
    const int64_t xcs = a.g.cs * 8;
    const int woff = (tid >> 2) * WP + (tid & 3) * 4;   // LDS offset of weight float4 #tid (+ i*128*WP for #tid+512i)

    // ---- this lane's operand addresses (tile independent)
    const int p = wq * 32 + l31;
    const int sub = p / S::PT, ty = (p % S::PT) / TW, tx = p % TW;
    // row xr of B^T d:  t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3   ->   t = d[ra] + sigma * d[rb]
    const int ra = xr == 0 ? 0 : xr == 2 ? 2 : 1;
    const int rb = xr == 2 ? 1 : xr == 3 ? 3 : 2;
    const float sigma = xr == 1 ? 1.f : -1.f;
    const int rbase = WLDS + ((sub * S::RH + 2 * ty) * 2 * S::RW2 + tx) * RP + 4 * h;
    const int rA = rbase + ra * 2 * S::RW2 * RP, rB = rbase + rb * 2 * S::RW2 * RP;
    const int abase = ((4 * h) * 64 + l31) * WP + 4 * xr;

    // ---- staging descriptors: LDS side is tile independent, global side is recomputed per tile
    int loff[S::NLD];
#pragma unroll
    for (int i = 0; i < S::NLD; ++i) {
        const int e = tid + NTHR * i;
        const int half = e & 1, px = e >> 1;
        const int c = px % (2 * S::RW2);
        const int r = (px / (2 * S::RW2)) % S::RH;
        const int sb = px / (2 * S::RW2 * S::RH);
        loff[i] = WLDS + (((sb * S::RH + r) * 2 + (c & 1)) * S::RW2 + (c >> 1)) * RP + half * 4;
    }
    // current tile (no suffix) and next tile of this workgroup (suffix n)
    uint32_t goff[S::NLD], goffn[S::NLD];   // byte offsets inside one channel block of x
    const float *wsrc, *wsrcn;              // packed U of the tile's cout block (+ tid*16 bytes per lane)
    int ct, ctn;                            // tile coordinates: cout block, position group
    uint32_t pw, pwn;
    auto advance = [&]() {                  // next -> current
        ct = ctn; pw = pwn; wsrc = wsrcn;
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) goff[i] = goffn[i];
    };
    auto describe = [&](int64_t logical) {  // fills the "next" set
        pwn = a.d_nct.div((uint32_t)logical);
        ctn = (int)((uint32_t)logical - (uint32_t)pwn * (uint32_t)a.nct);
        wsrcn = a.w + (int64_t)ctn * a.ncb * 8192;
        int t = tid;
        asm volatile("" : "+v"(t));   // recompute the per-lane constants per tile instead of keeping them live
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) {
            const int e = t + NTHR * i;
            const int half = e & 1, px = e >> 1;
            const int c = px % (2 * S::RW2);
            const int r = (px / (2 * S::RW2)) % S::RH;
            const int sb = px / (2 * S::RW2 * S::RH);
            const uint32_t s = (uint32_t)pwn * S::NSUB + sb;
            const uint32_t b = a.d_img.div(s);
            const uint32_t rem = s - b * (uint32_t)per_img;
            const uint32_t tyb = a.d_ntx.div(rem), txb = rem - tyb * a.ntx;
            const int gr = 2 * tyb * TH + r, gc = 2 * txb * TW + c;
            const bool ok = e < S::RAW4 && s < (uint32_t)a.nsr && gr < a.g.hp && gc < a.g.wp;
            // out-of-frame pixels read the (always zero) top-left border pixel of image 0 instead
            goffn[i] = 4u * (ok ? (uint32_t)((a.g.sl + (int64_t)b * a.g.plane + (int64_t)gr * a.g.wp + gc) * 8 + half * 4)
                              : (uint32_t)(a.g.sl * 8));
        }
    };

    f32x16 acc[2][4];
    float4 pr[S::NLD], pwt[4];
    float4 uA[2], uB[2], uC[2], uD[2], uE[2];
    float vA[4], vB[4];
    float tc[4][4];   // row xr of B^T d for the current block: [channel m][col j]

#ifdef DINV_WINO_TIMING
    int tile_k = 0;
#define DINV_STAMP(i) do { if (a.dbg && tid == 0 && tile_k < 4) a.dbg[(bid * 4 + tile_k) * 32 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DINV_STAMP(i) do { } while (0)
#endif
#define DINV_MFMA2(U, V, c2_, j0_)                                                                              \
    do {                                                                                                         \
        acc[c2_][j0_] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(U[c2_], j0_), V[j0_], acc[c2_][j0_], 0, 0, 0); \
        acc[c2_][(j0_) + 1] =                                                                                    \
            __builtin_amdgcn_mfma_f32_32x32x2f32(comp(U[c2_], (j0_) + 1), V[(j0_) + 1], acc[c2_][(j0_) + 1], 0, 0, 0); \
    } while (0)
#define DINV_VCALC(V, T)                                                                                         \
    do { V[0] = T[0] - T[2]; V[1] = T[1] + T[2]; V[2] = T[2] - T[1]; V[3] = T[1] - T[3]; } while (0)
#define DINV_COL(j) ((((j) & 1) * S::RW2 + ((j) >> 1)) * RP)
#define DINV_UREAD(U, buf, m)                                                                                    \
    do { U[0] = ld4((buf) + abase + (m) * 64 * WP); U[1] = ld4((buf) + abase + ((m) * 64 + 32) * WP); } while (0)

    // buffer loads: uniform base in an SGPR resource + 32-bit per-lane byte offset, so staging costs no address
    // VALU (every VALU instruction is paid for in MFMA time); the per-lane offset stays below 4 GB by construction
    auto ld4_so = [](const float* sbase, uint32_t byte_off) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase), 0, 0xffffffff, 0x00020000);
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
    };
    auto fetch_w = [&](int cb, int i) { pwt[i] = ld4_so(wsrc + (int64_t)cb * 8192 + i * 2048, (uint32_t)tid * 16u); };
    auto fetch_r = [&](int cb, int i) { pr[i] = ld4_so(a.x + cb * xcs, goff[i]); };
    auto stash_w = [&](float* buf, int i) { st4(buf + woff + i * 128 * WP, pwt[i]); };
    auto stash_r = [&](float* buf, int i) {
        if (S::RAW4 % NTHR == 0 || i + 1 < S::NLD || tid + NTHR * i < S::RAW4) st4(buf + loff[i], pr[i]);
    };
    auto tcalc = [&](const float* buf) {
        float4 dA[4], dB[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { dA[j] = ld4(buf + rA + DINV_COL(j)); dB[j] = ld4(buf + rB + DINV_COL(j)); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) tc[m][j] = fmaf(comp(dB[j], m), sigma, comp(dA[j], m));
    };

    // ---- Staging parts: W0..W3 (weight float4s) and R0,R1 (pixels).  In steady state W0-W2 run one block ahead
    // (written to LDS in steps 0/1) and W3,R0,R1 two blocks ahead (written in steps 2/3, when the current buffer
    // is already dead), so the 6 ds_write_b128 per wave are spread over the whole block.
    // State at a tile's entry: pwt[0..2] = block 0 U parts 0-2, (w3x, r1x) = block 0 U part 3 + pixels,
    // (pwt[3], pr) = block 1 U part 3 + pixels.  Produced by first_loads() or by the last blocks of the previous tile.
    float4 w3x, r1x0, r1x1;
    static_assert(S::NLD <= 2, "staging assumes at most two pixel float4s per thread");
    auto first_loads = [&]() {
        const int c1 = a.ncb > 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) fetch_w(0, i);
        w3x = ld4_so(wsrc + 3 * 2048, (uint32_t)tid * 16u);
        r1x0 = ld4_so(a.x, goff[0]);
        if (S::NLD > 1) r1x1 = ld4_so(a.x, goff[S::NLD - 1]);
        fetch_w(c1, 3);
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) fetch_r(c1, i);
    };

    int64_t jt = bid >> 3;
    if (jt >= a.per_xcd || (bid & 7) * a.per_xcd + jt >= a.nwg) return;
    describe((bid & 7) * a.per_xcd + jt);
    advance();
    first_loads();
    for (;;) {
    DINV_STAMP(0);
    // the next tile of this workgroup: its first loads are issued from the last three blocks of this tile (peeled
    // below: no branch in the steady-state loop)
    const bool more = jt + a.slots < a.per_xcd && (bid & 7) * a.per_xcd + jt + a.slots < a.nwg;
    // (the last tile of a workgroup "prefetches" itself again: valid addresses, unused data, no second code path)
    describe((bid & 7) * a.per_xcd + jt + (more ? a.slots : 0));
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c2][j][r] = 0.f;
    {   // ---- tile prologue: stage the entry state, re-issue for blocks 1 and 2
        const int c1 = a.ncb > 1 ? 1 : 0, c2b = a.ncb > 2 ? 2 : a.ncb - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) stash_w(lds, i);
        st4(lds + woff + 3 * 128 * WP, w3x);
        stash_w(lds + S::BUF, 3);
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) stash_r(lds + S::BUF, i);
        pr[0] = r1x0;
        if (S::NLD > 1) pr[S::NLD - 1] = r1x1;
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) stash_r(lds, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) fetch_w(c1, i);
        fetch_w(c2b, 3);
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) fetch_r(c2b, i);
    }
    lds_barrier();
    tcalc(lds);
    DINV_UREAD(uA, lds, 0);
    DINV_VCALC(vA, tc[0]);

    DINV_STAMP(1);
    // One 8-channel block = 4 steps (m = channel 4h+m) of 8 MFMAs.  Everything else rides one step ahead:
    //   step 0: read U(m=1); V(m=1); write the staged weights of block cb+1 to LDS, re-issue the loads for cb+2
    //   step 1: read U(m=2), U(m=3); V(m=2); same for the staged input pixels;   barrier
    //   step 2: V(m=3); read the two patch rows of block cb+1
    //   step 3: row xr of B^T d for block cb+1; read U(cb+1, m=0); V(cb+1, m=0)
    // rel_tag: 0 = steady state; 3, 2, 1 = third-last ... last block of a tile whose successor's first loads are
    // issued from here (compile-time variants, peeled behind the loop)
    auto block = [&](int cb, auto parity_tag, auto rel_tag, float4 (&u0)[2], float4 (&u0n)[2]) {
        constexpr int PAR = decltype(parity_tag)::value;   // cb & 1, compile time: LDS addresses fold to immediates
        constexpr int REL = decltype(rel_tag)::value;
        const float* cur = lds + PAR * S::BUF;
        float* nxt = lds + (1 - PAR) * S::BUF;
        const int cb2 = cb + 2 < a.ncb ? cb + 2 : a.ncb - 1;   // re-issued loads past the end re-read the last block
        const int cb3 = cb + 3 < a.ncb ? cb + 3 : a.ncb - 1;
        float* curw = lds + PAR * S::BUF;
        float4 dA[2], dB[2];
        // Issue is in order and the older wave of a SIMD wins the MFMA pipe, so a wave often runs alone: every
        // filler is therefore placed by hand between MFMA pairs (sched_barrier pins it), LDS reads right after the
        // first pair of the step BEFORE the one that consumes them (>= 6 MFMAs = 384 cycles of cover).
#define DINV_SB() __builtin_amdgcn_sched_barrier(0)
        // ---- step 0 (m = 0)
        if (cb == 2 || cb == 3) DINV_STAMP(8 + (cb - 2) * 8 + 0);
        DINV_MFMA2(u0, vA, 0, 0); DINV_SB();
        DINV_UREAD(uB, cur, 1); DINV_SB();
        DINV_MFMA2(u0, vA, 0, 2); DINV_SB();
        DINV_VCALC(vB, tc[1]); DINV_SB();
        DINV_MFMA2(u0, vA, 1, 0); DINV_SB();
        if (REL != 1) { stash_w(nxt, 0); stash_w(nxt, 1); }
        if (REL == 2) {          // block ncb: the next tile's block 0
            pwt[0] = ld4_so(wsrcn, (uint32_t)tid * 16u); pwt[1] = ld4_so(wsrcn + 2048, (uint32_t)tid * 16u);
        } else if (REL != 1) {
            fetch_w(cb2, 0); fetch_w(cb2, 1);
        }
        DINV_SB();
        DINV_MFMA2(u0, vA, 1, 2); DINV_SB();
        if (cb == 2 || cb == 3) DINV_STAMP(8 + (cb - 2) * 8 + 1);
        // ---- step 1 (m = 1)
        DINV_MFMA2(uB, vB, 0, 0); DINV_SB();
        DINV_UREAD(uE, cur, 2);
        DINV_UREAD(uC, cur, 3); DINV_SB();
        DINV_MFMA2(uB, vB, 0, 2); DINV_SB();
        DINV_VCALC(vA, tc[2]); DINV_SB();
        DINV_MFMA2(uB, vB, 1, 0); DINV_SB();
        if (REL != 1) stash_w(nxt, 2);
        if (REL == 2) pwt[2] = ld4_so(wsrcn + 2 * 2048, (uint32_t)tid * 16u);
        else if (REL != 1) fetch_w(cb2, 2);
        DINV_SB();
        DINV_MFMA2(uB, vB, 1, 2); DINV_SB();
        if (cb == 2 || cb == 3) DINV_STAMP(8 + (cb - 2) * 8 + 2);
        lds_barrier();
        if (cb == 2 || cb == 3) DINV_STAMP(8 + (cb - 2) * 8 + 3);
        DINV_SB();
        // ---- step 2 (m = 2): everything the next block needs first is read right behind the barrier; the patch
        // rows come in two column pairs so that only 16 registers of raw pixels are live at a time
        DINV_MFMA2(uE, vA, 0, 0); DINV_SB();
        DINV_UREAD(u0n, nxt, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) { dA[j] = ld4(nxt + rA + DINV_COL(j)); dB[j] = ld4(nxt + rB + DINV_COL(j)); }
        DINV_SB();
        DINV_MFMA2(uE, vA, 0, 2); DINV_SB();
        DINV_VCALC(vB, tc[3]); DINV_SB();     // last use of this block's tc: it is overwritten in place below
        DINV_MFMA2(uE, vA, 1, 0); DINV_SB();
        // block cb+2's U part 3 + pixels: this buffer's block cb is dead behind the barrier
        if (REL == 0 || REL == 3) { stash_w(curw, 3); stash_r(curw, 0); }
        if (REL == 2) { w3x = pwt[3]; r1x0 = pr[0]; }      // they are the next tile's block 0 parts: park them
        if (REL == 0) { fetch_w(cb3, 3); fetch_r(cb3, 0); }
        if (REL == 3) { pwt[3] = ld4_so(wsrcn + 3 * 2048, (uint32_t)tid * 16u); pr[0] = ld4_so(a.x, goffn[0]); }
        if (REL == 2) { pwt[3] = ld4_so(wsrcn + 8192 + 3 * 2048, (uint32_t)tid * 16u); pr[0] = ld4_so(a.x + xcs, goffn[0]); }
        DINV_SB();
        DINV_MFMA2(uE, vA, 1, 2); DINV_SB();
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) tc[m][j] = fmaf(comp(dB[j], m), sigma, comp(dA[j], m));
#pragma unroll
        for (int j = 0; j < 2; ++j) { dA[j] = ld4(nxt + rA + DINV_COL(2 + j)); dB[j] = ld4(nxt + rB + DINV_COL(2 + j)); }
        DINV_SB();
        if (cb == 2 || cb == 3) DINV_STAMP(8 + (cb - 2) * 8 + 4);
        // ---- step 3 (m = 3)
        DINV_MFMA2(uC, vB, 0, 0); DINV_SB();
        DINV_MFMA2(uC, vB, 0, 2); DINV_SB();
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) tc[m][2 + j] = fmaf(comp(dB[j], m), sigma, comp(dA[j], m));
        DINV_VCALC(vA, tc[0]); DINV_SB();
        DINV_MFMA2(uC, vB, 1, 0); DINV_SB();
        if (S::NLD > 1) {
            if (REL == 0 || REL == 3) stash_r(curw, S::NLD - 1);
            if (REL == 2) r1x1 = pr[S::NLD - 1];
            if (REL == 0) fetch_r(cb3, S::NLD - 1);
            if (REL == 3) pr[S::NLD - 1] = ld4_so(a.x, goffn[S::NLD - 1]);
            if (REL == 2) pr[S::NLD - 1] = ld4_so(a.x + xcs, goffn[S::NLD - 1]);
        }
        DINV_SB();
        DINV_MFMA2(uC, vB, 1, 2); DINV_SB();
#undef DINV_SB
    };
    {
        // two blocks per trip so the register state returns to the same names (no copies on the back edge) and
        // the LDS buffer parity is a compile-time constant
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using R0 = std::integral_constant<int, 0>;
        // steady-state pairs, then the last four blocks peeled (block count even and >= 4: checked on the host)
#pragma unroll 1
        for (int cb = 0; cb + 4 < a.ncb; cb += 2) {
            block(cb, P0{}, R0{}, uA, uD);
            block(cb + 1, P1{}, R0{}, uD, uA);
        }
        block(a.ncb - 4, P0{}, R0{}, uA, uD);
        block(a.ncb - 3, P1{}, std::integral_constant<int, 3>{}, uD, uA);
        block(a.ncb - 2, P0{}, std::integral_constant<int, 2>{}, uA, uD);
        block(a.ncb - 1, P1{}, std::integral_constant<int, 1>{}, uD, uA);
    }
#undef DINV_MFMA2
#undef DINV_VCALC
#undef DINV_COL
#undef DINV_UREAD

    DINV_STAMP(2);
    // ---- epilogue: s = M A for this wave's Winograd row, exchange, then A^T s for channel sub-block rj = xr
    int te = tid;
    asm volatile("" : "+v"(te));      // same: (sub, ty, tx, h) are cheaper to recompute than to keep across the loop
    const int ep = wq * 32 + (te & 31), eh = (te >> 5) & 1;
    const int esub = ep / S::PT, ety = (ep % S::PT) / TW, etx = ep % TW;
    const uint32_t s = (uint32_t)pw * S::NSUB + esub;
    const uint32_t b = a.d_img.div(s);
    const uint32_t rem = s - b * (uint32_t)per_img;
    const uint32_t tyb = a.d_ntx.div(rem), txb = rem - tyb * a.ntx;
    const int oy = 2 * (tyb * TH + ety), ox = 2 * (txb * TW + etx);
    const bool live = s < (uint32_t)a.nsr && oy < a.g.h && ox < a.g.w;
    const bool okx = ox + 1 < a.g.w, oky = oy + 1 < a.g.h;
    // per-lane byte offsets inside one channel block; lanes/pixels outside the image get an out-of-range offset:
    // buffer loads then return 0 and buffer stores are dropped, so the epilogue has no branches
    const int64_t pix = a.g.sl + (int64_t)b * a.g.plane + (int64_t)(oy + 1) * a.g.wp + ox + 1;
    uint32_t loffs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool ok = live && !((q >> 1) && !oky) && !((q & 1) && !okx);
        loffs[q] = ok ? (uint32_t)((pix + (int64_t)(q >> 1) * a.g.wp + (q & 1)) * 32 + 16 * eh) : 0xffffffffu;
    }
    const int ect = ct;          // this tile's cout block is still needed below
    float4 rv[2][4];
    if (NRES) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int q = 0; q < 4; ++q)   // fetched now so that their latency hides behind the exchange
                rv[c2][q] = ld4_so(a.res + ((int64_t)ect * 8 + c2 * 4 + xr) * xcs, loffs[q]);
    }
    DINV_STAMP(3);
    lds_barrier();   // the staging buffers are dead: reuse LDS as E[wq][src row][rj][c2][dx][lane][4]
    DINV_STAMP(4);
    float* ex = lds + wq * (4 * 4 * 2 * 2 * 256) + (te & 63) * 4;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int rj = 0; rj < 4; ++rj) {
            float s0[4], s1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * rj + k;
                const float m0 = acc[c2][0][r], m1 = acc[c2][1][r], m2 = acc[c2][2][r], m3 = acc[c2][3][r];
                s0[k] = m0 + m1 + m2;
                s1[k] = m1 - m2 - m3;
            }
            float* e = ex + (((xr * 4 + rj) * 2 + c2) * 2) * 256;
            st4(e, make_float4(s0[0], s0[1], s0[2], s0[3]));
            st4(e + 256, make_float4(s1[0], s1[1], s1[2], s1[3]));
        }
    // ---- next tile: describe it and issue its first loads now, the accumulators are dead, so there are
    // registers for them; they fly during the exchange reads, the output stores and the tile turnover
    jt += a.slots;
    if (more) advance();
    lds_barrier();
    DINV_STAMP(5);
    float4 qv[2][2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[c2][dx][i] = ld4(ex + (((i * 4 + xr) * 2 + c2) * 2 + dx) * 256);
    // Last LDS access of the tile: barrier here, so the output stores (store-issue bound: ~4k cycles for the
    // workgroup's 64 KB) overlap with the next tile's staging instead of holding every wave at a barrier.
    lds_barrier();
    DINV_STAMP(6);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        float4 o[4];   // [dy*2+dx], channels 4h..4h+3 of block ect*8 + c2*4 + xr
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float4* q = qv[c2][dx];
            o[dx] = add4(add4(q[0], q[1]), q[2]);
            o[2 + dx] = make_float4(q[1].x - q[2].x - q[3].x, q[1].y - q[2].y - q[3].y, q[1].z - q[2].z - q[3].z,
                                    q[1].w - q[2].w - q[3].w);
        }
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.y + ((int64_t)ect * 8 + c2 * 4 + xr) * xcs, 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 val = o[q];
            if (RELU) val = make_float4(fmaxf(val.x, 0.f), fmaxf(val.y, 0.f), fmaxf(val.z, 0.f), fmaxf(val.w, 0.f));
            if (NRES) val = add4(val, rv[c2][q]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yrsrc, loffs[q], 0, 0);
        }
    }
    DINV_STAMP(7);
    if (!more) break;
#ifdef DINV_WINO_TIMING
    ++tile_k;
#endif
    }   // tile loop
}

template <int TH, int TW, bool RELU, int NRES>
int launch_shape(WinoArgs a, int tiles_y, int tiles_x, hipStream_t st) {
    using S = Shape<TH, TW>;
    a.nty = (int32_t)ceil_div(tiles_y, TH);
    a.ntx = (int32_t)ceil_div(tiles_x, TW);
    a.nsr = (int64_t)a.g.batch * a.nty * a.ntx;
    a.nwg = ceil_div(a.nsr, S::NSUB) * a.nct;
    a.per_xcd = ceil_div(a.nwg, 8);
    DINV_REQUIRE(a.nsr + 64 < (1ll << 31) && a.nwg < (1ll << 31), "winograd conv: too many tiles for 32-bit indexing");
    a.d_img = make_fastdiv((uint32_t)(a.nty * a.ntx));
    a.d_ntx = make_fastdiv((uint32_t)a.ntx);
    a.d_nct = make_fastdiv((uint32_t)a.nct);
    const size_t shm = S::LDSF * sizeof(float);
    static std::atomic<uint64_t> configured{0};   // per instantiation: bit d = attribute set on device d
    auto kern = conv3x3_wino_kernel<TH, TW, RELU, NRES>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(3, "hipGetDevice failed");
    const uint64_t bit = 1ull << (dev & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return fail(3, "hipFuncSetAttribute(max dynamic LDS) failed");
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    const int cpx = cus_per_xcd(dev);
    a.slots = (int32_t)(a.per_xcd < cpx ? a.per_xcd : cpx);
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.slots * 8)), dim3(NTHR), shm, st, a);
    DINV_CHECK_LAUNCH();
    return 0;
}

template <bool RELU, int NRES>
int launch_any(const WinoArgs& a, hipStream_t st) {
    const int tyn = (a.g.h + 1) / 2, txn = (a.g.w + 1) / 2;
    // rectangle shape with the least padded-tile waste; ties go to the widest (fewest halo loads)
    const int shapes[3][2] = {{4, 16}, {8, 8}, {4, 4}};
    int best = 0;
    double bw = 1e30;
    for (int i = 0; i < 3; ++i) {
        const double w = (double)ceil_div(tyn, shapes[i][0]) * shapes[i][0] * ceil_div(txn, shapes[i][1]) * shapes[i][1];
        if (w < bw * 0.999) { bw = w; best = i; }
    }
    switch (best) {
        case 0: return launch_shape<4, 16, RELU, NRES>(a, tyn, txn, st);
        case 1: return launch_shape<8, 8, RELU, NRES>(a, tyn, txn, st);
        default: return launch_shape<4, 4, RELU, NRES>(a, tyn, txn, st);
    }
}

}  // namespace

#ifdef DINV_WINO_TIMING
static long long* g_wino_dbg = nullptr;
extern "C" void dinv_debug_wino_timing(long long* p) { g_wino_dbg = p; }
#endif

extern "C" int dinv_conv3x3_winograd(const dinv_act_geom* g, const float* x, const float* w_wino, int32_t cin,
                                     int32_t cout, float* y, const float* res1, int32_t relu, dinv_stream_t stream) {
    if (check_geom(g)) return 1;
    DINV_REQUIRE(x && w_wino && y, "null pointer");
    DINV_REQUIRE(cin >= 32 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0,
                 "winograd conv needs cin %% 16 == 0, cin >= 32 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    DINV_REQUIRE(g->cs * 32 < (1ll << 32), "winograd conv: one channel block must stay below 4 GB (32-bit buffer offsets)");
    WinoArgs a{};
    a.g = make_geom(*g);
    a.x = x; a.w = w_wino; a.y = y; a.res = res1;
    a.ncb = cin / 8; a.nct = cout / 64;
#ifdef DINV_WINO_TIMING
    a.dbg = g_wino_dbg;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (relu) return launch_any<true, 0>(a, st);
    if (res1) return launch_any<false, 1>(a, st);
    return launch_any<false, 0>(a, st);
}
