// DRUNet 3x3 convolution, Winograd F(2x2, 3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Same operator as conv3x3_kernel in drunet.hip (nn.Conv2d 3x3 s1 p1 no bias inside the ResBlocks, reference
// deepinv/models/drunet.py:403-434), same "padded pixel rows, channels blocked by 8" activation layout, but the
// MFMA work is cut 2.25x: every 2x2 output tile needs 16 multiplies per (ci, co) instead of 36.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// * U = G g G^T is precomputed on the host (fp64 -> fp32), packed [cout/64][cin/8][ci 8][co 64][xi 16].
// * Workgroup = 4 waves = 64 couts x 64 tile positions; wave (wc, wq) owns 32 couts x 32 positions x all 16
//   Winograd points xi: 16 MFMA accumulators (256 AGPRs), one wave per SIMD, one workgroup per CU.
// * Per 8-channel block the raw input region (zero border included, so no edge cases) and the 32 KB slice of U
//   are staged in LDS (double buffered, one barrier per block, next block prefetched into registers during
//   the MFMAs).  Each lane builds V = B^T d B for its own position and its 4 channels (4h..4h+3, h = lane>>5
//   = MFMA k index) from 16 ds_read_b128 and 128 adds, and feeds 64 MFMAs with it.
// * The 64 positions of a workgroup are NSUB rectangles of TH x TW tiles enumerated over (batch, tile rows,
//   tile cols), so the 40x40 and 80x80 levels fill the MFMA columns as well as 320x320 does.
// * Epilogue: per-lane A^T M A, fused ReLU / residual add, float4 stores of interior pixels only.
#include "drunet_common.hpp"

using namespace dinv;
using namespace dinv_drunet;

namespace {

constexpr int WP = 20;                 // LDS pitch of one (ci, co) row of U: 16 xi + 4 pad (conflict-free b128)
constexpr int WLDS = 8 * 64 * WP;      // floats of U per channel block in LDS
constexpr int RP = 12;                 // LDS pitch of one staged pixel: 8 channels + 4 pad

struct WinoArgs {
    Geom g;
    const float* x;
    const float* w;
    float* y;
    const float* res;
    int32_t ncb, nct, nty, ntx;
    int64_t nsr, nwg, per_xcd;
};

template <int TH, int TW>
struct Shape {
    static constexpr int PT = TH * TW;             // tile positions per rectangle
    static constexpr int NSUB = 64 / PT;           // rectangles per workgroup
    static constexpr int RH = 2 * TH + 2;          // staged rows per rectangle
    static constexpr int RW2 = TW + 1;             // staged columns per parity
    static constexpr int RAWF = NSUB * RH * 2 * RW2 * RP;
    static constexpr int RAW4 = NSUB * RH * 2 * RW2 * 2;   // float4 loads to stage one block
    static constexpr int NLD = (RAW4 + 255) / 256;
    static constexpr int BUF = WLDS + RAWF;        // floats per LDS buffer
    static_assert(PT <= 64 && 64 % PT == 0, "rectangle must divide the 64 positions");
};

template <int TH, int TW, bool RELU, int NRES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv3x3_wino_kernel(WinoArgs a) {
    using S = Shape<TH, TW>;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wc = wave & 1, wq = wave >> 1;

    // consecutive block ids land on different XCDs: give each XCD a contiguous range of the logical order so
    // the cout tiles of one position tile (and neighbouring position tiles) share an L2
    const int64_t bid = blockIdx.x;
    const int64_t logical = (bid & 7) * a.per_xcd + (bid >> 3);
    if (logical >= a.nwg) return;
    const int ct = (int)(logical % a.nct);
    const int64_t pw = logical / a.nct;
    const int64_t per_img = (int64_t)a.nty * a.ntx;

    // ---- staging descriptors (independent of the channel block)
    int64_t goff[S::NLD];
    int loff[S::NLD];
    bool gok[S::NLD];
#pragma unroll
    for (int i = 0; i < S::NLD; ++i) {
        const int e = tid + 256 * i;
        const int half = e & 1, px = e >> 1;
        const int c = px % (2 * S::RW2);
        const int r = (px / (2 * S::RW2)) % S::RH;
        const int sub = px / (2 * S::RW2 * S::RH);
        const int64_t s = pw * S::NSUB + sub;
        const int64_t b = s / per_img;
        const int rem = (int)(s - b * per_img);
        const int tyb = rem / a.ntx, txb = rem - tyb * a.ntx;
        const int gr = 2 * tyb * TH + r, gc = 2 * txb * TW + c;
        gok[i] = e < S::RAW4 && s < a.nsr && gr < a.g.hp && gc < a.g.wp;
        goff[i] = (a.g.sl + b * a.g.plane + (int64_t)gr * a.g.wp + gc) * 8 + half * 4;
        loff[i] = WLDS + (((sub * S::RH + r) * 2 + (c & 1)) * S::RW2 + (c >> 1)) * RP + half * 4;
    }
    const float* wsrc = a.w + (int64_t)ct * a.ncb * 8192 + tid * 4;
    const int64_t xcs = a.g.cs * 8;

    float4 pr[S::NLD], pwt[8];
    auto fetch = [&](int cb) {
        const float* xb = a.x + cb * xcs;
#pragma unroll
        for (int i = 0; i < S::NLD; ++i) pr[i] = gok[i] ? ld4(xb + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wb = wsrc + (int64_t)cb * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) pwt[i] = ld4(wb + i * 1024);
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < S::NLD; ++i)
            if (S::RAW4 % 256 == 0 || i + 1 < S::NLD || tid + 256 * i < S::RAW4) st4(buf + loff[i], pr[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;          // float4 index: (ci*64 + co)*4 + q
            st4(buf + (f >> 2) * WP + (f & 3) * 4, pwt[i]);
        }
    };

    // ---- this lane's operand addresses
    const int p = wq * 32 + l31;
    const int sub = p / S::PT, ty = (p % S::PT) / TW, tx = p % TW;
    const int rbase = WLDS + ((sub * S::RH + 2 * ty) * 2 * S::RW2 + tx) * RP + 4 * h;
    const int abase = ((4 * h) * 64 + wc * 32 + l31) * WP;

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    fetch(0);
    stash(lds);
    __syncthreads();

    for (int cb = 0; cb < a.ncb; ++cb) {
        const float* buf = lds + (cb & 1) * S::BUF;
        const bool more = cb + 1 < a.ncb;
        if (more) fetch(cb + 1);

        float4 d[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[i * 4 + j] = ld4(buf + rbase + ((i * 2 + (j & 1)) * S::RW2 + (j >> 1)) * RP);
        // B^T d along rows, for the 4 channels at once
        float4 t[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
            t[j] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            t[4 + j] = add4(d1, d2);
            t[8 + j] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            t[12 + j] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float4 u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = ld4(buf + abase + m * 64 * WP + q * 4);
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t0 = comp(t[4 * i], m), t1 = comp(t[4 * i + 1], m), t2 = comp(t[4 * i + 2], m),
                            t3 = comp(t[4 * i + 3], m);
                v[4 * i] = t0 - t2;
                v[4 * i + 1] = t1 + t2;
                v[4 * i + 2] = t2 - t1;
                v[4 * i + 3] = t1 - t3;
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(u[xi >> 2], xi & 3), v[xi], acc[xi], 0, 0, 0);
        }
        if (more) stash(lds + ((cb + 1) & 1) * S::BUF);
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A per (co, position); lane holds co = 8*rj + 4*h + (0..3) for rj = 0..3
    const int64_t s = pw * S::NSUB + sub;
    if (s >= a.nsr) return;
    const int64_t b = s / per_img;
    const int rem = (int)(s - b * per_img);
    const int tyb = rem / a.ntx, txb = rem - tyb * a.ntx;
    const int oy = 2 * (tyb * TH + ty), ox = 2 * (txb * TW + tx);
    if (oy >= a.g.h || ox >= a.g.w) return;
    const bool okx = ox + 1 < a.g.w, oky = oy + 1 < a.g.h;
    const int64_t pix = a.g.sl + b * a.g.plane + (int64_t)(oy + 1) * a.g.wp + ox + 1;
#pragma unroll
    for (int rj = 0; rj < 4; ++rj) {
        float o[4][4];  // [output pixel dy*2+dx][channel]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * rj + k;
            float sv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[4 * i][r], m1 = acc[4 * i + 1][r], m2 = acc[4 * i + 2][r], m3 = acc[4 * i + 3][r];
                sv[i][0] = m0 + m1 + m2;
                sv[i][1] = m1 - m2 - m3;
            }
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                o[dx][k] = sv[0][dx] + sv[1][dx] + sv[2][dx];
                o[2 + dx][k] = sv[1][dx] - sv[2][dx] - sv[3][dx];
            }
        }
        const int64_t cbo = (int64_t)ct * 8 + wc * 4 + rj;
        const int64_t base = (cbo * a.g.cs + pix) * 8 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dy = q >> 1, dx = q & 1;
            if ((dy && !oky) || (dx && !okx)) continue;
            const int64_t off = base + ((int64_t)dy * a.g.wp + dx) * 8;
            float4 val = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
            if (RELU) val = make_float4(fmaxf(val.x, 0.f), fmaxf(val.y, 0.f), fmaxf(val.z, 0.f), fmaxf(val.w, 0.f));
            if (NRES) val = add4(val, ld4(a.res + off));
            st4(a.y + off, val);
        }
    }
}

template <int TH, int TW, bool RELU, int NRES>
int launch_shape(WinoArgs a, int tiles_y, int tiles_x, hipStream_t st) {
    using S = Shape<TH, TW>;
    a.nty = (int32_t)ceil_div(tiles_y, TH);
    a.ntx = (int32_t)ceil_div(tiles_x, TW);
    a.nsr = (int64_t)a.g.batch * a.nty * a.ntx;
    a.nwg = ceil_div(a.nsr, S::NSUB) * a.nct;
    a.per_xcd = ceil_div(a.nwg, 8);
    const size_t shm = 2 * S::BUF * sizeof(float);
    static bool once = false;  // per instantiation
    auto kern = conv3x3_wino_kernel<TH, TW, RELU, NRES>;
    if (!once) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return fail(3, "hipFuncSetAttribute(max dynamic LDS) failed");
        once = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.per_xcd * 8)), dim3(256), shm, st, a);
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

extern "C" int dinv_conv3x3_winograd(const dinv_act_geom* g, const float* x, const float* w_wino, int32_t cin,
                                     int32_t cout, float* y, const float* res1, int32_t relu, dinv_stream_t stream) {
    if (check_geom(g)) return 1;
    DINV_REQUIRE(x && w_wino && y, "null pointer");
    DINV_REQUIRE(cin >= 8 && cin % 8 == 0 && cout >= 64 && cout % 64 == 0,
                 "winograd conv needs cin %% 8 == 0 and cout %% 64 == 0 (got %d,%d)", cin, cout);
    DINV_REQUIRE(!(relu && res1), "relu and residual are not combined in DRUNet");
    WinoArgs a{};
    a.g = make_geom(*g);
    a.x = x; a.w = w_wino; a.y = y; a.res = res1;
    a.ncb = cin / 8; a.nct = cout / 64;
    hipStream_t st = (hipStream_t)stream;
    if (relu) return launch_any<true, 0>(a, st);
    if (res1) return launch_any<false, 1>(a, st);
    return launch_any<false, 0>(a, st);
}
