// Wave-autonomous 2-D MultiCoilMRI pipelines (round 5) for image heights H = R * 64 (R in {4, 5, 8}) and widths in {256, 320, 512}.
//
// The column transform of length H is split  DFT_H = (64-point DFTs over u) o (twiddle W_H^(u q)) o (radix-R butterfly over j),
// row index h = u + 64 j, output index k = q + R k'  (decimation in frequency; the inverse runs the transposed chain).  The
// radix-R stage only combines R image rows element by element, so it rides along with the ROW transform of those R rows, and
// what is left for the column pass are 64-point transforms of tiles of 64 rows x 32 columns that fit one wave:
//
//   A   :  [x, S -> rows F_W of the R rows {u + 64 j} + radix-R  -> t]   [t -> 64-point columns -> y * mask (planar)]
//   A^T :  [y * mask -> 64-point columns^-1 -> t]   [t, S -> radix-R^-1 + rows F_W^-1 of R rows, coil sum in registers -> x]
//   A^T A: first pass of A, [64-point columns, M^2, 64-point columns^-1 in one tile], last pass of A^T
//
// Every kernel is built from independent WAVES (fft_wave.hpp): a wave owns its tile and its LDS region, exchanges data with
// wave-local ordering only (no s_barrier), and several waves per CU sit in different phases of load / transform / store.
// The intermediate `t` ([B, N, R, 64, W] interleaved: block q holds the rows u = 0..63 of residue q) is written once and
// read once per operator (A, A^T: 2.7x / 2.6x the algorithmic bytes; the workgroup-cooperative pipelines of mri.hip moved
// 3.0x / 4.5x / 23x and ran at 2-3.4 TB/s per pass).
// Centred transforms: input index i sits at position (i + N/2) mod N, output k at (k + N/2) mod N, on both axes.
#pragma once
#include <type_traits>
#include "fft_launch.hpp"

#ifndef DINV_MRIW_MINW
#define DINV_MRIW_MINW 2      // waves per SIMD the rows kernels are compiled for
#endif
#ifndef DINV_MRIW_WPB
#define DINV_MRIW_WPB 4       // waves per workgroup of the rows kernels (they share the LDS allocation and the twiddle table)
#endif

namespace dinv {
namespace mriw {

constexpr int CB = 64;      // rows of a column block (the 64-point transforms)
constexpr int CG = 32;      // columns of a column-pass tile (128 B of a planar row, 256 B of an interleaved one)

__device__ __forceinline__ void unpack_c4(const float4& a, const float4& b, float2 (&v)[4]) {
    v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w); v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
}

// =====================================================================================================================
// rows pass, decimation in frequency along the columns (first pass of A and of A^T):
//   YIN = false:  t[b, n, q, u, :] = ( radix-R over j of  F_W   ( S_n[h_j, :] x[b, h_j, :] ) ) W_H^(u q)          (in = x, S)
//   YIN = true :  t[b, n, q, u, :] = ( radix-R^-1 over j of F_W^-1( M[h_j, :] y[b, n, h_j, :] ) ) conj W_H^(u q)   (in = y planar)
// h_j = (u + 64 j + H/2) mod H.  One wave per (image (b, n), u): R rows of W complex = 12.8 KB of LDS at 320 x 320.
// =====================================================================================================================
template <class P, int R, bool YIN, int WPB>
__global__ __launch_bounds__(64 * WPB, DINV_MRIW_MINW) void rows_dif_kernel(const float* __restrict__ in, const float2* __restrict__ maps,
                                                               const float* __restrict__ mask, float2* __restrict__ t, int ncoil,
                                                               int aux_batch, int64_t ntiles, const void* table_w, const void* table_h,
                                                               float scale) {
    constexpr bool INV = YIN;
    using TF = TileFft<P, INV, true, R, 64>;
    constexpr int N = P::N, H = R * CB, R1 = P::R1, M1 = P::M1, T1 = M1 / 4;
    constexpr int NS1 = (R * T1 + 63) / 64;          // stage-1 item slots per lane
    constexpr int NSE = (N / 4 + 63) / 64;           // epilogue slots per lane (4 consecutive positions of all R rows each)
    static_assert(P::STAGES == 3 && N % 8 == 0 && (N / 2) % 4 == 0, "three-stage row plans, quads stay quads under the centre shift");
    constexpr size_t WREG = (TF::lds_floats2 + 1) / 2 * 2;      // float2 per wave (16-byte aligned regions)
    __shared__ __attribute__((aligned(16))) float2 buf_all[(size_t)WPB * WREG];
    __shared__ __attribute__((aligned(16))) float2 tw[TF::TAB];        // per-stage twiddles W^(u q) of the row plan
    TF::fill_twiddle_table(tw, reinterpret_cast<const float2*>(table_w), threadIdx.x, 64 * WPB);
    __syncthreads();
    const float2* twh = reinterpret_cast<const float2*>(table_h);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float2* buf = buf_all + (size_t)wv * WREG;
    constexpr int cw = N / 2, ch = H / 2;
    const int64_t vol = (int64_t)H * N;
    const int64_t stride = (int64_t)gridDim.x * WPB;
    // row of line `line` for block offset u, as an element offset
    auto row_off = [&](int line, int u) __attribute__((always_inline)) {
        int h = u + CB * line + ch;
        if (h >= H) h -= H;
        return (unsigned)(h * N);
    };
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wv; tile < ntiles; tile += stride) {
        const unsigned T = (unsigned)tile, p = T / CB, u = T - p * CB, b = p / (unsigned)ncoil, n = p - b * (unsigned)ncoil;
        // YIN: planar rows of coil n of y[b] and of the mask;  else: planar x[b] and interleaved S_n
        const float* pre = YIN ? in + ((int64_t)b * 2 * ncoil + n) * vol : in + (int64_t)b * 2 * vol;
        const int64_t pim = YIN ? (int64_t)ncoil * vol : vol;
        const float2* sp = (!YIN && maps) ? maps + ((int64_t)(aux_batch > 1 ? b : 0) * ncoil + n) * vol : nullptr;
        const float* mre = (YIN && mask) ? mask + (int64_t)(aux_batch > 1 ? b : 0) * 2 * vol : nullptr;
        const float2 wq = twh[u];          // column twiddle of the epilogue, requested here: its latency hides behind the stages
        wave_lds_sync();     // the previous tile's epilogue has read `buf`
        // ---------------- stage 1: x * S (or M * y) straight from global memory -> LDS (x and the maps of a slice are shared by
        // the waves that run its coils at the same time: L2 / Infinity-Cache hits)
#pragma unroll
        for (int s = 0; s < NS1; ++s) {
            const int w = lane + 64 * s, line = w / T1, u0 = (w - line * T1) * 4;
            if (w >= R * T1) continue;
            const unsigned ro = row_off(line, (int)u);
            float4 rre[R1], rim[R1], sa[R1], sb[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                int n0 = u0 + M1 * j + cw;
                if (n0 >= N) n0 -= N;
                const unsigned o = ro + (unsigned)n0;
                rre[j] = ld_f4(pre + o);
                rim[j] = ld_f4(pre + pim + o);
                if (sp) {
                    sa[j] = reinterpret_cast<const float4*>(sp + o)[0];
                    sb[j] = reinterpret_cast<const float4*>(sp + o)[1];
                }
                if (mre) {
                    sa[j] = ld_f4(mre + o);
                    sb[j] = ld_f4(mre + vol + o);
                }
            }
            float2 xv[R1][4];
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                if (mre) {      // float multiply exactly as mri.py:271
                    rre[j] = make_float4(sa[j].x * rre[j].x, sa[j].y * rre[j].y, sa[j].z * rre[j].z, sa[j].w * rre[j].w);
                    rim[j] = make_float4(sb[j].x * rim[j].x, sb[j].y * rim[j].y, sb[j].z * rim[j].z, sb[j].w * rim[j].w);
                }
                xv[j][0] = make_float2(rre[j].x, rim[j].x); xv[j][1] = make_float2(rre[j].y, rim[j].y);
                xv[j][2] = make_float2(rre[j].z, rim[j].z); xv[j][3] = make_float2(rre[j].w, rim[j].w);
                if (sp) {
                    xv[j][0] = cmul(make_float2(sa[j].x, sa[j].y), xv[j][0]); xv[j][1] = cmul(make_float2(sa[j].z, sa[j].w), xv[j][1]);
                    xv[j][2] = cmul(make_float2(sb[j].x, sb[j].y), xv[j][2]); xv[j][3] = cmul(make_float2(sb[j].z, sb[j].w), xv[j][3]);
                }
            }
            TF::v4_stage1_item_tab(buf, tw, line, u0, xv);
        }
        wave_lds_sync();
        TF::template v4_stage2_tab<WaveSync>(buf, tw, R, lane);
        TF::last_inplace(buf, R, scale, lane);
        wave_lds_sync();
        // ---------------- epilogue: radix-R butterfly over the R rows + column twiddle W_H^(u q), store block rows q * 64 + u
        float2* tout = t + (int64_t)p * vol + (int64_t)u * N;
#pragma unroll
        for (int s = 0; s < NSE; ++s) {
            const int pq = lane + 64 * s;
            if (pq >= N / 4) continue;
            float2 o[R][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int k = 4 * pq + e + cw;            // position 4 pq + e holds output k
                if (k >= N) k -= N;
                float2 v[R];
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = buf[TF::pos_of(j, k)];
                Bfly<R, INV>::run(v);
                apply_twiddle_powers<R, INV>(v, wq);
#pragma unroll
                for (int q = 0; q < R; ++q) o[q][e] = v[q];
            }
#pragma unroll
            for (int q = 0; q < R; ++q) st_c4(tout + (int64_t)q * CB * N + 4 * pq, o[q]);
        }
    }
}

// =====================================================================================================================
// 64-point column transforms on a tile of 64 rows x 32 columns held by ONE wave: lane (g, cq) = (lane / 8, lane % 8) owns the rows
// g + 8 m (m = 0..7) of 4 adjacent columns.  8-point transforms over m in registers, twiddle W_64^(g k2), an 8 x 8 transpose
// across the eight lane groups through the wave's LDS region, 8-point transforms over g: output index g' + 8 k1 in the lane
// group g' - the same layout as the input, so two transforms chain without moving data.
// =====================================================================================================================
constexpr int C64_PITCH = 8 * 32 + 2;     // float2 per outer index of the transpose (2 spare: the lane groups start 4 banks apart)
constexpr int C64_LDS = 8 * C64_PITCH;    // float2 per wave

template <bool INV>
__device__ __forceinline__ void col64(float2 (&v)[8][4], float2* buf, float2 w64g, int g, int cq) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float2 a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = v[m][e];
        Bfly<8, INV>::run(a);
        apply_twiddle_powers<8, INV>(a, w64g);      // a[k2] *= W_64^(g k2)
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m][e] = a[m];
    }
    wave_lds_sync();                                 // whoever read this region last is done
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) st_c4(buf + k2 * C64_PITCH + g * 32 + cq * 4, v[k2]);
    wave_lds_sync();
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) ld_c4(buf + g * C64_PITCH + n1 * 32 + cq * 4, v[n1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float2 a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = v[m][e];
        Bfly<8, INV>::run(a);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m][e] = a[m];
    }
}

// the same transform on HALF a tile (64 rows x 16 columns: 2 adjacent columns per lane) - for the pass that also keeps a coil sum
// and a prefetched tile in registers
constexpr int C64H_PITCH = 8 * 16 + 2;

template <bool INV>
__device__ __forceinline__ void col64_half(float2 (&v)[8][2], float2* buf, float2 w64g, int g, int cq) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float2 a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = v[m][e];
        Bfly<8, INV>::run(a);
        apply_twiddle_powers<8, INV>(a, w64g);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m][e] = a[m];
    }
    wave_lds_sync();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        *reinterpret_cast<float4*>(buf + k2 * C64H_PITCH + g * 16 + cq * 2) = make_float4(v[k2][0].x, v[k2][0].y, v[k2][1].x, v[k2][1].y);
    wave_lds_sync();
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const float4 r = *reinterpret_cast<const float4*>(buf + g * C64H_PITCH + n1 * 16 + cq * 2);
        v[n1][0] = make_float2(r.x, r.y);
        v[n1][1] = make_float2(r.z, r.w);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float2 a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = v[m][e];
        Bfly<8, INV>::run(a);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m][e] = a[m];
    }
}

// MODE 0: t -> y * mask (planar)           (last pass of A)
// MODE 1: y * mask (planar) -> t           (decimation-in-time first pass of A^T: not used by the pipelines below, kept for tests)
// MODE 2: t -> F, M^2, F^-1 -> t in place  (middle pass of A^T A)
template <int R, int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void cols64_kernel(float2* __restrict__ t, float* __restrict__ y, const float* __restrict__ mask,
                                                             int ncoil, int mask_batch, int W, int64_t ntiles, const void* table_h,
                                                             float scale) {
    constexpr int H = R * CB, ch = H / 2;
    __shared__ __attribute__((aligned(16))) float2 buf_all[(size_t)WPB * C64_LDS];
    const float2* twh = reinterpret_cast<const float2*>(table_h);
    const int lane = threadIdx.x & 63, g = lane >> 3, cq = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float2* buf = buf_all + (size_t)wv * C64_LDS;
    const float2 w64g = twh[R * g];                  // W_64^g = W_H^(R g)
    const int ncg = W / CG;
    const int64_t vol = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * WPB;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wv; tile < ntiles; tile += stride) {
        const unsigned T = (unsigned)tile, pq_ = T / (unsigned)ncg, cg = T - pq_ * (unsigned)ncg, p = pq_ / R, q = pq_ - p * R;
        const unsigned b = p / (unsigned)ncoil, n = p - b * (unsigned)ncoil;
        const unsigned col = cg * CG + cq * 4;
        float2* tt = t + (int64_t)p * vol + (int64_t)q * CB * W + col;                    // block q of image p
        float* yre = y ? y + ((int64_t)b * 2 * ncoil + n) * vol + col : nullptr;          // planar rows of coil n, slice b
        const float* mre = mask ? mask + (int64_t)(mask_batch > 1 ? b : 0) * 2 * vol + col : nullptr;
        // position (row of the image / of k-space) of index q + R k'
        auto hpos = [&](int kp) __attribute__((always_inline)) {
            int h = (int)q + R * kp + ch;
            if (h >= H) h -= H;
            return (unsigned)h * (unsigned)W;
        };
        float2 v[8][4];
        if (MODE == 1) {
            float4 re[8], im[8], mr[8], mi[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const unsigned o = hpos(g + 8 * m);
                re[m] = ld_f4(yre + o);
                im[m] = ld_f4(yre + (int64_t)ncoil * vol + o);
                if (mre) { mr[m] = ld_f4(mre + o); mi[m] = ld_f4(mre + vol + o); }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (mre) {      // float multiply exactly as mri.py:271
                    re[m] = make_float4(mr[m].x * re[m].x, mr[m].y * re[m].y, mr[m].z * re[m].z, mr[m].w * re[m].w);
                    im[m] = make_float4(mi[m].x * im[m].x, mi[m].y * im[m].y, mi[m].z * im[m].z, mi[m].w * im[m].w);
                }
                v[m][0] = make_float2(re[m].x, im[m].x); v[m][1] = make_float2(re[m].y, im[m].y);
                v[m][2] = make_float2(re[m].z, im[m].z); v[m][3] = make_float2(re[m].w, im[m].w);
            }
            col64<true>(v, buf, w64g, g, cq);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k1][e] = cscale(v[k1][e], scale);
                st_c4(tt + (unsigned)(g + 8 * k1) * (unsigned)W, v[k1]);
            }
        } else {
            float4 ra[8], rb[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4* src = reinterpret_cast<const float4*>(tt + (unsigned)(g + 8 * m) * (unsigned)W);
                ra[m] = src[0];
                rb[m] = src[1];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) unpack_c4(ra[m], rb[m], v[m]);
            col64<false>(v, buf, w64g, g, cq);
            if (MODE == 0) {
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) {
                    const unsigned o = hpos(g + 8 * k1);
                    float4 re = make_float4(v[k1][0].x * scale, v[k1][1].x * scale, v[k1][2].x * scale, v[k1][3].x * scale);
                    float4 im = make_float4(v[k1][0].y * scale, v[k1][1].y * scale, v[k1][2].y * scale, v[k1][3].y * scale);
                    if (mre) {  // float multiply exactly as mri.py:271; masked-out samples are written as exact zeros
                        const float4 mr = ld_f4(mre + o), mi = ld_f4(mre + vol + o);
                        re = make_float4(mr.x * re.x, mr.y * re.y, mr.z * re.z, mr.w * re.w);
                        im = make_float4(mi.x * im.x, mi.y * im.y, mi.z * im.z, mi.w * im.w);
                    }
                    st_f4(yre + o, re);
                    st_f4(yre + (int64_t)ncoil * vol + o, im);
                }
            } else {
                if (mre) {
#pragma unroll
                    for (int k1 = 0; k1 < 8; ++k1) {
                        const unsigned o = hpos(g + 8 * k1);
                        const float4 mr = ld_f4(mre + o), mi = ld_f4(mre + vol + o);
                        // y = m * (F t), then m * y on the way back: the two float multiplies of A followed by A^T
                        v[k1][0] = make_float2(mr.x * (mr.x * (v[k1][0].x * scale)), mi.x * (mi.x * (v[k1][0].y * scale)));
                        v[k1][1] = make_float2(mr.y * (mr.y * (v[k1][1].x * scale)), mi.y * (mi.y * (v[k1][1].y * scale)));
                        v[k1][2] = make_float2(mr.z * (mr.z * (v[k1][2].x * scale)), mi.z * (mi.z * (v[k1][2].y * scale)));
                        v[k1][3] = make_float2(mr.w * (mr.w * (v[k1][3].x * scale)), mi.w * (mi.w * (v[k1][3].y * scale)));
                    }
                } else {
#pragma unroll
                    for (int k1 = 0; k1 < 8; ++k1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[k1][e] = cscale(v[k1][e], scale);
                }
                col64<true>(v, buf, w64g, g, cq);
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[k1][e] = cscale(v[k1][e], scale);
                    st_c4(tt + (unsigned)(g + 8 * k1) * (unsigned)W, v[k1]);
                }
            }
        }
    }
}

// last pass of A^T:  x[b, h, :] = sum_n conj(S_n[h, :]) (64-point columns^-1 of t[b, n, q, :, :])[k'],  h = (q + R k' + H/2) mod H.
// One wave per (slice b, residue q, 32 columns) walks the coils: the coil sum stays in the lane that owns the output positions
// (fixed order n = 0 .. N-1: deterministic), the next coil's tile is requested as soon as the registers of the current one
// have been consumed.
template <int R, int WPB, bool MAPS>
__global__ __launch_bounds__(64 * WPB, 2) void cols64_combine_kernel(const float2* __restrict__ t, const float2* __restrict__ maps,
                                                                     float* __restrict__ x, int ncoil, int maps_batch, int W,
                                                                     int64_t ntiles, const void* table_h, float scale) {
    constexpr int H = R * CB, ch = H / 2;
    __shared__ __attribute__((aligned(16))) float2 buf_all[(size_t)WPB * C64_LDS];
    const float2* twh = reinterpret_cast<const float2*>(table_h);
    const int lane = threadIdx.x & 63, g = lane >> 3, cq = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float2* buf = buf_all + (size_t)wv * C64_LDS;
    const float2 w64g = twh[R * g];
    const int ncg = W / CG;
    const int64_t vol = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * WPB;
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wv; tile < ntiles; tile += stride) {
        const unsigned T = (unsigned)tile, bq = T / (unsigned)ncg, cg = T - bq * (unsigned)ncg, b = bq / R, q = bq - b * R;
        const unsigned col = cg * CG + cq * 4;
        unsigned hoff[8];       // image rows of this lane's outputs k' = g + 8 k1
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            int h = (int)q + R * (g + 8 * k1) + ch;
            if (h >= H) h -= H;
            hoff[k1] = (unsigned)h * (unsigned)W + col;
        }
        // The tile is worked in two halves of 16 columns (lane = 2 + 2 adjacent columns): the coil sum of the whole tile (64
        // registers), ONE half in flight through the transform (32) and the NEXT half on its way from memory (32) fit the 256
        // registers of a wave at two per SIMD - the whole-tile form (64 + 64 + 64) spilled 50 of them.
        float2 acc[8][4];
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[k1][e] = make_float2(0.f, 0.f);
        float4 rn[8], sn[8];        // the next half tile (requested before this half's transform) and the sensitivities of its
                                    // output positions (requested behind this half's accumulation, which frees the buffer)
        const unsigned toff = (unsigned)g * (unsigned)W + col;      // (uniform base + 32-bit lane offset: no 64-bit address registers)
        auto issue = [&](int n, int half) __attribute__((always_inline)) {
            const float2* tu = t + ((int64_t)b * ncoil + n) * vol + (int64_t)q * CB * W + 2 * half;
#pragma unroll
            for (int m = 0; m < 8; ++m)     // (byte offsets: base in SGPRs + a 32-bit lane offset is one addressing mode)
                rn[m] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tu + (int64_t)m * 8 * W) + toff * 8u);
        };
        auto issue_maps = [&](int n, int half) __attribute__((always_inline)) {
            if (MAPS) {
                const float2* sp = maps + ((int64_t)(maps_batch > 1 ? b : 0) * ncoil + n) * vol + 2 * half;
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) sn[k1] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sp) + hoff[k1] * 8u);
            }
        };
        auto half_step = [&](int n, auto half_) __attribute__((always_inline)) {
            constexpr int HF = decltype(half_)::value;
            float2 v[8][2];
#pragma unroll
            for (int m = 0; m < 8; ++m) { v[m][0] = make_float2(rn[m].x, rn[m].y); v[m][1] = make_float2(rn[m].z, rn[m].w); }
            __builtin_amdgcn_sched_barrier(0);                 // (a load has no reason to wait for its buffer's last reader unless told)
            // the next half flies during this half's transform (behind the last coil: that coil again, unused - no branch, so
            // that the buffers stay one set of registers)
            const int nn = n + 1 < ncoil ? n + 1 : n;
            if (HF == 0) issue(n, 1);
            else issue(nn, 0);
            col64_half<true>(v, buf, w64g, g, cq);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) {
                if (MAPS) {
                    acc[k1][2 * HF] = cadd(acc[k1][2 * HF], cmulc(v[k1][0], make_float2(sn[k1].x, sn[k1].y)));           // conj(S) * v
                    acc[k1][2 * HF + 1] = cadd(acc[k1][2 * HF + 1], cmulc(v[k1][1], make_float2(sn[k1].z, sn[k1].w)));
                } else {
                    acc[k1][2 * HF] = cadd(acc[k1][2 * HF], v[k1][0]);
                    acc[k1][2 * HF + 1] = cadd(acc[k1][2 * HF + 1], v[k1][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HF == 0) issue_maps(n, 1);
            else issue_maps(nn, 0);
        };
        issue(0, 0);
        issue_maps(0, 0);
        for (int n = 0; n < ncoil; ++n) {
            half_step(n, std::integral_constant<int, 0>{});
            half_step(n, std::integral_constant<int, 1>{});
        }
        float* xre = x + (int64_t)b * 2 * vol;
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            st_f4(xre + hoff[k1], make_float4(acc[k1][0].x * scale, acc[k1][1].x * scale, acc[k1][2].x * scale, acc[k1][3].x * scale));
            st_f4(xre + vol + hoff[k1], make_float4(acc[k1][0].y * scale, acc[k1][1].y * scale, acc[k1][2].y * scale, acc[k1][3].y * scale));
        }
    }
}

// =====================================================================================================================
// last pass of the adjoint:  x[b, h_j, :] = sum_n conj(S_n[h_j, :]) ( radix-R^-1 over q of conj(W_H^(u q)) F_W^-1 t[b, n, q, u, :] )
// (the radix-R stage acts element by element along the rows, so it commutes with the row transform and runs as its epilogue,
// exactly as in the forward pass).  One wave per (slice b, u) walks the coils; the coil sum stays in the registers of the lane
// that owns an output position (fixed order n = 0 .. N-1: deterministic).
// =====================================================================================================================
template <class P, int R, int WPB>
__global__ __launch_bounds__(64 * WPB, DINV_MRIW_MINW) void rows_combine_kernel(const float2* __restrict__ t, const float2* __restrict__ maps,
                                                                   float* __restrict__ x, int ncoil, int maps_batch, int64_t ntiles,
                                                                   const void* table_w, const void* table_h, float scale) {
    using TF = TileFft<P, true, true, R, 64>;
    constexpr int N = P::N, H = R * CB, R1 = P::R1, M1 = P::M1, T1 = M1 / 4;
    constexpr int NS1 = (R * T1 + 63) / 64;
    constexpr int NSE = (N / 4 + 63) / 64;
    static_assert(P::STAGES == 3 && N % 8 == 0 && (N / 2) % 4 == 0, "three-stage row plans, quads stay quads under the centre shift");
    constexpr size_t WREG = (TF::lds_floats2 + 1) / 2 * 2;
    __shared__ __attribute__((aligned(16))) float2 buf_all[(size_t)WPB * WREG];
    __shared__ __attribute__((aligned(16))) float2 tw[TF::TAB];        // per-stage twiddles W^(u q) of the row plan
    TF::fill_twiddle_table(tw, reinterpret_cast<const float2*>(table_w), threadIdx.x, 64 * WPB);
    __syncthreads();
    const float2* twh = reinterpret_cast<const float2*>(table_h);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float2* buf = buf_all + (size_t)wv * WREG;
    constexpr int cw = N / 2, ch = H / 2;
    const int64_t vol = (int64_t)H * N;
    const int64_t stride = (int64_t)gridDim.x * WPB;
    auto row_off = [&](int line, int u) __attribute__((always_inline)) {
        int h = u + CB * line + ch;
        if (h >= H) h -= H;
        return (unsigned)(h * N);
    };
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wv; tile < ntiles; tile += stride) {
        const unsigned T = (unsigned)tile, b = T / CB, u = T - b * CB;
        const float2 wq = twh[u];
        float2 acc[NSE][R][4];
#pragma unroll
        for (int s = 0; s < NSE; ++s)
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[s][j][e] = make_float2(0.f, 0.f);
        for (int n = 0; n < ncoil; ++n) {
            const float2* tin = t + ((int64_t)b * ncoil + n) * vol + (int64_t)u * N;      // row u of block q: + q * 64 * N
            const float2* sp = maps ? maps + ((int64_t)(maps_batch > 1 ? b : 0) * ncoil + n) * vol : nullptr;
            wave_lds_sync();     // the previous coil's epilogue has read `buf`
            // ---------------- stage 1: the R block rows (line = q) straight from global memory -> LDS
#pragma unroll
            for (int s = 0; s < NS1; ++s) {
                const int w = lane + 64 * s, line = w / T1, u0 = (w - line * T1) * 4;
                if (w >= R * T1) continue;
                float4 ra[R1], rb[R1];
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    int n0 = u0 + M1 * j + cw;
                    if (n0 >= N) n0 -= N;
                    const float4* src = reinterpret_cast<const float4*>(tin + (unsigned)(line * CB * N + n0));
                    ra[j] = src[0];
                    rb[j] = src[1];
                }
                float2 xv[R1][4];
#pragma unroll
                for (int j = 0; j < R1; ++j) unpack_c4(ra[j], rb[j], xv[j]);
                TF::v4_stage1_item_tab(buf, tw, line, u0, xv);
            }
            wave_lds_sync();
            TF::template v4_stage2_tab<WaveSync>(buf, tw, R, lane);
            TF::last_inplace(buf, R, scale, lane);
            wave_lds_sync();
            // ---------------- epilogue: conj column twiddle, radix-R^-1 over the block rows, conj(S) * . accumulated per position
#pragma unroll
            for (int s = 0; s < NSE; ++s) {
                const int pq = lane + 64 * s;
                if (pq >= N / 4) continue;
                float2 o[R][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = 4 * pq + e + cw;            // position 4 pq + e holds output k of the row transform
                    if (k >= N) k -= N;
                    float2 v[R];
#pragma unroll
                    for (int q = 0; q < R; ++q) v[q] = buf[TF::pos_of(q, k)];
                    apply_twiddle_powers<R, true>(v, wq);
                    Bfly<R, true>::run(v);
#pragma unroll
                    for (int j = 0; j < R; ++j) o[j][e] = v[j];
                }
                if (sp) {
                    float4 sa[R], sb[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const float4* src = reinterpret_cast<const float4*>(sp + row_off(j, (int)u) + 4 * pq);
                        sa[j] = src[0];
                        sb[j] = src[1];
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        float2 sv[4];
                        unpack_c4(sa[j], sb[j], sv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[s][j][e] = cadd(acc[s][j][e], cmulc(o[j][e], sv[e]));     // conj(S) * v
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < R; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[s][j][e] = cadd(acc[s][j][e], o[j][e]);
                }
            }
        }
        // ---------------- store the coil sum (planar)
        float* xre = x + (int64_t)b * 2 * vol;
#pragma unroll
        for (int s = 0; s < NSE; ++s) {
            const int pq = lane + 64 * s;
            if (pq >= N / 4) continue;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const unsigned o = row_off(j, (int)u) + 4 * pq;
                st_f4(xre + o, make_float4(acc[s][j][0].x, acc[s][j][1].x, acc[s][j][2].x, acc[s][j][3].x));
                st_f4(xre + vol + o, make_float4(acc[s][j][0].y, acc[s][j][1].y, acc[s][j][2].y, acc[s][j][3].y));
            }
        }
    }
}

// ------------------------------------------------------------------ host side
// op: 0 = A, 1 = A^T, 2 = A^T A.  The last passes of A^T / A^T A walk the coils inside one wave (B * R * W / 32 resp. B * 64 waves):
// below ~1000 of them the chip is not filled and the workgroup-cooperative pipelines of mri.hip are faster (measured at 4 slices
// of 8 coils 320 x 320: 0.072 / 0.089 ms against 0.045 / 0.070 ms)
inline bool wave2d_ok(const dinv_mri_desc* d, int op) {
    if (d->ndim != 2) return false;
    const int H = d->dims[0], W = d->dims[1];
    if (!(H == 256 || H == 320 || H == 512) || !(W == 256 || W == 320 || W == 512)) return false;
    const int64_t images = (int64_t)d->batch * d->coils;
    if (op != 0 && (int64_t)d->batch * CB < 1024 && d->reserved != 1) return false;
    return images * CB * (W / CG) * (H / CB) < (1ll << 31) && (int64_t)H * W < (1ll << 24);
}

inline int resident_waves_grid(int64_t wave_tiles, int wpb) {
    int cus = 256, dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
    const int64_t resident = std::max<int64_t>((int64_t)cus * 4 * DINV_MRIW_MINW / wpb, cus);       // DINV_MRIW_MINW waves per SIMD
    return (int)std::min<int64_t>(ceil_div(wave_tiles, wpb), resident);
}

#define DINV_MRIW_WIDTHS(X) X(256) X(320) X(512)

// waves per workgroup of a rows kernel: DINV_MRIW_WPB where that many tiles (+ the twiddle table) fit the 160 KB of LDS, else 4
template <class P, int R>
constexpr int rows_wpb() {
    using TF = TileFft<P, false, true, R, 64>;
    constexpr size_t per_wave = ((TF::lds_floats2 + 1) / 2 * 2) * sizeof(float2), tab = TF::TAB * sizeof(float2);
    return per_wave * DINV_MRIW_WPB + tab <= 160 * 1024 ? DINV_MRIW_WPB : 4;
}

template <int R, bool YIN>
int launch_rows_dif(int W, const float* in, const float2* maps, const float* mask, float2* t, int64_t images, int ncoil, int aux_batch,
                    const void* tw, const void* th, hipStream_t s) {
    const int64_t ntiles = images * CB;
    const float sc = 1.0f / sqrtf((float)W);
    switch (W) {
#define DINV_CASE(NN) case NN: { constexpr int WPB = rows_wpb<typename PlanForS<NN>::P, R>(); hipLaunchKernelGGL((rows_dif_kernel<typename PlanForS<NN>::P, R, YIN, WPB>), dim3((unsigned)resident_waves_grid(ntiles, WPB)), dim3(64 * WPB), 0, s, in, maps, mask, t, ncoil, aux_batch, ntiles, tw, th, sc); } break;
        DINV_MRIW_WIDTHS(DINV_CASE)
#undef DINV_CASE
        default: return fail(2, "mri wave pipeline: unsupported width %d", W);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

template <int R>
int launch_cols64_combine(int W, const float2* t, const float2* maps, float* x, int64_t batch, int ncoil, int maps_batch, const void* th,
                          hipStream_t s) {
    constexpr int WPB = 4;
    const int64_t ntiles = batch * R * (W / CG);
    const unsigned grid = (unsigned)resident_waves_grid(ntiles, WPB);
    const float hs = 1.0f / sqrtf((float)(R * CB));
    if (maps) hipLaunchKernelGGL((cols64_combine_kernel<R, WPB, true>), dim3(grid), dim3(64 * WPB), 0, s, t, maps, x, ncoil, maps_batch, W, ntiles, th, hs);
    else hipLaunchKernelGGL((cols64_combine_kernel<R, WPB, false>), dim3(grid), dim3(64 * WPB), 0, s, t, maps, x, ncoil, maps_batch, W, ntiles, th, hs);
    DINV_CHECK_LAUNCH();
    return 0;
}

template <int R>
int launch_rows_combine(int W, const float2* t, const float2* maps, float* x, int64_t batch, int ncoil, int maps_batch, const void* tw,
                        const void* th, hipStream_t s) {
    const int64_t ntiles = batch * CB;
    const float sc = 1.0f / sqrtf((float)W);
    switch (W) {
#define DINV_CASE(NN) case NN: { constexpr int WPB = rows_wpb<typename PlanForS<NN>::P, R>(); hipLaunchKernelGGL((rows_combine_kernel<typename PlanForS<NN>::P, R, WPB>), dim3((unsigned)resident_waves_grid(ntiles, WPB)), dim3(64 * WPB), 0, s, t, maps, x, ncoil, maps_batch, ntiles, tw, th, sc); } break;
        DINV_MRIW_WIDTHS(DINV_CASE)
#undef DINV_CASE
        default: return fail(2, "mri wave pipeline: unsupported width %d", W);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

template <int R, int MODE>
int launch_cols64(int W, float2* t, float* y, const float* mask, int64_t images, int ncoil, int mask_batch, const void* th, hipStream_t s) {
    constexpr int WPB = 4;
    const int64_t ntiles = images * R * (W / CG);
    const unsigned grid = (unsigned)resident_waves_grid(ntiles, WPB);
    const float hs = 1.0f / sqrtf((float)(R * CB));
    hipLaunchKernelGGL((cols64_kernel<R, MODE, WPB>), dim3(grid), dim3(64 * WPB), 0, s, t, y, mask, ncoil, mask_batch, W, ntiles, th, hs);
    DINV_CHECK_LAUNCH();
    return 0;
}

// op: 0 = A (x -> y), 1 = A^T (y -> x), 2 = A^T A (x -> out)
inline int run_wave2d(const dinv_mri_desc* d, int op, const float* in, const float2* maps, const float* mask, float* out, float2* t,
                      hipStream_t s) {
    const int H = d->dims[0], W = d->dims[1];
    const int64_t images = (int64_t)d->batch * d->coils;
    const void *th = d->table[0], *tw = d->table[1];
    int e = 0;
#define DINV_RUN(RR)                                                                                                              \
    do {                                                                                                                          \
        if (op == 0) {                                                                                                            \
            if ((e = (launch_rows_dif<RR, false>(W, in, maps, nullptr, t, images, d->coils, d->maps_batch, tw, th, s)))) return e; \
            return launch_cols64<RR, 0>(W, t, out, mask, images, d->coils, d->mask_batch, th, s);                                 \
        } else if (op == 1) {                                                                                                     \
            if ((e = (launch_rows_dif<RR, true>(W, in, nullptr, mask, t, images, d->coils, d->mask_batch, tw, th, s)))) return e;  \
            return launch_cols64_combine<RR>(W, t, maps, out, d->batch, d->coils, d->maps_batch, th, s);                          \
        } else {                                                                                                                  \
            if ((e = (launch_rows_dif<RR, false>(W, in, maps, nullptr, t, images, d->coils, d->maps_batch, tw, th, s)))) return e; \
            if ((e = launch_cols64<RR, 2>(W, t, nullptr, mask, images, d->coils, d->mask_batch, th, s))) return e;                \
            return launch_rows_combine<RR>(W, t, maps, out, d->batch, d->coils, d->maps_batch, tw, th, s);                        \
        }                                                                                                                         \
    } while (0)
    switch (H / CB) {
        case 4: DINV_RUN(4); break;
        case 5: DINV_RUN(5); break;
        case 8: DINV_RUN(8); break;
        default: break;
    }
#undef DINV_RUN
    return fail(2, "mri wave pipeline: unsupported height %d", H);
}

}  // namespace mriw
}  // namespace dinv
