// Kernel templates + launch helpers for one-axis FFT passes over HBM-resident tensors.
//
//   rows pass : tensor viewed [nlines, N], N contiguous.  A 256-thread workgroup owns `lpb`
//               consecutive lines; 64 lanes run along N so every global access is a
//               contiguous 256/512-byte wave transaction.
//   cols pass : tensor viewed [P, N, Q], Q contiguous.  A workgroup owns an N x tq strip of
//               one p; tq lanes run along Q (tq*8 B = 128 B segments for tq = 16).
//
// The `Io` functor (passed by value) fuses the pointwise work of the calling operator into
// the load/store phase (coil-map multiply, mask multiply, planar<->interleaved layout
// change), so each pass is exactly one read and one write of the tensor.
#pragma once
#include "fft_core.hpp"
#include <cstdlib>
#include <type_traits>

#include "fft_static.hpp"
#include "fft_wave.hpp"

namespace dinv {

// ------------------------------------------------------------------ 16-byte vector access helpers
__device__ __forceinline__ void ld_c4(const float2* p, float2 (&v)[4]) {  // 4 interleaved complex = 2 x float4
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w); v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
}
__device__ __forceinline__ void st_c4(float2* p, const float2 (&v)[4]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
}
__device__ __forceinline__ float4 ld_f4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st_f4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <class T, class = void> struct io_has_vec4 : std::false_type {};
template <class T> struct io_has_vec4<T, std::void_t<decltype(T::has_vec4)>> : std::bool_constant<T::has_vec4> {};

// ------------------------------------------------------------------ plain complex IO
struct C2CIo {
    const float2* in;
    float2* out;
    struct RowCtx { int64_t base; };
    struct ColCtx { int64_t base; int64_t q; };
    int64_t n_, q_;  // filled by the launcher
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const { return RowCtx{line * n_}; }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const { return in[c.base + n]; }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { out[c.base + k] = v; }
    static constexpr bool has_vec4 = true;
    __device__ __forceinline__ void load4(const RowCtx& c, int n0, float2 (&v)[4]) const { ld_c4(in + c.base + n0, v); }
    __device__ __forceinline__ void store4(const RowCtx& c, int k0, const float2 (&v)[4]) const { st_c4(out + c.base + k0, v); }
    // wave-autonomous rows pass (fft_wave.hpp): a tile's base pointers are wave-uniform (scalar registers), lanes add a
    // 32-bit element offset; loads are left in flight as raw registers and finished at their first use
    struct Raw4 { float4 a, b; };
    struct TileCtx { const float2* in; float2* out; };
    __host__ bool wave_rows_ok(int, int64_t) const { return true; }     // rows are independent: any tile of consecutive rows
    __device__ __forceinline__ TileCtx tile_ctx(int64_t line0) const { return TileCtx{in + line0 * n_, out + line0 * n_}; }
    __device__ __forceinline__ void load4_raw(const TileCtx& c, unsigned off, Raw4& r) const {
        r.a = reinterpret_cast<const float4*>(c.in + off)[0];
        r.b = reinterpret_cast<const float4*>(c.in + off)[1];
    }
    struct Mask4 {};
    __device__ __forceinline__ bool has_mask() const { return false; }
    __device__ __forceinline__ void load_mask4(const TileCtx&, unsigned, Mask4&) const {}
    __device__ __forceinline__ void unpack4(const Raw4& r, const Mask4&, float2 (&v)[4]) const {
        v[0] = make_float2(r.a.x, r.a.y); v[1] = make_float2(r.a.z, r.a.w); v[2] = make_float2(r.b.x, r.b.y); v[3] = make_float2(r.b.z, r.b.w);
    }
    __device__ __forceinline__ void store4(const TileCtx& c, unsigned off, const float2 (&v)[4]) const { st_c4(c.out + off, v); }
    __device__ __forceinline__ ColCtx col_ctx(int64_t p, int64_t q) const { return ColCtx{p * n_ * q_ + q, q_}; }
    __device__ __forceinline__ float2 load(const ColCtx& c, int k) const { return in[c.base + (int64_t)k * c.q]; }
    __device__ __forceinline__ void store(const ColCtx& c, int k, float2 v) const { out[c.base + (int64_t)k * c.q] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// ------------------------------------------------------------------ kernels
template <class Io, bool INV>
__global__ __launch_bounds__(256) void fft_rows_kernel(Io io, int64_t nlines, int lpb, dinv_fft_plan plan,
                                                       const void* table, int centered, float scale) {
    DINV_DYN_LDS(unsigned char, smem);
    const int N = plan.n;
    const int LS = (N % 2 == 0) ? N + 1 : N;
    const int tid = threadIdx.x;
    LdsCarve L = carve_lds(smem, N, lpb, LS, plan.generic != 0);
    load_tables(L.tw, L.perm, table, N, tid, 256);
    __syncthreads();
    const int64_t line0 = (int64_t)blockIdx.x * lpb;
    const int lines = (int)min((int64_t)lpb, nlines - line0);
    const int c = centered ? N / 2 : 0;
    const int lane = tid & 63, wv = tid >> 6;
    for (int l = wv; l < lines; l += 4) {
        const typename Io::RowCtx ctx = io.row_ctx(line0 + l);
        float2* dst = L.buf + l * LS;
        for (int n = lane; n < N; n += 64) {
            int np = n - c;
            if (np < 0) np += N;
            dst[L.perm[np]] = io.load(ctx, n);
        }
    }
    const float2* res = tile_fft<INV>(plan, L.buf, L.alt, L.tw, lines, LS, tid, 256);
    for (int l = wv; l < lines; l += 4) {
        const typename Io::RowCtx ctx = io.row_ctx(line0 + l);
        const float2* src = res + l * LS;
        for (int k = lane; k < N; k += 64) {
            int kp = k - c;
            if (kp < 0) kp += N;
            io.store(ctx, k, cscale(src[kp], scale));
        }
    }
}

template <class Io, bool INV>
__global__ __launch_bounds__(256) void fft_cols_kernel(Io io, int64_t Q, int tq, int64_t qtiles,
                                                       dinv_fft_plan plan, const void* table, int centered,
                                                       float scale) {
    DINV_DYN_LDS(unsigned char, smem);
    const int N = plan.n;
    const int LS = (N % 2 == 0) ? N + 1 : N;
    const int tid = threadIdx.x;
    LdsCarve L = carve_lds(smem, N, tq, LS, plan.generic != 0);
    load_tables(L.tw, L.perm, table, N, tid, 256);
    __syncthreads();
    const int64_t p = blockIdx.x / qtiles;
    const int64_t q0 = (blockIdx.x - p * qtiles) * tq;
    const int cols = (int)min((int64_t)tq, Q - q0);
    const int c = centered ? N / 2 : 0;
    const int tx = tid % tq, ty = tid / tq, rpp = 256 / tq;
    typename Io::ColCtx ctx = io.col_ctx(p, q0 + (tx < cols ? tx : 0));
    if (tx < cols) {
        float2* dst = L.buf + tx * LS;
        for (int k = ty; k < N; k += rpp) {
            int kp = k - c;
            if (kp < 0) kp += N;
            dst[L.perm[kp]] = io.load(ctx, k);
        }
    }
    const float2* res = tile_fft<INV>(plan, L.buf, L.alt, L.tw, cols, LS, tid, 256);
    if (tx < cols) {
        const float2* src = res + tx * LS;
        for (int k = ty; k < N; k += rpp) {
            int kp = k - c;
            if (kp < 0) kp += N;
            io.store(ctx, k, cscale(src[kp], scale));
        }
    }
}

// ------------------------------------------------------------------ tile sizing
inline int rows_lines_per_block(const dinv_fft_plan& p) {
    const int LS = fft_line_stride(p.n);
    const size_t per_line = (size_t)LS * 8 * (p.generic ? 2 : 1);
    int lpb = (int)(24576 / per_line);
    if (lpb < 1) lpb = 1;
    if (lpb > 64) lpb = 64;
    return lpb;
}

inline int cols_tile_width(const dinv_fft_plan& p, int64_t Q) {
    const int LS = fft_line_stride(p.n);
    const size_t per_line = (size_t)LS * 8 * (p.generic ? 2 : 1);
    int tq = 64;
    while (tq > 1 && (size_t)tq * per_line > 49152) tq >>= 1;
    if (tq > 16 && (size_t)tq * per_line > 40960) tq = 16;
    while (tq > 1 && tq / 2 >= Q) tq >>= 1;
    return tq;
}

template <class K>
inline int set_lds_limit(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(100 + (int)e, "hipFuncSetAttribute(lds=%zu): %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

// ------------------------------------------------------------------ static-plan dispatch
constexpr int kMaxGrid = 256 * 8;
#ifndef DINV_WAVE_WPB
#define DINV_WAVE_WPB 4      // waves per workgroup of the wave-autonomous rows pass (they only share the LDS allocation)
#endif
#ifndef DINV_WAVE_PREFETCH
#define DINV_WAVE_PREFETCH 1  // issue the next tile's loads behind stage 1 of the current one
#endif
#ifndef DINV_WAVE_MINW
#define DINV_WAVE_MINW 2     // waves per SIMD it is compiled for (register budget 256): 8 tiles of 10 KB in flight per CU
#endif  // grid-stride over tiles: enough workgroups to fill 256 CUs several times

template <int N> struct RowsL { static constexpr int value = N >= 512 ? 8 : (N >= 128 ? 16 : 32); };
template <int N> struct ColsL { static constexpr int value = N == 16 ? 256 : 16; };

template <class T, class = void> struct io_has_raw4 : std::false_type {};
template <class T> struct io_has_raw4<T, std::void_t<typename T::Raw4>> : std::true_type {};
template <class T, class = void> struct io_planar_store : std::false_type {};
template <class T> struct io_planar_store<T, std::void_t<decltype(T::planar_store)>> : std::bool_constant<T::planar_store> {};

template <int N, class Io, int L>
inline int launch_rows_static_L(Io io, int64_t nlines, const void* table, int inverse, int centered, float scale,
                                hipStream_t s) {
    using P = std::conditional_t<io_planar_store<Io>::value, typename PlanForS<N>::P, typename PlanFor<N>::P>;
    const int64_t ntiles = ceil_div(nlines, L);
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, kMaxGrid);
    if constexpr (io_has_vec4<Io>::value && P::STAGES >= 2 && P::M1 % 4 == 0 && (P::N / (P::STAGES == 3 ? P::R3 : P::R2)) % 4 == 0) {
        if (centered == 0 || (P::N / 2) % 4 == 0) {
#ifndef DINV_NO_WAVE_ROWS
            if constexpr (io_has_raw4<Io>::value && N >= 256) {
                // wave-autonomous pass (fft_wave.hpp): one wave per tile of LW rows, no workgroup barrier, persistent waves
                constexpr int LW = WaveRowsL<P>::value, WPB = DINV_WAVE_WPB, MINW = DINV_WAVE_MINW;
                constexpr bool PF = DINV_WAVE_PREFETCH != 0;
                if (io.wave_rows_ok(LW, nlines)) {
                    const int64_t wtiles = ceil_div(nlines, LW);
                    int cus = 256;
                    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
                    const int64_t resident = (int64_t)cus * (4 * MINW / WPB);          // workgroups the chip holds at MINW waves per SIMD
                    const unsigned wgrid = (unsigned)std::min<int64_t>(ceil_div(wtiles, WPB), resident);
                    if (inverse)
                        hipLaunchKernelGGL((fft_rows_wave_kernel<P, Io, true, LW, WPB, MINW, PF>), dim3(wgrid), dim3(64 * WPB), 0, s, io,
                                           nlines, wtiles, table, centered, scale);
                    else
                        hipLaunchKernelGGL((fft_rows_wave_kernel<P, Io, false, LW, WPB, MINW, PF>), dim3(wgrid), dim3(64 * WPB), 0, s, io,
                                           nlines, wtiles, table, centered, scale);
                    DINV_CHECK_LAUNCH();
                    return 0;
                }
            }
#endif
            if (inverse)
                hipLaunchKernelGGL((fft_rows_static_v4_kernel<P, Io, true, L>), dim3(grid), dim3(256), 0, s, io, nlines,
                                   ntiles, table, centered, scale);
            else
                hipLaunchKernelGGL((fft_rows_static_v4_kernel<P, Io, false, L>), dim3(grid), dim3(256), 0, s, io, nlines,
                                   ntiles, table, centered, scale);
            DINV_CHECK_LAUNCH();
            return 0;
        }
    }
    if (inverse)
        hipLaunchKernelGGL((fft_rows_static_kernel<P, Io, true, L>), dim3(grid), dim3(256), 0, s, io, nlines, ntiles,
                           table, centered, scale);
    else
        hipLaunchKernelGGL((fft_rows_static_kernel<P, Io, false, L>), dim3(grid), dim3(256), 0, s, io, nlines, ntiles,
                           table, centered, scale);
    DINV_CHECK_LAUNCH();
    return 0;
}

template <int N, class Io>
inline int launch_rows_static(Io io, int64_t nlines, const void* table, int inverse, int centered, float scale,
                              hipStream_t s) {
    return launch_rows_static_L<N, Io, RowsL<N>::value>(io, nlines, table, inverse, centered, scale, s);
}

template <int N, class Io>
inline int launch_cols_static(Io io, int64_t P_, int64_t Q, const void* table, int inverse, int centered, float scale,
                              hipStream_t s, int group = 1) {
    using P = typename PlanFor<N>::P;
    constexpr int L = ColsL<N>::value;
    const int64_t qtiles = ceil_div(Q, L);
    const int64_t ntiles = P_ * qtiles;
    if (group > 1 && P_ % group != 0) group = 1;
    const int64_t padded = group > 1 ? ceil_div(ntiles, (int64_t)8 * group) * 8 * group : ntiles;
    const unsigned grid = (unsigned)std::min<int64_t>(padded, kMaxGrid);
    if (inverse)
        hipLaunchKernelGGL((fft_cols_static_kernel<P, Io, true, L>), dim3(grid), dim3(256), 0, s, io, Q, qtiles, ntiles,
                           table, centered, scale, group);
    else
        hipLaunchKernelGGL((fft_cols_static_kernel<P, Io, false, L>), dim3(grid), dim3(256), 0, s, io, Q, qtiles, ntiles,
                           table, centered, scale, group);
    DINV_CHECK_LAUNCH();
    return 0;
}

#define DINV_STATIC_SIZES(X) X(64) X(128) X(256) X(320) X(512)

template <class Io>
inline int launch_rows(Io io, int64_t nlines, const dinv_fft_plan& plan, const void* table, int inverse,
                       int centered, float scale, hipStream_t s) {
    if (nlines == 0) return 0;
    io.set_geometry(plan.n, 1);
    switch (plan.n) {
#define DINV_CASE(NN) case NN: return launch_rows_static<NN, Io>(io, nlines, table, inverse, centered, scale, s);
        DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
        default: break;
    }
    const int lpb = rows_lines_per_block(plan);
    const size_t lds = fft_lds_bytes(plan, lpb);
    DINV_REQUIRE(lds <= kMaxLdsBytes, "fft length %d does not fit the 160 KiB LDS tile (%zu B)", plan.n, lds);
    const int64_t blocks = ceil_div(nlines, lpb);
    DINV_REQUIRE(blocks < (1ll << 31), "too many fft lines (%lld)", (long long)nlines);
    if (inverse) {
        if (int e = set_lds_limit(fft_rows_kernel<Io, true>, lds)) return e;
        hipLaunchKernelGGL((fft_rows_kernel<Io, true>), dim3((unsigned)blocks), dim3(256), lds, s, io, nlines, lpb,
                           plan, table, centered, scale);
    } else {
        if (int e = set_lds_limit(fft_rows_kernel<Io, false>, lds)) return e;
        hipLaunchKernelGGL((fft_rows_kernel<Io, false>), dim3((unsigned)blocks), dim3(256), lds, s, io, nlines,
                           lpb, plan, table, centered, scale);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

template <class Io>
inline int launch_cols(Io io, int64_t P, int64_t Q, const dinv_fft_plan& plan, const void* table, int inverse,
                       int centered, float scale, hipStream_t s, int group = 1) {
    if (P == 0 || Q == 0) return 0;
    io.set_geometry(plan.n, Q);
    switch (plan.n) {
#define DINV_CASE(NN) case NN: return launch_cols_static<NN, Io>(io, P, Q, table, inverse, centered, scale, s, group);
        DINV_CASE(16) DINV_CASE(32) DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
        default: break;
    }
    const int tq = cols_tile_width(plan, Q);
    const size_t lds = fft_lds_bytes(plan, tq);
    DINV_REQUIRE(lds <= kMaxLdsBytes, "fft length %d does not fit the 160 KiB LDS tile (%zu B)", plan.n, lds);
    const int64_t qtiles = ceil_div(Q, tq);
    const int64_t blocks = P * qtiles;
    DINV_REQUIRE(blocks < (1ll << 31), "too many fft tiles (%lld)", (long long)blocks);
    if (inverse) {
        if (int e = set_lds_limit(fft_cols_kernel<Io, true>, lds)) return e;
        hipLaunchKernelGGL((fft_cols_kernel<Io, true>), dim3((unsigned)blocks), dim3(256), lds, s, io, Q, tq,
                           qtiles, plan, table, centered, scale);
    } else {
        if (int e = set_lds_limit(fft_cols_kernel<Io, false>, lds)) return e;
        hipLaunchKernelGGL((fft_cols_kernel<Io, false>), dim3((unsigned)blocks), dim3(256), lds, s, io, Q, tq,
                           qtiles, plan, table, centered, scale);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

}  // namespace dinv
