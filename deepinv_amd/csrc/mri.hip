// MRI / MultiCoilMRI forward and adjoint:  y = M F (S_n x),  x = sum_n conj(S_n) F^H (M y_n).
//
// Reference semantics: deepinv/physics/mri.py:254-272 (A), :284-324 (A_adjoint),
// deepinv/utils/mixins.py:149-180 (planar<->complex, centred orthonormal FFT).
// The reference runs 3 ATen launches + 2 layout copies + 2 broadcast multiplies per call;
// here the coil multiply, the planar<->interleaved change, both fft shifts, the 1/sqrt(N)
// scaling, the mask multiply and the coil reduction are folded into the load/store phases
// of the FFT passes, so the tensor makes one HBM round trip per transformed axis.
//
//   forward 2-D : rows(W)  [x,S -> t]          ; cols(H) [t -> y*mask]
//   forward 3-D : rows(W)  [x,S -> t]          ; cols(H) [t -> t] ; cols(D) [t -> y*mask]
//   adjoint 2-D : cols(H)  [y*mask -> t]       ; rows(W)+coil-combine [t,S -> x]
//   adjoint 3-D : cols(D)  [y*mask -> t]       ; cols(H) [t -> t] ; rows(W)+combine
// t is a complex64 scratch of B*N*vol elements supplied by the caller.
#include "fft_core.hpp"
#include "fft_launch.hpp"

#include "mri_wave.hpp"

using namespace dinv;

namespace {

// ---- rows pass of the forward op: planar x (optionally times coil map) -> interleaved t
struct RowsCoilLoadIo {
    const float* x;      // [B,2,R,W]
    const float2* maps;  // [mb,N,R,W] or null
    float2* t;           // [B,N,R,W]
    int32_t ncoil, maps_batch;
    int64_t R, W;
    int64_t n_, q_;
    struct RowCtx { int64_t xre, xim, s, o; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const {
        const int64_t r = line % R;
        const int64_t bn = line / R;
        const int64_t n = bn % ncoil;
        const int64_t b = bn / ncoil;
        RowCtx c;
        c.xre = ((b * 2) * R + r) * W;
        c.xim = c.xre + R * W;
        c.s = (((maps_batch > 1 ? b : 0) * ncoil + n) * R + r) * W;
        c.o = line * W;
        return c;
    }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const {
        float2 v = make_float2(x[c.xre + n], x[c.xim + n]);
        if (maps) v = cmul(maps[c.s + n], v);
        return v;
    }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { t[c.o + k] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// ---- last cols pass of the forward op: interleaved t -> planar y times mask
struct ColsPlanarMaskStoreIo {
    const float2* t;    // [P=B*N, Na, Q]
    float* y;           // [B,2,N,Na,Q]
    const float* mask;  // [mb,2,Na,Q] or null
    int32_t ncoil, mask_batch;
    int64_t n_, q_;
    struct RowCtx {};
    struct ColCtx { int64_t tin, yre, yim, mre, mim; };
    __device__ __forceinline__ ColCtx col_ctx(int64_t p, int64_t q) const {
        const int64_t vol = n_ * q_;
        const int64_t b = p / ncoil, n = p % ncoil;
        ColCtx c;
        c.tin = p * vol + q;
        c.yre = ((b * 2) * ncoil + n) * vol + q;
        c.yim = c.yre + (int64_t)ncoil * vol;
        c.mre = ((mask_batch > 1 ? b : 0) * 2) * vol + q;
        c.mim = c.mre + vol;
        return c;
    }
    __device__ __forceinline__ float2 load(const ColCtx& c, int k) const { return t[c.tin + (int64_t)k * q_]; }
    __device__ __forceinline__ void store(const ColCtx& c, int k, float2 v) const {
        const int64_t o = (int64_t)k * q_;
        if (mask) {
            // multiply by the float mask exactly as mri.py:271 does (never skip the write:
            // y must be dense with exact zeros where mask == 0, test_physics.py:1052-1077)
            v.x = mask[c.mre + o] * v.x;
            v.y = mask[c.mim + o] * v.y;
        }
        y[c.yre + o] = v.x;
        y[c.yim + o] = v.y;
    }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// ---- first cols pass of the adjoint: planar y times mask -> interleaved t
struct ColsPlanarMaskLoadIo {
    const float* y;     // [B,2,N,Na,Q]
    const float* mask;  // [mb,2,Na,Q] or null
    float2* t;          // [P=B*N, Na, Q]
    int32_t ncoil, mask_batch;
    int64_t n_, q_;
    struct RowCtx {};
    struct ColCtx { int64_t tout, yre, yim, mre, mim; };
    __device__ __forceinline__ ColCtx col_ctx(int64_t p, int64_t q) const {
        const int64_t vol = n_ * q_;
        const int64_t b = p / ncoil, n = p % ncoil;
        ColCtx c;
        c.tout = p * vol + q;
        c.yre = ((b * 2) * ncoil + n) * vol + q;
        c.yim = c.yre + (int64_t)ncoil * vol;
        c.mre = ((mask_batch > 1 ? b : 0) * 2) * vol + q;
        c.mim = c.mre + vol;
        return c;
    }
    __device__ __forceinline__ float2 load(const ColCtx& c, int k) const {
        const int64_t o = (int64_t)k * q_;
        float2 v = make_float2(y[c.yre + o], y[c.yim + o]);
        if (mask) {
            v.x = mask[c.mre + o] * v.x;
            v.y = mask[c.mim + o] * v.y;
        }
        return v;
    }
    __device__ __forceinline__ void store(const ColCtx& c, int k, float2 v) const { t[c.tout + (int64_t)k * q_] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// ---- last pass of the adjoint: inverse rows FFT of every coil + sum_n conj(S_n) * .
// One workgroup owns `lpb` (b,r) lines and walks the coils; the coil sum lives in an LDS
// accumulator (each thread always touches the same elements, so no race, no atomics:
// the reduction order n = 0..N-1 is fixed -> deterministic).
__global__ __launch_bounds__(256) void mri_rows_combine_kernel(const float2* __restrict__ t,
                                                               const float2* __restrict__ maps,
                                                               float* __restrict__ x, int64_t nlines, int64_t R,
                                                               int ncoil, int maps_batch, int lpb,
                                                               dinv_fft_plan plan, const void* table,
                                                               int centered, float scale) {
    DINV_DYN_LDS(unsigned char, smem);
    const int N = plan.n;
    const int LS = (N % 2 == 0) ? N + 1 : N;
    const int tid = threadIdx.x;
    LdsCarve L = carve_lds(smem, N, lpb, LS, plan.generic != 0);
    float2* acc = (plan.generic ? L.alt : L.buf) + (size_t)lpb * LS;
    load_tables(L.tw, L.perm, table, N, tid, 256);
    const int64_t line0 = (int64_t)blockIdx.x * lpb;
    const int lines = (int)min((int64_t)lpb, nlines - line0);
    const int c = centered ? N / 2 : 0;
    const int lane = tid & 63, wv = tid >> 6;
    for (int l = wv; l < lines; l += 4)
        for (int k = lane; k < N; k += 64) acc[l * LS + k] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int coil = 0; coil < ncoil; ++coil) {
        for (int l = wv; l < lines; l += 4) {
            const int64_t line = line0 + l;
            const int64_t b = line / R, r = line - b * R;
            const float2* src = t + ((b * ncoil + coil) * R + r) * (int64_t)N;
            float2* dst = L.buf + l * LS;
            for (int n = lane; n < N; n += 64) {
                int np = n - c;
                if (np < 0) np += N;
                dst[L.perm[np]] = src[n];
            }
        }
        const float2* res = tile_fft<true>(plan, L.buf, L.alt, L.tw, lines, LS, tid, 256);
        for (int l = wv; l < lines; l += 4) {
            const int64_t line = line0 + l;
            const int64_t b = line / R, r = line - b * R;
            const float2* s = maps ? maps + (((maps_batch > 1 ? b : 0) * ncoil + coil) * R + r) * (int64_t)N : nullptr;
            for (int k = lane; k < N; k += 64) {
                int kp = k - c;
                if (kp < 0) kp += N;
                float2 v = cscale(res[l * LS + kp], scale);
                if (s) v = cmulc(v, s[k]);  // conj(S) * v
                float2 a = acc[l * LS + k];
                acc[l * LS + k] = cadd(a, v);
            }
        }
        __syncthreads();  // res (LDS) is overwritten by the next coil's load phase
    }
    for (int l = wv; l < lines; l += 4) {
        const int64_t line = line0 + l;
        const int64_t b = line / R, r = line - b * R;
        float* xre = x + ((b * 2) * R + r) * (int64_t)N;
        float* xim = xre + R * (int64_t)N;
        for (int k = lane; k < N; k += 64) {
            float2 a = acc[l * LS + k];
            xre[k] = a.x;
            xim[k] = a.y;
        }
    }
}

// static-plan variant of the coil-combine pass: the coil sum is accumulated in registers by the threads that
// own the last-stage outputs (fixed (line,k) per thread across coils -> deterministic order n = 0..N-1)
struct CombineLine { int64_t t0, s0, x0; };  // offsets of coil 0 in t / maps, and of the output row

template <class P, int L>
__global__ __launch_bounds__(256) void mri_rows_combine_static_kernel(const float2* __restrict__ t,
                                                                      const float2* __restrict__ maps,
                                                                      float* __restrict__ x, int64_t nlines, int64_t R,
                                                                      int ncoil, int maps_batch, int64_t ntiles,
                                                                      const void* table, int centered, float scale) {
    using TF = TileFft<P, true, true, L>;
    constexpr int N = P::N;
    __shared__ __attribute__((aligned(16))) float2 buf[TF::lds_floats2];
    __shared__ CombineLine cl[L];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int c = centered ? N / 2 : 0;
    const int64_t coil_stride = R * (int64_t)N;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t line0 = tile * L;
        const int lines = (int)min((int64_t)L, nlines - line0);
        __syncthreads();
        if (tid < lines) {
            const int64_t line = line0 + tid;
            const int64_t b = line / R, r = line - b * R;
            CombineLine v;
            v.t0 = ((b * ncoil) * R + r) * (int64_t)N;
            v.s0 = (((maps_batch > 1 ? b : 0) * ncoil) * R + r) * (int64_t)N;
            v.x0 = ((b * 2) * R + r) * (int64_t)N;
            cl[tid] = v;
        }
        __syncthreads();
        float2 acc[TF::NSL][TF::RL];
#pragma unroll
        for (int a = 0; a < TF::NSL; ++a)
#pragma unroll
            for (int q = 0; q < TF::RL; ++q) acc[a][q] = make_float2(0.f, 0.f);
        for (int coil = 0; coil < ncoil; ++coil) {
            const int64_t co = coil * coil_stride;
            TF::run(buf, tw, lines, c, scale, tid,
                    [&](int, int, int line, int n) { return t[cl[line].t0 + co + n]; },
                    [&](int slot, int line, int k, int q, float2 v) {
                        if (maps) v = cmulc(v, maps[cl[line].s0 + co + k]);  // conj(S) * v
                        acc[slot][q] = cadd(acc[slot][q], v);
                    });
            __syncthreads();  // buf is rewritten by the next coil's stage 1
        }
        // store: same item decomposition as the last stage of TileFft::run
        constexpr int Q2N = (P::STAGES == 3) ? P::R2 : 1;
#pragma unroll
        for (int slot = 0; slot < TF::NSL; ++slot) {
            const int w = tid + 256 * slot;
            const int line = w / TF::KL, i = w - line * TF::KL;
            if (w >= L * TF::KL || line >= lines) continue;
            const int q1 = i % P::R1, q2 = i / P::R1;
            float* xre = x + cl[line].x0;
            float* xim = xre + coil_stride;
#pragma unroll
            for (int q = 0; q < TF::RL; ++q) {
                int k = q1 + P::R1 * q2 + P::R1 * Q2N * q + c;
                if (k >= N) k -= N;
                xre[k] = acc[slot][q].x;
                xim[k] = acc[slot][q].y;
            }
        }
    }
}

template <int N>
int launch_combine_static(const float2* t, const float2* maps, float* x, int64_t nlines, int64_t R, int ncoil,
                          int maps_batch, const void* table, float scale, hipStream_t s) {
    using P = typename PlanFor<N>::P;
    constexpr int L = RowsL<N>::value;
    const int64_t ntiles = ceil_div(nlines, L);
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, kMaxGrid);
    hipLaunchKernelGGL((mri_rows_combine_static_kernel<P, L>), dim3(grid), dim3(256), 0, s, t, maps, x, nlines, R, ncoil,
                       maps_batch, ntiles, table, 1, scale);
    DINV_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================
// Static-plan pipeline (all transformed sizes in {16,32,64,128,256,320,512}): pass order chosen so that the big
// planar k-space tensor y is only ever touched along its contiguous rows, and x / coil maps are read once per
// workgroup instead of once per coil / batch element:
//   forward : cols(first axis) with coil loop [x cached in registers, S -> t] ; (cols(H) in place) ; rows(W) [t -> y*mask]
//   adjoint : rows(W) [y*mask -> t] ; (cols(H) in place) ; cols(first axis) + coil combine [t,S -> x]
// =====================================================================================================

// rows pass, forward: interleaved t -> planar y * mask   (lines = (b, n, r) rows of length W)
struct RowsPlanarMaskStoreIo {
    static constexpr bool planar_store = true;  // -> store-friendly plan (64-wide output runs)
    const float2* t;
    float* y;
    const float* mask;
    int32_t ncoil, mask_batch;
    int64_t R, W;  // rows per volume, row length
    int64_t n_, q_;
    struct RowCtx { int64_t tin, yre, yim, mre, mim; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const {
        const int64_t r = line % R, bn = line / R;
        const int64_t n = bn % ncoil, b = bn / ncoil;
        const int64_t vol = R * W;
        RowCtx c;
        c.tin = line * W;
        c.yre = ((b * 2) * ncoil + n) * vol + r * W;
        c.yim = c.yre + (int64_t)ncoil * vol;
        c.mre = ((mask_batch > 1 ? b : 0) * 2) * vol + r * W;
        c.mim = c.mre + vol;
        return c;
    }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const { return t[c.tin + n]; }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const {
        if (mask) {  // float multiply exactly as mri.py:271; masked-out samples are written as exact zeros
            v.x = mask[c.mre + k] * v.x;
            v.y = mask[c.mim + k] * v.y;
        }
        y[c.yre + k] = v.x;
        y[c.yim + k] = v.y;
    }
    static constexpr bool has_vec4 = true;
    __device__ __forceinline__ void load4(const RowCtx& c, int n0, float2 (&v)[4]) const { ld_c4(t + c.tin + n0, v); }
    __device__ __forceinline__ void store4(const RowCtx& c, int k0, const float2 (&v)[4]) const {
        float4 re = make_float4(v[0].x, v[1].x, v[2].x, v[3].x), im = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
        if (mask) {
            const float4 mr = ld_f4(mask + c.mre + k0), mi = ld_f4(mask + c.mim + k0);
            re = make_float4(mr.x * re.x, mr.y * re.y, mr.z * re.z, mr.w * re.w);
            im = make_float4(mi.x * im.x, mi.y * im.y, mi.z * im.z, mi.w * im.w);
        }
        st_f4(y + c.yre + k0, re);
        st_f4(y + c.yim + k0, im);
    }
    // wave-autonomous rows pass (fft_wave.hpp): a tile of LW consecutive rows must lie inside one (b, n) image; its base
    // pointers are wave-uniform, lanes add a 32-bit element offset
    __host__ bool wave_rows_ok(int lw, int64_t nlines) const { return R % lw == 0 && (uint64_t)R * (uint64_t)W < (1ull << 31) && nlines < (1ll << 31); }
    struct Raw4 { float4 a, b; };
    struct TileCtx { const float2* tin; float* yre; float* yim; const float* mre; const float* mim; };
    __device__ __forceinline__ TileCtx tile_ctx(int64_t line0) const {      // (32-bit divisions: wave_rows_ok bounds the sizes)
        const unsigned l0 = (unsigned)line0, Ru = (unsigned)R, bn = l0 / Ru, r = l0 - bn * Ru;
        const unsigned b = bn / (unsigned)ncoil, n = bn - b * (unsigned)ncoil;
        const int64_t vol = R * W;
        const int64_t yre = ((int64_t)(b * 2) * ncoil + n) * vol + (int64_t)r * W, mre = ((mask_batch > 1 ? (int64_t)b : 0) * 2) * vol + (int64_t)r * W;
        return TileCtx{t + line0 * W, y + yre, y + yre + (int64_t)ncoil * vol, mask ? mask + mre : nullptr, mask ? mask + mre + vol : nullptr};
    }
    __device__ __forceinline__ void load4_raw(const TileCtx& c, unsigned off, Raw4& r) const {
        r.a = reinterpret_cast<const float4*>(c.tin + off)[0];
        r.b = reinterpret_cast<const float4*>(c.tin + off)[1];
    }
    struct Mask4 {};
    __device__ __forceinline__ bool has_mask() const { return false; }      // (of the LOADED side: the mask is applied by store4)
    __device__ __forceinline__ void load_mask4(const TileCtx&, unsigned, Mask4&) const {}
    __device__ __forceinline__ void unpack4(const Raw4& r, const Mask4&, float2 (&v)[4]) const {
        v[0] = make_float2(r.a.x, r.a.y); v[1] = make_float2(r.a.z, r.a.w); v[2] = make_float2(r.b.x, r.b.y); v[3] = make_float2(r.b.z, r.b.w);
    }
    __device__ __forceinline__ void store4(const TileCtx& c, unsigned off, const float2 (&v)[4]) const {
        float4 re = make_float4(v[0].x, v[1].x, v[2].x, v[3].x), im = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
        if (mask) {  // float multiply exactly as mri.py:271; masked-out samples are written as exact zeros
            const float4 mr = ld_f4(c.mre + off), mi = ld_f4(c.mim + off);
            re = make_float4(mr.x * re.x, mr.y * re.y, mr.z * re.z, mr.w * re.w);
            im = make_float4(mi.x * im.x, mi.y * im.y, mi.z * im.z, mi.w * im.w);
        }
        st_f4(c.yre + off, re);
        st_f4(c.yim + off, im);
    }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// rows pass, adjoint: planar y * mask -> interleaved t
struct RowsPlanarMaskLoadIo {
    const float* y;
    const float* mask;
    float2* t;
    int32_t ncoil, mask_batch;
    int64_t R, W;
    int64_t n_, q_;
    struct RowCtx { int64_t tout, yre, yim, mre, mim; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const {
        const int64_t r = line % R, bn = line / R;
        const int64_t n = bn % ncoil, b = bn / ncoil;
        const int64_t vol = R * W;
        RowCtx c;
        c.tout = line * W;
        c.yre = ((b * 2) * ncoil + n) * vol + r * W;
        c.yim = c.yre + (int64_t)ncoil * vol;
        c.mre = ((mask_batch > 1 ? b : 0) * 2) * vol + r * W;
        c.mim = c.mre + vol;
        return c;
    }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const {
        float2 v = make_float2(y[c.yre + n], y[c.yim + n]);
        if (mask) {
            v.x = mask[c.mre + n] * v.x;
            v.y = mask[c.mim + n] * v.y;
        }
        return v;
    }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { t[c.tout + k] = v; }
    static constexpr bool has_vec4 = true;
    __device__ __forceinline__ void load4(const RowCtx& c, int n0, float2 (&v)[4]) const {
        float4 re = ld_f4(y + c.yre + n0), im = ld_f4(y + c.yim + n0);
        if (mask) {
            const float4 mr = ld_f4(mask + c.mre + n0), mi = ld_f4(mask + c.mim + n0);
            re = make_float4(mr.x * re.x, mr.y * re.y, mr.z * re.z, mr.w * re.w);
            im = make_float4(mi.x * im.x, mi.y * im.y, mi.z * im.z, mi.w * im.w);
        }
        v[0] = make_float2(re.x, im.x); v[1] = make_float2(re.y, im.y);
        v[2] = make_float2(re.z, im.z); v[3] = make_float2(re.w, im.w);
    }
    __device__ __forceinline__ void store4(const RowCtx& c, int k0, const float2 (&v)[4]) const { st_c4(t + c.tout + k0, v); }
    // wave-autonomous rows pass (fft_wave.hpp): the mask multiply happens when the registers are consumed
    __host__ bool wave_rows_ok(int lw, int64_t nlines) const { return R % lw == 0 && (uint64_t)R * (uint64_t)W < (1ull << 31) && nlines < (1ll << 31); }
    struct Raw4 { float4 re, im; };
    struct TileCtx { float2* tout; const float* yre; const float* yim; const float* mre; const float* mim; };
    __device__ __forceinline__ TileCtx tile_ctx(int64_t line0) const {      // (32-bit divisions: wave_rows_ok bounds the sizes)
        const unsigned l0 = (unsigned)line0, Ru = (unsigned)R, bn = l0 / Ru, r = l0 - bn * Ru;
        const unsigned b = bn / (unsigned)ncoil, n = bn - b * (unsigned)ncoil;
        const int64_t vol = R * W;
        const int64_t yre = ((int64_t)(b * 2) * ncoil + n) * vol + (int64_t)r * W, mre = ((mask_batch > 1 ? (int64_t)b : 0) * 2) * vol + (int64_t)r * W;
        return TileCtx{t + line0 * W, y + yre, y + yre + (int64_t)ncoil * vol, mask ? mask + mre : nullptr, mask ? mask + mre + vol : nullptr};
    }
    __device__ __forceinline__ void load4_raw(const TileCtx& c, unsigned off, Raw4& r) const {
        r.re = ld_f4(c.yre + off);
        r.im = ld_f4(c.yim + off);
    }
    // (the mask - small, L2-resident - is fetched when the samples are consumed: its latency is covered by the other waves
    // of the CU, and the 8 registers per element group it would occupy while in flight are what limits their number)
    struct Mask4 { float4 mr, mi; };
    __device__ __forceinline__ bool has_mask() const { return mask != nullptr; }
    __device__ __forceinline__ void load_mask4(const TileCtx& c, unsigned off, Mask4& m) const {
        m.mr = ld_f4(c.mre + off);
        m.mi = ld_f4(c.mim + off);
    }
    __device__ __forceinline__ void unpack4(const Raw4& r, const Mask4& m, float2 (&v)[4]) const {
        float4 re = r.re, im = r.im;
        if (mask) {
            re = make_float4(m.mr.x * re.x, m.mr.y * re.y, m.mr.z * re.z, m.mr.w * re.w);
            im = make_float4(m.mi.x * im.x, m.mi.y * im.y, m.mi.z * im.z, m.mi.w * im.w);
        }
        v[0] = make_float2(re.x, im.x); v[1] = make_float2(re.y, im.y);
        v[2] = make_float2(re.z, im.z); v[3] = make_float2(re.w, im.w);
    }
    __device__ __forceinline__ void store4(const TileCtx& c, unsigned off, const float2 (&v)[4]) const { st_c4(c.tout + off, v); }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// cols pass along the first volume axis, forward: planar x times coil map -> interleaved t (p = b*ncoil + n)
struct ColsCoilLoadIo {
    const float* x;
    const float2* maps;
    float2* t;
    int32_t ncoil, maps_batch;
    int64_t n_, q_;
    struct RowCtx {};
    struct ColCtx { int64_t xre, xim, s, o; };
    __device__ __forceinline__ ColCtx col_ctx(int64_t p, int64_t q) const {
        const int64_t vol = n_ * q_;
        const int64_t b = p / ncoil, n = p % ncoil;
        ColCtx c;
        c.xre = (b * 2) * vol + q;
        c.xim = c.xre + vol;
        c.s = ((maps_batch > 1 ? b : 0) * ncoil + n) * vol + q;
        c.o = p * vol + q;
        return c;
    }
    __device__ __forceinline__ float2 load(const ColCtx& c, int k) const {
        const int64_t o = (int64_t)k * q_;
        float2 v = make_float2(x[c.xre + o], x[c.xim + o]);
        if (maps) v = cmul(maps[c.s + o], v);
        return v;
    }
    __device__ __forceinline__ void store(const ColCtx& c, int k, float2 v) const { t[c.o + (int64_t)k * q_] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// ---- first pass of the forward operator, fused: t[b,n] = F_axis0( S[n] . x[b] ) on tiles of L columns.
// The planar x tile and the interleaved coil-map tile are fetched with 16-byte-per-lane loads (all of a thread's loads in
// flight before the first use), multiplied, and written to the LDS tile in natural order; the transform then runs in place
// (TileFft::run<LDS_IN>) and its last stage stores t.  Replaces the streaming expansion kernel + an in-place column pass:
// one write of t instead of a write, a read and a write (210 of the 630 MB that A moved per call at cfg2).
// The coils of one (slice, column tile) run on one XCD at the same time (blocks b and b + 8; observed placement, speed
// only), so x is read from HBM once and from that XCD's L2 seven times.
template <class P, int L, int NT>
__global__ __launch_bounds__(NT) void mri_cols_expand_fwd_kernel(const float* __restrict__ x, const float2* __restrict__ maps,
                                                                  float2* __restrict__ t, int ncoil, int maps_batch, int64_t Q,
                                                                  int64_t qtiles, int64_t nsets, const void* table, float scale) {
    using TF = TileFft<P, false, false, L, NT>;
    constexpr int N = P::N, QL = L / 4, NI = (N * QL + NT - 1) / NT;
    static_assert(L % 4 == 0 && P::STAGES > 1, "wide loads need 4 | L; single-stage lengths use the plain column pass");
    __shared__ __attribute__((aligned(16))) float2 buf[(size_t)N * L];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x, line = tid % L;
    const int c = N / 2;
    const int64_t vol = (int64_t)N * Q;
    const int64_t padded = ceil_div_dev(nsets, 8) * 8 * ncoil;
    for (int64_t T = blockIdx.x; T < padded; T += gridDim.x) {
        const int64_t chunk = T / (8 * ncoil), within = T - chunk * (8 * ncoil);
        const int64_t set = chunk * 8 + within % 8;
        const int n = (int)(within / 8);
        if (set >= nsets) continue;
        const int64_t b = set / qtiles, q0 = (set - b * qtiles) * L;
        const int cols = (int)min((int64_t)L, Q - q0);
        const float* xre = x + (b * 2) * vol + q0;
        const float2* sp = maps ? maps + ((maps_batch > 1 ? b : 0) * ncoil + n) * vol + q0 : nullptr;
        float4 xr[NI], xi[NI], sa[NI], sb[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int item = tid + NT * i, row = item / QL, quad = item - row * QL;
            const bool ok = item < N * QL && 4 * quad < cols;
            const int64_t o = ok ? (int64_t)row * Q + 4 * quad : 0;
            xr[i] = ld_f4(xre + o);
            xi[i] = ld_f4(xre + vol + o);
            if (sp) {
                sa[i] = reinterpret_cast<const float4*>(sp + o)[0];
                sb[i] = reinterpret_cast<const float4*>(sp + o)[1];
            }
        }
        __syncthreads();   // the previous tile's last stage has left the LDS tile
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int item = tid + NT * i, row = item / QL, quad = item - row * QL;
            if (item >= N * QL) continue;
            float2 v[4] = {make_float2(xr[i].x, xi[i].x), make_float2(xr[i].y, xi[i].y), make_float2(xr[i].z, xi[i].z),
                           make_float2(xr[i].w, xi[i].w)};
            if (sp) {
                v[0] = cmul(make_float2(sa[i].x, sa[i].y), v[0]);
                v[1] = cmul(make_float2(sa[i].z, sa[i].w), v[1]);
                v[2] = cmul(make_float2(sb[i].x, sb[i].y), v[2]);
                v[3] = cmul(make_float2(sb[i].z, sb[i].w), v[3]);
            }
            st_c4(buf + row * L + 4 * quad, v);
        }
        __syncthreads();
        float2* o = t + (b * ncoil + n) * vol + q0 + (line < cols ? line : 0);
        TF::template run<true>(buf, tw, cols, c, scale, tid,
                               [&](int, int, int, int nn) { return buf[nn * L + line]; },
                               [&](int, int, int k, int, float2 v) { o[(int64_t)k * Q] = v; });
    }
}

// ---- last pass of the adjoint, fused: x[b] = sum_n conj(S[n]) . F^H_axis0( t[b,n] ) on tiles of L columns.  One workgroup
// owns a (slice, column tile) and walks the coils; the stage-1 inputs AND the coil-map values of the NEXT coil are loaded
// into registers before the current coil is transformed, so the HBM stream never waits for a transform; the coil sum is
// accumulated in registers by the thread that owns an output position (fixed order n = 0..N-1: deterministic).
// Replaces an in-place column pass + the streaming combination kernel: one read of t instead of a read, a write and a read.
template <class P, int L, int NT>
__global__ __launch_bounds__(NT) void mri_cols_combine_inv_kernel(const float2* __restrict__ t,
                                                                  const float2* __restrict__ maps,
                                                                  float* __restrict__ x, int ncoil, int maps_batch,
                                                                  int64_t Q, int64_t qtiles, int64_t ntiles,
                                                                  const void* table, float scale) {
    using TF = TileFft<P, true, false, L, NT>;
    constexpr int N = P::N;
    __shared__ __attribute__((aligned(16))) float2 buf[P::STAGES > 1 ? TF::lds_floats2 : 1];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x, line = tid % L;
    const int c = N / 2;
    const int64_t vol = (int64_t)N * Q;
    constexpr int Q2N = (P::STAGES == 3) ? P::R2 : 1;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t b = tile / qtiles;
        const int64_t q0 = (tile - b * qtiles) * L;
        const int cols = (int)min((int64_t)L, Q - q0);
        const int64_t q = q0 + (line < cols ? line : 0);
        float2 acc[TF::NSL][TF::RL];
#pragma unroll
        for (int a = 0; a < TF::NSL; ++a)
#pragma unroll
            for (int r = 0; r < TF::RL; ++r) acc[a][r] = make_float2(0.f, 0.f);
        // output position k of (slot, r): the last-stage item decomposition of TileFft::run
        auto out_k = [&](int slot, int r) {
            const int w = tid + NT * slot, i = w / L;
            int k = (P::STAGES == 1 ? r : (i % P::R1) + P::R1 * (i / P::R1) + P::R1 * Q2N * r) + c;
            return k >= N ? k - N : k;
        };
        // two register sets (A, B) alternate: the loads of coil n + 1 are in flight while coil n is transformed
        float2 va[TF::NS1][P::R1], vb[TF::NS1][P::R1];
#define DINV_FETCH(coil, V)                                                                                           \
        do {                                                                                                          \
            const float2* in_ = t + ((b * ncoil + (coil)) * vol + q);                                                 \
            TF::load_inputs(V, cols, c, tid, [&](int, int, int, int n) { return in_[(int64_t)n * Q]; });              \
        } while (0)
#define DINV_XFORM(coil, V)                                                                                           \
        do {                                                                                                          \
            const float2* s_ = maps ? maps + (((maps_batch > 1 ? b : 0) * ncoil + (coil)) * vol + q) : nullptr;       \
            if (P::STAGES > 1) __syncthreads(); /* the previous transform has left the LDS tile */                    \
            TF::run_regs(buf, tw, cols, c, scale, tid, V, [&](int slot, int, int k, int r, float2 v) {                \
                if (s_) v = cmulc(v, s_[(int64_t)k * Q]); /* conj(S) * v */                                           \
                acc[slot][r] = cadd(acc[slot][r], v);                                                                 \
            });                                                                                                       \
        } while (0)
        DINV_FETCH(0, va);
        for (int coil = 0; coil < ncoil; coil += 2) {
            if (coil + 1 < ncoil) DINV_FETCH(coil + 1, vb);
            DINV_XFORM(coil, va);
            if (coil + 1 < ncoil) {
                if (coil + 2 < ncoil) DINV_FETCH(coil + 2, va);
                DINV_XFORM(coil + 1, vb);
            }
        }
#undef DINV_FETCH
#undef DINV_XFORM
        float* xre = x + (b * 2) * vol + q;
        float* xim = xre + vol;
#pragma unroll
        for (int slot = 0; slot < TF::NSL; ++slot) {
            if (tid + NT * slot >= L * TF::KL || line >= cols) continue;
#pragma unroll
            for (int r = 0; r < TF::RL; ++r) {
                const int k = out_k(slot, r);
                xre[(int64_t)k * Q] = acc[slot][r].x;
                xim[(int64_t)k * Q] = acc[slot][r].y;
            }
        }
    }
}

// coil combination as a streaming pass: x[b] = sum_n conj(S[n]) * t[b,n], 4 pixels (2 x 16 B of t) per thread,
// the coil maps of a pixel quad stay in registers across a chunk of batch elements
constexpr int CB = 4;  // batch elements per thread
template <int NC>
__global__ __launch_bounds__(256) void mri_coil_combine_kernel(const float2* __restrict__ t,
                                                               const float2* __restrict__ maps,
                                                               float* __restrict__ x, int64_t vol, int batch,
                                                               int ncoil, int maps_batch) {
    const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pix = quad * 4;
    if (pix >= vol) return;
    const int b0 = blockIdx.y * CB;
    float2 sv[NC][4];
    const bool shared_maps = maps_batch <= 1;
    const bool cached = maps && shared_maps && ncoil <= NC;  // maps of this pixel quad live in registers
    if (cached) {
#pragma unroll
        for (int n = 0; n < NC; ++n)
            if (n < ncoil) ld_c4(maps + (int64_t)n * vol + pix, sv[n]);
    }
    for (int bb = 0; bb < CB; ++bb) {
        const int b = b0 + bb;
        if (b >= batch) break;
        float2 acc[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
        for (int n0 = 0; n0 < ncoil; n0 += NC) {
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                if (n0 + n >= ncoil) break;
                float2 tv[4];
                ld_c4(t + ((int64_t)b * ncoil + n0 + n) * vol + pix, tv);
                if (maps) {
                    if (!cached) ld_c4(maps + ((int64_t)(shared_maps ? 0 : b) * ncoil + n0 + n) * vol + pix, sv[n]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = cadd(acc[e], cmulc(tv[e], sv[n][e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = cadd(acc[e], tv[e]);
                }
            }
        }
        float* xre = x + ((int64_t)b * 2) * vol + pix;
        st_f4(xre, make_float4(acc[0].x, acc[1].x, acc[2].x, acc[3].x));
        st_f4(xre + vol, make_float4(acc[0].y, acc[1].y, acc[2].y, acc[3].y));
    }
}

template <int N>
int launch_cols_expand_fwd(const float* x, const float2* maps, float2* t, int64_t B, int ncoil, int maps_batch, int64_t Q,
                           const void* table, float scale, hipStream_t s) {
    using P = typename PlanFor<N>::P;
    if constexpr (P::STAGES > 1) {
        constexpr int L = ColsL<N>::value;
        const int64_t qtiles = ceil_div(Q, L), nsets = B * qtiles;
        const int64_t padded = ceil_div(nsets, 8) * 8 * ncoil;
        const unsigned grid = (unsigned)std::min<int64_t>(padded, 4 * kMaxGrid);
        // 256 threads: two workgroups per CU at N = 320 (210 VGPRs), so one tile's loads overlap another's transform
        // (measured at cfg2: 111 us; 512 threads / 148 VGPRs = one workgroup per CU: 135 us; 512 threads capped at 128
        // VGPRs (spills): 171 us)
        constexpr int NT = 256;
        hipLaunchKernelGGL((mri_cols_expand_fwd_kernel<P, L, NT>), dim3(grid), dim3(NT), 0, s, x, maps, t, ncoil, maps_batch, Q,
                           qtiles, nsets, table, scale);
        DINV_CHECK_LAUNCH();
        return 0;
    } else {
        return fail(2, "no fused expand pass for length %d", N);
    }
}

template <int N>
int launch_cols_combine_inv(const float2* t, const float2* maps, float* x, int64_t B, int ncoil, int maps_batch,
                            int64_t Q, const void* table, float scale, hipStream_t s) {
    using P = typename PlanFor<N>::P;
    constexpr int L = ColsL<N>::value;
    const int64_t qtiles = ceil_div(Q, L), ntiles = B * qtiles;
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, 4 * kMaxGrid);
    constexpr int NT = N >= 256 ? 512 : 256;
    hipLaunchKernelGGL((mri_cols_combine_inv_kernel<P, L, NT>), dim3(grid), dim3(NT), 0, s, t, maps, x, ncoil, maps_batch, Q,
                       qtiles, ntiles, table, scale);
    DINV_CHECK_LAUNCH();
    return 0;
}

// ---- normal operator, middle pass along the LAST (contiguous) axis: t <- F^H_W( M^2 . F_W t ) on tiles of L rows; the forward
// transform emits masked k-space rows into a second LDS tile in natural order, the inverse transform reads them from
// there.  This is the variant the normal operator uses.
// Measured at cfg2 (B = 32, 8 coils, 320 x 320; us per pass): expand 52-60 + cols 78 + THIS ~145 + cols^-1 71 + combine 38
// = 0.389 ms (3-D cfg4 volume pair: 0.493 vs 0.578 ms).  The transforms themselves, not HBM, set the pace of this pass:
// two of them on one 420 MB round trip take as long as two single-transform passes would.
// The column variant above needs 171 us for its 420 MB round trip (234 VGPRs) and forces the planar 4-byte rows passes
// on both ends (116 + 193 us): 0.473 ms in all vs 0.487 ms for dinv_mri_forward + dinv_mri_adjoint.  A first rows
// variant that kept the forward outputs in registers (one LDS tile, L = 16) ran at 256 VGPRs + 22 AGPRs = one wave per
// SIMD and took 234 us.
template <class P, int L>
__global__ __launch_bounds__(256) void mri_rows_normal_kernel(float2* __restrict__ t, const float* __restrict__ mask,
                                                              int64_t nlines, int64_t R, int ncoil, int mask_batch,
                                                              int64_t ntiles, const void* table, float scale) {
    using TF = TileFft<P, false, true, L>;
    using TI = TileFft<P, true, true, L>;
    constexpr int N = P::N;
    constexpr int KS = N + 4;   // row pitch of the k-space tile: rows start 32 B apart modulo the bank width
    __shared__ __attribute__((aligned(16))) float2 buf[TF::lds_floats2];
    __shared__ __attribute__((aligned(16))) float2 ksp[(size_t)L * KS];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int c = N / 2;
    const int64_t vol = R * (int64_t)N;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t line0 = tile * L;
        const int lines = (int)min((int64_t)L, nlines - line0);
        __syncthreads();   // previous tile's inverse transform has left `buf`
        TF::run(buf, tw, lines, c, scale, tid,
                [&](int, int, int line, int n) { return t[(line0 + line) * N + n]; },
                [&](int, int line, int k, int, float2 v) {
                    if (mask) {
                        const int64_t gl = line0 + line;               // (b, n, r) row
                        const int64_t bn = gl / R, r = gl - bn * R;
                        const float* m = mask + ((mask_batch > 1 ? bn / ncoil : 0) * 2) * vol + r * N + k;
                        const float m0 = m[0], m1 = m[vol];
                        v.x = m0 * (m0 * v.x);
                        v.y = m1 * (m1 * v.y);
                    }
                    ksp[line * KS + k] = v;
                });
        __syncthreads();   // masked k-space rows complete; every read of `buf` by the forward transform is done
        TI::run(buf, tw, lines, c, scale, tid,
                [&](int, int, int line, int n) { return ksp[line * KS + n]; },
                [&](int, int line, int k, int, float2 v) { t[(line0 + line) * N + k] = v; });
    }
}

// rows per workgroup (two LDS tiles of L rows): measured at 320 x 320: L = 8 0.413 ms, L = 4 0.389 ms, L = 2 0.453 ms for the
// whole normal operator; the 16-byte-per-lane variant of the transforms (run_v4) was slower here (0.431 ms at L = 8).
template <int N> struct RowsNormalL { static constexpr int value = N >= 512 ? 4 : (N >= 320 ? 4 : (N >= 256 ? 8 : (N >= 128 ? 16 : 32))); };

template <int N>
int launch_rows_normal(float2* t, const float* mask, int64_t nlines, int64_t R, int ncoil, int mask_batch, const void* table,
                       float scale, hipStream_t s) {
    using P = typename PlanFor<N>::P;
    constexpr int L = RowsNormalL<N>::value;
    const int64_t ntiles = ceil_div(nlines, L);
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, 4 * kMaxGrid);
    hipLaunchKernelGGL((mri_rows_normal_kernel<P, L>), dim3(grid), dim3(256), 0, s, t, mask, nlines, R, ncoil, mask_batch,
                       ntiles, table, scale);
    DINV_CHECK_LAUNCH();
    return 0;
}

#define DINV_ALL_STATIC(X) X(16) X(32) X(64) X(128) X(256) X(320) X(512)

bool all_static(const dinv_mri_desc* d) {
    for (int i = 0; i < d->ndim; ++i)
        if (!has_static_plan(d->dims[i])) return false;
    return d->dims[d->ndim - 1] >= 64;  // the rows pass has static kernels from 64 up
}

int validate(const dinv_mri_desc* d) {
    DINV_REQUIRE(d != nullptr, "null descriptor");
    DINV_REQUIRE(d->ndim == 2 || d->ndim == 3, "ndim must be 2 or 3, got %d", d->ndim);
    DINV_REQUIRE(d->batch >= 0 && d->coils >= 1, "bad batch/coils %d/%d", d->batch, d->coils);
    DINV_REQUIRE(d->coil_dim == 1 || d->coils == 1, "single-coil layout needs coils == 1");
    for (int i = 0; i < d->ndim; ++i) {
        DINV_REQUIRE(d->dims[i] >= 1, "bad dim %d", d->dims[i]);
        DINV_REQUIRE(d->plan[i].n == d->dims[i], "plan[%d].n=%d does not match dim %d", i, d->plan[i].n, d->dims[i]);
        DINV_REQUIRE(d->table[i] != nullptr, "null fft table %d", i);
    }
    DINV_REQUIRE(d->mask_batch == 0 || d->mask_batch == 1 || d->mask_batch == d->batch,
                 "mask batch %d incompatible with batch %d", d->mask_batch, d->batch);
    DINV_REQUIRE(d->maps_batch == 0 || d->maps_batch == 1 || d->maps_batch == d->batch,
                 "coil-map batch %d incompatible with batch %d", d->maps_batch, d->batch);
    return 0;
}

inline int64_t volume(const dinv_mri_desc* d) {
    int64_t v = 1;
    for (int i = 0; i < d->ndim; ++i) v *= d->dims[i];
    return v;
}

}  // namespace

extern "C" size_t dinv_mri_workspace_bytes(const dinv_mri_desc* d) {
    if (!d || d->ndim < 2 || d->ndim > 3) return 0;
    return (size_t)d->batch * d->coils * volume(d) * sizeof(float2);
}

extern "C" int dinv_mri_forward(const dinv_mri_desc* d, const float* x, const float* maps, const float* mask,
                                float* y, void* workspace, size_t ws_bytes, dinv_stream_t stream) {
    if (int e = validate(d)) return e;
    if (d->batch == 0) return 0;
    DINV_REQUIRE(x && y && workspace, "null tensor pointer");
    DINV_REQUIRE(ws_bytes >= dinv_mri_workspace_bytes(d), "workspace too small: %zu < %zu", ws_bytes,
                 dinv_mri_workspace_bytes(d));
    DINV_REQUIRE((d->maps_batch != 0) == (maps != nullptr), "maps pointer / maps_batch mismatch");
    DINV_REQUIRE((d->mask_batch != 0) == (mask != nullptr), "mask pointer / mask_batch mismatch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nd = d->ndim;
    const int64_t W = d->dims[nd - 1];
    const int64_t vol = volume(d);
    const int64_t R = vol / W;
    float2* t = reinterpret_cast<float2*>(workspace);
    const int64_t P = (int64_t)d->batch * d->coils;
#ifndef DINV_NO_WAVE2D
    if (mriw::wave2d_ok(d, 0)) return mriw::run_wave2d(d, 0, x, reinterpret_cast<const float2*>(maps), mask, y, t, s);
#endif

    if (all_static(d)) {
        const float2* mp = reinterpret_cast<const float2*>(maps);
        const int64_t N0 = d->dims[0], Q0 = vol / N0;
        const float sc0 = 1.0f / sqrtf((float)N0);
        int e = 0;
        if (Q0 % 4 == 0 && N0 > 16) {
            // fused first pass: coil expansion + transform along the first axis, t written once
            switch (d->dims[0]) {
#define DINV_CASE(NN) case NN: e = launch_cols_expand_fwd<NN>(x, mp, t, d->batch, d->coils, d->maps_batch, Q0, d->table[0], sc0, s); break;
                DINV_CASE(32) DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
            }
        } else {
            // short first axis (a 3-D volume's depth): single-stage transform straight from x and the maps
            ColsCoilLoadIo cio{x, mp, t, d->coils, d->maps_batch, 0, 0};
            e = launch_cols(cio, P, Q0, d->plan[0], d->table[0], 0, 1, sc0, s, d->coils);
        }
        if (e) return e;
        if (nd == 3) {
            C2CIo mio{t, t, 0, 0};
            if ((e = launch_cols(mio, P * d->dims[0], W, d->plan[1], d->table[1], 0, 1, 1.0f / sqrtf((float)d->dims[1]), s))) return e;
        }
        RowsPlanarMaskStoreIo sio{t, y, mask, d->coils, d->mask_batch, R, W, 0, 0};
        return launch_rows(sio, P * R, d->plan[nd - 1], d->table[nd - 1], 0, 1, 1.0f / sqrtf((float)W), s);
    }
    RowsCoilLoadIo rio{x, reinterpret_cast<const float2*>(maps), t, d->coils, d->maps_batch, R, W, 0, 0};
    if (int e = launch_rows(rio, P * R, d->plan[nd - 1], d->table[nd - 1], 0, 1, 1.0f / sqrtf((float)W), s)) return e;
    if (nd == 3) {
        const int64_t H = d->dims[1];
        C2CIo mio{t, t, 0, 0};
        if (int e = launch_cols(mio, P * d->dims[0], W, d->plan[1], d->table[1], 0, 1, 1.0f / sqrtf((float)H), s)) return e;
    }
    const int64_t Na = d->dims[0];
    ColsPlanarMaskStoreIo cio{t, y, mask, d->coils, d->mask_batch, 0, 0};
    return launch_cols(cio, P, vol / Na, d->plan[0], d->table[0], 0, 1, 1.0f / sqrtf((float)Na), s);
}

extern "C" int dinv_mri_adjoint(const dinv_mri_desc* d, const float* y, const float* maps, const float* mask,
                                float* x, void* workspace, size_t ws_bytes, dinv_stream_t stream) {
    if (int e = validate(d)) return e;
    if (d->batch == 0) return 0;
    DINV_REQUIRE(x && y && workspace, "null tensor pointer");
    DINV_REQUIRE(ws_bytes >= dinv_mri_workspace_bytes(d), "workspace too small: %zu < %zu", ws_bytes,
                 dinv_mri_workspace_bytes(d));
    DINV_REQUIRE((d->maps_batch != 0) == (maps != nullptr), "maps pointer / maps_batch mismatch");
    DINV_REQUIRE((d->mask_batch != 0) == (mask != nullptr), "mask pointer / mask_batch mismatch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nd = d->ndim;
    const int64_t W = d->dims[nd - 1];
    const int64_t vol = volume(d);
    const int64_t R = vol / W;
    float2* t = reinterpret_cast<float2*>(workspace);
    const int64_t P = (int64_t)d->batch * d->coils;
#ifndef DINV_NO_WAVE2D
    if (mriw::wave2d_ok(d, 1)) return mriw::run_wave2d(d, 1, y, reinterpret_cast<const float2*>(maps), mask, x, t, s);
#endif

    if (all_static(d)) {
        RowsPlanarMaskLoadIo lio{y, mask, t, d->coils, d->mask_batch, R, W, 0, 0};
        if (int e = launch_rows(lio, P * R, d->plan[nd - 1], d->table[nd - 1], 1, 1, 1.0f / sqrtf((float)W), s)) return e;
        if (nd == 3) {
            C2CIo mio{t, t, 0, 0};
            if (int e = launch_cols(mio, P * d->dims[0], W, d->plan[1], d->table[1], 1, 1, 1.0f / sqrtf((float)d->dims[1]), s)) return e;
        }
        const float2* mp = reinterpret_cast<const float2*>(maps);
        const int64_t N0 = d->dims[0], Q0 = vol / N0;
        const float sc0 = 1.0f / sqrtf((float)N0);
        if (vol % 4 == 0 && N0 > 16) {
            // C2C pass along the first axis in place, then a wide streaming coil combination.  A fused form (one workgroup
            // walking the coils of a column tile, next coil's loads in flight during the transform) was built in round 3 and
            // measured 2x slower than these two passes (152 vs 44 + 27 us at cfg2): the serial coil loop leaves too few
            // independent tiles in flight
            C2CIo fio{t, t, 0, 0};
            if (int e = launch_cols(fio, P, Q0, d->plan[0], d->table[0], 1, 1, sc0, s)) return e;
            const dim3 grid((unsigned)ceil_div(vol / 4, 256), (unsigned)ceil_div(d->batch, CB));
            hipLaunchKernelGGL((mri_coil_combine_kernel<8>), grid, dim3(256), 0, s, t, mp, x, vol, d->batch, d->coils,
                               d->maps_batch);
            DINV_CHECK_LAUNCH();
            return 0;
        }
        // short first axis (a 3-D volume's depth): single-stage inverse transform with the coil sum in its store phase
        return launch_cols_combine_inv<16>(t, mp, x, d->batch, d->coils, d->maps_batch, Q0, d->table[0], sc0, s);
    }
    const int64_t Na = d->dims[0];
    ColsPlanarMaskLoadIo cio{y, mask, t, d->coils, d->mask_batch, 0, 0};
    if (int e = launch_cols(cio, P, vol / Na, d->plan[0], d->table[0], 1, 1, 1.0f / sqrtf((float)Na), s)) return e;
    if (nd == 3) {
        const int64_t H = d->dims[1];
        C2CIo mio{t, t, 0, 0};
        if (int e = launch_cols(mio, P * d->dims[0], W, d->plan[1], d->table[1], 1, 1, 1.0f / sqrtf((float)H), s)) return e;
    }
    // rows pass + coil combine
    const dinv_fft_plan& pw = d->plan[nd - 1];
    {
        const float2* mp = reinterpret_cast<const float2*>(maps);
        const int64_t nl = (int64_t)d->batch * R;
        const float sc = 1.0f / sqrtf((float)W);
        switch (pw.n) {
#define DINV_CASE(NN) case NN: return launch_combine_static<NN>(t, mp, x, nl, R, d->coils, d->maps_batch, d->table[nd - 1], sc, s);
            DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
            default: break;
        }
    }
    int lpb = rows_lines_per_block(pw);
    if (lpb > 16) lpb = 16;
    const int LS = fft_line_stride(pw.n);
    const size_t lds = fft_lds_bytes(pw, lpb) + (size_t)lpb * LS * sizeof(float2);
    DINV_REQUIRE(lds <= kMaxLdsBytes, "fft length %d does not fit the 160 KiB LDS tile", pw.n);
    if (int e = set_lds_limit(mri_rows_combine_kernel, lds)) return e;
    const int64_t nlines = (int64_t)d->batch * R;
    const int64_t blocks = ceil_div(nlines, lpb);
    hipLaunchKernelGGL(mri_rows_combine_kernel, dim3((unsigned)blocks), dim3(256), lds, s, t,
                       reinterpret_cast<const float2*>(maps), x, nlines, R, d->coils, d->maps_batch, lpb, pw,
                       d->table[nd - 1], 1, 1.0f / sqrtf((float)W));
    DINV_CHECK_LAUNCH();
    return 0;
}

// A^T A x without the k-space tensor (what L2.grad / CG call every iteration):
//   rows(W) [x, S -> t] ; (cols(H) in place) ; cols(first axis): F, M^2, F^H in one LDS tile ; (cols(H)^-1) ;
//   rows(W)^-1 + coil combine [t, S -> out]
// 2-D: 4 passes over t instead of the 10 of dinv_mri_forward + dinv_mri_adjoint (which also write and re-read y).
static bool normal_ok(const dinv_mri_desc* d) {
    if (!all_static(d)) return false;
    switch (d->dims[d->ndim - 1]) {
#define DINV_CASE(NN) case NN: return true;
        DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
        default: return false;
    }
}

extern "C" int dinv_mri_normal_supported(const dinv_mri_desc* d) { return d && validate(d) == 0 && normal_ok(d) ? 1 : 0; }

extern "C" int dinv_mri_normal(const dinv_mri_desc* d, const float* x, const float* maps, const float* mask, float* out,
                               void* workspace, size_t ws_bytes, dinv_stream_t stream) {
    if (int e = validate(d)) return e;
    if (d->batch == 0) return 0;
    DINV_REQUIRE(normal_ok(d), "dinv_mri_normal: unsupported sizes (see dinv_mri_normal_supported)");
    DINV_REQUIRE(x && out && workspace, "null tensor pointer");
    DINV_REQUIRE(ws_bytes >= dinv_mri_workspace_bytes(d), "workspace too small: %zu < %zu", ws_bytes,
                 dinv_mri_workspace_bytes(d));
    DINV_REQUIRE((d->maps_batch != 0) == (maps != nullptr), "maps pointer / maps_batch mismatch");
    DINV_REQUIRE((d->mask_batch != 0) == (mask != nullptr), "mask pointer / mask_batch mismatch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nd = d->ndim;
    const int64_t W = d->dims[nd - 1];
    const int64_t vol = volume(d);
    const int64_t R = vol / W;
    float2* t = reinterpret_cast<float2*>(workspace);
    const int64_t P = (int64_t)d->batch * d->coils;
    const float2* mp = reinterpret_cast<const float2*>(maps);
#ifndef DINV_NO_WAVE2D
    if (mriw::wave2d_ok(d, 2)) return mriw::run_wave2d(d, 2, x, mp, mask, out, t, s);
#endif
    const float scw = 1.0f / sqrtf((float)W);
    const int64_t N0 = d->dims[0], Q0 = vol / N0;
    const float sc0 = 1.0f / sqrtf((float)N0);
    // first pass (x, S -> t, transformed along the first axis) -> (cols(H)) -> rows(W): F, M^2, F^H in one tile ->
    // (cols(H)^-1) -> cols(first axis)^-1 -> coil combination: the k-space tensor is never formed
    int e = 0;
    C2CIo fio{t, t, 0, 0};
    if (Q0 % 4 == 0 && N0 > 16) {
        switch (d->dims[0]) {
#define DINV_CASE(NN) case NN: e = launch_cols_expand_fwd<NN>(x, mp, t, d->batch, d->coils, d->maps_batch, Q0, d->table[0], sc0, s); break;
            DINV_CASE(32) DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
        }
    } else {
        ColsCoilLoadIo cio{x, mp, t, d->coils, d->maps_batch, 0, 0};
        e = launch_cols(cio, P, Q0, d->plan[0], d->table[0], 0, 1, sc0, s, d->coils);
    }
    if (e) return e;
    if (nd == 3)
        if ((e = launch_cols(fio, P * d->dims[0], W, d->plan[1], d->table[1], 0, 1, 1.0f / sqrtf((float)d->dims[1]), s))) return e;
    switch ((int)W) {
#define DINV_CASE(NN) case NN: e = launch_rows_normal<NN>(t, mask, P * R, R, d->coils, d->mask_batch, d->table[nd - 1], scw, s); break;
        DINV_STATIC_SIZES(DINV_CASE)
#undef DINV_CASE
        default: return fail(2, "dinv_mri_normal: no static rows plan for %d", (int)W);
    }
    if (e) return e;
    if (nd == 3)
        if ((e = launch_cols(fio, P * d->dims[0], W, d->plan[1], d->table[1], 1, 1, 1.0f / sqrtf((float)d->dims[1]), s))) return e;
    if (vol % 4 == 0 && N0 > 16) {
        if ((e = launch_cols(fio, P, Q0, d->plan[0], d->table[0], 1, 1, sc0, s))) return e;
        const dim3 grid((unsigned)ceil_div(vol / 4, 256), (unsigned)ceil_div(d->batch, CB));
        hipLaunchKernelGGL((mri_coil_combine_kernel<8>), grid, dim3(256), 0, s, t, mp, out, vol, d->batch, d->coils, d->maps_batch);
        DINV_CHECK_LAUNCH();
        return 0;
    }
    return launch_cols_combine_inv<16>(t, mp, out, d->batch, d->coils, d->maps_batch, Q0, d->table[0], sc0, s);
}
