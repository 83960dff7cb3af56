// Loop algebra of the iteration drivers as hand-written kernels (gfx950): the PGD gradient step, the CG vector
// updates and the per-sample dot products.
//
// Replaces the ATen elementwise / reduction launches behind
//   fStepPGD.forward + L2.grad        deepinv/optim/optim_iterators/pgd.py:137-139, optim/data_fidelity.py:335-338
//   conjugate_gradient                deepinv/optim/linear/conjugate_gradient.py:48-75, linear/utils.py:6-26
// All streams are 16 bytes per lane; dot products are reduced with wave shuffles, then across the waves of a
// workgroup through LDS, then across workgroups by a second fixed-order pass (no atomics: deterministic).
#include "common.hpp"

#include <cmath>

using namespace dinv;

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// out = clamp(a*x + b*y + c*z + d, lo, hi)   (y, z optional; lo = -inf, hi = +inf: no clamp)
// (torch.clamp propagates NaN; fminf / fmaxf return the non-NaN operand, so a NaN is passed through explicitly: a diverged
// iterate must not turn into a finite-looking -inf / 0)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }
__global__ __launch_bounds__(256) void lincomb_kernel(int64_t n, float a, const float* __restrict__ x, float b,
                                                      const float* __restrict__ y, float c,
                                                      const float* __restrict__ z, float d, float lo, float hi,
                                                      float* __restrict__ out) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = ld4(x + 4 * i);
        v = make_float4(fmaf(a, v.x, d), fmaf(a, v.y, d), fmaf(a, v.z, d), fmaf(a, v.w, d));
        if (y) { const float4 u = ld4(y + 4 * i); v = make_float4(fmaf(b, u.x, v.x), fmaf(b, u.y, v.y), fmaf(b, u.z, v.z), fmaf(b, u.w, v.w)); }
        if (z) { const float4 u = ld4(z + 4 * i); v = make_float4(fmaf(c, u.x, v.x), fmaf(c, u.y, v.y), fmaf(c, u.z, v.z), fmaf(c, u.w, v.w)); }
        st4(out + 4 * i, make_float4(clampf(v.x, lo, hi), clampf(v.y, lo, hi), clampf(v.z, lo, hi), clampf(v.w, lo, hi)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float v = fmaf(a, x[i], d);
        if (y) v = fmaf(b, y[i], v);
        if (z) v = fmaf(c, z[i], v);
        out[i] = clampf(v, lo, hi);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// partial[b][blk] = sum over this block's slice of x[b,:] * y[b,:]
__global__ __launch_bounds__(256) void dot_partial_kernel(int64_t n, const float* __restrict__ x,
                                                          const float* __restrict__ y, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * n;
    const float* yb = y + (int64_t)b * n;
    float acc = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 u = ld4(xb + 4 * i), v = ld4(yb + 4 * i);
        acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc = fmaf(xb[n4 * 4 + threadIdx.x], yb[n4 * 4 + threadIdx.x], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b] = sum_k partial[b][k] in fixed order (one wave per sample)
__global__ __launch_bounds__(64) void dot_final_kernel(int nblk, const float* __restrict__ partial, float* __restrict__ out) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 64) acc += partial[(int64_t)b * nblk + k];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[b] = acc;
}

// CG updates with per-sample scalars kept on the device (conjugate_gradient.py:55-66):
//   mode 0:  alpha_b = num[b] / (den[b] + eps);  x += alpha_b p ;  r -= alpha_b Ap
//   mode 1:  beta_b  = num[b] / (den[b] + eps);  p  = r + beta_b p
// `done` (optional device flag): once the solve has converged the updates are skipped, so iterations issued after
// convergence leave x, r, p exactly as the reference's `break` would (conjugate_gradient.py:61).
__global__ __launch_bounds__(256) void cg_update_kernel(int mode, int64_t n, const float* __restrict__ num,
                                                        const float* __restrict__ den, float eps,
                                                        float* __restrict__ v0, float* __restrict__ v1,
                                                        const float* __restrict__ w0, const float* __restrict__ w1,
                                                        const int32_t* __restrict__ done) {
    if (done && done[0]) return;
    const int b = blockIdx.y;
    const float s = num[b] / (den[b] + eps);
    const int64_t base = (int64_t)b * n, n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t o = base + 4 * i;
        if (mode == 0) {
            const float4 p = ld4(w0 + o), ap = ld4(w1 + o);
            float4 x = ld4(v0 + o), r = ld4(v1 + o);
            x = make_float4(fmaf(s, p.x, x.x), fmaf(s, p.y, x.y), fmaf(s, p.z, x.z), fmaf(s, p.w, x.w));
            r = make_float4(fmaf(-s, ap.x, r.x), fmaf(-s, ap.y, r.y), fmaf(-s, ap.z, r.z), fmaf(-s, ap.w, r.w));
            st4(v0 + o, x); st4(v1 + o, r);
        } else {
            const float4 r = ld4(w0 + o);
            float4 p = ld4(v0 + o);
            p = make_float4(fmaf(s, p.x, r.x), fmaf(s, p.y, r.y), fmaf(s, p.z, r.z), fmaf(s, p.w, r.w));
            st4(v0 + o, p);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t o = base + n4 * 4 + threadIdx.x;
        if (mode == 0) { v0[o] = fmaf(s, w0[o], v0[o]); v1[o] = fmaf(-s, w1[o], v1[o]); }
        else v0[o] = fmaf(s, v0[o], w0[o]);
    }
}

// done |= all_b(res[b] < tol2[b])   (the reference's torch.all(res_new < tol), evaluated on the device)
__global__ __launch_bounds__(64) void cg_check_kernel(int batch, const float* __restrict__ res,
                                                      const float* __restrict__ tol2, int32_t* __restrict__ done) {
    int ok = 1;
    for (int b = threadIdx.x; b < batch; b += 64) ok &= (res[b] < tol2[b]) ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ok &= __shfl_xor(ok, m);
    if (threadIdx.x == 0 && ok) done[0] = 1;
}

// out[i] = s[i] / (d[i mod period] + add): a complex spectrum over a real symbol that is shared by the leading (batch, channel)
// dimensions - the pointwise division of the closed-form proxes (blur.py:331-363, forward.py:1212-1234)
__global__ __launch_bounds__(256) void cdiv_real_kernel(int64_t n, int64_t period, const float2* __restrict__ s,
                                                        const float* __restrict__ d, float add, float2* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float den = d[i % period] + add;
        const float2 v = s[i];
        out[i] = make_float2(v.x / den, v.y / den);
    }
}

// real spectra (or real / imaginary planes) against a real singular-value mask m shared by the leading dimensions
// (DecomposablePhysics, forward.py:1212-1252):  mode 0  out = x / (m m + add)   (prox_l2),
//                                               mode 1  out = x * (m > 1e-5 ? 1 / m : 0)   (A_dagger)
__global__ __launch_bounds__(256) void mask_solve_kernel(int mode, int64_t n, int64_t period, const float* __restrict__ x,
                                                         const float* __restrict__ m, float add, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float mv = m[i % period];
        // m m, then + add: two roundings, as the reference's tensor expression has them (the product is made opaque so that the
        // compiler cannot contract it into a fused multiply-add; __fmul_rn is a plain product to it)
        float sq = mv * mv;
#ifndef DINV_EMU
        asm volatile("" : "+v"(sq));
#endif
        out[i] = mode == 0 ? x[i] / (sq + add) : x[i] * (mv > 1e-5f ? 1.0f / mv : 0.0f);
    }
}

inline unsigned stream_blocks(int64_t n) { return (unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(n / 4 + 1, 256), 1), 2048); }

}  // namespace

static int affine_launch(int64_t n, float a, const float* x, float b, const float* y, float c, const float* z, float d,
                         float lo, float hi, float* out, dinv_stream_t stream) {
    DINV_REQUIRE(n >= 0 && x && out, "bad arguments");
    if (n == 0) return 0;
    DINV_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)y | (uintptr_t)z) % 16 == 0, "tensors must be 16-byte aligned");
    hipLaunchKernelGGL(lincomb_kernel, dim3(stream_blocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, a, x,
                       b, y, c, z, d, lo, hi, out);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_lincomb(int64_t n, float a, const float* x, float b, const float* y, float c, const float* z,
                            float* out, dinv_stream_t stream) {
    return affine_launch(n, a, x, b, y, c, z, 0.f, -INFINITY, INFINITY, out, stream);
}

extern "C" int dinv_affine(int64_t n, float a, const float* x, float b, const float* y, float c, const float* z, float d,
                           float lo, float hi, float* out, dinv_stream_t stream) {
    DINV_REQUIRE(lo <= hi, "empty clamp interval [%g, %g]", (double)lo, (double)hi);
    return affine_launch(n, a, x, b, y, c, z, d, lo, hi, out, stream);
}

extern "C" int dinv_cdiv_real(int64_t n, int64_t period, const float* s, const float* d, float add, float* out,
                              dinv_stream_t stream) {
    DINV_REQUIRE(n >= 0 && period > 0 && s && d && out, "bad arguments");
    if (n == 0) return 0;
    DINV_REQUIRE(((uintptr_t)s | (uintptr_t)out) % 8 == 0, "complex tensors must be 8-byte aligned");
    hipLaunchKernelGGL(cdiv_real_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 2048)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), n, period, reinterpret_cast<const float2*>(s), d, add,
                       reinterpret_cast<float2*>(out));
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_mask_solve(int32_t mode, int64_t n, int64_t period, const float* x, const float* m, float add, float* out,
                               dinv_stream_t stream) {
    DINV_REQUIRE((mode == 0 || mode == 1) && n >= 0 && period > 0 && x && m && out, "bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_solve_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 2048)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), mode, n, period, x, m, add, out);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t dinv_batched_dot_blocks(int64_t n) { return (int32_t)std::min<int64_t>(std::max<int64_t>(ceil_div(n / 4 + 1, 1024), 1), 256); }

extern "C" int dinv_batched_dot(int32_t batch, int64_t n, const float* x, const float* y, float* out, float* partial,
                                dinv_stream_t stream) {
    DINV_REQUIRE(batch >= 0 && n >= 0 && x && y && out && partial, "bad arguments");
    if (batch == 0) return 0;
    DINV_REQUIRE(batch <= 65535, "batch too large");
    DINV_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0 && n % 4 == 0, "dot operands must be 16-byte aligned with n %% 4 == 0");
    const int nblk = dinv_batched_dot_blocks(n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nblk, batch), dim3(256), 0, s, n, x, y, partial);
    hipLaunchKernelGGL(dot_final_kernel, dim3(batch), dim3(64), 0, s, nblk, partial, out);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_cg_update(int32_t mode, int32_t batch, int64_t n, const float* num, const float* den, float eps,
                              float* v0, float* v1, const float* w0, const float* w1, dinv_stream_t stream) {
    DINV_REQUIRE((mode == 0 || mode == 1) && batch >= 0 && n >= 0 && num && den && v0 && w0, "bad arguments");
    DINV_REQUIRE(mode == 1 || (v1 && w1), "mode 0 needs r and Ap");
    if (batch == 0 || n == 0) return 0;
    DINV_REQUIRE(batch <= 65535 && n % 4 == 0, "batch too large or n %% 4 != 0");
    hipLaunchKernelGGL(cg_update_kernel, dim3(stream_blocks(n), batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       mode, n, num, den, eps, v0, v1, w0, w1, (const int32_t*)nullptr);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_cg_update_masked(int32_t mode, int32_t batch, int64_t n, const float* num, const float* den, float eps,
                                     float* v0, float* v1, const float* w0, const float* w1, const int32_t* done,
                                     dinv_stream_t stream) {
    DINV_REQUIRE((mode == 0 || mode == 1) && batch >= 0 && n >= 0 && num && den && v0 && w0 && done, "bad arguments");
    DINV_REQUIRE(mode == 1 || (v1 && w1), "mode 0 needs r and Ap");
    if (batch == 0 || n == 0) return 0;
    DINV_REQUIRE(batch <= 65535 && n % 4 == 0, "batch too large or n %% 4 != 0");
    hipLaunchKernelGGL(cg_update_kernel, dim3(stream_blocks(n), batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       mode, n, num, den, eps, v0, v1, w0, w1, done);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_cg_check(int32_t batch, const float* res, const float* tol2, int32_t* done, dinv_stream_t stream) {
    DINV_REQUIRE(batch >= 0 && res && tol2 && done, "bad arguments");
    if (batch == 0) return 0;
    hipLaunchKernelGGL(cg_check_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), batch, res, tol2, done);
    DINV_CHECK_LAUNCH();
    return 0;
}
