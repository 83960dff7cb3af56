// Parallel-beam Radon transform, its exact adjoint, and the ramp filter (gfx950).
//
// Reference semantics (deepinv/physics/functional/radon.py):
//   forward  : Radon.forward :252-309 = zero-pad W -> G = ceil(sqrt2 W) (:261-268) [or circle mask :270-283],
//              F.grid_sample(x, affine_grid(R(theta))) bilinear / align_corners=True / zero padding (:7-10,
//              :334-341), then sum over the rotated rows.   out layout [B,C,G,A].
//   adjoint  : the reference obtains the exact transpose by autograd (deepinv/physics/forward.py:1302-1362,
//              tomography.py:311-350); here it is a deterministic *gather* (no atomics): for every
//              image pixel and angle the <= 4x4 lattice samples that can touch it are re-evaluated with the
//              bit-identical coordinate code of the forward kernel, so <Ax,v> = <x,A^T v> up to summation order.
//   ramp     : AbstractFilter.forward/_get_fourier_filter :79-149 (zero-pad to P >= 2N, rfft * F, irfft, crop)
//              == linear convolution with h[0] = 1/2, h[odd d] = -2/(pi d)^2, h[even d] = 0  (P >= 2N: no wrap).
//
// The 3 GB precomputed grid and the 1.5 GB/image sampled intermediate of the reference are never
// materialised: coordinates are recomputed from cos/sin/linspace tables.  Images (forward) and sinograms
// (adjoint) are first re-packed batch-innermost ([..][NB]) so one bilinear tap of NB images is a single
// 4*NB-byte vector load and the coordinate/weight arithmetic is amortised over the NB images.
#include "common.hpp"

#pragma clang fp contract(off)  // forward and adjoint must round the sample coordinates identically

using namespace dinv;

namespace {

struct RadonGeom {
    int32_t n_img, W, G, pad, A, circle, NB, groups;
    float scale;  // 1/operator_norm (tomography.py:253-254)
};

// sample position (ix -> column, iy -> row) for grid = [gx, gy] R^T (radon.py:334-341), unnormalised as ATen's grid_sampler
// (align_corners=True): the fan-beam kernels, whose lattice is not uniform
__device__ __forceinline__ void sample_pos(float c, float s, float xj, float xi, float gm1, float& ix, float& iy) {
    const float gx = fmaf(c, xj, s * xi);
    const float gy = fmaf(-s, xj, c * xi);
    ix = ((gx + 1.0f) * 0.5f) * gm1;
    iy = ((gy + 1.0f) * 0.5f) * gm1;
}

// Parallel beam: the lattice is the rotation of the UNIFORM base grid xn = linspace(-1, 1, G) of affine_grid (radon.py:252-342), so
// in pixel units the sample of ray j at step i is   (ix, iy) = ctr + R (j - ctr, i - ctr),   ctr = (G - 1) / 2:
//   ix = fma(s, i - ctr, fma(c, j - ctr, ctr)),   iy = fma(c, i - ctr, fma(-s, j - ctr, ctr))
// - one fused multiply-add per coordinate and step from a per-ray base (j - ctr and i - ctr are exact), two roundings of at most
// half an ulp of G each (6e-5 pixel at G = 729), against nine operations per step for the chain above, whose own roundings are of
// the same size.  Bit-identical in the forward and adjoint kernels of radon.hip and radon_tiled.hip: the adjoint's tap weights
// are the forward's.  (The base grid is the reference's linspace by construction; the `xn` arguments of the entry points keep
// their place in the ABI, the parallel-beam kernels do not read them.)
__device__ __forceinline__ void ray_base(float c, float s, float dj, float ctr, float& bx, float& by) {
    bx = fmaf(c, dj, ctr);
    by = fmaf(-s, dj, ctr);
}
__device__ __forceinline__ void lattice_pos(float c, float s, float bx, float by, float di, float& ix, float& iy) {
    ix = fmaf(s, di, bx);
    iy = fmaf(c, di, by);
}

// ---- x [n_img, W, W] -> xp [groups][(G+2)][(G+2)][NB], zero ring + zero padding, optional circle mask
template <int NB>
__global__ void radon_pack_image(RadonGeom g, const float* __restrict__ x, float* __restrict__ xp) {
    const int GP = g.G + 2;
    const int64_t total = (int64_t)g.groups * GP * GP;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(idx / ((int64_t)GP * GP));
        const int rem = (int)(idx - (int64_t)grp * GP * GP);
        const int r = rem / GP - 1 - g.pad, c = rem % GP - 1 - g.pad;  // original image coordinates
        float v[NB];
        bool in = r >= 0 && r < g.W && c >= 0 && c < g.W;
        if (in && g.circle) {
            // radon.py:270-283: mask = (xax^2 + yax^2 <= 1), axes = 2*k/(W-1) - 1
            const float ya = 2.0f * (float)c / (float)(g.W - 1) - 1.0f;
            const float xa = 2.0f * (float)r / (float)(g.W - 1) - 1.0f;
            in = (xa * xa + ya * ya) <= 1.0f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            v[k] = (in && n < g.n_img) ? x[((int64_t)n * g.W + r) * g.W + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) xp[idx * NB + k] = v[k];
    }
}

// ---- forward: one thread = one ray (detector j, angle a) of NB images, marching the G rotated rows
template <int NB>
__global__ __launch_bounds__(256) void radon_fwd_kernel(RadonGeom g, const float* __restrict__ xp,
                                                        const float* __restrict__ xn, const float2* __restrict__ cs,
                                                        float* __restrict__ sino) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int a = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int grp = blockIdx.z;
    if (j >= g.G || a >= g.A) return;
    const float2 t = cs[a];
    const float c = t.x, s = t.y;
    const float ctr = 0.5f * (float)(g.G - 1);
    const int GP = g.G + 2;
    const float* img = xp + (int64_t)grp * GP * GP * NB;
    float bx, by;
    ray_base(c, s, (float)j - ctr, ctr, bx, by);
    float acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.f;
    for (int i = 0; i < g.G; ++i) {
        float ix, iy;
        lattice_pos(c, s, bx, by, (float)i - ctr, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        const float tx = ix - fx, ty = iy - fy;
        int x0 = (int)fx, y0 = (int)fy;
        const bool ok = x0 >= -1 && x0 <= g.G - 1 && y0 >= -1 && y0 <= g.G - 1;
        if (!ok) continue;
        const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty);
        const float w10 = (1.0f - tx) * ty, w11 = tx * ty;
        const float* p = img + ((int64_t)(y0 + 1) * GP + (x0 + 1)) * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            acc[k] = fmaf(w00, p[k], acc[k]);
            acc[k] = fmaf(w01, p[NB + k], acc[k]);
            acc[k] = fmaf(w10, p[(int64_t)GP * NB + k], acc[k]);
            acc[k] = fmaf(w11, p[(int64_t)GP * NB + NB + k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = grp * NB + k;
        if (n < g.n_img) sino[((int64_t)n * g.G + j) * g.A + a] = acc[k] * g.scale;
    }
}

// ---- sino [n_img, G, A] -> sp [groups][A][G][NB]
template <int NB>
__global__ void radon_pack_sino(RadonGeom g, const float* __restrict__ sino, float* __restrict__ sp) {
    const int64_t total = (int64_t)g.groups * g.A * g.G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(idx / ((int64_t)g.A * g.G));
        const int rem = (int)(idx - (int64_t)grp * g.A * g.G);
        const int a = rem / g.G, j = rem - a * g.G;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            sp[idx * NB + k] = n < g.n_img ? sino[((int64_t)n * g.G + j) * g.A + a] : 0.f;
        }
    }
}

// ---- exact adjoint as a gather: one thread = one image pixel of NB images, loop over angles
template <int NB>
__global__ __launch_bounds__(256) void radon_adj_kernel(RadonGeom g, const float* __restrict__ sp,
                                                        const float* __restrict__ xn, const float2* __restrict__ cs,
                                                        float* __restrict__ x) {
    DINV_DYN_LDS(float, smem);
    float2* cs_s = reinterpret_cast<float2*>(smem + ((g.G + 1) / 2) * 2);  // A  (behind the G words the launch still reserves)
    for (int i = threadIdx.x; i < g.A; i += 256) cs_s[i] = cs[i];
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int grp = blockIdx.z;
    if (col >= g.W || row >= g.W) return;
    const int px = col + g.pad, py = row + g.pad;  // padded-grid pixel (adjoint of the zero pad = crop)
    const float gm1 = (float)(g.G - 1);
    const float ctr = 0.5f * gm1;
    const float dx = (float)px - ctr, dy = (float)py - ctr;
    float acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.f;
    bool live = true;
    if (g.circle) {
        const float ya = 2.0f * (float)col / (float)(g.W - 1) - 1.0f;
        const float xa = 2.0f * (float)row / (float)(g.W - 1) - 1.0f;
        live = (xa * xa + ya * ya) <= 1.0f;
    }
    if (live) {
        for (int a = 0; a < g.A; ++a) {
            const float c = cs_s[a].x, s = cs_s[a].y;
            // approximate inverse map (only used to pick the candidate window)
            const int j0 = (int)floorf(c * dx - s * dy + ctr) - 1;
            const int i0 = (int)floorf(s * dx + c * dy + ctr) - 1;
            const float* sa = sp + ((int64_t)grp * g.A + a) * g.G * NB;
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) {
                const int j = j0 + dj;
                if (j < 0 || j >= g.G) continue;
                float bx, by;
                ray_base(c, s, (float)j - ctr, ctr, bx, by);
                float wsum = 0.f;
#pragma unroll
                for (int di = 0; di < 4; ++di) {
                    const int i = i0 + di;
                    if (i < 0 || i >= g.G) continue;
                    float ix, iy;
                    lattice_pos(c, s, bx, by, (float)i - ctr, ix, iy);
                    const float fx = floorf(ix), fy = floorf(iy);
                    const float tx = ix - fx, ty = iy - fy;
                    const int ex = px - (int)fx, ey = py - (int)fy;  // 0 -> tap weight (1-t), 1 -> t
                    const float wx = ex == 0 ? 1.0f - tx : (ex == 1 ? tx : 0.f);
                    const float wy = ey == 0 ? 1.0f - ty : (ey == 1 ? ty : 0.f);
                    wsum += wx * wy;
                }
                if (wsum != 0.f) {
                    const float* v = sa + (int64_t)j * NB;
#pragma unroll
                    for (int k = 0; k < NB; ++k) acc[k] = fmaf(wsum, v[k], acc[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = grp * NB + k;
        if (n < g.n_img) x[((int64_t)n * g.W + row) * g.W + col] = acc[k] * g.scale;
    }
}

// =====================================================================================================
// Fan-beam geometry (fan_beam_grid, deepinv/physics/functional/radon.py:16-52; Radon.forward with fan_beam=True
// :252-342; Tomography uses the exact adjoint and RampFilter + adjoint for FBP, tomography.py:229-350).
// Ray (detector d, angle a) samples the G points  R(theta_a) (xm_i, yd_d * sc_i), i = 0..G-1:
//   xm = linspace(-1,1,G) runs along the central ray, yd = linspace(-1,1,n_det) across the detector, and the stretch
//   sc_i = 0.5 * L_det * (xm_i + r_src) / (r_src + r_det) grows linearly from the source to the detector.
// With sample_pos(c, s, xj := xm_i, xi := yd_d * sc_i) the arithmetic is that of the parallel-beam kernels with the
// roles of the two lattice axes exchanged; the same gather structure applies (first-generation kernels: one thread per
// ray / per pixel, operands through L1).  sino layout [n_img, n_det, A].
// =====================================================================================================
template <int NB>
__global__ __launch_bounds__(256) void radon_fan_fwd_kernel(RadonGeom g, int n_det, const float* __restrict__ xp,
                                                            const float* __restrict__ xm, const float* __restrict__ sc,
                                                            const float* __restrict__ yd, const float2* __restrict__ cs,
                                                            float* __restrict__ sino) {
    DINV_DYN_LDS(float, tab);      // xm [G], sc [G]
    float* xm_s = tab;
    float* sc_s = tab + g.G;
    for (int i = threadIdx.x; i < g.G; i += 256) { xm_s[i] = xm[i]; sc_s[i] = sc[i]; }
    __syncthreads();
    const int d = blockIdx.x * 64 + (threadIdx.x & 63);
    const int a = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int grp = blockIdx.z;
    if (d >= n_det || a >= g.A) return;
    const float2 t = cs[a];
    const float c = t.x, s = t.y;
    const float gm1 = (float)(g.G - 1);
    const int GP = g.G + 2;
    const float* img = xp + (int64_t)grp * GP * GP * NB;
    const float ydd = yd[d];
    float acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.f;
    for (int i = 0; i < g.G; ++i) {
        float ix, iy;
        sample_pos(c, s, xm_s[i], ydd * sc_s[i], gm1, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        // range test in floating point: far-away detector pixels produce coordinates beyond the int range
        if (!(fx >= -1.0f && fx <= (float)(g.G - 1) && fy >= -1.0f && fy <= (float)(g.G - 1))) continue;
        const float tx = ix - fx, ty = iy - fy;
        const int x0 = (int)fx, y0 = (int)fy;
        const float w00 = (1.0f - tx) * (1.0f - ty), w01 = tx * (1.0f - ty);
        const float w10 = (1.0f - tx) * ty, w11 = tx * ty;
        const float* p = img + ((int64_t)(y0 + 1) * GP + (x0 + 1)) * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            acc[k] = fmaf(w00, p[k], acc[k]);
            acc[k] = fmaf(w01, p[NB + k], acc[k]);
            acc[k] = fmaf(w10, p[(int64_t)GP * NB + k], acc[k]);
            acc[k] = fmaf(w11, p[(int64_t)GP * NB + NB + k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = grp * NB + k;
        if (n < g.n_img) sino[((int64_t)n * n_det + d) * g.A + a] = acc[k] * g.scale;
    }
}

// exact adjoint as a gather: one thread = one image pixel of NB images.  For every angle the pixel is rotated back into the fan
// frame (qx along the central ray, qy across).  A sample (march index i, detector d) touches the pixel iff both of its
// coordinates lie within one pixel of it, i.e. iff its offset, seen in the fan frame, lies inside the rotated unit square: the
// march indices within |c| + |s| pixels of qx (two or three of them) and, for each, the detectors within (|c| + |s|) / stretch_i
// detector pitches of qy / stretch_i (one or two at the usual geometries).  Both index ranges follow in closed form - no search,
// no division (the reciprocal stretch is tabulated) - and every candidate is evaluated with the FORWARD kernel's coordinate
// code; its weight is the product of the two hat functions clamp(1 - |t - p|), i.e. exactly the bilinear tap weight the
// forward gives that pixel (zero outside the support, so a candidate too many costs nothing but its instructions).
template <int NB>
__global__ __launch_bounds__(256) void radon_fan_adj_kernel(RadonGeom g, int n_det, const float* __restrict__ sp,
                                                            const float* __restrict__ xm, const float* __restrict__ sc,
                                                            const float* __restrict__ yd, const float2* __restrict__ cs,
                                                            float* __restrict__ x) {
    DINV_DYN_LDS(float, tab);      // xm [G], sc [G], 1 / sc [G], yd [n_det]
    float* xm_s = tab;
    float* sc_s = tab + g.G;
    float* rsc_s = tab + 2 * g.G;  // reciprocal stretch (0 where the stretch vanishes: that march index sees every detector)
    float* yd_s = tab + 3 * g.G;
    for (int i = threadIdx.x; i < g.G; i += 256) {
        const float v = sc[i];
        xm_s[i] = xm[i]; sc_s[i] = v; rsc_s[i] = fabsf(v) > 1e-20f ? 1.0f / v : 0.f;
    }
    for (int i = threadIdx.x; i < n_det; i += 256) yd_s[i] = yd[i];
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int grp = blockIdx.z;
    if (col >= g.W || row >= g.W) return;
    const int px = col + g.pad, py = row + g.pad;
    const float fpx = (float)px, fpy = (float)py;
    const float gm1 = (float)(g.G - 1), dm1 = (float)(n_det - 1), hgm1 = 0.5f * gm1, hdm1 = 0.5f * dm1;
    const float gx = 2.0f * fpx / gm1 - 1.0f, gy = 2.0f * fpy / gm1 - 1.0f;   // pixel centre, normalised
    float acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.f;
    bool live = true;
    if (g.circle) {
        const float ya = 2.0f * (float)col / (float)(g.W - 1) - 1.0f;
        const float xa = 2.0f * (float)row / (float)(g.W - 1) - 1.0f;
        live = (xa * xa + ya * ya) <= 1.0f;
    }
    if (live) {
        for (int a = 0; a < g.A; ++a) {
            const float c = cs[a].x, s = cs[a].y;
            const float qx = c * gx - s * gy, qy = s * gx + c * gy;          // inverse rotation (candidate ranges only)
            const float reach = fabsf(c) + fabsf(s);                         // half width of the rotated unit square, in pixels
            const float qi = (qx + 1.0f) * hgm1;                             // march coordinate of the pixel
            // march indices within reach (+ 0.01 for the rounding of this inverse map) of qi
            int ilo = (int)ceilf(qi - reach - 0.01f), ihi = (int)floorf(qi + reach + 0.01f);
            ilo = ilo < 0 ? 0 : ilo;
            ihi = ihi > g.G - 1 ? g.G - 1 : ihi;
            const float reach_n = reach / hgm1;                              // the same in normalised units
            const float* sa = sp + ((int64_t)grp * g.A + a) * n_det * NB;
            for (int i = ilo; i <= ihi; ++i) {
                const float sci = sc_s[i], xmi = xm_s[i], rsc = rsc_s[i];
                int dlo = 0, dhi = n_det - 1;
                if (rsc != 0.f && n_det > 1) {
                    const float dc = fmaf(qy * rsc, hdm1, hdm1);             // detector coordinate of the pixel at this march index
                    const float m = fmaf(reach_n * fabsf(rsc), hdm1, 0.01f); // half width of its support there, in detector pitches
                    const float lo = ceilf(dc - m), hi = floorf(dc + m);
                    if (!(hi >= 0.0f && lo <= dm1)) continue;
                    dlo = lo > 0.0f ? (int)lo : 0;
                    dhi = hi < dm1 ? (int)hi : n_det - 1;
                }
                for (int d = dlo; d <= dhi; ++d) {
                    float ix, iy;
                    sample_pos(c, s, xmi, yd_s[d] * sci, gm1, ix, iy);       // the forward's coordinates, bit for bit
                    const float wx = fmaxf(1.0f - fabsf(ix - fpx), 0.0f), wy = fmaxf(1.0f - fabsf(iy - fpy), 0.0f);
                    const float w = wx * wy;
                    const float* v = sa + (int64_t)d * NB;
#pragma unroll
                    for (int k = 0; k < NB; ++k) acc[k] = fmaf(w, v[k], acc[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = grp * NB + k;
        if (n < g.n_img) x[((int64_t)n * g.W + row) * g.W + col] = acc[k] * g.scale;
    }
}

// ---- interpolating back-projection (IRadon.forward, radon.py:396-444; used when adjoint_via_backprop=False):
// reco[y][x] = sum_a bilinear(sino, col = ixtab[a] (~ a), row = ((x*cos - y*sin + 1)/2)(G-1)), zero padding,
// on the G x G grid, cropped to W x W, optional disc mask.  One thread per output pixel of one image.
__global__ __launch_bounds__(256) void iradon_kernel(RadonGeom g, const float* __restrict__ sino,
                                                     const float* __restrict__ xn, const float2* __restrict__ cs,
                                                     const float* __restrict__ ixtab, float* __restrict__ out) {
    DINV_DYN_LDS(float, smem);
    float2* cs_s = reinterpret_cast<float2*>(smem);
    float* ix_s = smem + 2 * g.A;
    for (int i = threadIdx.x; i < g.A; i += 256) { cs_s[i] = cs[i]; ix_s[i] = ixtab[i]; }
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (col >= g.W || row >= g.W) return;
    const float xg = xn[col + g.pad], yg = xn[row + g.pad];  // meshgrid(linspace(-1,1,G)), radon.py:446-456
    const float gm1 = (float)(g.G - 1);
    const float* sn = sino + (int64_t)n * g.G * g.A;
    float acc = 0.f;
    const bool live = !g.circle || (xg * xg + yg * yg <= 1.0f);
    if (live) {
        for (int a = 0; a < g.A; ++a) {
            const float t = xg * cs_s[a].x - yg * cs_s[a].y;   // _XYtoT (radon.py:458-461)
            const float iy = ((t + 1.0f) * 0.5f) * gm1;
            const float ix = ix_s[a];
            const float fy = floorf(iy), fx = floorf(ix);
            const float ty = iy - fy, tx = ix - fx;
            const int y0 = (int)fy, x0 = (int)fx;
            float v = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = y0 + dy, xx = x0 + dx;
                    if (yy >= 0 && yy < g.G && xx >= 0 && xx < g.A) {
                        const float w = (dy ? ty : 1.0f - ty) * (dx ? tx : 1.0f - tx);
                        v = fmaf(w, sn[(int64_t)yy * g.A + xx], v);
                    }
                }
            acc += v;
        }
    }
    out[((int64_t)n * g.W + row) * g.W + col] = acc * g.scale;
}

// ---- ramp filter along the detector axis: out[n][j][a] = sum_m h[j-m] y[n][m][a]
constexpr int RJ = 8;  // outputs per thread
__global__ __launch_bounds__(256) void ramp_kernel(int n_img, int N, int A, const float* __restrict__ y,
                                                   float* __restrict__ out) {
    DINV_DYN_LDS(float, h_s);  // h[d], d = 0..N-1
    for (int d = threadIdx.x; d < N; d += 256) {
        float v = 0.f;
        if (d == 0) v = 0.5f;
        else if (d & 1) {
            const float pd = 3.14159265358979323846f * (float)d;
            v = -2.0f / (pd * pd);
        }
        h_s[d] = v;
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * RJ;
    const int n = blockIdx.z;
    if (a >= A) return;
    float acc[RJ];
#pragma unroll
    for (int q = 0; q < RJ; ++q) acc[q] = 0.f;
    const float* src = y + (int64_t)n * N * A + a;
    for (int m = 0; m < N; ++m) {
        const float v = src[(int64_t)m * A];
#pragma unroll
        for (int q = 0; q < RJ; ++q) {
            int d = j0 + q - m;
            d = d < 0 ? -d : d;
            acc[q] = fmaf(h_s[d < N ? d : 0], (d < N) ? v : 0.f, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < RJ; ++q)
        if (j0 + q < N) out[((int64_t)n * N + j0 + q) * A + a] = acc[q];
}

int check_desc(const dinv_radon_desc* d, RadonGeom* g) {
    DINV_REQUIRE(d != nullptr, "null descriptor");
    DINV_REQUIRE(d->n_img >= 0 && d->width >= 2 && d->grid >= d->width && d->n_angles >= 1, "bad radon geometry");
    DINV_REQUIRE(d->pad_before >= 0 && d->pad_before + d->width <= d->grid, "bad padding");
    g->n_img = d->n_img; g->W = d->width; g->G = d->grid; g->pad = d->pad_before; g->A = d->n_angles;
    g->circle = d->circle; g->scale = d->scale;
    g->NB = d->n_img >= 8 ? 8 : d->n_img >= 4 ? 4 : d->n_img >= 2 ? 2 : 1;
    g->groups = d->n_img == 0 ? 0 : (d->n_img + g->NB - 1) / g->NB;
    return 0;
}

}  // namespace

extern "C" size_t dinv_radon_workspace_bytes(const dinv_radon_desc* d, int32_t adjoint) {
    RadonGeom g;
    if (!d || check_desc(d, &g)) return 0;
    if (adjoint) return (size_t)g.groups * g.A * g.G * g.NB * sizeof(float);
    return (size_t)g.groups * (g.G + 2) * (g.G + 2) * g.NB * sizeof(float);
}

#define DINV_NB_DISPATCH(NBV, STMT)                                  \
    switch (NBV) {                                                   \
        case 8: { constexpr int NB = 8; STMT; } break;               \
        case 4: { constexpr int NB = 4; STMT; } break;               \
        case 2: { constexpr int NB = 2; STMT; } break;               \
        default: { constexpr int NB = 1; STMT; } break;              \
    }

extern "C" int dinv_radon_forward(const dinv_radon_desc* d, const float* x, const float* xn, const float* cs,
                                  float* sino, void* ws, size_t ws_bytes, dinv_stream_t stream) {
    RadonGeom g;
    if (int e = check_desc(d, &g)) return e;
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(x && xn && cs && sino && ws, "null pointer");
    DINV_REQUIRE(ws_bytes >= dinv_radon_workspace_bytes(d, 0), "workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* xp = reinterpret_cast<float*>(ws);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    const int64_t npk = (int64_t)g.groups * (g.G + 2) * (g.G + 2);
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    const dim3 grid((g.G + 63) / 64, (g.A + 3) / 4, g.groups);
    DINV_NB_DISPATCH(g.NB, {
        hipLaunchKernelGGL(radon_pack_image<NB>, dim3(pk_blocks), dim3(256), 0, s, g, x, xp);
        hipLaunchKernelGGL(radon_fwd_kernel<NB>, grid, dim3(256), g.G * sizeof(float), s, g, xp, xn, cs2, sino);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_radon_adjoint(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                                  float* x, void* ws, size_t ws_bytes, dinv_stream_t stream) {
    RadonGeom g;
    if (int e = check_desc(d, &g)) return e;
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(x && xn && cs && sino && ws, "null pointer");
    DINV_REQUIRE(ws_bytes >= dinv_radon_workspace_bytes(d, 1), "workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* sp = reinterpret_cast<float*>(ws);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    const int64_t npk = (int64_t)g.groups * g.A * g.G;
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    const dim3 grid((g.W + 63) / 64, (g.W + 3) / 4, g.groups);
    const size_t lds = (size_t)(((g.G + 1) / 2) * 2) * sizeof(float) + (size_t)g.A * sizeof(float2);
    DINV_REQUIRE(lds <= 64 * 1024, "too many angles/detectors for the LDS tables (%zu B)", lds);
    DINV_NB_DISPATCH(g.NB, {
        hipLaunchKernelGGL(radon_pack_sino<NB>, dim3(pk_blocks), dim3(256), 0, s, g, sino, sp);
        hipLaunchKernelGGL(radon_adj_kernel<NB>, grid, dim3(256), lds, s, g, sp, xn, cs2, x);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_radon_backproject(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                                      const float* ixtab, float* out, dinv_stream_t stream) {
    RadonGeom g;
    if (int e = check_desc(d, &g)) return e;
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(sino && xn && cs && ixtab && out, "null pointer");
    DINV_REQUIRE(g.n_img <= 65535, "too many images per call");
    const size_t lds = (size_t)g.A * 3 * sizeof(float);
    DINV_REQUIRE(lds <= 64 * 1024, "too many angles for the LDS tables");
    hipLaunchKernelGGL(iradon_kernel, dim3((g.W + 63) / 64, (g.W + 3) / 4, g.n_img), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), g, sino, xn, reinterpret_cast<const float2*>(cs), ixtab, out);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_radon_ramp(int32_t n_img, int32_t n_det, int32_t n_angles, const float* sino, float* out,
                               dinv_stream_t stream) {
    DINV_REQUIRE(n_img >= 0 && n_det >= 1 && n_angles >= 1, "bad ramp geometry");
    if (n_img == 0) return 0;
    DINV_REQUIRE(sino && out && sino != out, "null or aliased pointer");
    DINV_REQUIRE((size_t)n_det * sizeof(float) <= 64 * 1024, "detector axis too long (%d)", n_det);
    DINV_REQUIRE(n_img <= 65535, "too many sinograms per call");
    hipLaunchKernelGGL(ramp_kernel, dim3((n_angles + 255) / 256, (n_det + RJ - 1) / RJ, n_img), dim3(256),
                       n_det * sizeof(float), reinterpret_cast<hipStream_t>(stream), n_img, n_det, n_angles, sino, out);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t dinv_radon_fan_workspace_bytes(const dinv_radon_desc* d, int32_t n_det, int32_t adjoint) {
    RadonGeom g;
    if (!d || n_det < 1 || check_desc(d, &g)) return 0;
    if (adjoint) return (size_t)g.groups * g.A * n_det * g.NB * sizeof(float);
    return (size_t)g.groups * (g.G + 2) * (g.G + 2) * g.NB * sizeof(float);
}

extern "C" int dinv_radon_fan_forward(const dinv_radon_desc* d, int32_t n_det, const float* x, const float* xm,
                                      const float* sc, const float* yd, const float* cs, float* sino, void* ws,
                                      size_t ws_bytes, dinv_stream_t stream) {
    RadonGeom g;
    if (int e = check_desc(d, &g)) return e;
    DINV_REQUIRE(n_det >= 1, "bad detector count %d", n_det);
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(x && xm && sc && yd && cs && sino && ws, "null pointer");
    DINV_REQUIRE(ws_bytes >= dinv_radon_fan_workspace_bytes(d, n_det, 0), "workspace too small");
    DINV_REQUIRE((size_t)2 * g.G * sizeof(float) <= 64 * 1024, "grid too large for the LDS tables");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* xp = reinterpret_cast<float*>(ws);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    const int64_t npk = (int64_t)g.groups * (g.G + 2) * (g.G + 2);
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    const dim3 grid((n_det + 63) / 64, (g.A + 3) / 4, g.groups);
    DINV_NB_DISPATCH(g.NB, {
        hipLaunchKernelGGL(radon_pack_image<NB>, dim3(pk_blocks), dim3(256), 0, s, g, x, xp);
        hipLaunchKernelGGL(radon_fan_fwd_kernel<NB>, grid, dim3(256), 2 * g.G * sizeof(float), s, g, n_det, xp, xm, sc, yd, cs2, sino);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_radon_fan_adjoint(const dinv_radon_desc* d, int32_t n_det, const float* sino, const float* xm,
                                      const float* sc, const float* yd, const float* cs, float* x, void* ws,
                                      size_t ws_bytes, dinv_stream_t stream) {
    RadonGeom g;
    if (int e = check_desc(d, &g)) return e;
    DINV_REQUIRE(n_det >= 1, "bad detector count %d", n_det);
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(x && xm && sc && yd && cs && sino && ws, "null pointer");
    DINV_REQUIRE(ws_bytes >= dinv_radon_fan_workspace_bytes(d, n_det, 1), "workspace too small");
    const size_t lds = (size_t)(3 * g.G + n_det) * sizeof(float);
    DINV_REQUIRE(lds <= 64 * 1024, "grid / detector too large for the LDS tables (%zu B)", lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* sp = reinterpret_cast<float*>(ws);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    RadonGeom gs = g;
    gs.G = n_det;   // radon_pack_sino: [n_img, n_det, A] -> [groups][A][n_det][NB]
    const int64_t npk = (int64_t)g.groups * g.A * n_det;
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    const dim3 grid((g.W + 63) / 64, (g.W + 3) / 4, g.groups);
    DINV_NB_DISPATCH(g.NB, {
        hipLaunchKernelGGL(radon_pack_sino<NB>, dim3(pk_blocks), dim3(256), 0, s, gs, sino, sp);
        hipLaunchKernelGGL(radon_fan_adj_kernel<NB>, grid, dim3(256), lds, s, g, n_det, sp, xm, sc, yd, cs2, x);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}
