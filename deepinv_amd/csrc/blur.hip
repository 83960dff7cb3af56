// Blur / Downsampling spatial kernels and real<->half-complex FFT passes (gfx950).
//
// Reference semantics:
//   conv2d            deepinv/physics/functional/convolution.py:42-107   true convolution (flipped filter),
//                     pad (pw-iw, pw, ph-ih, ph) with mode valid|circular|reflect|replicate|constant, per-(b,c)
//                     filters via grouped conv;   Downsampling.A = conv2d(...)[..., ::f, ::f]  (blur.py:255-283)
//   conv_transpose2d  convolution.py:110-164 + _apply_transpose_padding :689-758 (fold the padded border back);
//                     Downsampling.A_adjoint = conv_transpose2d(zero-insert_f(y))            (blur.py:285-329)
//   rfft2 / irfft2    BlurFFT.V_adjoint/U/U_adjoint/V (blur.py:639-657), _circular_conv_fft (convolution.py:837-865)
//
// The transposed convolution is a gather: every output pixel enumerates its pre-images under the padding
// map (<= 3 per axis, a contiguous range for `replicate`), so no atomics and a fixed summation order.
#include "fft_core.hpp"
#include "fft_launch.hpp"

using namespace dinv;

namespace {

enum PadMode { PAD_VALID = 0, PAD_CIRCULAR = 1, PAD_REFLECT = 2, PAD_REPLICATE = 3, PAD_CONSTANT = 4 };

struct ConvGeom {
    int32_t B, C, H, W;      // full-resolution image (input of A)
    int32_t fb, fc, h, w;    // filter [fb in {1,B}, fc in {1,C}, h, w]
    int32_t mode, stride;
    int32_t pt, pl, pb, pr;  // pads (top,left,bottom,right); 0 for valid
    int32_t Ho, Wo;          // output of A
    uint32_t smagic;         // ceil(2^32 / stride) when every padded coordinate is below 2^16 (q / stride = mulhi(q, smagic)), else 0
};

// q / stride and q % stride for a padded coordinate q >= 0 (a hardware-less integer division costs ~30 instructions: the transposed
// strided convolution needs one pair per candidate row / column)
__device__ __forceinline__ void divmod_stride(const ConvGeom& g, int q, int& quo, int& rem) {
    quo = g.smagic ? (int)__umulhi((unsigned)q, g.smagic) : q / g.stride;
    rem = q - quo * g.stride;
}

// source index of padded coordinate q (relative to the unpadded axis), or -1 for a zero tap
__device__ __forceinline__ int pad_map(int q, int n, int mode) {
    if (q >= 0 && q < n) return q;
    switch (mode) {
        case PAD_CIRCULAR: { int r = q % n; return r < 0 ? r + n : r; }
        case PAD_REFLECT: return q < 0 ? -q : 2 * (n - 1) - q;
        case PAD_REPLICATE: return q < 0 ? 0 : n - 1;
        default: return -1;
    }
}

// y[b,c,io,jo] = sum_{u,v} kf[u,v] * xpad[io*s+u, jo*s+v],  kf[u,v] = k[h-1-u, w-1-v]
__global__ __launch_bounds__(256) void conv2d_pad_kernel(ConvGeom g, const float* __restrict__ x,
                                                         const float* __restrict__ k, float* __restrict__ y) {
    DINV_DYN_LDS(float, ks);  // flipped filter of this (b,c)
    const int bc = blockIdx.z, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * g.h * g.w;
    for (int i = threadIdx.x; i < g.h * g.w; i += 256) ks[i] = kf[g.h * g.w - 1 - i];
    __syncthreads();
    const int jo = blockIdx.x * 64 + (threadIdx.x & 63);
    const int io = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (io >= g.Ho || jo >= g.Wo) return;
    const float* img = x + (int64_t)bc * g.H * g.W;
    float acc = 0.f;
    for (int u = 0; u < g.h; ++u) {
        const int r = pad_map(io * g.stride + u - g.pt, g.H, g.mode);
        if (r < 0) continue;
        const float* row = img + (int64_t)r * g.W;
        for (int v = 0; v < g.w; ++v) {
            const int cc = pad_map(jo * g.stride + v - g.pl, g.W, g.mode);
            if (cc >= 0) acc = fmaf(ks[u * g.w + v], row[cc], acc);
        }
    }
    y[((int64_t)bc * g.Ho + io) * g.Wo + jo] = acc;
}

// pre-images of unpadded index t under pad_map, in padded coordinates [0, n+pt+pb): up to 3 ranges [lo,hi]
struct Pre { int lo[3], hi[3], n; };
__device__ __forceinline__ Pre preimages(int t, int n, int p0, int p1, int mode) {
    Pre r;
    r.n = 1;
    r.lo[0] = r.hi[0] = t + p0;
    const int np = n + p0 + p1;
    if (mode == PAD_CIRCULAR) {
        int a = t + p0 - n, b = t + p0 + n;
        if (a >= 0) { r.lo[r.n] = r.hi[r.n] = a; ++r.n; }
        if (b < np) { r.lo[r.n] = r.hi[r.n] = b; ++r.n; }
    } else if (mode == PAD_REFLECT) {
        if (t >= 1 && t <= p0) { r.lo[r.n] = r.hi[r.n] = p0 - t; ++r.n; }
        const int q = 2 * (n - 1) - t;  // must land in [n, n+p1-1] and differ from t
        if (q >= n && q <= n + p1 - 1) { r.lo[r.n] = r.hi[r.n] = q + p0; ++r.n; }
    } else if (mode == PAD_REPLICATE) {
        if (t == 0) r.lo[0] = 0;
        if (t == n - 1) r.hi[0] = np - 1;
    }
    return r;
}

// x[b,c,r,cc] = sum_{pr in pre(r), pc in pre(cc)} sum_{u,v} kf[u,v] * yz[pr-u, pc-v]
// with yz the stride-s zero-inserted measurement (yz[a,b] = y[a/s,b/s] when both divisible).
__global__ __launch_bounds__(256) void conv2d_pad_transpose_kernel(ConvGeom g, const float* __restrict__ y,
                                                                   const float* __restrict__ k,
                                                                   float* __restrict__ x) {
    DINV_DYN_LDS(float, ks);
    const int bc = blockIdx.z, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * g.h * g.w;
    for (int i = threadIdx.x; i < g.h * g.w; i += 256) ks[i] = kf[g.h * g.w - 1 - i];
    __syncthreads();
    const int cc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= g.H || cc >= g.W) return;
    const float* meas = y + (int64_t)bc * g.Ho * g.Wo;
    const int s = g.stride;
    // extent of the zero-inserted / padded-conv output grid: positions io*s, io < Ho
    const Pre prow = preimages(r, g.H, g.pt, g.pb, g.mode);
    const Pre pcol = preimages(cc, g.W, g.pl, g.pr, g.mode);
    float acc = 0.f;
    for (int ir = 0; ir < prow.n; ++ir)
        for (int pr = prow.lo[ir]; pr <= prow.hi[ir]; ++pr) {
            int io0, u0;
            divmod_stride(g, pr, io0, u0);
            // taps u = u0 + t s (pr - u divisible by s), measurement row io = (pr - u) / s = io0 - t: no division inside
            for (int u = u0, io = io0; u < g.h && io >= 0; u += s, --io) {
                if (io >= g.Ho) continue;
                const float* mrow = meas + (int64_t)io * g.Wo;
                const float* krow = ks + u * g.w;
                for (int ic = 0; ic < pcol.n; ++ic)
                    for (int pc = pcol.lo[ic]; pc <= pcol.hi[ic]; ++pc) {
                        int jo0, v0;
                        divmod_stride(g, pc, jo0, v0);
                        for (int v = v0, jo = jo0; v < g.w && jo >= 0; v += s, --jo)
                            if (jo < g.Wo) acc = fmaf(krow[v], mrow[jo], acc);
                    }
            }
        }
    x[((int64_t)bc * g.H + r) * g.W + cc] = acc;
}

// gradient of the convolution w.r.t. its filter (what autograd through F.conv2d gives the reference: blind / learned kernels,
// least_squares_implicit_backward, deepinv/optim/linear/least_squares.py:315-339):
//   dk[b,c,h-1-u,w-1-v] = sum_{io,jo} gy[b,c,io,jo] * xpad[io*s+u, jo*s+v]      per (b, c) plane, fixed summation order
// one workgroup per (filter tap, plane): the taps of a plane read the same two images, shifted (L2)
__global__ __launch_bounds__(256) void conv2d_filter_grad_kernel(ConvGeom g, const float* __restrict__ x,
                                                                 const float* __restrict__ gy, float* __restrict__ dk) {
    __shared__ float red[4];
    const int tap = blockIdx.x, u = tap / g.w, v = tap - u * g.w;
    const int bc = blockIdx.y;
    const float* img = x + (int64_t)bc * g.H * g.W;
    const float* go = gy + (int64_t)bc * g.Ho * g.Wo;
    float acc = 0.f;
    for (int io = threadIdx.x >> 6; io < g.Ho; io += 4) {
        const int r = pad_map(io * g.stride + u - g.pt, g.H, g.mode);
        if (r < 0) continue;
        const float* row = img + (int64_t)r * g.W;
        for (int jo = threadIdx.x & 63; jo < g.Wo; jo += 64) {
            const int cc = pad_map(jo * g.stride + v - g.pl, g.W, g.mode);
            if (cc >= 0) acc = fmaf(go[(int64_t)io * g.Wo + jo], row[cc], acc);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dk[(int64_t)bc * g.h * g.w + (g.h * g.w - 1 - tap)] = (red[0] + red[1]) + (red[2] + red[3]);
}

int make_geom(const dinv_conv_desc* d, ConvGeom* g) {
    DINV_REQUIRE(d != nullptr, "null descriptor");
    DINV_REQUIRE(d->batch >= 0 && d->channels >= 1 && d->height >= 1 && d->width >= 1, "bad image geometry");
    DINV_REQUIRE(d->fh >= 1 && d->fw >= 1 && (d->fbatch == 1 || d->fbatch == d->batch) &&
                 (d->fchannels == 1 || d->fchannels == d->channels), "filter shape not broadcastable");
    DINV_REQUIRE(d->mode >= PAD_VALID && d->mode <= PAD_CONSTANT, "unknown padding mode %d", d->mode);
    DINV_REQUIRE(d->stride >= 1, "bad stride");
    g->B = d->batch; g->C = d->channels; g->H = d->height; g->W = d->width;
    g->fb = d->fbatch; g->fc = d->fchannels; g->h = d->fh; g->w = d->fw;
    g->mode = d->mode; g->stride = d->stride;
    g->smagic = (d->height + d->fh < 65536 && d->width + d->fw < 65536 && d->stride < 65536)
                    ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)d->stride - 1) / (uint64_t)d->stride) : 0u;
    int fullH, fullW;
    if (d->mode == PAD_VALID) {
        g->pt = g->pl = g->pb = g->pr = 0;
        DINV_REQUIRE(d->height >= d->fh && d->width >= d->fw, "filter larger than image in 'valid' mode");
        fullH = d->height - d->fh + 1; fullW = d->width - d->fw + 1;
    } else {
        const int ph = d->fh / 2, pw = d->fw / 2, ih = (d->fh - 1) % 2, iw = (d->fw - 1) % 2;
        g->pt = ph - ih; g->pb = ph; g->pl = pw - iw; g->pr = pw;
        if (d->mode == PAD_CIRCULAR) DINV_REQUIRE(ph <= d->height && pw <= d->width, "circular padding wider than the image");
        if (d->mode == PAD_REFLECT) DINV_REQUIRE(ph < d->height && pw < d->width, "reflect padding must be smaller than the image");
        fullH = d->height; fullW = d->width;
    }
    g->Ho = (fullH + d->stride - 1) / d->stride;
    g->Wo = (fullW + d->stride - 1) / d->stride;
    DINV_REQUIRE((int64_t)g->B * g->C <= 65535, "too many (batch*channel) planes per call");
    DINV_REQUIRE((size_t)g->h * g->w * sizeof(float) <= 64 * 1024, "filter too large for LDS");
    return 0;
}

// ------------------------------------------------------------------ 3-D (volumes [B,C,D,H,W]; conv3d / conv_transpose3d,
// convolution.py:333-452: the same padding rule per axis, stride 1)
struct ConvGeom3 {
    int32_t B, C, D, H, W;
    int32_t fb, fc, d, h, w;
    int32_t mode;
    int32_t pf, pt, pl, pk, pb, pr;   // pads front / top / left / back / bottom / right
    int32_t Do, Ho, Wo;
};

// y[b,c,ko,io,jo] = sum_{t,u,v} kf[t,u,v] * xpad[ko+t, io+u, jo+v],  kf = k flipped along all three axes
__global__ __launch_bounds__(256) void conv3d_pad_kernel(ConvGeom3 g, const float* __restrict__ x, const float* __restrict__ k,
                                                         float* __restrict__ y) {
    DINV_DYN_LDS(float, ks);
    const int nk = g.d * g.h * g.w;
    const int z = blockIdx.z, bc = z / g.Do, ko = z - bc * g.Do, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * nk;
    for (int i = threadIdx.x; i < nk; i += 256) ks[i] = kf[nk - 1 - i];
    __syncthreads();
    const int jo = blockIdx.x * 64 + (threadIdx.x & 63);
    const int io = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (io >= g.Ho || jo >= g.Wo) return;
    const float* vol = x + (int64_t)bc * g.D * g.H * g.W;
    float acc = 0.f;
    for (int t = 0; t < g.d; ++t) {
        const int dd = pad_map(ko + t - g.pf, g.D, g.mode);
        if (dd < 0) continue;
        for (int u = 0; u < g.h; ++u) {
            const int r = pad_map(io + u - g.pt, g.H, g.mode);
            if (r < 0) continue;
            const float* row = vol + ((int64_t)dd * g.H + r) * g.W;
            for (int v = 0; v < g.w; ++v) {
                const int cc = pad_map(jo + v - g.pl, g.W, g.mode);
                if (cc >= 0) acc = fmaf(ks[(t * g.h + u) * g.w + v], row[cc], acc);
            }
        }
    }
    y[(((int64_t)bc * g.Do + ko) * g.Ho + io) * g.Wo + jo] = acc;
}

// exact transpose as a gather over the pre-images of the output voxel under the padding map (fixed summation order)
__global__ __launch_bounds__(256) void conv3d_pad_transpose_kernel(ConvGeom3 g, const float* __restrict__ y,
                                                                   const float* __restrict__ k, float* __restrict__ x) {
    DINV_DYN_LDS(float, ks);
    const int nk = g.d * g.h * g.w;
    const int z = blockIdx.z, bc = z / g.D, dd = z - bc * g.D, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * nk;
    for (int i = threadIdx.x; i < nk; i += 256) ks[i] = kf[nk - 1 - i];
    __syncthreads();
    const int cc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= g.H || cc >= g.W) return;
    const float* meas = y + (int64_t)bc * g.Do * g.Ho * g.Wo;
    const Pre pdep = preimages(dd, g.D, g.pf, g.pk, g.mode);
    const Pre prow = preimages(r, g.H, g.pt, g.pb, g.mode);
    const Pre pcol = preimages(cc, g.W, g.pl, g.pr, g.mode);
    float acc = 0.f;
    for (int id = 0; id < pdep.n; ++id)
        for (int pd = pdep.lo[id]; pd <= pdep.hi[id]; ++pd)
            for (int t = 0; t < g.d; ++t) {
                const int ko = pd - t;
                if (ko < 0) break;
                if (ko >= g.Do) continue;
                for (int ir = 0; ir < prow.n; ++ir)
                    for (int pr = prow.lo[ir]; pr <= prow.hi[ir]; ++pr)
                        for (int u = 0; u < g.h; ++u) {
                            const int io = pr - u;
                            if (io < 0) break;
                            if (io >= g.Ho) continue;
                            const float* mrow = meas + ((int64_t)ko * g.Ho + io) * g.Wo;
                            for (int ic = 0; ic < pcol.n; ++ic)
                                for (int pc = pcol.lo[ic]; pc <= pcol.hi[ic]; ++pc)
                                    for (int v = 0; v < g.w; ++v) {
                                        const int jo = pc - v;
                                        if (jo < 0) break;
                                        if (jo < g.Wo) acc = fmaf(ks[(t * g.h + u) * g.w + v], mrow[jo], acc);
                                    }
                        }
            }
    x[(((int64_t)bc * g.D + dd) * g.H + r) * g.W + cc] = acc;
}

// filter gradient per (b, c) plane: one workgroup per (tap, plane)
__global__ __launch_bounds__(256) void conv3d_filter_grad_kernel(ConvGeom3 g, const float* __restrict__ x,
                                                                 const float* __restrict__ gy, float* __restrict__ dk) {
    __shared__ float red[4];
    const int nk = g.d * g.h * g.w;
    const int tap = blockIdx.x, t = tap / (g.h * g.w), uv = tap - t * g.h * g.w, u = uv / g.w, v = uv - u * g.w;
    const int bc = blockIdx.y;
    const float* vol = x + (int64_t)bc * g.D * g.H * g.W;
    const float* go = gy + (int64_t)bc * g.Do * g.Ho * g.Wo;
    float acc = 0.f;
    for (int ko = 0; ko < g.Do; ++ko) {
        const int dd = pad_map(ko + t - g.pf, g.D, g.mode);
        if (dd < 0) continue;
        for (int io = threadIdx.x >> 6; io < g.Ho; io += 4) {
            const int r = pad_map(io + u - g.pt, g.H, g.mode);
            if (r < 0) continue;
            const float* row = vol + ((int64_t)dd * g.H + r) * g.W;
            const float* grow = go + ((int64_t)ko * g.Ho + io) * g.Wo;
            for (int jo = threadIdx.x & 63; jo < g.Wo; jo += 64) {
                const int cc = pad_map(jo + v - g.pl, g.W, g.mode);
                if (cc >= 0) acc = fmaf(grow[jo], row[cc], acc);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dk[(int64_t)bc * nk + (nk - 1 - tap)] = (red[0] + red[1]) + (red[2] + red[3]);
}

int make_geom3(const dinv_conv3d_desc* d, ConvGeom3* g) {
    DINV_REQUIRE(d != nullptr, "null descriptor");
    DINV_REQUIRE(d->batch >= 0 && d->channels >= 1 && d->depth >= 1 && d->height >= 1 && d->width >= 1, "bad volume geometry");
    DINV_REQUIRE(d->fd >= 1 && d->fh >= 1 && d->fw >= 1 && (d->fbatch == 1 || d->fbatch == d->batch) &&
                 (d->fchannels == 1 || d->fchannels == d->channels), "filter shape not broadcastable");
    DINV_REQUIRE(d->mode >= PAD_VALID && d->mode <= PAD_CONSTANT, "unknown padding mode %d", d->mode);
    g->B = d->batch; g->C = d->channels; g->D = d->depth; g->H = d->height; g->W = d->width;
    g->fb = d->fbatch; g->fc = d->fchannels; g->d = d->fd; g->h = d->fh; g->w = d->fw;
    g->mode = d->mode;
    if (d->mode == PAD_VALID) {
        g->pf = g->pt = g->pl = g->pk = g->pb = g->pr = 0;
        DINV_REQUIRE(d->depth >= d->fd && d->height >= d->fh && d->width >= d->fw, "filter larger than volume in 'valid' mode");
        g->Do = d->depth - d->fd + 1; g->Ho = d->height - d->fh + 1; g->Wo = d->width - d->fw + 1;
    } else {
        const int pd = d->fd / 2, ph = d->fh / 2, pw = d->fw / 2;
        g->pf = pd - (d->fd - 1) % 2; g->pk = pd; g->pt = ph - (d->fh - 1) % 2; g->pb = ph; g->pl = pw - (d->fw - 1) % 2; g->pr = pw;
        if (d->mode == PAD_CIRCULAR) DINV_REQUIRE(pd <= d->depth && ph <= d->height && pw <= d->width, "circular padding wider than the volume");
        if (d->mode == PAD_REFLECT) DINV_REQUIRE(pd < d->depth && ph < d->height && pw < d->width, "reflect padding must be smaller than the volume");
        g->Do = d->depth; g->Ho = d->height; g->Wo = d->width;
    }
    DINV_REQUIRE((int64_t)g->B * g->C * std::max(g->D, g->Do) <= 65535, "too many (batch * channel * depth) planes per call");
    DINV_REQUIRE((size_t)g->d * g->h * g->w * sizeof(float) <= 64 * 1024, "filter too large for LDS");
    return 0;
}

// ------------------------------------------------------------------ real <-> half-complex row passes
struct RealRowsLoadIo {   // real [L, W] -> complex half spectrum [L, W/2+1] (first W/2+1 bins of the full FFT)
    const float* x;
    float2* out;
    int32_t wh;
    int64_t n_, q_;
    struct RowCtx { int64_t i, o; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const { return RowCtx{line * n_, line * wh}; }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const { return make_float2(x[c.i + n], 0.f); }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { if (k < wh) out[c.o + k] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

struct HalfRowsStoreRealIo {  // half spectrum [L, W/2+1] -> real [L, W]  (pocketfft c2r conventions)
    const float2* in;
    float* out;
    int32_t wh;
    int64_t n_, q_;
    struct RowCtx { int64_t i, o; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const { return RowCtx{line * wh, line * n_}; }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const {
        const int N = (int)n_;
        float2 v;
        if (n < wh) {
            v = in[c.i + n];
            if (n == 0 || (2 * n == N)) v.y = 0.f;  // imaginary parts of DC / Nyquist are ignored by c2r
        } else {
            v = in[c.i + (N - n)];
            v.y = -v.y;  // Hermitian extension
        }
        return v;
    }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { out[c.o + k] = v.x; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

}  // namespace

extern "C" int dinv_conv2d_out_size(const dinv_conv_desc* d, int32_t* ho, int32_t* wo) {
    ConvGeom g;
    if (int e = make_geom(d, &g)) return e;
    *ho = g.Ho; *wo = g.Wo;
    return 0;
}

extern "C" int dinv_conv2d(const dinv_conv_desc* d, const float* x, const float* filter, float* y, dinv_stream_t stream) {
    ConvGeom g;
    if (int e = make_geom(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && filter && y, "null pointer");
    hipLaunchKernelGGL(conv2d_pad_kernel, dim3((g.Wo + 63) / 64, (g.Ho + 3) / 4, g.B * g.C), dim3(256),
                       g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, x, filter, y);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv2d_transpose(const dinv_conv_desc* d, const float* y, const float* filter, float* x,
                                     dinv_stream_t stream) {
    ConvGeom g;
    if (int e = make_geom(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && filter && y, "null pointer");
    hipLaunchKernelGGL(conv2d_pad_transpose_kernel, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.B * g.C), dim3(256),
                       g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, y, filter, x);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv2d_filter_grad(const dinv_conv_desc* d, const float* x, const float* gy, float* dk_planes,
                                       dinv_stream_t stream) {
    ConvGeom g;
    if (int e = make_geom(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && gy && dk_planes, "null pointer");
    hipLaunchKernelGGL(conv2d_filter_grad_kernel, dim3(g.h * g.w, g.B * g.C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g,
                       x, gy, dk_planes);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3d_out_size(const dinv_conv3d_desc* d, int32_t* dout, int32_t* ho, int32_t* wo) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    *dout = g.Do; *ho = g.Ho; *wo = g.Wo;
    return 0;
}

extern "C" int dinv_conv3d(const dinv_conv3d_desc* d, const float* x, const float* filter, float* y, dinv_stream_t stream) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && filter && y, "null pointer");
    hipLaunchKernelGGL(conv3d_pad_kernel, dim3((g.Wo + 63) / 64, (g.Ho + 3) / 4, g.B * g.C * g.Do), dim3(256),
                       g.d * g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, x, filter, y);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3d_transpose(const dinv_conv3d_desc* d, const float* y, const float* filter, float* x, dinv_stream_t stream) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && filter && y, "null pointer");
    hipLaunchKernelGGL(conv3d_pad_transpose_kernel, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.B * g.C * g.D), dim3(256),
                       g.d * g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, y, filter, x);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3d_filter_grad(const dinv_conv3d_desc* d, const float* x, const float* gy, float* dk_planes,
                                       dinv_stream_t stream) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && gy && dk_planes, "null pointer");
    hipLaunchKernelGGL(conv3d_filter_grad_kernel, dim3(g.d * g.h * g.w, g.B * g.C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       g, x, gy, dk_planes);
    DINV_CHECK_LAUNCH();
    return 0;
}

// x real [P,H,W] -> half spectrum [P,H,W/2+1] (interleaved complex); unnormalised * scale
extern "C" int dinv_rfft2(const float* x, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
                          const dinv_fft_plan* plan_w, const void* table_w, float scale, dinv_stream_t stream) {
    DINV_REQUIRE(x && out && plan_h && plan_w && table_h && table_w, "null pointer");
    if (P == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int H = plan_h->n, W = plan_w->n, Wh = W / 2 + 1;
    RealRowsLoadIo rio{x, reinterpret_cast<float2*>(out), Wh, 0, 0};
    if (int e = launch_rows(rio, P * H, *plan_w, table_w, 0, 0, scale, s)) return e;
    C2CIo cio{reinterpret_cast<const float2*>(out), reinterpret_cast<float2*>(out), 0, 0};
    return launch_cols(cio, P, Wh, *plan_h, table_h, 0, 0, 1.0f, s);
}

// half spectrum [P,H,W/2+1] -> real [P,H,W]; `ws` holds P*H*(W/2+1) complex (input is left untouched)
extern "C" int dinv_irfft2(const float* in, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
                           const dinv_fft_plan* plan_w, const void* table_w, float scale, void* ws, size_t ws_bytes,
                           dinv_stream_t stream) {
    DINV_REQUIRE(in && out && plan_h && plan_w && table_h && table_w && ws, "null pointer");
    if (P == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int H = plan_h->n, W = plan_w->n, Wh = W / 2 + 1;
    DINV_REQUIRE(ws_bytes >= (size_t)P * H * Wh * sizeof(float2), "workspace too small");
    float2* t = reinterpret_cast<float2*>(ws);
    C2CIo cio{reinterpret_cast<const float2*>(in), t, 0, 0};
    if (int e = launch_cols(cio, P, Wh, *plan_h, table_h, 1, 0, 1.0f, s)) return e;
    HalfRowsStoreRealIo rio{t, out, Wh, 0, 0};
    return launch_rows(rio, P * H, *plan_w, table_w, 1, 0, scale, s);
}
