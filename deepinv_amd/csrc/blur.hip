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

constexpr int64_t kMaxGridZ = 65535;      // planes per launch (grid.y / grid.z limit)

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

// ---- LDS-tiled form for stride 1 (Blur.A with every padding mode; Blur.A_adjoint for valid / circular / constant, whose transpose
// is again a gather with an index map).  A workgroup owns 64 x 64 outputs of one plane: the (64 + h - 1) x (64 + w4) input patch
// goes through pad_map ONCE per element into LDS, a thread owns 4 x 4 outputs and walks the patch rows; per row and chunk of four
// filter columns it reads 8 consecutive patch values (two 16-byte LDS reads) and one 16-byte row of filter taps (the same address
// in every lane), and the taps of the previous three filter rows slide along in registers - the filter table carries three zero
// rows above and below and zero columns up to a multiple of 4, so the inner loop has no conditions: 64 multiply-adds per three
// LDS reads, no integer division, no global access.
struct TileConv {
    int32_t C, fb, fc;
    int32_t Hi, Wi, Ho, Wo;
    int32_t h, w;
    int32_t off_r, off_c;   // input coordinate of tap (u, v) for output (r, c): (r + u - off_r, c + v - off_c), then pad_map
    int32_t mode, flip;     // flip: taps are k[h w - 1 - (u w + v)] (true convolution), else k[u w + v] (its transpose)
    int32_t pw, pitch;      // patch columns, LDS row pitch (floats)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one patch row of one chunk of (up to) four filter columns: output row i of the thread (MASK bit i) multiplies it with filter row
// r - i, whose four taps are one 16-byte LDS read (the same address in every lane).  The 4 x 4 outputs are held as PAIRS of
// neighbouring columns, so that a multiply-add is one v_pk_fma_f32 on (tap, tap) x (patch[c], patch[c + 1]); the pairs that start
// at an odd patch column (taps 1 and 3) are re-packed from the even ones with six register moves.
// acc += taps[HALF] * v on both halves, the tap pair as SRC0 (broadcast by op_sel[0] / op_sel_hi[0]).  Written out as an instruction
// because the form hipcc picks for `acc += t.y * v` - the pair as src1, op_sel:[0,1,0] - is the one that returns wrong LOW results in
// lanes 48..63 of a wave while a wave of another kernel executes bf16 MFMAs on the same SIMD (scripts/r06/probe/pk_forms_probe.hip,
// DESIGN 3.6); src0-high for the low result is clean there.
template <int HALF>
__device__ __forceinline__ void pk_fma_tap(f32x2& acc, f32x2 taps, f32x2 v) {
#ifdef DINV_EMU
    acc += (HALF ? taps.y : taps.x) * v;
#else
    if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(taps), "v"(v));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(taps), "v"(v));
#endif
}

template <int MASK, int NV>
__device__ __forceinline__ void tile_step(f32x2 (&acc)[4][2], const float* __restrict__ prow, const float* __restrict__ krow, int w4) {
    const float4 a = *reinterpret_cast<const float4*>(prow);
    const float2 b = *reinterpret_cast<const float2*>(prow + 4);
    const float2 c = *reinterpret_cast<const float2*>(prow + 6);
    const f32x2 E0 = {a.x, a.y}, E1 = {a.z, a.w}, E2 = {b.x, b.y};
    const f32x2 O0 = {a.y, a.z}, O1 = {a.w, b.x}, O2 = {b.y, c.x};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!((MASK >> i) & 1)) continue;
        const float4 t = *reinterpret_cast<const float4*>(krow - i * w4);
        const f32x2 t01 = {t.x, t.y}, t23 = {t.z, t.w};
        if (NV > 0) { pk_fma_tap<0>(acc[i][0], t01, E0); pk_fma_tap<0>(acc[i][1], t01, E1); }
        if (NV > 1) { pk_fma_tap<1>(acc[i][0], t01, O0); pk_fma_tap<1>(acc[i][1], t01, O1); }
        if (NV > 2) { pk_fma_tap<0>(acc[i][0], t23, E1); pk_fma_tap<0>(acc[i][1], t23, E2); }
        if (NV > 3) { pk_fma_tap<1>(acc[i][0], t23, O1); pk_fma_tap<1>(acc[i][1], t23, O2); }
    }
}

template <int NV>
__device__ __forceinline__ void tile_chunk(f32x2 (&acc)[4][2], const float* __restrict__ prow, const float* __restrict__ krow, int pitch,
                                           int w4, int h) {
    if (h >= 4) {
        // filter row r - i exists for 0 <= r - i < h: the first and the last three patch rows serve fewer output rows
        tile_step<1, NV>(acc, prow, krow, w4);
        tile_step<3, NV>(acc, prow + pitch, krow + w4, w4);
        tile_step<7, NV>(acc, prow + 2 * pitch, krow + 2 * w4, w4);
        for (int r = 3; r < h; ++r) tile_step<15, NV>(acc, prow + r * pitch, krow + r * w4, w4);
        tile_step<14, NV>(acc, prow + h * pitch, krow + h * w4, w4);
        tile_step<12, NV>(acc, prow + (h + 1) * pitch, krow + (h + 1) * w4, w4);
        tile_step<8, NV>(acc, prow + (h + 2) * pitch, krow + (h + 2) * w4, w4);
    } else {       // short filters: every step with all four rows (the table's zero rows above and below make the surplus vanish)
        for (int r = 0; r < h + 3; ++r) tile_step<15, NV>(acc, prow + r * pitch, krow + r * w4, w4);
    }
}

__global__ __launch_bounds__(256) void conv2d_tiled_kernel(TileConv g, const float* __restrict__ in, const float* __restrict__ k,
                                                           float* __restrict__ out) {
    DINV_DYN_LDS(float, smem);
    const int w4 = (g.w + 3) & ~3, ph = 64 + g.h - 1;
    float* kt = smem;                                   // [(h + 6)][w4]: rows -3 .. h + 2 of the (flipped) filter, zero outside
    float* patch = smem + (g.h + 6) * w4;               // [ph][pitch]
    const int bc = blockIdx.z, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * g.h * g.w;
    const int tid = threadIdx.x;
    for (int i = tid; i < (g.h + 6) * w4; i += 256) {
        const int u = i / w4 - 3, v = i - (u + 3) * w4;
        float t = 0.f;
        if (u >= 0 && u < g.h && v < g.w) t = g.flip ? kf[g.h * g.w - 1 - (u * g.w + v)] : kf[u * g.w + v];
        kt[i] = t;
    }
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* img = in + (int64_t)bc * g.Hi * g.Wi;
    for (int pr = tid / 64; pr < ph; pr += 4) {
        const int rr = pad_map(r0 + pr - g.off_r, g.Hi, g.mode);
        const bool rok = rr >= 0 && rr < g.Hi;
        for (int pc = tid & 63; pc < g.pw; pc += 64) {
            const int cc = pad_map(c0 + pc - g.off_c, g.Wi, g.mode);
            patch[pr * g.pitch + pc] = (rok && cc >= 0 && cc < g.Wi) ? img[(int64_t)rr * g.Wi + cc] : 0.f;
        }
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    f32x2 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
    const float* pbase = patch + (ty * 4) * g.pitch + tx * 4;
    const float* kbase = kt + 3 * w4;
    const int nfull = g.w >> 2, rem = g.w & 3;
    for (int q = 0; q < nfull; ++q) tile_chunk<4>(acc, pbase + 4 * q, kbase + 4 * q, g.pitch, w4, g.h);
    if (rem == 1) tile_chunk<1>(acc, pbase + 4 * nfull, kbase + 4 * nfull, g.pitch, w4, g.h);
    else if (rem == 2) tile_chunk<2>(acc, pbase + 4 * nfull, kbase + 4 * nfull, g.pitch, w4, g.h);
    else if (rem == 3) tile_chunk<3>(acc, pbase + 4 * nfull, kbase + 4 * nfull, g.pitch, w4, g.h);
    float* o = out + (int64_t)bc * g.Ho * g.Wo;
    const int oc = c0 + tx * 4;
    const bool vec = (g.Wo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (((int64_t)g.Ho * g.Wo) % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int orow = r0 + ty * 4 + i;
        if (orow >= g.Ho) continue;
        float* dst = o + (int64_t)orow * g.Wo + oc;
        const float v4[4] = {acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y};
        if (vec && oc + 3 < g.Wo) {
            *reinterpret_cast<float4*>(dst) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (oc + j < g.Wo) dst[j] = v4[j];
        }
    }
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

// the LDS-tiled kernel for a stride-1 geometry; returns 1 when it took the call, 0 when the caller's general kernel must
inline int launch_tiled(const ConvGeom& g, bool transpose, const float* in, const float* k, float* out, hipStream_t s, int* took) {
    *took = 0;
    if (g.stride != 1) return 0;
    if (transpose && !(g.mode == PAD_VALID || g.mode == PAD_CIRCULAR || g.mode == PAD_CONSTANT)) return 0;
    TileConv t;
    t.C = g.C; t.fb = g.fb; t.fc = g.fc; t.h = g.h; t.w = g.w;
    if (!transpose) {
        t.Hi = g.H; t.Wi = g.W; t.Ho = g.Ho; t.Wo = g.Wo; t.off_r = g.pt; t.off_c = g.pl; t.mode = g.mode; t.flip = 1;
    } else {
        // x[r, c] = sum_{u, v} kf[u, v] y[r + pt - u, c + pl - v]: with u' = h - 1 - u a gather with the UNFLIPPED filter
        // (zero outside y for valid / constant, periodic for circular)
        t.Hi = g.Ho; t.Wi = g.Wo; t.Ho = g.H; t.Wo = g.W; t.off_r = g.h - 1 - g.pt; t.off_c = g.w - 1 - g.pl;
        t.mode = g.mode == PAD_CIRCULAR ? PAD_CIRCULAR : PAD_CONSTANT; t.flip = 0;
    }
    const int w4 = (g.w + 3) & ~3;
    t.pw = 64 + w4;
    t.pitch = t.pw;
    const size_t lds = ((size_t)(g.h + 6) * w4 + (size_t)(64 + g.h - 1) * t.pitch) * sizeof(float);
    if (lds > 64 * 1024) return 0;
    hipLaunchKernelGGL(conv2d_tiled_kernel, dim3((t.Wo + 63) / 64, (t.Ho + 63) / 64, g.B * g.C), dim3(256), lds, s, t, in, k, out);
    DINV_CHECK_LAUNCH();
    *took = 1;
    return 0;
}

// ---- LDS-tiled forms for stride s >= 2 (Downsampling.A / A_adjoint: bicubic x4 on 256 x 256 is a 16 x 16 filter at stride 4).  The
// general kernels above read every tap from global memory through pad_map (57 us) or walk pre-image lists with integer divisions and
// spill (91 us) for 12.6 MB of input: ten times what the bytes cost.  Forward: a workgroup owns 16 x 16 outputs; its
// (15 s + h) x (15 s + w) input patch goes through pad_map once per element into LDS, stored DE-INTERLEAVED by column phase
// (column j s + b at [b][j]) so that the 16 lanes of an output row read consecutive words for a tap; two LDS reads per
// multiply-add, no global access and no division in the loop.  Transposed (valid / constant, circular with sizes divisible by
// the stride): x[r, c] = sum over the taps u = (r + pt) mod s + t s, v = (c + pl) mod s + q s of kf[u, v] y[(r + pt) / s - t,
// (c + pl) / s - q]: a workgroup owns 64 x 64 outputs and stages the (64 / s + (h - 1) / s + 2)^2 measurements they touch.
struct StrideConv {
    int32_t C, fb, fc;
    int32_t Hi, Wi, Ho, Wo;     // forward: input / output;  transposed: Hi x Wi = the measurement y, Ho x Wo = the image x
    int32_t h, w, s, pt, pl, mode;
    int32_t PH, PWs, pitch;     // forward: patch rows, columns per phase, row pitch;  transposed: PH x PWs staged measurements, pitch
};

// FAST4 (stride 4, filter width a multiple of 4 - bicubic x4): the patch keeps its natural column order; the four taps v .. v + 3 of
// lane tx are the 16 bytes at column 4 tx + v, so the 16 lanes of an output row read 256 contiguous bytes with ONE ds_read_b128 and the
// four taps come with one more (the same address in every lane): 2 LDS reads per 4 multiply-adds, 64 trips per output instead of 256
// dependent ones (29.6 -> us measured in the launcher's comment).
template <bool FAST4>
__global__ __launch_bounds__(256) void conv2d_strided_kernel(StrideConv g, const float* __restrict__ in, const float* __restrict__ k,
                                                             float* __restrict__ out) {
    DINV_DYN_LDS(float, smem);
    float* ks = smem;                      // flipped filter [h][w]
    float* patch = smem + g.h * g.w;       // [PH][pitch]; general form: a row = s phases of PWs columns (column j s + b at [b][j])
    const int bc = blockIdx.z, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * g.h * g.w;
    const int tid = threadIdx.x;
    for (int i = tid; i < g.h * g.w; i += 256) ks[i] = kf[g.h * g.w - 1 - i];
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    const float* img = in + (int64_t)bc * g.Hi * g.Wi;
    const int pwide = g.s * g.PWs;
    // four elements per trip: their global loads are in flight together
    for (int idx0 = tid; idx0 < g.PH * pwide; idx0 += 1024) {
        float val[4];
        int dst[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = idx0 + 256 * e;
            const int pr = idx / pwide, pc = idx - pr * pwide;
            const int rr = pad_map(r0 * g.s + pr - g.pt, g.Hi, g.mode);
            const int cc = pad_map(c0 * g.s + pc - g.pl, g.Wi, g.mode);
            const int j = pc / g.s, ph = pc - j * g.s;
            const bool ok = idx < g.PH * pwide && rr >= 0 && rr < g.Hi && cc >= 0 && cc < g.Wi;
            dst[e] = idx < g.PH * pwide ? pr * g.pitch + (FAST4 ? pc : ph * g.PWs + j) : -1;
            val[e] = ok ? img[(int64_t)rr * g.Wi + cc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (dst[e] >= 0) patch[dst[e]] = val[e];
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    float acc = 0.f;
    if (FAST4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int u = 0; u < g.h; ++u) {
            const float* prow = patch + (ty * 4 + u) * g.pitch + tx * 4;
            const float* krow = ks + u * g.w;
#pragma unroll 4
            for (int v = 0; v < g.w; v += 4) {
                const float4 p4 = *reinterpret_cast<const float4*>(prow + v);
                const float4 t4 = *reinterpret_cast<const float4*>(krow + v);
                a0 = fmaf(t4.x, p4.x, a0); a1 = fmaf(t4.y, p4.y, a1); a2 = fmaf(t4.z, p4.z, a2); a3 = fmaf(t4.w, p4.w, a3);
            }
        }
        acc = (a0 + a1) + (a2 + a3);
    } else {
        for (int u = 0; u < g.h; ++u) {
            const float* prow = patch + (ty * g.s + u) * g.pitch + tx;
            const float* krow = ks + u * g.w;
            for (int ph = 0; ph < g.s; ++ph) {
                const float* pp = prow + ph * g.PWs;
                for (int v = ph, q = 0; v < g.w; v += g.s, ++q) acc = fmaf(krow[v], pp[q], acc);
            }
        }
    }
    const int io = r0 + ty, jo = c0 + tx;
    if (io < g.Ho && jo < g.Wo) out[((int64_t)bc * g.Ho + io) * g.Wo + jo] = acc;
}

// FAST16 (stride 4, 16 x 16 filter - bicubic x4): the 16 outputs of a thread (rows r0 + ty + 4 i) share their tap phases, so the 4 x 4
// taps they use are read ONCE into registers, and measurement row R serves the outputs i = R - R0 + t (t = 0 .. 3): the thread walks the
// 19 rows it touches, four values each - 76 + 16 LDS reads for 256 multiply-adds instead of 512.
template <bool FAST16>
__global__ __launch_bounds__(256) void conv2d_strided_transpose_kernel(StrideConv g, const float* __restrict__ y,
                                                                       const float* __restrict__ k, float* __restrict__ x) {
    DINV_DYN_LDS(float, smem);
    float* ks = smem;                      // flipped filter [h][w]
    float* yp = smem + g.h * g.w;          // [PH][pitch] measurements
    const int bc = blockIdx.z, b = bc / g.C, c = bc % g.C;
    const float* kf = k + ((int64_t)(g.fb > 1 ? b : 0) * g.fc + (g.fc > 1 ? c : 0)) * g.h * g.w;
    const int tid = threadIdx.x;
    for (int i = tid; i < g.h * g.w; i += 256) ks[i] = kf[g.h * g.w - 1 - i];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int base_r = (r0 + g.pt) / g.s - (g.h - 1) / g.s, base_c = (c0 + g.pl) / g.s - (g.w - 1) / g.s;
    const float* meas = y + (int64_t)bc * g.Hi * g.Wi;
    for (int idx = tid; idx < g.PH * g.PWs; idx += 256) {
        const int i = idx / g.PWs, j = idx - i * g.PWs;
        int rr = base_r + i, cc = base_c + j;
        if (g.mode == PAD_CIRCULAR) {
            rr %= g.Hi; if (rr < 0) rr += g.Hi;
            cc %= g.Wi; if (cc < 0) cc += g.Wi;
        }
        yp[i * g.pitch + j] = (rr >= 0 && rr < g.Hi && cc >= 0 && cc < g.Wi) ? meas[(int64_t)rr * g.Wi + cc] : 0.f;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    const int cx = c0 + tx;
    const int C0 = (cx + g.pl) / g.s, bph = (cx + g.pl) - C0 * g.s;
    const float* ycol = yp + (C0 - base_c);
    if (FAST16) {
        const int Rb = (r0 + ty + g.pt) >> 2, a = (r0 + ty + g.pt) & 3;
        float T[4][4], acc[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[t][q] = ks[(a + 4 * t) * 16 + bph + 4 * q];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int rel = -3; rel < 16; ++rel) {
            const float* yrow = ycol + (Rb + rel - base_r) * g.pitch;
            const float y0 = yrow[0], y1 = yrow[-1], y2 = yrow[-2], y3 = yrow[-3];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = rel + t;
                if (i >= 0 && i < 16) acc[i] += (T[t][0] * y0 + T[t][1] * y1) + (T[t][2] * y2 + T[t][3] * y3);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + ty + 4 * i;
            if (r < g.Ho && cx < g.Wo) x[((int64_t)bc * g.Ho + r) * g.Wo + cx] = acc[i];
        }
        return;
    }
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i;
        const int R0 = (r + g.pt) / g.s, a = (r + g.pt) - R0 * g.s;
        float acc = 0.f;
        for (int u = a, t = 0; u < g.h; u += g.s, ++t) {
            const float* yrow = ycol + (R0 - t - base_r) * g.pitch;
            const float* krow = ks + u * g.w;
#pragma unroll 4
            for (int v = bph, q = 0; v < g.w; v += g.s, ++q) acc = fmaf(krow[v], yrow[-q], acc);
        }
        if (r < g.Ho && cx < g.Wo) x[((int64_t)bc * g.Ho + r) * g.Wo + cx] = acc;
    }
}

// the strided LDS-tiled kernels; *took = 1 when they ran the call
inline int launch_strided(const ConvGeom& g, bool transpose, const float* in, const float* k, float* out, hipStream_t s, int* took) {
    *took = 0;
    if (g.stride < 2 || g.stride > 16) return 0;
    StrideConv t;
    t.C = g.C; t.fb = g.fb; t.fc = g.fc; t.h = g.h; t.w = g.w; t.s = g.stride; t.pt = g.pt; t.pl = g.pl; t.mode = g.mode;
    if (!transpose) {
        t.Hi = g.H; t.Wi = g.W; t.Ho = g.Ho; t.Wo = g.Wo;
        t.PH = 15 * g.stride + g.h;
        t.PWs = 15 + (g.w + g.stride - 1) / g.stride;
        t.pitch = g.stride * t.PWs;
        const bool fast4 = g.stride == 4 && g.w % 4 == 0;
        if (fast4) t.pitch = (t.pitch + 3) & ~3;            // 16-byte rows
        else while (t.pitch % 8 != 4) ++t.pitch;            // rows s apart land 16 banks apart at stride 4
        const size_t lds = ((size_t)g.h * g.w + (size_t)t.PH * t.pitch) * sizeof(float);
        if (lds > 64 * 1024) return 0;
        const dim3 grid((g.Wo + 15) / 16, (g.Ho + 15) / 16, g.B * g.C);
        if (fast4) hipLaunchKernelGGL(conv2d_strided_kernel<true>, grid, dim3(256), lds, s, t, in, k, out);
        else hipLaunchKernelGGL(conv2d_strided_kernel<false>, grid, dim3(256), lds, s, t, in, k, out);
    } else {
        if (!(g.mode == PAD_VALID || g.mode == PAD_CONSTANT || (g.mode == PAD_CIRCULAR && g.H % g.stride == 0 && g.W % g.stride == 0)))
            return 0;
        t.Hi = g.Ho; t.Wi = g.Wo; t.Ho = g.H; t.Wo = g.W;
        t.PH = 63 / g.stride + (g.h - 1) / g.stride + 2;
        t.PWs = 63 / g.stride + (g.w - 1) / g.stride + 2;
        t.pitch = t.PWs | 1;
        const size_t lds = ((size_t)g.h * g.w + (size_t)t.PH * t.pitch) * sizeof(float);
        if (lds > 64 * 1024) return 0;
        const dim3 grid((g.W + 63) / 64, (g.H + 63) / 64, g.B * g.C);
        if (g.stride == 4 && g.h == 16 && g.w == 16) hipLaunchKernelGGL(conv2d_strided_transpose_kernel<true>, grid, dim3(256), lds, s, t, in, k, out);
        else hipLaunchKernelGGL(conv2d_strided_transpose_kernel<false>, grid, dim3(256), lds, s, t, in, k, out);
    }
    DINV_CHECK_LAUNCH();
    *took = 1;
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
    int32_t z0;                       // first (batch, channel, depth) plane of this launch: grids are cut at 65535 planes (grid.z)
};

// y[b,c,ko,io,jo] = sum_{t,u,v} kf[t,u,v] * xpad[ko+t, io+u, jo+v],  kf = k flipped along all three axes
__global__ __launch_bounds__(256) void conv3d_pad_kernel(ConvGeom3 g, const float* __restrict__ x, const float* __restrict__ k,
                                                         float* __restrict__ y) {
    DINV_DYN_LDS(float, ks);
    const int nk = g.d * g.h * g.w;
    const int z = blockIdx.z + g.z0, bc = z / g.Do, ko = z - bc * g.Do, b = bc / g.C, c = bc % g.C;
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
    const int z = blockIdx.z + g.z0, bc = z / g.D, dd = z - bc * g.D, b = bc / g.C, c = bc % g.C;
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
    const int bc = blockIdx.y + g.z0;
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
    DINV_REQUIRE((int64_t)g->B * g->C * std::max(g->D, g->Do) < (1ll << 31), "too many (batch * channel * depth) planes per call");
    g->z0 = 0;
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


// ------------------------------------------------------------------ BlurFFT: the spectral symbol and the fused operator
// DecomposablePhysics of BlurFFT (deepinv/physics/blur.py:639-657, forward.py:1080-1117, 1212-1252): every operator of the class is
//     out = irfft2( SYMBOL( rfft2(x) ) ),     SYMBOL(v) = post( scale( pre(v) ) )   per frequency bin,
//   pre   = v * conj(angle)   (U_adjoint, blur.py:650-653)                     - A_adjoint, A_A_adjoint, A_dagger
//   scale = m (.) v per real / imaginary component (`mask * view_as_real(.)`)  - A, A_adjoint          (mode 1)
//           (m m) (.) v        (`mask.conj() * mask * .`)                      - A_adjoint_A, A_A_adjoint   (mode 2)
//           v / (m m + add)    (forward.py:1223-1234)                          - prox_l2, add = 1 / gamma   (mode 3)
//           v * (m > 1e-5 ? 1 / m : 0)   (forward.py:1247-1252)                - A_dagger                   (mode 4)
//   post  = v * angle         (U, blur.py:645-648)                             - A, A_A_adjoint
// `mask` is the reference's buffer [Ps, H, Wh, 2] (two real values per bin), `angle` complex64 [Ps, H, Wh]; spectrum plane p uses
// symbol plane p % Ps (Ps = C for a filter shared by the batch, B C for per-sample filters).
struct Symbol {
    const float2* mask;
    const float2* angle;
    int64_t planes, plane_bins;     // Ps, H * Wh
    int32_t wh;
    int32_t pre_conj, scale_mode, post;
    float add;
    __device__ __forceinline__ float2 apply(float2 v, int64_t p, int k, int q) const {
        const int64_t i = (p % planes) * plane_bins + (int64_t)k * wh + q;
        float2 a = make_float2(1.f, 0.f);
        if (pre_conj | post) a = angle[i];
        if (pre_conj) v = cmulc(v, a);
        if (scale_mode) {
            const float2 m = mask[i];
            switch (scale_mode) {
                case 1: v.x *= m.x; v.y *= m.y; break;
                case 2: v.x *= m.x * m.x; v.y *= m.y * m.y; break;
                case 3: {
                    float sx = m.x * m.x, sy = m.y * m.y;     // m m, then + add: two roundings, as the reference's tensor expression
#ifndef DINV_EMU
                    asm volatile("" : "+v"(sx), "+v"(sy));
#endif
                    v.x = v.x / (sx + add); v.y = v.y / (sy + add);
                    break;
                }
                default:
                    v.x *= m.x > 1e-5f ? 1.0f / m.x : 0.0f;
                    v.y *= m.y > 1e-5f ? 1.0f / m.y : 0.0f;
            }
        }
        if (post) v = cmul(v, a);
        return v;
    }
};

// in-place symbol multiply of a contiguous half spectrum [P, H, Wh] (general sizes; U / U_adjoint alone)
__global__ __launch_bounds__(256) void spectrum_symbol_kernel(Symbol sym, const float2* __restrict__ in, float2* __restrict__ out,
                                                              int64_t n, int64_t pitch) {
    const int64_t per_plane = sym.plane_bins / sym.wh * pitch;      // H * pitch
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / per_plane, r = i - p * per_plane;
        const int k = (int)(r / pitch), q = (int)(r - (int64_t)k * pitch);
        if (q < sym.wh) out[i] = sym.apply(in[i], p, k, q);
    }
}

// rows passes on the PITCHED intermediate t [P, H, pitch] (pitch = Wh rounded up to 16 complex: every 16-column strip of the
// column pass is one aligned 128-byte segment per row; the pad columns are never read as data)
template <bool V4>
struct RealRowsPitchedIo {   // real [L, W] -> t rows
    const float* x;
    float2* out;
    int32_t wh, pitch;
    int64_t n_, q_;
    struct RowCtx { int64_t i, o; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const { return RowCtx{line * n_, line * pitch}; }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const { return make_float2(x[c.i + n], 0.f); }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { if (k < wh) out[c.o + k] = v; }
    static constexpr bool has_vec4 = V4;
    __device__ __forceinline__ void load4(const RowCtx& c, int n0, float2 (&v)[4]) const {
        const float4 a = ld_f4(x + c.i + n0);
        v[0] = make_float2(a.x, 0.f); v[1] = make_float2(a.y, 0.f); v[2] = make_float2(a.z, 0.f); v[3] = make_float2(a.w, 0.f);
    }
    __device__ __forceinline__ void store4(const RowCtx& c, int k0, const float2 (&v)[4]) const {
        if (k0 < wh) st_c4(out + c.o + k0, v);      // a group that straddles Wh spills into the pad columns (pitch >= Wh + 3)
    }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

template <bool V4>
struct HalfRowsPitchedIo {  // t rows -> real [L, W]  (pocketfft c2r conventions: imaginary parts of DC / Nyquist ignored)
    const float2* in;
    float* out;
    int32_t wh, pitch;
    int64_t n_, q_;
    struct RowCtx { int64_t i, o; };
    struct ColCtx {};
    __device__ __forceinline__ RowCtx row_ctx(int64_t line) const { return RowCtx{line * pitch, line * n_}; }
    __device__ __forceinline__ float2 load(const RowCtx& c, int n) const {
        const int N = (int)n_;
        float2 v;
        if (n < wh) {
            v = in[c.i + n];
            if (n == 0 || (2 * n == N)) v.y = 0.f;
        } else {
            v = in[c.i + (N - n)];
            v.y = -v.y;
        }
        return v;
    }
    __device__ __forceinline__ void store(const RowCtx& c, int k, float2 v) const { out[c.o + k] = v.x; }
    static constexpr bool has_vec4 = V4;
    __device__ __forceinline__ void load4(const RowCtx& c, int n0, float2 (&v)[4]) const {
        if (n0 > 0 && n0 + 3 < wh && 2 * (n0 + 3) < (int)n_) {
            ld_c4(in + c.i + n0, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = load(c, n0 + e);
        }
    }
    __device__ __forceinline__ void store4(const RowCtx& c, int k0, const float2 (&v)[4]) const {
        st_f4(out + c.o + k0, make_float4(v[0].x, v[1].x, v[2].x, v[3].x));
    }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

struct PitchedColsIo {      // column pass over the first Q columns of t [P, N, pitch]
    const float2* in;
    float2* out;
    int64_t pitch;
    int64_t n_, q_;
    struct RowCtx {};
    struct ColCtx { int64_t base; };
    __device__ __forceinline__ ColCtx col_ctx(int64_t p, int64_t q) const { return ColCtx{p * n_ * pitch + q}; }
    __device__ __forceinline__ float2 load(const ColCtx& c, int k) const { return in[c.base + (int64_t)k * pitch]; }
    __device__ __forceinline__ void store(const ColCtx& c, int k, float2 v) const { out[c.base + (int64_t)k * pitch] = v; }
    __host__ void set_geometry(int64_t n, int64_t q) { n_ = n; q_ = q; }
};

// The middle pass of the fused operator: t <- F_H^-1( SYMBOL( F_H t ) ) on strips of L columns - the forward column transform
// leaves its outputs, symbol applied, in a second LDS tile in natural order; the inverse transform reads them from there.  The
// spectrum crosses HBM once in each direction instead of three times (forward columns, symbol, inverse columns).
template <class P, int L, int NT>
__global__ __launch_bounds__(NT, (NT == 512 && P::N <= 256) ? 4 : 2) void blurfft_cols_kernel(float2* __restrict__ t, int64_t pitch, int64_t Q, int64_t qtiles,
                                                           int64_t ntiles, const void* table, Symbol sym) {
    using TF = TileFft<P, false, false, L, NT>;
    using TI = TileFft<P, true, false, L, NT>;
    constexpr int N = P::N;
    __shared__ __attribute__((aligned(16))) float2 buf[TF::lds_floats2];
    __shared__ __attribute__((aligned(16))) float2 ksp[(size_t)L * N];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int line = tid % L;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p = tile / qtiles;
        const int64_t q0 = (tile - p * qtiles) * L;
        const int cols = (int)min((int64_t)L, Q - q0);
        const int q = (int)q0 + (line < cols ? line : 0);
        float2* col = t + p * (int64_t)N * pitch + q;
        __syncthreads();   // the previous tile's inverse transform has left `buf` and `ksp`
        TF::run(buf, tw, cols, 0, 1.0f, tid,
                [&](int, int, int, int n) { return col[(int64_t)n * pitch]; },
                [&](int, int ln, int k, int, float2 v) { ksp[k * L + ln] = sym.apply(v, p, k, q); });
        __syncthreads();
        TI::run(buf, tw, cols, 0, 1.0f, tid,
                [&](int, int, int ln, int n) { return ksp[n * L + ln]; },
                [&](int, int, int k, int, float2 v) { col[(int64_t)k * pitch] = v; });
    }
}

template <int N> struct BlurColsL { static constexpr int value = N >= 512 ? 8 : 16; };

template <int N>
int launch_blurfft_cols(float2* t, int64_t pitch, int64_t P_, int64_t Q, const void* table, const Symbol& sym, hipStream_t s) {
    using P = typename PlanFor<N>::P;
    constexpr int L = BlurColsL<N>::value;
    const int64_t qtiles = ceil_div(Q, L), ntiles = P_ * qtiles;
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, 4 * kMaxGrid);
    // 512 threads from 256 points up: half the register arrays per thread (202 -> ~110 registers at N = 256), twice the waves per
    // compute unit behind the same two LDS tiles
    constexpr int NT = N >= 256 ? 512 : 256;
    hipLaunchKernelGGL((blurfft_cols_kernel<P, L, NT>), dim3(grid), dim3(NT), 0, s, t, pitch, Q, qtiles, ntiles, table, sym);
    DINV_CHECK_LAUNCH();
    return 0;
}

int make_symbol(Symbol* sym, const float* mask, const float* angle, int64_t symbol_planes, int H, int Wh, int32_t flags, float add) {
    const int scale_mode = (flags >> 4) & 7;
    DINV_REQUIRE(symbol_planes >= 1, "symbol_planes must be >= 1");
    DINV_REQUIRE(scale_mode <= 4, "unknown scale mode %d", scale_mode);
    DINV_REQUIRE(scale_mode == 0 || mask, "this symbol needs the singular values (mask)");
    DINV_REQUIRE(!(flags & 3) || angle, "this symbol needs the phases (angle)");
    *sym = Symbol{reinterpret_cast<const float2*>(mask), reinterpret_cast<const float2*>(angle), symbol_planes, (int64_t)H * Wh, Wh,
                  (flags & 1) ? 1 : 0, scale_mode, (flags & 2) ? 1 : 0, add};
    return 0;
}

}  // namespace

extern "C" size_t dinv_blurfft_workspace_bytes(int64_t P, int32_t H, int32_t W) {
    const int64_t pitch = ceil_div(W / 2 + 1, 16) * 16;
    return (size_t)(P * H * pitch) * sizeof(float2);
}

extern "C" int dinv_spectrum_symbol(const float* spec_in, float* spec_out, int64_t P, int32_t H, int32_t Wh, const float* mask,
                                    const float* angle, int64_t symbol_planes, int32_t flags, float add, dinv_stream_t stream) {
    DINV_REQUIRE(spec_in && spec_out && P >= 0 && H >= 1 && Wh >= 1, "bad arguments");
    Symbol sym;
    if (int e = make_symbol(&sym, mask, angle, symbol_planes, H, Wh, flags, add)) return e;
    const int64_t n = P * H * (int64_t)Wh;
    if (n == 0) return 0;
    hipLaunchKernelGGL(spectrum_symbol_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), sym, reinterpret_cast<const float2*>(spec_in),
                       reinterpret_cast<float2*>(spec_out), n, (int64_t)Wh);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_blurfft_apply(const float* x, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
                                  const dinv_fft_plan* plan_w, const void* table_w, const float* mask, const float* angle,
                                  int64_t symbol_planes, int32_t flags, float add, float scale, void* ws, size_t ws_bytes,
                                  dinv_stream_t stream) {
    DINV_REQUIRE(x && out && plan_h && plan_w && table_h && table_w && ws, "null pointer");
    if (P == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int H = plan_h->n, W = plan_w->n, Wh = W / 2 + 1;
    const int64_t pitch = ceil_div(Wh, 16) * 16;
    DINV_REQUIRE(ws_bytes >= dinv_blurfft_workspace_bytes(P, H, W), "workspace too small");
    DINV_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)ws) % 16 == 0, "tensors must be 16-byte aligned");
    Symbol sym;
    if (int e = make_symbol(&sym, mask, angle, symbol_planes, H, Wh, flags, add)) return e;
    float2* t = reinterpret_cast<float2*>(ws);
    // scale of the transform pair (ortho: 1 / (H W)) applied once, in the first pass
    if (W % 4 == 0) {
        RealRowsPitchedIo<true> rio{x, t, Wh, (int32_t)pitch, 0, 0};
        if (int e = launch_rows(rio, P * H, *plan_w, table_w, 0, 0, scale, s)) return e;
    } else {       // 4-element groups need W % 4 == 0: scalar accesses otherwise
        RealRowsPitchedIo<false> rio{x, t, Wh, (int32_t)pitch, 0, 0};
        if (int e = launch_rows(rio, P * H, *plan_w, table_w, 0, 0, scale, s)) return e;
    }
    bool fused = false;
    switch (H) {
#define DINV_CASE(NN) case NN: fused = true; if (int e = launch_blurfft_cols<NN>(t, pitch, P, Wh, table_h, sym, s)) return e; break;
        DINV_CASE(64) DINV_CASE(128) DINV_CASE(256) DINV_CASE(320) DINV_CASE(512)
#undef DINV_CASE
        default: break;
    }
    if (!fused) {      // other heights: forward columns, symbol, inverse columns as three passes of the generic engine
        PitchedColsIo cio{t, t, pitch, 0, 0};
        if (int e = launch_cols(cio, P, Wh, *plan_h, table_h, 0, 0, 1.0f, s)) return e;
        const int64_t n = P * H * pitch;
        hipLaunchKernelGGL(spectrum_symbol_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, sym, t,
                           t, n, pitch);
        DINV_CHECK_LAUNCH();
        if (int e = launch_cols(cio, P, Wh, *plan_h, table_h, 1, 0, 1.0f, s)) return e;
    }
    if (W % 4 == 0) {
        HalfRowsPitchedIo<true> oio{t, out, Wh, (int32_t)pitch, 0, 0};
        return launch_rows(oio, P * H, *plan_w, table_w, 1, 0, 1.0f, s);
    }
    HalfRowsPitchedIo<false> oio{t, out, Wh, (int32_t)pitch, 0, 0};
    return launch_rows(oio, P * H, *plan_w, table_w, 1, 0, 1.0f, s);
}

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
    int took = 0;
    if (int e = launch_tiled(g, false, x, filter, y, reinterpret_cast<hipStream_t>(stream), &took)) return e;
    if (took) return 0;
    if (int e = launch_strided(g, false, x, filter, y, reinterpret_cast<hipStream_t>(stream), &took)) return e;
    if (took) return 0;
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
    int took = 0;
    if (int e = launch_tiled(g, true, y, filter, x, reinterpret_cast<hipStream_t>(stream), &took)) return e;
    if (took) return 0;
    if (int e = launch_strided(g, true, y, filter, x, reinterpret_cast<hipStream_t>(stream), &took)) return e;
    if (took) return 0;
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
    for (int64_t z0 = 0, nz = (int64_t)g.B * g.C * g.Do; z0 < nz; z0 += kMaxGridZ) {      // grid.z holds 65535 planes per launch
        g.z0 = (int32_t)z0;
        hipLaunchKernelGGL(conv3d_pad_kernel, dim3((g.Wo + 63) / 64, (g.Ho + 3) / 4, (unsigned)std::min<int64_t>(nz - z0, kMaxGridZ)), dim3(256),
                           g.d * g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, x, filter, y);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3d_transpose(const dinv_conv3d_desc* d, const float* y, const float* filter, float* x, dinv_stream_t stream) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && filter && y, "null pointer");
    for (int64_t z0 = 0, nz = (int64_t)g.B * g.C * g.D; z0 < nz; z0 += kMaxGridZ) {
        g.z0 = (int32_t)z0;
        hipLaunchKernelGGL(conv3d_pad_transpose_kernel, dim3((g.W + 63) / 64, (g.H + 3) / 4, (unsigned)std::min<int64_t>(nz - z0, kMaxGridZ)),
                           dim3(256), g.d * g.h * g.w * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, y, filter, x);
    }
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_conv3d_filter_grad(const dinv_conv3d_desc* d, const float* x, const float* gy, float* dk_planes,
                                       dinv_stream_t stream) {
    ConvGeom3 g;
    if (int e = make_geom3(d, &g)) return e;
    if (g.B == 0) return 0;
    DINV_REQUIRE(x && gy && dk_planes, "null pointer");
    for (int64_t z0 = 0, nz = (int64_t)g.B * g.C; z0 < nz; z0 += kMaxGridZ) {
        g.z0 = (int32_t)z0;
        hipLaunchKernelGGL(conv3d_filter_grad_kernel, dim3(g.d * g.h * g.w, (unsigned)std::min<int64_t>(nz - z0, kMaxGridZ)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), g, x, gy, dk_planes);
    }
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
