// Measurement synthesis on the device: additive Gaussian noise and the Cartesian MRI mask generators.
//
// Reference semantics:
//   GaussianNoise.forward           deepinv/physics/noise.py:197-330       y = x + sigma_b * N(0, 1)
//   Random / GaussianMaskGenerator  deepinv/physics/generator/mri.py:134-196, 262-301
//       per (batch, time) row: n_lines columns drawn WITHOUT replacement with probabilities pdf (zero on the centre
//       band), the centre band always sampled, every image row gets the same columns
//   EquispacedMaskGenerator         mri.py:304-384   columns round(arange((t + offset_b) % a, W - 1, a)), random offset_b
//   PolyOrderMaskGenerator          mri.py:199-281   every column an independent Bernoulli draw with probability pdf[w]
//       (the polynomial variable density shifted by bisection to the target rate; centre band probability 1)
//
// The reference draws from torch's generators with a Python loop per sample; here one launch serves the whole batch.
// Random numbers come from Philox4x32-10 (counter based: element i of a call uses counter (offset + i / 4), key = seed),
// so a call is reproducible from (seed, offset) and independent of the launch geometry.  The VALUES differ from torch's
// stream (another use of the same generator family), the DISTRIBUTIONS are the reference's: sampling n columns without
// replacement from pdf is done as "Gumbel top-n" (keys log p_w + G_w with G_w standard Gumbel; the n largest keys are
// distributed exactly like n successive draws without replacement from p - the Plackett-Luce identity that
// torch.multinomial(replacement=False) also implements).
#include "common.hpp"

using namespace dinv;

namespace {

struct Philox {
    uint32_t c[4], k[2];
    __device__ __forceinline__ Philox(uint64_t seed, uint64_t ctr, uint32_t stream) {
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = stream; c[3] = 0;
    }
    __device__ __forceinline__ void round_() {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int r = 0; r < 10; ++r) round_();
    }
};

// uniform in (0, 1]: never 0, so that log() is finite
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604645e-8f + 2.9802322e-8f; }

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    const float t = 6.28318530717958647692f * u01(b);
    n0 = r * cosf(t);
    n1 = r * sinf(t);
}

// y = x + sigma_b * N(0,1): 4 elements per thread from one Philox block; sigma: scalar (per_sample == 0) or [batch]
__global__ __launch_bounds__(256) void gaussian_noise_kernel(int64_t n, int64_t per_sample, const float* __restrict__ x,
                                                             const float* __restrict__ sigma, float sigma_scalar,
                                                             uint64_t seed, uint64_t offset, float* __restrict__ y) {
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        Philox ph(seed, offset + (uint64_t)q, 0u);
        ph.run();
        float z[4];
        box_muller(ph.c[0], ph.c[1], z[0], z[1]);
        box_muller(ph.c[2], ph.c[3], z[2], z[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = 4 * q + e;
            if (i < n) {
                const float s = sigma ? sigma[i / per_sample] : sigma_scalar;
                y[i] = fmaf(s, z[e], x ? x[i] : 0.f);
            }
        }
    }
}

constexpr int MAXW = 4096;

struct MaskArgs {
    int32_t rows;        // batch * T
    int32_t T, C, H, W;
    int32_t n_lines, c_lo, c_hi;   // columns [c_lo, c_hi) = always-sampled centre band
    int32_t mode;        // 0: random lines with probabilities pdf ; 1: equispaced with random offset ; 2: Bernoulli(pdf[w]) per column
    double accel;        // equispaced: adjusted acceleration (mri.py:357-359)
    int32_t n_offsets;   // equispaced: offset_b uniform in [0, n_offsets)
    uint64_t seed, offset;
};

// one workgroup per (batch, time) row: decide the W columns, then write them to every channel and image row
__global__ __launch_bounds__(256) void mask_lines_kernel(MaskArgs a, const float* __restrict__ pdf, float* __restrict__ mask) {
    __shared__ float key[MAXW];
    __shared__ unsigned char line[MAXW];
    const int row = blockIdx.x, b = row / a.T, t = row - b * a.T;
    const int tid = threadIdx.x;
    for (int w = tid; w < a.W; w += 256) line[w] = (w >= a.c_lo && w < a.c_hi) ? 1 : 0;
    if (a.mode == 0) {
        // Gumbel keys; a zero-probability column can never be selected
        for (int w = tid; w < a.W; w += 256) {
            Philox ph(a.seed, a.offset + (uint64_t)row * ((MAXW + 3) / 4) + (uint64_t)(w >> 2), 1u);
            ph.run();
            const float u = u01(ph.c[w & 3]);
            const float p = pdf[w];
            key[w] = p > 0.f ? logf(p) - logf(-logf(u)) : -3.0e38f;
        }
        __syncthreads();
        for (int w = tid; w < a.W; w += 256) {
            const float kw = key[w];
            if (kw > -1.0e38f) {
                int rank = 0;   // columns with a larger key (ties: the smaller index wins)
                for (int v = 0; v < a.W; ++v) rank += (key[v] > kw || (key[v] == kw && v < w)) ? 1 : 0;
                if (rank < a.n_lines) line[w] = 1;
            }
        }
    } else if (a.mode == 2) {
        // torch.bernoulli(pdf) (mri.py:273-281): column w is sampled iff u < pdf[w], u uniform; here u in (0, 1] and the test is
        // u <= p, so p = 1 (the centre band) always fires and p = 0 never does
        for (int w = tid; w < a.W; w += 256) {
            Philox ph(a.seed, a.offset + (uint64_t)row * ((MAXW + 3) / 4) + (uint64_t)(w >> 2), 3u);
            ph.run();
            line[w] = u01(ph.c[w & 3]) <= pdf[w] ? 1 : 0;
        }
    } else {
        __syncthreads();
        // offset_b uniform in [0, n_offsets); columns round(start + k * accel) for start + k * accel < W - 1
        Philox ph(a.seed, a.offset + (uint64_t)b, 2u);
        ph.run();
        const int off = a.n_offsets > 0 ? (int)(ph.c[0] % (uint32_t)a.n_offsets) : 0;
        const double start = fmod((double)(t + off), a.accel);
        for (int k = tid;; k += 256) {
            const double v = start + (double)k * a.accel;
            if (!(v < (double)(a.W - 1))) break;
            const int col = (int)rintf((float)v);     // arange in fp32, round half to even, cast (mri.py:371-380)
            if (col >= 0 && col < a.W) line[col] = 1;
        }
    }
    __syncthreads();
    // mask[b, c, t, h, w]
    const int64_t rowlen = a.W;
    for (int c = 0; c < a.C; ++c) {
        float* dst = mask + (((int64_t)b * a.C + c) * a.T + t) * a.H * rowlen;
        const int64_t total = (int64_t)a.H * rowlen;
        for (int64_t e = tid; e < total; e += 256) dst[e] = line[e % rowlen] ? 1.0f : 0.0f;
    }
}

}  // namespace

extern "C" int dinv_gaussian_noise(int64_t n, int64_t per_sample, const float* x, const float* sigma_dev,
                                   float sigma_scalar, uint64_t seed, uint64_t offset, float* y, dinv_stream_t stream) {
    DINV_REQUIRE(n >= 0 && y && per_sample >= 1, "bad arguments");
    if (n == 0) return 0;
    const int64_t nq = (n + 3) / 4;
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(nq, 256), 8192);
    hipLaunchKernelGGL(gaussian_noise_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, per_sample,
                       x, sigma_dev, sigma_scalar, seed, offset, y);
    DINV_CHECK_LAUNCH();
    return 0;
}

extern "C" int dinv_mri_mask_lines(int32_t batch, int32_t channels, int32_t times, int32_t height, int32_t width,
                                   int32_t n_lines, int32_t center_lo, int32_t center_hi, int32_t mode,
                                   const float* pdf_dev, double accel, int32_t n_offsets, uint64_t seed, uint64_t offset,
                                   float* mask, dinv_stream_t stream) {
    DINV_REQUIRE(batch >= 0 && channels >= 1 && times >= 1 && height >= 1 && width >= 1 && mask, "bad mask geometry");
    DINV_REQUIRE(width <= MAXW, "mask width %d above the generator's limit %d", width, MAXW);
    DINV_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (random lines), 1 (equispaced) or 2 (Bernoulli columns)");
    DINV_REQUIRE(mode == 1 || pdf_dev, "random-line and Bernoulli masks need the column probabilities");
    DINV_REQUIRE(mode != 1 || accel > 0.0, "equispaced masks need a positive acceleration");
    DINV_REQUIRE(n_lines >= 0 && center_lo >= 0 && center_lo <= center_hi && center_hi <= width, "bad line counts");
    if (batch == 0) return 0;
    MaskArgs a{batch * times, times, channels, height, width, n_lines, center_lo, center_hi, mode, accel, n_offsets, seed, offset};
    hipLaunchKernelGGL(mask_lines_kernel, dim3((unsigned)(batch * times)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, pdf_dev, mask);
    DINV_CHECK_LAUNCH();
    return 0;
}
