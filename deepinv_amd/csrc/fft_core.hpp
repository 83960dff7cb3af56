// In-LDS mixed-radix FFT engine for gfx950.
//
// One workgroup owns a tile of `lines` independent sequences of length N that live in
// LDS as float2 buf[line * LS + pos].  The transform is an in-place decimation-in-time
// Cooley-Tukey: the caller scatters input element n to LDS position perm[n] (mixed-radix
// digit reversal, free because the global->LDS copy is a scatter anyway), the stages run
// innermost radix first, and the result comes out in natural order, so the store phase
// reads LDS linearly.  Radices 2,3,4,5,8 are register butterflies; any other (prime)
// factor takes the generic O(r^2) stage which ping-pongs into a second LDS buffer.
// Twiddles exp(-2 pi i t / N) are a per-length table built on the host in fp64
// (dinv_fft_plan_init) and staged into LDS once per workgroup.
//
// Centred transforms (MRIMixin.fft = ifftshift -> fft -> fftshift,
// reference deepinv/utils/mixins.py:159-180) never move data: both shifts are the index
// map idx -> (idx - N/2) mod N applied on the global side of the load and of the store.
#pragma once
#include "common.hpp"

// The butterflies and stages are __host__ __device__ so that tests/csrc/host_fft_emul.hip can
// execute the very same code on the CPU (single "thread": tid = 0, nthr = 1) in the GPU-less
// build container.  That emulation is test infrastructure; the product never calls it.
#if defined(__HIP_DEVICE_COMPILE__)
#define DINV_SYNC() __syncthreads()
#define DINV_CLZ(x) __clz(x)
#else
#define DINV_SYNC() ((void)0)
#define DINV_CLZ(x) __builtin_clz((unsigned)(x))
#endif
#define DINV_HD __host__ __device__ __forceinline__

namespace dinv {

__host__ __device__ inline size_t fft_table_bytes(int n) { return (size_t)n * (sizeof(float2) + sizeof(int)); }

// ------------------------------------------------------------------ butterflies
template <bool INV>
DINV_HD float2 rot90(float2 a) {
    // forward: multiply by -i ; inverse: multiply by +i
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, bool INV>
struct Bfly;

template <bool INV>
struct Bfly<2, INV> {
    static DINV_HD void run(float2 (&v)[2]) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <bool INV>
struct Bfly<3, INV> {
    static DINV_HD void run(float2 (&v)[3]) {
        const float s = 0.86602540378443864676f;
        float2 t1 = cadd(v[1], v[2]);
        float2 t2 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        float2 t3 = cscale(csub(v[1], v[2]), s);
        float2 r = rot90<INV>(t3);  // fwd: -i*t3
        v[0] = cadd(v[0], t1);
        v[1] = cadd(t2, r);
        v[2] = csub(t2, r);
    }
};

template <bool INV>
struct Bfly<4, INV> {
    static DINV_HD void run(float2 (&v)[4]) {
        float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        float2 c = cadd(v[1], v[3]), d = rot90<INV>(csub(v[1], v[3]));
        v[0] = cadd(a, c);
        v[2] = csub(a, c);
        v[1] = cadd(b, d);
        v[3] = csub(b, d);
    }
};

template <bool INV>
struct Bfly<5, INV> {
    static DINV_HD void run(float2 (&v)[5]) {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
        float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
        float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
        float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
        float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        float2 b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        float2 r1 = rot90<INV>(b1), r2 = rot90<INV>(b2);  // fwd: -i*b
        v[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
        v[1] = cadd(a1, r1);
        v[4] = csub(a1, r1);
        v[2] = cadd(a2, r2);
        v[3] = csub(a2, r2);
    }
};

template <bool INV>
struct Bfly<8, INV> {
    static DINV_HD void run(float2 (&v)[8]) {
        const float h = 0.70710678118654752440f;
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        Bfly<4, INV>::run(e);
        Bfly<4, INV>::run(o);
        // o[k] *= W8^k  (forward W8 = exp(-i pi/4))
        float2 o1 = INV ? make_float2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
                        : make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
        float2 o2 = rot90<INV>(o[2]);
        float2 o3 = INV ? make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
                        : make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
        v[0] = cadd(e[0], o[0]);
        v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);
        v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);
        v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);
        v[7] = csub(e[3], o3);
    }
};

// ------------------------------------------------------------------ stages
// In-place radix-R stage.  M = length of the already-transformed sub-blocks (= stride
// between the R inputs of one butterfly); block length is R*M.
template <int R, bool INV>
DINV_HD void stage_reg(float2* buf, const float2* tw, int N, int M, int lines,
                                          int LS, int tid, int nthr) {
    const int per_line = N / R;
    const int total = per_line * lines;
    const int twstep = N / (R * M);
    const bool pow2 = (M & (M - 1)) == 0;
    const int sh = 31 - DINV_CLZ(M);
    for (int g = tid; g < total; g += nthr) {
        const int line = g / per_line;
        const int u = g - line * per_line;
        int blk, k;
        if (pow2) {
            blk = u >> sh;
            k = u & (M - 1);
        } else {
            blk = u / M;
            k = u - blk * M;
        }
        float2* p = buf + line * LS + blk * (R * M) + k;
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = p[j * M];
        if (M > 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) {
                float2 w = tw[j * k * twstep];
                v[j] = INV ? cmulc(v[j], w) : cmul(v[j], w);
            }
        }
        Bfly<R, INV>::run(v);
#pragma unroll
        for (int j = 0; j < R; ++j) p[j * M] = v[j];
    }
}

// Out-of-place generic radix stage (any R): one output element per loop trip.
template <bool INV>
DINV_HD void stage_generic(const float2* src, float2* dst, const float2* tw, int N,
                                              int R, int M, int lines, int LS, int tid, int nthr) {
    const int total = N * lines;
    const int L = R * M;
    const int twstep = N / L;
    const int rstep = N / R;
    for (int g = tid; g < total; g += nthr) {
        const int line = g / N;
        const int e = g - line * N;
        const int blk = e / L;
        const int rem = e - blk * L;
        const int q = rem / M;
        const int k = rem - q * M;
        const float2* p = src + line * LS + blk * L + k;
        float ax = 0.f, ay = 0.f;
        int t1 = 0;  // j*k*twstep  (< N)
        int t2 = 0;  // (j*q mod R) * rstep
        for (int j = 0; j < R; ++j) {
            int t = t1 + t2;
            if (t >= N) t -= N;
            float2 w = tw[t];
            float2 x = p[j * M];
            float2 y = INV ? cmulc(x, w) : cmul(x, w);
            ax += y.x;
            ay += y.y;
            t1 += k * twstep;
            t2 += q * rstep;
            if (t2 >= N) t2 -= N;
        }
        dst[line * LS + e] = make_float2(ax, ay);
    }
}

// Runs all stages on the tile.  `buf` holds the permuted input; `alt` is the ping-pong
// buffer (only touched when the plan has a generic stage).  Returns the buffer that holds
// the natural-order result.  Starts and ends with a workgroup barrier.
template <bool INV>
DINV_HD float2* tile_fft(const dinv_fft_plan& plan, float2* buf, float2* alt,
                                            const float2* tw, int lines, int LS, int tid, int nthr) {
    const int N = plan.n;
    int M = 1;
    float2* cur = buf;
    for (int s = plan.nstages - 1; s >= 0; --s) {
        const int R = plan.radix[s];
        DINV_SYNC();
        switch (R) {
            case 2: stage_reg<2, INV>(cur, tw, N, M, lines, LS, tid, nthr); break;
            case 3: stage_reg<3, INV>(cur, tw, N, M, lines, LS, tid, nthr); break;
            case 4: stage_reg<4, INV>(cur, tw, N, M, lines, LS, tid, nthr); break;
            case 5: stage_reg<5, INV>(cur, tw, N, M, lines, LS, tid, nthr); break;
            case 8: stage_reg<8, INV>(cur, tw, N, M, lines, LS, tid, nthr); break;
            default: {
                stage_generic<INV>(cur, alt, tw, N, R, M, lines, LS, tid, nthr);
                float2* t = cur;
                cur = alt;
                alt = t;
            }
        }
        M *= R;
    }
    DINV_SYNC();
    return cur;
}

// Stage the per-length tables into LDS.  Layout: float2 tw[N] ; int perm[N].
DINV_HD void load_tables(float2* tw_s, int* perm_s, const void* table, int N, int tid,
                                            int nthr) {
    const float2* tw_g = reinterpret_cast<const float2*>(table);
    const int* perm_g = reinterpret_cast<const int*>(tw_g + N);
    for (int i = tid; i < N; i += nthr) {
        tw_s[i] = tw_g[i];
        perm_s[i] = perm_g[i];
    }
}

// LDS line stride: odd (in float2 units) so that column tiles scatter conflict-free.
inline int fft_line_stride(int n) { return (n % 2 == 0) ? n + 1 : n; }

// LDS carve: [tw N*8][perm N*4 rounded to 8][buf lines*LS*8][alt lines*LS*8 if generic]
inline size_t fft_lds_bytes(const dinv_fft_plan& p, int lines) {
    const int LS = fft_line_stride(p.n);
    size_t b = (size_t)p.n * 8 + (((size_t)p.n * 4 + 15) / 16) * 16;
    b = ((b + 15) / 16) * 16;
    b += (size_t)lines * LS * 8 * (p.generic ? 2 : 1);
    return b;
}

struct LdsCarve {
    float2* tw;
    int* perm;
    float2* buf;
    float2* alt;
};

DINV_HD LdsCarve carve_lds(unsigned char* smem, int N, int lines, int LS, bool generic) {
    LdsCarve c;
    c.tw = reinterpret_cast<float2*>(smem);
    c.perm = reinterpret_cast<int*>(smem + (size_t)N * 8);
    size_t off = (size_t)N * 8 + (((size_t)N * 4 + 15) / 16) * 16;
    off = ((off + 15) / 16) * 16;
    c.buf = reinterpret_cast<float2*>(smem + off);
    c.alt = generic ? c.buf + (size_t)lines * LS : c.buf;
    return c;
}

constexpr size_t kMaxLdsBytes = 160 * 1024;

}  // namespace dinv
