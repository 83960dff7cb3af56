// LDS-tiled parallel-beam Radon transform and its exact adjoint for gfx950, plus the FFT ramp filter.
//
// Same arithmetic as csrc/radon.hip (reference semantics: deepinv/physics/functional/radon.py:252-342, exact adjoint
// = forward.py:1302-1362 / tomography.py:311-350, ramp filter radon.py:79-162); what changes is where the operands
// live while they are gathered:
//
//   forward  A workgroup = 64 adjacent rays x KW angles (one angle per wave) x NB images.  The rays of a chunk all run
//            "mostly along" one image axis (class PLAIN: |cos| >= |sin|, band coordinate v = row; class SWAP: the
//            transposed image copy is used and v = column), so the image is consumed in bands of BH rows: the band's
//            window [BH+1 rows] x [ww columns] x NB images is staged into LDS with 16-byte coalesced loads, every
//            ray marches through the samples whose upper-left tap lies in the band and reads its four taps from LDS
//            (one ds_read_b128 per tap and 4 images), then the workgroup moves on to the next band.  Every ray
//            accumulates its own line integral in registers: no atomics, no cross-workgroup reduction.  The window
//            column range per (chunk, ray block, band) is a host-built table (dinv_radon_plan_init, fp64 geometry with
//            a one-pixel margin); angle lists whose chunks do not fit the window width are flagged (`fits` = 0) and
//            keep the gather kernels of radon.hip.
//   adjoint  deterministic gather (one thread = one pixel of NB images, like radon.hip) with the sinogram segment a
//            16x16 pixel tile can touch staged in LDS per angle chunk, a 3x3 instead of a 4x4 candidate window (the
//            lattice points within |cos|+|sin| of the pixel), and the tap weight evaluated as clamp(1-|t|) of the
//            *bit-identical* forward sample position, which equals the forward's bilinear weight (see weight_of()).
//   ramp     zero-pad to P = 2^ceil(log2(2 N)) -> FFT -> x F -> inverse FFT -> crop, in ONE kernel per column tile:
//            two adjacent angle columns travel as the real and imaginary part of one complex sequence (the filter is
//            real and even), the forward transform is decimation-in-frequency (transposed DIT stages) so that its
//            digit-reversed output feeds the inverse decimation-in-time stages directly - no permutation pass.
#include "common.hpp"
#include "fft_core.hpp"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)  // forward and adjoint must round the sample coordinates identically (and like radon.hip)

using namespace dinv;

// Two objects are built from this file (csrc/Makefile): DINV_RADON_PART = 1 is dinv_radon_forward_tiled alone, compiled WITH SLP
// vectorisation (its window arithmetic packs into v_pk_* forms that pay: 1.16 against 1.31 ms at config 3, and none of them is the
// op_sel[1] = 1 form of DESIGN.md 3.6 - tests/test_abi.py scans for it); DINV_RADON_PART = 2 is everything else, without SLP like the
// rest of the library (the adjoint is 17 % faster that way: 0.71 against 0.86 ms).  0 = the whole file (host emulation).
#ifndef DINV_RADON_PART
#define DINV_RADON_PART 0
#endif

#ifdef DINV_EMU
extern "C" { int dinv_emu_window_misses = 0; int dinv_emu_segment_misses = 0; }
#endif

namespace {

constexpr int KWMAX = 8;     // most angles per workgroup (one per wave; the plan picks 8, 4, 2 or 1)
constexpr int BH = 16;       // band height
constexpr int WWPREF = 128;  // window width (columns) up to which two workgroups of 8 images share a CU's LDS;
                             // a multiple of 16: see the pitch note in dinv_radon_plan_init
constexpr int JW = 32;       // sinogram segment length staged per angle in the adjoint
constexpr int KA = 16;       // angles per staging round in the adjoint
constexpr int JPAD = 32;     // zero entries before/after each packed sinogram row
constexpr int MAXG = 4096;   // detector count limit of the tiled kernels (LDS table of the base grid)

struct TiledGeom {
    int32_t n_img, W, G, pad, A, circle, groups;
    int32_t PW, PH;                 // packed image pitch / rows (one zero ring + band overhang)
    int32_t njb, nbands;
    int32_t WW;                     // LDS window pitch in columns (forward only; from the plan)
    uint32_t row_magic;             // ceil(2^32 / (WW * planes)): chunk index -> window row by multiply-high
    float scale;
};

// identical to radon.hip (see the note there): sample position (ix -> column, iy -> row) of lattice point (j, i) of the rotated
// uniform grid, (ix, iy) = ctr + R (j - ctr, i - ctr): a per-ray base and one fused multiply-add per coordinate and step
__device__ __forceinline__ void ray_base(float c, float s, float dj, float ctr, float& bx, float& by) {
    bx = fmaf(c, dj, ctr);
    by = fmaf(-s, dj, ctr);
}
__device__ __forceinline__ void lattice_pos(float c, float s, float bx, float by, float di, float& ix, float& iy) {
    ix = fmaf(s, di, bx);
    iy = fmaf(c, di, by);
}

// bilinear weight of integer pixel p for a sample at position t along one axis.  With f = floor(t): p == f gives
// 1 - (t - f) and p == f + 1 gives t - f, both EXACTLY what grid_sample's weights are (t - p is exact by Sterbenz,
// and 1 - (1 - frac) is exact because frac is a multiple of ulp(t) >= 2^-23); any other p clamps to 0.
__device__ __forceinline__ float weight_of(float t, float p) {
    const float w = 1.0f - fabsf(t - p);
    return w > 0.f ? w : 0.f;
}

template <int NB> struct Vec { static constexpr int V = NB >= 4 ? 4 : NB; static constexpr int PLANES = NB / V; };

// ---------------------------------------------------------------------------------------------------- pack
// x [n_img, W, W] -> two packed copies [groups][PH][PW][NB]: plain (row = grid y + 1, col = grid x + 1) and
// transposed (row = grid x + 1, col = grid y + 1); zero ring, zero padding, optional inscribed-disc mask
template <int NB>
__global__ void radon_pack_image2(TiledGeom g, const float* __restrict__ x, float* __restrict__ xp,
                                  float* __restrict__ xpt) {
    const int64_t per = (int64_t)g.PH * g.PW;
    const int64_t total = (int64_t)g.groups * per;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(idx / per);
        const int rem = (int)(idx - (int64_t)grp * per);
        const int pr = rem / g.PW, pc = rem - pr * g.PW;
        const int r = pr - 1 - g.pad, c = pc - 1 - g.pad;   // original image coordinates (row, col)
        bool in = r >= 0 && r < g.W && c >= 0 && c < g.W;
        bool in_t = in;
        if (in && g.circle) {   // radon.py:270-283 (symmetric in r and c)
            const float ya = 2.0f * (float)c / (float)(g.W - 1) - 1.0f;
            const float xa = 2.0f * (float)r / (float)(g.W - 1) - 1.0f;
            in = in_t = (xa * xa + ya * ya) <= 1.0f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            const bool ok = n < g.n_img;
            xp[idx * NB + k] = (in && ok) ? x[((int64_t)n * g.W + r) * g.W + c] : 0.f;
            xpt[idx * NB + k] = (in_t && ok) ? x[((int64_t)n * g.W + c) * g.W + r] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward
template <int V> using vf = float __attribute__((ext_vector_type(V)));

// block = 64 * kw threads (kw = angles per chunk, plan->kw); LDS: xn[G] ; win[PLANES][BH+1][WW][V]
template <int NB, bool SWAP, int MAXPF>
__global__ __launch_bounds__(512, MAXPF <= 10 ? 4 : 2) void radon_fwd_tiled_kernel(TiledGeom g, const float* __restrict__ xp,
                                                              const float* __restrict__ xn,
                                                              const float2* __restrict__ cs,
                                                              const int32_t* __restrict__ chunk_angles,
                                                              const int32_t* __restrict__ chunk_dir,
                                                              const int32_t* __restrict__ wtab,
                                                              const float* __restrict__ norm,
                                                              float* __restrict__ sino, int chunk_base) {
    constexpr int V = Vec<NB>::V, PLANES = Vec<NB>::PLANES;
    using VF = vf<V>;
    DINV_DYN_LDS(float, lds);
    float* win = lds + ((g.G + 3) & ~3);      // (the first G words are reserved by the launch and unused)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kw = blockDim.x >> 6, nthr = blockDim.x;
    const int WW = g.WW;
    const int jb = blockIdx.x, ch = chunk_base + blockIdx.y, grp = blockIdx.z;
    const int a = chunk_angles[ch * kw + wv];
    const int dir = chunk_dir[ch];
    // ds_read_b128 is serviced in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): give each group
    // 16 ADJACENT rays, whose taps are adjacent window columns, so that a group spans as few bank rows as possible
    const int l5 = lane & 31;
    const bool g0 = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int ray = (lane & 32) + (g0 ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12))
                                      : 16 + (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)));
    const int j = jb * 64 + ray;
    const bool active = a >= 0 && j < g.G;
    const float ctr = 0.5f * (float)(g.G - 1);
    float c = 0.f, s = 0.f, bx = 0.f, by = 0.f;
    if (active) {
        const float2 t = cs[a];
        c = t.x; s = t.y;
        ray_base(c, s, (float)j - ctr, ctr, bx, by);
    }
    VF acc[PLANES];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) acc[pl] = VF(0.f);
    int u0 = 0, v0 = 0;
    float tu = 0.f, tv = 0.f;
    const float dirf = (float)dir;
    auto eval = [&](float dcur) {
        float ix, iy;
        lattice_pos(c, s, bx, by, dcur, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        if (SWAP) { u0 = (int)fy; v0 = (int)fx; tu = iy - fy; tv = ix - fx; }
        else      { u0 = (int)fx; v0 = (int)fy; tu = ix - fx; tv = iy - fy; }
    };
    // The steps of a ray whose upper-left tap lies inside the zero-ringed image, floor(ix), floor(iy) in [-1, G-1], form ONE interval
    // [ilo, ihi] (a line meets a square in a segment): it is found once per ray - an analytic estimate, widened by a step and
    // then trimmed with the very test the samples would have failed - and the march runs over that interval only: no range
    // test, no clamp and no zero-weight sample inside the loop (at 45 degrees a fifth of the lattice lies outside the image).
    const float gtop = (float)(g.G - 1);
    auto inside = [&](int ii) {
        float ix, iy;
        lattice_pos(c, s, bx, by, (float)ii - ctr, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        return fx >= -1.0f && fx <= gtop && fy >= -1.0f && fy <= gtop;
    };
    int nleft = 0;               // samples left on this ray
    float di = 0.f;              // step coordinate i - ctr of the current sample (exact: half-integers)
    if (active) {
        float lo = -ctr, hi = gtop - ctr;                 // in units of i - ctr
        auto clip = [&](float b, float slope) {           // -1 <= b + slope * d < G
            if (fabsf(slope) > 1e-12f) {
                const float t1 = (-1.0f - b) / slope, t2 = ((float)g.G - b) / slope;
                lo = fmaxf(lo, fminf(t1, t2));
                hi = fminf(hi, fmaxf(t1, t2));
            } else if (!(b >= -1.0f && b < (float)g.G)) {
                hi = lo - 4.0f;                           // parallel to that side and outside: empty
            }
        };
        clip(bx, s);
        clip(by, c);
        lo = fminf(lo, gtop - ctr + 2.0f);                // (a ray far outside: keep the estimate inside the int range)
        hi = fmaxf(hi, -ctr - 2.0f);
        int ilo = (int)floorf(lo + ctr) - 1, ihi = (int)ceilf(hi + ctr) + 1;
        ilo = ilo < 0 ? 0 : ilo;
        ihi = ihi > g.G - 1 ? g.G - 1 : ihi;
        while (ilo <= ihi && !inside(ilo)) ++ilo;
        while (ihi >= ilo && !inside(ihi)) --ihi;
        nleft = ihi >= ilo ? ihi - ilo + 1 : 0;
        di = (float)(dir > 0 ? ilo : ihi) - ctr;
        eval(di);
    }
    const int32_t* wt = wtab + ((int64_t)ch * g.njb + jb) * g.nbands;
    const float* img = xp + (int64_t)grp * g.PH * g.PW * NB;
    const int plane_stride = (BH + 1) * WW * V;
    // Window staging is software pipelined: the 16-byte chunks of band b+1 are loaded into registers BEFORE the rays
    // march through band b (all loads in flight together, their latency hidden behind the march) and written to LDS
    // after the barrier that ends band b.  A window row is staged at the full pitch WW (constant chunk -> (row, col)
    // map: one multiply-high per chunk); columns beyond the band's own width are never read by a valid sample.
    // MAXPF = chunks per thread held in registers (9 covers 8-wave workgroups, 18 the 4-wave ones)
    const unsigned rowlen = (unsigned)(WW * PLANES);   // chunks per window row
    const unsigned total = (unsigned)(BH + 1) * rowlen;
    const unsigned magic = g.row_magic;              // ceil(2^32 / rowlen)
    VF pre[MAXPF];
    // chunk -> (row, column chunk) is the same in every band: global and LDS offsets are computed once; per band only
    // the (uniform) base address of the window changes
    int goff[MAXPF], loff[MAXPF];
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
        const unsigned q = tid + k * nthr;
        const unsigned r = __umulhi(q, magic), cq = q - r * rowlen;
        const unsigned col = cq / PLANES, pl = cq - col * PLANES;
        goff[k] = (int)(r * (unsigned)(g.PW * NB) + cq * V);
        loff[k] = (int)(pl * plane_stride + (r * WW + col) * V);
    }
    auto band_base = [&](int band) -> const float* {
        const int wx0 = (wt[band] & 0xffff) - 8;
        return img + ((int64_t)band * BH * g.PW + (wx0 + 1)) * NB;
    };
    auto issue = [&](int band) {
        const float* base = band_base(band);
#pragma unroll
        for (int k = 0; k < MAXPF; ++k)
            if (tid + k * nthr < total) pre[k] = *reinterpret_cast<const VF*>(base + goff[k]);
    };
    auto commit = [&](int band) {
#pragma unroll
        for (int k = 0; k < MAXPF; ++k)
            if (tid + k * nthr < total) *reinterpret_cast<VF*>(win + loff[k]) = pre[k];
        const float* base = band_base(band);
        for (unsigned q = tid + MAXPF * nthr; q < total; q += nthr) {   // few-wave workgroups: the rest, synchronously
            const unsigned r = __umulhi(q, magic), cq = q - r * rowlen;
            const unsigned col = cq / PLANES, pl = cq - col * PLANES;
            *reinterpret_cast<VF*>(win + pl * plane_stride + (r * WW + col) * V) =
                *reinterpret_cast<const VF*>(base + r * (unsigned)(g.PW * NB) + cq * V);
        }
    };
    issue(0);
    for (int band = 0; band < g.nbands; ++band) {
        const int wx0 = (wt[band] & 0xffff) - 8;   // first window column (grid coordinate)
#ifdef DINV_EMU
        const int ww = wt[band] >> 16;
#endif
        const int vb = -1 + band * BH;
        __syncthreads();   // every ray is done with the previous window
        commit(band);
        __syncthreads();
        if (band + 1 < g.nbands) issue(band + 1);
        if (active) {
            // one basic block per sample; the position of the NEXT sample is evaluated between the tap reads and the multiply-adds
            // of the current one
            const int vend = band == g.nbands - 1 ? 0x7fffffff : vb + BH;
            while (nleft > 0 && v0 < vend) {
                const int col = u0 - wx0, row = v0 - vb;            // inside the window by construction (the plan; the interval)
#ifdef DINV_EMU
                if (col < 0 || col > ww - 2 || row < 0 || row > BH - 1) ++dinv_emu_window_misses;   // host emulation only
#endif
                const float a0 = 1.0f - tu, b0 = 1.0f - tv;
                const float w00 = a0 * b0, w01 = tu * b0, w10 = a0 * tv, w11 = tu * tv;
                const float* p = win + (row * WW + col) * V;
                VF t00[PLANES], t01[PLANES], t10[PLANES], t11[PLANES];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const float* pp = p + pl * plane_stride;
                    t00[pl] = *reinterpret_cast<const VF*>(pp);
                    t01[pl] = *reinterpret_cast<const VF*>(pp + V);
                    t10[pl] = *reinterpret_cast<const VF*>(pp + WW * V);
                    t11[pl] = *reinterpret_cast<const VF*>(pp + WW * V + V);
                }
                --nleft;
                di += dirf;
                eval(di);      // the position of the NEXT sample between the tap reads and the multiply-adds of this one
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float r = acc[pl][e];
                        r = fmaf(w00, t00[pl][e], r);
                        r = fmaf(w01, t01[pl][e], r);
                        r = fmaf(w10, t10[pl][e], r);
                        r = fmaf(w11, t11[pl][e], r);
                        acc[pl][e] = r;
                    }
            }
        }
    }
    if (active) {
        const float nrm = norm ? norm[0] : 1.0f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            if (n < g.n_img) {
                const float v = acc[k / V][k % V] * g.scale;
                sino[((int64_t)n * g.G + j) * g.A + a] = norm ? v / nrm : v;
            }
        }
    }
}

#if DINV_RADON_PART != 1
// ---------------------------------------------------------------------------------------------------- adjoint
// sino [n_img, G, A] -> sp [groups][A][JPAD + G + JPAD][NB] (zero padded rows)
template <int NB>
__global__ void radon_pack_sino2(TiledGeom g, const float* __restrict__ sino, float* __restrict__ sp) {
    const int GJ = g.G + 2 * JPAD;
    const int64_t total = (int64_t)g.groups * g.A * GJ;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(idx / ((int64_t)g.A * GJ));
        const int rem = (int)(idx - (int64_t)grp * g.A * GJ);
        const int a = rem / GJ, j = rem - a * GJ - JPAD;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            sp[idx * NB + k] = (n < g.n_img && j >= 0 && j < g.G) ? sino[((int64_t)n * g.G + j) * g.A + a] : 0.f;
        }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void radon_adj_tiled_kernel(TiledGeom g, const float* __restrict__ sp,
                                                              const float* __restrict__ xn,
                                                              const float2* __restrict__ cs,
                                                              const float* __restrict__ norm, float* __restrict__ x) {
    constexpr int V = Vec<NB>::V, PLANES = Vec<NB>::PLANES;
    using VF = vf<V>;
    DINV_DYN_LDS(float, lds);
    float* seg = lds + ((g.G + 3) & ~3);                   // [KA][PLANES][JW][V]  (the first G words: reserved by the launch, unused)
    float2* cs_s = reinterpret_cast<float2*>(seg + KA * JW * NB);   // [KA]
    int* jlo_s = reinterpret_cast<int*>(cs_s + KA);        // [KA]
    const int tid = threadIdx.x;
    const int col = blockIdx.x * 16 + (tid & 15);
    const int row = blockIdx.y * 16 + (tid >> 4);
    const int grp = blockIdx.z;
    const int GJ = g.G + 2 * JPAD;
    const float gm1 = (float)(g.G - 1);
    const float ctr = 0.5f * gm1;
    const int px = col + g.pad, py = row + g.pad;   // padded-grid pixel (adjoint of the zero pad = crop)
    const float fpx = (float)px, fpy = (float)py;
    const float dx = fpx - ctr, dy = fpy - ctr;
    const float dxc = (float)(blockIdx.x * 16 + g.pad) + 7.5f - ctr;   // tile centre
    const float dyc = (float)(blockIdx.y * 16 + g.pad) + 7.5f - ctr;
    bool live = col < g.W && row < g.W;
    if (live && g.circle) {
        const float ya = 2.0f * (float)col / (float)(g.W - 1) - 1.0f;
        const float xa = 2.0f * (float)row / (float)(g.W - 1) - 1.0f;
        live = (xa * xa + ya * ya) <= 1.0f;
    }
    float acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.f;
    for (int a0 = 0; a0 < g.A; a0 += KA) {
        __syncthreads();
        if (tid < KA) {
            const int a = a0 + tid;
            float2 t = make_float2(1.f, 0.f);
            int jlo = 0;
            if (a < g.A) {
                t = cs[a];
                jlo = (int)floorf(t.x * dxc - t.y * dyc + ctr) - 13;
                jlo = jlo < -JPAD ? -JPAD : (jlo > g.G + JPAD - JW ? g.G + JPAD - JW : jlo);
            }
            cs_s[tid] = t;
            jlo_s[tid] = jlo;
        }
        __syncthreads();
        {   // stage seg[ai][pl][q][V] <- sp[grp][a0+ai][JPAD + jlo + q][pl*V ..]; a thread's loads are all issued
            // before the first LDS write (KA * JW * PLANES chunks / 256 threads = 4 per thread for 8 images)
            constexpr int CH = KA * JW * PLANES, PER = (CH + 255) / 256;
            VF v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int f = tid + k * 256;
                const int ai = f / (JW * PLANES), r = f - ai * (JW * PLANES);
                const int q = r / PLANES, pl = r - q * PLANES;
                const int a = a0 + ai;
                v[k] = VF(0.f);
                if (f < CH && a < g.A)
                    v[k] = *reinterpret_cast<const VF*>(sp + (((int64_t)grp * g.A + a) * GJ + (JPAD + jlo_s[ai] + q)) * NB + pl * V);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int f = tid + k * 256;
                const int ai = f / (JW * PLANES), r = f - ai * (JW * PLANES);
                const int q = r / PLANES, pl = r - q * PLANES;
                if (f < CH) *reinterpret_cast<VF*>(seg + (((int64_t)ai * PLANES + pl) * JW + q) * V) = v[k];
            }
        }
        __syncthreads();
        if (live) {
            const int na = g.A - a0 < KA ? g.A - a0 : KA;
            for (int ai = 0; ai < na; ++ai) {
                const float c = cs_s[ai].x, s = cs_s[ai].y;
                // approximate inverse map (selects the candidate window only): lattice coordinates of the pixel
                const float ux = c * dx - s * dy + ctr, uy = s * dx + c * dy + ctr;
                const float w = fabsf(c) + fabsf(s) + 1e-3f;
                const int jc = (int)ceilf(ux - w), ic = (int)ceilf(uy - w);
                // lattice points outside the grid are moved far away instead of being masked: their weights clamp to zero
                float dif[3];
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const int i = ic + di;
                    dif[di] = (unsigned)i < (unsigned)g.G ? (float)i - ctr : 1.0e9f;
                }
                const int jl = jlo_s[ai];
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int j = jc + dj;
                    float bx, by;
                    ray_base(c, s, (unsigned)j < (unsigned)g.G ? (float)j - ctr : 1.0e9f, ctr, bx, by);
                    float wsum = 0.f;
#pragma unroll
                    for (int di = 0; di < 3; ++di) {
                        float ix, iy;
                        lattice_pos(c, s, bx, by, dif[di], ix, iy);
                        wsum += weight_of(ix, fpx) * weight_of(iy, fpy);
                    }
                    int q = j - jl;
#ifdef DINV_EMU
                    if ((q < 0 || q > JW - 1) && wsum != 0.f) ++dinv_emu_segment_misses;
#endif
                    q = q < 0 ? 0 : (q > JW - 1 ? JW - 1 : q);   // out-of-segment candidates have wsum == 0
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        const VF v = *reinterpret_cast<const VF*>(seg + ((ai * PLANES + pl) * JW + q) * V);
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[pl * V + e] = fmaf(wsum, v[e], acc[pl * V + e]);
                    }
                }
            }
        }
    }
    if (col < g.W && row < g.W) {
        const float nrm = norm ? norm[0] : 1.0f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int n = grp * NB + k;
            if (n < g.n_img) {
                const float v = acc[k] * g.scale;
                x[((int64_t)n * g.W + row) * g.W + col] = norm ? v / nrm : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- ramp filter
// transposed (decimation-in-frequency) counterpart of stage_reg: butterfly first, then the twiddles on the outputs
template <int R, bool INV>
__device__ __forceinline__ void stage_reg_dif(float2* buf, const float2* tw, int N, int M, int lines, int LS, int tid,
                                              int nthr) {
    const int per_line = N / R;
    const int total = per_line * lines;
    const int twstep = N / (R * M);
    const int sh = 31 - __clz(M);   // M is a power of two here
    for (int w = tid; w < total; w += nthr) {
        const int line = w / per_line;
        const int u = w - line * per_line;
        const int blk = u >> sh, k = u & (M - 1);
        float2* p = buf + line * LS + blk * (R * M) + k;
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = p[j * M];
        Bfly<R, INV>::run(v);
        if (M > 1) {
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const float2 t = tw[j * k * twstep];
                v[j] = INV ? cmulc(v[j], t) : cmul(v[j], t);
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) p[j * M] = v[j];
    }
}

// y [n_img, N, A] -> out, filtered along N.  One workgroup = CT complex columns (2 CT angles) of one sinogram.
// LDS: tw[P] (float2) ; buf[CT][P+1] (float2).  filt[P] = F[k] stored at the digit-reversed position of k.
__global__ __launch_bounds__(256) void ramp_fft_kernel(int n_img, int N, int A, int P, int CT, dinv_fft_plan plan,
                                                       const void* __restrict__ table,
                                                       const float* __restrict__ filt, const float* __restrict__ y,
                                                       float* __restrict__ out) {
    DINV_DYN_LDS(float2, lds2);
    float2* tw = lds2;
    float2* buf = lds2 + P;
    const int LS = P + 1;
    const int tid = threadIdx.x;
    const float2* tw_g = reinterpret_cast<const float2*>(table);
    for (int i = tid; i < P; i += 256) tw[i] = tw_g[i];
    const int n = blockIdx.y;
    const int c0 = blockIdx.x * CT;              // first complex column of this tile
    const int ncol = (A + 1) / 2;                // complex columns (the last one is half empty when A is odd)
    const float* src = y + (int64_t)n * N * A;
    float* dst = out + (int64_t)n * N * A;
    // load: element (m, cc) = (y[m][2c], y[m][2c+1]); rows N..P-1 are the zero padding.  Eight independent loads per
    // thread are in flight before the first LDS write.
    constexpr int U = 8;
    for (int e0 = tid; e0 < N * CT; e0 += 256 * U) {
        float2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 256;
            const int m = e / CT, cc = e - m * CT;
            const int a0 = 2 * (c0 + cc);
            v[u] = make_float2(0.f, 0.f);
            if (e < N * CT && c0 + cc < ncol) {
                v[u].x = src[(int64_t)m * A + a0];
                if (a0 + 1 < A) v[u].y = src[(int64_t)m * A + a0 + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 256;
            const int m = e / CT, cc = e - m * CT;
            if (e < N * CT) buf[cc * LS + m] = v[u];
        }
    }
    for (int e = N * CT + tid; e < P * CT; e += 256) {
        const int m = e / CT, cc = e - m * CT;
        buf[cc * LS + m] = make_float2(0.f, 0.f);
    }
    // forward transform, decimation in frequency: stages outermost first, natural order in, digit-reversed out
    {
        int M = P;
        for (int st = 0; st < plan.nstages; ++st) {
            const int R = plan.radix[st];
            M /= R;
            __syncthreads();
            switch (R) {
                case 2: stage_reg_dif<2, false>(buf, tw, P, M, CT, LS, tid, 256); break;
                case 4: stage_reg_dif<4, false>(buf, tw, P, M, CT, LS, tid, 256); break;
                default: stage_reg_dif<8, false>(buf, tw, P, M, CT, LS, tid, 256); break;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < P * CT; e += 256) {
        const int cc = e / P, pos = e - cc * P;
        const float f = filt[pos];
        float2 v = buf[cc * LS + pos];
        buf[cc * LS + pos] = make_float2(v.x * f, v.y * f);
    }
    // inverse transform, decimation in time on the digit-reversed data (tile_fft starts and ends with a barrier)
    const float2* res = tile_fft<true>(plan, buf, buf, tw, CT, LS, tid, 256);
    const float inv = 1.0f / (float)P;
    for (int e = tid; e < N * CT; e += 256) {
        const int m = e / CT, cc = e - m * CT;
        const int a0 = 2 * (c0 + cc);
        if (c0 + cc < ncol) {
            const float2 v = res[cc * LS + m];
            dst[(int64_t)m * A + a0] = v.x * inv;
            if (a0 + 1 < A) dst[(int64_t)m * A + a0 + 1] = v.y * inv;
        }
    }
}

#endif  // DINV_RADON_PART != 1

// ---------------------------------------------------------------------------------------------------- host side
int make_geom(const dinv_radon_desc* d, TiledGeom* g, int* NBsel) {
    DINV_REQUIRE(d != nullptr, "null descriptor");
    DINV_REQUIRE(d->n_img >= 0 && d->width >= 2 && d->grid >= d->width && d->n_angles >= 1, "bad radon geometry");
    DINV_REQUIRE(d->pad_before >= 0 && d->pad_before + d->width <= d->grid, "bad padding");
    DINV_REQUIRE(d->grid <= MAXG, "detector count %d above the tiled kernels' limit %d", d->grid, MAXG);
    g->n_img = d->n_img; g->W = d->width; g->G = d->grid; g->pad = d->pad_before; g->A = d->n_angles;
    g->circle = d->circle; g->scale = d->scale;
    int NB = d->n_img >= 8 ? 8 : d->n_img >= 4 ? 4 : d->n_img >= 2 ? 2 : 1;
    *NBsel = NB;
    g->groups = d->n_img == 0 ? 0 : (d->n_img + NB - 1) / NB;
    g->njb = (d->grid + 63) / 64;
    g->nbands = (d->grid + 1 + BH - 1) / BH;
    g->PH = g->nbands * BH + 1;
    g->PW = d->grid + 2;
    g->WW = 0;
    g->row_magic = 0;
    return 0;
}

struct PlanLayout {
    int nchunks_max;
    size_t off_angles, off_dir, off_wtab, words;
};
PlanLayout plan_layout(int A, int njb, int nbands, int kw) {
    PlanLayout L;
    L.nchunks_max = (A + kw - 1) / kw + 4;   // at most one partly filled chunk per (class, direction)
    L.off_angles = 0;
    L.off_dir = L.off_angles + (size_t)L.nchunks_max * kw;
    L.off_wtab = L.off_dir + (size_t)L.nchunks_max;
    L.words = L.off_wtab + (size_t)L.nchunks_max * njb * nbands;
    return L;
}
constexpr int kSlackFloats = 4096;   // tail of the packed image: windows may run past the end of the last row

#define DINV_NB_DISPATCH(NBV, STMT)                                  \
    switch (NBV) {                                                   \
        case 8: { constexpr int NB = 8; STMT; } break;               \
        case 4: { constexpr int NB = 4; STMT; } break;               \
        case 2: { constexpr int NB = 2; STMT; } break;               \
        default: { constexpr int NB = 1; STMT; } break;              \
    }

template <class K>
int set_dyn_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(100 + (int)e, "hipFuncSetAttribute(lds=%zu): %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

}  // namespace

#if DINV_RADON_PART != 1
extern "C" size_t dinv_radon_plan_bytes(const dinv_radon_desc* d) {
    TiledGeom g;
    int NB;
    if (!d || make_geom(d, &g, &NB)) return 0;
    size_t words = 0;
    for (int kw = 1; kw <= KWMAX; kw *= 2) words = std::max(words, plan_layout(g.A, g.njb, g.nbands, kw).words);
    return words * sizeof(int32_t);
}

namespace {
// chunks of `kw` angles by (class, direction) and the window of every (chunk, ray block, band); returns the widest
int build_plan(const TiledGeom& g, const float* cs_host, int kw, int32_t* blob, int* nplain_out, int* nch_out) {
    const PlanLayout L = plan_layout(g.A, g.njb, g.nbands, kw);
    std::memset(blob, 0, L.words * sizeof(int32_t));
    // ---- chunks: PLAIN (|cos| >= |sin|) first, then SWAP; inside a class one run per marching direction
    std::vector<std::vector<int>> cls(4);   // 0: plain +, 1: plain -, 2: swap +, 3: swap -
    for (int a = 0; a < g.A; ++a) {
        const float c = cs_host[2 * a], s = cs_host[2 * a + 1];
        const bool swap = std::fabs(s) > std::fabs(c);
        const float slope = swap ? s : c;   // d(band coordinate)/di
        cls[(swap ? 2 : 0) + (slope > 0.f ? 0 : 1)].push_back(a);
    }
    int nch = 0, nplain = 0;
    for (int k = 0; k < 4; ++k) {
        for (size_t p = 0; p < cls[k].size(); p += kw) {
            for (int w = 0; w < kw; ++w)
                blob[L.off_angles + (size_t)nch * kw + w] = p + w < cls[k].size() ? cls[k][p + w] : -1;
            blob[L.off_dir + nch] = (k & 1) ? -1 : 1;
            ++nch;
        }
        if (k == 1) nplain = nch;
    }
    // ---- windows (fp64 geometry, one pixel of margin on each side for the fp32 sample positions)
    const double ctr = 0.5 * (g.G - 1);
    int worst = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool swap = ch >= nplain;
        for (int jb = 0; jb < g.njb; ++jb) {
            const int jlo = jb * 64, jhi = std::min(jb * 64 + 63, g.G - 1);
            for (int band = 0; band < g.nbands; ++band) {
                const double vlo = -1.0 + (double)band * BH, vhi = vlo + BH;
                double umin = 1e30, umax = -1e30;
                for (int w = 0; w < kw; ++w) {
                    const int a = blob[L.off_angles + (size_t)ch * kw + w];
                    if (a < 0) continue;
                    const double c = cs_host[2 * a], s = cs_host[2 * a + 1];
                    // along a ray: v(di) = ctr + vj*dj + sl*di, u(di) = ctr + uj*dj + ul*di   (v = band coordinate)
                    const double vj = swap ? c : -s, uj = swap ? -s : c;
                    const double sl = swap ? s : c, ul = swap ? c : s;
                    auto take = [&](double dj, double di) {
                        double u = ctr + uj * dj + ul * di;
                        u = std::min(std::max(u, -2.0), (double)g.G + 1.0);
                        umin = std::min(umin, u); umax = std::max(umax, u);
                    };
                    // the (ray, step) region of this band is a convex polygon: rays jlo..jhi, steps of the lattice
                    // widened by one, band coordinate in [vlo, vhi]; u is linear on it -> extremes at its vertices
                    for (int e = 0; e < 2; ++e) {
                        const double dj = (e ? jhi : jlo) - ctr;
                        double dlo = -ctr - 1.0, dhi = ctr + 1.0;
                        double b0 = (vlo - ctr - vj * dj) / sl, b1 = (vhi - ctr - vj * dj) / sl;
                        if (b0 > b1) std::swap(b0, b1);
                        dlo = std::max(dlo, b0); dhi = std::min(dhi, b1);
                        if (dlo > dhi) continue;   // this ray has no sample in the band
                        take(dj, dlo);
                        take(dj, dhi);
                    }
                    if (vj != 0.0)
                        for (double di : {-ctr - 1.0, ctr + 1.0})
                            for (double v : {vlo, vhi}) {
                                const double dj = (v - ctr - sl * di) / vj;
                                if (dj >= jlo - ctr && dj <= jhi - ctr) take(dj, di);
                            }
                }
                int wx0 = -1, ww = 2;
                if (umin <= umax) {
                    wx0 = std::max((int)std::floor(umin) - 1, -1);   // column -1 is the zero ring: nothing to its left
                    ww = (int)std::floor(umax) + 2 - wx0 + 1;
                    worst = std::max(worst, ww);
                }
                blob[L.off_wtab + ((size_t)ch * g.njb + jb) * g.nbands + band] = (ww << 16) | (wx0 + 8);
            }
        }
    }
    *nplain_out = nplain;
    *nch_out = nch;
    return worst;
}
}  // namespace

// Host-side plan.  cs_host: [A][2] fp32 (cos, sin) exactly as the device table holds them.  The number of angles per
// workgroup is the largest of 8, 4, 2, 1 whose widest window stays within WWPREF columns (angle lists with coarse or
// irregular spacing get fewer angles per workgroup; one angle always fits).  A caller may force the count by passing
// plan->kw = 1, 2, 4 or 8 on entry (0: automatic; the emulated CPU tests sweep it).
// Window pitch: a window cell (4 images = 16 bytes) of (row, col) sits in LDS bank slot (row * pitch + col) mod 16.  The 16
// lanes one ds_read_b128 cycle serves are 16 adjacent rays, i.e. cells (col0 + ~k cos, row0 - ~k sin): with pitch = 0 mod
// 16 the slot is col mod 16 and only two rays that share a column in different rows collide (shallow angles: none; 45
// degrees: two-way), every other residue also collides across rows (pitch = 8 mod 16: (col + 8, row + 1) with (col, row),
// two-way at nearly every angle).  Simulated LDS cycles per tap read averaged over the angles of a class: 1.46 at 0 mod 16,
// 1.91 at 8 (the round-2 pitch 136), 2.36 at 12 (the pitch 124 that config 3 got): so the pitch is rounded up to 16.
extern "C" int dinv_radon_plan_init(const dinv_radon_desc* d, const float* cs_host, dinv_radon_plan* plan,
                                    void* host_blob) {
    TiledGeom g;
    int NB;
    if (int e = make_geom(d, &g, &NB)) return e;
    DINV_REQUIRE(cs_host && plan && host_blob, "null pointer");
    int32_t* blob = reinterpret_cast<int32_t*>(host_blob);
    const int forced = plan->kw;
    std::memset(plan, 0, sizeof(*plan));
    int kw = KWMAX, worst = 0, nplain = 0, nch = 0;
    for (;; kw >>= 1) {
        if (forced == 1 || forced == 2 || forced == 4 || forced == 8) kw = forced;
        worst = build_plan(g, cs_host, kw, blob, &nplain, &nch);
        if (worst <= WWPREF || kw == 1 || forced) break;
    }
    plan->grid = g.G; plan->n_angles = g.A; plan->kw = kw; plan->band_h = BH;
    plan->win_w = std::max(16, (worst + 15) & ~15);
    plan->n_jblocks = g.njb; plan->n_bands = g.nbands; plan->n_chunks_plain = nplain; plan->n_chunks_swap = nch - nplain;
    plan->fits = 1;
    plan->blob_words = (int32_t)plan_layout(g.A, g.njb, g.nbands, kw).words;
    plan->widest_window = worst;
    return 0;
}

extern "C" size_t dinv_radon_tiled_workspace_bytes(const dinv_radon_desc* d, int32_t adjoint) {
    TiledGeom g;
    int NB;
    if (!d || make_geom(d, &g, &NB)) return 0;
    if (adjoint) return (size_t)g.groups * g.A * (g.G + 2 * JPAD) * NB * sizeof(float);
    return (2 * (size_t)g.groups * g.PH * g.PW * NB + kSlackFloats) * sizeof(float);
}

#endif  // DINV_RADON_PART != 1

#if DINV_RADON_PART != 2
#define DINV_FWD_LAUNCH(PF)                                                                                              \
    do {                                                                                                                 \
        if (int e = set_dyn_lds(radon_fwd_tiled_kernel<NB, false, PF>, lds)) return e;                                  \
        if (int e = set_dyn_lds(radon_fwd_tiled_kernel<NB, true, PF>, lds)) return e;                                   \
        if (plan->n_chunks_plain > 0)                                                                                    \
            hipLaunchKernelGGL((radon_fwd_tiled_kernel<NB, false, PF>), dim3(g.njb, plan->n_chunks_plain, g.groups),    \
                               dim3(64 * kw), lds, s, g, (const float*)xp, xn, cs2, blob + L.off_angles,                \
                               blob + L.off_dir, blob + L.off_wtab, norm_dev, sino, 0);                                  \
        if (plan->n_chunks_swap > 0)                                                                                     \
            hipLaunchKernelGGL((radon_fwd_tiled_kernel<NB, true, PF>), dim3(g.njb, plan->n_chunks_swap, g.groups),      \
                               dim3(64 * kw), lds, s, g, (const float*)xpt, xn, cs2, blob + L.off_angles,               \
                               blob + L.off_dir, blob + L.off_wtab, norm_dev, sino, plan->n_chunks_plain);               \
    } while (0)

extern "C" int dinv_radon_forward_tiled(const dinv_radon_desc* d, const dinv_radon_plan* plan, const void* plan_dev,
                                        const float* x, const float* xn, const float* cs, const float* norm_dev,
                                        float* sino, void* ws, size_t ws_bytes, dinv_stream_t stream) {
    TiledGeom g;
    int NBsel;
    if (int e = make_geom(d, &g, &NBsel)) return e;
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(plan && plan_dev && x && xn && cs && sino && ws, "null pointer");
    DINV_REQUIRE(plan->grid == g.G && plan->n_angles == g.A && plan->n_jblocks == g.njb && plan->n_bands == g.nbands,
                 "plan does not match the descriptor");
    DINV_REQUIRE(ws_bytes >= dinv_radon_tiled_workspace_bytes(d, 0), "workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int kw = plan->kw;
    DINV_REQUIRE(kw == 1 || kw == 2 || kw == 4 || kw == 8, "bad plan (kw = %d)", kw);
    const PlanLayout L = plan_layout(g.A, g.njb, g.nbands, kw);
    const int32_t* blob = reinterpret_cast<const int32_t*>(plan_dev);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    const int64_t npk = (int64_t)g.groups * g.PH * g.PW;
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    g.WW = plan->win_w;
    {
        const uint64_t rowlen = (uint64_t)g.WW * (NBsel >= 8 ? 2 : 1);
        g.row_magic = (uint32_t)((((uint64_t)1 << 32) + rowlen - 1) / rowlen);
    }
    DINV_REQUIRE(g.groups <= 65535 && plan->n_chunks_plain <= 65535 && plan->n_chunks_swap <= 65535, "grid too large");
    DINV_NB_DISPATCH(NBsel, {
        float* xp = reinterpret_cast<float*>(ws);
        float* xpt = xp + (size_t)g.groups * g.PH * g.PW * NB;
        const size_t lds = ((size_t)((g.G + 3) & ~3) + (size_t)(BH + 1) * g.WW * NB) * sizeof(float);
        DINV_REQUIRE(lds <= kMaxLdsBytes && (size_t)g.WW * NB <= kSlackFloats,
                     "window of %d columns x %d images does not fit the LDS: use dinv_radon_forward", g.WW, NB);
        hipLaunchKernelGGL(radon_pack_image2<NB>, dim3(pk_blocks), dim3(256), 0, s, g, x, xp, xpt);
        if (kw == 8) DINV_FWD_LAUNCH(9); else DINV_FWD_LAUNCH(18);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}

#endif  // DINV_RADON_PART != 2

#if DINV_RADON_PART != 1
extern "C" int dinv_radon_adjoint_tiled(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                                        const float* norm_dev, float* x, void* ws, size_t ws_bytes,
                                        dinv_stream_t stream) {
    TiledGeom g;
    int NBsel;
    if (int e = make_geom(d, &g, &NBsel)) return e;
    if (g.n_img == 0) return 0;
    DINV_REQUIRE(x && xn && cs && sino && ws, "null pointer");
    DINV_REQUIRE(ws_bytes >= dinv_radon_tiled_workspace_bytes(d, 1), "workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float2* cs2 = reinterpret_cast<const float2*>(cs);
    const int64_t npk = (int64_t)g.groups * g.A * (g.G + 2 * JPAD);
    const unsigned pk_blocks = (unsigned)std::min<int64_t>(ceil_div(npk, 256), 65535);
    DINV_REQUIRE(g.groups <= 65535, "too many images per call");
    DINV_NB_DISPATCH(NBsel, {
        float* sp = reinterpret_cast<float*>(ws);
        const size_t lds = ((size_t)((g.G + 3) & ~3) + (size_t)KA * JW * NB) * sizeof(float) + KA * sizeof(float2) +
                           KA * sizeof(int);
        if (int e = set_dyn_lds(radon_adj_tiled_kernel<NB>, lds)) return e;
        hipLaunchKernelGGL(radon_pack_sino2<NB>, dim3(pk_blocks), dim3(256), 0, s, g, sino, sp);
        hipLaunchKernelGGL(radon_adj_tiled_kernel<NB>, dim3((g.W + 15) / 16, (g.W + 15) / 16, g.groups), dim3(256), lds, s,
                           g, (const float*)sp, xn, cs2, norm_dev, x);
    });
    DINV_CHECK_LAUNCH();
    return 0;
}

// ---- ramp filter: P, the Fourier filter of the reference and its digit-reversed device table
extern "C" int32_t dinv_radon_ramp_padded_size(int32_t n_det) {
    // max(64, 2^ceil(log2(2 n)))  (radon.py:96-98)
    int32_t p = 64;
    while (p < 2 * n_det) p *= 2;
    return p;
}

// host_filter_out[P]: F[k] (k = 0..P-1, Hermitian-extended; F = 2 rfft(f), radon.py:151-162) at position perm[k] of the
// P-point plan.  fft_host_table: the table dinv_fft_plan_init wrote for length P.
extern "C" int dinv_radon_ramp_filter_init(int32_t P, const void* fft_host_table, float* host_filter_out) {
    DINV_REQUIRE(P >= 64 && (P & (P - 1)) == 0, "padded length must be a power of two >= 64");
    DINV_REQUIRE(fft_host_table && host_filter_out, "null pointer");
    const int* perm = reinterpret_cast<const int*>(reinterpret_cast<const float*>(fft_host_table) + 2 * (size_t)P);
    // f[0] = 1/4, f[odd m] = -1/(pi n_m)^2 with n = [1,3,..,P/2-1 | P/2-1,..,3,1]; the reference builds f in fp32
    // and takes torch.fft.rfft of it: restated in fp64 on the fp32 coefficients (the transform of 2048 terms in fp64
    // is exact to fp32 resolution)
    std::vector<double> f((size_t)P, 0.0);
    f[0] = 0.25;
    const float pi32 = 3.14159265358979323846f;
    for (int m = 1; m < P; m += 2) {
        const int idx = (m - 1) / 2;                       // index into n
        const int half = P / 4;                            // len(arange(1, P/2+1, 2))
        const float nm = idx < half ? (float)(2 * idx + 1) : (float)(P / 2 - 1 - 2 * (idx - half));
        const float t = pi32 * nm;
        f[m] = (double)(-1.0f / (t * t));
    }
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < P; ++k) {
        double re = 0.0;
        for (int m = 0; m < P; ++m) {
            if (f[m] == 0.0) continue;
            const int64_t r = ((int64_t)k * m) % P;
            re += f[m] * std::cos(two_pi * (double)r / (double)P);
        }
        host_filter_out[perm[k]] = (float)(2.0 * re);
    }
    return 0;
}

extern "C" int dinv_radon_ramp_fft(int32_t n_img, int32_t n_det, int32_t n_angles, int32_t P,
                                   const dinv_fft_plan* plan, const void* fft_table_dev, const float* filter_dev,
                                   const float* sino, float* out, dinv_stream_t stream) {
    DINV_REQUIRE(n_img >= 0 && n_det >= 1 && n_angles >= 1, "bad ramp geometry");
    if (n_img == 0) return 0;
    DINV_REQUIRE(plan && fft_table_dev && filter_dev && sino && out && sino != out, "null or aliased pointer");
    DINV_REQUIRE(P == dinv_radon_ramp_padded_size(n_det) && plan->n == P && plan->generic == 0, "plan does not match");
    DINV_REQUIRE(n_img <= 65535, "too many sinograms per call");
    int CT = 8;
    while (CT > 1 && ((size_t)P + (size_t)CT * (P + 1)) * sizeof(float2) > kMaxLdsBytes) CT >>= 1;
    const size_t lds = ((size_t)P + (size_t)CT * (P + 1)) * sizeof(float2);
    DINV_REQUIRE(lds <= kMaxLdsBytes, "detector axis too long for the in-LDS ramp filter (%d)", n_det);
    if (int e = set_dyn_lds(ramp_fft_kernel, lds)) return e;
    const int ncol = (n_angles + 1) / 2;
    hipLaunchKernelGGL(ramp_fft_kernel, dim3((ncol + CT - 1) / CT, n_img), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), n_img, n_det, n_angles, P, CT, *plan, fft_table_dev,
                       filter_dev, sino, out);
    DINV_CHECK_LAUNCH();
    return 0;
}
#endif  // DINV_RADON_PART != 1
