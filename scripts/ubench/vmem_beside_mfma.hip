// Microbenchmark for the open question of DESIGN.md 3.2 (Winograd bf16-split conv): what does one global_load_dwordx4 (1 KB per
// wave) cost the wave that issues it in the middle of a v_mfma_f32_32x32x16_bf16 stream?  The conv kernel loads its A fragments
// straight from L2 (12 loads per 72 MFMAs and wave) and builds its V stage from 12 more; diagnostic builds priced the first at a
// sixth and the second at a quarter of the kernel although neither bandwidth nor latency explains it.  This program measures
// the time per MFMA of a pure MFMA stream with L loads per 12 MFMAs interleaved, for one and two waves per SIMD and for an
// L1-, L2- and HBM-sized working set, loads waited for one iteration late (two register sets) so that their latency is hidden.
// build: hipcc -O3 --offload-arch=gfx950 vmem_beside_mfma.hip -o vmem_beside_mfma ; run: ./vmem_beside_mfma
// (first attempt on the last GPU seconds of round 3: exited after 1 s without output - not yet debugged on hardware)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compiler-counted forms (the first version issued the loads from inline asm with "=v" outputs: hipcc does not count such loads,
// reused their destination registers for addresses, and the late data faulted the kernel - cdna_hip_programming.md 5.7 item 1);
// every MFMA / load group is pinned by a scheduling barrier, the waits are the compiler's (first use one half-iteration later)
#define MFMA(ACC) do { ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, ACC, 0, 0, 0); } while (0)
#define LOAD(DST, PTR) do { DST = *(PTR); } while (0)

// L loads per 12 MFMAs, spread evenly; NT = 256 (one wave per SIMD) or 512 (two); LS = lane stride in 16-byte units (1: a wave
// reads 1 KB contiguous, like the conv's weight fragments; 4: 16 bytes out of every 64, like its four-pixel activation windows)
template <int L, int NT, int LS = 1>
__global__ __launch_bounds__(NT) void k(const u32x4* src, float* out, int iters, unsigned mask) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + 3 * i); }
    u32x4 s0[L > 0 ? L : 1], s1[L > 0 ? L : 1];
    unsigned sink = 0;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned idx = (blockIdx.x * 64 + wave * 8) * 64 * LS + lane * LS;      // in 16-byte units
    for (int i = 0; i < (L > 0 ? L : 1); ++i) s0[i] = s1[i] = u32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                MFMA(acc[m & 3]);
                if (L > 0 && (m * L) / 12 != ((m + 1) * L) / 12) {       // the (m*L/12)-th load of this iteration goes behind MFMA m
                    const int li = (m * L) / 12;
                    const u32x4* p = src + (idx & mask);
                    idx += 64 * 13 * LS + (LS > 1 ? 1 : 0);
                    if (half == 0) LOAD(s0[li], p); else LOAD(s1[li], p);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the loads of the PREVIOUS half are consumed here: at most this half's L loads stay in flight
            if (L > 0) {
#pragma unroll
                for (int i = 0; i < L; ++i) sink ^= half == 0 ? s1[i].x : s0[i].x;
            }
        }
    }
    float s = (float)sink;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int L, int NT, int LS = 1>
static void run(const u32x4* src, float* out, size_t span_bytes, const char* name) {
    const int iters = 4000, blocks = 256;      // one block per CU: 256 threads = one wave per SIMD, 512 = two
    const unsigned mask = (unsigned)(span_bytes / 16 - 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<L, NT, LS>), dim3(blocks), dim3(NT), 0, 0, src, out, 200, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<L, NT, LS>), dim3(blocks), dim3(NT), 0, 0, src, out, iters, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(err)); fflush(stdout); return; }
    // MFMAs per SIMD: waves per SIMD x iters x 12
    const double per_simd = (double)(NT / 256) * iters * 12;
    printf("{\"loads_per_12_mfma\": %d, \"block\": %d, \"lane_stride_bytes\": %d, \"span\": \"%s\", \"ms\": %.4f, \"ns_per_mfma_slot\": %.3f}\n",
           L, NT, 16 * LS, name, ms, ms * 1e6 / per_simd);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    u32x4* src; float* out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 512 * 512 * sizeof(float)) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    struct { size_t b; const char* n; } spans[] = {{(size_t)1 << 14, "16KB (L1)"}, {(size_t)1 << 21, "2MB (L2)"}, {(size_t)1 << 30, "1GB (HBM)"}};
    for (auto sp : spans) {
        run<0, 256>(src, out, sp.b, sp.n); run<1, 256>(src, out, sp.b, sp.n); run<2, 256>(src, out, sp.b, sp.n);
        run<4, 256>(src, out, sp.b, sp.n); run<6, 256>(src, out, sp.b, sp.n);
        run<0, 512>(src, out, sp.b, sp.n); run<2, 512>(src, out, sp.b, sp.n); run<4, 512>(src, out, sp.b, sp.n);
        run<2, 256, 4>(src, out, sp.b, sp.n); run<4, 256, 4>(src, out, sp.b, sp.n); run<2, 512, 4>(src, out, sp.b, sp.n); run<4, 512, 4>(src, out, sp.b, sp.n);
    }
    return 0;
}
