// TEST INFRASTRUCTURE - a single-OS-thread emulation of the HIP execution model, just large enough to run the
// product's kernel SOURCES (deepinv_amd/csrc/*.hip) on the host in the GPU-less build container:
//   * every thread of a workgroup is a ucontext fiber; __syncthreads() and the wave-level exchange primitives are
//     cooperative barriers; workgroups run one after the other (so `__shared__` objects are plain statics);
//   * a wavefront is 64 consecutive linear thread ids, exactly as on gfx950;
//   * MFMA builtins are emulated from the documented fragment layouts (cdna_hip_programming.md section 3).
// Nothing under deepinv_amd/ includes this file: the product is compiled by hipcc against the real runtime.  The tests
// build tests/emu/libdeepinv_amd_emu.so from the same sources with `-I tests/emu/include` and call its C-ABI entry
// points on host buffers (tests/test_emu_*.py).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define DINV_EMU 1
#ifndef __HIP_DEVICE_COMPILE__
#define __HIP_DEVICE_COMPILE__ 1
#endif
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static

// ------------------------------------------------------------------ vector types
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ------------------------------------------------------------------ runtime stubs
typedef struct emu_stream_* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetDeviceCount(int* c) { *c = 0; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 0;
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 256; };
constexpr int hipDeviceAttributeMultiprocessorCount = 0;
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 16; return hipSuccess; }   // 2 "CUs" per XCD: short tile lists per workgroup
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }

// ------------------------------------------------------------------ the fiber scheduler
namespace emu {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = true;
    dim3 tid;
};
struct Barrier { int count = 0; unsigned gen = 0; };
struct State {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::function<void()> body;
    int cur = -1, nthreads = 0, alive = 0;
    Barrier block_bar;
    std::vector<Barrier> wave_bar;
    std::vector<int> wave_alive;
    std::vector<std::vector<uint64_t>> wave_buf;   // per wave: scratch for cross-lane exchange (up to 64 x 64 words)
    std::vector<unsigned char> dynsmem;
};
inline State g;
}  // namespace emu
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
inline int linear_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }
inline void yield() {
    Fiber& f = g.fibers[g.cur];
    swapcontext(&f.ctx, &g.sched);
}
inline void trampoline() {
    g.body();
    Fiber& f = g.fibers[g.cur];
    f.done = true;
    --g.alive;
    --g.wave_alive[g.cur / 64];
    swapcontext(&f.ctx, &g.sched);
}
// cooperative barrier over a group whose number of live members is *expected (re-read while waiting: a member that
// returned from the kernel no longer takes part, like a terminated wave on the hardware)
inline void barrier_wait(Barrier& b, const int* expected) {
    const unsigned gen = b.gen;
    ++b.count;
    for (;;) {
        if (b.gen != gen) return;
        if (b.count >= *expected) { b.count = 0; ++b.gen; return; }
        yield();
    }
}
inline void run_block(const std::function<void()>& body, dim3 block) {
    const int n = (int)(block.x * block.y * block.z);
    if ((int)g.fibers.size() < n) g.fibers.resize(n);
    g.nthreads = g.alive = n;
    g.body = body;
    g.block_bar = Barrier();
    const int nw = (n + 63) / 64;
    g.wave_bar.assign(nw, Barrier());
    g.wave_alive.assign(nw, 0);
    g.wave_buf.assign(nw, std::vector<uint64_t>(64 * 64));
    for (int t = 0; t < n; ++t) {
        Fiber& f = g.fibers[t];
        if (!f.stack) f.stack = (char*)malloc(kStack);
        f.done = false;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        ++g.wave_alive[t / 64];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    long spins = 0;
    while (g.alive > 0) {
        const int before = g.alive;
        const unsigned gen0 = g.block_bar.gen;
        for (int t = 0; t < n; ++t) {
            Fiber& f = g.fibers[t];
            if (f.done) continue;
            g.cur = t;
            threadIdx = f.tid;
            swapcontext(&g.sched, &f.ctx);
        }
        // a released barrier whose waiters have not been resumed yet shows up as a generation change
        if (g.alive == before && g.block_bar.gen == gen0) {
            // maybe a barrier whose last expected member exited: release it
            if (g.block_bar.count > 0 && g.block_bar.count >= g.alive) { g.block_bar.count = 0; ++g.block_bar.gen; continue; }
            bool released = false;
            for (size_t w = 0; w < g.wave_bar.size(); ++w)
                if (g.wave_bar[w].count > 0 && g.wave_bar[w].count >= g.wave_alive[w]) {
                    g.wave_bar[w].count = 0; ++g.wave_bar[w].gen; released = true;
                }
            if (released) continue;
            if (++spins > 1000000) { fprintf(stderr, "hip_emu: deadlock (barrier never completes)\n"); abort(); }
        } else spins = 0;
    }
}
template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    gridDim = grid;
    blockDim = block;
    if (g.dynsmem.size() < shmem + 64) g.dynsmem.resize(shmem + 64);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                run_block([&] { kernel(args...); }, block);
            }
}
inline void yield_if_fiber() {}     // workgroups run one after the other: a spin on another workgroup's flag can only time out
inline int lane() { return linear_tid() & 63; }
inline int wave() { return linear_tid() >> 6; }
// all live lanes of the wave deposit `words` 64-bit words, synchronise, and may then read anybody's
inline uint64_t* wave_exchange_begin(const uint64_t* mine, int words) {
    const int w = wave(), l = lane();
    std::vector<uint64_t>& buf = g.wave_buf[w];
    for (int i = 0; i < words; ++i) buf[(size_t)l * 64 + i] = mine[i];
    barrier_wait(g.wave_bar[w], &g.wave_alive[w]);
    return buf.data();
}
// ordering point of a wave's LDS traffic (the product's wave_lds_sync): on the hardware the lanes of a wave run in lock step,
// here its fibers meet at a barrier
inline void wave_sync() {
    const int w = wave();
    barrier_wait(g.wave_bar[w], &g.wave_alive[w]);
}
inline void wave_exchange_end() {
    const int w = wave();
    barrier_wait(g.wave_bar[w], &g.wave_alive[w]);
}
}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::emu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)
// dynamic LDS: `extern __shared__ T name[];` on the device
#define DINV_DYN_LDS(T, name) T* name = reinterpret_cast<T*>(::emu::g.dynsmem.data())

inline void __syncthreads() { emu::barrier_wait(emu::g.block_bar, &emu::g.alive); }
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

// ------------------------------------------------------------------ cross-lane
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    uint64_t mine = 0;
    std::memcpy(&mine, &v, sizeof(T));
    uint64_t* buf = emu::wave_exchange_begin(&mine, 1);
    const int l = emu::lane();
    const int s = (l & ~(width - 1)) | (src & (width - 1));
    T r;
    std::memcpy(&r, &buf[(size_t)s * 64], sizeof(T));
    emu::wave_exchange_end();
    return r;
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, emu::lane() ^ m, width); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = emu::lane();
    return __shfl(v, ((l & (width - 1)) + (int)d < width) ? l + (int)d : l, width);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = emu::lane();
    return __shfl(v, ((l & (width - 1)) >= (int)d) ? l - (int)d : l, width);
}
inline int emu_readfirstlane(int v) { return __shfl(v, 0, 64); }
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane(x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ::emu::yield_if_fiber()
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4      // (the __hip_atomic_* builtins themselves are clang's, also on the host)
#endif
// buffer resources (range-checked loads / stores: offsets >= num_records are dropped / read as zero)
struct __amdgpu_buffer_rsrc_t { char* base; unsigned num; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num, int) { return {(char*)p, (unsigned)num}; }
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned off, int soff, int) {
    emu_u32x4 v = {0, 0, 0, 0};
    if ((uint64_t)off + soff + 16 <= r.num) std::memcpy(&v, r.base + off + soff, 16);
    return v;
}
typedef unsigned int emu_u32x2 __attribute__((ext_vector_type(2)));
inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, unsigned off, int soff, int) {
    emu_u32x2 v = {0, 0};
    if ((uint64_t)off + soff + 8 <= r.num) std::memcpy(&v, r.base + off + soff, 8);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off, int soff, int) {
    if ((uint64_t)off + soff + 16 <= r.num) std::memcpy(r.base + off + soff, &v, 16);
}
#define __builtin_amdgcn_wavefrontsize() 64

// ------------------------------------------------------------------ math / bit helpers
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
using std::min;
using std::max;
inline float fminf_(float a, float b) { return a < b ? a : b; }

// ------------------------------------------------------------------ MFMA (fragment layouts: cdna_hip_programming.md §3)
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));
inline float emu_bf16_to_f32(short h) { return __uint_as_float(((unsigned)(unsigned short)h) << 16); }
// D[i][j] += sum_k A[i][k] B[k][j];  lane l holds A[i = l&31][k = 8(l>>5) .. +7], B[k = 8(l>>5) .. +7][j = l&31],
// C/D element r of lane l is row (r&3) + 8(r>>2) + 4(l>>5), column l&31
inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
    uint64_t mine[4];
    std::memcpy(&mine[0], &a, 16);
    std::memcpy(&mine[2], &b, 16);
    uint64_t* buf = emu::wave_exchange_begin(mine, 4);
    const int l = emu::lane(), j = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int h = 0; h < 2; ++h) {
            short av[8], bv[8];
            std::memcpy(av, &buf[(size_t)(i + 32 * h) * 64], 16);
            std::memcpy(bv, &buf[(size_t)(j + 32 * h) * 64 + 2], 16);
            for (int k = 0; k < 8; ++k) acc = std::fmaf(emu_bf16_to_f32(av[k]), emu_bf16_to_f32(bv[k]), acc);
        }
        d[r] = acc;
    }
    emu::wave_exchange_end();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_f32_32x32x16_bf16(a, b, c)
// f32 32x32x2: lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
    uint64_t mine[1];
    float ab[2] = {a, b};
    std::memcpy(mine, ab, 8);
    uint64_t* buf = emu::wave_exchange_begin(mine, 1);
    const int l = emu::lane(), j = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int k = 0; k < 2; ++k) {
            float pa[2], pb[2];
            std::memcpy(pa, &buf[(size_t)(i + 32 * k) * 64], 8);
            std::memcpy(pb, &buf[(size_t)(j + 32 * k) * 64], 8);
            acc = std::fmaf(pa[0], pb[1], acc);
        }
        d[r] = acc;
    }
    emu::wave_exchange_end();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32(a, b, c)
// f32 16x16x4: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D element r of lane l is row 4(l>>4) + r, column l&15
inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
    uint64_t mine[1];
    float ab[2] = {a, b};
    std::memcpy(mine, ab, 8);
    uint64_t* buf = emu::wave_exchange_begin(mine, 1);
    const int l = emu::lane(), j = l & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = d[r];
        for (int k = 0; k < 4; ++k) {
            float pa[2], pb[2];
            std::memcpy(pa, &buf[(size_t)(i + 16 * k) * 64], 8);
            std::memcpy(pb, &buf[(size_t)(j + 16 * k) * 64], 8);
            acc = std::fmaf(pa[0], pb[1], acc);
        }
        d[r] = acc;
    }
    emu::wave_exchange_end();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32(a, b, c)
