// Microbenchmark: do VALU / LDS instructions issued by the SAME wave hide in the shadow of v_mfma_f32_32x32x2_f32
// when there is one wave per SIMD?  build: hipcc -O3 --offload-arch=gfx950 mfma_coissue.hip -o mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, bool AGPR, int NT = 256>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 256, NT / 256))) void k(float* out, int iters) {
    __shared__ float lds[4096];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = i;
    __syncthreads();
    const float* lp = lds + (threadIdx.x & 63) * 4;
    float4 l0 = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(b));
#pragma unroll
            for (int l = 0; l < NL; ++l) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"((unsigned)(size_t)lp));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = l0.x;
    for (int i = 0; i < 8; ++i) s += f[i] + acc[i][0] + acc[i][7];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two values per lane and instruction) beside the fp32 MFMA: NV packed instructions per MFMA
template <int NV, bool FMA, int NT = 256>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 256, NT / 256))) void kp(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    f32x2 f[8], c = {1.0f, 0.5f};
    for (int i = 0; i < 8; ++i) f[i] = (f32x2){a + i, a - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f[v % 8]) : "v"(c));
                else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i].x + f[i].y + acc[i][0] + acc[i][7];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NV, bool FMA, int NT = 256>
void runp(const char* name, float* d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kp<NV, FMA, NT><<<blocks, NT>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kp<NV, FMA, NT><<<blocks, NT>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s NPK=%2d fma=%d waves/simd=%d : %.1f ns per MFMA per SIMD\n", name, NV, (int)FMA, NT / 256,
           ms * 1e6 / (iters * 8.0 * (NT / 256)));
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));
// same experiment with the bf16 matrix core (v_mfma_f32_32x32x16_bf16, 8 passes = 32 cycles): does VALU hide there?
template <int NV, int NT = 256>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 256, NT / 256))) void kb(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(0x3f80); }
    float f[8], c = 1.0f;
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + acc[i][0] + acc[i][7];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NV, int NT = 256>
void runb(const char* name, float* d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kb<NV, NT><<<blocks, NT>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kb<NV, NT><<<blocks, NT>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("bf16 %-22s NV=%2d waves/simd=%d : %.1f ns per MFMA per SIMD (32 cyc @2.4GHz = 13.3 ns)\n", name, NV, NT / 256,
           ms * 1e6 / (iters * 8.0 * (NT / 256)));
}

template <int NV, int NL, bool AGPR, int NT = 256>
void run(const char* name, float* d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NL, AGPR, NT><<<blocks, NT>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NL, AGPR, NT><<<blocks, NT>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s NV=%2d NL=%d agpr=%d waves/simd=%d : %.1f ns per MFMA per SIMD\n", name, NV, NL, (int)AGPR, NT / 256,
           ms * 1e6 / (iters * 8.0 * (NT / 256)));
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    run<0, 0, true>("mfma only", d);
    run<2, 0, true>("mfma + 2 valu", d);
    run<4, 0, true>("mfma + 4 valu", d);
    run<8, 0, true>("mfma + 8 valu", d);
    run<12, 0, true>("mfma + 12 valu", d);
    run<16, 0, true>("mfma + 16 valu", d);
    run<0, 1, true>("mfma + 1 ds_read_b128", d);
    run<4, 1, true>("mfma + 4 valu + 1 ds_read", d);
    run<0, 0, false>("mfma only (vgpr acc)", d);
    run<4, 0, false>("mfma + 4 valu (vgpr acc)", d);
    run<8, 0, false>("mfma + 8 valu (vgpr acc)", d);
    run<0, 0, true, 512>("2w: mfma only", d);
    run<4, 0, true, 512>("2w: mfma + 4 valu", d);
    run<8, 0, true, 512>("2w: mfma + 8 valu", d);
    run<12, 0, true, 512>("2w: mfma + 12 valu", d);
    run<16, 0, true, 512>("2w: mfma + 16 valu", d);
    run<8, 1, true, 512>("2w: mfma + 8 valu + 1 ds", d);
    run<8, 0, true, 1024>("4w: mfma + 8 valu", d);
    run<16, 0, true, 1024>("4w: mfma + 16 valu", d);
    runp<2, false>("mfma + 2 pk_add", d);
    runp<4, false>("mfma + 4 pk_add", d);
    runp<8, false>("mfma + 8 pk_add", d);
    runp<4, true>("mfma + 4 pk_fma", d);
    runp<2, false, 512>("2w: mfma + 2 pk_add", d);
    runp<4, false, 512>("2w: mfma + 4 pk_add", d);
    runp<8, false, 512>("2w: mfma + 8 pk_add", d);
    runp<4, true, 512>("2w: mfma + 4 pk_fma", d);
    runp<8, true, 512>("2w: mfma + 8 pk_fma", d);
    runb<0>("mfma only", d);
    runb<2>("mfma + 2 valu", d);
    runb<4>("mfma + 4 valu", d);
    runb<8>("mfma + 8 valu", d);
    runb<0, 512>("2w: mfma only", d);
    runb<4, 512>("2w: mfma + 4 valu", d);
    runb<8, 512>("2w: mfma + 8 valu", d);
    return 0;
}
