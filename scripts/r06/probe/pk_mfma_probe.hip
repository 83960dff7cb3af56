// Does a packed fp32 FMA chain lose results when bf16 MFMA waves of ANOTHER kernel share the SIMDs?  (scripts/r06/race_hunt8.py: the
// tail convolution's outputs differ run to run beside a bf16-split launch - the low half of packed results, lanes 48..63.)
// Kernel A: every lane runs the same chain twice, once with v_pk_fma_f32 and once with two v_fma_f32, and counts mismatches per
// lane.  Kernel B: a register-only loop of v_mfma_f32_32x32x16_bf16.  A runs alone, then beside B on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void pk_chain(const float* __restrict__ u, const float* __restrict__ w, int n, int mode,
                                                unsigned* __restrict__ bad_lane, unsigned* __restrict__ bad_half) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < n; ++i) {
        f32x2 uu = {u[(gid + i) & 4095], u[(gid + 7 * i + 1) & 4095]};
        const f32x2 ws = {w[i & 255], w[(i & 255) + 1]};   // uniform: scalar loads
        float wvv = w[(i & 255) + (lane & 0)];
        if (mode == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(uu), "s"(ws));
        else {
            f32x2 wp = {wvv, wvv};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(uu), "v"(wp));
        }
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s0) : "v"(uu.x), "v"(wvv));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s1) : "v"(uu.y), "v"(wvv));
    }
    if (acc.x != s0) { atomicAdd(bad_lane + lane, 1u); atomicAdd(bad_half + 0, 1u); }
    if (acc.y != s1) { atomicAdd(bad_lane + lane, 1u); atomicAdd(bad_half + 1, 1u); }
}

__global__ __launch_bounds__(256, 2) void mfma_loop(float* __restrict__ out, int n) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x - i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    float *u, *w, *out; unsigned *bad_lane, *bad_half;
    CK(hipMalloc(&u, 4096 * 4)); CK(hipMalloc(&w, 512 * 4)); CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMalloc(&bad_lane, 64 * 4)); CK(hipMalloc(&bad_half, 2 * 4));
    float hu[4096], hw[512];
    srand(1);
    for (float& v : hu) v = (rand() % 2001 - 1000) / 1000.f;
    for (float& v : hw) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(u, hu, sizeof hu, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    for (int mode = 0; mode < 2; ++mode)
        for (int partner = 0; partner < 2; ++partner) {
            CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_half, 0, 8));
            CK(hipDeviceSynchronize());
            for (int rep = 0; rep < 20; ++rep) {
                if (partner) hipLaunchKernelGGL(mfma_loop, dim3(2048), dim3(256), 0, sb, out, 4000);
                hipLaunchKernelGGL(pk_chain, dim3(4096), dim3(256), 0, sa, u, w, 4000, mode, bad_lane, bad_half);
                if (partner) hipLaunchKernelGGL(mfma_loop, dim3(2048), dim3(256), 0, sb, out, 4000);
                CK(hipDeviceSynchronize());
            }
            unsigned hl[64], hh[2];
            CK(hipMemcpy(hl, bad_lane, sizeof hl, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh, bad_half, sizeof hh, hipMemcpyDeviceToHost));
            unsigned rows[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) rows[l / 16] += hl[l];
            printf("{\"pk_source\": \"%s\", \"partner\": \"%s\", \"mismatches_lanes_0_15\": %u, \"16_31\": %u, \"32_47\": %u, \"48_63\": %u, \"low_half\": %u, \"high_half\": %u}\n",
                   mode == 0 ? "sgpr" : "vgpr", partner ? "bf16 mfma loop" : "none", rows[0], rows[1], rows[2], rows[3], hh[0], hh[1]);
        }
    return 0;
}
