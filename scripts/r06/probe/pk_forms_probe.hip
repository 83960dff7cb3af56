// Which packed-fp32 instruction form loses results beside the bf16-split convolution?  (tail3x3_shift_kernel built with
// SLP-packed fp32 ops drops single terms in lanes 48..63 while conv3x3_wsplit_kernel runs on another stream.)
// Kernel A mimics the tail kernel's instruction mix: 16-byte global loads, v_mov_b32_dpp wave shifts, then ONE packed form per
// accumulator pair (the op_sel variants hipcc emitted there), each checked in the same lane against scalar v_fma / v_mul / v_add on
// the same inputs.  Partner = the product's own dinv_conv3x3_wsplit (dlopen of libdeepinv_amd.so) on a second stream.
//   hipcc -O2 --offload-arch=gfx950 pk_forms_probe.hip -o pk_forms_probe -ldl ;  ./pk_forms_probe path/to/libdeepinv_amd.so
#include <hip/hip_runtime.h>
#include "../../../include/deepinv_amd.h"
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NF = 10;

#define PK3(NAME, TXT)                                                                              \
    __device__ __forceinline__ f32x2 NAME(f32x2 a, f32x2 b, f32x2 c) {                              \
        f32x2 d;                                                                                    \
        asm volatile(TXT : "=v"(d) : "v"(a), "v"(b), "v"(c));                                       \
        return d;                                                                                   \
    }
#define PK2(NAME, TXT)                                                                              \
    __device__ __forceinline__ f32x2 NAME(f32x2 a, f32x2 b) {                                       \
        f32x2 d;                                                                                    \
        asm volatile(TXT : "=v"(d) : "v"(a), "v"(b));                                               \
        return d;                                                                                   \
    }
PK3(fma_101, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")
PK3(fma_011, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")
PK3(fma_pln, "v_pk_fma_f32 %0, %1, %2, %3")
PK3(fma_s100, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]")
PK3(fma_s001, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]")
PK3(fma_s010, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]")
PK2(mul_sw, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
PK2(mul_pln, "v_pk_mul_f32 %0, %1, %2")
PK2(add_pln, "v_pk_add_f32 %0, %1, %2")
PK2(mul_h01, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")

__device__ __forceinline__ float dpp_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float fm(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float ml(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float ad(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// bad[form][row of 16 lanes][half]
__global__ __launch_bounds__(256) void forms_kernel(const float4* __restrict__ x, const float4* __restrict__ w, int64_t n4, int iters,
                                                    unsigned* __restrict__ bad) {
    const int lane = threadIdx.x & 63, row = lane >> 4;
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) % n4;
    f32x2 acc[NF];
    float r0[NF], r1[NF];
    for (int f = 0; f < NF; ++f) { acc[f] = f32x2{0.f, 0.f}; r0[f] = 0.f; r1[f] = 0.f; }
    unsigned wrong[NF][2] = {};
    for (int it = 0; it < iters; ++it) {
        const float4 a = x[i], b = x[(i + 977) % n4];
        const float4 q = w[(it * 64 + (lane & 0)) & 1023];           // "weights": the same address in every lane
        i = (i + 4099) % n4;
        const float s0 = dpp_prev(a.x), s1 = dpp_prev(a.y);          // lane shifts feed the packed ops, as in the tail kernel
        const f32x2 u = {s0, s1}, v = {b.x, b.y}, ww = {q.x, q.y};
        // one packed form per accumulator; the scalar twin on the same inputs
        acc[0] = fma_101(u, ww, acc[0]);  r0[0] = fm(u.x, ww.x, r0[0]); r1[0] = fm(u.y, ww.x, r1[0]);
        acc[1] = fma_011(u, ww, acc[1]);  r0[1] = fm(u.x, ww.x, r0[1]); r1[1] = fm(u.x, ww.y, r1[1]);
        acc[2] = fma_pln(u, ww, acc[2]);  r0[2] = fm(u.x, ww.x, r0[2]); r1[2] = fm(u.y, ww.y, r1[2]);
        acc[3] = fma_s100(u, ww, acc[3]); r0[3] = fm(u.y, ww.x, r0[3]); r1[3] = fm(u.x, ww.y, r1[3]);
        { const f32x2 c = acc[4]; acc[4] = fma_s001(u, ww, c); const float t0 = fm(u.x, ww.x, c.y), t1 = fm(u.y, ww.y, c.x); r0[4] = t0; r1[4] = t1; }
        acc[5] = fma_s010(u, ww, acc[5]); r0[5] = fm(u.x, ww.y, r0[5]); r1[5] = fm(u.y, ww.x, r1[5]);
        { const f32x2 m = mul_sw(ww, u); acc[6] = fma_pln(v, ww, m); r0[6] = fm(v.x, ww.x, ml(ww.x, u.y)); r1[6] = fm(v.y, ww.y, ml(ww.y, u.x)); }
        { const f32x2 m = mul_pln(ww, u); acc[7] = add_pln(acc[7], m); r0[7] = ad(r0[7], ml(ww.x, u.x)); r1[7] = ad(r1[7], ml(ww.y, u.y)); }
        { const f32x2 m = add_pln(u, v); acc[8] = fma_101(m, ww, acc[8]); r0[8] = fm(ad(u.x, v.x), ww.x, r0[8]); r1[8] = fm(ad(u.y, v.y), ww.x, r1[8]); }
        { const f32x2 m = mul_h01(u, ww); acc[9] = add_pln(acc[9], m); r0[9] = ad(r0[9], ml(u.x, ww.x)); r1[9] = ad(r1[9], ml(u.x, ww.y)); }
        for (int f = 0; f < NF; ++f) {
            if (f == 6) { wrong[f][0] += acc[f].x != r0[f]; wrong[f][1] += acc[f].y != r1[f]; continue; }
            if (acc[f].x != r0[f]) { ++wrong[f][0]; acc[f].x = r0[f]; }      // count once, then follow the scalar chain again
            if (acc[f].y != r1[f]) { ++wrong[f][1]; acc[f].y = r1[f]; }
        }
    }
    for (int f = 0; f < NF; ++f)
        for (int h = 0; h < 2; ++h)
            if (wrong[f][h]) atomicAdd(bad + (f * 4 + row) * 2 + h, wrong[f][h]);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// synthetic partners (registers only unless said): 0 bf16 32x32x16 MFMA, 1 fp32 32x32x2 MFMA, 2 bf16 16x16x32 MFMA, 3 LDS reads, 4 fp32 FMAs
template <int KIND>
__global__ __launch_bounds__(256, 2) void partner_kernel(float* __restrict__ out, int n) {
    __shared__ float lds[4096];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i) & 7); b[i] = (__bf16)(float)((blockIdx.x - i) & 7); }
    f32x16 c0 = {0}, c1 = {0};
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 d0 = {0}, d1 = {0};
    float s = 0.f, t = 1.0f + threadIdx.x * 1e-6f;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); }
        if (KIND == 1) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t, s, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, t, c1, 0, 0, 0); }
        if (KIND == 2) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, d1, 0, 0, 0); }
        if (KIND == 3) { s += lds[(threadIdx.x * 4 + i) & 4095]; s += lds[(threadIdx.x * 8 + 3 * i) & 4095]; }
        if (KIND == 4) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s) : "v"(t)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(s)); }
        if (KIND == 5) { d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t, s, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, t, d1, 0, 0, 0); }
        if (KIND == 6) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            const s16x4 a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
            d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b4, a4, d1, 0, 0, 0);
        }
        if (KIND == 7) {
            typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
            h16x8 ha, hb;
            for (int q = 0; q < 8; ++q) { ha[q] = (_Float16)(float)a[q]; hb[q] = (_Float16)(float)b[q]; }
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, d1, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + t + d0[0] + d1[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "deepinv_amd/libdeepinv_amd.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    typedef int (*geom_init_t)(int32_t, int32_t, int32_t, void*);
    typedef int (*wsplit_t)(const void*, const void*, const void*, int32_t, int32_t, void*, const float*, int32_t, void*);
    geom_init_t geom_init = (geom_init_t)dlsym(lib, "dinv_act_geom_init");
    wsplit_t wsplit = (wsplit_t)dlsym(lib, "dinv_conv3x3_wsplit");
    if (!geom_init || !wsplit) { printf("symbols missing\n"); return 1; }
    dinv_act_geom geo;
    void* gbuf = &geo;
    if (geom_init(8, 256, 256, gbuf)) { printf("geom_init failed\n"); return 1; }
    const int64_t cs = geo.cs;
    const size_t act_bytes = (size_t)8 * cs * 8 * sizeof(float);          // 64 channels = 8 blocks
    float *xb, *rb, *yb; void* wb;
    CK(hipMalloc(&xb, act_bytes)); CK(hipMalloc(&rb, act_bytes)); CK(hipMalloc(&yb, act_bytes));
    const size_t wbytes = (size_t)64 * 64 * 3 * 4 * 2 * 2;                // [1][4][3][4][2][2][64][8] bf16
    CK(hipMalloc(&wb, wbytes * 2));
    {
        std::vector<float> h(act_bytes / 4);
        srand(3);
        for (float& v : h) v = (rand() % 2001 - 1000) / 1000.f;
        CK(hipMemcpy(xb, h.data(), act_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(rb, h.data(), act_bytes, hipMemcpyHostToDevice));
        std::vector<uint16_t> hw(wbytes);
        for (uint16_t& v : hw) v = (uint16_t)(0x3c00 + rand() % 512) | (rand() & 1 ? 0x8000 : 0);     // bf16 around +-0.01
        CK(hipMemcpy(wb, hw.data(), wbytes * 2, hipMemcpyHostToDevice));
    }
    const int64_t n4 = 1 << 22;
    float4 *x, *w; unsigned* bad;
    CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&w, 1024 * 16)); CK(hipMalloc(&bad, NF * 8 * 4));
    {
        std::vector<float> h(n4 * 4);
        for (float& v : h) v = (rand() % 2001 - 1000) / 1000.f;
        CK(hipMemcpy(x, h.data(), n4 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), 1024 * 16, hipMemcpyHostToDevice));
    }
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    const char* names[NF] = {"fma op_sel_hi:[1,0,1]", "fma op_sel_hi:[0,1,1]", "fma plain", "fma op_sel:[1,0,0] hi:[0,1,1]", "fma op_sel:[0,0,1] hi:[1,1,0]",
                             "fma op_sel:[0,1,0] hi:[1,0,1]", "mul op_sel:[0,1] hi:[1,0] -> fma", "mul plain -> add", "add -> fma", "mul op_sel_hi:[0,1] -> add"};
    const char* pnames[10] = {"none", "conv3x3_wsplit (the product's kernel)", "loop of v_mfma_f32_32x32x16_bf16", "loop of v_mfma_f32_32x32x2_f32",
                              "loop of v_mfma_f32_16x16x32_bf16", "loop of LDS reads", "loop of v_fma_f32", "loop of v_mfma_f32_16x16x4_f32",
                              "loop of v_mfma_f32_16x16x16_bf16", "loop of v_mfma_f32_16x16x32_f16"};
    float* pout; CK(hipMalloc(&pout, 2048 * 256 * 4));
    for (int partner = 0; partner < 10; ++partner) {
        CK(hipMemset(bad, 0, NF * 8 * 4)); CK(hipDeviceSynchronize());
        auto go = [&]() {
            switch (partner) {
                case 1: for (int k = 0; k < 2; ++k) wsplit(gbuf, xb, wb, 64, 64, yb, rb, 0, sb); break;
                case 2: hipLaunchKernelGGL(partner_kernel<0>, dim3(2048), dim3(256), 0, sb, pout, 3000); break;
                case 3: hipLaunchKernelGGL(partner_kernel<1>, dim3(2048), dim3(256), 0, sb, pout, 1500); break;
                case 4: hipLaunchKernelGGL(partner_kernel<2>, dim3(2048), dim3(256), 0, sb, pout, 6000); break;
                case 5: hipLaunchKernelGGL(partner_kernel<3>, dim3(2048), dim3(256), 0, sb, pout, 6000); break;
                case 6: hipLaunchKernelGGL(partner_kernel<4>, dim3(2048), dim3(256), 0, sb, pout, 20000); break;
                case 7: hipLaunchKernelGGL(partner_kernel<5>, dim3(2048), dim3(256), 0, sb, pout, 3000); break;
                case 8: hipLaunchKernelGGL(partner_kernel<6>, dim3(2048), dim3(256), 0, sb, pout, 6000); break;
                case 9: hipLaunchKernelGGL(partner_kernel<7>, dim3(2048), dim3(256), 0, sb, pout, 6000); break;
                default: break;
            }
        };
        for (int rep = 0; rep < 30; ++rep) {
            go();
            hipLaunchKernelGGL(forms_kernel, dim3(2560), dim3(256), 0, sa, x, w, n4, 400, bad);
            go();
            CK(hipDeviceSynchronize());
        }
        unsigned h[NF * 8];
        CK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
        for (int f = 0; f < NF; ++f) {
            unsigned tot = 0;
            for (int k = 0; k < 8; ++k) tot += h[f * 8 + k];
            if (partner > 1 && tot == 0) continue;                       // print the clean forms for "none" and the product's kernel only
            printf("{\"partner\": \"%s\", \"form\": \"%s\", \"mismatches_by_lane_row_lo_hi\": [", pnames[partner], names[f]);
            for (int r = 0; r < 4; ++r) printf("[%u, %u]%s", h[(f * 4 + r) * 2], h[(f * 4 + r) * 2 + 1], r < 3 ? ", " : "");
            printf("]}\n");
        }
        unsigned tot = 0;
        for (unsigned v : h) tot += v;
        if (partner > 1 && tot == 0) printf("{\"partner\": \"%s\", \"all_forms_clean\": true}\n", pnames[partner]);
    }
    return 0;
}
