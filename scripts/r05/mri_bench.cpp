// Stand-alone A/B timing of the MultiCoilMRI entry points through the C-ABI (no torch: a fresh GPU box spends 1-2 minutes importing
// it).  Every argument is a build of libdeepinv_amd.so (the product or a variant compiled with other -D switches); each is
// dlopen'ed and runs dinv_mri_forward / _adjoint / _normal on the SAME seeded problem of BASELINE config 2 (32 slices, 8 coils,
// 320x320) and of config 4 (2 volumes, 12 coils, 16x256x256): ms per call (HIP events over `reps` back-to-back calls) and the
// relative l2 distance of every output to the first library's (0 = bit-identical).
//   hipcc -O2 --offload-arch=gfx950 scripts/r05/mri_bench.cpp -Iinclude -ldl -o scripts/r05/mri_bench
//   scripts/r05/mri_bench deepinv_amd/libdeepinv_amd.so scripts/r05/variants/libv1.so ...
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "deepinv_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Lib {
    void* h;
    decltype(&dinv_fft_table_bytes) table_bytes;
    decltype(&dinv_fft_plan_init) plan_init;
    decltype(&dinv_mri_workspace_bytes) ws_bytes;
    decltype(&dinv_mri_forward) fwd;
    decltype(&dinv_mri_adjoint) adj;
    decltype(&dinv_mri_normal) nrm;
    decltype(&dinv_last_error) err;
};

static Lib open_lib(const char* path) {
    Lib l{};
    l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { printf("dlopen %s: %s\n", path, dlerror()); exit(1); }
#define SYM(f, n) l.f = (decltype(l.f))dlsym(l.h, n); if (!l.f) { printf("missing %s in %s\n", n, path); exit(1); }
    SYM(table_bytes, "dinv_fft_table_bytes") SYM(plan_init, "dinv_fft_plan_init") SYM(ws_bytes, "dinv_mri_workspace_bytes")
    SYM(fwd, "dinv_mri_forward") SYM(adj, "dinv_mri_adjoint") SYM(nrm, "dinv_mri_normal") SYM(err, "dinv_last_error")
    return l;
}

struct Problem { int B, coils, nd, dims[3]; const char* name; };

static double rel(const std::vector<float>& a, const std::vector<float>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { const double d = (double)a[i] - b[i]; num += d * d; den += (double)b[i] * b[i]; }
    return std::sqrt(num / (den > 0 ? den : 1));
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: mri_bench lib.so [lib2.so ...] [--reps N]\n"); return 1; }
    int reps = 30;
    std::vector<std::string> libs;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else libs.push_back(argv[i]);
    }
    const Problem probs[] = {{32, 8, 2, {320, 320, 0}, "cfg2 32x8x320x320"}, {2, 12, 3, {16, 256, 256}, "cfg4 2x12x16x256x256"},
                             {4, 8, 2, {320, 320, 0}, "cfg2/8 4x8x320x320"}};
    for (const Problem& pr : probs) {
        int64_t vol = 1;
        for (int i = 0; i < pr.nd; ++i) vol *= pr.dims[i];
        const size_t nx = (size_t)pr.B * 2 * vol, ny = (size_t)pr.B * 2 * pr.coils * vol, nm = (size_t)pr.coils * vol * 2, nk = 2 * (size_t)vol;
        std::mt19937 rng(7);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> hx(nx), hy(ny), hmaps(nm), hmask(nk);
        for (auto& v : hx) v = nd(rng);
        for (auto& v : hy) v = nd(rng);
        for (auto& v : hmaps) v = nd(rng) * 0.35f;
        for (int64_t i = 0; i < vol; ++i) hmask[i] = hmask[vol + i] = (rng() % 4 == 0) ? 1.f : 0.f;
        float *dx, *dy, *dmaps, *dmask, *dout_x, *dout_y;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dmaps, nm * 4)); CK(hipMalloc(&dmask, nk * 4));
        CK(hipMalloc(&dout_x, nx * 4)); CK(hipMalloc(&dout_y, ny * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dmaps, hmaps.data(), nm * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dmask, hmask.data(), nk * 4, hipMemcpyHostToDevice));
        std::vector<float> ref_f, ref_a, ref_n;
        for (size_t li = 0; li < libs.size(); ++li) {
            Lib l = open_lib(libs[li].c_str());
            dinv_mri_desc d{};
            d.batch = pr.B; d.coils = pr.coils; d.ndim = pr.nd; d.mask_batch = 1; d.maps_batch = 1; d.coil_dim = 1;
            std::vector<void*> tabs;
            for (int i = 0; i < pr.nd; ++i) {
                d.dims[i] = pr.dims[i];
                const size_t tb = l.table_bytes(pr.dims[i]);
                std::vector<unsigned char> host(tb);
                if (l.plan_init(pr.dims[i], &d.plan[i], host.data())) { printf("plan_init: %s\n", l.err()); return 1; }
                void* t;
                CK(hipMalloc(&t, tb)); CK(hipMemcpy(t, host.data(), tb, hipMemcpyHostToDevice));
                d.table[i] = t; tabs.push_back(t);
            }
            const size_t wsb = l.ws_bytes(&d);
            void* ws;
            CK(hipMalloc(&ws, wsb));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto timeit = [&](auto fn) {
                for (int i = 0; i < 5; ++i) if (fn()) { printf("dinv error: %s\n", l.err()); exit(1); }
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) fn();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return ms / reps;
            };
            const float tf = timeit([&] { return l.fwd(&d, dx, dmaps, dmask, dout_y, ws, wsb, nullptr); });
            std::vector<float> of(ny); CK(hipMemcpy(of.data(), dout_y, ny * 4, hipMemcpyDeviceToHost));
            const float ta = timeit([&] { return l.adj(&d, dy, dmaps, dmask, dout_x, ws, wsb, nullptr); });
            std::vector<float> oa(nx); CK(hipMemcpy(oa.data(), dout_x, nx * 4, hipMemcpyDeviceToHost));
            const float tn = timeit([&] { return l.nrm(&d, dx, dmaps, dmask, dout_x, ws, wsb, nullptr); });
            std::vector<float> on(nx); CK(hipMemcpy(on.data(), dout_x, nx * 4, hipMemcpyDeviceToHost));
            if (li == 0) { ref_f = of; ref_a = oa; ref_n = on; }
            const double alg = ((double)nx + ny) * 4 + nm * 4 + nk * 4;
            printf("{\"problem\": \"%s\", \"lib\": \"%s\", \"A_ms\": %.4f, \"AT_ms\": %.4f, \"ATA_ms\": %.4f, \"A_GBps\": %.0f, \"AT_GBps\": %.0f, "
                   "\"diff_A\": %.3g, \"diff_AT\": %.3g, \"diff_ATA\": %.3g}\n", pr.name, libs[li].c_str(), tf, ta, tn, alg / tf / 1e6, alg / ta / 1e6,
                   rel(of, ref_f), rel(oa, ref_a), rel(on, ref_n));
            fflush(stdout);
            CK(hipFree(ws));
            for (void* t : tabs) CK(hipFree(t));
            // (the library stays loaded: kernels of a dlclose'd code object may still be referenced by the runtime)
        }
        CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dmaps)); CK(hipFree(dmask)); CK(hipFree(dout_x)); CK(hipFree(dout_y));
    }
    return 0;
}
