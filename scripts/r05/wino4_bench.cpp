// (round 5: + dinv_conv3x3_winograd4_bf16x3, the same kernel with the multiplies as a three-part bf16 split on the bf16 matrix cores)
// Stand-alone check + timing of the fp32 Winograd kernels through the C-ABI (no torch: a fresh GPU box spends 1-2 minutes
// importing it).  For every DRUNet level of BASELINE config 2 (B slices): dinv_conv3x3_winograd4 (F(4x4,3x3)) against
// dinv_conv3x3_winograd (F(2x2,3x3)) and against an fp64 evaluation of the defining sum at sampled outputs; then ms per launch
// of both kernels in the three epilogue forms a ResBlock uses (plain, ReLU, + residual).
//   hipcc -O2 --offload-arch=gfx950 scripts/r05/wino4_bench.cpp -Iinclude -Ldeepinv_amd -ldeepinv_amd -o gpurun_out/wino4_bench
//   LD_LIBRARY_PATH=deepinv_amd gpurun_out/wino4_bench [batch]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <random>
#include <vector>
#include "deepinv_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define DK(x) do { int r_ = (x); if (r_) { printf("dinv error %d: %s (%s:%d)\n", r_, dinv_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static const double BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
static const double G4[6][3] = {{0.25, 0, 0}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6}, {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0, 0, 1}};
// Winograd point (6 * row + col) of each point slot of the packed U (deepinv_amd/hip/drunet.py: WINOGRAD4_POINT_SLOTS)
static const int SLOT4[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17, 9, 10, 11, 18, 19, 20, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35, 27, 28, 29};
static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};

#ifdef TIMING
extern "C" void dinv_debug_wino4_timing(long long* p);   // timing build of drunet_wino4.hip linked into this executable
#endif

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int lvmask = argc > 3 ? atoi(argv[3]) : 15;
    const bool quick = argc > 4;      // timing of the F(4x4) kernel only (diagnostic builds: the results are not meaningful)
    int ndev = 0;
    DK(dinv_device_count(&ndev));
    printf("{\"devices\": %d, \"abi\": %d, \"batch\": %d}\n", ndev, dinv_version(), B);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const int levels[4][2] = {{320, 64}, {160, 128}, {80, 256}, {40, 512}};
    void* dws = nullptr;                 // workspace of the tail split (argv[5] = "nosplit": none)
    size_t wsb = 0;
    if (!(argc > 5)) {
        wsb = dinv_conv3x3_winograd4_workspace_bytes();
        CK(hipMalloc(&dws, wsb));
        CK(hipMemset(dws, 0, wsb));
    }
    for (int lv = 0; lv < 4; ++lv) {
        if (!((lvmask >> lv) & 1)) continue;
        const int H = levels[lv][0], W = H, C = levels[lv][1];
        dinv_act_geom g;
        DK(dinv_act_geom_init(B, H, W, &g));
        const size_t nact = (size_t)(C / 8) * g.cs * 8;
        std::vector<float> hx(nact, 0.f), hr(nact, 0.f);
        for (int cb = 0; cb < C / 8; ++cb)
            for (int b = 0; b < B; ++b)
                for (int r = 1; r <= H; ++r)
                    for (int c = 1; c <= W; ++c) {
                        float* p = &hx[((size_t)cb * g.cs + g.sl + (size_t)b * g.plane + (size_t)r * g.wp + c) * 8];
                        float* q = &hr[((size_t)cb * g.cs + g.sl + (size_t)b * g.plane + (size_t)r * g.wp + c) * 8];
                        for (int k = 0; k < 8; ++k) { p[k] = nd(rng); q[k] = nd(rng); }
                    }
        std::vector<float> hw((size_t)C * C * 9);
        const float sc = 1.f / (3.f * std::sqrt((float)C));
        for (auto& v : hw) v = nd(rng) * sc;
        // packs
        const int nct = C / 64, ncb = C / 8;
        std::vector<float> u4((size_t)nct * ncb * 8 * 9 * 64 * 4), u2((size_t)nct * ncb * 8 * 64 * 16);
        for (int co = 0; co < C; ++co)
            for (int ci = 0; ci < C; ++ci) {
                const float* gk = &hw[((size_t)co * C + ci) * 9];
                double t4[6][3], U4[6][6], t2[4][3], U2[4][4];
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) { t4[i][j] = 0; for (int k = 0; k < 3; ++k) t4[i][j] += G4[i][k] * gk[k * 3 + j]; }
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { U4[i][j] = 0; for (int k = 0; k < 3; ++k) U4[i][j] += t4[i][k] * G4[j][k]; }
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) { t2[i][j] = 0; for (int k = 0; k < 3; ++k) t2[i][j] += G2[i][k] * gk[k * 3 + j]; }
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { U2[i][j] = 0; for (int k = 0; k < 3; ++k) U2[i][j] += t2[i][k] * G2[j][k]; }
                const int ct = co / 64, c2 = (co % 64) / 32, r = co % 32, cb = ci / 8, hh = (ci % 8) / 4, m = ci % 4;
                for (int sl = 0; sl < 36; ++sl) {
                    const int q = sl / 9, k = sl % 9, pt = SLOT4[sl];
                    u4[(((((size_t)ct * ncb + cb) * 8 + (c2 * 4 + q)) * 9 + k) * 64 + (hh * 32 + r)) * 4 + m] = (float)U4[pt / 6][pt % 6];
                }
                for (int xi = 0; xi < 16; ++xi)
                    u2[((((size_t)ct * ncb + cb) * 8 + (ci % 8)) * 64 + (co % 64)) * 16 + xi] = (float)U2[xi / 4][xi % 4];
            }
        // the three-part bf16 split of u4 in the packing of dinv_conv3x3_winograd4_bf16x3 (deepinv_amd/hip/drunet.py:
        // pack_winograd4_bf16x3_weight): per (ct, cb, wave, point) [lane 64][um 4 | uh 4] then [lane 64][ul 4]
        auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        auto bff = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
        std::vector<uint16_t> u3(u4.size() * 3);
        for (size_t pw = 0; pw < u4.size() / 256; ++pw)
            for (int ln = 0; ln < 64; ++ln)
                for (int m = 0; m < 4; ++m) {
                    const float v = u4[pw * 256 + ln * 4 + m];
                    const uint16_t h = bf(v); const float r1 = v - bff(h);
                    const uint16_t mi = bf(r1); const uint16_t lo = bf(r1 - bff(mi));
                    u3[pw * 768 + ln * 8 + m] = mi; u3[pw * 768 + ln * 8 + 4 + m] = h; u3[pw * 768 + 512 + ln * 4 + m] = lo;
                }
        float* du3;
        CK(hipMalloc(&du3, u3.size() * 2));
        CK(hipMemcpy(du3, u3.data(), u3.size() * 2, hipMemcpyHostToDevice));
        float *dx, *dr, *dy4, *dy2, *du4, *du2;
        CK(hipMalloc(&dx, nact * 4)); CK(hipMalloc(&dr, nact * 4)); CK(hipMalloc(&dy4, nact * 4)); CK(hipMalloc(&dy2, nact * 4));
        CK(hipMalloc(&du4, u4.size() * 4)); CK(hipMalloc(&du2, u2.size() * 4));
        CK(hipMemcpy(dx, hx.data(), nact * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dr, hr.data(), nact * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(du4, u4.data(), u4.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(du2, u2.data(), u2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dy4, 0, nact * 4)); CK(hipMemset(dy2, 0, nact * 4));
        if (!quick) {
        // ---- correctness: residual form (conv + r)
        std::vector<float> y4b(nact);
        DK(dinv_conv3x3_winograd4_bf16x3(&g, dx, du3, C, C, dy4, dr, 0, dws, wsb, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y4b.data(), dy4, nact * 4, hipMemcpyDeviceToHost));
        DK(dinv_conv3x3_winograd4(&g, dx, du4, C, C, dy4, dr, 0, dws, wsb, nullptr));
        DK(dinv_conv3x3_winograd(&g, dx, du2, C, C, dy2, dr, 0, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<float> y4(nact), y2(nact);
        CK(hipMemcpy(y4.data(), dy4, nact * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y2.data(), dy2, nact * 4, hipMemcpyDeviceToHost));
        double d42 = 0, n2 = 0;
        for (size_t i = 0; i < nact; ++i) { const double d = (double)y4[i] - y2[i]; d42 += d * d; n2 += (double)y2[i] * y2[i]; }
        // frames untouched?
        double frame = 0;
        for (int cb = 0; cb < C / 8; ++cb)
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < g.wp; ++c)
                    for (int k = 0; k < 8; ++k) {
                        frame += std::fabs(y4[((size_t)cb * g.cs + g.sl + (size_t)b * g.plane + c) * 8 + k]);
                        frame += std::fabs(y4[((size_t)cb * g.cs + g.sl + (size_t)b * g.plane + (size_t)(H + 1) * g.wp + c) * 8 + k]);
                    }
        // sampled fp64 reference
        std::uniform_int_distribution<int> ub(0, B - 1), uh(0, H - 1), uc(0, C - 1);
        double e4 = 0, e2 = 0, nr = 0, e4b = 0;
        const int ns = 4000;
        for (int s = 0; s < ns; ++s) {
            int b = ub(rng), oy = uh(rng), ox = uh(rng), co = uc(rng);
            if (s < 64) { oy = (s & 1) ? H - 1 - (s % 3) : (s % 3); ox = (s & 2) ? W - 1 - (s % 5) : (s % 5); }   // corners / edges too
            double acc = 0;
            for (int ci = 0; ci < C; ++ci)
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx_ = 0; dx_ < 3; ++dx_) {
                        const size_t p = ((size_t)(ci / 8) * g.cs + g.sl + (size_t)b * g.plane + (size_t)(oy + dy) * g.wp + ox + dx_) * 8 + ci % 8;
                        acc += (double)hw[((size_t)co * C + ci) * 9 + dy * 3 + dx_] * hx[p];
                    }
            const size_t po = ((size_t)(co / 8) * g.cs + g.sl + (size_t)b * g.plane + (size_t)(oy + 1) * g.wp + ox + 1) * 8 + co % 8;
            acc += hr[po];
            e4b += (y4b[po] - acc) * (y4b[po] - acc); e4 += (y4[po] - acc) * (y4[po] - acc); e2 += (y2[po] - acc) * (y2[po] - acc); nr += acc * acc;
        }
        double d4b = 0;
        for (size_t i = 0; i < nact; ++i) { const double d = (double)y4b[i] - y4[i]; d4b += d * d; }
        printf("{\"level\": %d, \"H\": %d, \"C\": %d, \"rel_f4_vs_f2\": %.3e, \"rel_f4_vs_fp64\": %.3e, \"rel_f4bf16x3_vs_fp64\": %.3e, \"rel_f4bf16x3_vs_f4\": %.3e, \"rel_f2_vs_fp64\": %.3e, \"frame_abs_sum\": %.1f}\n",
               lv, H, C, std::sqrt(d42 / n2), std::sqrt(e4 / nr), std::sqrt(e4b / nr), std::sqrt(d4b / n2), std::sqrt(e2 / nr), frame);
        fflush(stdout);
        }
        // ---- timing
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int kern = 0; kern < (quick ? 2 : 3); ++kern)
            for (int mode = quick ? 1 : 0; mode < 3; ++mode) {
                auto run = [&]() {
                    if (kern == 0) DK(dinv_conv3x3_winograd4(&g, dx, du4, C, C, dy4, mode == 2 ? dr : nullptr, mode == 1, dws, wsb, nullptr));
                    else if (kern == 1) DK(dinv_conv3x3_winograd4_bf16x3(&g, dx, du3, C, C, dy4, mode == 2 ? dr : nullptr, mode == 1, dws, wsb, nullptr));
                    else DK(dinv_conv3x3_winograd(&g, dx, du2, C, C, dy2, mode == 2 ? dr : nullptr, mode == 1, nullptr));
                };
                for (int i = 0; i < 3; ++i) run();
                CK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < reps; ++i) run();
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                const double gf = 2.0 * 9 * C * (double)C * B * H * W / 1e9;
                printf("{\"level\": %d, \"kernel\": \"%s\", \"mode\": \"%s\", \"ms\": %.4f, \"direct_equiv_TFLOPs\": %.1f}\n", lv,
                       kern == 0 ? "winograd4" : kern == 1 ? "winograd4_bf16x3" : "winograd2", mode == 0 ? "plain" : mode == 1 ? "relu" : "res", ms, gf / ms);
                fflush(stdout);
            }
#ifdef TIMING
        {   // per-phase cycle stamps of wave 0 of every workgroup, tiles 1 and 2 (steady state), averaged over the workgroups
            const int NW = 256;
            long long* dd;
            CK(hipMalloc(&dd, (size_t)NW * 4 * 16 * 8));
            for (int kb = 0; kb < 2; ++kb)
            for (int mode = 1; mode < 3; ++mode) {
                CK(hipMemset(dd, 0, (size_t)NW * 4 * 16 * 8));
                dinv_debug_wino4_timing(dd);
                if (kb) DK(dinv_conv3x3_winograd4_bf16x3(&g, dx, du3, C, C, dy4, mode == 2 ? dr : nullptr, mode == 1, dws, wsb, nullptr));
                else DK(dinv_conv3x3_winograd4(&g, dx, du4, C, C, dy4, mode == 2 ? dr : nullptr, mode == 1, dws, wsb, nullptr));
                CK(hipDeviceSynchronize());
                dinv_debug_wino4_timing(nullptr);
                std::vector<long long> hd((size_t)NW * 4 * 16);
                CK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
                double ph[11] = {0}; int n = 0;
                for (int w = 0; w < NW; ++w)
                    for (int k = 1; k < 3; ++k) {
                        const long long* t = &hd[((size_t)w * 4 + k) * 16];
                        if (!t[0] || !t[10]) continue;
                        for (int i = 1; i <= 10; ++i) ph[i] += (double)(t[i] - t[i - 1]);
                        ph[0] += (double)(t[10] - t[0]);
                        ++n;
                    }
                printf("{\"level\": %d, \"bf16x3\": %d, \"mode\": \"%s\", \"tiles\": %d, \"ticks_per_tile\": %.0f, \"prologue_stage\": %.0f, \"prologue_transform\": %.0f, "
                       "\"main_loop\": %.0f, \"e_barrier1\": %.0f, \"e_write0\": %.0f, \"e_finish0\": %.0f, \"e_barrier3\": %.0f, \"e_write1_next\": %.0f, "
                       "\"e_finish1\": %.0f, \"e_barrier5\": %.0f}\n", lv, kb, mode == 1 ? "relu" : "res", n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n,
                       ph[5] / n, ph[6] / n, ph[7] / n, ph[8] / n, ph[9] / n, ph[10] / n);
                fflush(stdout);
            }
            CK(hipFree(dd));
        }
#endif
        CK(hipFree(dx)); CK(hipFree(dr)); CK(hipFree(dy4)); CK(hipFree(dy2)); CK(hipFree(du4)); CK(hipFree(du2)); CK(hipFree(du3));
    }
    return 0;
}
