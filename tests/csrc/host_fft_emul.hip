// TEST INFRASTRUCTURE (not part of the product library).
// Runs the *same* butterfly / stage / permutation code as the device FFT engine
// (deepinv_amd/csrc/fft_core.hpp) on the host with a single emulated thread, so the
// GPU-less build container can check the engine's arithmetic against numpy.fft.
#include <vector>

#include "../../deepinv_amd/csrc/fft_core.hpp"

using namespace dinv;

extern "C" int dinv_fft_plan_init(int32_t n, dinv_fft_plan* plan, void* host_table);

// in/out: interleaved complex [lines, n]
extern "C" int emul_fft_lines(const float* in, float* out, int lines, int n, int inverse, int centered,
                              float scale) {
    dinv_fft_plan plan;
    std::vector<unsigned char> table(fft_table_bytes(n));
    if (int e = dinv_fft_plan_init(n, &plan, table.data())) return e;
    const float2* tw = reinterpret_cast<const float2*>(table.data());
    const int* perm = reinterpret_cast<const int*>(tw + n);
    const int LS = fft_line_stride(n);
    std::vector<float2> buf((size_t)lines * LS), alt((size_t)lines * LS);
    const int c = centered ? n / 2 : 0;
    const float2* src = reinterpret_cast<const float2*>(in);
    for (int l = 0; l < lines; ++l)
        for (int i = 0; i < n; ++i) {
            int ip = i - c;
            if (ip < 0) ip += n;
            buf[(size_t)l * LS + perm[ip]] = src[(size_t)l * n + i];
        }
    const float2* res = inverse ? tile_fft<true>(plan, buf.data(), alt.data(), tw, lines, LS, 0, 1)
                                : tile_fft<false>(plan, buf.data(), alt.data(), tw, lines, LS, 0, 1);
    float2* dst = reinterpret_cast<float2*>(out);
    for (int l = 0; l < lines; ++l)
        for (int k = 0; k < n; ++k) {
            int kp = k - c;
            if (kp < 0) kp += n;
            dst[(size_t)l * n + k] = cscale(res[(size_t)l * LS + kp], scale);
        }
    return 0;
}
