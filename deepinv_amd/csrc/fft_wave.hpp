// Wave-autonomous rows passes of the static-plan FFT engine (round 5).
//
// The workgroup-cooperative passes of fft_static.hpp are paced by tile latency, not by HBM bandwidth: a 256-thread
// workgroup walks load -> stage 1 -> barrier -> stage 2 -> barrier -> stage 3 -> store per 41-KB tile, two such
// workgroups fit a CU (166-206 registers), and nothing of tile i + 1 is in flight while tile i is transformed
// (measured: ~10 us per tile, 1.9-3.4 TB/s per pass at BASELINE config 2).
//
// Here ONE WAVE owns a tile of LW rows (4 rows of 320 complex = 10 KB of LDS): the three stages exchange data through
// the wave's private LDS region, and because the DS operations of one wave execute in program order there is NO
// s_barrier anywhere in the kernel - every wave runs its own load / transform / store pipeline, a dozen of them per CU
// in different phases.  The loads of the wave's NEXT tile are issued right after stage 1 has consumed the registers of
// the current one, so they are in flight during stages 2-3 and the stores (one register set, no double buffer).
// The arithmetic (butterflies, twiddles, item decomposition) is TileFft's, instantiated for 64 threads and LW lines.
#pragma once
#include "fft_static.hpp"

namespace dinv {

// Order the LDS traffic of ONE wave: the hardware executes a wave's DS instructions in program order, so all that is
// needed is that the compiler does not move LDS accesses across this point (wavefront-scope fence + scheduling
// barrier: no instruction is emitted).  Host emulation: a cooperative barrier over the fibers of the wave.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef DINV_EMU
    ::emu::wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

struct WaveSync { static __device__ __forceinline__ void sync() { wave_lds_sync(); } };

// rows per wave tile: as many as fill the 64 lanes of stage 1 (M1 / 4 lanes per row), at least 4
template <class P> struct WaveRowsL { static constexpr int value = (P::M1 / 4) <= 8 ? 8 : 4; };

// Io requirements (besides what the cooperative kernels use):
//   Raw4, Mask4                   the registers one 4-element load (of data / of mask values) leaves in flight
//   TileCtx tile_ctx(line0)       base pointers of a tile of consecutive rows (line0 wave-uniform: scalar registers)
//   load4_raw(tc, off, raw)       issue the loads of 4 consecutive elements at element offset `off` (32 bits) of the tile
//   has_mask() / load_mask4(..)   mask values of the same 4 elements (fetched at consumption: small, L2-resident)
//   unpack4(raw, m, v)            finish them (mask multiply, planar -> interleaved)
//   store4(tc, off, v)            store 4 consecutive outputs
template <class P, class Io, bool INV, int LW, int WPB, int MINW, bool PF>
__global__ __launch_bounds__(64 * WPB, MINW) void fft_rows_wave_kernel(Io io, int64_t nlines, int64_t ntiles, const void* table,
                                                                      int centered, float scale) {
    using TF = TileFft<P, INV, true, LW, 64>;
    constexpr int R1 = P::R1, M1 = P::M1, N = P::N;
    constexpr int T1 = M1 / 4;
    static_assert(TF::NSV1 == 1 && LW * T1 <= 64, "one stage-1 item per lane");
    __shared__ __attribute__((aligned(16))) float2 buf_all[(size_t)WPB * TF::lds_floats2];
    // the twiddle table in LDS (2.5 KB at N = 320, shared by the waves of the workgroup: the only workgroup-wide step of the
    // kernel): a table read is then an LDS read inside the stage instead of a dependent L2 round trip
    __shared__ __attribute__((aligned(16))) float2 tw[TF::TAB];        // W^(u q) of the two twiddled stages (fft_static.hpp)
    TF::fill_twiddle_table(tw, reinterpret_cast<const float2*>(table), threadIdx.x, 64 * WPB);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float2* buf = buf_all + (size_t)wv * TF::lds_floats2;
    const int c = centered ? N / 2 : 0;
    const int64_t stride = (int64_t)gridDim.x * WPB;
    int64_t tile = (int64_t)blockIdx.x * WPB + wv;
    // stage-1 item of this lane: row `l1` of the tile, elements u0 .. u0 + 3 (+ M1 j)
    const int l1 = lane / T1, u0 = (lane - l1 * T1) * 4;
    const bool act1 = lane < LW * T1;
    unsigned off1[R1];      // element offsets of this lane's stage-1 loads inside a tile
#pragma unroll
    for (int j = 0; j < R1; ++j) {
        int n0 = u0 + M1 * j + c;
        if (n0 >= N) n0 -= N;
        off1[j] = (unsigned)(l1 * N + n0);
    }
    typename Io::Raw4 raw[R1];

    auto issue = [&](int64_t t) __attribute__((always_inline)) {
        const int64_t line0 = t * LW;
        const int lines = (int)min((int64_t)LW, nlines - line0);
        if (act1 && l1 < lines) {
            const typename Io::TileCtx tc = io.tile_ctx(line0);
#pragma unroll
            for (int j = 0; j < R1; ++j) io.load4_raw(tc, off1[j], raw[j]);
        }
    };
    if (PF && tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += stride) {
        const int64_t line0 = tile * LW;
        const int lines = (int)min((int64_t)LW, nlines - line0);
        const typename Io::TileCtx tc = io.tile_ctx(line0);
        if (!PF) issue(tile);
        wave_lds_sync();     // the previous tile's last stage has read `buf`
        // ---------------- stage 1: registers -> LDS
        if (act1 && l1 < lines) {
            typename Io::Mask4 m[R1];
            if (io.has_mask()) {
#pragma unroll
                for (int j = 0; j < R1; ++j) io.load_mask4(tc, off1[j], m[j]);
            }
            float2 x[R1][4];
#pragma unroll
            for (int j = 0; j < R1; ++j) io.unpack4(raw[j], m[j], x[j]);
            TF::v4_stage1_item_tab(buf, tw, l1, u0, x);
        }
        if (PF && tile + stride < ntiles) issue(tile + stride);      // next tile's loads fly during stages 2, 3 and the stores
        wave_lds_sync();
        TF::template v4_stage2_tab<WaveSync>(buf, tw, lines, lane);
        TF::v4_last(buf, lines, c, scale, lane, [&](int, int line, int k0, int, const float2 (&v)[4]) {
            io.store4(tc, (unsigned)(line * N + k0), v);
        });
    }
}

}  // namespace dinv
