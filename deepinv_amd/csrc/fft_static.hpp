// Compile-time planned FFT passes for the sizes the BASELINE configs use (16, 64, 128, 256, 320, 512).
//
// Decimation-in-frequency, at most three radix stages N = R1*R2*R3, laid out so that
//   * stage 1 reads its R1 inputs straight from global memory (lanes run along the contiguous index, so each
//     of the R1 loads is a fully coalesced wave transaction) -- no permutation table, no staging pass;
//   * stages exchange data through LDS with lane-linear (conflict-free) writes and reads;
//   * the last stage writes its R outputs straight to global memory, the thread -> (q1,q2) mapping chosen so
//     that for every output register consecutive lanes hit consecutive addresses;
//   * twiddles: one table read per butterfly (W^u) and its powers by complex multiplication.
// LDS traffic per element: 2 writes + 2 reads (the generic engine needs 4 + 4 plus table reads).
// Centred transforms fold both shifts into the global index maps ((idx + c) mod N), as in fft_core.hpp.
#pragma once
#include "fft_core.hpp"

namespace dinv {

__host__ __device__ inline int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ radix-16 butterfly (4x4 DIF, constants)
template <bool INV>
struct Bfly<16, INV> {
    static DINV_HD void run(float2 (&v)[16]) {
        // n = 4a + b ; k = q + 4r
        const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos/sin(pi/8)
        const float h = 0.70710678118654752440f;
        float2 y[4][4];  // y[b][q]
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float2 t[4] = {v[b], v[4 + b], v[8 + b], v[12 + b]};
            Bfly<4, INV>::run(t);
#pragma unroll
            for (int q = 0; q < 4; ++q) y[b][q] = t[q];
        }
        // twiddle W16^(b*q), forward W16 = exp(-2 pi i/16)
        const float2 w1 = make_float2(c1, INV ? s1 : -s1);
        const float2 w2 = make_float2(h, INV ? h : -h);
        const float2 w3 = make_float2(s1, INV ? c1 : -c1);
        const float2 w4 = make_float2(0.f, INV ? 1.f : -1.f);
        const float2 w6 = make_float2(-h, INV ? h : -h);
        const float2 w9 = make_float2(-c1, INV ? -s1 : s1);
        y[1][1] = cmul(y[1][1], w1); y[1][2] = cmul(y[1][2], w2); y[1][3] = cmul(y[1][3], w3);
        y[2][1] = cmul(y[2][1], w2); y[2][2] = cmul(y[2][2], w4); y[2][3] = cmul(y[2][3], w6);
        y[3][1] = cmul(y[3][1], w3); y[3][2] = cmul(y[3][2], w6); y[3][3] = cmul(y[3][3], w9);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float2 t[4] = {y[0][q], y[1][q], y[2][q], y[3][q]};
            Bfly<4, INV>::run(t);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[q + 4 * r] = t[r];
        }
    }
};

template <bool INV>
struct Bfly<1, INV> {
    static DINV_HD void run(float2 (&v)[1]) {}
};

// v[q] *= w^q (q >= 1); conj for the inverse transform.  Powers by multiplication (depth <= 4).
template <int R, bool INV>
DINV_HD void apply_twiddle_powers(float2 (&v)[R], float2 w) {
    if (INV) w.y = -w.y;
    if constexpr (R >= 2) {
        float2 p[R];
        p[1] = w;
#pragma unroll
        for (int q = 2; q < R; ++q) p[q] = (q % 2 == 0) ? cmul(p[q / 2], p[q / 2]) : cmul(p[q - 1], w);
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], p[q]);
    }
}

template <int N_, int R1_, int R2_, int R3_>
struct StaticPlan {
    static constexpr int N = N_, R1 = R1_, R2 = R2_, R3 = R3_;
    static_assert(R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static constexpr int M1 = N / R1;   // sub-transform length after stage 1
    static constexpr int M2 = M1 / R2;  // == R3
    static constexpr int STAGES = (R3 > 1) ? 3 : (R2 > 1 ? 2 : 1);
    // ROW layout [line][q1*M1P + rest]: pad the q1 stride so that stage-3 reads (q1 fastest) spread over banks
    static constexpr int M1P = (M1 % 16 == 0) ? M1 + 1 : M1;
    static constexpr int LSR = R1 * M1P;
    static constexpr int K1 = M1, K2 = N / R2, K3 = N / R3;  // work items per line per stage
};

template <int N> struct PlanFor { static constexpr bool ok = false; };
template <> struct PlanFor<16>  { static constexpr bool ok = true; using P = StaticPlan<16, 16, 1, 1>; };
template <> struct PlanFor<32>  { static constexpr bool ok = true; using P = StaticPlan<32, 4, 8, 1>; };
template <> struct PlanFor<64>  { static constexpr bool ok = true; using P = StaticPlan<64, 8, 8, 1>; };
template <> struct PlanFor<128> { static constexpr bool ok = true; using P = StaticPlan<128, 2, 8, 8>; };
template <> struct PlanFor<256> { static constexpr bool ok = true; using P = StaticPlan<256, 4, 8, 8>; };
template <> struct PlanFor<320> { static constexpr bool ok = true; using P = StaticPlan<320, 5, 8, 8>; };
template <> struct PlanFor<512> { static constexpr bool ok = true; using P = StaticPlan<512, 8, 8, 8>; };

// store-friendly plans (R1 = 8): the last stage then emits k = i + 64*q3 with i = lane, i.e. 64 consecutive
// outputs per register -> full 256-byte wave stores even for planar (4-byte) outputs
template <int N> struct PlanForS { using P = typename PlanFor<N>::P; };
template <> struct PlanForS<128> { using P = StaticPlan<128, 8, 8, 2>; };
template <> struct PlanForS<256> { using P = StaticPlan<256, 8, 8, 4>; };
template <> struct PlanForS<320> { using P = StaticPlan<320, 8, 8, 5>; };

inline bool has_static_plan(int n) { return n == 16 || n == 32 || n == 64 || n == 128 || n == 256 || n == 320 || n == 512; }

// ------------------------------------------------------------------ tile transform
// ROW = true : tile of L lines, lanes run along the line (LDS [line][LSR]); global access through
//              io.load/store(RowCtx, idx), one RowCtx per line cached in LDS.
// ROW = false: tile of L columns of one outer index p, lanes run along the columns (LDS [pos][L]);
//              each thread owns one column (ColCtx in registers).
// Output is delivered through `emit(item_slot, line, k, value)` so that the coil-combine pass can accumulate.
// NT: threads of the workgroup (512 halves the per-thread register arrays of the long column transforms)
template <class P, bool INV, bool ROW, int L, int NT = 256>
struct TileFft {
    static constexpr int N = P::N;
    static __device__ __forceinline__ int addr(int line, int q1, int rest) {
        return ROW ? line * P::LSR + q1 * P::M1P + rest : (q1 * P::M1 + rest) * L + line;
    }
    static constexpr size_t lds_floats2 = ROW ? (size_t)L * P::LSR : (size_t)L * N;

    static constexpr int NS1 = (L * P::K1 + NT - 1) / NT;                                  // item slots per thread, stage 1
    static constexpr int RL = (P::STAGES == 3) ? P::R3 : (P::STAGES == 2 ? P::R2 : P::R1);  // radix of the last stage
    static constexpr int KL = N / RL;                                                      // last-stage items per line
    static constexpr int NSL = (L * KL + NT - 1) / NT;                                        // slots per thread, last stage

    // item -> (line, index-within-line); lanes run along the index (ROW) or along the lines (COL)
    static __device__ __forceinline__ void split(int w, int K, int& line, int& i) {
        if (ROW) { line = w / K; i = w - line * K; } else { i = w / L; line = w - i * L; }
    }

    // LDS_IN: the inputs themselves live in `buf` (natural order, whatever addressing `load` uses): every thread first
    // pulls ALL its stage-1 inputs into registers, then the workgroup synchronises, and only then are the stage-1
    // outputs written over the same LDS tile (two transforms back to back on one tile, see mri_cols_normal_kernel).
    // stage-1 inputs of this thread pulled into registers by the caller (software pipelining across tiles: the loads of
    // the NEXT tile are in flight while the current one is transformed); feed them back through run_regs()
    template <class LoadF>
    static __device__ __forceinline__ void load_inputs(float2 (&vin)[NS1][P::R1], int lines, int c, int tid, LoadF load) {
#pragma unroll
        for (int slot = 0; slot < NS1; ++slot) {
            const int w = tid + NT * slot;
            int line, u;
            split(w, P::K1, line, u);
            const bool ok = w < L * P::K1 && line < lines;
#pragma unroll
            for (int j = 0; j < P::R1; ++j) {
                int n = u + P::M1 * j + c;
                if (n >= N) n -= N;
                vin[slot][j] = ok ? load(slot, j, line, n) : make_float2(0.f, 0.f);
            }
        }
    }
    template <class EmitF>
    static __device__ __forceinline__ void run_regs(float2* buf, const float2* __restrict__ tw, int lines, int c, float scale,
                                                    int tid, const float2 (&vin)[NS1][P::R1], EmitF emit) {
        run<false>(buf, tw, lines, c, scale, tid, [&](int slot, int j, int, int) { return vin[slot][j]; }, emit);
    }

    template <bool LDS_IN = false, class LoadF, class EmitF>
    static __device__ __forceinline__ void run(float2* buf, const float2* __restrict__ tw, int lines, int c,
                                               float scale, int tid, LoadF load, EmitF emit) {
        constexpr int R1 = P::R1, R2 = P::R2, M1 = P::M1, M2 = P::M2;
        float2 vin[LDS_IN ? NS1 : 1][R1];
        if constexpr (LDS_IN) {
#pragma unroll
            for (int slot = 0; slot < NS1; ++slot) {
                const int w = tid + NT * slot;
                int line, u;
                split(w, P::K1, line, u);
                if (w >= L * P::K1 || line >= lines) continue;
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    int n = u + M1 * j + c;
                    if (n >= N) n -= N;
                    vin[slot][j] = load(slot, j, line, n);
                }
            }
            __syncthreads();
        }
        // ---------------- stage 1 : global -> registers -> LDS  (or straight to the output when single-stage)
#pragma unroll
        for (int slot = 0; slot < NS1; ++slot) {
            const int w = tid + NT * slot;
            int line, u;
            split(w, P::K1, line, u);
            if (w >= L * P::K1 || line >= lines) continue;
            float2 v[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                if constexpr (LDS_IN) {
                    v[j] = vin[slot][j];
                } else {
                    int n = u + M1 * j + c;
                    if (n >= N) n -= N;
                    v[j] = load(slot, j, line, n);
                }
            }
            Bfly<R1, INV>::run(v);
            if constexpr (P::STAGES == 1) {
#pragma unroll
                for (int q = 0; q < R1; ++q) {
                    int k = q + c;
                    if (k >= N) k -= N;
                    emit(slot, line, k, q, cscale(v[q], scale));
                }
            } else {
                apply_twiddle_powers<R1, INV>(v, tw[u]);  // W_N^(u q)
#pragma unroll
                for (int q = 0; q < R1; ++q) buf[addr(line, q, u)] = v[q];
            }
        }
        if constexpr (P::STAGES == 1) return;
        __syncthreads();
        if constexpr (P::STAGES == 3) {
            // ---------------- stage 2 : LDS -> LDS, items (line, q1, u') with u' fastest
            constexpr int NS2 = (L * P::K2 + NT - 1) / NT;
#pragma unroll
            for (int slot = 0; slot < NS2; ++slot) {
                const int w = tid + NT * slot;
                int line, i;
                split(w, P::K2, line, i);
                if (w >= L * P::K2 || line >= lines) continue;
                const int q1 = i / M2, u = i % M2;
                float2 v[R2];
#pragma unroll
                for (int j = 0; j < R2; ++j) v[j] = buf[addr(line, q1, u + M2 * j)];
                Bfly<R2, INV>::run(v);
                apply_twiddle_powers<R2, INV>(v, tw[R1 * u]);  // W_M1^(u q2) = W_N^(R1 u q2)
#pragma unroll
                for (int q = 0; q < R2; ++q) buf[addr(line, q1, q * M2 + u)] = v[q];
            }
            __syncthreads();
        }
        // ---------------- last stage : LDS -> output, items (line, q2, q1) with q1 fastest
        constexpr int Q2N = (P::STAGES == 3) ? R2 : 1;  // number of q2 values
#pragma unroll
        for (int slot = 0; slot < NSL; ++slot) {
            const int w = tid + NT * slot;
            int line, i;
            split(w, KL, line, i);
            if (w >= L * KL || line >= lines) continue;
            const int q1 = i % R1, q2 = i / R1;
            float2 v[RL];
#pragma unroll
            for (int j = 0; j < RL; ++j) v[j] = buf[addr(line, q1, (Q2N > 1 ? q2 * RL : 0) + j)];
            Bfly<RL, INV>::run(v);
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                int k = q1 + R1 * q2 + R1 * Q2N * q + c;
                if (k >= N) k -= N;
                emit(slot, line, k, q, cscale(v[q], scale));
            }
        }
    }

    // ---- the pieces of run_v4 for callers that pipeline tiles themselves (fft_wave.hpp): one stage-1 item from registers,
    // and everything after stage 1 with a caller-chosen synchronisation (workgroup barrier or wave-local ordering)
    static constexpr int NSV1 = (L * (P::M1 / 4) + NT - 1) / NT;
    // (`tw` may point to an LDS copy of the table: a global-memory twiddle read is a dependent L2 round trip per stage)
    static __device__ __forceinline__ void v4_stage1_item(float2* buf, const float2* tw, int line, int u0,
                                                          const float2 (&x)[P::R1][4]) {
        constexpr int R1 = P::R1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float2 v[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) v[j] = x[j][e];
            Bfly<R1, INV>::run(v);
            apply_twiddle_powers<R1, INV>(v, tw[u0 + e]);
#pragma unroll
            for (int q = 0; q < R1; ++q) buf[addr(line, q, u0 + e)] = v[q];
        }
    }
    template <class Sync, class EmitV>
    static __device__ __forceinline__ void v4_finish(float2* buf, const float2* tw, int lines, int c, float scale,
                                                     int tid, EmitV emitv) {
        static_assert(ROW, "v4_finish is a rows-pass piece");
        constexpr int R1 = P::R1, R2 = P::R2, M2 = P::M2;
        if constexpr (P::STAGES == 3) {
            constexpr int NS2 = (L * P::K2 + NT - 1) / NT;
#pragma unroll
            for (int slot = 0; slot < NS2; ++slot) {
                const int w = tid + NT * slot;
                int line, i;
                split(w, P::K2, line, i);
                if (w >= L * P::K2 || line >= lines) continue;
                const int q1 = i / M2, u = i % M2;
                float2 v[R2];
#pragma unroll
                for (int j = 0; j < R2; ++j) v[j] = buf[addr(line, q1, u + M2 * j)];
                Bfly<R2, INV>::run(v);
                apply_twiddle_powers<R2, INV>(v, tw[R1 * u]);
#pragma unroll
                for (int q = 0; q < R2; ++q) buf[addr(line, q1, q * M2 + u)] = v[q];
            }
            Sync::sync();
        }
        constexpr int Q2N = (P::STAGES == 3) ? R2 : 1;
        constexpr int TL = KL / 4;
        constexpr int NSVL = (L * TL + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NSVL; ++slot) {
            const int w = tid + NT * slot;
            const int line = w / TL, i0 = (w - line * TL) * 4;
            if (w >= L * TL || line >= lines) continue;
            float2 o[RL][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e;
                const int q1 = i % R1, q2 = i / R1;
                float2 v[RL];
#pragma unroll
                for (int j = 0; j < RL; ++j) v[j] = buf[addr(line, q1, (Q2N > 1 ? q2 * RL : 0) + j)];
                Bfly<RL, INV>::run(v);
#pragma unroll
                for (int q = 0; q < RL; ++q) o[q][e] = cscale(v[q], scale);
            }
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                int k0 = i0 + R1 * Q2N * q + c;
                if (k0 >= N) k0 -= N;
                emitv(slot, line, k0, q, o[q]);
            }
        }
    }

    // ---- twiddles from a per-plan LDS table instead of powers by multiplication (R - 2 complex multiplies per butterfly are
    // traded for R - 1 LDS reads; the wave-autonomous kernels are bound by vector-ALU issue, not by LDS):
    //   T1[(q - 1) * M1 + u] = W_N^(u q), u < M1, 1 <= q < R1        T2[(q - 1) * M2 + u] = W_N^(R1 u q), u < M2, 1 <= q < R2
    static constexpr int TAB1 = (P::R1 - 1) * P::M1, TAB2 = (P::STAGES == 3) ? (P::R2 - 1) * P::M2 : 0;
    static constexpr int TAB = (TAB1 + TAB2 + 1) / 2 * 2;       // float2 entries (kept even: 16-byte aligned neighbours)
    static __device__ __forceinline__ void fill_twiddle_table(float2* tab, const float2* __restrict__ tw, int tid, int nthreads) {
        for (int i = tid; i < TAB1; i += nthreads) tab[i] = tw[(i % P::M1) * (i / P::M1 + 1)];
        if constexpr (P::STAGES == 3)
            for (int i = tid; i < TAB2; i += nthreads) tab[TAB1 + i] = tw[P::R1 * (i % P::M2) * (i / P::M2 + 1)];
    }
    static __device__ __forceinline__ float2 twmul(float2 v, float2 w) { return INV ? cmulc(v, w) : cmul(v, w); }
    static __device__ __forceinline__ void v4_stage1_item_tab(float2* buf, const float2* tab, int line, int u0,
                                                              const float2 (&x)[P::R1][4]) {
        constexpr int R1 = P::R1, M1 = P::M1;
        float2 w[R1 - 1 > 0 ? R1 - 1 : 1][4];
#pragma unroll
        for (int q = 1; q < R1; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[q - 1][e] = tab[(q - 1) * M1 + u0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float2 v[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) v[j] = x[j][e];
            Bfly<R1, INV>::run(v);
            buf[addr(line, 0, u0 + e)] = v[0];
#pragma unroll
            for (int q = 1; q < R1; ++q) buf[addr(line, q, u0 + e)] = twmul(v[q], w[q - 1][e]);
        }
    }
    template <class Sync>
    static __device__ __forceinline__ void v4_stage2_tab(float2* buf, const float2* tab, int lines, int tid) {
        static_assert(ROW && P::STAGES == 3, "three-stage rows plans");
        constexpr int R2 = P::R2, M2 = P::M2;
        constexpr int NS2 = (L * P::K2 + NT - 1) / NT;
        const float2* t2 = tab + TAB1;
#pragma unroll
        for (int slot = 0; slot < NS2; ++slot) {
            const int w = tid + NT * slot;
            int line, i;
            split(w, P::K2, line, i);
            if (w >= L * P::K2 || line >= lines) continue;
            const int q1 = i / M2, u = i % M2;
            float2 v[R2];
#pragma unroll
            for (int j = 0; j < R2; ++j) v[j] = buf[addr(line, q1, u + M2 * j)];
            Bfly<R2, INV>::run(v);
            buf[addr(line, q1, u)] = v[0];
#pragma unroll
            for (int q = 1; q < R2; ++q) buf[addr(line, q1, q * M2 + u)] = twmul(v[q], t2[(q - 1) * M2 + u]);
        }
        Sync::sync();
    }

    // ---- pieces for the wave-autonomous MRI passes (mri_wave.hpp): stage 2 alone, the last stage IN PLACE (the output
    // k = q1 + R1 q2 + R1 Q2N q of an item overwrites the item's input j = q), and the LDS position of output k after it
    template <class Sync>
    static __device__ __forceinline__ void v4_stage2(float2* buf, const float2* tw, int lines, int tid) {
        static_assert(ROW && P::STAGES == 3, "three-stage rows plans");
        constexpr int R1 = P::R1, R2 = P::R2, M2 = P::M2;
        constexpr int NS2 = (L * P::K2 + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NS2; ++slot) {
            const int w = tid + NT * slot;
            int line, i;
            split(w, P::K2, line, i);
            if (w >= L * P::K2 || line >= lines) continue;
            const int q1 = i / M2, u = i % M2;
            float2 v[R2];
#pragma unroll
            for (int j = 0; j < R2; ++j) v[j] = buf[addr(line, q1, u + M2 * j)];
            Bfly<R2, INV>::run(v);
            apply_twiddle_powers<R2, INV>(v, tw[R1 * u]);
#pragma unroll
            for (int q = 0; q < R2; ++q) buf[addr(line, q1, q * M2 + u)] = v[q];
        }
        Sync::sync();
    }
    static __device__ __forceinline__ void last_inplace(float2* buf, int lines, float scale, int tid) {
        static_assert(ROW && P::STAGES == 3, "three-stage rows plans");
        constexpr int R1 = P::R1;
#pragma unroll
        for (int slot = 0; slot < NSL; ++slot) {
            const int w = tid + NT * slot;
            int line, i;
            split(w, KL, line, i);
            if (w >= L * KL || line >= lines) continue;
            const int q1 = i % R1, q2 = i / R1;
            float2 v[RL];
#pragma unroll
            for (int j = 0; j < RL; ++j) v[j] = buf[addr(line, q1, q2 * RL + j)];
            Bfly<RL, INV>::run(v);
#pragma unroll
            for (int q = 0; q < RL; ++q) buf[addr(line, q1, q2 * RL + q)] = cscale(v[q], scale);
        }
    }
    // the last stage with 4 adjacent outputs per emit (as in run_v4), and the same item decomposition without the arithmetic
    template <class EmitV>
    static __device__ __forceinline__ void v4_last(float2* buf, int lines, int c, float scale, int tid, EmitV emitv) {
        static_assert(ROW && P::STAGES == 3, "three-stage rows plans");
        constexpr int R1 = P::R1, Q2N = P::R2, TL = KL / 4, NSVL = (L * TL + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NSVL; ++slot) {
            const int w = tid + NT * slot;
            const int line = w / TL, i0 = (w - line * TL) * 4;
            if (w >= L * TL || line >= lines) continue;
            float2 o[RL][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e;
                const int q1 = i % R1, q2 = i / R1;
                float2 v[RL];
#pragma unroll
                for (int j = 0; j < RL; ++j) v[j] = buf[addr(line, q1, q2 * RL + j)];
                Bfly<RL, INV>::run(v);
#pragma unroll
                for (int q = 0; q < RL; ++q) o[q][e] = cscale(v[q], scale);
            }
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                int k0 = i0 + R1 * Q2N * q + c;
                if (k0 >= N) k0 -= N;
                emitv(slot, line, k0, q, o[q]);
            }
        }
    }
    template <class F>
    static __device__ __forceinline__ void v4_last_positions(int lines, int c, int tid, F f) {
        constexpr int R1 = P::R1, Q2N = P::R2, TL = KL / 4, NSVL = (L * TL + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NSVL; ++slot) {
            const int w = tid + NT * slot;
            const int line = w / TL, i0 = (w - line * TL) * 4;
            if (w >= L * TL || line >= lines) continue;
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                int k0 = i0 + R1 * Q2N * q + c;
                if (k0 >= N) k0 -= N;
                f(slot, line, k0, q);
            }
        }
    }
    static __device__ __forceinline__ int pos_of(int line, int k) {      // where last_inplace left output k of `line`
        constexpr int R1 = P::R1, R2 = P::R2;
        const int q1 = k % R1, r = k / R1, q2 = r % R2, q = r / R2;
        return addr(line, q1, q2 * RL + q);
    }

    // ---- ROW-mode variant with 4 adjacent elements per thread on the global side: every global access is a
    // 16-byte-per-lane vector access (the guide's "vectorize ALWAYS" rule; 4-byte-per-lane streams top out near
    // 2.7 TB/s on this chip, 16-byte ones reach >5 TB/s).  loadv(line, n0, out[4]) / emitv(slot, line, k0, q, v[4]).
    template <class LoadV, class EmitV>
    static __device__ __forceinline__ void run_v4(float2* buf, const float2* __restrict__ tw, int lines, int c,
                                                  float scale, int tid, LoadV loadv, EmitV emitv) {
        static_assert(ROW, "run_v4 is a rows-pass variant");
        constexpr int R1 = P::R1, R2 = P::R2, M1 = P::M1, M2 = P::M2;
        static_assert(M1 % 4 == 0 && KL % 4 == 0 && N % 8 == 0 && P::STAGES >= 2, "vector width 4 needs 4 | M1, KL");
        constexpr int T1 = M1 / 4;                       // threads per line, stage 1
        constexpr int NSV1 = (L * T1 + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NSV1; ++slot) {
            const int w = tid + NT * slot;
            const int line = w / T1, u0 = (w - line * T1) * 4;
            if (w >= L * T1 || line >= lines) continue;
            float2 x[R1][4];
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                int n0 = u0 + M1 * j + c;
                if (n0 >= N) n0 -= N;
                loadv(line, n0, x[j]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float2 v[R1];
#pragma unroll
                for (int j = 0; j < R1; ++j) v[j] = x[j][e];
                Bfly<R1, INV>::run(v);
                apply_twiddle_powers<R1, INV>(v, tw[u0 + e]);
#pragma unroll
                for (int q = 0; q < R1; ++q) buf[addr(line, q, u0 + e)] = v[q];
            }
        }
        __syncthreads();
        if constexpr (P::STAGES == 3) {
            constexpr int NS2 = (L * P::K2 + NT - 1) / NT;
#pragma unroll
            for (int slot = 0; slot < NS2; ++slot) {
                const int w = tid + NT * slot;
                int line, i;
                split(w, P::K2, line, i);
                if (w >= L * P::K2 || line >= lines) continue;
                const int q1 = i / M2, u = i % M2;
                float2 v[R2];
#pragma unroll
                for (int j = 0; j < R2; ++j) v[j] = buf[addr(line, q1, u + M2 * j)];
                Bfly<R2, INV>::run(v);
                apply_twiddle_powers<R2, INV>(v, tw[R1 * u]);
#pragma unroll
                for (int q = 0; q < R2; ++q) buf[addr(line, q1, q * M2 + u)] = v[q];
            }
            __syncthreads();
        }
        constexpr int Q2N = (P::STAGES == 3) ? R2 : 1;
        constexpr int TL = KL / 4;                       // threads per line, last stage
        constexpr int NSVL = (L * TL + NT - 1) / NT;
#pragma unroll
        for (int slot = 0; slot < NSVL; ++slot) {
            const int w = tid + NT * slot;
            const int line = w / TL, i0 = (w - line * TL) * 4;
            if (w >= L * TL || line >= lines) continue;
            float2 o[RL][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e;
                const int q1 = i % R1, q2 = i / R1;
                float2 v[RL];
#pragma unroll
                for (int j = 0; j < RL; ++j) v[j] = buf[addr(line, q1, (Q2N > 1 ? q2 * RL : 0) + j)];
                Bfly<RL, INV>::run(v);
#pragma unroll
                for (int q = 0; q < RL; ++q) o[q][e] = cscale(v[q], scale);
            }
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                int k0 = i0 + R1 * Q2N * q + c;
                if (k0 >= N) k0 -= N;
                emitv(slot, line, k0, q, o[q]);
            }
        }
    }
};

// ------------------------------------------------------------------ kernels
template <class P, class Io, bool INV, int L>
__global__ __launch_bounds__(256) void fft_rows_static_kernel(Io io, int64_t nlines, int64_t ntiles,
                                                              const void* table, int centered, float scale) {
    using TF = TileFft<P, INV, true, L>;
    __shared__ __attribute__((aligned(16))) float2 buf[TF::lds_floats2];
    __shared__ typename Io::RowCtx ctxs[L];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int c = centered ? P::N / 2 : 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t line0 = tile * L;
        const int lines = (int)min((int64_t)L, nlines - line0);
        __syncthreads();  // previous tile fully consumed (buf and ctxs)
        if (tid < lines) ctxs[tid] = io.row_ctx(line0 + tid);
        __syncthreads();
        TF::run(buf, tw, lines, c, scale, tid,
                [&](int, int, int line, int n) { return io.load(ctxs[line], n); },
                [&](int, int line, int k, int, float2 v) { io.store(ctxs[line], k, v); });
    }
}

template <class P, class Io, bool INV, int L>
__global__ __launch_bounds__(256) void fft_rows_static_v4_kernel(Io io, int64_t nlines, int64_t ntiles,
                                                                 const void* table, int centered, float scale) {
    using TF = TileFft<P, INV, true, L>;
    __shared__ __attribute__((aligned(16))) float2 buf[TF::lds_floats2];
    __shared__ typename Io::RowCtx ctxs[L];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int c = centered ? P::N / 2 : 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t line0 = tile * L;
        const int lines = (int)min((int64_t)L, nlines - line0);
        __syncthreads();
        if (tid < lines) ctxs[tid] = io.row_ctx(line0 + tid);
        __syncthreads();
        TF::run_v4(buf, tw, lines, c, scale, tid,
                   [&](int line, int n0, float2 (&out)[4]) { io.load4(ctxs[line], n0, out); },
                   [&](int, int line, int k0, int, const float2 (&v)[4]) { io.store4(ctxs[line], k0, v); });
    }
}

template <class P, class Io, bool INV, int L>
__global__ __launch_bounds__(256) void fft_cols_static_kernel(Io io, int64_t Q, int64_t qtiles, int64_t ntiles,
                                                              const void* table, int centered, float scale,
                                                              int group) {
    using TF = TileFft<P, INV, false, L>;
    __shared__ __attribute__((aligned(16))) float2 buf[P::STAGES > 1 ? TF::lds_floats2 : 1];
    const float2* tw = reinterpret_cast<const float2*>(table);
    const int tid = threadIdx.x;
    const int c = centered ? P::N / 2 : 0;
    const int line = tid % L;
    // `group` > 1: the `group` consecutive outer indices p = g*group + m share input data (the coils of one
    // slice share x).  Blocks b and b+8 run on the same XCD (observed placement; speed only, never
    // correctness), so the members of a sharing set are laid out 8 apart and hit that XCD's L2.
    const int64_t padded = group > 1 ? ceil_div_dev(ntiles, (int64_t)8 * group) * 8 * group : ntiles;
    for (int64_t T = blockIdx.x; T < padded; T += gridDim.x) {
        int64_t tile = T;
        if (group > 1) {
            const int64_t chunk = T / (8 * group), within = T - chunk * (8 * group);
            const int64_t set = chunk * 8 + within % 8, member = within / 8;
            const int64_t g = set / qtiles, qt = set - g * qtiles;
            tile = (g * group + member) * qtiles + qt;
            if (tile >= ntiles) continue;
        }
        const int64_t p = tile / qtiles;
        const int64_t q0 = (tile - p * qtiles) * L;
        const int cols = (int)min((int64_t)L, Q - q0);
        const typename Io::ColCtx ctx = io.col_ctx(p, q0 + (line < cols ? line : 0));
        if (P::STAGES > 1) __syncthreads();
        TF::run(buf, tw, cols, c, scale, tid,
                [&](int, int, int, int n) { return io.load(ctx, n); },
                [&](int, int, int k, int, float2 v) { io.store(ctx, k, v); });
    }
}

}  // namespace dinv
