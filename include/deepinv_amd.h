/*
 * deepinv_amd — C-ABI of the MI355X (gfx950) hot-path library  (libdeepinv_amd.so)
 *
 * Drop-in boundary: every entry point below replaces the ATen call
 * sequence behind one method of the reference's Python operator API.  The reference
 * (deepinv v0.4.1) is 100 % Python/PyTorch, so "the FFI the reference would bind" is a
 * ctypes binding from its operator classes; INTEGRATION.md shows that stub.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch types.  All tensor pointers are DEVICE
 *     pointers to contiguous fp32 buffers owned by the caller (PyTorch's caching
 *     allocator).  The library never allocates or frees device memory and never
 *     synchronises: every kernel is enqueued on the caller's `stream`.
 *   - complex tensors are interleaved (re,im) fp32 pairs (torch.complex64 layout);
 *     deepinv's "[B,2,...]" real-pair tensors are called *planar*.
 *   - return value 0 = success; non-zero = error, message via dinv_last_error()
 *     (thread-local).  The Python host raises RuntimeError on non-zero.
 *   - entry points are re-entrant per (device, stream).
 */
#ifndef DEEPINV_AMD_H
#define DEEPINV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dinv_stream_t; /* hipStream_t */

#define DINV_MAX_STAGES 16

/* ------------------------------------------------------------------------- */
/* library / error                                                            */
/* ------------------------------------------------------------------------- */
const char* dinv_last_error(void);
int dinv_version(void);   /* 9 = this header (adds dinv_conv_wgrad_3x3x3); 8: (adds dinv_blurfft_apply, dinv_blurfft_workspace_bytes, dinv_spectrum_symbol); 7: (adds dinv_conv3x3_winograd4_last_split, dinv_conv3x3_winograd4_bf16x3, dinv_conv2d/3d_filter_grad, dinv_conv3d*, dinv_cdiv_real, dinv_mask_solve and the dinv_mri_desc.reserved test hook; 6: adds dinv_affine and dinv_conv_down2x2_bf16x3; the parallel-beam Radon entry points stopped reading xn; 5: natural point order in the packed weights of dinv_conv3x3_winograd4; 4: before dinv_conv3x3_winograd4; 3: round 3 before dinv_conv3x3_wsplit; 2: round 2; 1: the round-1 entry points only) */
/* number of visible HIP devices (0 when no GPU): used by the host to fail loudly */
int dinv_device_count(int* count);

/* ------------------------------------------------------------------------- */
/* FFT plans (in-LDS mixed-radix engine, csrc/fft_core.hpp)                    */
/* replaces torch.fft.{fftn,ifftn,rfft2,irfft2,fftshift,ifftshift} call sites  */
/* deepinv/utils/mixins.py:159-180, deepinv/physics/blur.py:639-657            */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t n;        /* transform length */
    int32_t nstages;  /* number of radix stages */
    int32_t generic;  /* 1 when a stage uses the generic (prime) radix path */
    int32_t reserved;
    int32_t radix[DINV_MAX_STAGES]; /* outermost first */
} dinv_fft_plan;

/* bytes of the per-length table (n twiddles float2 + n int32 permutation) */
size_t dinv_fft_table_bytes(int32_t n);
/* fill `plan` and the HOST table buffer; caller uploads the table to the device */
int dinv_fft_plan_init(int32_t n, dinv_fft_plan* plan, void* host_table);

/* Generic centred/plain complex FFT along one axis of an interleaved complex tensor
 * viewed as [outer, n, inner] (inner == 1 -> contiguous axis).  `inverse`: 0 forward
 * (exp(-i..)), 1 inverse.  `centered`: ifftshift -> fft -> fftshift semantics of
 * MRIMixin.fft (deepinv/utils/mixins.py:171-180).  `scale` multiplies the output
 * (1/sqrt(n) for norm="ortho").  In-place (in == out) is allowed. */
int dinv_fft_c2c_axis(const float* in, float* out, int64_t outer, int64_t inner,
                      const dinv_fft_plan* plan, const void* table_dev, int32_t inverse,
                      int32_t centered, float scale, dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* MRI / MultiCoilMRI  (deepinv/physics/mri.py:80-163, 254-324;                */
/*                      deepinv/utils/mixins.py:149-206)                       */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t batch;       /* B */
    int32_t coils;       /* N (1 for single-coil MRI) */
    int32_t ndim;        /* 2 or 3 transformed dims */
    int32_t dims[3];     /* ndim==2: {H,W,0}; ndim==3: {D,H,W} */
    int32_t mask_batch;  /* 0 = no mask, 1 = shared, B = per-sample; mask is [mb,2,vol] fp32 */
    int32_t maps_batch;  /* 0 = no coil maps, 1 = shared, B = per-sample; maps are [mb,N,vol] c64 */
    int32_t coil_dim;    /* 1: k-space has a coil dim  [B,2,N,vol] (MultiCoilMRI);
                            0: k-space is [B,2,vol] (single-coil MRI, coils must be 1) */
    int32_t reserved;    /* 0.  (1 = take the wave-autonomous 2-D pipelines of csrc/mri_wave.hpp whenever the sizes allow, also
                            below the batch size from which they pay off: lets tests reach them with small inputs) */
    dinv_fft_plan plan[3];     /* one per transformed dim, same order as dims */
    const void*   table[3];    /* device tables matching plan[] */
} dinv_mri_desc;

/* bytes of scratch the two calls below need (complex [B,N,vol] intermediate) */
size_t dinv_mri_workspace_bytes(const dinv_mri_desc* d);

/* y = M .* F(S_n .* x)      x:[B,2,vol] planar  ->  y:[B,2,(N,)vol] planar
 * replaces MultiCoilMRI.A (mri.py:254-272) and MRI.A = U(mask*V_adjoint(x))
 * (forward.py:1080-1096 with mri.py:99-104) */
int dinv_mri_forward(const dinv_mri_desc* d, const float* x, const float* maps,
                     const float* mask, float* y, void* workspace, size_t ws_bytes,
                     dinv_stream_t stream);

/* x = sum_n conj(S_n) .* F^H(M .* y_n)    y:[B,2,(N,)vol] -> x:[B,2,vol]
 * replaces MultiCoilMRI.A_adjoint (mri.py:284-324, rss=False) and MRI.A_adjoint */
int dinv_mri_adjoint(const dinv_mri_desc* d, const float* y, const float* maps,
                     const float* mask, float* x, void* workspace, size_t ws_bytes,
                     dinv_stream_t stream);

/* out = A^T A x = sum_n conj(S_n) F^H( M^2 F(S_n x) )   x, out: [B,2,vol] planar.
 * What L2.grad (deepinv/optim/data_fidelity.py:335-338: A_adjoint(A(x) - y)) and the CG prox
 * (deepinv/physics/forward.py:794-812, A_adjoint_A) evaluate every iteration; the k-space tensor is
 * never materialised (forward, mask^2 and inverse transform of the first axis share one LDS tile).
 * Same workspace as above.  Only for statically planned sizes: query dinv_mri_normal_supported first
 * (1 = supported); callers fall back to dinv_mri_forward + dinv_mri_adjoint otherwise. */
int dinv_mri_normal_supported(const dinv_mri_desc* d);
int dinv_mri_normal(const dinv_mri_desc* d, const float* x, const float* maps,
                    const float* mask, float* out, void* workspace, size_t ws_bytes,
                    dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* DRUNet convolutions on the fp32 matrix cores                                */
/* (deepinv/models/drunet.py:200-210 forward_unet; layers :39-101, 323-434,    */
/*  524-602).  Activations live in "padded pixel rows, channels blocked by 8": */
/*     act[c/8][sl + b*plane + r*wp + col][c%8],  (r,col) in a zero-bordered   */
/*     (H+2) x wp frame, wp = roundup(W+2, 4); one channel-block row is        */
/*     `cs` pixels x 8 floats.                                                 */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, height, width; /* unpadded size of this U-Net level */
    int32_t hp, wp;               /* padded frame */
    int64_t plane;                /* hp*wp */
    int64_t np;                   /* batch*plane : flattened padded pixels */
    int64_t sl;                   /* leading slack pixels */
    int64_t cs;                   /* pixels per channel-block row (incl. slack) */
} dinv_act_geom;

/* fills the geometry for a level; a buffer of C channels needs ceil(C/8)*cs*8 floats, zero-initialised once */
int dinv_act_geom_init(int32_t batch, int32_t height, int32_t width, dinv_act_geom* g);

/* NCHW image (+ noise-level map as extra channel: drunet.py:238-251) -> first channel block (cin + 1 <= 8).
 * sigma_mode 0: scalar `sigma_scalar`; 1: per-sample tensor [B]; 2: map [B,1,H,W] */
int dinv_act_pack(const dinv_act_geom* g, const float* x, int32_t cin, const float* sigma,
                  int32_t sigma_mode, float sigma_scalar, float* act, dinv_stream_t stream);
int dinv_act_unpack(const dinv_act_geom* g, const float* act, int32_t cout, float* y, dinv_stream_t stream);

/* y = [relu]( conv3x3(x (+x2)) ) (+res1) (+res2), stride 1, zero padding 1, no bias
 * (nn.Conv2d in drunet.py `conv(... mode="C")`, ResBlock :403-434).
 * w_packed: [cout/MT][cin/8][9 taps][MT][8], MT = cout_tile in {32, 64} (64: fewer, fatter workgroups; 32:
 * twice as many workgroups, used when the 64-wide grid would not fill the chip); cin % 8 == 0 (zero-pad),
 * cout % MT == 0 (zero-pad); only the first ceil(cout_valid/8) output channel blocks are written. */
int dinv_conv3x3(const dinv_act_geom* g, const float* x, const float* x2, const float* w_packed,
                 int32_t cin, int32_t cout, int32_t cout_valid, int32_t cout_tile, float* y, const float* res1,
                 const float* res2, int32_t relu, dinv_stream_t stream);
/* fp32 3x3x3 convolution (nn.Conv3d of DRUNet with dim = 3, drunet.py:39-263) of volumes stored as stacks of
 * depth + 2 slices (see the 3-D section below), ONE launch: y = [relu](conv3x3x3(x)) (+res1), the padding slices of y
 * written as zeros.  w_packed: [cout/MT][dz 3][cin/8][9 taps][MT][8]; cout_tile = MT in {16, 32, 64}; MT = 16 (cout
 * padded to 16) selects the thin-layer kernel on the 16x16x4 fp32 MFMA tile (also accepted by dinv_conv3x3: x2 = res2 =
 * NULL).  x must be readable one slice before and after the buffer's range.  relu: bit 0 = ReLU; bit 1 (thin kernel
 * only, with res1, without bit 0) = res1 is a GATE: y = res1 > 0 ? conv : 0 (ReLU backward in the epilogue of the data-
 * gradient convolution; torch.autograd's threshold_backward after conv3d's input gradient). */
int dinv_conv3x3x3(const dinv_act_geom* g, const float* x, const float* w_packed, int32_t cin, int32_t cout,
                   int32_t cout_valid, int32_t cout_tile, float* y, const float* res1, int32_t relu, int32_t depth,
                   dinv_stream_t stream);
/* Last DRUNet layer: y[first channel block, channels 0..cout-1] = conv3x3(x (+x2)), 1 <= cout <= 4, stride 1,
 * zero padding 1, no bias (m_tail, drunet.py:39-101, input x + x1 at :210), on the vector ALU (HBM-bound).
 * w_tail: [cin/8][9 taps][cout][8]; cin % 8 == 0. */
int dinv_conv3x3_tail(const dinv_act_geom* g, const float* x, const float* x2, const float* w_tail, int32_t cin,
                      int32_t cout, float* y, dinv_stream_t stream);
/* Same operator as dinv_conv3x3 (no x2, one optional residual) through Winograd F(2x2,3x3): 2.25x fewer MFMA
 * flops, results equal up to fp32 rounding of the transforms (~1e-6 relative).
 * w_wino: U = G g G^T per (cout, cin), packed [cout/64][cin/8][ci 8][co 64][16]; cin % 16 == 0, cin >= 32,
 * cout % 64 == 0 (other shapes: dinv_conv3x3);
 * relu and res1 are mutually exclusive (ResBlock conv1 / conv2, drunet.py:403-434). */
int dinv_conv3x3_winograd(const dinv_act_geom* g, const float* x, const float* w_wino, int32_t cin, int32_t cout,
                          float* y, const float* res1, int32_t relu, dinv_stream_t stream);
/* Same operator through Winograd F(4x4,3x3) on the fp32 matrix cores: 36 fp32 multiplies per 4x4 output tile and (cin, cout)
 * instead of 144 (4x fewer MFMA flops than the direct form, 1.78x fewer than dinv_conv3x3_winograd), fp32 accumulation; the
 * transforms add 2-3e-6 relative per layer against an fp64 convolution (csrc/drunet_wino4.hip: U straight from L2 as MFMA
 * fragments, V = B^T d B computed once per workgroup through LDS, 64 couts x 32 tiles x 36 points per workgroup).
 * w_wino4: U = G g G^T per (cout, cin) packed [cout/64][cin/8][wave 8 = 4 (cout half) + q][slot 9][lane 64][4]; the 36 point
 *   slots 9 q + k hold the Winograd points (6 row + col) 0-5, 6-8 | 12-17, 9-11 | 18-23, 24-26 | 30-35, 27-29: a full row, then
 *   half a row, per wave (deepinv_amd/hip/drunet.py: pack_winograd4_weight, WINOGRAD4_POINT_SLOTS; ABI version 6 - version 5
 *   stored the points in natural order); cin % 16 == 0, cout % 64 == 0, height % 4 == 0, width % 4 == 0;
 * relu and res1 are mutually exclusive (ResBlock conv1 / conv2, drunet.py:403-434).
 * workspace (optional, may be NULL): dinv_conv3x3_winograd4_workspace_bytes() bytes of device memory, ZERO-FILLED once by the
 *   caller and then left to the library (it keeps its ticket words zero between launches).  With a workspace the tiles of the
 *   last, incomplete round of workgroups are cut into 2 / 4 / 8 parts along the input channels so that every compute unit
 *   works on them: the parts write partial outputs, the part that finishes last adds them in part order (deterministic). */
size_t dinv_conv3x3_winograd4_workspace_bytes(void);
int dinv_conv3x3_winograd4(const dinv_act_geom* g, const float* x, const float* w_wino4, int32_t cin, int32_t cout,
                           float* y, const float* res1, int32_t relu, void* workspace, size_t workspace_bytes,
                           dinv_stream_t stream);
/* The same operator, workspace and tail split with every fp32 multiply evaluated on the BF16 matrix cores as a THREE-part operand
 * split (x = xh + xm + xl, round to nearest even each: exact to 2^-24) and SIX products (um vm + uh vh, um vh + uh vl, uh vm +
 * ul vh; the three dropped cross terms are <= 2^-23 |u||v| in the worst case, ~2^-26 rms), fp32 accumulation: per-layer error at or
 * below the fp32 form's (1.1-2.9e-6 against 1.2-3.2e-6 at the DRUNet levels) at 3/8 of its matrix-pipe time
 * (csrc/drunet_wino4.hip, BF3 = true).  Opt-in: sustained, the package power limit makes both forms equally fast.
 * w_wino4x3: the U of dinv_conv3x3_winograd4, already split, in the same [cout/64][cin/8][wave 8][slot 9] order; per slot
 *   1536 bytes = [lane 64][um 4 | uh 4] then [lane 64][ul 4] as bf16 (deepinv_amd/hip/drunet.py: pack_winograd4_bf16x3_weight). */
int dinv_conv3x3_winograd4_bf16x3(const dinv_act_geom* g, const float* x, const void* w_wino4x3, int32_t cin, int32_t cout,
                                  float* y, const float* res1, int32_t relu, void* workspace, size_t workspace_bytes,
                                  dinv_stream_t stream);
/* What the calling thread's last dinv_conv3x3_winograd4 did with its incomplete last round: the number of parts each tail tile
 * was cut into (1 = not cut) and the tail tiles per XCD (tests and diagnostics; either pointer may be NULL). */
int dinv_conv3x3_winograd4_last_split(int32_t* split_f, int32_t* n_tail_tiles);
/* Same operator on the BF16 matrix cores with a two-part exact operand split (x = xh + xl with xh = bf16(x),
 * xl = bf16(x - xh); three products ah*bl + al*bh + ah*bh, fp32 accumulate): per output
 * |y - y_exact| <= 3 * 2^-16 * (|w| conv |x|), 2-4e-6 relative per layer on random data (csrc/drunet_split2d.hip: 2-D pixel
 * tiles, whole-step LDS stages).  This is the "bf16split" setting of the ONE precision switch of the denoiser; the fp32
 * setting uses dinv_conv3x3_winograd / dinv_conv3x3.
 * w_split: [cout/64][cin/16][dy 3][plane hi/lo][dx 3][cblk 2][row 64][8] bf16, the rows of each 32-row tile permuted so
 *   that row 8g + 4h + e carries cout 16(g>>1) + 8h + 4(g&1) + e (deepinv_amd/hip/drunet.py: pack_split2d_weight);
 *   cin % 16 == 0, cout % 64 == 0.
 * flags: bit 0  x is PRE-SPLIT: each pixel's 8-channel block holds 8 bf16 high parts then 8 bf16 low parts in the 32 bytes
 *               an fp32 block occupies (what this kernel would split an fp32 block into);
 *        bit 1  write y pre-split (no residual then);  bit 2  ReLU (no residual then);
 *        bit 3  GATE: res1 is not added but gates the output, y = res1 > 0 ? conv : 0 (the data gradient through the ReLU of
 *               a ResBlock: res1 = the forward pass's ReLU output; replaces a separate dinv_relu_backward pass; fp32 in / out);
 *        bits 8-9  pixels per workgroup: 0 = chosen from the grid size, 1 = 128, 2 = 256.
 * In a ResBlock (drunet.py:403-434) conv1 runs with flags 2|4 into a scratch buffer that conv2 reads with flag 1 and res1 = x.
 * The geometry must come from dinv_act_geom_init of this library version (trailing slack for the halo rows of the last tile). */
int dinv_conv3x3_split(const dinv_act_geom* g, const void* x, const void* w_split, int32_t cin, int32_t cout, void* y,
                       const float* res1, int32_t flags, dinv_stream_t stream);
/* The same operator (fp32 in / out; replaces the two 3x3 convolutions of a ResBlock, drunet.py:403-434) as Winograd F(2,3)
 * along image rows on the bf16 matrix cores with the same two-part operand split: 12 instead of 18 three-product multiplies
 * per output pixel pair, input channel and kernel row (csrc/drunet_wsplit.hip).  Needs an even image width.
 * w_wsplit: [cout/64][cin/16][dy 3][point 4][m 2][plane hi/lo][lane 64][8] bf16 holding U0 = g0, U1 = (g0+g1+g2)/2,
 *   U2 = (g0-g1+g2)/2, U3 = g2 of kernel row dy (formed in fp64, then split), lane = 32 (ci / 8 % 2) + row, rows permuted as
 *   for dinv_conv3x3_split (deepinv_amd/hip/drunet.py: pack_wsplit_weight).  flags: bit 2 ReLU (no residual then). */
int dinv_conv3x3_wsplit(const dinv_act_geom* g, const void* x, const void* w_wsplit, int32_t cin, int32_t cout, void* y,
                        const float* res1, int32_t flags, dinv_stream_t stream);
/* 3x3x3 convolution (nn.Conv3d in DRUNet(dim=3), drunet.py:39-263) of volumes stored as stacks of depth + 2 slices
 * (g->batch = volumes x (depth + 2), one zero slice at each end of a volume) in ONE launch: the K loop of
 * dinv_conv3x3_split also runs over the three depth taps (tap dz reads the slices shifted by dz - 1), the padding slices of
 * y are written as zeros.  x needs one readable plane in front of its first and behind its last slice; same flags.
 * w_split: [cout/64][dz 3][cin/16][dy 3][plane 2][dx 3][cblk 2][row 64][8] bf16 (hip/drunet.py: pack_split3d_weight). */
int dinv_conv3x3x3_split(const dinv_act_geom* g, const void* x, const void* w_split, int32_t cin, int32_t cout, void* y,
                         const float* res1, int32_t flags, int32_t depth, dinv_stream_t stream);

/* 2x2 stride-2 convolution (downsample_strideconv, drunet.py:524-552) on the bf16 matrix cores with the same exact
 * two-part operand split; w_split: [tap = dy*2+dx][Cin/16][plane hi/lo][cblk 2][Cout][ci 8] bf16.  Same operator as
 * dinv_conv_down2x2 (fp32 pipe). */
int dinv_conv_down2x2_bf16s(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                            const void* w_split, int32_t cin, int32_t cout, float* y, dinv_stream_t stream);
/* The same stride-2 convolution with a THREE-part bf16 operand split (x = xh + xm + xl: all 24 significand bits of an fp32 operand)
 * and six products (ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm; the dropped terms are below 2^-24 |a||b|), fp32 accumulation:
 * the fp32-equivalent form used by `conv_precision = "fp32"` (the fp32-MFMA kernel dinv_conv_down2x2 is bound by that pipe).
 * w_split3: [tap 4][cin/16][plane 3][cblk 2][cout][8] bf16 (deepinv_amd/hip/drunet.py: pack_down_bf16x3_weight);
 * cin % 16 == 0, cout % 64 == 0. */
int dinv_conv_down2x2_bf16x3(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const void* w_split3,
                             int32_t cin, int32_t cout, float* y, dinv_stream_t stream);

/* 2x2 stride-2 transposed convolution (upsample_convtranspose, drunet.py:493-521) of x (+ x2, the U-Net skip add) on
 * the bf16 matrix cores; w_split: [Cin/16][tap = dy*2+dx][plane hi/lo][cblk 2][Cout][ci 8] bf16.  Same operator as
 * dinv_conv_up2x2 (fp32 pipe); writes interior output pixels only. */
int dinv_conv_up2x2_bf16s(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                          const void* w_split, int32_t cin, int32_t cout, float* y, dinv_stream_t stream);

/* ---- training through the DRUNet prior (deepinv/unfolded/unfolded.py:116-226 with deepinv/models/drunet.py:39-263).
 * Data gradients reuse the forward kernels with re-packed weights (3x3: transposed + flipped; 2x2 down <-> 2x2 up).
 * Weight gradients:  dw[m][n][t] (+)= sum_p S[m][p] * L[n][map(p) + off_t]
 *   taps == 9: 3x3 conv, S = dL/dy [m = Cout], L = x [n = Cin], one grid (gs == gl), dw laid out [Cout][Cin][3][3];
 *   taps == 4: 2x2 stride-2 conv:   S = dL/dy on the half grid gs, L = x on the full grid gl -> [Cout][Cin][2][2];
 *              2x2 transposed conv: S = x on the half grid, L = dL/dy on the full grid       -> [Cin][Cout][2][2].
 * Both tensors in the padded channel-blocked activation layout with zero frames; fp32 matrix cores; deterministic
 * (per-slice partial sums in `ws`, fixed-order reduction).  accumulate != 0 adds to dw. */
size_t dinv_conv_wgrad_workspace_bytes(const dinv_act_geom* gs, int32_t m, int32_t n, int32_t taps);
int dinv_conv_wgrad(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m, const float* l,
                    int32_t n, int32_t taps, float* dw, int32_t accumulate, void* ws, size_t ws_bytes,
                    dinv_stream_t stream);
/* grad <- grad where act > 0 else 0 (ReLU backward on whole activation buffers; n floats, n % 4 == 0) */
int dinv_relu_backward(int64_t n, const float* act, float* grad, dinv_stream_t stream);

/* ---- 3-D volumes (DRUNet with dim = 3, deepinv/models/drunet.py:39-263 with Conv3d / ConvTranspose3d) on the 2-D
 * kernels: a volume of D slices occupies D + 2 consecutive images of the padded layout (a zero slice at each end).
 *   3x3x3 convolution = ONE launch of dinv_conv3x3x3_split / dinv_conv3x3x3 (the depth taps are part of the kernel's K
 *     loop: tap dz reads x shifted by dz - 1 slices = pointer + (dz - 1) * plane * 8 floats); its weight gradient is
 *     dinv_conv_wgrad per depth tap with L shifted the same way;
 *   2x2x2 stride-2 layers pair slice z of the half grid with slice 2 z + dz of the full grid: the _3d entry points
 *     below take the depth of the half-grid volume and the depth tap dz in {0, 1}; zero slices are skipped / kept zero. */
int dinv_conv_down2x2_bf16s_3d(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                               const void* w_split, int32_t cin, int32_t cout, float* y, int32_t depth_out,
                               int32_t dz, int32_t accumulate, dinv_stream_t stream);
int dinv_conv_up2x2_bf16s_3d(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                             const void* w_split, int32_t cin, int32_t cout, float* y, int32_t depth_in, int32_t dz,
                             dinv_stream_t stream);
/* weight gradient of a 3x3x3 layer in ONE call (the three depth taps as the second grid dimension of one launch, one
 * reduction): l = the layer input shifted by -1 slice (depth tap 0), depth_stride = floats from one depth tap to the
 * next (= plane * 8); dw [m][n][3][3][3]; ws of 3 x dinv_conv_wgrad_workspace_bytes(g, m, n, 9).  Replaces three
 * dinv_conv_wgrad calls + a stack (torch.autograd's conv3d weight gradient, deepinv/models/drunet.py:323-434 with dim=3). */
int dinv_conv_wgrad_3x3x3(const dinv_act_geom* g, const float* s, int32_t m, const float* l, int32_t n,
                          int64_t depth_stride, float* dw, int32_t accumulate, void* ws, size_t ws_bytes,
                          dinv_stream_t stream);
/* weight gradient of the depth tap dz of a 2x2x2 stride-2 (transposed) convolution: S on the half grid (depth_s slices
 * per volume), L on the full grid; dw [m][n][2][2] */
int dinv_conv_wgrad_3d(const dinv_act_geom* gs, const dinv_act_geom* gl, const float* s, int32_t m, const float* l,
                       int32_t n, float* dw, int32_t accumulate, void* ws, size_t ws_bytes, int32_t depth_s,
                       int32_t dz, dinv_stream_t stream);
/* 2x2 stride-2 conv (downsample_strideconv, drunet.py:524-552); w: [4 taps][cin/8][cout][8] */
int dinv_conv_down2x2(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x,
                      const float* w, int32_t cin, int32_t cout, float* y, dinv_stream_t stream);
/* 2x2 stride-2 transposed conv of (x + x2) (upsample_convtranspose, drunet.py:493-521);
 * w: [4 taps][cin/8][cout][8] */
int dinv_conv_up2x2(const dinv_act_geom* gin, const dinv_act_geom* gout, const float* x, const float* x2,
                    const float* w, int32_t cin, int32_t cout, float* y, dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Tomography: parallel-beam Radon, exact adjoint (gather form), ramp filter   */
/* (deepinv/physics/functional/radon.py:74-173, 176-342;                       */
/*  deepinv/physics/tomography.py:229-350; adjoint = forward.py:1302-1362)     */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t n_img;      /* B*C images */
    int32_t width;      /* W (square image) */
    int32_t grid;       /* G = ceil(sqrt2*W) (circle=0) or W (circle=1): detector count */
    int32_t pad_before; /* (W+pad)//2 - W//2 (radon.py:261-268); 0 when circle */
    int32_t n_angles;   /* A */
    int32_t circle;     /* multiply by the inscribed-disc mask (radon.py:270-283) */
    float   scale;      /* 1/operator_norm applied to the output (tomography.py:253-254) */
    int32_t reserved;
} dinv_radon_desc;

size_t dinv_radon_workspace_bytes(const dinv_radon_desc* d, int32_t adjoint);
/* x:[n_img,W,W] -> sino:[n_img,G,A].  xn:[G] = linspace(-1,1,G) (affine_grid base grid, fp32);
 * cs:[A][2] = (cos,sin) of the angles in fp32 radians, both built by the host exactly as the
 * reference builds them (radon.py:70-71, 334-341).
 * The rotated lattice of that UNIFORM base grid is ctr + R (j - ctr, i - ctr) in pixel units (ctr = (G-1)/2); since ABI
 * version 6 the parallel-beam kernels (these two entry points and the _tiled ones) evaluate it in that form - one fused
 * multiply-add per coordinate and step, identical in forward and adjoint - and no longer read `xn` (the argument keeps its
 * place; dinv_radon_backproject and the fan-beam entry points still use their grids). */
int dinv_radon_forward(const dinv_radon_desc* d, const float* x, const float* xn, const float* cs,
                       float* sino, void* ws, size_t ws_bytes, dinv_stream_t stream);
/* exact transpose of dinv_radon_forward: sino:[n_img,G,A] -> x:[n_img,W,W] */
int dinv_radon_adjoint(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                       float* x, void* ws, size_t ws_bytes, dinv_stream_t stream);
/* interpolating back-projection of IRadon.forward (radon.py:396-444), the inexact adjoint used when
 * adjoint_via_backprop=False: sino:[n_img,G,A] -> out:[n_img,W,W] * scale.  ixtab:[A] are the fp32 column
 * coordinates ((X+1)/2)(A-1), X = a*2/(A-1)-1, exactly as the reference grid holds them (radon.py:474-489). */
int dinv_radon_backproject(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                           const float* ixtab, float* out, dinv_stream_t stream);
/* ramp filter along the detector axis of sino:[n_img,n_det,A] (RampFilter, radon.py:74-173) */
int dinv_radon_ramp(int32_t n_img, int32_t n_det, int32_t n_angles, const float* sino, float* out,
                    dinv_stream_t stream);

/* ---- fan-beam geometry (fan_beam_grid, functional/radon.py:16-52; Radon(fan_beam=True) :205-342; Tomography with
 * fan_beam=True uses the exact adjoint and RampFilter + adjoint for FBP, tomography.py:229-350).
 * sino:[n_img,n_det,A].  Host tables, all fp32, built as the reference builds its grid: xm:[G] = linspace(-1,1,G)
 * (march along the central ray), yd:[n_det] = linspace(-1,1,n_det), sc:[G] = 0.5*L_det*(xm+r_src)/(r_src+r_det) with
 * the radii / detector length scaled by 2/(G*pixel_spacing); cs as above.  d->grid is the padded image size G. */
size_t dinv_radon_fan_workspace_bytes(const dinv_radon_desc* d, int32_t n_det, int32_t adjoint);
int dinv_radon_fan_forward(const dinv_radon_desc* d, int32_t n_det, const float* x, const float* xm,
                           const float* sc, const float* yd, const float* cs, float* sino, void* ws,
                           size_t ws_bytes, dinv_stream_t stream);
/* exact transpose of dinv_radon_fan_forward (deterministic gather): sino:[n_img,n_det,A] -> x:[n_img,W,W] */
int dinv_radon_fan_adjoint(const dinv_radon_desc* d, int32_t n_det, const float* sino, const float* xm,
                           const float* sc, const float* yd, const float* cs, float* x, void* ws,
                           size_t ws_bytes, dinv_stream_t stream);

/* ---- LDS-tiled kernels (csrc/radon_tiled.hip): same operators, same arithmetic, operands staged in LDS.
 * The plan is host-built once per (angles, grid): angle chunks by marching class/direction and the window
 * column range of every (chunk, 64-ray block, 16-row band).  `fits` == 0 (irregular angle lists whose chunks
 * span more than the LDS window) means: call dinv_radon_forward instead. */
typedef struct {
    int32_t grid, n_angles, kw, band_h, win_w, n_jblocks, n_bands;
    int32_t n_chunks_plain, n_chunks_swap;
    int32_t fits;
    int32_t blob_words;      /* size of the device blob in 32-bit words (= dinv_radon_plan_bytes / 4) */
    int32_t widest_window;   /* diagnostic: the largest window width any (chunk, block, band) needs */
    int32_t reserved[4];
} dinv_radon_plan;
size_t dinv_radon_plan_bytes(const dinv_radon_desc* d);
/* cs_host:[A][2] fp32 (cos,sin) as uploaded to the device; host_blob: dinv_radon_plan_bytes(d) bytes, to be
 * copied to the device unchanged.  plan->kw on ENTRY: 0 = choose the angles per workgroup (the largest of 8, 4, 2, 1
 * whose widest window fits 128 columns), 1 / 2 / 4 / 8 = force that count; every other field is output.
 * win_w (the LDS pitch) is a multiple of 16 so that a window cell's bank slot is its column modulo 16. */
int dinv_radon_plan_init(const dinv_radon_desc* d, const float* cs_host, dinv_radon_plan* plan, void* host_blob);
size_t dinv_radon_tiled_workspace_bytes(const dinv_radon_desc* d, int32_t adjoint);
/* norm_dev: optional device scalar; when non-NULL the result is DIVIDED by it (tomography.py:253-254: the
 * operator norm stays on the device, no host read-back per call); d->scale still multiplies. */
int dinv_radon_forward_tiled(const dinv_radon_desc* d, const dinv_radon_plan* plan, const void* plan_dev,
                             const float* x, const float* xn, const float* cs, const float* norm_dev,
                             float* sino, void* ws, size_t ws_bytes, dinv_stream_t stream);
int dinv_radon_adjoint_tiled(const dinv_radon_desc* d, const float* sino, const float* xn, const float* cs,
                             const float* norm_dev, float* x, void* ws, size_t ws_bytes, dinv_stream_t stream);
/* FFT ramp filter exactly as the reference pads it (radon.py:79-162): P = dinv_radon_ramp_padded_size(n_det),
 * plan/table = dinv_fft_plan_init(P), filter = dinv_radon_ramp_filter_init(P, host table) uploaded. */
int32_t dinv_radon_ramp_padded_size(int32_t n_det);
int dinv_radon_ramp_filter_init(int32_t P, const void* fft_host_table, float* host_filter_out);
int dinv_radon_ramp_fft(int32_t n_img, int32_t n_det, int32_t n_angles, int32_t P, const dinv_fft_plan* plan,
                        const void* fft_table_dev, const float* filter_dev, const float* sino, float* out,
                        dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Measurement synthesis on the device: additive Gaussian noise                */
/* (deepinv/physics/noise.py:197-330) and Cartesian MRI acceleration masks     */
/* (deepinv/physics/generator/mri.py:15-384).  Philox4x32-10: element i of a   */
/* call uses counter offset + i/4 under key `seed`.                            */
/* ------------------------------------------------------------------------- */
/* y[i] = x[i] + sigma * N(0,1); sigma = sigma_dev[i / per_sample] when sigma_dev != NULL, else sigma_scalar;
 * x may be NULL (pure noise). */
int dinv_gaussian_noise(int64_t n, int64_t per_sample, const float* x, const float* sigma_dev, float sigma_scalar,
                        uint64_t seed, uint64_t offset, float* y, dinv_stream_t stream);
/* mask[batch, channels, times, height, width] of k-space columns.  mode 0: per (batch, time) row, n_lines columns
 * without replacement with probabilities pdf_dev[width] (zero on the centre band [center_lo, center_hi), which is
 * always sampled); mode 1: equispaced columns round(arange((t + offset_b) % accel, width - 1, accel)) with
 * offset_b uniform in [0, n_offsets); mode 2 (ABI 8): every column of a row an independent Bernoulli draw with probability
 * pdf_dev[w] (PolyOrderMaskGenerator, mri.py:199-281; the centre band is whatever pdf_dev says, [center_lo, center_hi) is also
 * forced on). */
int dinv_mri_mask_lines(int32_t batch, int32_t channels, int32_t times, int32_t height, int32_t width, int32_t n_lines,
                        int32_t center_lo, int32_t center_hi, int32_t mode, const float* pdf_dev, double accel,
                        int32_t n_offsets, uint64_t seed, uint64_t offset, float* mask, dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Blur / Downsampling: padded true convolution, its exact transpose, and the  */
/* real<->half-complex 2-D FFT used by BlurFFT                                 */
/* (deepinv/physics/functional/convolution.py:42-164, 689-758, 837-865;        */
/*  deepinv/physics/blur.py:255-329, 639-657)                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, channels, height, width;  /* full-resolution image x: [B,C,H,W] */
    int32_t fbatch, fchannels, fh, fw;       /* filter [fb in {1,B}, fc in {1,C}, fh, fw] */
    int32_t mode;    /* 0 valid, 1 circular, 2 reflect, 3 replicate, 4 constant(zeros) */
    int32_t stride;  /* 1 for Blur; Downsampling factor otherwise (output = conv[::s, ::s]) */
} dinv_conv_desc;

int dinv_conv2d_out_size(const dinv_conv_desc* d, int32_t* ho, int32_t* wo);
/* y = (k (*) pad(x))[::s, ::s]   (conv2d, convolution.py:42-107; Downsampling.A, blur.py:255-283) */
int dinv_conv2d(const dinv_conv_desc* d, const float* x, const float* filter, float* y, dinv_stream_t stream);
/* exact transpose of dinv_conv2d: y:[B,C,Ho,Wo] -> x:[B,C,H,W]
 * (conv_transpose2d + _apply_transpose_padding, convolution.py:110-164, 689-758) */
int dinv_conv2d_transpose(const dinv_conv_desc* d, const float* y, const float* filter, float* x,
                          dinv_stream_t stream);
/* gradient of dinv_conv2d w.r.t. the filter, per (batch, channel) plane: dk_planes [B,C,fh,fw] from x [B,C,H,W] and the output
 * gradient gy [B,C,Ho,Wo]; the caller sums the planes a broadcast filter is shared by (what autograd through F.conv2d gives
 * the reference: convolution.py:42-107, consumed by least_squares_implicit_backward, least_squares.py:315-339).  By adjointness
 * the same call with (x := gradient w.r.t. the transpose's output, gy := the transpose's input) is the filter gradient of
 * dinv_conv2d_transpose. */
int dinv_conv2d_filter_grad(const dinv_conv_desc* d, const float* x, const float* gy, float* dk_planes, dinv_stream_t stream);

/* The same three operators on volumes [B,C,D,H,W] with filters [fb,fc,fd,fh,fw] (conv3d / conv_transpose3d, convolution.py:333-452;
 * Blur on 5-D tensors, blur.py:535-561): the 2-D padding rule on every axis, stride 1. */
typedef struct {
    int32_t batch, channels, depth, height, width;
    int32_t fbatch, fchannels, fd, fh, fw;
    int32_t mode;    /* as dinv_conv_desc */
    int32_t reserved;
} dinv_conv3d_desc;
int dinv_conv3d_out_size(const dinv_conv3d_desc* d, int32_t* dout, int32_t* ho, int32_t* wo);
int dinv_conv3d(const dinv_conv3d_desc* d, const float* x, const float* filter, float* y, dinv_stream_t stream);
int dinv_conv3d_transpose(const dinv_conv3d_desc* d, const float* y, const float* filter, float* x, dinv_stream_t stream);
int dinv_conv3d_filter_grad(const dinv_conv3d_desc* d, const float* x, const float* gy, float* dk_planes, dinv_stream_t stream);

/* rfft2 / irfft2 over the last two dims of a real [P,H,W] tensor (half spectrum [P,H,W/2+1] complex);
 * unnormalised transforms times `scale`. */
int dinv_rfft2(const float* x, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
               const dinv_fft_plan* plan_w, const void* table_w, float scale, dinv_stream_t stream);
int dinv_irfft2(const float* in, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
                const dinv_fft_plan* plan_w, const void* table_w, float scale, void* ws, size_t ws_bytes,
                dinv_stream_t stream);

/* BlurFFT as ONE call: out = irfft2( SYMBOL( rfft2(x) ) ) over the last two dims of a real [P,H,W] tensor - every operator of
 * deepinv's BlurFFT / DecomposablePhysics (deepinv/physics/blur.py:639-657 V_adjoint / U / U_adjoint / V;
 * deepinv/physics/forward.py:1080-1117 A / A_adjoint / A_adjoint_A / A_A_adjoint; :1212-1252 prox_l2 / A_dagger) with no
 * arithmetic between the transforms left to the caller.  SYMBOL(v) = post(scale(pre(v))) per frequency bin:
 *   flags bit 0 (DINV_SYM_PRE_CONJ_ANGLE)  v <- v * conj(angle)                 (U_adjoint)
 *   flags bits 4-6 scale mode: 0 none; 1  v <- m (.) v  (per real / imaginary component, `mask * view_as_real(v)`);
 *                              2  v <- (m m) (.) v;  3  v <- v / (m m + add)  (prox_l2: add = 1 / gamma);
 *                              4  v <- v * (m > 1e-5 ? 1 / m : 0)  (A_dagger)
 *   flags bit 1 (DINV_SYM_POST_ANGLE)      v <- v * angle                       (U)
 * mask: the reference's `mask` buffer [Ps,H,W/2+1,2] (float pairs), angle: its `angle` buffer [Ps,H,W/2+1] complex64 (either may
 * be null when the flags do not use it); spectrum plane p uses symbol plane p % Ps.  `scale` multiplies the result (ortho pair:
 * 1 / (H W)).  For H in {64,128,256,320,512} the forward column transform, the symbol and the inverse column transform are ONE
 * pass over an LDS tile (the half spectrum crosses HBM once each way); other heights run them as three passes.  `ws`: caller
 * scratch of dinv_blurfft_workspace_bytes(P,H,W) bytes. */
#define DINV_SYM_PRE_CONJ_ANGLE 1
#define DINV_SYM_POST_ANGLE 2
#define DINV_SYM_SCALE(mode) ((mode) << 4)
size_t dinv_blurfft_workspace_bytes(int64_t P, int32_t H, int32_t W);
int dinv_blurfft_apply(const float* x, float* out, int64_t P, const dinv_fft_plan* plan_h, const void* table_h,
                       const dinv_fft_plan* plan_w, const void* table_w, const float* mask, const float* angle,
                       int64_t symbol_planes, int32_t flags, float add, float scale, void* ws, size_t ws_bytes,
                       dinv_stream_t stream);
/* the symbol alone on a contiguous half spectrum [P,H,Wh] (interleaved complex; in place when spec_out == spec_in): the
 * multiplications of BlurFFT.U / U_adjoint and of the three-call form rfft2 -> symbol -> irfft2 */
int dinv_spectrum_symbol(const float* spec_in, float* spec_out, int64_t P, int32_t H, int32_t Wh, const float* mask,
                         const float* angle, int64_t symbol_planes, int32_t flags, float add, dinv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Loop algebra of the iteration drivers                                       */
/* (fStepPGD + L2.grad: optim_iterators/pgd.py:137-139, data_fidelity.py:335-338; */
/*  conjugate_gradient: optim/linear/conjugate_gradient.py:48-75, utils.py:6-26) */
/* ------------------------------------------------------------------------- */
/* out = a*x + b*y + c*z  (y, z may be null); n floats, 16-byte aligned */
int dinv_lincomb(int64_t n, float a, const float* x, float b, const float* y, float c, const float* z,
                 float* out, dinv_stream_t stream);
/* out = clamp(a*x + b*y + c*z + d, lo, hi)  (y, z may be null; lo = -INFINITY, hi = +INFINITY: no clamp): the affine updates
 * of DiffPIR in one pass each (reference deepinv/sampling/diffusion.py:463-507: x / (2 sqrt(a_t)) + 0.5; clamp(2 D - 1, -1, 1) / 2
 * + 0.5 = clamp(D, 0, 1); x_{t-1} = s_a x0 + s_1ma sqrt(1 - zeta) eps + s_1ma sqrt(zeta) n with eps = (x - s_a' x0) / s_1ma'). */
int dinv_affine(int64_t n, float a, const float* x, float b, const float* y, float c, const float* z, float d,
                float lo, float hi, float* out, dinv_stream_t stream);
/* out[b] = <x[b,:], y[b,:]> for b < batch (n floats per sample, n % 4 == 0); deterministic two-stage reduction.
 * `partial` is caller scratch of batch * dinv_batched_dot_blocks(n) floats. */
int32_t dinv_batched_dot_blocks(int64_t n);
int dinv_batched_dot(int32_t batch, int64_t n, const float* x, const float* y, float* out, float* partial,
                     dinv_stream_t stream);
/* CG vector updates with per-sample scalars s_b = num[b] / (den[b] + eps) kept on the device:
 *   mode 0: v0 (x) += s_b * w0 (p) ; v1 (r) -= s_b * w1 (Ap)        mode 1: v0 (p) = w0 (r) + s_b * v0 (p) */
int dinv_cg_update(int32_t mode, int32_t batch, int64_t n, const float* num, const float* den, float eps,
                   float* v0, float* v1, const float* w0, const float* w1, dinv_stream_t stream);
/* Device-side convergence (conjugate_gradient.py:61 `if torch.all(res < tol): break` without a host round trip per
 * iteration): dinv_cg_check sets *done when every sample's residual is below its tolerance; the masked update is a
 * no-op once *done is set, so iterations issued after convergence leave the iterate exactly as the break would. */
int dinv_cg_update_masked(int32_t mode, int32_t batch, int64_t n, const float* num, const float* den, float eps,
                          float* v0, float* v1, const float* w0, const float* w1, const int32_t* done,
                          dinv_stream_t stream);
int dinv_cg_check(int32_t batch, const float* res, const float* tol2, int32_t* done, dinv_stream_t stream);
/* out[i] = s[i] / (d[i mod period] + add) for n interleaved complex values s over a real symbol d of `period` entries shared by the
 * leading (batch, channel) dimensions: the pointwise division of the closed-form proxes - Downsampling.prox_l2
 * (deepinv/physics/blur.py:331-363: mean_blocks(|K|^2) + 1/gamma) and DecomposablePhysics.prox_l2 with a real singular-value
 * mask (deepinv/physics/forward.py:1212-1234: |s|^2 + 1/gamma).  s and out may alias. */
int dinv_cdiv_real(int64_t n, int64_t period, const float* s, const float* d, float add, float* out, dinv_stream_t stream);
/* The pointwise solves of DecomposablePhysics with a REAL singular-value mask m of `period` entries shared by the leading
 * dimensions of x (deepinv/physics/forward.py:1212-1252):
 *   mode 0 (prox_l2):  out = x / (m m + add)       (add = 1 / gamma)
 *   mode 1 (A_dagger): out = x * (m > 1e-5 ? 1 / m : 0)
 * x and out may alias. */
int dinv_mask_solve(int32_t mode, int64_t n, int64_t period, const float* x, const float* m, float add, float* out,
                    dinv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPINV_AMD_H */
