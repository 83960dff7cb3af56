"""ctypes wrappers for the DRUNet MFMA convolution kernels (csrc/drunet.hip)."""
from __future__ import annotations

import ctypes

import torch

from . import check, lib, ptr, stream_ptr


class ActGeom(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
        ("hp", ctypes.c_int32), ("wp", ctypes.c_int32),
        ("plane", ctypes.c_int64), ("np", ctypes.c_int64), ("sl", ctypes.c_int64), ("cs", ctypes.c_int64),
    ]


_declared = False
# tuning constants of the fp32 setting (module attributes, no environment knobs): output tile of the Winograd kernel the
# ResBlock convolutions take (4: csrc/drunet_wino4.hip, 2: csrc/drunet_wino.hip) and the fewest workgroup tiles (64 couts x
# 32 tile positions) a launch must have for the F(4x4,3x3) kernel (one persistent workgroup per CU: 256 on MI355X)
FP32_WINOGRAD_TILE = 4
WINOGRAD4_MIN_TILES = 32
# True: the F(4x4,3x3) launches of the fp32 setting evaluate their multiplies as a three-part bf16 split, six products on the bf16
# matrix cores (dinv_conv3x3_winograd4_bf16x3: same per-layer accuracy, 3/8 of the matrix-pipe time).  Off by default: sustained,
# the package power limit makes the two forms equally fast at every batch (scripts/r05/bf16x3_e2e.py)
FP32_WINOGRAD4_BF16X3 = False
# False: the F(4x4,3x3) launches get no workspace, i.e. the incomplete last round of tiles is not cut along the input channels
# (diagnostic switch: bench.py --no-tail-split)
WINOGRAD4_TAIL_SPLIT = True


def _l():
    global _declared
    l = lib()
    if not _declared:
        vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
        G = ctypes.POINTER(ActGeom)
        l.dinv_act_geom_init.argtypes = [i32, i32, i32, G]
        l.dinv_act_pack.argtypes = [G, vp, i32, vp, i32, f32, vp, vp]
        l.dinv_act_unpack.argtypes = [G, vp, i32, vp, vp]
        l.dinv_conv3x3.argtypes = [G, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp]
        l.dinv_conv3x3_tail.argtypes = [G, vp, vp, vp, i32, i32, vp, vp]
        l.dinv_conv3x3_winograd.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, vp]
        l.dinv_conv3x3_winograd4.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, vp, ctypes.c_size_t, vp]
        l.dinv_conv3x3_winograd4_bf16x3.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, vp, ctypes.c_size_t, vp]
        l.dinv_conv3x3_winograd4_workspace_bytes.restype = ctypes.c_size_t
        l.dinv_conv3x3_winograd4_workspace_bytes.argtypes = []
        l.dinv_conv3x3_winograd4_last_split.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(i32)]
        l.dinv_conv3x3_split.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, vp]
        l.dinv_conv3x3_wsplit.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, vp]
        l.dinv_conv3x3x3_split.argtypes = [G, vp, vp, i32, i32, vp, vp, i32, i32, vp]
        l.dinv_conv3x3x3.argtypes = [G, vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, vp]
        l.dinv_conv_down2x2.argtypes = [G, G, vp, vp, i32, i32, vp, vp]
        l.dinv_conv_down2x2_bf16s.argtypes = [G, G, vp, vp, i32, i32, vp, vp]
        l.dinv_conv_down2x2_bf16x3.argtypes = [G, G, vp, vp, i32, i32, vp, vp]
        l.dinv_conv_up2x2_bf16s.argtypes = [G, G, vp, vp, vp, i32, i32, vp, vp]
        l.dinv_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
        l.dinv_conv_wgrad_workspace_bytes.argtypes = [G, i32, i32, i32]
        l.dinv_conv_wgrad.argtypes = [G, G, vp, i32, vp, i32, i32, vp, i32, vp, ctypes.c_size_t, vp]
        l.dinv_relu_backward.argtypes = [ctypes.c_int64, vp, vp, vp]
        l.dinv_conv_down2x2_bf16s_3d.argtypes = [G, G, vp, vp, i32, i32, vp, i32, i32, i32, vp]
        l.dinv_conv_up2x2_bf16s_3d.argtypes = [G, G, vp, vp, vp, i32, i32, vp, i32, i32, vp]
        l.dinv_conv_wgrad_3d.argtypes = [G, G, vp, i32, vp, i32, vp, i32, vp, ctypes.c_size_t, i32, i32, vp]
        l.dinv_conv_wgrad_3x3x3.argtypes = [G, vp, i32, vp, i32, ctypes.c_int64, vp, i32, vp, ctypes.c_size_t, vp]
        l.dinv_conv_up2x2.argtypes = [G, G, vp, vp, vp, i32, i32, vp, vp]
        _declared = True
    return l


def geom(batch: int, h: int, w: int) -> ActGeom:
    g = ActGeom()
    check(_l().dinv_act_geom_init(batch, h, w, ctypes.byref(g)))
    return g


def alloc(g: ActGeom, channels: int, device) -> torch.Tensor:
    """zero-initialised activation buffer [ceil(channels/8), cs, 8] (padded pixel rows, channels blocked by 8)"""
    return torch.zeros(((channels + 7) // 8, g.cs, 8), device=device, dtype=torch.float32)


def pack_conv3x3_weight(w: torch.Tensor, mt: int | None = None) -> tuple[torch.Tensor, int, int]:
    """OIHW [Cout,Cin,3,3] -> [Cout/MT][Cin/8][9 taps][MT][8] (zero padded). Returns (packed, cin_p, cout_p)."""
    cout, cin = w.shape[:2]
    cin_p = (cin + 7) // 8 * 8
    cout_p = (cout + 31) // 32 * 32
    if mt is None:
        mt = 64 if cout_p % 64 == 0 else 32
    wp = torch.zeros((cout_p, cin_p, 3, 3), device=w.device, dtype=torch.float32)
    wp[:cout, :cin] = w.detach().float()
    wp = wp.reshape(cout_p // mt, mt, cin_p // 8, 8, 9).permute(0, 2, 4, 1, 3).contiguous()
    return wp, cin_p, cout_p


def pack_tail_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout<=4, Cin, 3, 3] -> [Cin/8][9 taps][Cout][8] for conv3x3_tail"""
    cout, cin = w.shape[:2]
    if cout > 4 or cin % 8:
        raise ValueError(f"tail packing needs cout <= 4 and cin % 8 == 0, got {cout},{cin}")
    return w.detach().float().reshape(cout, cin // 8, 8, 9).permute(1, 3, 0, 2).contiguous()


def _pack_split_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> two-part bf16 split (hi = bf16(w), lo = bf16(w - hi)),
    [Cout/64][Cin/16][dy 3][plane 2][dx 3][cblk 2][co 64][ci 8] (bf16)"""
    cout, cin = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"bf16-split packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    w = w.detach().float()
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    planes = torch.stack((hi, lo))                                            # [2, Cout, Cin, 3, 3]
    planes = planes.reshape(2, cout // 64, 64, cin // 16, 2, 8, 3, 3)         # pl, ct, co, s, cblk, ci, dy, dx
    return planes.permute(1, 3, 6, 0, 7, 4, 2, 5).contiguous()               # ct, s, dy, pl, dx, cblk, co, ci


def split2d_row_perm() -> torch.Tensor:
    """MFMA A-row -> cout permutation inside a 32-row tile for csrc/drunet_split2d.hip: row 8g + 4h + e carries cout
    16 (g >> 1) + 8 h + 4 (g & 1) + e, so that the 16 accumulator registers of lane half h are the complete 8-channel blocks
    2k + h (k = 0, 1) of its pixel."""
    i = torch.arange(32)
    g, h, e = i >> 3, (i >> 2) & 1, i & 3
    return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + e


def pack_split2d_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> two-part bf16 split (hi = bf16(w), lo = bf16(w - hi)), packed for csrc/drunet_split2d.hip:
    [Cout/64][Cin/16][dy 3][plane 2][dx 3][cblk 2][row 64][ci 8] (bf16), rows of each 32-row tile permuted by
    `split2d_row_perm`"""
    packed = _pack_split_weight(w)                     # ct, s, dy, pl, dx, cblk, co 64, ci 8
    perm = split2d_row_perm().to(packed.device)
    idx = torch.cat((perm, 32 + perm))
    return packed.index_select(6, idx).contiguous()


def pack_wsplit_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> Winograd F(2,3) weights along the kernel columns, U0 = g0, U1 = (g0+g1+g2)/2,
    U2 = (g0-g1+g2)/2, U3 = g2 per kernel row (fp64, rounded once to fp32), two-part bf16 split, packed for
    csrc/drunet_wsplit.hip as MFMA A fragments: [Cout/64][Cin/16][dy 3][point 4][m 2][plane 2][cblk 2][row 32][ci 8] (bf16),
    rows of each 32-row tile permuted by `split2d_row_perm`"""
    cout, cin = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"Winograd bf16-split packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    g = w.detach().double()                                                     # [co, ci, dy, dx]
    u = torch.stack((g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) / 2, (g[..., 0] - g[..., 1] + g[..., 2]) / 2, g[..., 2])).float()
    hi = u.bfloat16()
    lo = (u - hi.float()).bfloat16()
    planes = torch.stack((hi, lo))                                               # [pl 2, k 4, co, ci, dy 3]
    planes = planes.reshape(2, 4, cout // 64, 2, 32, cin // 16, 2, 8, 3)         # pl, k, ct, m, r, s, cblk, ci, dy
    planes = planes.index_select(4, split2d_row_perm().to(planes.device))
    return planes.permute(2, 5, 8, 1, 3, 0, 6, 4, 7).contiguous()               # ct, s, dy, k, m, pl, cblk, r, ci


def pack_split3d_weight(w5: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3,3] -> the per-depth-tap packs of pack_split2d_weight stacked behind the cout tile:
    [Cout/64][dz 3][Cin/16][dy 3][plane 2][dx 3][cblk 2][row 64][ci 8] (bf16) for dinv_conv3x3x3_split"""
    return torch.stack([pack_split2d_weight(w5[:, :, dz]) for dz in range(3)], dim=1).contiguous()


def pack_winograd_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> U = G g G^T (fp64, rounded once) packed [Cout/64][Cin/8][ci 8][co 64][16];
    needs Cin % 16 == 0, Cin >= 32 and Cout % 64 == 0 (the ResBlock convs; the kernel pipelines channel blocks
    in pairs and peels the last four)."""
    cout, cin = w.shape[:2]
    if cin % 16 or cin < 32 or cout % 64:
        raise ValueError(f"winograd packing needs cin % 16 == 0, cin >= 32 and cout % 64 == 0, got {cin},{cout}")
    G = [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]]
    u = filter_transform(G, w.detach().double()).float().reshape(cout // 64, 64, cin // 8, 8, 16)
    return u.permute(0, 2, 3, 1, 4).contiguous()


def filter_transform(G, g: torch.Tensor) -> torch.Tensor:
    """G g G^T over the last two dimensions of g [..., 3, 3] for an n x 3 matrix G given as nested Python floats, as
    scalar-coefficient combinations of slices (fp64 elementwise arithmetic where the weights live): no matrix-multiply library is
    called for an n x 3 . 3 x 3 . 3 x n product per filter (a batched fp64 GEMM through rocBLAS / Tensile otherwise)"""
    def comb(coefs, parts):
        acc = None
        for c, p in zip(coefs, parts):
            if c != 0.0:
                acc = p * c if acc is None else acc + p * c
        return acc if acc is not None else torch.zeros_like(parts[0])

    rows = torch.stack([comb(Gi, [g[..., a, :] for a in range(3)]) for Gi in G], dim=-2)          # [..., n, 3] = G g
    return torch.stack([comb(Gj, [rows[..., :, b] for b in range(3)]) for Gj in G], dim=-1)       # [..., n, n] = (G g) G^T


def winograd4_matrices(dtype=torch.float64, device=None):
    """(B^T, G, A^T) of Winograd F(4x4, 3x3) (interpolation points 0, +-1, +-2, inf; Lavin & Gray 2016):
    Y = A^T [ (G g G^T) . (B^T d B) ] A for a 6x6 input patch d and a 3x3 filter g"""
    bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                       [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=dtype, device=device)
    g = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                      [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=dtype, device=device)
    at = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                      dtype=dtype, device=device)
    return bt, g, at


# Winograd point 6 * row + col held by each of the 36 point slots of csrc/drunet_wino4.hip: every wave's nine consecutive slots
# are a full row of the 6 x 6 transform followed by half a row (the halves of rows 1 and 4), so that every wave's epilogue
# retires the same six accumulators first
WINOGRAD4_POINT_SLOTS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17, 9, 10, 11,
                         18, 19, 20, 21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35, 27, 28, 29)


def pack_winograd4_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> U = G g G^T of Winograd F(4x4,3x3) (fp64, rounded once to fp32), packed for
    csrc/drunet_wino4.hip as the MFMA A fragments of the wave that uses them:
    [Cout/64][Cin/8][wave = 4 c2 + q][k 9][lane = 32 h + r][m 4] holds U[point slot 9 q + k][cout 64 ct + 32 c2 + r][cin 8 cb + 4 h + m];
    needs Cin % 16 == 0 and Cout % 64 == 0"""
    cout, cin = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"winograd F(4,3) packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    _, G, _ = winograd4_matrices(device="cpu")
    u = filter_transform(G.tolist(), w.detach().double()).float()       # [co, ci, 6, 6]
    u = u.reshape(cout, cin, 36)[:, :, list(WINOGRAD4_POINT_SLOTS)]     # slot order: see WINOGRAD4_POINT_SLOTS
    u = u.reshape(cout // 64, 2, 32, cin // 8, 2, 4, 4, 9)              # ct, c2, r, cb, h, m, q, k
    return u.permute(0, 3, 1, 6, 7, 4, 2, 5).contiguous()               # ct, cb, c2, q, k, h, r, m


def pack_winograd4_bf16x3_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout,Cin,3,3] -> the U of pack_winograd4_weight (fp64, rounded once to fp32) as its exact THREE-part bf16 split
    u = uh + um + ul (round to nearest even each), packed for dinv_conv3x3_winograd4_bf16x3: per (cout tile, channel block, wave,
    point) 1536 bytes = [lane 64][um 4 | uh 4] then [lane 64][ul 4] (bf16): the (um, uh) and (uh, ul) operand windows of the
    wave's three bf16 MFMAs, loaded as 16 + 8 bytes per lane.  Returns a bfloat16 tensor [Cout/64][Cin/8][8][9][768]"""
    u = pack_winograd4_weight(w)                                        # ct, cb, c2, q, k, h, r, m   (fp32)
    hi = u.bfloat16()
    r1 = u - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    ct, cb = u.shape[:2]
    mh = torch.stack((mid, hi), dim=-2).reshape(ct, cb, 8, 9, 512)      # [h, r][part 2][m 4]
    return torch.cat((mh, lo.reshape(ct, cb, 8, 9, 256)), dim=-1).contiguous()


def pack_down_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv2d k2 s2 weight [Cout,Cin,2,2] -> [tap=dy*2+dx][Cin/8][Cout][8]"""
    cout, cin = w.shape[:2]
    return w.detach().float().permute(2, 3, 1, 0).reshape(4, cin // 8, 8, cout).permute(0, 1, 3, 2).contiguous()


def pack_down_bf16s_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv2d k2 s2 weight [Cout,Cin,2,2] -> two-part bf16 split packed [tap=dy*2+dx][Cin/16][plane 2][cblk 2][Cout][ci 8]"""
    cout, cin = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"bf16-split down packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    w = w.detach().float()
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    planes = torch.stack((hi, lo)).reshape(2, cout, cin // 16, 2, 8, 2, 2)        # pl, co, s, cblk, ci, dy, dx
    return planes.permute(5, 6, 2, 0, 3, 1, 4).contiguous()                       # dy, dx, s, pl, cblk, co, ci


def pack_down_bf16x3_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv2d k2 s2 weight [Cout,Cin,2,2] -> THREE-part bf16 split (hi, mid, lo: 24 significand bits) packed
    [tap=dy*2+dx][Cin/16][plane 3][cblk 2][Cout][ci 8] for dinv_conv_down2x2_bf16x3 (six products: fp32-equivalent)"""
    cout, cin = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"bf16x3 down packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    w = w.detach().float()
    hi = w.bfloat16()
    r1 = w - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    planes = torch.stack((hi, mid, lo)).reshape(3, cout, cin // 16, 2, 8, 2, 2)   # pl, co, s, cblk, ci, dy, dx
    return planes.permute(5, 6, 2, 0, 3, 1, 4).contiguous()                       # dy, dx, s, pl, cblk, co, ci


def pack_up_bf16s_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d k2 s2 weight [Cin,Cout,2,2] -> two-part bf16 split packed [Cin/16][tap=dy*2+dx][plane 2][cblk 2][Cout][ci 8]"""
    cin, cout = w.shape[:2]
    if cin % 16 or cout % 64:
        raise ValueError(f"bf16-split up packing needs cin % 16 == 0 and cout % 64 == 0, got {cin},{cout}")
    w = w.detach().float()
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    planes = torch.stack((hi, lo)).reshape(2, cin // 16, 2, 8, cout, 2, 2)        # pl, s, cblk, ci, co, dy, dx
    return planes.permute(1, 5, 6, 0, 2, 4, 3).contiguous()                       # s, dy, dx, pl, cblk, co, ci


def pack_up_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d k2 s2 weight [Cin,Cout,2,2] -> [tap=dy*2+dx][Cin/8][Cout][8]"""
    cin, cout = w.shape[:2]
    return w.detach().float().permute(2, 3, 0, 1).reshape(4, cin // 8, 8, cout).permute(0, 1, 3, 2).contiguous()


# Packed weights of the training / 3-D paths are cached per source tensor: an unfolded network calls the denoiser
# `max_iter` times per training step with the same weights and every layer needs a pack (a dozen small torch ops) per
# call otherwise - more host time than the kernels take.  One entry per (kind, storage address, shape, sub): the entry
# remembers the tensor's version counter, and a call that finds another version REPLACES the entry (an optimizer step
# bumps the counter of every parameter, so the packs of the previous step are dropped right there instead of piling up).
# An entry keeps its source tensor alive, so the address cannot be handed to another tensor while the entry exists.
# Not detected: in-place updates through `p.data` (hand-written SGD / EMA: `.data` ops do not bump the counter) and
# inference-mode tensors (no counter) - call `clear_pack_cache()` after such an update (load_state_dict, torch.optim and
# `deepinv_amd.training.Trainer` need nothing: they bump the counter / call it).
_PACKS: dict = {}


def clear_pack_cache():
    """drop every cached weight pack (after in-place weight updates the version counter does not see)"""
    _PACKS.clear()


def cached_pack(kind, w, make, sub=0):
    try:
        ver = w._version
    except RuntimeError:      # inference tensors carry no version counter
        ver = -1
    key = (kind, w.data_ptr(), tuple(w.shape), sub)
    hit = _PACKS.get(key)
    if hit is None or hit[1] != ver:
        if hit is None and len(_PACKS) > 4096:
            _PACKS.clear()
        hit = _PACKS[key] = (make(), ver, w)
    return hit[0]


def pack_input(g, x, sigma, act):
    dev = x.device
    if isinstance(sigma, torch.Tensor):
        if sigma.numel() == 1:
            mode, sp, sv = 0, None, float(sigma.item()) if not sigma.is_cuda else None
            if sv is None:  # device scalar: avoid a host sync, expand on device
                sigma = sigma.reshape(1).expand(g.batch).contiguous().float()
                mode, sp, sv = 1, sigma, 0.0
        elif sigma.numel() == g.batch:
            mode, sp, sv = 1, sigma.reshape(-1).contiguous().float().to(dev), 0.0
        else:
            mode, sp, sv = 2, sigma.contiguous().float().to(dev), 0.0
    else:
        mode, sp, sv = 0, None, float(sigma)
    check(_l().dinv_act_pack(ctypes.byref(g), ptr(x), x.shape[1], ptr(sp), mode, sv, ptr(act), stream_ptr(dev)))


def unpack_output(g, act, cout, y):
    check(_l().dinv_act_unpack(ctypes.byref(g), ptr(act), cout, ptr(y), stream_ptr(y.device)))


# ---- optional live profiling: HIP events around every conv3x3 launch on the launch stream
_prof = None


def profile_begin():
    global _prof
    _prof = []


def profile_end():
    """per-kernel totals of the 3x3 conv launches since profile_begin():
    {kernel: {"ms", "launches", "direct_flops" (2*9*Cin*Cout*pixels), "mfma_flops" (executed on the MFMA pipe:
    the same for the direct kernel, 16/36 of it per 2x2 tile for Winograd F(2x2,3x3))}}"""
    global _prof
    rec, _prof = _prof, None
    out = {}
    if not rec:
        return out
    torch.cuda.synchronize()
    for e0, e1, name, direct, mfma in rec:
        d = out.setdefault(name, {"ms": 0.0, "launches": 0, "direct_flops": 0.0, "mfma_flops": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += 1
        d["direct_flops"] += direct
        d["mfma_flops"] += mfma
    return out


def conv3x3(g, x, wpk, cin, cout, y, cout_valid=None, x2=None, res1=None, res2=None, relu=False, cin_valid=None):
    """wpk: packed weight tensor [cout/MT][cin/8][9][MT][8]; MT is read off its shape"""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _conv3x3(g, x, wpk, cin, cout, y, cout_valid, x2, res1, res2, relu)
        e1.record()
        co = cout if cout_valid is None else cout_valid
        ci = cin if cin_valid is None else cin_valid
        fl = 2.0 * 9 * ci * co * g.batch * g.height * g.width
        _prof.append((e0, e1, "conv3x3_kernel", fl, fl))
        return
    _conv3x3(g, x, wpk, cin, cout, y, cout_valid, x2, res1, res2, relu)


def _conv3x3(g, x, wpk, cin, cout, y, cout_valid=None, x2=None, res1=None, res2=None, relu=False):
    check(_l().dinv_conv3x3(ctypes.byref(g), ptr(x), ptr(x2), ptr(wpk), cin, cout, cout if cout_valid is None else cout_valid,
                            int(wpk.shape[3]), ptr(y), ptr(res1), ptr(res2), int(relu), stream_ptr(y.device)))


def pack_conv3x3x3_weight(w5: torch.Tensor) -> tuple[torch.Tensor, int, int]:
    """[Cout, Cin, 3, 3, 3] -> fp32 pack of dinv_conv3x3x3: [cout/MT][dz][cin/8][9 taps][MT][8] (zero padded; MT = 16 for
    Cout <= 16 - the thin-layer kernel -, else 64 / 32 as in pack_conv3x3_weight).  Returns (packed, cin_p, cout_p)."""
    cout, cin = w5.shape[:2]
    cin_p = (cin + 7) // 8 * 8
    if cout <= 16:
        mt, cout_p = 16, 16
    else:
        cout_p = (cout + 31) // 32 * 32
        mt = 64 if cout_p % 64 == 0 else 32
    wp = torch.zeros((cout_p, cin_p, 3, 3, 3), device=w5.device, dtype=torch.float32)
    wp[:cout, :cin] = w5.detach().float()
    # (co/mt, mt, ci/8, 8, dz, 9) -> (co/mt, dz, ci/8, 9, mt, 8)
    wp = wp.reshape(cout_p // mt, mt, cin_p // 8, 8, 3, 9).permute(0, 4, 2, 5, 1, 3).contiguous()
    return wp, cin_p, cout_p


def conv3x3x3(g, x, wpk, cin, cout, y, depth, cout_valid=None, res1=None, relu=False, gate=False):
    """fp32 3x3x3 convolution of volumes stored as stacks of depth + 2 slices, ONE launch (depth taps inside the K loop
    of csrc/drunet.hip: conv3x3_kernel / conv3_thin_kernel); wpk from pack_conv3x3x3_weight; views as for conv3x3x3_split.
    gate (thin kernel): y = res1 > 0 ? conv : 0"""
    check(_l().dinv_conv3x3x3(ctypes.byref(g), ptr(x), ptr(wpk), cin, cout, cout if cout_valid is None else cout_valid,
                              int(wpk.shape[4]), ptr(y), ptr(res1), int(relu) | (2 if gate else 0), int(depth), stream_ptr(y.device)))


def conv3x3_tail(g, x, wtail, cin, cout, y, x2=None):
    """last layer on the vector ALU: y[:cout] = conv3x3(x (+x2)); wtail from pack_tail_weight"""
    check(_l().dinv_conv3x3_tail(ctypes.byref(g), ptr(x), ptr(x2), ptr(wtail), cin, cout, ptr(y), stream_ptr(y.device)))


def conv3x3_split(g, x, wsplit, cin, cout, y, res1=None, relu=False, x_presplit=False, y_presplit=False, gate=False):
    """y = [relu](conv3x3(x)) (+res1) on the bf16 matrix cores, two-part exact operand split, 2-D pixel tiles
    (csrc/drunet_split2d.hip); wsplit from pack_split2d_weight.  `x_presplit` / `y_presplit`: the activation buffer holds
    (8 bf16 high parts | 8 bf16 low parts) per pixel and channel block instead of 8 fp32 values (same 32 bytes)."""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    flags = (1 if x_presplit else 0) | (2 if y_presplit else 0) | (4 if relu else 0) | (8 if gate else 0)   # gate: y = res1 > 0 ? conv : 0
    check(_l().dinv_conv3x3_split(ctypes.byref(g), ptr(x), ptr(wsplit), cin, cout, ptr(y), ptr(res1), flags,
                                  stream_ptr(y.device)))
    if _prof is not None:
        e1.record()
        fl = 2.0 * 9 * cin * cout * g.batch * g.height * g.width
        _prof.append((e0, e1, "conv3x3_split2d_kernel", fl, 3.0 * fl))


def conv3x3_wsplit(g, x, wws, cin, cout, y, res1=None, relu=False):
    """y = [relu](conv3x3(x)) (+res1): Winograd F(2,3) along rows on the bf16 matrix cores, two-part operand split
    (csrc/drunet_wsplit.hip); wws from pack_wsplit_weight; even image width"""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_l().dinv_conv3x3_wsplit(ctypes.byref(g), ptr(x), ptr(wws), cin, cout, ptr(y), ptr(res1), 4 if relu else 0,
                                   stream_ptr(y.device)))
    if _prof is not None:
        e1.record()
        fl = 2.0 * 9 * cin * cout * g.batch * g.height * g.width
        _prof.append((e0, e1, "conv3x3_wsplit_kernel", fl, 2.0 * fl))      # 12 of 18 multiplies, three products each


def conv3x3x3_split(g, x, wsplit, cin, cout, y, depth, res1=None, relu=False, x_presplit=False, y_presplit=False, gate=False):
    """3x3x3 convolution of volumes stored as stacks of depth + 2 slices, one launch (depth taps inside the K loop of
    csrc/drunet_split2d.hip); wsplit from pack_split3d_weight; x / y / res1 are views whose first plane is a volume's
    leading zero slice, with one more readable plane on each side of x"""
    flags = (1 if x_presplit else 0) | (2 if y_presplit else 0) | (4 if relu else 0) | (8 if gate else 0)   # gate: y = res1 > 0 ? conv : 0
    check(_l().dinv_conv3x3x3_split(ctypes.byref(g), ptr(x), ptr(wsplit), cin, cout, ptr(y), ptr(res1), flags, int(depth),
                                    stream_ptr(y.device)))


def conv3x3_winograd(g, x, wino, cin, cout, y, res1=None, relu=False):
    """y = [relu](conv3x3(x)) (+res1) via Winograd F(2x2,3x3); wino from pack_winograd_weight"""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_l().dinv_conv3x3_winograd(ctypes.byref(g), ptr(x), ptr(wino), cin, cout, ptr(y), ptr(res1), int(relu),
                                     stream_ptr(y.device)))
    if _prof is not None:
        e1.record()
        tiles = g.batch * ((g.height + 1) // 2) * ((g.width + 1) // 2)
        _prof.append((e0, e1, "conv3x3_wino_kernel", 2.0 * 9 * cin * cout * g.batch * g.height * g.width,
                      2.0 * 16 * cin * cout * tiles))


_W4_WS: dict = {}


def winograd4_workspace(device):
    """the workspace of dinv_conv3x3_winograd4's tail split for the CURRENT stream of `device` (zero-filled once; the library
    keeps its ticket words zero between launches).  Launches on one stream are ordered, so one buffer serves every layer issued
    there; two streams of a device must not share one (their tail parts would take each other's tickets and partial outputs),
    hence the key (device, stream)."""
    if not WINOGRAD4_TAIL_SPLIT:
        return None
    device = torch.device(device)
    if device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    else:
        key = (str(device), 0)
    ws = _W4_WS.get(key)
    if ws is None:
        ws = _W4_WS[key] = torch.zeros(_l().dinv_conv3x3_winograd4_workspace_bytes(), device=device, dtype=torch.uint8)
    return ws


def winograd4_last_split():
    """(parts per tail tile, tail tiles per XCD) of this thread's last conv3x3_winograd4 launch"""
    f, n = ctypes.c_int32(0), ctypes.c_int32(0)
    check(_l().dinv_conv3x3_winograd4_last_split(ctypes.byref(f), ctypes.byref(n)))
    return f.value, n.value


def conv3x3_winograd4(g, x, wino4, cin, cout, y, res1=None, relu=False, workspace=None):
    """y = [relu](conv3x3(x)) (+res1) via Winograd F(4x4,3x3) on the fp32 matrix cores (csrc/drunet_wino4.hip); wino4 from
    pack_winograd4_weight; height and width multiples of 4; `workspace` (winograd4_workspace(device)) enables the tail split"""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_l().dinv_conv3x3_winograd4(ctypes.byref(g), ptr(x), ptr(wino4), cin, cout, ptr(y), ptr(res1), int(relu),
                                      ptr(workspace), 0 if workspace is None else workspace.numel(), stream_ptr(y.device)))
    if _prof is not None:
        e1.record()
        tiles = g.batch * (g.height // 4) * (g.width // 4)
        _prof.append((e0, e1, "conv3x3_wino4_kernel", 2.0 * 9 * cin * cout * g.batch * g.height * g.width,
                      2.0 * 36 * cin * cout * tiles))


def conv3x3_winograd4_bf16x3(g, x, wino4x3, cin, cout, y, res1=None, relu=False, workspace=None):
    """conv3x3_winograd4 with the multiplies as a three-part bf16 split, six products on the bf16 matrix cores (fp32-equivalent:
    csrc/drunet_wino4.hip, BF3); wino4x3 from pack_winograd4_bf16x3_weight"""
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_l().dinv_conv3x3_winograd4_bf16x3(ctypes.byref(g), ptr(x), ptr(wino4x3), cin, cout, ptr(y), ptr(res1), int(relu),
                                             ptr(workspace), 0 if workspace is None else workspace.numel(), stream_ptr(y.device)))
    if _prof is not None:
        e1.record()
        tiles = g.batch * (g.height // 4) * (g.width // 4)
        # executed: 6 bf16 products per fp32 multiply of the F(4x4) form
        _prof.append((e0, e1, "conv3x3_wino4_kernel<bf16x3>", 2.0 * 9 * cin * cout * g.batch * g.height * g.width,
                      2.0 * 6 * 36 * cin * cout * tiles))


def down2x2(gi, go, x, w, cin, cout, y):
    check(_l().dinv_conv_down2x2(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(w), cin, cout, ptr(y), stream_ptr(y.device)))


def down2x2_bf16s(gi, go, x, wsplit, cin, cout, y):
    """the same 2x2 stride-2 convolution on the bf16 matrix cores (weights from pack_down_bf16s_weight)"""
    check(_l().dinv_conv_down2x2_bf16s(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(wsplit), cin, cout, ptr(y),
                                       stream_ptr(y.device)))


def down2x2_bf16x3(gi, go, x, wsplit3, cin, cout, y):
    """the same 2x2 stride-2 convolution with the three-part operand split and six products: fp32-equivalent
    (weights from pack_down_bf16x3_weight)"""
    check(_l().dinv_conv_down2x2_bf16x3(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(wsplit3), cin, cout, ptr(y),
                                        stream_ptr(y.device)))


def up2x2_bf16s(gi, go, x, x2, wsplit, cin, cout, y):
    """the same transposed convolution on the bf16 matrix cores (weights from pack_up_bf16s_weight)"""
    check(_l().dinv_conv_up2x2_bf16s(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(x2), ptr(wsplit), cin, cout, ptr(y),
                                     stream_ptr(y.device)))


def up2x2(gi, go, x, x2, w, cin, cout, y):
    check(_l().dinv_conv_up2x2(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(x2), ptr(w), cin, cout, ptr(y),
                               stream_ptr(y.device)))


def conv_wgrad(gs, gl, s, m, l, n, taps, dw=None, accumulate=False):
    """dw[m][n][t] (+)= sum_p S[m][p] L[n][map(p) + off_t] (csrc/drunet_bwd.hip); returns dw of shape [m, n, k, k]"""
    k = 3 if taps == 9 else 2
    if dw is None:
        dw = torch.empty((m, n, k, k), device=s.device, dtype=torch.float32)
        accumulate = False
    ws = torch.empty(_l().dinv_conv_wgrad_workspace_bytes(ctypes.byref(gs), m, n, taps), device=s.device, dtype=torch.uint8)
    check(_l().dinv_conv_wgrad(ctypes.byref(gs), ctypes.byref(gl), ptr(s), m, ptr(l), n, taps, ptr(dw), int(accumulate), ptr(ws),
                               ws.numel(), stream_ptr(s.device)))
    return dw


def conv_wgrad_3x3x3(g, s, m, l_first, n, depth_stride):
    """[m, n, 3, 3, 3] weight gradient of a 3x3x3 layer in one call: the three depth taps are the second grid dimension of ONE launch
    and of one reduction (csrc/drunet_bwd.hip: dinv_conv_wgrad_3x3x3); l_first = the layer input shifted by -1 slice"""
    dw = torch.empty((m, n, 3, 3, 3), device=s.device, dtype=torch.float32)
    ws = torch.empty(3 * _l().dinv_conv_wgrad_workspace_bytes(ctypes.byref(g), m, n, 9), device=s.device, dtype=torch.uint8)
    check(_l().dinv_conv_wgrad_3x3x3(ctypes.byref(g), ptr(s), m, ptr(l_first), n, int(depth_stride), ptr(dw), 0, ptr(ws), ws.numel(),
                                     stream_ptr(s.device)))
    return dw


def relu_backward(act, grad):
    """grad <- grad * (act > 0) in place, on whole activation buffers"""
    check(_l().dinv_relu_backward(grad.numel(), ptr(act), ptr(grad), stream_ptr(grad.device)))
    return grad


def down2x2_bf16s_3d(gi, go, x, wsplit, cin, cout, y, depth_out, dz, accumulate):
    """depth tap dz of a 2x2x2 stride-2 convolution (volumes as stacks of D + 2 slices); wsplit = pack_down_bf16s_weight(w[:, :, dz])"""
    check(_l().dinv_conv_down2x2_bf16s_3d(ctypes.byref(gi), ctypes.byref(go), ptr(x), ptr(wsplit), cin, cout, ptr(y), depth_out, dz,
                                          int(accumulate), stream_ptr(y.device)))


def up2x2_bf16s_3d(gi, go, x, wsplit, cin, cout, y, depth_in, dz):
    """depth tap dz of a 2x2x2 stride-2 transposed convolution: writes the slices 2 z + dz of the output volume"""
    check(_l().dinv_conv_up2x2_bf16s_3d(ctypes.byref(gi), ctypes.byref(go), ptr(x), None, ptr(wsplit), cin, cout, ptr(y), depth_in,
                                        dz, stream_ptr(y.device)))


def conv_wgrad_3d(gs, gl, s, m, l, n, depth_s, dz):
    """[m, n, 2, 2] weight gradient of depth tap dz of a 2x2x2 stride-2 layer"""
    dw = torch.empty((m, n, 2, 2), device=s.device, dtype=torch.float32)
    ws = torch.empty(_l().dinv_conv_wgrad_workspace_bytes(ctypes.byref(gs), m, n, 4), device=s.device, dtype=torch.uint8)
    check(_l().dinv_conv_wgrad_3d(ctypes.byref(gs), ctypes.byref(gl), ptr(s), m, ptr(l), n, ptr(dw), 0, ptr(ws), ws.numel(), depth_s,
                                  dz, stream_ptr(s.device)))
    return dw
