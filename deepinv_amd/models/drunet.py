"""DRUNet denoiser on the hand-written HIP kernels (csrc/drunet*.hip) - inference, training and ``dim=3``.

Module tree and parameter names are identical to the reference (deepinv/models/drunet.py:39-101:
``m_head``, ``m_down{1,2,3}``, ``m_body``, ``m_up{3,2,1}``, ``m_tail``; ResBlock ``res.0``/``res.2``)
so reference ``state_dict``s / ``.pth`` checkpoints load unchanged.  The ``nn.Conv*`` modules only hold the
parameters: no forward of this class runs a PyTorch-ROCm (MIOpen) convolution, and an architecture the kernels do
not cover raises instead of falling back (the PyTorch graph of the same module lives in ``tests/torch_drunet.py``,
where it serves as an independent GPU reference).

* inference (``torch.no_grad`` / eval, 2-D): 64 HIP launches, activations stay in padded channel-blocked planes;
* gradients requested (``deepinv.unfolded``): ``models/drunet_train.py`` (forward + backward as one autograd node);
* ``dim=3``: ``models/drunet3d.py``.
"""
from __future__ import annotations


import torch
import torch.nn as nn

from ..hip import HipExtensionError
from ..hip import drunet as K
from . import drunet3d, drunet_train
from .base import Denoiser


# The ONE precision switch of the denoiser: how the 56 ResBlock 3x3 convolutions multiply.
#   "fp32":      fp32 multiplies on the fp32 matrix cores (Winograd F(4x4,3x3) kernel; F(2x2,3x3) / direct kernels for the
#                shapes it does not take) - the reference's arithmetic type, and the default since round 4
#   "bf16split": every fp32 operand as two bf16 parts, three products on the bf16 matrix cores, fp32 accumulation
#                (csrc/drunet_wsplit.hip / drunet_split2d.hip; <= 2^-16 per operand, 4-6e-6 per layer, DRUNet output within 1e-4 of
#                the fp32 path): the throughput setting, 15-20 % faster per convolution, opt-in
# Default for new models: `deepinv_amd.models.drunet.DEFAULT_CONV_PRECISION`; per model: `model.conv_precision = "bf16split"`.
CONV_PRECISIONS = ("bf16split", "fp32")
DEFAULT_CONV_PRECISION = "fp32"


def _conv_nd(dim):
    return {2: nn.Conv2d, 3: nn.Conv3d}[dim]


def _convT_nd(dim):
    return {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dim]


class ResBlock(nn.Module):
    """x + conv(relu(conv(x)))  (drunet.py:403-434, mode 'CRC')"""

    def __init__(self, channels, dim=2):
        super().__init__()
        C = _conv_nd(dim)
        self.res = nn.Sequential(C(channels, channels, 3, 1, 1, bias=False), nn.ReLU(inplace=True),
                                 C(channels, channels, 3, 1, 1, bias=False))

    def forward(self, x):
        return x + self.res(x)


def weights_init_drunet(m):
    """orthogonal init, gain 0.2 (drunet.py:689-692)"""
    if m.__class__.__name__.find("Conv") != -1:
        nn.init.orthogonal_(m.weight.data, gain=0.2)


def run_replicate_padded(net, x, multiple=16):
    """Evaluate `net` on `x` grown (edge replication, at the far end of every spatial axis) to the next multiple of `multiple`,
    and give back the original extent - what the reference does for shapes its U-Net cannot take directly
    (deepinv/models/drunet.py:252-256 -> models/utils.py:49-61)."""
    extent = tuple(x.shape[2:])
    grow = [(-n) % multiple for n in extent]
    pads = []
    for extra in reversed(grow):            # F.pad lists the last axis first, (near, far) per axis
        pads += [0, extra]
    out = net(nn.functional.pad(x, pads, mode="replicate")) if any(grow) else net(x)
    return out[(slice(None), slice(None)) + tuple(slice(0, n) for n in extent)]


def run_four_windows(net, x, field=32):
    """Evaluate `net` on the four corner windows of a 2-D input - each reaching `field`-aligned past the centre, so that the half
    of the image a window is responsible for lies at least a receptive field away from its cut - and stitch the four owned
    halves together (the reference's strategy for large shapes that are not multiples of 8: drunet.py:257-262 ->
    models/utils.py:64-98)."""
    H, W = x.shape[-2:]
    span = lambda n: (n // 2 // field + 1) * field
    out = None
    for top in (True, False):
        rows = slice(0, span(H)) if top else slice(H - span(H), H)
        for left in (True, False):
            cols = slice(0, span(W)) if left else slice(W - span(W), W)
            piece = net(x[..., rows, cols])
            if out is None:
                out = piece.new_zeros(*piece.shape[:2], H, W)
            # a near window owns the first half of the axis (the first n // 2 entries of its output), a far window the rest
            # (the last n - n // 2 entries of its output)
            pr = slice(0, H // 2) if top else slice(piece.shape[-2] - (H - H // 2), piece.shape[-2])
            pc = slice(0, W // 2) if left else slice(piece.shape[-1] - (W - W // 2), piece.shape[-1])
            out[..., slice(0, H // 2) if top else slice(H // 2, H), slice(0, W // 2) if left else slice(W // 2, W)] = piece[..., pr, pc]
    return out


_CUS: dict = {}


def _compute_units(device) -> int:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CUS:
        _CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count) if device.type == "cuda" else 256
    return _CUS[idx]


class DRUNet(Denoiser):
    def __init__(self, in_channels=3, out_channels=3, nc=(64, 128, 256, 512), nb=4, act_mode="R",
                 downsample_mode="strideconv", upsample_mode="convtranspose", pretrained=None,
                 pretrained_2d_isotropic=False, device=None, dim=2):
        super().__init__()
        if act_mode != "R" or downsample_mode != "strideconv" or upsample_mode != "convtranspose":
            raise NotImplementedError("only the default DRUNet architecture (ReLU / strideconv / convtranspose) is "
                                      "on the accelerated path")
        dim = int(str(dim).lower().replace("d", "")) if not isinstance(dim, int) else dim
        if dim not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        C, T = _conv_nd(dim), _convT_nd(dim)
        self.in_channels, self.out_channels, self.nc, self.nb = in_channels, out_channels, tuple(nc), nb
        cin = in_channels + 1  # + noise level map
        self.m_head = C(cin, nc[0], 3, 1, 1, bias=False)
        self.m_down1 = nn.Sequential(*[ResBlock(nc[0], dim) for _ in range(nb)], C(nc[0], nc[1], 2, 2, 0, bias=False))
        self.m_down2 = nn.Sequential(*[ResBlock(nc[1], dim) for _ in range(nb)], C(nc[1], nc[2], 2, 2, 0, bias=False))
        self.m_down3 = nn.Sequential(*[ResBlock(nc[2], dim) for _ in range(nb)], C(nc[2], nc[3], 2, 2, 0, bias=False))
        # a single module is not wrapped (the reference's `sequential` helper, drunet.py:279-297): with nb = 1 the body's
        # state_dict keys are m_body.res.{0,2}.weight
        self.m_body = ResBlock(nc[3], dim) if nb == 1 else nn.Sequential(*[ResBlock(nc[3], dim) for _ in range(nb)])
        self.m_up3 = nn.Sequential(T(nc[3], nc[2], 2, 2, 0, bias=False), *[ResBlock(nc[2], dim) for _ in range(nb)])
        self.m_up2 = nn.Sequential(T(nc[2], nc[1], 2, 2, 0, bias=False), *[ResBlock(nc[1], dim) for _ in range(nb)])
        self.m_up1 = nn.Sequential(T(nc[1], nc[0], 2, 2, 0, bias=False), *[ResBlock(nc[0], dim) for _ in range(nb)])
        self.m_tail = C(nc[0], out_channels, 3, 1, 1, bias=False)
        self.dim = dim
        self.conv_precision = DEFAULT_CONV_PRECISION
        # forward pass of the TRAINING node: "fp32" keeps the ReLU masks identical to an fp32 reference,
        # "bf16split" runs the inference kernels
        self.train_forward_precision = "fp32"
        if pretrained is not None:
            if pretrained in ("download", "download_2d"):
                raise RuntimeError("no network access: pass pretrained=<path to .pth> or None")
            self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=True)
            self.eval()
        else:
            self.apply(weights_init_drunet)
        self._engine = None
        if device is not None:
            self.to(device)

    def _noise_map(self, x, sigma):
        """drunet.py:226-249"""
        if isinstance(sigma, torch.Tensor):
            if sigma.ndim > 0:
                if sigma.shape == (x.size(0), 1, *x.shape[2:]):
                    return sigma
                if sigma.shape in [(x.size(0),), (x.size(0), 1, *[1] * self.dim)] or sigma.numel() == 1:
                    m = sigma.reshape(-1, 1, *[1] * self.dim).to(x)
                    return m.expand(x.size(0), 1, *x.shape[2:])
                raise ValueError("Incorrect shape, sigma should be of shape (1,), (batch_size,) or "
                                 f"(batch_size, 1, height, width, (depth)), got {tuple(sigma.shape)}")
            return torch.ones((x.size(0), 1, *x.shape[2:]), device=x.device) * sigma.to(x.device)
        return torch.full((x.size(0), 1, *x.shape[2:]), float(sigma), device=x.device, dtype=x.dtype)

    def _sigma_operand(self, x, sigma):
        """the noise level in the form the pack kernel takes it - a float, [B] values or a [B,1,H,W] map - after the shape checks
        of `_noise_map` (drunet.py:226-249)"""
        if not isinstance(sigma, torch.Tensor):
            return float(sigma)
        if sigma.ndim == 0 or sigma.numel() == 1:
            return sigma if sigma.is_cuda else float(sigma)
        if sigma.shape == (x.size(0), 1, *x.shape[2:]) or sigma.shape in [(x.size(0),), (x.size(0), 1, *[1] * self.dim)]:
            return sigma
        raise ValueError("Incorrect shape, sigma should be of shape (1,), (batch_size,) or "
                         f"(batch_size, 1, height, width, (depth)), got {tuple(sigma.shape)}")

    def _use_hip(self, x):
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        return self.dim == 2 and not needs_grad

    def _use_hip_train(self):
        """gradients requested: the hand-written backward (models/drunet_train.py) for the 2-D architectures it covers"""
        return drunet_train.supported(self)

    def _precision(self):
        if self.conv_precision not in CONV_PRECISIONS:
            raise ValueError(f"conv_precision must be one of {CONV_PRECISIONS}, got {self.conv_precision!r}")
        return self.conv_precision

    def forward(self, x, sigma):
        if not x.is_cuda:
            raise HipExtensionError("deepinv_amd.models.DRUNet runs only on a HIP device; there is no CPU fallback")
        if self._use_hip(x):
            if all(v % 8 == 0 and v > 31 for v in x.shape[2:]):
                # the pack kernel writes the noise-level channel itself (scalar, one value per sample, or a map): no concatenated
                # copy of the input is built
                return self._hip_forward(x, self._sigma_operand(x, sigma))
            run = lambda inp: self._hip_forward(inp[:, :-1], inp[:, -1:])
        elif drunet3d.supported(self):
            run = lambda inp: drunet3d.forward3d(self, inp)             # volumes as stacks of slices on the 2-D kernels
        elif self._use_hip_train():
            run = lambda inp: drunet_train.forward_train(self, inp)     # forward AND backward on the HIP kernels
        else:   # no second backend: what the HIP kernels do not cover is an error, not a PyTorch-ROCm graph
            raise NotImplementedError(f"DRUNet(dim={self.dim}, nc={self.nc}) with gradients: only 4-level architectures are "
                                      "covered by the hand-written HIP forward / backward (models/drunet_train.py, drunet3d.py)")
        xin = torch.cat((x, self._noise_map(x, sigma)), 1)
        safe = all(s % 8 == 0 and s > 31 for s in xin.shape[2:])
        if safe:
            return run(xin)
        if self.training or any(xin.size(2 + i) < 64 for i in range(self.dim)):
            return run_replicate_padded(run, xin, multiple=16)
        if self.dim == 3:
            raise NotImplementedError("the four-window evaluation of large shapes that are not multiples of 8 is 2-D only (as in the reference)")
        return run_four_windows(run, xin, field=64)

    # ------------------------------------------------------------------ MFMA inference engine
    def _weights_version(self):
        # tensors created under torch.inference_mode() carry no version counter (reading it raises)
        return (tuple(0 if p.is_inference() else p._version for p in self.parameters())
                + tuple(p.data_ptr() for p in self.parameters()))

    def _prepare(self, device):
        ver = (self._weights_version(), self._precision(), K.FP32_WINOGRAD4_BF16X3)     # the packs depend on the precision
        if self._engine is not None and self._engine["ver"] == ver and self._engine["device"] == device:
            return self._engine
        e = {"ver": ver, "device": device, "ws": {}, "split": self._precision() == "bf16split"}
        split = e["split"]

        def c3(m):
            """packs of one 3x3 conv: direct fp32 (64- and 32-wide cout tiles; _pick() chooses per launch geometry) and, per
            precision, the bf16-split pack or the fp32 Winograd pack"""
            w = m.weight.to(device)
            p64 = K.pack_conv3x3_weight(w)
            p32 = K.pack_conv3x3_weight(w, mt=32) if p64[0].shape[3] == 64 else p64
            ok = w.shape[0] % 64 == 0 and w.shape[1] % 16 == 0
            s2d = K.pack_split2d_weight(w) if (split and ok) else None
            wino = K.pack_winograd_weight(w) if (not split and ok and w.shape[1] >= 32) else None
            wsp = K.pack_wsplit_weight(w) if (split and ok) else None
            wino4 = K.pack_winograd4_weight(w) if (not split and ok) else None
            wino4x3 = K.pack_winograd4_bf16x3_weight(w) if (not split and ok and K.FP32_WINOGRAD4_BF16X3) else None
            return (p64, p32, wino, s2d, wsp, wino4, wino4x3)

        e["head"] = c3(self.m_head)
        e["tail"] = c3(self.m_tail)
        wt = self.m_tail.weight
        e["tail_valu"] = K.pack_tail_weight(wt.to(device)) if (wt.shape[0] <= 4 and wt.shape[1] % 8 == 0) else None
        for name in ("m_down1", "m_down2", "m_down3"):
            seq = getattr(self, name)
            e[name] = [(c3(b.res[0]), c3(b.res[2])) for b in list(seq)[:-1]]
            wd = seq[-1].weight.to(device)
            e[name + "_s"] = K.pack_down_weight(wd)
            okd = wd.shape[0] % 64 == 0 and wd.shape[1] % 16 == 0
            e[name + "_sb"] = K.pack_down_bf16s_weight(wd) if (split and okd) else None
            # fp32 setting: three-part bf16 split, six products (fp32-equivalent; the fp32-MFMA kernel is bound by that pipe)
            e[name + "_s3"] = K.pack_down_bf16x3_weight(wd) if (not split and okd) else None
        body = [self.m_body] if isinstance(self.m_body, ResBlock) else list(self.m_body)
        e["m_body"] = [(c3(b.res[0]), c3(b.res[2])) for b in body]
        for name in ("m_up3", "m_up2", "m_up1"):
            seq = getattr(self, name)
            wu = seq[0].weight.to(device)
            e[name + "_s"] = K.pack_up_weight(wu)
            e[name + "_sb"] = K.pack_up_bf16s_weight(wu) if (split and wu.shape[0] % 16 == 0 and wu.shape[1] % 64 == 0) else None
            e[name] = [(c3(b.res[0]), c3(b.res[2])) for b in list(seq)[1:]]
        self._engine = e
        return e

    def _workspace(self, e, B, H, W, device, lane=0):
        key = (B, H, W, lane)
        ws = e["ws"].get(key)
        if ws is None:
            if any(k[1:3] != (H, W) or k[3] == lane for k in e["ws"]):
                e["ws"].clear()  # one image geometry at a time keeps the footprint bounded (its batch lanes live side by side)
            nc = self.nc
            g = [K.geom(B, H >> i, W >> i) for i in range(4)]
            cin_p = e["head"][0][1]
            ws = {"g": g, "in": K.alloc(g[0], cin_p, device), "out": K.alloc(g[0], self.out_channels, device)}
            for i in range(4):
                # skip tensor x_{i+1}, two ping-pong buffers and the ResBlock temporary
                for nm in ("skip", "a", "b", "t"):
                    ws[f"{nm}{i}"] = K.alloc(g[i], nc[i], device)
            e["ws"][key] = ws
        return ws

    @staticmethod
    def _pick(g, packs):
        """64-wide cout tiles unless that grid would leave the chip under-filled (< 3 rounds of the 512 resident
        workgroup slots): then 32-wide tiles double the number of workgroups (small per-GPU batches)."""
        p64, p32 = packs[:2]
        if p64[0].shape[3] == 64 and ((g.np + 255) // 256) * (p64[2] // 64) < 1536:
            return p32
        return p64

    def _conv_fp32(self, g, pk, x, y, relu=False, res1=None):
        """one ResBlock convolution in fp32 arithmetic: Winograd F(4x4,3x3) kernel (csrc/drunet_wino4.hip) where the image
        sides are multiples of 4 and the launch has enough 64-cout x 32-tile workgroup tiles to occupy the chip, else the
        F(2x2,3x3) kernel, else the direct MFMA kernel"""
        if (pk[5] is not None and K.FP32_WINOGRAD_TILE == 4 and g.height % 4 == 0 and g.width % 4 == 0
                and -(-g.batch * (g.height // 4) * (g.width // 4) // 32) * (pk[0][2] // 64) >= K.WINOGRAD4_MIN_TILES):
            # inside a batch lane the last, incomplete round of tiles is left to the other lane (see batch_lanes) - unless the
            # WHOLE launch is less than one round (fewer tiles than compute units: the deep levels of a small lane), where cutting
            # the tiles along the input channels is what fills the chip
            tiles = -(-g.batch * (g.height // 4) * (g.width // 4) // 32) * (pk[0][2] // 64)
            wsp = K.winograd4_workspace(x.device) if (self._tail_split or tiles < _compute_units(x.device)) else None
            if pk[6] is not None:       # (K.FP32_WINOGRAD4_BF16X3 when the packs were built)
                K.conv3x3_winograd4_bf16x3(g, x, pk[6], pk[0][1], pk[0][2], y, res1=res1, relu=relu, workspace=wsp)
            else:
                K.conv3x3_winograd4(g, x, pk[5], pk[0][1], pk[0][2], y, res1=res1, relu=relu, workspace=wsp)
            return
        if pk[2] is not None:
            K.conv3x3_winograd(g, x, pk[2], pk[0][1], pk[0][2], y, res1=res1, relu=relu)
            return
        (w, ci, co) = self._pick(g, pk)
        K.conv3x3(g, x, w, ci, co, y, relu=relu, res1=res1)

    def _res_block(self, g, pk1, pk2, x, t, y):
        """y = x + conv2(relu(conv1(x))) (drunet.py:403-434); `t` is scratch.  bf16-split precision: Winograd F(2,3) along
        rows on the bf16 matrix cores (csrc/drunet_wsplit.hip: 1.5x fewer matrix instructions, measured 12-25 % faster per
        level at 4 and 32 slices) wherever the image width is even; else the direct kernel, whose conv1 writes its ReLU output
        pre-split (the parts conv2 would form anyway) so that conv2 stages it by plain copies"""
        if pk1[4] is not None and pk2[4] is not None and g.width % 2 == 0:
            K.conv3x3_wsplit(g, x, pk1[4], pk1[0][1], pk1[0][2], t, relu=True)
            K.conv3x3_wsplit(g, t, pk2[4], pk2[0][1], pk2[0][2], y, res1=x)
        elif pk1[3] is not None and pk2[3] is not None:
            K.conv3x3_split(g, x, pk1[3], pk1[0][1], pk1[0][2], t, relu=True, y_presplit=True)
            K.conv3x3_split(g, t, pk2[3], pk2[0][1], pk2[0][2], y, res1=x, x_presplit=True)
        else:
            self._conv_fp32(g, pk1, x, t, relu=True)
            self._conv_fp32(g, pk2, t, y, res1=x)

    def _res_chain(self, g, blocks, c, x, a, b, t):
        """run ResBlocks: returns the buffer holding the result (never `x` itself is overwritten)"""
        cur = x
        bufs = [a, b]
        for i, (pk1, pk2) in enumerate(blocks):
            dst = bufs[i % 2]
            self._res_block(g, pk1, pk2, cur, t, dst)
            cur = dst
        return cur

    # ---- batch lanes: the batch cut into `batch_lanes` contiguous parts, each run through the whole network on its own HIP stream.
    # The units of a batch are independent through the network, and every convolution launch ends in an incomplete round of
    # workgroups (a persistent workgroup per compute unit; 800 tiles over 256 units = 3.125 rounds at 4 slices) during which most
    # of the chip idles: with two lanes the other lane's launch fills the units as they fall idle, and the lanes' epilogues (the
    # HBM bursts of a launch) no longer coincide.  1 = one launch sequence (default).  Results do not depend on it (every unit is
    # computed by the same instruction sequence; only the tile a unit shares with a neighbour of the batch changes).
    # "auto" = two lanes for every batch of at least two units.  Measured on MI355X (cfg2, graph replay, ms per 50-iteration step;
    # one lane with the channel split of the last round / two lanes with it / two lanes without it): 4 slices 327 / 311 / 304,
    # 8 slices 602 / 538 / 513, 16 slices 1081 / 1060 / 987, 32 slices 2013 / 2039 / 1952 (profiles/r06_lanes.jsonl).  With a second
    # lane filling the idle units, cutting the last round's tiles along the input channels (csrc/drunet_wino4.hip: partial outputs
    # through a workspace, a ticket, the last part adds them) only costs - the lanes run the F(4x4) launches WITHOUT it, and a
    # unit's result then does not depend on the batch it is computed in at all (no summation order depends on the tile round).
    batch_lanes = "auto"
    LANES_MIN_PIXELS = 4 * 128 * 128
    _tail_split = True

    def _lanes(self, x):
        B = x.shape[0]
        if not x.is_cuda and x.device.type != "meta":     # (host emulation of the kernels, tests/emu_backend.py: no streams)
            return 1
        if B * x.shape[-2] * x.shape[-1] < self.LANES_MIN_PIXELS:      # launches of a few tiles: nothing to fill, nothing to time
            return 1
        if self.batch_lanes == "auto":
            return 2 if B >= 2 else 1
        return max(1, min(int(self.batch_lanes), B))

    def _run_lanes(self, x, sigma_map, streams):
        """the batch lanes of one call on the given streams (one contiguous part of the batch each); the current stream waits for
        all of them"""
        from ..hip import split_batch

        B, dev = x.shape[0], x.device
        y = torch.empty((B, self.out_channels, *x.shape[2:]), device=dev, dtype=torch.float32)
        cur = torch.cuda.current_stream(dev)
        for i, (s, (b0, b1)) in enumerate(zip(streams, split_batch(B, len(streams)))):
            sg = sigma_map
            if isinstance(sg, torch.Tensor) and sg.numel() > 1:      # one value per sample, or a map: this lane's units
                sg = sg.reshape(B, *sg.shape[1:])[b0:b1] if sg.shape[0] == B else sg
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                self._tail_split = False        # (see batch_lanes: the other lane fills the last round)
                try:
                    self._hip_forward_lane(x[b0:b1], sg, lane=i, out=y[b0:b1])
                finally:
                    self._tail_split = True
        for s in streams:
            cur.wait_stream(s)
        return y

    def _calibrated_lane_streams(self, x, sigma_map, lanes):
        """The streams of the lanes on this device - decided ONCE per process and (device, lanes) by timing this very call: one launch
        sequence against `lanes` concurrent ones, on up to four candidate stream sets from PyTorch's pool.  Two lanes whose streams
        landed on one hardware queue run one after the other and lose (hip/__init__.py: _LANE_STREAMS); such a set is passed over,
        and when no set is at least as fast as the single sequence the lanes are switched off for the process (None).  Not decided
        while a HIP graph is being captured (no timing there): one lane until an eager call has calibrated."""
        from .. import hip as H

        key = H.lane_key(x.device, lanes)
        if key in H._LANE_STREAMS:
            return H._LANE_STREAMS[key]
        if torch.cuda.is_current_stream_capturing():
            return None
        dev = x.device

        def gpu_ms(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)

        self._hip_forward_lane(x, sigma_map)                    # packs the weights, creates the buffers
        t_one = min(gpu_ms(lambda: self._hip_forward_lane(x, sigma_map)) for _ in range(3))     # minima of three: one slow sample
        # (another process on the host, a clock ramp) must not decide for the whole process
        best = None
        for attempt in range(4):
            cand = [torch.cuda.Stream(dev) for _ in range(lanes)]
            self._run_lanes(x, sigma_map, cand)                 # (the lanes' own buffers)
            t = min(gpu_ms(lambda: self._run_lanes(x, sigma_map, cand)) for _ in range(3))
            if t <= 1.03 * t_one and (best is None or t < best[0]):
                best = (t, cand)
            if best is not None and (attempt >= 1 or t < 0.9 * t_one):
                break
        H._LANE_STREAMS[key] = best[1] if best is not None else None
        self._lane_calibration = {"one_sequence_ms": round(t_one, 3), "lanes_ms": None if best is None else round(best[0], 3),
                                  "lanes": lanes, "attempts": attempt + 1, "batch": int(x.shape[0])}
        return H._LANE_STREAMS[key]

    def _hip_forward(self, x, sigma_map):
        lanes = self._lanes(x)
        if lanes > 1:
            x = x.contiguous().float()
            streams = self._calibrated_lane_streams(x, sigma_map, lanes)
            if streams is not None:
                return self._run_lanes(x, sigma_map, streams)
        return self._hip_forward_lane(x, sigma_map)

    def _hip_forward_lane(self, x, sigma_map, lane=0, out=None):
        """forward_unet (drunet.py:200-210) as 64 conv launches + pack/unpack."""
        dev = x.device
        e = self._prepare(dev)
        B, _, H, W = x.shape
        ws = self._workspace(e, B, H, W, dev, lane)
        g = ws["g"]
        nc = self.nc
        x = x.contiguous().float()
        K.pack_input(g[0], x, sigma_map, ws["in"])
        (wh, cih, coh) = self._pick(g[0], e["head"])
        K.conv3x3(g[0], ws["in"], wh, cih, coh, ws["skip0"], cin_valid=self.in_channels + 1)  # x1
        cur = ws["skip0"]
        downs = ("m_down1", "m_down2", "m_down3")
        for i, name in enumerate(downs):
            r = self._res_chain(g[i], e[name], nc[i], cur, ws[f"a{i}"], ws[f"b{i}"], ws[f"t{i}"])
            if e[name + "_sb"] is not None:   # same arithmetic as the ResBlock convs
                K.down2x2_bf16s(g[i], g[i + 1], r, e[name + "_sb"], nc[i], nc[i + 1], ws[f"skip{i + 1}"])   # x2, x3, x4
            elif e[name + "_s3"] is not None:
                K.down2x2_bf16x3(g[i], g[i + 1], r, e[name + "_s3"], nc[i], nc[i + 1], ws[f"skip{i + 1}"])
            else:
                K.down2x2(g[i], g[i + 1], r, e[name + "_s"], nc[i], nc[i + 1], ws[f"skip{i + 1}"])
            cur = ws[f"skip{i + 1}"]
        r = self._res_chain(g[3], e["m_body"], nc[3], cur, ws["a3"], ws["b3"], ws["t3"])
        skip_add = ws["skip3"]  # x + x4 is fused into the up-conv's operand load
        for i, name in zip((2, 1, 0), ("m_up3", "m_up2", "m_up1")):
            if e[name + "_sb"] is not None:
                K.up2x2_bf16s(g[i + 1], g[i], r, skip_add, e[name + "_sb"], nc[i + 1], nc[i], ws[f"t{i}"])
            else:
                K.up2x2(g[i + 1], g[i], r, skip_add, e[name + "_s"], nc[i + 1], nc[i], ws[f"t{i}"])
            # t{i} holds the up-conv output; run the ResBlocks with a/b ping-pong and a fresh temporary
            r = self._res_chain_from_t(g[i], e[name], ws[f"t{i}"], ws[f"a{i}"], ws[f"b{i}"])
            skip_add = ws[f"skip{i}"]
        if e["tail_valu"] is not None:   # m_tail(x + x1)
            K.conv3x3_tail(g[0], r, e["tail_valu"], nc[0], self.out_channels, ws["out"], x2=ws["skip0"])
        else:
            (wt, cit, cot) = self._pick(g[0], e["tail"])
            K.conv3x3(g[0], r, wt, cit, cot, ws["out"], cout_valid=self.out_channels, x2=ws["skip0"])
        y = out if out is not None else torch.empty((B, self.out_channels, H, W), device=dev, dtype=torch.float32)
        K.unpack_output(g[0], ws["out"], self.out_channels, y)
        return y

    def _res_chain_from_t(self, g, blocks, t_in, a, b):
        """ResBlocks whose input lives in the `t` buffer: rotate roles so nothing is clobbered."""
        cur, tmp, other = t_in, a, b
        for (pk1, pk2) in blocks:
            self._res_block(g, pk1, pk2, cur, tmp, other)
            cur, other = other, cur
        return cur
