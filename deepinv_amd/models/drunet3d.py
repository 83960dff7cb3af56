"""DRUNet with ``dim=3`` (Conv3d / ConvTranspose3d, deepinv/models/drunet.py:39-263) forward + backward on the 2-D HIP
kernels - BASELINE config 4's denoiser (unfolded PGD on 3-D multi-coil MRI, deepinv/unfolded/unfolded.py:116-226).

A volume of D slices lives in the padded channel-blocked activation layout as D + 2 consecutive "images" (one zero
slice at each end), so that

* a 3x3x3 convolution is ONE launch of the 2-D-tile kernels with the three depth taps inside the K loop (tap dz reads
  the input shifted by dz - 1 slices), ReLU / residual / zeroed padding slices in the epilogue: ``dinv_conv3x3x3_split``
  (bf16-split arithmetic) or ``dinv_conv3x3x3`` (fp32; layers of <= 16 output channels on the 16x16x4 MFMA tile);
* the 2x2x2 stride-2 convolution / transposed convolution are two launches of the 2-D kernels with the slice pairing
  z <-> 2 z + dz done inside the kernel (``dinv_conv_down2x2_bf16s_3d`` / ``dinv_conv_up2x2_bf16s_3d``);
* data gradients are the same operators on re-packed weights (flipped + transposed 3x3x3 filters; down <-> up);
* weight gradients are ``dinv_conv_wgrad`` per depth tap (shifted views) and ``dinv_conv_wgrad_3d`` for the 2x2x2 layers.

Activation buffers are allocated with channel counts rounded up to 64 and every kernel gets its weights zero-padded to
what IT needs (bf16-split kernels: Cin to 16, Cout to 64; fp32 direct kernel: Cin to 8, Cout to 32; weight gradients:
the true counts) - the small config-4 network has 16 / 32 / 64 / 128 channels; padded channels stay exactly zero
through convolutions, ReLUs and residual adds.
The forward pass of the 3x3x3 convolutions runs on the fp32 matrix cores when gradients are requested (same reasoning
as models/drunet_train.py: ReLU masks of an fp32 reference), on the bf16-split kernels otherwise."""
from __future__ import annotations


import torch

from ..hip import drunet as K
from ..hip import elementwise as ew


def supported(model) -> bool:
    return model.dim == 3 and len(model.nc) == 4


def _r64(c):
    return (c + 63) // 64 * 64


def _r16(c):
    return (c + 15) // 16 * 16


_POOL: dict = {}        # (device, channels, level shape) -> released activation buffers
POOL_MAX_BYTES = None          # cap of the free list; None = 40 % of the device's memory (config 4's training step holds ~45 GB of
                               # activations for its backward pass: under the 16-GiB cap of rounds 3-5 sixty 655-MB buffers per step
                               # fell off the list and were zero-filled again, 5 ms of the 154-ms step)
_POOL_CAPS: dict = {}
CHECK_RECYCLED = False         # debugging aid (the GPU test switches it on): verify the invariant below on every reuse
_pool_bytes = 0
_pool_sig = None               # (device, B, D, H, W) of the last call: another problem shape drops the whole list


def _pool_cap(device):
    if POOL_MAX_BYTES is not None:
        return POOL_MAX_BYTES
    if device not in _POOL_CAPS:
        _POOL_CAPS[device] = int(0.4 * torch.cuda.get_device_properties(device).total_memory) if device.type == "cuda" else 16 << 30
    return _POOL_CAPS[device]


class Vol:
    """activation volume: tensor [C/8, cs, 8] with one guard plane in front, so that slice-shifted views stay inside.

    Buffers are recycled through a free list keyed by (device, channel count, level shape) instead of being zero-filled
    per layer (the fills were 6 % of config 4's training step): every kernel that writes a volume writes ALL of its
    in-range pixels (exact zeros on frames and padding slices) and nothing outside, so a released buffer still has
    zero guard planes, zero slack and zero padded channel blocks - the only things a fresh ``torch.zeros`` adds."""

    def __init__(self, lv, channels, device):
        self.lv = lv
        self.sig = _pool_sig          # buffers of an earlier problem shape are not recycled when they die later
        self.key = (torch.device(device), int(channels), lv.B, lv.D, lv.H, lv.W)
        free = _POOL.get(self.key)
        if free:
            global _pool_bytes
            self.t = free.pop()
            _pool_bytes -= self.t.numel() * 4
            if CHECK_RECYCLED:
                self._check_clean(channels)
        else:
            self.t = torch.zeros((_r64(channels) // 8, lv.g.cs, 8), device=device, dtype=torch.float32)

    def _check_clean(self, channels):
        """the invariant recycling rests on: frames, padding slices, guard planes, slack and padded channel blocks of a
        released buffer are exactly zero (only interior voxels of real channel blocks may hold anything)"""
        lv, t = self.lv, self.t.clone()
        b, d, h, w = lv.B, lv.D, lv.H, lv.W
        vol = t[:, lv.guard + lv.g.sl: lv.guard + lv.g.sl + lv.g.np].view(t.shape[0], b, d + 2, h + 2, -1, 8)
        vol[:(int(channels) + 7) // 8, :, 1:-1, 1:h + 1, 1:w + 1] = 0
        bad = int(torch.count_nonzero(t))
        if bad:
            raise AssertionError(f"recycled activation buffer {self.key}: {bad} non-zero values outside the interior")

    def __del__(self):
        global _pool_bytes
        try:
            t, key = self.t, self.key
            if self.sig == _pool_sig and _pool_bytes + t.numel() * 4 <= _pool_cap(t.device):
                _POOL.setdefault(key, []).append(t)
                _pool_bytes += t.numel() * 4
        except Exception:       # interpreter shutdown
            pass

    def view(self, dz=0):
        return self.t[:, self.lv.guard + dz * self.lv.g.plane:]


class Level:
    def __init__(self, B, D, H, W):
        self.B, self.D, self.H, self.W = B, D, H, W
        self.g = K.geom(B * (D + 2), H, W)
        self.guard = int(self.g.plane)
        self.g.cs = (self.g.cs + 2 * self.guard + 3) // 4 * 4


def _pad_w(w, d0, d1):
    if w.shape[0] == d0 and w.shape[1] == d1:
        return w
    out = torch.zeros((d0, d1, *w.shape[2:]), device=w.device, dtype=torch.float32)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _cached(kind, w, dz, make):
    return K.cached_pack(kind, w, make, sub=dz)


def conv3(lv, w5, x: Vol, relu=False, res: Vol | None = None, fp32=False, flip=False, x_presplit=False,
          y_presplit=False, gate: Vol | None = None) -> Vol:
    """3x3x3 convolution, stride 1, zero padding 1, no bias; w5 [Cout, Cin, 3, 3, 3] (true channel counts).
    flip: convolve with the transposed, tap-reversed filter instead (the data gradient of the same layer);
    gate: the forward pass's ReLU output whose sign masks the result (ReLU backward).
    ONE launch either way (the depth taps are part of the kernel's K loop, ReLU and the zero padding slices in its
    epilogue): bf16-split arithmetic (csrc/drunet_split2d.hip) or fp32 (csrc/drunet.hip: thin head / tail layers and the
    mask-exact training forward; layers of <= 16 output channels on the 16x16x4 MFMA tile)"""
    cout, cin = (w5.shape[1], w5.shape[0]) if flip else w5.shape[:2]
    y = Vol(lv, cout, x.t.device)
    # 16 -> 16 channels is ONE 16 x 16 x 4 MFMA tile of the fp32 thin kernel; the bf16-split kernel pads the outputs to its 64-row
    # tile (measured at config 4's level 0: 0.43 ms against 0.32 ms), so those layers take the fp32 kernel in every setting
    thin = cin <= 16 and cout <= 16
    if cin >= 16 and cout >= 16 and not fp32 and not thin:
        pk = _cached(("c3x3", flip), w5, 0, lambda: K.pack_split3d_weight(_pad_w(_flip_t(w5) if flip else w5, _r64(cout), _r16(cin))))
        r1 = gate if gate is not None else res
        K.conv3x3x3_split(lv.g, x.view(), pk, _r16(cin), _r64(cout), y.view(), lv.D, res1=r1.view() if r1 is not None else None,
                          relu=relu, x_presplit=x_presplit, y_presplit=y_presplit, gate=gate is not None)
        return y
    assert not (x_presplit or y_presplit)
    pk, cip, cop = _cached(("c3f", flip), w5, 0, lambda: K.pack_conv3x3x3_weight(_flip_t(w5) if flip else w5))
    if gate is not None and int(pk.shape[4]) == 16:          # thin kernel: ReLU backward in the epilogue
        assert res is None and not relu
        K.conv3x3x3(lv.g, x.view(), pk, cip, cop, y.view(), lv.D, cout_valid=cout, res1=gate.view(), gate=True)
        return y
    K.conv3x3x3(lv.g, x.view(), pk, cip, cop, y.view(), lv.D, cout_valid=cout, res1=res.view() if res is not None else None,
                relu=relu)
    if gate is not None:
        K.relu_backward(gate.t, y.t)
    return y


def down(lvi, lvo, w5, x: Vol) -> Vol:
    """2x2x2 stride-2 convolution; w5 [Cout, Cin, 2, 2, 2]"""
    cout, cin = w5.shape[:2]
    y = Vol(lvo, cout, x.t.device)
    cop, cip = _r64(cout), _r16(cin)
    for dz in range(2):
        pk = _cached("down", w5, dz, lambda dz=dz: K.pack_down_bf16s_weight(_pad_w(w5[:, :, dz], cop, cip)))
        K.down2x2_bf16s_3d(lvi.g, lvo.g, x.view(), pk, cip, cop, y.view(), lvo.D, dz, dz > 0)
    return y


def up(lvi, lvo, w5, x: Vol) -> Vol:
    """2x2x2 stride-2 transposed convolution; w5 [Cin, Cout, 2, 2, 2]"""
    cin, cout = w5.shape[:2]
    y = Vol(lvo, cout, x.t.device)
    cip, cop = _r16(cin), _r64(cout)
    for dz in range(2):
        pk = _cached("up", w5, dz, lambda dz=dz: K.pack_up_bf16s_weight(_pad_w(w5[:, :, dz], cip, cop)))
        K.up2x2_bf16s_3d(lvi.g, lvo.g, x.view(), pk, cip, cop, y.view(), lvi.D, dz)
    return y


def add(lv, a: Vol, b: Vol) -> Vol:
    out = Vol(lv, a.key[1], a.t.device)
    ew.lincomb(1.0, a.t, 1.0, b.t, out=out.t)
    return out


def release_buffers():
    """drop the recycled activation buffers (kept between calls of the SAME problem shape, at most POOL_MAX_BYTES; a call
    with another shape drops them by itself)"""
    global _pool_bytes
    _POOL.clear()
    _pool_bytes = 0


def _blk(model, prefix, k):
    """parameter-name prefix of ResBlock k of a stage (with nb = 1 the body is a bare ResBlock: 'm_body', not 'm_body.0')"""
    return prefix if (prefix == "m_body" and model.nb == 1) else f"{prefix}.{k}"


def _flip_t(w5):
    return w5.flip(2, 3, 4).transpose(0, 1).contiguous()


def wgrad3(lv, gout: Vol, x: Vol, m, n):
    """[m, n, 3, 3, 3] weight gradient of conv3 (S = dL/dy, L = x shifted by the depth tap)"""
    return K.conv_wgrad_3x3x3(lv.g, gout.view(), m, x.view(-1), n, int(lv.g.plane) * 8)


def wgrad2(lvs, lvl, small: Vol, m, large: Vol, n):
    """[m, n, 2, 2, 2] weight gradient of a 2x2x2 stride-2 layer (S on the half grid, L on the full grid)"""
    return torch.stack([K.conv_wgrad_3d(lvs.g, lvl.g, small.view(), m, large.view(), n, lvs.D, dz) for dz in range(2)], dim=2)


class DRUNet3dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, xin, *params):
        names = [n for n, _ in model.named_parameters()]
        nb, nc = model.nb, model.nc
        dev = xin.device
        B, C, D, H, Wd = xin.shape
        if D % 8 or H % 8 or Wd % 8:
            raise ValueError("3-D DRUNet on the HIP kernels needs depth, height and width to be multiples of 8")
        train = any(ctx.needs_input_grad[1:])
        # 3x3x3 convolutions in fp32 arithmetic (csrc/drunet.hip) or as bf16 split products: the training node follows
        # `train_forward_precision` (ReLU masks identical to an fp32 reference), inference follows `conv_precision`
        f32 = (getattr(model, "train_forward_precision", "fp32") if train else model.conv_precision) == "fp32"
        global _pool_sig
        if _pool_sig != (dev, B, D, H, Wd):       # buffers of another problem shape would never be reused: free them
            release_buffers()
            _pool_sig = (dev, B, D, H, Wd)
        lv = [Level(B, D >> i, H >> i, Wd >> i) for i in range(4)]
        W = {n: p.detach().float() for n, p in zip(names, params)}      # true shapes; each kernel pads what it needs
        # pack the input volume: [B, C, D, H, W] -> slices [B (D+2), C, H, W] with zero end slices
        x2 = torch.nn.functional.pad(xin.detach().float().permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1))
        x2 = x2.reshape(B * (D + 2), C, H, Wd).contiguous()
        x_act = Vol(lv[0], C, dev)
        K.pack_input(lv[0].g, x2[:, :-1].contiguous(), x2[:, -1:].contiguous(), x_act.view())
        saved = {"x_act": x_act, "res": {}, "down_in": {}, "up_in": {}}

        def res_chain(l, prefix, first, cur):
            for k in range(first, first + nb):
                w1, w2 = W[f"{_blk(model, prefix, k)}.res.0.weight"], W[f"{_blk(model, prefix, k)}.res.2.weight"]
                # inference: the ReLU temporary travels pre-split (consumed only by the second convolution); training keeps
                # it in fp32 (the backward pass reads its sign)
                ps = (not train) and min(w1.shape[0], w1.shape[1], w2.shape[0], w2.shape[1]) >= 16 and max(w1.shape[0], w1.shape[1]) > 16
                a1 = conv3(l, w1, cur, relu=True, fp32=f32, y_presplit=ps)
                out = conv3(l, w2, a1, res=cur, fp32=f32, x_presplit=ps)
                if train:
                    saved["res"][f"{prefix}.{k}"] = (cur, a1)
                cur = out
            return cur

        x1 = conv3(lv[0], W["m_head.weight"], x_act, fp32=f32)
        skips = [x1]
        cur = x1
        for i, name in enumerate(("m_down1", "m_down2", "m_down3")):
            r = res_chain(lv[i], name, 0, cur)
            saved["down_in"][name] = r
            cur = down(lv[i], lv[i + 1], W[f"{name}.{nb}.weight"], r)
            skips.append(cur)
        cur = res_chain(lv[3], "m_body", 0, cur)
        for i, name in zip((2, 1, 0), ("m_up3", "m_up2", "m_up1")):
            s = add(lv[i + 1], cur, skips[i + 1])
            saved["up_in"][name] = s
            cur = up(lv[i + 1], lv[i], W[f"{name}.0.weight"], s)
            cur = res_chain(lv[i], name, 1, cur)
        s0 = add(lv[0], cur, x1)
        saved["tail_in"] = s0
        y_act = conv3(lv[0], W["m_tail.weight"], s0, fp32=f32)
        y2 = torch.empty((B * (D + 2), model.out_channels, H, Wd), device=dev, dtype=torch.float32)
        K.unpack_output(lv[0].g, y_act.view(), model.out_channels, y2)
        y = y2.view(B, D + 2, model.out_channels, H, Wd)[:, 1:-1].permute(0, 2, 1, 3, 4).contiguous()
        if train:
            ctx.model, ctx.names, ctx.W, ctx.lv, ctx.saved = model, names, W, lv, saved
            ctx.in_shape = (B, C, D, H, Wd)
        return y

    @staticmethod
    def backward(ctx, gy):
        model, names, W, lv, saved = ctx.model, ctx.names, ctx.W, ctx.lv, ctx.saved
        nb = model.nb
        B, C, D, H, Wd = ctx.in_shape
        dev = gy.device
        want_w = any(ctx.needs_input_grad[2:])
        dW = {}

        def keep(name, grad):
            dW[name] = grad

        def res_back(l, prefix, first, gout):
            for k in range(first + nb - 1, first - 1, -1):
                x_in, a1 = saved["res"][f"{prefix}.{k}"]
                n1, n2 = f"{_blk(model, prefix, k)}.res.0.weight", f"{_blk(model, prefix, k)}.res.2.weight"
                if want_w:
                    keep(n2, wgrad3(l, gout, a1, W[n2].shape[0], W[n2].shape[1]))
                gt = conv3(l, W[n2], gout, flip=True, gate=a1)       # ReLU backward in the epilogue (gate = the forward activation)
                if want_w:
                    keep(n1, wgrad3(l, gt, x_in, W[n1].shape[0], W[n1].shape[1]))
                gout = conv3(l, W[n1], gt, res=gout, flip=True)
            return gout

        g2 = torch.nn.functional.pad(gy.contiguous().float().permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, 0, 1, 1))
        g2 = g2.reshape(B * (D + 2), model.out_channels, H, Wd).contiguous()
        gy_act = Vol(lv[0], model.out_channels, dev)
        K.pack_input(lv[0].g, g2, 0.0, gy_act.view())
        wt = W["m_tail.weight"]
        if want_w:
            keep("m_tail.weight", wgrad3(lv[0], gy_act, saved["tail_in"], wt.shape[0], wt.shape[1]))
        gcur = conv3(lv[0], wt, gy_act, flip=True)
        gskip = {0: gcur}
        for i, name in zip((0, 1, 2), ("m_up1", "m_up2", "m_up3")):
            gcur = res_back(lv[i], name, 1, gcur)
            wu = W[f"{name}.0.weight"]        # [Cin (level i+1), Cout (level i), 2, 2, 2]
            if want_w:
                keep(f"{name}.0.weight", wgrad2(lv[i + 1], lv[i], saved["up_in"][name], wu.shape[0], gcur, wu.shape[1]))
            gcur = down(lv[i], lv[i + 1], wu, gcur)
            gskip[i + 1] = gcur
        gcur = add(lv[3], res_back(lv[3], "m_body", 0, gcur), gskip[3])
        for i, name in zip((2, 1, 0), ("m_down3", "m_down2", "m_down1")):
            wd = W[f"{name}.{nb}.weight"]     # [Cout (level i+1), Cin (level i), 2, 2, 2]
            if want_w:
                keep(f"{name}.{nb}.weight", wgrad2(lv[i + 1], lv[i], gcur, wd.shape[0], saved["down_in"][name], wd.shape[1]))
            gcur = up(lv[i + 1], lv[i], wd, gcur)
            gcur = add(lv[i], res_back(lv[i], name, 0, gcur), gskip[i])
        wh = W["m_head.weight"]
        if want_w:
            keep("m_head.weight", wgrad3(lv[0], gcur, saved["x_act"], wh.shape[0], wh.shape[1]))
        gx = None
        if ctx.needs_input_grad[1]:
            gin = conv3(lv[0], wh, gcur, flip=True)
            g2 = torch.empty((B * (D + 2), C, H, Wd), device=dev, dtype=torch.float32)
            K.unpack_output(lv[0].g, gin.view(), C, g2)
            gx = g2.view(B, D + 2, C, H, Wd)[:, 1:-1].permute(0, 2, 1, 3, 4).contiguous()
        ctx.saved = None
        grads = [dW.get(n) if need else None for n, need in zip(names, ctx.needs_input_grad[2:])]
        return (None, gx, *grads)


def forward3d(model, xin):
    """DRUNet(dim=3)(xin) as one autograd node; xin = cat(volume, noise map) [B, C+1, D, H, W]"""
    return DRUNet3dFunction.apply(model, xin, *[p for _, p in model.named_parameters()])
