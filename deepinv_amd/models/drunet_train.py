"""DRUNet forward + backward on the HIP kernels (2-D), for training through ``deepinv.unfolded``
(deepinv/unfolded/unfolded.py:116-226 with a DRUNet prior, deepinv/models/drunet.py:39-263).

The reference differentiates the 64 convolutions with autograd (ATen / cuDNN kernels).  Here the whole network is ONE
``torch.autograd.Function``:

* forward: the same kernels as inference (bf16-split 3x3 / 2x2 convolutions, direct fp32 kernel for the thin head and
  tail), every ResBlock input and post-ReLU activation kept for the backward pass;
* data gradients: the forward kernels again, on re-packed weights - d/dx of a 3x3 convolution is the 3x3 convolution
  with the transposed, spatially flipped filter; d/dx of the 2x2 stride-2 convolution is the 2x2 stride-2 transposed
  convolution with the same filter, and vice versa;
* weight gradients: ``dinv_conv_wgrad`` (csrc/drunet_bwd.hip, fp32 matrix cores, deterministic);
* ReLU backward and the skip-connection sums: ``dinv_relu_backward`` / ``dinv_lincomb``.

Gradients are returned for the input image, the noise-level map (the trainable ``g_param`` of unfolded PnP) and every
convolution weight.  Double backward is not supported.

ReLU masks: the bf16-split convolutions are fp32-class (a few 1e-6 relative), which is enough to flip ``relu'(z)`` for
the handful of pre-activations that lie within that distance of zero; one flipped mask entry changes one row of a
weight gradient by ~1/sqrt(pixels) (measured: 5e-3 .. 1e-2 on the 256 / 512-channel levels at 2 x 12 x 16 pixels, against
7e-7 without flips; data gradients are unaffected: 3e-6).  The FORWARD pass of the training path therefore runs on the
fp32 matrix cores by default (direct kernels, 3e-7 per layer: the masks are those of an fp32 reference);
``DINV_DRUNET_TRAIN_PRECISION=bf16s`` uses the inference kernels instead (faster forward, mask flips at the 1e-6 level).
The backward pass always uses the bf16-split kernels for the data gradients (no ReLU decision depends on them)."""
from __future__ import annotations


import torch

from ..hip import drunet as K
from ..hip import elementwise as ew


def supported(model) -> bool:
    """any 4-level 2-D DRUNet: channel counts that are not multiples of 64 (what the 2x2 kernels take) are zero-padded -
    padded channels stay exactly zero through convolutions, ReLUs and residual adds, their gradients are sliced away"""
    return model.dim == 2 and len(model.nc) == 4


def _r64(c):
    return (c + 63) // 64 * 64


def _pad_w(w, d0, d1):
    if w.shape[0] == d0 and w.shape[1] == d1:
        return w
    out = torch.zeros((d0, d1, *w.shape[2:]), device=w.device, dtype=torch.float32)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _blk(model, prefix, k):
    """parameter-name prefix of ResBlock k of a stage (with nb = 1 the body is a bare ResBlock: 'm_body', not 'm_body.0')"""
    return prefix if (prefix == "m_body" and model.nb == 1) else f"{prefix}.{k}"


def _flip_t(w):
    """filter of the data-gradient convolution: [Cout,Cin,3,3] -> [Cin,Cout,3,3], taps reversed"""
    return w.flip(2, 3).transpose(0, 1).contiguous()


def _fp32_forward(model) -> bool:
    v = getattr(model, "train_forward_precision", "fp32")
    if v not in ("fp32", "bf16split"):
        raise ValueError(f"train_forward_precision must be fp32 or bf16split, got {v}")
    return v == "fp32"


def _conv3(g, w, x, relu=False, res1=None, fp32=False, flip=False, gate=None):
    """y = [relu](conv3x3(x, w)) (+ res1) on activation buffers; bf16-split kernel where the shapes allow it.
    flip: convolve with the transposed, tap-reversed filter (the data gradient of the same layer); packs are cached per
    weight tensor and version (hip/drunet.py: cached_pack).  gate: the forward pass's ReLU output whose sign masks the
    result (ReLU backward fused into the epilogue of the data-gradient convolution)"""
    cout, cin = (w.shape[1], w.shape[0]) if flip else w.shape[:2]
    y = K.alloc(g, cout, x.device)
    src = lambda: _flip_t(w) if flip else w  # noqa: E731
    if cout % 64 == 0 and cin % 16 == 0 and not fp32:
        K.conv3x3_split(g, x, K.cached_pack(("c3s", flip), w, lambda: K.pack_split2d_weight(src())), cin, cout, y,
                        res1=gate if gate is not None else res1, relu=relu, gate=gate is not None)
    else:
        wpk, ci_p, co_p = K.cached_pack(("c3d", flip), w, lambda: K.pack_conv3x3_weight(src()))
        K.conv3x3(g, x, wpk, ci_p, co_p, y, cout_valid=cout, res1=res1, relu=relu)
        if gate is not None:
            K.relu_backward(gate, y)
    return y


def _down(gi, go, w, x, fp32=False):
    """2x2 stride-2 convolution with a [Cout,Cin,2,2] filter (also: data gradient of the transposed convolution)"""
    cout, cin = w.shape[:2]
    y = K.alloc(go, cout, x.device)
    if cin % 16 == 0 and not fp32:
        K.down2x2_bf16s(gi, go, x, K.cached_pack("dns", w, lambda: K.pack_down_bf16s_weight(w)), cin, cout, y)
    else:
        K.down2x2(gi, go, x, K.cached_pack("dnd", w, lambda: K.pack_down_weight(w)), cin, cout, y)
    return y


def _up(gi, go, w, x, fp32=False):
    """2x2 stride-2 transposed convolution with a [Cin,Cout,2,2] filter (also: data gradient of the strided one)"""
    cin, cout = w.shape[:2]
    y = K.alloc(go, cout, x.device)
    if cin % 16 == 0 and not fp32:
        K.up2x2_bf16s(gi, go, x, None, K.cached_pack("ups", w, lambda: K.pack_up_bf16s_weight(w)), cin, cout, y)
    else:
        K.up2x2(gi, go, x, None, K.cached_pack("upd", w, lambda: K.pack_up_weight(w)), cin, cout, y)
    return y


def _add(a, b):
    return ew.lincomb(1.0, a, 1.0, b)


class DRUNetFunction(torch.autograd.Function):
    """``y = DRUNet(xin)`` with ``xin = cat(image, noise map)``; parameters are passed explicitly (named_parameters order)"""

    @staticmethod
    def forward(ctx, model, xin, *params):
        names = [n for n, _ in model.named_parameters()]
        W = {}
        for n, p in zip(names, params):
            p = p.detach().float()
            if n == "m_head.weight":
                W[n] = _pad_w(p, _r64(p.shape[0]), p.shape[1])
            elif n == "m_tail.weight":
                W[n] = _pad_w(p, p.shape[0], _r64(p.shape[1]))
            else:
                W[n] = _pad_w(p, _r64(p.shape[0]), _r64(p.shape[1]))
        nb, nc = model.nb, model.nc
        dev = xin.device
        B, C, H, Wd = xin.shape
        g = [K.geom(B, H >> i, Wd >> i) for i in range(4)]
        xin = xin.detach().contiguous().float()
        x_act = K.alloc(g[0], C, dev)
        K.pack_input(g[0], xin[:, :-1].contiguous(), xin[:, -1:].contiguous(), x_act)
        saved = {"x_act": x_act, "res": {}, "down_in": {}, "up_in": {}}
        f32 = _fp32_forward(model)

        def res_chain(gl, prefix, first, cur):
            for k in range(first, first + nb):
                a1 = _conv3(gl, W[f"{_blk(model, prefix, k)}.res.0.weight"], cur, relu=True, fp32=f32)
                out = _conv3(gl, W[f"{_blk(model, prefix, k)}.res.2.weight"], a1, res1=cur, fp32=f32)
                saved["res"][f"{prefix}.{k}"] = (cur, a1)
                cur = out
            return cur

        x1 = _conv3(g[0], W["m_head.weight"], x_act)
        skips = [x1]
        cur = x1
        for i, name in enumerate(("m_down1", "m_down2", "m_down3")):
            r = res_chain(g[i], name, 0, cur)
            saved["down_in"][name] = r
            cur = _down(g[i], g[i + 1], W[f"{name}.{nb}.weight"], r, fp32=f32)
            skips.append(cur)
        cur = res_chain(g[3], "m_body", 0, cur)
        for i, name in zip((2, 1, 0), ("m_up3", "m_up2", "m_up1")):
            s = _add(cur, skips[i + 1])
            saved["up_in"][name] = s
            cur = _up(g[i + 1], g[i], W[f"{name}.0.weight"], s, fp32=f32)
            cur = res_chain(g[i], name, 1, cur)
        s0 = _add(cur, x1)
        saved["tail_in"] = s0
        y_act = _conv3(g[0], W["m_tail.weight"], s0)
        y = torch.empty((B, model.out_channels, H, Wd), device=dev, dtype=torch.float32)
        K.unpack_output(g[0], y_act, model.out_channels, y)
        ctx.model, ctx.names, ctx.W, ctx.g, ctx.saved = model, names, W, g, saved
        ctx.shapes = {n: tuple(p.shape) for n, p in zip(names, params)}
        ctx.in_channels = C
        return y

    @staticmethod
    def backward(ctx, gy):
        model, names, W, g, saved = ctx.model, ctx.names, ctx.W, ctx.g, ctx.saved
        nb, nc = model.nb, model.nc
        dev = gy.device
        want_w = any(ctx.needs_input_grad[2:])
        dW = {}

        def wgrad(name, gs, gl, s, l, taps):
            if want_w:
                m, n = W[name].shape[:2]
                sh = ctx.shapes[name]
                dW[name] = K.conv_wgrad(gs, gl, s, m, l, n, taps)[:sh[0], :sh[1]].contiguous()

        def res_back(gl, prefix, first, gout):
            for k in range(first + nb - 1, first - 1, -1):
                x_in, a1 = saved["res"][f"{prefix}.{k}"]
                w1, w2 = W[f"{_blk(model, prefix, k)}.res.0.weight"], W[f"{_blk(model, prefix, k)}.res.2.weight"]
                wgrad(f"{_blk(model, prefix, k)}.res.2.weight", gl, gl, gout, a1, 9)
                gt = _conv3(gl, w2, gout, flip=True, gate=a1)
                wgrad(f"{_blk(model, prefix, k)}.res.0.weight", gl, gl, gt, x_in, 9)
                gout = _conv3(gl, w1, gt, res1=gout, flip=True)
            return gout

        gy = gy.contiguous().float()
        gy_act = K.alloc(g[0], model.out_channels, dev)
        K.pack_input(g[0], gy, 0.0, gy_act)
        wgrad("m_tail.weight", g[0], g[0], gy_act, saved["tail_in"], 9)
        gcur = _conv3(g[0], W["m_tail.weight"], gy_act, flip=True)
        gskip = {0: gcur}                     # s0 = u0 + x1
        for i, name in zip((0, 1, 2), ("m_up1", "m_up2", "m_up3")):
            gcur = res_back(g[i], name, 1, gcur)
            wu = W[f"{name}.0.weight"]        # [Cin = nc[i+1], Cout = nc[i], 2, 2]
            wgrad(f"{name}.0.weight", g[i + 1], g[i], saved["up_in"][name], gcur, 4)
            gcur = _down(g[i], g[i + 1], wu, gcur)          # d/ds of convT(s, wu) = conv_s2 with the same filter
            gskip[i + 1] = gcur               # s_{i+1} = (level i+1 result) + x_{i+2}
        gcur = _add(res_back(g[3], "m_body", 0, gcur), gskip[3])
        for i, name in zip((2, 1, 0), ("m_down3", "m_down2", "m_down1")):
            wd = W[f"{name}.{nb}.weight"]     # [Cout = nc[i+1], Cin = nc[i], 2, 2]
            wgrad(f"{name}.{nb}.weight", g[i + 1], g[i], gcur, saved["down_in"][name], 4)
            gcur = _up(g[i + 1], g[i], wd, gcur)            # d/dr of conv_s2(r, wd) = convT with the same filter
            gcur = _add(res_back(g[i], name, 0, gcur), gskip[i])
        wgrad("m_head.weight", g[0], g[0], gcur, saved["x_act"], 9)
        gx = None
        if ctx.needs_input_grad[1]:
            gin_act = _conv3(g[0], W["m_head.weight"], gcur, flip=True)
            B, H, Wd = g[0].batch, g[0].height, g[0].width
            gx = torch.empty((B, ctx.in_channels, H, Wd), device=dev, dtype=torch.float32)
            K.unpack_output(g[0], gin_act, ctx.in_channels, gx)
        ctx.saved = None                      # free the activations
        grads = [dW.get(n) if need else None for n, need in zip(names, ctx.needs_input_grad[2:])]
        return (None, gx, *grads)


def forward_train(model, xin):
    """DRUNet(xin) recorded as one autograd node (all parameters of `model` are inputs of the node)"""
    return DRUNetFunction.apply(model, xin, *[p for _, p in model.named_parameters()])
