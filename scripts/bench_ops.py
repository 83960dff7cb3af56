"""Operator micro-benchmarks on one MI355X: A / A_adjoint time, algorithmic GB/s vs HBM peak.
Usage: python scripts/bench_ops.py [mri2d|mri3d|...]"""
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepinv_amd as dinv  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_mri(B, coils, img, three_d):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 2, *img, generator=g).to(dev)
    maps = (torch.randn(1, coils, *img, dtype=torch.complex64, generator=g) / coils ** 0.5).to(dev)
    mask = (torch.rand(*img, generator=g) > 0.75).float().to(dev)
    phys = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, *img), three_d=three_d, device=dev)
    y = phys.A(x)
    vol = 1
    for n in img:
        vol *= n
    alg = B * 2 * vol * 4 + B * 2 * coils * vol * 4 + coils * vol * 8 + 2 * vol * 4
    tA = timeit(lambda: phys.A(x))
    tT = timeit(lambda: phys.A_adjoint(y))
    tN = timeit(lambda: phys.A_adjoint_A(x))
    tC = timeit(lambda: phys.A_adjoint(phys.A(x)))
    algN = 2 * B * 2 * vol * 4 + coils * vol * 8 + 2 * vol * 4      # read x, write A^T A x (+ maps, mask once)
    print(json.dumps({"op": "MultiCoilMRI.A_adjoint_A", "B": B, "coils": coils, "img": img, "ms": tN * 1e3,
                      "composite_ms": tC * 1e3, "alg_MB": algN / 1e6, "GBps": algN / tN / 1e9,
                      "t_traffic_floor_MB": 4 * B * coils * vol * 8 / 1e6,
                      "scratch_GBps": 4 * B * coils * vol * 8 / tN / 1e9}))
    for name, t in (("A", tA), ("A_adjoint", tT)):
        print(json.dumps({"op": f"MultiCoilMRI.{name}", "B": B, "coils": coils, "img": img, "ms": t * 1e3,
                          "alg_MB": alg / 1e6, "GBps": alg / t / 1e9, "frac_hbm_peak": alg / t / HBM_PEAK}))


def bench_radon(B, W, nang):
    dev = torch.device("cuda:0")
    phys = dinv.physics.Tomography(angles=nang, img_width=W, normalize=False, device=dev)
    x = torch.rand(B, 1, W, W, device=dev)
    y = phys.A(x)
    G = y.shape[2]
    alg = B * (W * W + G * nang) * 4
    for name, fn in (("A", lambda: phys.A(x)), ("A_adjoint", lambda: phys.A_adjoint(y)), ("ramp", lambda: phys.filter(y)),
                     ("fbp", lambda: phys.A_dagger(y, fbp=True))):
        t = timeit(fn, iters=5, warmup=1)
        print(json.dumps({"op": f"Tomography.{name}", "B": B, "W": W, "angles": nang, "ms": t * 1e3, "ms_per_img": t * 1e3 / B,
                          "alg_MB": alg / 1e6, "GBps": alg / t / 1e9, "Gsamples_per_s": B * G * G * nang / t / 1e9}))


def bench_drunet(B, cin, H, W, gflop_per_img, precision=None):
    dev = torch.device("cuda:0")
    model = dinv.models.DRUNet(cin, cin, pretrained=None).to(dev).eval()
    if precision:
        model.conv_precision = precision
    x = torch.rand(B, cin, H, W, device=dev)
    with torch.no_grad():
        t = timeit(lambda: model(x, 0.05), iters=5, warmup=2)
    tf = gflop_per_img * B / t / 1e3
    print(json.dumps({"op": "DRUNet.forward", "conv_precision": model.conv_precision, "B": B, "cin": cin, "img": [H, W], "ms": t * 1e3, "TFLOPs": tf,
                      "frac_fp32_mfma_peak": tf / 157.3}))


def bench_conv_levels(B):
    """direct vs Winograd ResBlock conv at the four DRUNet levels"""
    from deepinv_amd.hip import drunet as K
    dev = torch.device("cuda:0")
    for lvl, c in enumerate((64, 128, 256, 512)):
        H = 320 >> lvl
        g = K.geom(B, H, H)
        x, y = K.alloc(g, c, dev), K.alloc(g, c, dev)
        x.normal_()
        w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
        wd, ci, co = K.pack_conv3x3_weight(w)
        ww = K.pack_winograd_weight(w)
        fl = 2.0 * 9 * c * c * B * H * H
        td = timeit(lambda: K.conv3x3(g, x, wd, ci, co, y, relu=True), iters=20, warmup=3)
        tw = timeit(lambda: K.conv3x3_winograd(g, x, ww, c, c, y, relu=True), iters=20, warmup=3)
        ws = K.pack_split2d_weight(w)
        ts = timeit(lambda: K.conv3x3_split(g, x, ws, c, c, y, relu=True), iters=20, warmup=3)
        r = K.alloc(g, c, dev)
        tsr = timeit(lambda: K.conv3x3_split(g, x, ws, c, c, y, res1=r), iters=20, warmup=3)
        print(json.dumps({"op": "conv3x3", "B": B, "hw": H, "c": c, "direct_ms": td * 1e3, "wino_ms": tw * 1e3,
                          "bf16s_ms": ts * 1e3, "bf16s_res_ms": tsr * 1e3, "direct_TF": fl / td / 1e12,
                          "wino_effTF": fl / tw / 1e12, "bf16s_effTF": fl / ts / 1e12,
                          "bf16s_executed_TF": 3 * fl / ts / 1e12, "bf16s_frac_bf16_peak": 3 * fl / ts / 2.5e15}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["mri2d", "mri3d"]
    if "mri2d" in which:
        bench_mri(32, 8, (320, 320), False)
    if "mri3d" in which:
        bench_mri(2, 12, (16, 256, 256), True)
    if "radon" in which:
        bench_radon(8, 512, 720)
    if "drunet" in which:
        bench_drunet(32, 2, 320, 320, 433.4)
    if "drunet_fp32" in which:
        bench_drunet(32, 2, 320, 320, 433.4, "fp32")
    if "drunet4_fp32" in which:
        bench_drunet(4, 2, 320, 320, 433.4, "fp32")
    if "convlv" in which:
        bench_conv_levels(32)
        bench_conv_levels(4)
    if "drunet4" in which:
        bench_drunet(4, 2, 320, 320, 433.4)
