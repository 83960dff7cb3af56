#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): PnP-PGD slices/sec (50 iterations) on 2-D 8-coil
320x320 MRI with a 4x radial mask and the DRUNet(2->2) denoiser, global batch 32, plus
A / A_adjoint GB/s against the HBM roofline.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full 50-iteration PnP-PGD reconstruction of the global batch (inputs already
resident in HBM).  Multi-GPU: the batch is sharded in contiguous slabs, one process per GPU,
reconstructions combined by one RCCL all-gather per step (inside the timed region);
total work is fixed as N grows ("scaling": "strong", the north_star's >=6x@8 target).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # {kernel: {bytes_per_launch, commit, config}}
MFMA_BF16_PEAK = 2500e12     # FLOP/s dense bf16 matrix (v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12   # FLOP/s dense fp32 matrix (v_mfma_f32_32x32x2_f32)


def make_problem(dinv, B_local, offset, H, W, coils, device):
    """Synthetic cfg-2 inputs, seeded per global slice index so that the global
    batch is identical for every N."""
    g = torch.Generator().manual_seed(0)
    maps = torch.randn(1, coils, H, W, dtype=torch.complex64, generator=g)
    maps = maps / maps.abs().pow(2).sum(dim=1, keepdim=True).sqrt()
    mask = dinv.utils.radial_mask(H, W, 80)
    xs, ns = [], []
    for i in range(offset, offset + B_local):
        gi = torch.Generator().manual_seed(1000 + i)
        xs.append(torch.rand(1, 2, H, W, generator=gi))
        ns.append(torch.randn(1, 2, coils, H, W, generator=gi))
    x = torch.cat(xs).to(device)
    noise = torch.cat(ns).to(device)
    physics = dinv.physics.MultiCoilMRI(mask=mask, coil_maps=maps, img_size=(2, H, W), device=device)
    y = physics.A(x) + 0.01 * noise * physics.mask[:, :, None]
    return physics, x, y, maps, mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="global batch (slices)")
    ap.add_argument("--iters", type=int, default=50, help="PGD iterations per reconstruction")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--coils", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the operator rows of configs 3-5")
    args = ap.parse_args()

    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext
    from deepinv_amd.hip import drunet as K

    # rank / device / rendez-vous conventions, slab partition and the gather are those of deepinv_amd.distributed
    # (the same code the world-size-2 gloo tests run on CPU tensors, tests/test_distributed_cpu.py)
    ctx = BatchParallelContext(backend="nccl")   # "nccl" is RCCL on ROCm
    world, rank = ctx.world_size, ctx.rank
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    ctx.__enter__()
    device = ctx.device
    if device.type != "cuda":
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")

    H = W = args.size
    if args.batch % world:
        raise SystemExit("global batch must be divisible by the number of GPUs")
    slab = ctx.slab(args.batch)
    B_local = slab.stop - slab.start
    physics, x_true, y, maps, mask = make_problem(dinv, B_local, slab.start, H, W, args.coils, device)

    torch.manual_seed(0)
    denoiser = dinv.models.DRUNet(2, 2, pretrained=None).to(device).eval()
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(denoiser), stepsize=1.0, g_param=0.05,
                           max_iter=args.iters, early_stop=False)
    def step():
        return ctx.all_gather_batch(model(y, physics), args.batch)     # one RCCL all-gather per step (identity at N = 1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    K.profile_begin()          # HIP events around every conv3x3 launch, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    conv_prof = K.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    # ---- operator GB/s (outside the timed region)
    ops = []
    def time_op(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    def op_row(name, cfg, batch, fn, alg, n=20, **extra):
        t = time_op(fn, n)
        row = {"op": name, "config": cfg, "batch": batch, "ms": round(t * 1e3, 4), "alg_MB": round(alg / 1e6, 2),
               "GBps": round(alg / t / 1e9, 1), "frac_hbm_peak": round(alg / t / HBM_PEAK, 4)}
        for k, v in extra.items():      # *_per_s extras are work counts: divide by the measured time
            row[k] = round(v / t, 1) if k.endswith("_per_s") else v
        ops.append(row)

    alg = (B_local * 2 * H * W + B_local * 2 * args.coils * H * W) * 4 + args.coils * H * W * 8 + 2 * H * W * 4
    op_row("MultiCoilMRI.A", "cfg2", B_local, lambda: physics.A(x_true), alg)
    op_row("MultiCoilMRI.A_adjoint", "cfg2", B_local, lambda: physics.A_adjoint(y), alg)
    op_row("MultiCoilMRI.A_adjoint_A", "cfg2", B_local, lambda: physics.A_adjoint_A(x_true), 2 * B_local * 2 * H * W * 4
           + args.coils * H * W * 8 + 2 * H * W * 4)
    if world == 1 and not args.no_other_configs:
        other_config_ops(dinv, device, op_row)

    if rank == 0:
        slices_per_s = args.batch * args.steps / elapsed
        # dominant kernel = the one with the largest share of the timed region
        kname, kp = max(conv_prof.items(), key=lambda kv: kv[1]["ms"]) if conv_prof else ("none", None)
        achieved = kp["mfma_flops"] / (kp["ms"] * 1e-3) if kp else 0.0
        bf16 = "bf16" in kname   # bf16-split convolution: priced against the bf16 matrix peak
        peak = MFMA_BF16_PEAK if bf16 else MFMA_F32_PEAK
        all_ms = sum(v["ms"] for v in conv_prof.values())
        all_direct = sum(v["direct_flops"] for v in conv_prof.values())
        from deepinv_amd.models.drunet import _resblock_conv
        conv_mode = ("bf16x" + os.environ["DINV_CONV_BF16X3"] if os.environ.get("DINV_CONV_BF16X3") in ("2", "3")
                     else _resblock_conv())
        dtype = {"bf16s": "f32 in/out; ResBlock convs = bf16 x2 exact operand split (3 products), f32 accumulate",
                 "bf16x2": "f32 in/out; ResBlock convs = bf16 x2 exact operand split (3 products), f32 accumulate",
                 "bf16x3": "f32 in/out; ResBlock convs = bf16 x3 exact operand split (6 products), f32 accumulate"}.get(conv_mode, "f32")
        # HBM bytes per launch of the dominant kernel: measured by separate rocprofv3 --pmc passes (TCC_EA0_RDREQ x 64 B x 2
        # [gfx950 wide-load correction] + TCC_EA0_WRREQ x 64 B, averaged over the launches of one DRUNet call) and
        # recorded, with the commit and configuration they were taken at, in profiles/pmc_traffic.json
        traffic = None
        try:
            rec = json.load(open(PMC_TRAFFIC_FILE)).get(kname)
            if rec and rec.get("config") == {"batch": B_local, "height": H, "width": W}:
                traffic = rec["bytes_per_launch"]
        except (OSError, ValueError):
            pass
        res = {
            "metric": "PnP-PGD slices/sec (50 iters), 2D MRI 8-coil 320x320, 4x radial mask, DRUNet",
            "value": round(slices_per_s, 4), "unit": "slices/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 2D MRI 8-coil 320x320, 4x radial mask (80 spokes), PnP-PGD 50 it + "
                                   "DRUNet(2->2, random init), global batch %d" % args.batch,
                       "global_batch": args.batch, "per_gpu_batch": B_local, "iters": args.iters,
                       "parallelism": f"dp{world}", "collective": "all_gather(reconstruction)" if world > 1 else "none",
                       "resblock_conv": conv_mode, "loop_graph": os.environ.get("DINV_LOOP_GRAPH", "0") == "1"},
            # achieved = flops EXECUTED on the matrix pipe by the dominant kernel / its HIP-event time: 3 products per
            # multiply for the bf16 split kernel, 16/36 of the direct count for fp32 Winograd F(2x2,3x3); the effective
            # direct-convolution rate of all 3x3 convs is reported next to it
            "roofline": {"bound": "mfma", "kernel": kname + (" (DRUNet 3x3 conv, v_mfma_f32_32x32x16_bf16, split operands)" if bf16
                                                              else " (DRUNet 3x3 conv, v_mfma_f32_32x32x2_f32)"),
                         "achieved": round(achieved / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "launches": kp["launches"] if kp else 0,
                         "avg_launch_ms": round(kp["ms"] / max(kp["launches"], 1), 4) if kp else 0.0,
                         "share_of_step": round(kp["ms"] * 1e-3 / elapsed, 4) if kp else 0.0,
                         "conv3x3_direct_equiv_TFLOPs": round(all_direct / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                         "kernels": {k: {"ms": round(v["ms"], 2), "launches": v["launches"]} for k, v in conv_prof.items()}},
            "operators": ops,
        }
        if not args.no_cpu_baseline and world == 1:   # CPU baseline: rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(denoiser, maps, mask, H, W, args.coils, args.iters, y0=y[:1].cpu(),
                                               x_gpu0=out[:1].cpu())
            res["parity_rel_err_50it"] = res["cpu_baseline"].pop("parity_rel_err")
        print(json.dumps(res))
    ctx.__exit__(None, None, None)


def other_config_ops(dinv, device, op_row):
    """operator rows of BASELINE configs[2..4] at their per-GPU shard shapes (SURVEY 8d byte counts);
    the Radon rows also carry bilinear samples/s (that operator is gather-rate bound, not HBM bound)"""
    g = torch.Generator().manual_seed(0)
    # cfg3: Tomography 512x512, 720 angles, 8 images per GPU
    B, W, A = 8, 512, 720
    phys = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=True, device=device)
    x = torch.rand(B, 1, W, W, generator=g).to(device)
    y = phys.A(x)
    G = y.shape[2]
    alg = B * (W * W + G * A) * 4
    smp = float(B) * G * G * A
    op_row("Tomography.A", "cfg3", B, lambda: phys.A(x), alg, n=5, Gsamples_per_s=smp / 1e9)
    op_row("Tomography.A_adjoint", "cfg3", B, lambda: phys.A_adjoint(y), alg, n=5, Gsamples_per_s=smp / 1e9)
    op_row("Tomography.fbp", "cfg3", B, lambda: phys.A_dagger(y, fbp=True), alg + 2 * B * G * A * 4, n=5)
    del phys, x, y
    # cfg4: 3-D MultiCoilMRI 12 coils 16x256x256, 2 volumes per GPU
    B, coils, vol = 2, 12, (16, 256, 256)
    nv = vol[0] * vol[1] * vol[2]
    x = torch.rand(B, 2, *vol, generator=g).to(device)
    maps = (torch.randn(1, coils, *vol, dtype=torch.complex64, generator=g) / coils ** 0.5).to(device)
    mask = torch.zeros(*vol)
    mask[..., ::4] = 1
    mask[..., 118:138] = 1
    phys = dinv.physics.MultiCoilMRI(mask=mask.to(device), coil_maps=maps, img_size=(2, *vol), three_d=True, device=device)
    y = phys.A(x)
    alg = B * 2 * nv * 4 + B * 2 * coils * nv * 4 + coils * nv * 8 + 2 * nv * 4
    op_row("MultiCoilMRI3D.A", "cfg4", B, lambda: phys.A(x), alg)
    op_row("MultiCoilMRI3D.A_adjoint", "cfg4", B, lambda: phys.A_adjoint(y), alg)
    # cfg4's loop: one training step (forward + backward) of 10-iteration unfolded PGD with the 3-D DRUNet prior
    torch.manual_seed(0)
    den = dinv.models.DRUNet(2, 2, nc=(16, 32, 64, 128), nb=1, pretrained=None, dim=3).to(device)
    net = dinv.unfolded.unfolded_builder("PGD", data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den),
                                         params_algo={"stepsize": 1.0, "g_param": 0.05, "lambda": 1.0}, max_iter=10,
                                         trainable_params=["stepsize", "g_param"], device=device).to(device)

    def train_step():
        net.zero_grad()
        (net(y, phys) - x).pow(2).mean().backward()

    op_row("unfolded PGD x10 + DRUNet3D(16..128, nb=1): training step", "cfg4", B, train_step, 0, n=2,
           volumes_per_s=float(B), denoiser_backend="hip (models/drunet3d.py)")
    del phys, x, y, maps, den, net
    # cfg5: Downsampling x4 (bicubic, circular) on 3x256x256, 16 images per GPU
    B, img = 16, (3, 256, 256)
    phys = dinv.physics.Downsampling(img_size=img, filter="bicubic", factor=4, padding="circular", device=device)
    x = torch.rand(B, *img, generator=g).to(device)
    y = phys.A(x)
    z = torch.rand(B, *img, generator=g).to(device)
    alg = B * 3 * (256 * 256 + 64 * 64) * 4
    op_row("Downsampling.A", "cfg5", B, lambda: phys.A(x), alg)
    op_row("Downsampling.A_adjoint", "cfg5", B, lambda: phys.A_adjoint(y), alg)
    op_row("Downsampling.prox_l2", "cfg5", B, lambda: phys.prox_l2(z, y, 0.7), 2 * B * 3 * 256 * 256 * 4 + B * 3 * 64 * 64 * 4)


def cpu_baseline(denoiser, maps, mask, H, W, coils, iters, y0=None, x_gpu0=None):
    """The oracle ("port": same ATen CPU call sequence as the reference) timed on this box's host cores on a bounded
    sample: the FULL `iters`-iteration PnP-PGD reconstruction of slice 0 of the GPU batch (per-slice CPU cost is batch
    independent).  The CPU result is kept: its relative distance to the GPU reconstruction of the same measurement is
    the headline-configuration parity figure (`parity_rel_err_50it`, north_star bound 1e-4)."""
    from oracle import drunet_cpu as OD
    from oracle import optim_cpu as OO
    from oracle import physics_cpu as OP

    cores = os.cpu_count() or 1
    sd = {k: v.detach().cpu() for k, v in denoiser.state_dict().items()}
    A = lambda v: OP.multicoil_A(v, maps, mask)
    AT = lambda v: OP.multicoil_AT(v, maps, mask)
    if y0 is None:
        y0 = A(torch.rand(1, 2, H, W, generator=torch.Generator().manual_seed(1000)))
    den = lambda u, s: OD.drunet(sd, u, s)
    calib = {}
    with torch.no_grad():
        # give the CPU its best shot: oneDNN/MKL at batch 1 degrade badly when oversubscribed (256 threads on this
        # box: 52 s / iteration), so calibrate the thread count on one denoiser call each
        probe = torch.rand(1, 2, H, W, generator=torch.Generator().manual_seed(7))
        best = None
        for nt in [c for c in (8, 16, 32, 64, 128) if c <= cores] or [cores]:
            torch.set_num_threads(nt)
            den(probe.new_zeros(1, 2, H, W), 0.05)
            t0 = time.perf_counter()
            den(probe, 0.05)
            dt = time.perf_counter() - t0
            calib[str(nt)] = round(dt, 3)
            if best is None or dt < best[1]:
                best = (nt, dt)
            if dt > 3 * best[1]:
                break
        torch.set_num_threads(best[0])
        OO.pnp_pgd(y0, A, AT, den, max_iter=1)  # warm-up
        t0 = time.perf_counter()
        xk = OO.pnp_pgd(y0, A, AT, den, max_iter=iters)
        dt = time.perf_counter() - t0
    err = None
    if x_gpu0 is not None:
        err = float((x_gpu0.double() - xk.double()).norm() / xk.double().norm())
    return {"value": round(1.0 / dt, 5), "unit": "slices/s", "cores": torch.get_num_threads(), "host_cores": cores,
            "kind": "port", "threads_calibration_s_per_denoiser_call": calib,
            "sample": f"slice 0 of the batch, all {iters} PGD iterations ({dt:.1f} s)",
            "parity_rel_err": None if err is None else float(f"{err:.3e}")}


if __name__ == "__main__":
    main()
