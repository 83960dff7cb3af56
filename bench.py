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
LDS_READ_PEAK_TBPS = 150.0  # aggregate ds_read_b64/b128 rate with every CU streaming, MI355X_MICROARCH.md (LDS section)


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
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the companion timing with fp32 multiplies")
    ap.add_argument("--cpu-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the operator rows of configs 3-5")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)

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

    def timed(nwarm, nsteps):
        for _ in range(nwarm):
            step()
        fence()
        K.profile_begin()          # HIP events around every conv3x3 launch, on the launch stream
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o = step()
        fence()
        dt = time.perf_counter() - t0
        prof = K.profile_end()
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return o, dt, prof

    out, elapsed, conv_prof = timed(args.warmup, args.steps)
    assert torch.isfinite(out).all()
    # companion leg in the reference's arithmetic type: fp32 multiplies on the fp32 matrix cores (the ONE precision switch
    # of the denoiser, models/drunet.py); fewer steps, its own ms_per_step, printed next to the headline
    fp32_leg = None
    if not args.no_fp32_leg:
        denoiser.conv_precision = "fp32"
        n32 = max(2, args.steps // 4)
        out32, el32, prof32 = timed(1, n32)
        denoiser.conv_precision = "bf16split"
        fp32_leg = {"value_fp32": round(args.batch * n32 / el32, 4), "ms_per_step_fp32": round(el32 / n32 * 1e3, 2),
                    "steps_fp32": n32, "dtype_fp32": "f32 (v_mfma_f32_32x32x2_f32, Winograd F(2x2,3x3) for the ResBlock convs)",
                    "rel_diff_bf16split_vs_fp32": float(f"{float((out - out32).norm() / out32.norm()):.3e}")}

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
            row[k] = round(v / t, 3 if k.startswith("TFLOP") else 1) if k.endswith("_per_s") else v
        if "lds_model_TB_per_s" in row:
            row["frac_of_lds_model"] = round(row["lds_model_TB_per_s"] / LDS_READ_PEAK_TBPS, 4)
            row.pop("frac_hbm_peak")    # meaningless for this operator
        ops.append(row)

    alg = (B_local * 2 * H * W + B_local * 2 * args.coils * H * W) * 4 + args.coils * H * W * 8 + 2 * H * W * 4
    op_row("MultiCoilMRI.A", "cfg2", B_local, lambda: physics.A(x_true), alg)
    op_row("MultiCoilMRI.A_adjoint", "cfg2", B_local, lambda: physics.A_adjoint(y), alg)
    op_row("MultiCoilMRI.A_adjoint_A", "cfg2", B_local, lambda: physics.A_adjoint_A(x_true), 2 * B_local * 2 * H * W * 4
           + args.coils * H * W * 8 + 2 * H * W * 4)
    if world == 1 and not args.no_other_configs:
        other_config_ops(dinv, device, op_row, lambda: ops[-1])

    if rank == 0:
        slices_per_s = args.batch * args.steps / elapsed
        # dominant kernel = the one with the largest share of the timed region
        kname, kp = max(conv_prof.items(), key=lambda kv: kv[1]["ms"]) if conv_prof else ("none", None)
        bf16 = "split" in kname or "bf16" in kname   # bf16-split convolution: priced against the bf16 matrix peak
        peak = MFMA_BF16_PEAK if bf16 else MFMA_F32_PEAK
        # roofline.achieved = ALGORITHMIC flops (direct 3x3 convolution: 2*9*Cin*Cout*B*H*W per launch) / HIP-event
        # time of the launches; the flops the kernel EXECUTES on the matrix pipe (three bf16 products per multiply; 16/36 of
        # the direct count per 2x2 tile for fp32 Winograd) are reported next to it as `executed`
        achieved = kp["direct_flops"] / (kp["ms"] * 1e-3) if kp else 0.0
        executed = kp["mfma_flops"] / (kp["ms"] * 1e-3) if kp else 0.0
        all_ms = sum(v["ms"] for v in conv_prof.values())
        all_direct = sum(v["direct_flops"] for v in conv_prof.values())
        dtype = ("f32 in/out; ResBlock convs = Winograd F(2,3) along rows, every multiply as bf16 x2 exact operand split (3 products), "
                 "f32 accumulate") if "wsplit" in kname else "f32 in/out; ResBlock convs = bf16 x2 exact operand split (3 products), f32 accumulate"
        # HBM bytes per launch of the dominant kernel: measured by separate rocprofv3 --pmc passes (TCC_EA0_RDREQ x 64 B x 2
        # [gfx950 wide-load correction] + TCC_EA0_WRREQ x 64 B, averaged over the launches of one DRUNet call) and
        # recorded, with the commit and configuration they were taken at, in profiles/pmc_traffic.json
        traffic = None
        pmc = {}
        try:
            pmc = json.load(open(PMC_TRAFFIC_FILE))
            rec = pmc.get(kname)
            if rec and rec.get("config") == {"batch": B_local, "height": H, "width": W}:
                traffic = rec["bytes_per_launch"]
        except (OSError, ValueError):
            pass
        for row in ops:     # measured HBM bytes of the operator rows (same counter method), where a pass was recorded
            rec = pmc.get("op:" + row["op"] + "@" + row["config"])
            if rec and rec.get("batch") == row["batch"]:
                row["pmc_MB"] = round(rec["bytes_per_call"] / 1e6, 1)
                row["pmc_over_alg"] = round(rec["bytes_per_call"] / 1e6 / max(row["alg_MB"], 1e-9), 2)
        res = {
            "metric": "PnP-PGD slices/sec (50 iters), 2D MRI 8-coil 320x320, 4x radial mask, DRUNet",
            "value": round(slices_per_s, 4), "unit": "slices/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 2D MRI 8-coil 320x320, 4x radial mask (80 spokes), PnP-PGD 50 it + "
                                   "DRUNet(2->2, random init), global batch %d" % args.batch,
                       "global_batch": args.batch, "per_gpu_batch": B_local, "iters": args.iters,
                       "parallelism": f"dp{world}", "collective": "all_gather(reconstruction)" if world > 1 else "none",
                       "conv_precision": "bf16split", "loop_graph": bool(getattr(model.fixed_point, "use_graph", False))},
            "roofline": {"bound": "mfma", "kernel": kname + (" (DRUNet 3x3 conv as Winograd F(2,3) along rows, v_mfma_f32_32x32x16_bf16, split operands)"
                                                              if "wsplit" in kname else
                                                              " (DRUNet 3x3 conv, v_mfma_f32_32x32x16_bf16, split operands)" if bf16
                                                              else " (DRUNet 3x3 conv, v_mfma_f32_32x32x2_f32)"),
                         "achieved": round(achieved / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "flops": "algorithmic: direct 3x3 convolution, 2*9*Cin*Cout*B*H*W per launch",
                         "executed": round(executed / 1e12, 2), "frac_executed": round(executed / peak, 4),
                         "launches": kp["launches"] if kp else 0,
                         "avg_launch_ms": round(kp["ms"] / max(kp["launches"], 1), 4) if kp else 0.0,
                         "share_of_step": round(kp["ms"] * 1e-3 / elapsed, 4) if kp else 0.0,
                         "conv3x3_direct_equiv_TFLOPs": round(all_direct / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                         "kernels": {k: {"ms": round(v["ms"], 2), "launches": v["launches"]} for k, v in conv_prof.items()}},
            "operators": ops,
        }
        if fp32_leg:
            res.update(fp32_leg)
        if not args.no_cpu_baseline and world == 1:   # CPU baseline: rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(denoiser, maps, mask, H, W, args.coils, args.iters, y_cpu=y.cpu(), x_gpu=out.cpu())
            res["parity_rel_err_50it"] = res["cpu_baseline"].pop("parity_rel_err_max")
            res["parity_slices"] = res["cpu_baseline"].pop("parity_slices")
        print(json.dumps(res))
    ctx.__exit__(None, None, None)


def drunet3d_forward_flops(den, vol):
    """2 * prod(kernel) * Cin * Cout * output voxels, summed over the conv layers of a DRUNet(dim=3) forward"""
    D, H, W = vol
    total = 0.0
    nc, nb = den.nc, den.nb
    lvl_vox = [D * H * W / 8 ** i for i in range(4)]
    cin = den.in_channels + 1
    total += 2 * 27 * cin * nc[0] * lvl_vox[0]                              # head
    for i in range(3):
        total += nb * 2 * (2 * 27 * nc[i] * nc[i] * lvl_vox[i]) * 2         # down-path and up-path ResBlocks of level i
        total += 2 * 8 * nc[i] * nc[i + 1] * lvl_vox[i + 1] * 2             # strided down conv + transposed up conv
    total += nb * 2 * (2 * 27 * nc[3] * nc[3] * lvl_vox[3])                 # body
    total += 2 * 27 * nc[0] * den.out_channels * lvl_vox[0]                 # tail
    return total


def other_config_ops(dinv, device, op_row, ops_last):
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
    # Radon is a gather-rate problem, not an HBM one: every bilinear sample reads 4 taps x 4 B per image from
    # the LDS window, so the model is the chip's aggregate ds_read rate (~150 TB/s for b64/b128 reads, MI355X_MICROARCH.md)
    op_row("Tomography.A", "cfg3", B, lambda: phys.A(x), alg, n=5, Gsamples_per_s=smp / 1e9, lds_model_TB_per_s=smp * 16 / 1e12)
    op_row("Tomography.A_adjoint", "cfg3", B, lambda: phys.A_adjoint(y), alg, n=5, Gsamples_per_s=smp / 1e9,
           lds_model_TB_per_s=smp * 16 / 1e12)
    op_row("Tomography.fbp", "cfg3", B, lambda: phys.A_dagger(y, fbp=True), alg + 2 * B * G * A * 4, n=5)
    del phys, y
    # the same geometry with fan-beam rays (first-generation gather kernels: stated, not tuned)
    fphys = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=False, fan_beam=True, device=device)
    fy = fphys.A(x)
    fsmp = float(B) * fy.shape[2] * G * A       # detector pixels x march steps x angles
    falg = B * (W * W + fy.shape[2] * A) * 4
    op_row("Tomography(fan_beam).A", "cfg3-geometry", B, lambda: fphys.A(x), falg, n=3, Gsamples_per_s=fsmp / 1e9)
    op_row("Tomography(fan_beam).A_adjoint", "cfg3-geometry", B, lambda: fphys.A_adjoint(fy), falg, n=3, Gsamples_per_s=fsmp / 1e9)
    del fphys, fy, x
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

    # algorithmic flops of one training step: forward convolutions of the 3-D DRUNet (2 k^3 Cin Cout voxels per layer) x 3
    # (forward + data gradient + weight gradient) x 10 iterations x B volumes
    fwd = drunet3d_forward_flops(den, vol)
    gfl = 3.0 * fwd * 10 * B / 1e9
    op_row("unfolded PGD x10 + DRUNet3D(16..128, nb=1): training step", "cfg4", B, train_step, 0, n=2,
           volumes_per_s=float(B), alg_GFLOP=round(gfl, 1), TFLOP_per_s=gfl / 1e3,
           denoiser_backend="hip (models/drunet3d.py)")
    r = ops_last()
    r["frac_bf16_mfma_peak"] = round(r["TFLOP_per_s"] * 1e12 / MFMA_BF16_PEAK, 5)
    r["frac_fp32_mfma_peak"] = round(r["TFLOP_per_s"] * 1e12 / MFMA_F32_PEAK, 4)
    for k in ("alg_MB", "GBps", "frac_hbm_peak"):
        r.pop(k, None)
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


def _pgd_cpu(sd, maps, mask, y, iters):
    from oracle import drunet_cpu as OD
    from oracle import optim_cpu as OO
    from oracle import physics_cpu as OP

    A = lambda v: OP.multicoil_A(v, maps, mask)
    AT = lambda v: OP.multicoil_AT(v, maps, mask)
    with torch.no_grad():
        return OO.pnp_pgd(y, A, AT, lambda u, s: OD.drunet(sd, u, s), max_iter=iters)


def cpu_worker(path):
    """one process of the whole-box CPU run: a few PGD iterations of ONE slice on `threads` cores; reports seconds per iteration"""
    job = torch.load(path)
    common = torch.load(job["common"])
    torch.set_num_threads(job["threads"])
    _pgd_cpu(common["sd"], common["maps"], common["mask"], job["y"], 1)      # warm-up (thread pools, oneDNN primitives)
    t0 = time.perf_counter()
    _pgd_cpu(common["sd"], common["maps"], common["mask"], job["y"], job["iters"])
    torch.save({"s_per_it": (time.perf_counter() - t0) / job["iters"]}, path + ".out")


def usable_cores():
    """cores this process may really use: the scheduler affinity and the cgroup CPU quota, not the machine's core count"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(denoiser, maps, mask, H, W, coils, iters, y_cpu, x_gpu, parity_slices=4, box_iters=3, box_timeout=90.0):
    """The oracle ("port": same ATen CPU call sequence as the reference) timed on this box's host cores on a bounded
    sample of the same workload, two ways:
      * one process on its best thread count (calibrated: oneDNN / MKL at batch 1 degrade when oversubscribed): median of
        three timed samples - slice 0 for all `iters` iterations, then twice for iters/5 iterations (per-iteration cost is
        constant) - this is `value` / `cores`;
      * the whole box (`whole_box`): P = usable cores / threads processes at once, each running `box_iters` iterations of a
        different slice; aggregate slices/s = P / (slowest per-iteration time x iters).  Bounded: the group is killed after
        `box_timeout` seconds (an oversubscribed or quota-limited box then reports `timed_out`).
    Slices 0 .. parity_slices-1 are reconstructed on the CPU for all `iters` iterations: the largest relative distance to the
    GPU reconstruction of the same measurement is the headline-configuration parity figure (`parity_rel_err_50it`,
    north_star bound 1e-4)."""
    import statistics
    import subprocess
    import tempfile

    from oracle import drunet_cpu as OD

    cores, usable = os.cpu_count() or 1, usable_cores()
    sd = {k: v.detach().cpu() for k, v in denoiser.state_dict().items()}
    calib = {}
    with torch.no_grad():
        probe = torch.rand(1, 2, H, W, generator=torch.Generator().manual_seed(7))
        best = None
        for nt in [c for c in (8, 16, 32, 64, 128) if c <= usable] or [usable]:
            torch.set_num_threads(nt)
            OD.drunet(sd, probe.new_zeros(1, 2, H, W), 0.05)
            t0 = time.perf_counter()
            OD.drunet(sd, probe, 0.05)
            dt = time.perf_counter() - t0
            calib[str(nt)] = round(dt, 3)
            if best is None or dt < best[1]:
                best = (nt, dt)
            if dt > 3 * best[1]:
                break
    threads = best[0]
    torch.set_num_threads(threads)
    _pgd_cpu(sd, maps, mask, y_cpu[:1], 1)   # warm-up
    samples, recs = [], {}
    t0 = time.perf_counter()
    recs[0] = _pgd_cpu(sd, maps, mask, y_cpu[:1], iters)
    samples.append((time.perf_counter() - t0) / iters)
    short = max(1, iters // 5)
    for _ in range(2):
        t0 = time.perf_counter()
        _pgd_cpu(sd, maps, mask, y_cpu[:1], short)
        samples.append((time.perf_counter() - t0) / short)
    per_it = statistics.median(samples)
    for i in range(1, min(parity_slices, y_cpu.shape[0])):        # more slices for the parity figure (not timed)
        recs[i] = _pgd_cpu(sd, maps, mask, y_cpu[i:i + 1], iters)
    # ---- whole box: P concurrent single-slice processes, a few iterations each
    nproc = max(1, min(usable // threads, y_cpu.shape[0], 32))
    whole = {"processes": nproc, "threads_each": threads, "usable_cores": usable}
    if nproc > 1:
        with tempfile.TemporaryDirectory() as tmp:
            paths = []
            common = os.path.join(tmp, "common.pt")
            torch.save({"sd": sd, "maps": maps, "mask": mask}, common)
            for i in range(nproc):
                pth = os.path.join(tmp, f"job{i}.pt")
                torch.save({"common": common, "y": y_cpu[i:i + 1].clone(), "iters": box_iters, "threads": threads}, pth)
                paths.append(pth)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="",
                       OMP_WAIT_POLICY="PASSIVE")
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", pth], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for pth in paths]
            timed_out = False
            for pr in procs:
                left = box_timeout - (time.perf_counter() - t0)
                try:
                    pr.wait(timeout=max(left, 0.1))
                except subprocess.TimeoutExpired:
                    timed_out = True
            if timed_out:
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()
            outs = [torch.load(pth + ".out") for pth in paths if os.path.exists(pth + ".out")]
            whole["wall_s_incl_startup"] = round(time.perf_counter() - t0, 1)
            if len(outs) == nproc and not timed_out:
                slowest = max(o["s_per_it"] for o in outs)
                whole.update(value=round(nproc / (slowest * iters), 5), unit="slices/s",
                             slowest_s_per_iteration=round(slowest, 4), sample=f"{box_iters} iterations per process")
            else:
                whole.update(timed_out=True, finished=len(outs))
    else:
        whole.update(value=round(1.0 / (per_it * iters), 5), unit="slices/s", sample="one process fills the usable cores")
    errs = {i: float((x_gpu[i:i + 1].double() - xk.double()).norm() / xk.double().norm()) for i, xk in recs.items()}
    return {"value": round(1.0 / (per_it * iters), 5), "unit": "slices/s", "cores": threads, "host_cores": cores,
            "usable_cores": usable, "kind": "port", "threads_calibration_s_per_denoiser_call": calib,
            "samples_s_per_iteration": [round(v, 4) for v in samples],
            "sample": f"slice 0 of the batch: median of 3 timed samples ({iters} it once, {short} it twice), "
                      f"{per_it * iters:.1f} s per slice",
            "whole_box": whole,
            "parity_rel_err_max": float(f"{max(errs.values()):.3e}"), "parity_slices": len(errs)}


if __name__ == "__main__":
    main()
