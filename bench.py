#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): PnP-PGD slices/sec (50 iterations) on 2-D 8-coil
320x320 MRI with a 4x radial mask and the DRUNet(2->2) denoiser, global batch 32, plus
A / A_adjoint GB/s against the HBM roofline.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full 50-iteration PnP-PGD reconstruction of the global batch (inputs already
resident in HBM).  The headline leg (`value`) runs the denoiser with fp32 multiplies on the fp32 matrix
cores (the reference's arithmetic type); the bf16-split throughput setting is a companion leg
(`value_bf16split`).  Multi-GPU: the batch is sharded in contiguous slabs, one process per GPU,
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

# (deepinv_amd sets the same default at import; bench.py touches the device before it imports the package)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # hardware queues for the streams of the batch lanes beside RCCL's: deepinv_amd/__init__.py

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # {kernel: {bytes_per_launch, commit, config}}
MFMA_BF16_PEAK = 2500e12     # FLOP/s dense bf16 matrix (v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12   # FLOP/s dense fp32 matrix (v_mfma_f32_32x32x2_f32)
LDS_READ_PEAK_TBPS = 150.0  # aggregate ds_read_b64/b128 rate with every CU streaming, MI355X_MICROARCH.md (LDS section)
VALU_PEAK_TINSTR = 78.6     # 1e12 lane-instructions/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (the fp32 vector peak / 2 flops per FMA)


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
    ap.add_argument("--no-split-leg", "--no-fp32-leg", dest="no_split_leg", action="store_true",
                    help="skip the companion timing of the bf16-split throughput setting")
    ap.add_argument("--cpu-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the operator rows of configs 3-5")
    ap.add_argument("--lanes", type=int, default=None,
                    help="DRUNet.batch_lanes: the per-GPU batch cut into this many parts, each run through the network on its own HIP "
                         "stream (models/drunet.py); default: the model's default (auto: 2 up to 16 slices of 320 x 320, else 1)")
    ap.add_argument("--no-tail-split", action="store_true", help="diagnostic: F(4x4) launches without the channel split of their last round")
    ap.add_argument("--bf16x3", action="store_true", help="diagnostic: the F(4x4) launches in their bf16 x 3 form (hip/drunet.py: FP32_WINOGRAD4_BF16X3)")
    ap.add_argument("--as-multi", action="store_true",
                    help="pre-flight of the multi-GPU code path on ONE rank: create the RCCL process group, replay the iteration as a HIP "
                         "graph, run agree_on_graph / barrier / all-gather / max-over-ranks exactly as an N > 1 run does (world size 1 "
                         "communicator), print the multi-GPU form of the JSON line (tests/test_loops_gpu.py)")
    ap.add_argument("--loop-graph", action="store_true",
                    help="replay the PGD iteration as a HIP graph also on one GPU (optim/fixed_point.py: use_graph; always on for "
                         "--gpus > 1); the per-launch HIP events of `roofline` then come from one eager step after the timed region")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)

    import deepinv_amd as dinv
    from deepinv_amd.distributed import BatchParallelContext
    from deepinv_amd.hip import drunet as K

    # rank / device / rendez-vous conventions, slab partition and the gather are those of deepinv_amd.distributed
    # (the same code the world-size-2 gloo tests run on CPU tensors, tests/test_distributed_cpu.py)
    ctx = BatchParallelContext(backend="nccl", init_always=args.as_multi)   # "nccl" is RCCL on ROCm
    world, rank = ctx.world_size, ctx.rank
    multi = world > 1 or args.as_multi      # the code path of an N > 1 run (every `multi` below was `world > 1`)
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
    if args.lanes is not None:
        denoiser.batch_lanes = args.lanes
    if args.no_tail_split:
        K.WINOGRAD4_TAIL_SPLIT = False
    if args.bf16x3:
        K.FP32_WINOGRAD4_BF16X3 = True
    lanes_asked = denoiser._lanes(torch.empty(B_local, 2, H, W, device="meta"))
    model = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(denoiser), stepsize=1.0, g_param=0.05,
                           max_iter=args.iters, early_stop=False)
    # The PGD iteration as a replayed HIP graph (optim/fixed_point.py: use_graph - one host call per iteration instead of ~80 launches):
    # +0.4 % at 32 slices per GPU, +6 % at 4, where the host-side launch gaps are a visible share of the iteration.  On for the
    # multi-GPU runs (small per-GPU batches) and on request; the one-GPU line keeps the eager loop, whose per-launch HIP events over
    # the timed region are what `roofline` is computed from.  A capture failure falls back to the eager loop and says so.
    graph_state = {"on": bool(args.loop_graph or multi), "error": None}
    model.fixed_point.use_graph = graph_state["on"]

    def reconstruct():
        """the loop on this rank's slab; a graph-capture failure switches this rank to the eager loop (no collective in here)"""
        if graph_state["on"] and not graph_state.get("proven"):
            try:
                rec = model(y, physics)
                graph_state["proven"] = True
                return rec
            except Exception as e:      # noqa: BLE001 - any capture problem: the eager loop instead of no result
                graph_state.update(on=False, error=f"{type(e).__name__}: {e}"[:300])
                model.fixed_point.use_graph = False
                torch.cuda.synchronize()
        return model(y, physics)

    def step():
        return ctx.all_gather_batch(reconstruct(), args.batch)     # one RCCL all-gather per step (identity at N = 1)

    def agree_on_graph():
        """every rank must run the same number of collectives: if the capture failed anywhere, all ranks run the eager loop"""
        if multi:
            ok = torch.tensor([1 if graph_state["on"] else 0], device=device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if graph_state["on"] and int(ok.item()) == 0:
                graph_state.update(on=False, error=graph_state["error"] or "capture failed on another rank")
                model.fixed_point.use_graph = False

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    power_samples, prof_steps = [], []

    def timed(nwarm, nsteps):
        for i in range(max(nwarm, 1 if graph_state["on"] and not graph_state.get("proven") else 0)):
            step()
            if i == 0:
                agree_on_graph()
        fence()
        # host thread: rocm-smi twice a second, no GPU work
        sampler = PowerSampler((device.index if device.index is not None else torch.cuda.current_device()) if rank == 0 else None)
        if not graph_state["on"]:
            K.profile_begin()      # HIP events around every conv3x3 launch, on the launch stream
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o = step()
        fence()
        dt = time.perf_counter() - t0
        prof = K.profile_end()
        power_samples.append(sampler.stop())
        prof_steps.append(nsteps)
        if graph_state["on"]:       # events cannot be recorded inside a graph: one eager step AFTER the timed region carries them
            model.fixed_point.use_graph = False
            K.profile_begin()
            step()
            fence()
            prof = K.profile_end()
            prof_steps[-1] = 1
            model.fixed_point.use_graph = True
        if multi:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return o, dt, prof

    # The headline leg runs the denoiser in the reference's arithmetic type: fp32 multiplies on the fp32 matrix cores
    # (conv_precision = "fp32": Winograd F(4x4,3x3) for the ResBlock convolutions).  The companion leg is the throughput
    # setting with narrower multiplies (bf16 x2 operand split, 16 significand bits per operand): fewer steps, its own
    # ms_per_step, reported under *_bf16split and never as `value`.
    denoiser.conv_precision = "fp32"
    out, elapsed, conv_prof = timed(args.warmup, args.steps)
    assert torch.isfinite(out).all()
    package = power_samples[0]
    split_leg = None
    if not args.no_split_leg:
        denoiser.conv_precision = "bf16split"
        n2 = max(2, args.steps // 4)
        out2, el2, prof2 = timed(1, n2)
        denoiser.conv_precision = "fp32"
        k2, p2 = max(prof2.items(), key=lambda kv: kv[1]["ms"])
        split_leg = {"value_bf16split": round(args.batch * n2 / el2, 4), "ms_per_step_bf16split": round(el2 / n2 * 1e3, 2),
                     "steps_bf16split": n2,
                     "dtype_bf16split": "f32 in/out; ResBlock convs = Winograd F(2,3) along rows, every multiply as bf16 x2 exact "
                                        "operand split (3 products: 16 significand bits per operand), f32 accumulate",
                     "roofline_bf16split": {"kernel": k2, "achieved": round(p2["mfma_flops"] / (p2["ms"] * 1e-3) / 1e12, 2),
                                            "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                            "frac": round(p2["mfma_flops"] / (p2["ms"] * 1e-3) / MFMA_BF16_PEAK, 4),
                                            "flops": "executed on the bf16 matrix pipe (three bf16 products per fp32 multiply)",
                                            "direct_equiv": round(p2["direct_flops"] / (p2["ms"] * 1e-3) / 1e12, 2),
                                            "avg_launch_ms": round(p2["ms"] / max(p2["launches"], 1), 4)},
                     "rel_diff_bf16split_vs_fp32": float(f"{float((out2 - out).norm() / out.norm()):.3e}")}

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
        if "valu_model_Tinstr_per_s" in row:
            row["frac_of_valu_model"] = round(row["valu_model_Tinstr_per_s"] / VALU_PEAK_TINSTR, 4)
            row.pop("frac_hbm_peak")
        ops.append(row)

    def loop_row(name, cfg, batch, fn, gflop, unit, precision):
        """one whole reconstruction loop of another BASELINE config at its per-GPU shard: wall time of the second of two runs"""
        with torch.no_grad():
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = fn()
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
        ops.append({"op": name, "config": cfg, "batch": batch, "ms": round(t * 1e3, 1), unit: round(batch / t, 3),
                    "denoiser_TFLOP_per_s": round(gflop / t / 1e3, 1), "conv_precision": precision,
                    "finite": bool(torch.isfinite(o).all())})

    alg = (B_local * 2 * H * W + B_local * 2 * args.coils * H * W) * 4 + args.coils * H * W * 8 + 2 * H * W * 4
    op_row("MultiCoilMRI.A", "cfg2", B_local, lambda: physics.A(x_true), alg)
    op_row("MultiCoilMRI.A_adjoint", "cfg2", B_local, lambda: physics.A_adjoint(y), alg)
    op_row("MultiCoilMRI.A_adjoint_A", "cfg2", B_local, lambda: physics.A_adjoint_A(x_true), 2 * B_local * 2 * H * W * 4
           + args.coils * H * W * 8 + 2 * H * W * 4)
    if not multi:
        blur_and_single_coil_rows(dinv, device, op_row)
    if not multi and not args.no_other_configs:
        other_config_ops(dinv, device, op_row, lambda: ops[-1], loop_row, ops.append)

    if rank == 0:
        slices_per_s = args.batch * args.steps / elapsed
        # dominant kernel = the one with the largest share of the timed region
        kname, kp = max(conv_prof.items(), key=lambda kv: kv[1]["ms"]) if conv_prof else ("none", None)
        peak = MFMA_F32_PEAK        # the headline leg multiplies in fp32 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
        # The roofline of the dominant kernel is the fp32 MATRIX PIPE: `achieved` = the flops the kernel EXECUTES on it per launch
        # / the HIP-event time of the launches, `frac` = achieved / peak (<= 1).  A Winograd kernel executes fewer multiplies than
        # the direct convolution it evaluates (36 instead of 144 per 4x4 output tile and channel pair for F(4x4,3x3)): the
        # ALGORITHMIC count of SURVEY 8d (2*9*Cin*Cout*B*H*W per launch) over the same time is reported beside it as
        # `direct_equiv` / `frac_direct_equiv` (that one can exceed 1 and is not a fraction of any roofline).
        # batch lanes > 1: the launches of the lanes run CONCURRENTLY (each on the compute units the other leaves idle), so the HIP-event
        # durations of the launches overlap in time; the time the kernel family occupies the chip is their sum / lanes
        from deepinv_amd import hip as HIP
        lanes = lanes_asked if (lanes_asked > 1 and HIP._LANE_STREAMS.get(HIP.lane_key(device, lanes_asked)) is not None) else 1
        kms = kp["ms"] / lanes if kp else 0.0
        achieved = kp["direct_flops"] / (kms * 1e-3) if kp else 0.0
        executed = kp["mfma_flops"] / (kms * 1e-3) if kp else 0.0
        all_ms = sum(v["ms"] for v in conv_prof.values())
        all_direct = sum(v["direct_flops"] for v in conv_prof.values())
        kdesc = {"conv3x3_wino4_kernel": "DRUNet 3x3 conv as Winograd F(4x4,3x3), v_mfma_f32_32x32x2_f32",
                 "conv3x3_wino_kernel": "DRUNet 3x3 conv as Winograd F(2x2,3x3), v_mfma_f32_32x32x2_f32",
                 "conv3x3_kernel": "DRUNet 3x3 conv, direct, v_mfma_f32_32x32x2_f32"}.get(kname, kname)
        dtype = ("f32: fp32 multiplies and accumulation on the fp32 matrix cores (v_mfma_f32_32x32x2_f32); ResBlock convs as "
                 + ("Winograd F(4x4,3x3)" if kname == "conv3x3_wino4_kernel" else "Winograd F(2x2,3x3)" if kname == "conv3x3_wino_kernel"
                    else "direct 3x3") + ", U = G g G^T formed in fp64 and rounded once; the three 2x2 stride-2 down convolutions "
                 "(2 % of the FLOPs) as an exact three-part bf16 operand split with six products and fp32 accumulation = all 24 "
                 "significand bits of both operands (layer_rel_err_vs_fp64: at the level of the fp32 kernel)")
        # HBM bytes per launch of the dominant kernel: measured by separate rocprofv3 --pmc passes (TCC_EA0_RDREQ x 64 B x 2
        # [gfx950 wide-load correction] + TCC_EA0_WRREQ x 64 B, averaged over the launches of one DRUNet call) and
        # recorded, with the commit and configuration they were taken at, in profiles/pmc_traffic.json
        # A recorded row is used only if the kernel sources it was measured on are the ones in the tree now (`sources_sha16`
        # = hash of the files that hold the kernels of that row, written by scripts/r05/merge_pmc.py): a stale row is dropped
        # and named in `pmc_stale`.
        traffic, mfma_busy = None, None
        pmc, stale = {}, []
        try:
            pmc = json.load(open(PMC_TRAFFIC_FILE))
        except (OSError, ValueError):
            pass

        def fresh(key):
            rec = pmc.get(key)
            if not rec:
                return None
            if rec.get("sources_sha16") != sources_sha16(rec.get("sources", [])):
                stale.append(key)
                return None
            return rec

        rec = fresh(kname)
        if rec and rec.get("config") == {"batch": B_local, "height": H, "width": W}:
            traffic = rec["bytes_per_launch"]
            mfma_busy = rec.get("mfma_busy")
        for row in ops:     # measured HBM bytes of the operator rows (same counter method), where a pass was recorded
            rec = fresh("op:" + row["op"] + "@" + row["config"])
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
                       "parallelism": f"dp{world}", "collective": "all_gather(reconstruction)" if multi else "none",
                       "conv_precision": "fp32", "batch_lanes": lanes, "batch_lanes_calibration": getattr(denoiser, "_lane_calibration", None), "loop_graph": graph_state["on"], "loop_graph_error": graph_state["error"],
                       **({"as_multi_preflight": True, "process_group": {"backend": dist.get_backend(), "world_size": dist.get_world_size()}}
                          if args.as_multi else {})},
            "roofline": {"bound": "mfma", "kernel": f"{kname} ({kdesc})",
                         "achieved": round(executed / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": round(executed / peak, 4), "traffic": traffic, "mfma_busy": mfma_busy,
                         "flops": "executed on the fp32 matrix pipe: 2*36*Cin*Cout*B*H*W/16 per launch for Winograd F(4x4,3x3) (a "
                                  "quarter of the direct convolution's 2*9*Cin*Cout*B*H*W = SURVEY 8d's algorithmic count, which is "
                                  "direct_equiv)",
                         "direct_equiv": round(achieved / 1e12, 2), "frac_direct_equiv": round(achieved / peak, 4),
                         "pmc_stale": stale,
                         # the package during the timed steps of this leg (rocm-smi from a host thread): DESIGN.md 3.2 - the F(4x4)
                         # kernel runs at the package power limit, the clock it sustains there is below the 2.4 GHz of `peak`
                         "package_during_timed_steps": package,
                         "launches": kp["launches"] if kp else 0,
                         "avg_launch_ms": round(kp["ms"] / max(kp["launches"], 1), 4) if kp else 0.0,
                         "share_of_step": round(kms * 1e-3 / (elapsed * prof_steps[0] / args.steps), 4) if kp else 0.0,
                         "events_from": "the timed steps" if not graph_state["on"] else
                                        "one eager step after the timed region (the timed steps replay a HIP graph: no events inside it)",
                         "conv3x3_direct_equiv_TFLOPs": round(all_direct / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                         "kernels": {k: {"ms": round(v["ms"], 2), "launches": v["launches"]} for k, v in conv_prof.items()}},
            "operators": ops,
        }
        if split_leg:
            res.update(split_leg)
        if not multi:
            res["layer_rel_err_vs_fp64"] = layer_errors(device)
        if not args.no_cpu_baseline and not multi:   # CPU baseline: rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(denoiser, maps, mask, H, W, args.coils, args.iters, y_cpu=y.cpu(), x_gpu=out.cpu())
            res["parity_rel_err_50it"] = res["cpu_baseline"].pop("parity_rel_err_max")
            res["parity_slices"] = res["cpu_baseline"].pop("parity_slices")
            if split_leg:
                res["parity_rel_err_50it_bf16split"] = float(f"{float((out2[:1].cpu().double() - res['cpu_baseline']['_rec0'].double()).norm() / res['cpu_baseline']['_rec0'].double().norm()):.3e}")
            res["cpu_baseline"].pop("_rec0", None)
            ug = unit_gain_parity(dinv, model, denoiser, y, physics, args.iters)
            if ug:
                res["parity_unit_gain_50it"] = ug
        print(json.dumps(res))
    ctx.__exit__(None, None, None)


class PowerSampler:
    """socket power and shader clock of one GPU, sampled by `rocm-smi` from a host thread while the timed steps run (no GPU work,
    nothing inside the timed region waits for it); None when rocm-smi is not there"""

    def __init__(self, index, period=0.5):
        import threading
        self.samples, self.index, self.period = [], index, period
        self._stop = threading.Event()
        self._thread = None
        if index is not None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower"], capture_output=True, text=True,
                                     timeout=5).stdout
                clk = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", txt)
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                if clk and pw:
                    self.samples.append((int(clk.group(1)), float(pw.group(1))))
            except (OSError, subprocess.SubprocessError):
                return
            self._stop.wait(self.period)

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join(timeout=6)
        if not self.samples:
            return None
        clk = sorted(c for c, _ in self.samples)
        pw = sorted(p for _, p in self.samples)
        return {"samples": len(self.samples), "sclk_mhz_median": clk[len(clk) // 2], "sclk_mhz_min": clk[0],
                "socket_power_w_median": pw[len(pw) // 2], "socket_power_w_max": pw[-1], "source": "rocm-smi --showclocks --showpower"}


def sources_sha16(files):
    """hash of the kernel source files a profiles/pmc_traffic.json row was measured on (paths relative to the repo root)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(files):
        try:
            h.update(open(os.path.join(ROOT, f), "rb").read())
        except OSError:
            h.update(b"missing:" + f.encode())
    return h.hexdigest()[:16]


def layer_errors(device, B=4, H=80, C=256):
    """relative l2 error of ONE ResBlock convolution (level-2 shape of the headline configuration: 256 channels, 80x80) of
    every kernel against an fp64 evaluation of the same convolution, random normal inputs and weights"""
    from deepinv_amd.hip import drunet as K

    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, H, H, generator=g).to(device)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(device)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    geo = K.geom(B, H, H)
    xa = K.alloc(geo, C, device)
    xa[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:H + 1] = x.view(B, -1, 8, H, H).permute(1, 0, 3, 4, 2)
    out = {}

    def err(run):
        ya = K.alloc(geo, C, device)
        run(ya)
        y = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:H + 1].permute(1, 0, 4, 2, 3).reshape(B, C, H, H)
        return float(f"{float((y.double() - ref).norm() / ref.norm()):.3e}")

    wd, ci, co = K.pack_conv3x3_weight(w)
    out["fp32_direct (conv3x3_kernel)"] = err(lambda ya: K.conv3x3(geo, xa, wd, ci, co, ya))
    w2 = K.pack_winograd_weight(w)
    out["fp32_winograd_F(2x2,3x3) (conv3x3_wino_kernel)"] = err(lambda ya: K.conv3x3_winograd(geo, xa, w2, C, C, ya))
    w4 = K.pack_winograd4_weight(w)
    out["fp32_winograd_F(4x4,3x3) (conv3x3_wino4_kernel)"] = err(lambda ya: K.conv3x3_winograd4(geo, xa, w4, C, C, ya))
    ws = K.pack_split2d_weight(w)
    out["bf16x2_direct (conv3x3_split2d_kernel)"] = err(lambda ya: K.conv3x3_split(geo, xa, ws, C, C, ya))
    ww = K.pack_wsplit_weight(w)
    out["bf16x2_winograd_F(2,3) (conv3x3_wsplit_kernel)"] = err(lambda ya: K.conv3x3_wsplit(geo, xa, ww, C, C, ya))
    # the 2x2 stride-2 convolution between levels 1 and 2 (128 -> 256 channels): fp32 MFMA, three-part bf16 split with six
    # products (what conv_precision = "fp32" runs), two-part split with three products (conv_precision = "bf16split")
    C1 = C // 2
    x1 = torch.randn(B, C1, 2 * H, 2 * H, generator=g).to(device)
    wd = (torch.randn(C, C1, 2, 2, generator=g) / (2.0 * C1 ** 0.5)).to(device)
    refd = torch.nn.functional.conv2d(x1.double(), wd.double(), stride=2)
    gi = K.geom(B, 2 * H, 2 * H)
    x1a = K.alloc(gi, C1, device)
    x1a[:, gi.sl:gi.sl + gi.np].view(-1, B, gi.hp, gi.wp, 8)[:, :, 1:2 * H + 1, 1:2 * H + 1] = x1.view(B, -1, 8, 2 * H, 2 * H).permute(1, 0, 3, 4, 2)

    def errd(run):
        ya = K.alloc(geo, C, device)
        run(ya)
        y = ya[:, geo.sl:geo.sl + geo.np].view(-1, B, geo.hp, geo.wp, 8)[:, :, 1:H + 1, 1:H + 1].permute(1, 0, 4, 2, 3).reshape(B, C, H, H)
        return float(f"{float((y.double() - refd).norm() / refd.norm()):.3e}")

    wdf, wd3, wd2 = K.pack_down_weight(wd), K.pack_down_bf16x3_weight(wd), K.pack_down_bf16s_weight(wd)
    out["down2x2_fp32 (down2x2_kernel)"] = errd(lambda ya: K.down2x2(gi, geo, x1a, wdf, C1, C, ya))
    out["down2x2_bf16x3_six_products (down2x2_bf16s_kernel<3>)"] = errd(lambda ya: K.down2x2_bf16x3(gi, geo, x1a, wd3, C1, C, ya))
    out["down2x2_bf16x2 (down2x2_bf16s_kernel<2>)"] = errd(lambda ya: K.down2x2_bf16s(gi, geo, x1a, wd2, C1, C, ya))
    return out


def unit_gain_parity(dinv, model, denoiser, y, physics, iters):
    """50-iteration parity with O(1)-gain ResBlock convolutions against the REAL reference: the denoiser takes the weights of
    tests/golden/cfg2_named.npz (made by tests/golden/make_golden_r4.py through deepinv itself on slice 0 of this very batch),
    the loop runs on the whole batch in both arithmetic settings, slice 0 is compared with the fixture's strided subsample.
    (checker only: the fixture and the seeded weight generator are test infrastructure)"""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "cfg2_named.npz")
    if not os.path.exists(path) or iters != 50:
        return None
    from oracle import drunet_cpu as OD
    d = np.load(path)
    st = int(d["stride"])
    keep = {k: v.detach().clone() for k, v in denoiser.state_dict().items()}
    prec = denoiser.conv_precision
    res = {"res_gain": float(d["res_gain"]), "reference": "deepinv v0.4.1 (tests/golden/cfg2_named.npz), slice 0"}
    # seven more slices of this batch through the reference (tests/golden/cfg2_slices.npz, make_golden_r5.py; gain 0.2 weights)
    more = os.path.join(ROOT, "tests", "golden", "cfg2_slices.npz")
    ds = np.load(more) if os.path.exists(more) and y.shape[0] == 32 else None
    try:
        for tag, gain in (("gain_0.2", None), ("unit_gain", float(d["res_gain"]))):
            denoiser.load_state_dict(OD.init_state_dict(2, 2, seed=int(d["drunet_seed"]), res_gain=gain))
            ref = torch.from_numpy(np.asarray(d["rec" + ("" if gain is None else "_gain")])).double()
            for p in ("fp32", "bf16split"):
                denoiser.conv_precision = p
                with torch.no_grad():
                    full = model(y, physics).detach().cpu()
                rec = full[:1].reshape(-1)[::st].double()
                res[f"{tag}_{p}"] = float(f"{float((rec - ref).norm() / ref.norm()):.3e}")
                if ds is not None and gain is None:
                    errs = []
                    for k, i in enumerate(ds["slices"]):
                        r = torch.from_numpy(np.asarray(ds["rec"][k])).double()
                        errs.append(float((full[int(i)].reshape(-1)[::int(ds["stride"])].double() - r).norm() / r.norm()))
                    res[f"{tag}_{p}_slices"] = {"slices": [0] + [int(i) for i in ds["slices"]],
                                                "max_rel_err": float(f"{max(errs + [res[f'{tag}_{p}']]):.3e}")}
    finally:
        denoiser.load_state_dict(keep)
        denoiser.conv_precision = prec
    return res


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


def blur_and_single_coil_rows(dinv, device, op_row):
    """the remaining operators north_star names, at a batch of 32: BlurFFT and Blur (9x9 Gaussian, sigma 2: BASELINE configs[0]'s
    filter) on [32,3,256,256], single-coil MRI on [32,2,320,320].  Algorithmic bytes: image in + image out (+ the symbol
    buffers mask [1,3,256,129,2] / angle [1,3,256,129] c64 once per call for BlurFFT; the k-space mask once per call for MRI)."""
    g = torch.Generator().manual_seed(5)
    B, img = 32, (3, 256, 256)
    x = torch.rand(B, *img, generator=g).to(device)
    k = dinv.physics.functional.gaussian_blur(psf_size=(9, 9), sigma=(2.0, 2.0)).to(device)
    pf = dinv.physics.BlurFFT(img_size=img, filter=k, device=device)
    y = pf.A(x)
    alg = 2 * x.numel() * 4 + pf.mask.numel() * 4 + pf.angle.numel() * 8
    op_row("BlurFFT.A", "cfg1-shape x32", B, lambda: pf.A(x), alg)
    op_row("BlurFFT.A_adjoint", "cfg1-shape x32", B, lambda: pf.A_adjoint(y), alg)
    op_row("BlurFFT.prox_l2", "cfg1-shape x32", B, lambda: pf.prox_l2(x, y, 1.3), 4 * x.numel() * 4 + pf.mask.numel() * 4 + pf.angle.numel() * 8)
    for pad in ("circular", "valid"):
        pb = dinv.physics.Blur(filter=k, padding=pad, device=device)
        yb = pb.A(x)
        algb = (x.numel() + yb.numel()) * 4
        op_row(f"Blur({pad}).A", "cfg1-shape x32", B, lambda: pb.A(x), algb, VALU_TFLOP_per_s=2.0 * 81 * yb.numel() / 1e12)
        op_row(f"Blur({pad}).A_adjoint", "cfg1-shape x32", B, lambda: pb.A_adjoint(yb), algb, VALU_TFLOP_per_s=2.0 * 81 * x.numel() / 1e12)
    del x, y, pf
    H = W = 320
    xm = torch.rand(B, 2, H, W, generator=g).to(device)
    mask = dinv.utils.radial_mask(H, W, 80).to(device)
    pm = dinv.physics.MRI(mask=mask, img_size=(2, H, W), device=device)
    ym = pm.A(xm)
    algm = 2 * xm.numel() * 4 + 2 * H * W * 4
    op_row("MRI.A", "single coil 320x320 x32", B, lambda: pm.A(xm), algm)
    op_row("MRI.A_adjoint", "single coil 320x320 x32", B, lambda: pm.A_adjoint(ym), algm)


def other_config_ops(dinv, device, op_row, ops_last, loop_row, loop_rows_append):
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
    # Radon is a gather-rate problem, not an HBM one.  ONE model per kernel, the same in DESIGN 3.3:
    #   forward: every bilinear sample reads 4 taps x 4 B per image from the LDS window with ds_read_b128 - the chip's aggregate
    #            rate for that instruction is ~150 TB/s (256 B/clk/CU x 256 CUs x 2.4 GHz = 157, measured 150: MI355X_MICROARCH.md,
    #            LDS section); `frac_of_lds_model` = (samples x 16 B / time) / 150 TB/s  (upper bound of the traffic: the march
    #            skips the out-of-image part of the lattice);
    #   adjoint: vector-ALU work - ~150 instructions per (pixel of 8 images, angle) (3 x 3 candidate lattice points, 9 per
    #            candidate + staging) against 256 CUs x 128 lanes x 2.4 GHz = 78.6e12 lane-instructions/s; `frac_of_valu_model`.
    op_row("Tomography.A", "cfg3", B, lambda: phys.A(x), alg, n=5, Gsamples_per_s=smp / 1e9, lds_model_TB_per_s=smp * 16 / 1e12)
    op_row("Tomography.A_adjoint", "cfg3", B, lambda: phys.A_adjoint(y), alg, n=5, Gsamples_per_s=smp / 1e9,
           valu_model_Tinstr_per_s=float(-(-B // 8)) * W * W * A * 150.0 / 1e12)
    op_row("Tomography.fbp", "cfg3", B, lambda: phys.A_dagger(y, fbp=True), alg + 2 * B * G * A * 4, n=5)
    # cfg3's loop at its per-GPU shard: FBP-initialised PnP-HQS, 30 iterations (prox by CG with the reference's default stopping rule),
    # DRUNet(1->1) (1109 GFLOP per 512x512 call), DPIR-style schedules stretched to 30 iterations.  With the fixture
    # tests/golden/cfg3_full.npz (made by tests/golden/make_golden_r5.py through the REAL reference) the shard is 8 copies of the
    # fixture's image - the reference's CG stops on `torch.all(residual < tol)` over the batch, so identical units stop where its
    # single-image run stopped - and the row carries the FULL-LENGTH parity of the reconstruction (checker only; the same helper is
    # the body of tests/test_named_shapes_gpu.py::test_cfg3_fbp_pnp_hqs_full_length_30_iterations).
    del phys, y
    fixture = os.path.join(ROOT, "tests", "golden", "cfg3_full.npz")
    if os.path.exists(fixture):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_named_shapes_gpu as NS
        d3 = NS.load("cfg3_full")
        res3 = NS.full_length_cfg3(dinv, device, d3, B=B, precisions=("fp32", "bf16split"), runs=2)
        for prec, r in res3.items():
            loop_rows_append({"op": "FBP + PnP-HQS 30 it (CG prox) + DRUNet(1->1): loop" + ("" if prec == "fp32" else " [bf16split]"),
                              "config": "cfg3", "batch": B, "ms": round(r["seconds"] * 1e3, 1), "images_per_s": round(B / r["seconds"], 3),
                              "denoiser_TFLOP_per_s": round(30 * 1109.0 * B / r["seconds"] / 1e3, 1), "conv_precision": prec,
                              "finite": r["finite"], "parity_rel_err": float(f"{r['vs_reference']:.3e}"),
                              "parity_rel_err_max_over_iterations": float(f"{r['trace_max']:.3e}"),
                              "A_adjoint_A_calls": r["ata_calls"], "A_adjoint_A_calls_reference": r["ata_calls_reference"],
                              "parity": "image 0 of the shard after all 30 iterations against deepinv.optim.HQS on the same seeds "
                                        "(tests/golden/cfg3_full.npz), and the worst iteration of the denoiser-output trace (it follows a prox whose "
                                        "truncated CG stopped one or two iterations apart from the reference's: "
                                        "tests/test_named_shapes_gpu.py::test_cfg3_fbp_pnp_hqs_full_length_30_iterations)"})
    else:
        import numpy as np
        phys = dinv.physics.Tomography(angles=A, img_width=W, circle=False, normalize=True, device=device)
        y = phys.A(x)
        torch.manual_seed(0)
        den3 = dinv.models.DRUNet(1, 1, pretrained=None).to(device).eval()
        s30 = np.logspace(np.log10(49 / 255.0), np.log10(0.02), 30).astype("float32")
        st30 = ((s30 / 0.02) ** 2 / 0.23).astype("float32")
        hqs = dinv.optim.HQS(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(den3), stepsize=list(map(float, st30)),
                             g_param=list(map(float, s30)), max_iter=30, early_stop=False,
                             custom_init=lambda yy, p: p.A_dagger(yy, fbp=True))
        for prec in ("fp32", "bf16split"):
            den3.conv_precision = prec
            loop_row("FBP + PnP-HQS 30 it (CG prox) + DRUNet(1->1): loop" + ("" if prec == "fp32" else " [bf16split]"), "cfg3", B,
                     lambda: hqs(y, phys), 30 * 1109.0 * B, unit="images_per_s", precision=prec)
        del phys, y, den3, hqs
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
    # cfg5's loop at its per-GPU shard: DiffPIR, 100 steps, DRUNet(3->3) (277.6 GFLOP per 256x256 call), closed-form prox.  The
    # problem is the one of tests/golden/cfg5_full.npz (made by tests/golden/make_golden_r5.py through the REAL reference: image 0,
    # its measurement noise, the denoiser weights and the torch.randn_like stream are regenerated from the stored seeds), so the row
    # carries the FULL-LENGTH parity of image 0 against deepinv.sampling.DiffPIR and against the fp64 evaluation of the same sample
    # path (checker only; the same helper is the body of tests/test_named_shapes_gpu.py::test_cfg5_diffpir_full_length_100_steps).
    del phys, x, y, z
    fixture = os.path.join(ROOT, "tests", "golden", "cfg5_full.npz")
    if os.path.exists(fixture):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_named_shapes_gpu as NS
        d5 = NS.load("cfg5_full")
        res5 = NS.full_length_cfg5(dinv, device, d5, B=B, precisions=("fp32", "bf16split"), runs=2)
        for prec, r in res5.items():
            loop_rows_append({"op": "DiffPIR 100 steps + DRUNet(3->3): loop" + ("" if prec == "fp32" else " [bf16split]"), "config": "cfg5",
                              "batch": B, "ms": round(r["seconds"] * 1e3, 1), "images_per_s": round(B / r["seconds"], 3),
                              "denoiser_TFLOP_per_s": round(100 * 277.6 * B / r["seconds"] / 1e3, 1), "conv_precision": prec,
                              "finite": r["finite"], "parity_rel_err": float(f"{r['vs_reference']:.3e}"),
                              "parity_rel_err_vs_fp64": float(f"{r['vs_fp64']:.3e}"),
                              "parity": "image 0 of the shard after all 100 steps against deepinv.sampling.DiffPIR on the same seeds "
                                        "(tests/golden/cfg5_full.npz); the reference's own fp32 rounding along this path is "
                                        f"{float(d5['out_err_vs_exact']):.1e} (its distance to the fp64 evaluation)"})


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
    # what backs kind = "port": the port and the REAL reference were run on this very problem (slice 0, 50 iterations) in the
    # build container, where the reference imports (tests/golden/make_golden_r4.py): identical results, same seconds per slice
    backing = None
    try:
        import numpy as np
        g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_named.npz"))
        backing = {"where": "build container, %d threads" % int(g["threads"]), "seconds_per_slice_reference": round(float(g["seconds_reference"]), 1),
                   "seconds_per_slice_port": round(float(g["seconds_port"]), 1), "rel_diff_port_vs_reference": float(g["port_vs_reference"])}
    except (OSError, KeyError):
        pass
    return {"_rec0": recs[0], "port_vs_reference": backing, "value": round(1.0 / (per_it * iters), 5), "unit": "slices/s", "cores": threads, "host_cores": cores,
            "usable_cores": usable, "kind": "port", "threads_calibration_s_per_denoiser_call": calib,
            "samples_s_per_iteration": [round(v, 4) for v in samples],
            "sample": f"slice 0 of the batch: median of 3 timed samples ({iters} it once, {short} it twice), "
                      f"{per_it * iters:.1f} s per slice",
            "whole_box": whole,
            "parity_rel_err_max": float(f"{max(errs.values()):.3e}"), "parity_slices": len(errs)}


if __name__ == "__main__":
    main()
