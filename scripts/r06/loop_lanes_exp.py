"""Experiment: batch lanes at the LOOP level - the two halves of the batch run their whole 50-iteration PnP-PGD loops on two streams
(joined once, at the end), optionally with lane B started half a DRUNet forward behind lane A, so that the two lanes sit at
different U-Net levels (lane A's epilogue-heavy full-resolution launches beside lane B's matrix-dense deep ones).  Compared with
the per-call lanes of DRUNet.batch_lanes (joined at the end of every denoiser call).
    python scripts/r06/loop_lanes_exp.py [batch] [graph 0/1]"""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deepinv_amd as dinv  # noqa: E402
from deepinv_amd.hip import drunet as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
GRAPH = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
physics, x_true, y, maps, mask = bench.make_problem(dinv, B, 0, 320, 320, 8, dev)
torch.manual_seed(0)
den = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()


def make_model(d):
    m = dinv.optim.PGD(data_fidelity=dinv.optim.L2(), prior=dinv.optim.PnP(d), stepsize=1.0, g_param=0.05, max_iter=50, early_stop=False)
    m.fixed_point.use_graph = GRAPH
    return m


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


# ---- reference points: one launch sequence, per-call lanes
den.batch_lanes = 1
m1 = make_model(den)
with torch.no_grad():
    t_one, ref = timeit(lambda: m1(y, physics))
    den.batch_lanes = 2
    t_call, out_call = timeit(lambda: m1(y, physics))
print(json.dumps({"batch": B, "graph": GRAPH, "one_sequence_ms": round(t_one, 1), "per_call_lanes_ms": round(t_call, 1),
                  "rel_diff": float((out_call - ref).norm() / ref.norm())}), flush=True)

# ---- loop-level lanes: two denoiser objects (own activation buffers), shared weights, one stream each
den_b = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
den_b.load_state_dict(den.state_dict())
dens = [den, den_b]
for d in dens:
    d.batch_lanes = 1
models = [make_model(d) for d in dens]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
if GRAPH:
    from deepinv_amd.optim import fixed_point as FP
    caps = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
half = B // 2
ys = [y[:half].contiguous(), y[half:].contiguous()]


def loop_lanes(offset):
    """one HOST THREAD per lane (the loops contain host synchronisations - parameter checks, graph capture - that would serialise
    two loops driven by one thread)"""
    import threading
    cur = torch.cuda.current_stream(dev)
    outs = [None, None]
    ev = torch.cuda.Event()
    state = {"calls": 0, "armed": offset, "recorded": threading.Event()}
    orig = K.down2x2_bf16x3

    def counting(*a, **kw):
        r = orig(*a, **kw)
        if state["armed"] and threading.current_thread().name == "lane0":
            state["calls"] += 1
            if state["calls"] == 3:         # the third stride-2 convolution of lane A's first forward: the middle of the U-Net
                ev.record(torch.cuda.current_stream(dev))
                state["armed"] = False
                state["recorded"].set()
        return r

    K.down2x2_bf16x3 = counting

    def run(i):
        torch.cuda.set_device(dev)
        with torch.no_grad():
            streams[i].wait_stream(cur)
            if i == 1 and offset:
                state["recorded"].wait(timeout=10)
                streams[1].wait_event(ev)
            with torch.cuda.stream(streams[i]):
                outs[i] = models[i](ys[i], physics)

    try:
        th = [threading.Thread(target=run, args=(i,), name=f"lane{i}") for i in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        K.down2x2_bf16x3 = orig
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


with torch.no_grad():
    for offset in (False, True):
        t, out = timeit(lambda: loop_lanes(offset))
        print(json.dumps({"batch": B, "graph": GRAPH, "loop_lanes_ms": round(t, 1), "half_forward_offset": offset,
                          "rel_diff": float((out - ref).norm() / ref.norm())}), flush=True)
