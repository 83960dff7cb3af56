"""repeat the full-length cfg5 run (tests/test_named_shapes_gpu.py: full_length_cfg5) with and without batch lanes and print the trace
errors of every run: is the 1.8e-3 glitch of one bf16-split trace step a race of the lanes?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_named_shapes_gpu as NS  # noqa: E402
import deepinv_amd as dinv  # noqa: E402

dev = torch.device("cuda:0")
d = NS.load("cfg5_full")
for lanes in ("auto", 1, "auto", 1):
    dinv.models.DRUNet.batch_lanes = lanes
    for rep in range(3):
        res = NS.full_length_cfg5(dinv, dev, d, precisions=("bf16split", "fp32"))
        print(lanes, rep, {k: (round(v["trace_vs_fp64_max"], 7), round(v["vs_fp64"], 8)) for k, v in res.items()}, flush=True)
