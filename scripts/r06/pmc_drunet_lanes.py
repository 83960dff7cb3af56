"""The fp32 DRUNet call of the bench (32 slices of 320 x 320) as it runs since round 6 - two batch lanes of 16 slices, F(4x4) launches without
the channel split of the last round - for the rocprofv3 counter passes (scripts/r06/pmc_drunet.sh).  Counter collection serialises the
kernels, so the lanes are FORCED here (no calibration by timing): the counters describe a lane's launch running alone."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402
import deepinv_amd.hip as H  # noqa: E402

dev = torch.device("cuda:0")
model = dinv.models.DRUNet(2, 2, pretrained=None).to(dev).eval()
model.conv_precision = "fp32"
model.batch_lanes = 2
H._LANE_STREAMS[H.lane_key(dev, 2)] = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
x = torch.rand(32, 2, 320, 320, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = model(x, 0.05)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
