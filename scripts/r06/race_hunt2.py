"""the bf16-split model under batch lanes is not reproducible run to run (scripts/r06/race_hunt.py): is it the concurrency on the device
or the host-side sequence?  V1: both lanes on ONE stream; V2: two streams, device synchronised between the lanes; V3: concurrent."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepinv_amd as dinv  # noqa: E402
import deepinv_amd.hip as H  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
den = dinv.models.DRUNet(3, 3, pretrained=None).to(dev).eval()
den.conv_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16split"
x = torch.rand(16, 3, 256, 256, generator=gen).to(dev)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with torch.no_grad():
    den.batch_lanes = 1
    ref = den(x, 0.1)
    ref_half = torch.cat((den(x[:8], 0.1), den(x[8:], 0.1)))
    print(json.dumps({"halves_vs_whole_max_abs": float((ref_half - ref).abs().max())}))
    den.batch_lanes = 2
    for name, streams, sync in (("V1 one stream", [s0, s0], False), ("V2 two streams, serialised", [s0, s1], True), ("V3 concurrent", [s0, s1], False)):
        H._LANE_STREAMS[H.lane_key(dev, 2)] = streams
        outs = []
        for _ in range(20):
            if sync:
                # serialise: run lane by lane with a device synchronisation in between (monkeypatched wait)
                orig = den._hip_forward_lane

                def synced(*a, **k):
                    r = orig(*a, **k)
                    torch.cuda.synchronize()
                    return r

                den._hip_forward_lane = synced
                try:
                    outs.append(den(x, 0.1))
                finally:
                    del den._hip_forward_lane
            else:
                outs.append(den(x, 0.1))
        torch.cuda.synchronize()
        diffs = [float((o - ref_half).abs().max()) for o in outs]
        print(json.dumps({"variant": name, "distinct": len(set(diffs)), "max_abs_diff_vs_halves": max(diffs), "min": min(diffs)}), flush=True)
