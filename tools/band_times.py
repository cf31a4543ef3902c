"""Diagnostic: device time of one sort-first part's share of the benchmark frame (what one of N GPUs
renders), for N = 1, 2, 4, 8, all on this GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vulkanrenderer_amd as vr  # noqa: E402

n, w, h = 2048, 3840, 2160
ctx = vr.VolumeRenderer(0)
raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.phantom_generate_device(raw.data_ptr(), 2, n)
ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n)
ctx.synchronize()
del raw
ctx.update_transfer_function(bench.TF_GRAPHS, reset_state=True)
ctx.set_sampling(bench.samples_for(n))
ctx.set_mode(1)
out = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
# the orbit of the benchmark: 1 degree a frame (BAND_STEP), all the way round
step = int(os.environ.get("BAND_STEP", "1"))
frames = 360 // step
outs = [out] + [torch.empty(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
PARTS = [int(x) for x in os.environ.get("BAND_PARTS", "1,2,4,8").split(",")]
for depth in [int(x) for x in os.environ.get("BAND_DEPTHS", "1,2").split(",")]:
  ctx.set_pipeline(depth)
  print("frames in flight:", depth, flush=True)
  for parts in PARTS:
    worst = 0.0
    for part in range(parts):
        for i in range(2):
            ctx.set_camera(*bench.orbit_camera(i))
            ctx.render_bands(w, h, part, parts, ptr=outs[i % 3].data_ptr())
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            ctx.set_camera(*bench.orbit_camera(i * step))
            ctx.render_bands(w, h, part, parts, ptr=outs[i % 3].data_ptr())
        ctx.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / frames * 1e3)
    print("parts %d: slowest part %.3f ms per frame" % (parts, worst), flush=True)
ctx.close()
