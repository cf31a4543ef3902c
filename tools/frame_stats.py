"""Counters of a few benchmark frames (fast mode), for tuning."""
import os
import sys

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
for step in (0, 30, 90, 200):
    ctx.set_camera(*bench.orbit_camera(step))
    st = ctx.render_device(w, h, out.data_ptr(), stats=True)
    print(step, {k: st[k] for k in ("rays_hit", "samples_evaluated", "cells_read", "bricks_touched", "long_rays", "replay_rays")},
          "taps/ray %.1f lookups/ray %.1f" % (st["samples_evaluated"] / st["rays_hit"], st["cells_read"] / st["rays_hit"]))
ctx.close()
