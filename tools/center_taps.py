"""Diagnostic: taps the fast march evaluates per ray in small rectangles of the benchmark frame."""
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
ctx.set_camera(*bench.orbit_camera(30))
for mode in (1, 2):
    ctx.set_mode(mode)
    for (x, y) in ((1920, 1080), (1600, 900), (2200, 1200), (1920, 600), (1400, 1080), (1920, 400)):
        img, st = ctx.render(w, h, stats=True, rect=(x, y, x + 8, y + 8))
        print("mode", mode, "rect", x, y, "hit", st["rays_hit"], "taps/ray %.1f" % (st["samples_evaluated"] / 64.0),
              "lookups/ray %.1f" % (st["cells_read"] / 64.0), "long", st["long_rays"], "replay", st["replay_rays"],
              "pixel", img[0, 0].tolist())
ctx.close()
