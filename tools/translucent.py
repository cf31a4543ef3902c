"""Diagnostic: a transfer function with low alpha everywhere (rays rarely become opaque) at 512^3, 1920x1080,
in the three modes: device time per frame, taps fetched, rays handed over and replayed."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import vulkanrenderer_amd as vr  # noqa: E402

XS = (0.0, 60.0, 300.0, 483.0)
TRANSLUCENT = [list(zip(XS, c)) for c in ((0.2, .9, 1, 1), (0.1, .3, .8, 1), (0.3, .1, .5, 1), (0.0, 0.004, 0.01, 0.02))]

n, w, h = 512, 1920, 1080
ctx = vr.VolumeRenderer(0)
raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.phantom_generate_device(raw.data_ptr(), 2, n)
ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n)
ctx.synchronize()
del raw
ctx.set_sampling(bench.samples_for(n))
out = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
stream = torch.cuda.ExternalStream(ctx.stream)
torch.cuda.synchronize()
for name, graphs in (("benchmark table", bench.TF_GRAPHS), ("translucent table", TRANSLUCENT)):
    ctx.update_transfer_function(graphs, reset_state=True)
    frames = {}
    for mode in ((0, 2, 1) if len(sys.argv) > 1 else (2, 1)):
        ctx.set_mode(mode)
        ms = []
        for k in range(6):
            ctx.set_camera(*bench.orbit_camera(40 * k))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            ctx.render_device(w, h, out.data_ptr())
            b.record(stream)
            ctx.synchronize()
            if k:
                ms.append(a.elapsed_time(b))
        ctx.set_camera(*bench.orbit_camera(40))
        st = ctx.render_device(w, h, out.data_ptr(), stats=True)
        ctx.synchronize()
        frames[mode] = out.cpu().numpy().copy()
        hit = max(1, st["rays_hit"])
        print("%s, mode %d: %.3f ms/frame; look-ups/ray %.1f, taps/ray %.1f (nominal %d), handed over %.1f%%, replayed %.1f%%, replay taps %.1f M, wave taps %.1f M" % (
            name, mode, float(np.mean(ms)), st["cells_read"] / hit, st["samples_evaluated"] / hit, bench.samples_for(n) + 2, 100.0 * st["long_rays"] / hit,
            100.0 * st["replay_rays"] / hit, st["replay_samples"] / 1e6, st["long_samples"] / 1e6))
    print("   frames identical:", bool(np.array_equal(frames[1], frames[2])))
ctx.close()
