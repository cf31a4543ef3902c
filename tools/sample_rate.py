"""Taps per second of the plain reference-order march (every lane busy, no skipping): the
sampling speed the other kernels can be compared with."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vulkanrenderer_amd as vr  # noqa: E402

for n, w, h in ((512, 1024, 1024), (2048, 1024, 1024)):
    ctx = vr.VolumeRenderer(0)
    raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.phantom_generate_device(raw.data_ptr(), 2, n)
    ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n)
    ctx.synchronize()
    del raw
    ctx.update_transfer_function(bench.TF_GRAPHS, reset_state=True)
    samples = bench.samples_for(n)
    ctx.set_sampling(samples)
    out = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.set_camera(*bench.orbit_camera(30))
    hits = ctx.count_hit_rays(w, h)
    for mode in (0, 2, 1):
        ctx.set_mode(mode)
        for _ in range(3):
            ctx.render_device(w, h, out.data_ptr())
        ms = ctx.march_kernel_times(2).mean()
        print("n=%d %dx%d mode %d: %.3f ms, nominal %.1f Gtaps/s" % (n, w, h, mode, ms, hits * (samples + 2) / ms / 1e6))
    ctx.close()
