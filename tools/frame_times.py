"""Per-kernel device times of a few benchmark frames, measured with HIP events around the whole frame
(fast march + long-ray march + replay) and around the fast march alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
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
frames = 24
for i in range(3):
    ctx.set_camera(*bench.orbit_camera(i))
    ctx.render_device(w, h, out.data_ptr())
ctx.synchronize()
t0 = time.perf_counter()
for i in range(frames):
    ctx.set_camera(*bench.orbit_camera(i * 15))
    ctx.render_device(w, h, out.data_ptr())
ctx.synchronize()
dt = (time.perf_counter() - t0) / frames * 1e3
print("%s: frame %.3f ms wall, fast march %.3f ms" % (os.environ.get("VR_LIB_NAME", "libvr_hip.so"), dt, float(np.mean(ctx.march_kernel_times(frames)))))
ctx.close()
