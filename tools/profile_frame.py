"""Renders a few frames of the benchmark workload and nothing else: the program to put after
`rocprofv3 ... --` when collecting counters for the march kernels."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import vulkanrenderer_amd as vr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--volume", type=int, default=2048)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--parts", type=int, default=1, help="render only sort-first part 0 of this many")
ap.add_argument("--shading", action="store_true", help="gradient lighting on")
ap.add_argument("--table", choices=("benchmark", "translucent"), default="benchmark")
a = ap.parse_args()
n = a.volume
ctx = vr.VolumeRenderer(0)
raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.phantom_generate_device(raw.data_ptr(), 2, n)
ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n)
ctx.synchronize()
del raw
ctx.update_transfer_function(bench.TRANSLUCENT_GRAPHS if a.table == "translucent" else bench.TF_GRAPHS, reset_state=True)
ctx.set_sampling(bench.samples_for(n))
ctx.set_mode(a.mode)
if a.shading:
    ctx.set_shading(True)
out = torch.empty(a.width * a.height * 4, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for i in range(a.frames):
    ctx.set_camera(*bench.orbit_camera(i * 30))
    if a.parts > 1:
        ctx.render_bands(a.width, a.height, 0, a.parts, ptr=out.data_ptr())
    else:
        ctx.render_device(a.width, a.height, out.data_ptr())
ctx.synchronize()
print("ms", ctx.march_kernel_times(a.frames))
ctx.close()
