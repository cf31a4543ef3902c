"""Wall time of vr_upload_volume from pageable host memory (copy over PCIe + re-tiling on the device).

python tools/upload_time.py [edge ...]   (default 1024; u16)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkanrenderer_amd as vr  # noqa: E402


def main():
    edges = [int(a) for a in sys.argv[1:]] or [1024]
    for n in edges:
        vox = np.empty((n, n, n), np.uint16)
        vox[:] = (np.arange(n, dtype=np.uint16) * 7 + 1)[None, None, :]
        with vr.VolumeRenderer(0) as ctx:
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.upload_volume(vox)
                ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
            print("edge %d: vr_upload_volume %.1f ms for %.2f GB of voxels = %.1f GB/s from pageable host memory" % (
                n, best * 1e3, vox.nbytes / 1e9, vox.nbytes / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
