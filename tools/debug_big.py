import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vulkanrenderer_amd as vr
import bench
for n in (512, 1024, 2048):
    ctx = vr.VolumeRenderer(0)
    raw = torch.empty(n**3, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    t=time.time()
    ctx.phantom_generate_device(raw.data_ptr(), 2, n); ctx.synchronize(); t1=time.time()-t
    nz = int((raw[::4097] != 0).sum().item())
    print(n, "phantom s", round(t1,3), "nonzero frac (strided)", nz / (raw.numel()/4097), "max", int(raw.view(torch.uint8).max().item()))
    t=time.time(); ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n); ctx.synchronize(); print(" upload s", round(time.time()-t,3), ctx.volume_info())
    del raw
    ctx.update_transfer_function(bench.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(bench.samples_for(n)); ctx.set_camera(math.radians(30), math.radians(20))
    for mode in (2, 1):
        ctx.set_mode(mode)
        img, st = ctx.render(512, 512, stats=True)
        print(" mode", mode, {k: st[k] for k in ("rays_hit","samples_evaluated","cells_read","bricks_touched","replay_rays","kernel_ms")}, "nonblack", float((img[...,:3].sum(-1)>0).mean()))
    ctx.close()
