"""vr_render_static: the entry point that allocates nothing, against vr_render on the same inputs.

Collected last on purpose (the file name sorts behind the other tests)."""
import math
import os

import numpy as np
import pytest

import conftest as cf

pytestmark = pytest.mark.gpu


def test_static_scratch_frames_equal_vr_render():
    import torch
    import vulkanrenderer_amd as vr
    from oracle import vr_oracle as o
    o.build()
    assert torch.cuda.is_available()
    vox = o.phantom(96, np.uint16)
    vol, _ = o.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    o.lut_update(lut, cf.TF_GRAPHS)
    samples = 333
    with vr.VolumeRenderer(0) as ctx:
        ctx.upload_volume(vox)
        ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        ctx.set_sampling(samples, (0, 0, 0, 0))
        ctx.set_mode(vr.MODE_REFERENCE_ORDER)      # the static entry point marches fast whatever mode is set
        # two calls in a row, the second with a smaller frame and another camera: what the first left in
        # the shared scratch (queues, tile ranges, counters) must not leak into the second
        big_probe = torch.zeros((700, 1000, 4), dtype=torch.uint8, device="cuda")
        for (w, h), cam in (((640, 360), cf.CAMERAS["h30_v20"]), ((200, 150), cf.CAMERAS["hm135_vm60"])):
            ctx.set_camera(*cam)
            want = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            got = torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.set_mode(vr.MODE_FAST)
            ctx.render_device(w, h, want.data_ptr())
            ctx.set_mode(vr.MODE_REFERENCE_ORDER)
            # the promise itself: while every allocation and release of the library is refused (and counted)
            # the call still succeeds, and the GPU's free memory is what it was
            ctx.synchronize()
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            os.environ["VR_TEST_FORBID_ALLOC"] = "1"
            try:
                ctx.render_static(w, h, got.data_ptr())
                ctx.render_static(w, h, got.data_ptr())
                ctx.synchronize()
            finally:
                del os.environ["VR_TEST_FORBID_ALLOC"]
            assert torch.cuda.mem_get_info()[0] == free_before, (w, h)
            assert np.array_equal(got.cpu().numpy(), want.cpu().numpy()), (w, h)
            # and both are the checker's frame
            vp, m = o.camera(cam[0], cam[1], w, h)
            ref = o.render(vol, lut, vp[4:8], m, w, h, samples)
            assert np.array_equal(got.cpu().numpy(), ref), (w, h)
        # the hook itself: vr_render's buffers for a frame size it has not seen are allocated on demand, and refused
        os.environ["VR_TEST_FORBID_ALLOC"] = "1"
        try:
            ctx.set_mode(vr.MODE_FAST)
            with pytest.raises(vr.VrError):
                ctx.render_device(1000, 700, big_probe.data_ptr())
        finally:
            del os.environ["VR_TEST_FORBID_ALLOC"]
        # a second context on the same GPU uses the same array (one after the other: fine)
        with vr.VolumeRenderer(0) as other:
            other.upload_volume(vox)
            other.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            other.set_sampling(samples, (0, 0, 0, 0))
            other.set_camera(*cf.CAMERAS["hm135_vm60"])
            again = torch.zeros((150, 200, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            other.render_static(200, 150, again.data_ptr())
            other.synchronize()
            assert np.array_equal(again.cpu().numpy(), got.cpu().numpy())
        # the limit: the reference's 1280x720 window fits, one row more does not
        assert vr.static_max_pixels() == 1280 * 720
        big = torch.zeros((721, 1280, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.set_camera(*cf.CAMERAS["h30_v20"])
        ctx.render_static(1280, 720, big.data_ptr())
        ctx.set_mode(vr.MODE_FAST)
        full = ctx.render(1280, 720)
        ctx.synchronize()
        assert np.array_equal(big[:720].cpu().numpy(), full)
        with pytest.raises(vr.VrError, match="static scratch"):
            ctx.render_static(1280, 721, big.data_ptr())
        with pytest.raises(vr.VrError, match="static scratch"):
            ctx.render_static(3840, 2160, big.data_ptr())
    with vr.VolumeRenderer(0) as empty:
        with pytest.raises(vr.VrError):
            empty.render_static(64, 64, big.data_ptr())        # no volume
