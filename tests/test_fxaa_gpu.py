"""GPU parity test for the FXAA pass (out-of-place reading of fxaa.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fxaa_bit_exact(scene, oracle_lightmaps):
    import torch

    import oracle
    from conftest import make_oracle
    from raymarching_amd import Engine

    W, H = 400, 300
    lm, _ = oracle_lightmaps
    ora = make_oracle(scene, lm[-1])
    frame = ora.render(W, H)
    ref = oracle.fxaa(frame)
    e = Engine(scene, W, H)
    src = torch.from_numpy(frame).cuda()
    dst = torch.zeros_like(src)
    e.fxaa_device(src.data_ptr(), dst.data_ptr(), 0, H)
    e.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got, ref)
    assert not np.array_equal(got, frame), "FXAA changed nothing: the test frame has no edges?"
    # bands: rows [r0,r1) written, everything else untouched
    dst2 = torch.full_like(src, 7)
    e.fxaa_device(src.data_ptr(), dst2.data_ptr(), 100, 163)
    e.synchronize()
    got2 = dst2.cpu().numpy()
    assert np.array_equal(got2[100:163], ref[100:163])
    assert (got2[:100] == 7).all() and (got2[163:] == 7).all()
    # noise image: exercises every branch of the tap selection
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    refn = oracle.fxaa(noise)
    srcn = torch.from_numpy(noise).cuda()
    dstn = torch.zeros_like(srcn)
    e.fxaa_device(srcn.data_ptr(), dstn.data_ptr(), 0, H)
    e.synchronize()
    assert np.array_equal(dstn.cpu().numpy(), refn)
    e.close()
