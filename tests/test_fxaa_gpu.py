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


@pytest.mark.parametrize("size", [(403, 301), (64, 40), (131, 9), (9, 131), (404, 301), (496, 50), (500, 37), (252, 64), (8, 40), (12, 131)])
def test_fxaa_around_the_edge_threshold(scene, size):
    """Images that sweep through the point where the taps leave the pixel: brightness ramps across,
    the amplitude of a small noise ramps down the image, so scaled directions of every size near 1
    occur at every brightness.  Odd sizes, and frames barely larger than the untouched border; widths
    that are multiples of 4 take the kernel with four pixels per lane (one wave's 248 columns, two
    waves' worth, a last wave with a few columns), the others the kernel with one."""
    import torch

    import oracle
    from raymarching_amd import Engine

    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    ramp = np.linspace(0.0, 255.0, W)[None, :, None]
    amp = np.linspace(0.0, 6.0, H)[:, None, None]
    img = np.clip(np.rint(ramp + amp * rng.standard_normal((H, W, 4))), 0, 255).astype(np.uint8)
    img[..., 3] = rng.integers(0, 256, (H, W), dtype=np.uint8)  # the fourth byte passes through
    ref = oracle.fxaa(img)
    e = Engine(scene, W, H)
    src = torch.from_numpy(img).cuda()
    dst = torch.full_like(src, 9)
    e.fxaa_device(src.data_ptr(), dst.data_ptr(), 0, H)
    e.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got, ref)
    changed = (got != img).any(axis=-1).mean()
    if W > 100 and H > 100:
        assert 0.05 < changed < 0.95, changed  # both kinds of pixel are present
    # a band that starts and ends inside the image
    if H > 20:
        dst2 = torch.full_like(src, 7)
        e.fxaa_device(src.data_ptr(), dst2.data_ptr(), 5, H - 6)
        e.synchronize()
        got2 = dst2.cpu().numpy()
        assert np.array_equal(got2[5:H - 6], ref[5:H - 6]) and (got2[:5] == 7).all() and (got2[H - 6:] == 7).all()
    e.close()
