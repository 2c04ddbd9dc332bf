"""rm_engine_compute_lightmaps_static: the lightmap precompute with its work buffers in the library's
static device array, called through the C ABI and compared bit for bit with
rm_engine_compute_lightmaps on the same scenes - twice in a row, the second time with a smaller
input, so that whatever the first call left in the shared array must not show."""
import numpy as np
import pytest

from raymarching_amd.scene import Scene, default_scene

pytestmark = pytest.mark.gpu


def _lightmaps(eng):
    return np.stack([eng.get_lightmap(i) for i in range(eng.n_objects)])


def _pair(scene, map_size, bounces):
    """(lightmaps of the allocating entry point, lightmaps of the static one, device bytes the static call took)."""
    import torch

    from raymarching_amd import Engine

    ref_eng = Engine(scene, 64, 64, map_size=map_size)
    ref_eng.compute_lightmaps(bounces)
    ref = _lightmaps(ref_eng)
    ref_eng.close()
    eng = Engine(scene, 64, 64, map_size=map_size)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    eng.compute_lightmaps_static(bounces)
    free_after = torch.cuda.mem_get_info()[0]
    got = _lightmaps(eng)
    eng.close()
    return ref, got, free_before - free_after


def test_static_lightmaps_equal_the_allocating_entry_point():
    from raymarching_amd import Engine, _lib

    base = default_scene()
    # first call: the default scene at the reference's MAP_SIZE (512 patches, 4 bounces)
    ref, got, taken = _pair(base, 8, 4)
    assert np.abs(ref).max() > 0
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    assert taken == 0, f"the static call took {taken} bytes of device memory"
    # second call, smaller input: six objects (one sphere and one wall gone), MAP_SIZE 4 (96 patches);
    # the array still holds the first call's clouds and form factors
    small = Scene(objects=[o for i, o in enumerate(base.objects) if i not in (0, 7)], camera_position=base.camera_position, light=base.light)
    ref2, got2, taken2 = _pair(small, 4, 3)
    assert np.abs(ref2).max() > 0
    assert np.array_equal(ref2.view(np.uint32), got2.view(np.uint32))
    assert taken2 == 0
    # and the first input again, after the smaller one
    ref3, got3, _ = _pair(base, 8, 4)
    assert np.array_equal(ref3.view(np.uint32), got3.view(np.uint32)) and np.array_equal(ref3.view(np.uint32), ref.view(np.uint32))
    # zero bounces: direct lighting only (lightmaps all zero with no emissive material), no form factors
    ref4, got4, _ = _pair(base, 8, 0)
    assert np.array_equal(ref4.view(np.uint32), got4.view(np.uint32))
    # beyond the static array: refused with a message, nothing computed
    big = Engine(base, 64, 64, map_size=16)  # 8 * 16 * 16 = 2048 patches
    with pytest.raises(_lib.RMError, match="2048 patches exceed the static scratch"):
        big.compute_lightmaps_static(4)
    assert np.abs(_lightmaps(big)).max() == 0
    big.compute_lightmaps(1)  # the allocating entry point still serves it
    assert np.abs(_lightmaps(big)).max() > 0
    big.close()
