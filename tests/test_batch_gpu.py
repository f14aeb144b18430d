"""GPU: device-side crop + augmentation (ssie_assemble_batch) is bit-exact against the reference fixtures
(cube, (x, y, mode)) -> patch from tests/golden/aux.npz (model.py:306-309 + utils.py:7-34)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_assemble_batch_bit_exact(golden_dir):
    import ssie
    ssie.load()
    from ssie_amd import hostlib as H
    g = np.load(os.path.join(golden_dir, "aux.npz"))
    cube = torch.from_numpy(g["aug_cube"]).cuda().contiguous()
    crops = [(0, int(x0), int(y0), int(m)) for (x0, y0, m) in g["aug_crops"]]
    batch = H.assemble_batch([cube], crops, 16, cube.shape[2])
    torch.cuda.synchronize()
    assert batch.shape == (8, cube.shape[2], 16, 16)
    for i, (_, x0, y0, m) in enumerate(crops):
        ref = torch.from_numpy(np.ascontiguousarray(g["aug_patch_%d" % m])).permute(2, 0, 1)
        assert torch.equal(batch[i].cpu(), ref), m
