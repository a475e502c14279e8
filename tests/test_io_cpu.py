"""Dataset arithmetic (reference io.py:38-58) against tests/golden/io_onehot.npz, produced by the reference's own
COCOStuffDataset.__getitem__ driven on fixed decoded tensors (tests/golden/make_golden.py io_onehot)."""
import os

import numpy as np
import torch

from tests.golden_util import GOLDEN_DIR


def _fixture():
    return np.load(os.path.join(GOLDEN_DIR, 'io_onehot.npz'))


def _write_files(z, folder):
    """The fixture's decoded bytes as files the dataset globs (*.jpg / *.png).  Both are written losslessly (PNG data; Pillow
    picks the decoder from the content, not the extension) so the decoded tensors are exactly the fixture's."""
    from PIL import Image
    os.makedirs(folder / 'img')
    os.makedirs(folder / 'mask')
    Image.fromarray(np.ascontiguousarray(z['img_u8'].transpose(1, 2, 0))).save(str(folder / 'img' / '000000000007.jpg'), format='PNG')
    Image.fromarray(z['lab_u8'][0]).save(str(folder / 'mask' / '000000000007.png'), format='PNG')


def test_float_items_equal_the_reference_dataset(tmp_path):
    from patchgan_amd.io import COCOStuffDataset
    z = _fixture()
    _write_files(z, tmp_path)
    ds = COCOStuffDataset(str(tmp_path / 'img'), str(tmp_path / 'mask'), labels=[int(v) for v in z['labels']], size=40)
    x, y = ds[0]
    assert x.dtype == torch.float32 and y.dtype == torch.float32
    assert np.array_equal(x.numpy(), z['x']) and np.array_equal(y.numpy(), z['y'])


def test_byte_items_carry_the_same_information(tmp_path):
    """device_pipeline=True hands over the decoded bytes; the arithmetic the GPU kernels then apply (pg_u8_to_f32: / 255.;
    pg_labels_to_onehot: (uint8)(v + 1) == label over the sorted labels) restated in numpy equals the fixture bit for bit.
    tests/test_step_gpu.py::test_device_input_kernels_vs_reference_fixture runs the kernels themselves."""
    from patchgan_amd.io import COCOStuffDataset
    z = _fixture()
    _write_files(z, tmp_path)
    labels = [int(v) for v in z['labels']]
    ds = COCOStuffDataset(str(tmp_path / 'img'), str(tmp_path / 'mask'), labels=labels, size=40, device_pipeline=True)
    img, lab = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (40, 56, 3) and lab.dtype == torch.uint8 and tuple(lab.shape) == (40, 56)
    x = img.numpy().transpose(2, 0, 1).astype(np.float32) / np.float32(255.)
    y = np.stack([((lab.numpy() + np.uint8(1)).astype(np.uint8) == v).astype(np.float32) for v in np.sort(labels)])
    assert np.array_equal(x, z['x']) and np.array_equal(y, z['y'])
