"""Host-side logic that needs no GPU: YAML schemas of patchgan_train, tiling helpers of patchgan_infer, dataset code."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import patchgan_oracle as O
from tests.golden_util import GOLDEN_DIR, probe

REF_YAML = """
dataset:
  type: COCOStuff
  augmentation: randomcrop+flip
  size: 256
train_data:
  images: /data/train
  masks: /data/train
  labels: [1, 2, 3, 4, 5, 6, 7]
validation_data:
  images: /data/val
  masks: /data/val
  labels: [1, 2, 3, 4, 5, 6, 7]
model_params:
  gen_filts: 32
  disc_filts: 16
  activation: relu
  use_dropout: True
  final_activation: sigmoid
  n_disc_layers: 5
checkpoint_path: ./checkpoints/checkpoint-COCO/
load_last_checkpoint: True
train_params:
  loss_type: weighted_bce
  seg_alpha: 200
  gen_learning_rate: 1.e-3
  disc_learning_rate: 1.e-3
  decay_rate: 0.95
  save_freq: 5
"""

NESTED_YAML = """
dataset:
  type: MyData
  size: 256
  in_channels: 1
  out_channels: 2
  data: {images: a, masks: b}
  train_val_split: [0.8, 0.2]
model_params:
  generator: {filters: 8, activation: leakyrelu}
  discriminator: {filters: 4, n_layers: 3, norm: true}
train_params: {loss_type: tversky, seg_alpha: 200, gen_learning_rate: 1.e-3, disc_learning_rate: 2.e-3}
"""


def test_parse_config_flat_legacy_schema():
    """The reference's own example (examples/train_coco.yaml values) uses the flat schema that train.py v0.2.2 rejects;
    both are accepted here."""
    from patchgan_amd.train import parse_config
    ds, tr, va, split, g, d = parse_config(yaml.safe_load(REF_YAML))
    assert tr['images'] == '/data/train' and va['masks'] == '/data/val' and split is None
    assert ds['labels'] == [1, 2, 3, 4, 5, 6, 7]
    assert g == {'filters': 32, 'activation': 'relu', 'use_dropout': True, 'final_activation': 'sigmoid'}
    assert d['filters'] == 16 and d['n_layers'] == 5 and d['norm'] is False


def test_parse_config_nested_schema_and_errors():
    from patchgan_amd.train import parse_config
    ds, tr, va, split, g, d = parse_config(yaml.safe_load(NESTED_YAML))
    assert split == [0.8, 0.2] and va is None and tr == {'images': 'a', 'masks': 'b'}
    assert g['filters'] == 8 and d['norm'] is True
    with pytest.raises(AttributeError):
        parse_config({'dataset': {'type': 'X'}, 'model_params': {}})


@pytest.mark.parametrize('tag', ['sq1024', 'sq600', 'sq300_1c'])
def test_tiling_matches_reference_goldens(tag):
    """n_crop / build_mask against vectors produced by the reference's own functions (square images, where the
    reference's tile indexing is well defined) and against the oracle restatement."""
    from patchgan_amd.infer import n_crop, build_mask
    z = np.load(os.path.join(GOLDEN_DIR, 'infer_tiles.npz'))
    c, h, w, size, overlap, thr = z[f'{tag}/params']
    c, h, w, size = int(c), int(h), int(w), int(size)
    g = torch.Generator().manual_seed(3)
    for t in ['sq1024', 'sq600', 'sq300_1c']:     # replay the generator stream of make_golden.run_infer_tiles
        cc, hh, ww, ss, ov, th = z[f'{t}/params']
        img = torch.rand(int(cc), int(hh), int(ww), generator=g)
        ncrops = int(z[f'{t}/ncrops'][0])
        masks = torch.rand(ncrops, int(cc), int(ss), int(ss), generator=g)
        if t == tag:
            break
    crops = n_crop(img, size, overlap)
    assert tuple(crops.shape) == tuple(z[f'{tag}/ncrops'])
    np.testing.assert_allclose(probe(crops), z[f'{tag}/crop_probe'], rtol=1e-7)
    assert torch.equal(crops, O.n_crop(img, size, overlap))
    m = build_mask(masks, size, (h, w), thr, overlap)
    assert tuple(m.shape) == tuple(z[f'{tag}/mask_shape'])
    np.testing.assert_allclose(probe(torch.as_tensor(np.ascontiguousarray(m))), z[f'{tag}/mask_probe'], rtol=1e-12)
    np.testing.assert_array_equal(m, O.build_mask(masks.numpy(), size, (h, w), thr, overlap))


def test_tiling_non_square_covers_every_pixel():
    """Where the reference's index (j*ncropsy+i) breaks (non-square: tiles left zero) ours stays exact."""
    from patchgan_amd.infer import n_crop, build_mask
    img = torch.rand(2, 600, 1024)
    crops = n_crop(img, 256, 0.9)
    assert crops.shape[0] == 3 * 5
    back = build_mask(crops, 256, (600, 1024), 0, 0.9)      # identity "prediction": reconstruct channel argmax
    np.testing.assert_array_equal(back, np.argmax(img.numpy(), axis=0))


def test_coco_dataset_and_plugin_loader(tmp_path):
    from PIL import Image
    from patchgan_amd.io import COCOStuffDataset, load_plugin_dataset
    rng = np.random.default_rng(0)
    for i in (7, 12):
        Image.fromarray(rng.integers(0, 255, (40, 50, 3), dtype=np.uint8)).save(tmp_path / f'{i:06d}.jpg')
        Image.fromarray(rng.integers(0, 4, (40, 50), dtype=np.uint8)).save(tmp_path / f'{i:06d}.png')
    ds = COCOStuffDataset(str(tmp_path), str(tmp_path), labels=[1, 3], size=32, augmentation='randomcrop')
    img, mask = ds[1]
    assert img.shape == (3, 32, 32) and mask.shape == (2, 32, 32) and img.dtype == torch.float32
    assert 0 <= img.min() and img.max() <= 1 and set(mask.unique().tolist()) <= {0.0, 1.0}
    (tmp_path / 'io.py').write_text("class Toy:\n    def __init__(self, a, b, size=1, augmentation=None):\n        self.n = 3\n")
    Toy = load_plugin_dataset('Toy', str(tmp_path / 'io.py'))
    assert Toy('a', 'b').n == 3
    with pytest.raises(ImportError):
        load_plugin_dataset('Missing', str(tmp_path / 'io.py'))
    with pytest.raises(FileNotFoundError):
        load_plugin_dataset('Toy', str(tmp_path / 'nope.py'))


def test_coco_dataset_device_pipeline_bytes(tmp_path):
    """f3: device_pipeline=True hands over the decoded bytes; converting them the way the reference's __getitem__ does
    (io.py:42-56: / 255., uint8 + 1 with 255 -> 0, one-hot) gives exactly the float item of the default mode."""
    from PIL import Image
    from patchgan_amd.io import COCOStuffDataset
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 4, (32, 32), dtype=np.uint8)
    lab[0, :5] = 255                                              # COCO-stuff "unlabeled": + 1 wraps to 0
    Image.fromarray(rng.integers(0, 255, (32, 32, 3), dtype=np.uint8)).save(tmp_path / '000001.jpg')
    Image.fromarray(lab).save(tmp_path / '000001.png')
    kw = dict(labels=[0, 1, 3], size=32, augmentation='resize')
    img_f, mask_f = COCOStuffDataset(str(tmp_path), str(tmp_path), **kw)[0]
    img_b, lab_b = COCOStuffDataset(str(tmp_path), str(tmp_path), device_pipeline=True, **kw)[0]
    assert img_b.dtype == torch.uint8 and img_b.shape == (32, 32, 3) and lab_b.dtype == torch.uint8 and lab_b.shape == (32, 32)
    assert torch.equal(img_b.permute(2, 0, 1).float() / 255., img_f)
    assert torch.equal(torch.stack([((lab_b + 1) == v).float() for v in (0, 1, 3)]), mask_f)
    assert mask_f[0, 0, :5].sum() == 5                            # label 0 selects the wrapped 255s, as in the reference
    with pytest.raises(NotImplementedError):
        COCOStuffDataset(str(tmp_path), str(tmp_path), labels=[1], size=32, augmentation='randomcrop', device_pipeline=True)


def test_losses_module_matches_oracle():
    from patchgan_amd import losses
    g = torch.Generator().manual_seed(0)
    p = torch.rand(2, 3, 8, 8, generator=g)
    y = (torch.rand(2, 3, 8, 8, generator=g) > 0.5).float()
    assert torch.allclose(losses.fc_tversky(y, p, 0.75, 0.75), O.fc_tversky(y, p, 0.75, 0.75))
    assert torch.allclose(losses.tversky(y, p, 0.75), O.tversky(y, p, 0.75))
    assert torch.allclose(losses.MAE_loss(y, p), O.mae_loss(y, p))
    assert torch.allclose(losses.bce_loss(p, y), O.bce(p, y))
