"""patchgan_train end to end on the GPU: YAML + ./io.py plugin -> 2 epochs -> checkpoints -> resume -> patchgan_infer."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

PLUGIN = '''
import os
import numpy as np
import torch
from torch.utils.data import Dataset


class Blobs(Dataset):
    """synthetic segmentation set: bright squares on noise; train ctor (images, masks, size=, augmentation=)"""
    def __init__(self, images, masks=None, size=256, augmentation=None, **kw):
        self.n = int(images) if str(images).isdigit() else 2
        self.size = size

    def __len__(self):
        return self.n

    def _item(self, i):
        g = torch.Generator().manual_seed(i)
        img = torch.rand(3, self.size, self.size, generator=g) * 0.3
        mask = torch.zeros(1, self.size, self.size)
        a = 32 + 16 * (i % 5)
        img[:, a:a + 64, a:a + 64] += 0.6
        mask[:, a:a + 64, a:a + 64] = 1
        return img, mask

    def __getitem__(self, i):
        if i >= self.n:
            raise IndexError
        img, mask = self._item(i)
        return (img, mask) if self.pair else img

    pair = True

    def get_filename(self, i):
        return f'blob_{i:03d}.png'

    @staticmethod
    def save_mask(mask, out_dir, fname):
        np.save(os.path.join(out_dir, fname + '.npy'), mask)


class BlobsInfer(Blobs):
    pair = False
'''


def test_train_resume_infer(tmp_path, monkeypatch, capsys):
    from patchgan_amd.train import patchgan_train
    from patchgan_amd.infer import patchgan_infer
    monkeypatch.chdir(tmp_path)
    (tmp_path / 'io.py').write_text(PLUGIN)
    cfg = {
        'dataset': {'type': 'Blobs', 'size': 256, 'in_channels': 3, 'out_channels': 1,
                    'train_data': {'images': '6', 'masks': ''}, 'validation_data': {'images': '2', 'masks': ''}},
        'model_params': {'generator': {'filters': 4, 'activation': 'leakyrelu', 'use_dropout': True},
                         'discriminator': {'filters': 4, 'n_layers': 3}},
        'checkpoint_path': str(tmp_path / 'ckpt'),
        'train_params': {'loss_type': 'tversky', 'seg_alpha': 200, 'gen_learning_rate': 1e-3, 'disc_learning_rate': 1e-3,
                         'decay_rate': 0.9, 'save_freq': 1},
    }
    (tmp_path / 'cfg.yaml').write_text(yaml.safe_dump(cfg))
    G_ep, D_ep = patchgan_train(['-c', 'cfg.yaml', '-n', '2', '-b', '2', '--dataloader_workers', '0'])
    assert len(G_ep) == 2 and all(np.isfinite(G_ep)) and all(np.isfinite(D_ep))
    assert G_ep[1] < G_ep[0] * 1.05
    files = sorted(os.listdir(tmp_path / 'ckpt'))
    assert files == ['discriminator_ep_001.pth', 'discriminator_ep_002.pth', 'generator_ep_001.pth', 'generator_ep_002.pth']
    sd = torch.load(tmp_path / 'ckpt' / 'generator_ep_002.pth')
    assert sd['encoder.0.model.DownConv0.weight'].shape == (4, 3, 4, 4) and sd['decoder.6.model.UpConv6.weight'].shape == (8, 1, 4, 4)
    # resume: load_last_checkpoint -> epoch 3 only
    cfg['load_last_checkpoint'] = True
    (tmp_path / 'cfg.yaml').write_text(yaml.safe_dump(cfg))
    G2, _ = patchgan_train(['-c', 'cfg.yaml', '-n', '3', '-b', '2', '--dataloader_workers', '0'])
    assert len(G2) == 1
    out = capsys.readouterr().out
    # resumed LR = lr * decay^((start-1)/decay_freq) with the CLI's default decay_freq = 5 (reference trainer.py:155-157)
    assert f'Epoch 3 -- lr: {1e-3 * 0.9 ** (2 / 5):5.3e}' in out
    # inference with the flat legacy schema infer.py reads
    icfg = {'dataset': {'type': 'BlobsInfer', 'dataset_path': '2', 'size': 256},
            'model_params': {'gen_filts': 4, 'disc_filts': 4, 'n_disc_layers': 3, 'activation': 'leakyrelu'},
            'checkpoint_paths': {'generator': str(tmp_path / 'ckpt' / 'generator_ep_003.pth'),
                                 'discriminator': str(tmp_path / 'ckpt' / 'discriminator_ep_003.pth')},
            'infer_params': {'output_path': str(tmp_path / 'pred'), 'threshold': 0.5}}
    (tmp_path / 'icfg.yaml').write_text(yaml.safe_dump(icfg))
    patchgan_infer(['-c', 'icfg.yaml'])
    m = np.load(tmp_path / 'pred' / 'blob_000.npy')
    assert m.shape == (256, 256) and set(np.unique(m)) <= {0.0, 1.0}
