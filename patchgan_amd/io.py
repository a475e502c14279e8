"""Dataset side of the hot path: the reference's ``COCOStuffDataset`` (patchgan/io.py:10-58) and its plugin contract.

Plugin contract kept from the reference (train.py:58-75, infer.py:107-125,164-174): a class named by ``dataset.type``
in ``./io.py`` of the working directory, constructed as ``Dataset(images_path, masks_path, size=, augmentation=, **kw)``
for training (items: ``(img float32 [Cin,S,S], mask float32 [Cout,S,S])``) and ``Dataset(dataset_path, **kw)`` for
inference (items: ``[C,H,W]`` tensors, plus ``get_filename(i)`` and a static ``save_mask(mask, out_dir, fname)``).

torchvision is not a dependency here: images are decoded with Pillow and resized with torch.nn.functional.interpolate
(bilinear, no antialias -- what ``transforms.Resize(antialias=None)`` does on tensors).

``device_pipeline=True`` (an extension) makes the dataset hand over the decoded bytes -- image uint8 [S,S,3]
and label map uint8 [S,S] -- and leaves ``/ 255.`` and the one-hot mask to the GPU (``Trainer.batch`` with uint8 input,
``pg_u8_to_f32`` / ``pg_labels_to_onehot``): a quarter of the H2D bytes and no per-pixel Python work in the loader.
"""
import glob
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def _read_image(path, mode):
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert(mode))
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr.copy()).permute(2, 0, 1)


class COCOStuffDataset(Dataset):
    augmentation = None

    def __init__(self, imgfolder, maskfolder, labels=[1], size=256, augmentation='resize', device_pipeline=False):
        self.images = np.asarray(sorted(glob.glob(os.path.join(imgfolder, "*.jpg"))))
        self.masks = np.asarray(sorted(glob.glob(os.path.join(maskfolder, "*.png"))))
        self.size = size
        self.labels = np.sort(labels)
        self.image_ids = [int(os.path.basename(p).replace('.jpg', '')) for p in self.images]
        self.mask_ids = [int(os.path.basename(p).replace('.png', '')) for p in self.masks]
        assert np.all(self.image_ids == self.mask_ids), "Image IDs and Mask IDs do not match!"
        self.flip = 0.25 if augmentation == 'randomcrop+flip' else 0.0
        self.augmentation = augmentation if augmentation in ('randomcrop', 'randomcrop+flip') else None
        self.device_pipeline = bool(device_pipeline)
        if self.device_pipeline and self.augmentation is not None:
            raise NotImplementedError("device_pipeline hands over undecoded-size bytes: use augmentation='resize' (none)")
        print(f"Loaded {len(self)} images")

    def __len__(self):
        return len(self.images)

    def _augment(self, stacked):
        if self.augmentation is None:
            return stacked
        s = torch.nn.functional.interpolate(stacked[None], size=(self.size, self.size), mode='bilinear',
                                            align_corners=False, antialias=False)[0]
        if self.flip > 0:
            if torch.rand(1).item() < self.flip:
                s = s.flip(-1)
            if torch.rand(1).item() < self.flip:
                s = s.flip(-2)
        return s

    def __getitem__(self, index):
        if self.device_pipeline:
            return (_read_image(self.images[index], 'RGB').permute(1, 2, 0).contiguous(),
                    _read_image(self.masks[index], 'L')[0].contiguous())
        img = _read_image(self.images[index], 'RGB').float() / 255.
        labels = (_read_image(self.masks[index], 'L') + 1).float()      # uint8 + 1 like read_image(...) + 1: 255 wraps to 0
        stacked = self._augment(torch.cat((img, labels), dim=0))
        img, labels = stacked[:3], stacked[3]
        mask = torch.zeros((len(self.labels), labels.shape[0], labels.shape[1]))
        for i, label in enumerate(self.labels):
            mask[i, labels == label] = 1
        return img, mask


def load_plugin_dataset(type_name, path='io.py'):
    """Load class `type_name` from ./io.py of the working directory, as the reference does (train.py:58-60)."""
    import importlib.machinery
    import importlib.util
    try:
        loader = importlib.machinery.SourceFileLoader('io_plugin', path)
        spec = importlib.util.spec_from_loader('io_plugin', loader)
        module = importlib.util.module_from_spec(spec)
        loader.exec_module(module)
    except FileNotFoundError:
        print("Make sure io.py is in the working directory!")
        raise
    try:
        return getattr(module, type_name)
    except AttributeError:
        print(f"io.py does not contain {type_name}")
        raise ImportError(f"io.py does not contain {type_name}")
