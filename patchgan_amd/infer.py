"""``patchgan_infer`` -- tiled inference (reference patchgan/infer.py:14-174) with the generator forward on the HIP path.

``predict_image`` is what ``patchgan_infer`` runs per image: ``pg_tiles_gather`` (image -> NHWC tile batch), the generator
forward, ``pg_tiles_blend`` (overlap average in float64, optional threshold, class argmax) -- three stages of HIP kernels
with no layout conversion in between.  ``n_crop`` / ``build_mask`` keep the reference's function signatures (NCHW tensors in
and out, any device) for user code that calls them directly.  The reference indexes tiles with ``j * ncropsy + i``
(infer.py:32,57), which is only correct for square images (it overwrites / skips tiles otherwise); here the row-major
index ``j * ncropsx + i`` is used, identical for square images and correct for the rest.
"""
import argparse
import os

import numpy as np
import torch
import tqdm
import yaml

from .disc import Discriminator
from .io import COCOStuffDataset, load_plugin_dataset
from .unet import UNet


def _starts(extent, size, eff):
    n = int(np.ceil(extent / eff))
    out = []
    for k in range(n):
        s = k * eff
        s -= max(s + size - extent, 0)
        out.append(s)
    return out


def n_crop(image, size, overlap):
    c, height, width = image.shape
    eff = int(overlap * size)
    ys, xs = _starts(height, size, eff), _starts(width, size, eff)
    crops = torch.zeros((len(xs) * len(ys), c, size, size), device=image.device)
    for j, sy in enumerate(ys):
        for i, sx in enumerate(xs):
            crops[j * len(xs) + i] = image[:, sy:sy + size, sx:sx + size]
    return crops


def build_mask(masks, crop_size, image_size, threshold, overlap):
    masks = torch.as_tensor(masks)
    n, c, height, width = masks.shape
    ih, iw = image_size
    dev = masks.device
    mask = torch.zeros((c, ih, iw), dtype=torch.float64, device=dev)
    count = torch.zeros((c, ih, iw), dtype=torch.float64, device=dev)
    eff = int(overlap * crop_size)
    ys, xs = _starts(ih, crop_size, eff), _starts(iw, crop_size, eff)
    for j, sy in enumerate(ys):
        for i, sx in enumerate(xs):
            mask[:, sy:sy + crop_size, sx:sx + crop_size] += masks[j * len(xs) + i].double()
            count[:, sy:sy + crop_size, sx:sx + crop_size] += 1
    mask = mask / count
    if threshold > 0:
        mask = (mask >= threshold).double()
    mask = mask.cpu().numpy()
    if c > 1:
        return np.argmax(mask, axis=0)
    return mask[0]


def predict_image(generator, image, size, overlap, threshold, max_tiles=None):
    """One image [C, H, W] (device tensor) -> mask as numpy: float64 [H, W] for a single-class generator, class index
    [H, W] otherwise -- reference infer.py:155-163 (n_crop -> generator -> build_mask) on the HIP path end to end.
    The tiles STREAM through the generator at most `max_tiles` per forward pass (default: 64 tiles of 256 x 256, scaled with the
    tile area; tiles are independent samples, so the result does not depend on it): a 1024 x 1024 image (25 tiles) is one pass, a
    4096 x 4096 one (324 tiles) six -- activation memory and every tensor's byte extent stay bounded whatever the image size."""
    from . import engine as E
    eng = generator.engine
    tiles = E.tiles_gather(image, size, overlap)
    pred = E.View.alloc(tiles.N, size, size, eng.output_nc, image.device)
    if max_tiles is None:
        max_tiles = max(1, (64 * 256 * 256) // (size * size))
    for t0 in range(0, tiles.N, max_tiles):
        n = min(max_tiles, tiles.N - t0)
        eng.forward(generator.flat, tiles.samples(t0, n), pred.samples(t0, n), False, 0)
    mask = E.tiles_blend(pred, tuple(image.shape[1:]), threshold, overlap)
    # device -> host through a pinned block of torch's caching host allocator (the numpy array keeps it alive; it returns to the cache when
    # the caller drops the mask): a pageable .cpu() of the 8-MB float64 mask took 0.3-1.2 ms of the 3-ms image, this one 0.15
    host = torch.empty(mask.shape, dtype=mask.dtype, pin_memory=True)
    host.copy_(mask, non_blocking=True)
    torch.cuda.current_stream(mask.device).synchronize()
    return host.numpy()


def patchgan_infer(argv=None):
    parser = argparse.ArgumentParser(prog='PatchGAN', description='Run PatchGAN inference')
    parser.add_argument('-c', '--config_file', required=True, type=str, help='Location of the config YAML file')
    parser.add_argument('--dataloader_workers', default=4, type=int)
    parser.add_argument('-d', '--device', default='auto', help='Device to use (CUDA=GPU)')
    parser.add_argument('--summary', default=True, action='store_true', help="Print summary of the models")
    args = parser.parse_args(argv)
    if args.device == 'cpu' or not torch.cuda.is_available():
        raise RuntimeError("patchgan_amd runs on a HIP device only (MI355X)")
    device = 'cuda'
    print(f"Running with {device}")

    with open(args.config_file, 'r') as infile:
        config = yaml.safe_load(infile)
    dataset_params = config['dataset']
    dataset_path = dataset_params['dataset_path']
    size = dataset_params.get('size', 256)
    dataset_kwargs = {}
    if dataset_params['type'] == 'COCOStuff':
        Dataset = COCOStuffDataset
        in_channels = 3
        labels = dataset_params.get('labels', [1])
        out_channels = len(labels)
        dataset_kwargs['labels'] = labels
    else:
        Dataset = load_plugin_dataset(dataset_params['type'])
        in_channels = dataset_params.get('in_channels', 3)
        out_channels = dataset_params.get('out_channels', 1)
    assert hasattr(Dataset, 'get_filename') and callable(Dataset.get_filename), \
        f"Dataset class {Dataset.__name__} must have the get_filename method which returns the image filename for a given index"
    assert hasattr(Dataset, 'save_mask') and callable(Dataset.save_mask), \
        f"Dataset class {Dataset.__name__} must have the save_mask method to save a mask cube for a given filename"
    datagen = Dataset(dataset_path, **dataset_kwargs)

    mp = config['model_params']
    if 'generator' in mp:
        gen_filts, activation = mp['generator']['filters'], mp['generator']['activation']
        final_activation = mp['generator'].get('final_activation', 'sigmoid')
        disc_filts, n_disc_layers = mp['discriminator']['filters'], mp['discriminator']['n_layers']
    else:
        gen_filts, disc_filts, n_disc_layers = mp['gen_filts'], mp['disc_filts'], mp['n_disc_layers']
        activation, final_activation = mp['activation'], mp.get('final_activation', 'sigmoid')
    generator = UNet(in_channels, out_channels, gen_filts, activation=activation, final_act=final_activation).to(device)
    discriminator = Discriminator(in_channels + out_channels, disc_filts, n_layers=n_disc_layers).to(device)
    if args.summary:
        from .train import print_summary
        print_summary('generator', generator)

    ck = config['checkpoint_paths']
    infer_params = config.get('infer_params', {})
    output_path = infer_params.get('output_path', 'predictions/')
    if not os.path.exists(output_path):
        os.makedirs(output_path)
        print(f"Created folder {output_path}")
    generator.eval()
    discriminator.eval()
    generator.load_state_dict(torch.load(ck['generator'], map_location=device))
    discriminator.load_state_dict(torch.load(ck['discriminator'], map_location=device))
    threshold = infer_params.get('threshold', 0)
    overlap = infer_params.get('overlap', 0.9)

    for i, data in enumerate(tqdm.tqdm(datagen, desc='Predicting', dynamic_ncols=True, ascii=True)):
        data = torch.as_tensor(data).to(device)
        out_fname, _ = os.path.splitext(datagen.get_filename(i))
        mask = predict_image(generator, data, size, overlap, threshold)
        Dataset.save_mask(mask, output_path, out_fname)


if __name__ == '__main__':
    patchgan_infer()
