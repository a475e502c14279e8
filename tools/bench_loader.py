#!/usr/bin/env python3
"""Loader throughput next to what the G+D step consumes (reference io.py:38-58, train.py:77-83).

Writes a synthetic COCO-stuff-like folder (JPEG images + PNG label maps of the given size), then times
``COCOStuffDataset`` -> ``DataLoader(shuffle, pin_memory, num_workers, persistent_workers)`` -> H2D copy for both item
formats -- float (the reference's: img float32 [3,S,S] + one-hot mask float32 [C,S,S]) and decoded bytes
(``device_pipeline=True``: uint8 image + uint8 label map, `/255.` and one-hot on the GPU) -- over a range of worker counts,
and prints one JSON line:  images/s per (format, workers), H2D bytes per image, the host cores available, and the
step's consumption for comparison (pass --step-images-per-sec, default the round-1 driver figure 1478).

    python tools/bench_loader.py [--size 256] [--files 512] [--batch 16] [--workers 0,4,8,16] [--seconds 6]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_files(folder, n, size, seed=0):
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(folder, 'img'))
    os.makedirs(os.path.join(folder, 'mask'))
    # smooth-ish content so that JPEG / PNG sizes resemble photographs / label maps rather than noise
    base = rng.integers(0, 256, (n, size // 8, size // 8, 3), dtype=np.uint8)
    lab = rng.integers(0, 12, (n, size // 16, size // 16), dtype=np.uint8)
    for i in range(n):
        im = Image.fromarray(base[i]).resize((size, size), Image.BICUBIC)
        im.save(os.path.join(folder, 'img', f'{i:012d}.jpg'), quality=90)
        Image.fromarray(lab[i]).resize((size, size), Image.NEAREST).save(os.path.join(folder, 'mask', f'{i:012d}.png'))


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--files', type=int, default=256)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--workers', default='0,4,8')
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--labels', default='1,2,3,4')
    ap.add_argument('--step-images-per-sec', type=float, default=1478.0)
    args = ap.parse_args()
    import torch
    from torch.utils.data import DataLoader
    from patchgan_amd.io import COCOStuffDataset
    labels = [int(v) for v in args.labels.split(',')]
    torch.set_num_threads(min(usable_cpus(), 16))      # the GPU box shows 256 logical CPUs under a 16-CPU quota: do not oversubscribe
    dev = torch.device('cuda') if torch.cuda.is_available() else None
    out = {'size': args.size, 'batch': args.batch, 'labels': len(labels), 'host_cpus': usable_cpus(), 'h2d': dev is not None,
           'step_images_per_sec': args.step_images_per_sec, 'results': []}
    with tempfile.TemporaryDirectory(prefix='pg_loader_') as folder:
        make_files(folder, args.files, args.size)
        for fmt in ('float', 'u8'):
            for nw in [int(v) for v in args.workers.split(',')]:
                ds = COCOStuffDataset(os.path.join(folder, 'img'), os.path.join(folder, 'mask'), labels=labels, size=args.size,
                                      augmentation='resize', device_pipeline=(fmt == 'u8'))
                # a COCO-sized epoch (the synthetic files repeated): the timed window never crosses an epoch restart
                ds = torch.utils.data.ConcatDataset([ds] * max(1, 65536 // len(ds)))
                kw = dict(num_workers=nw, persistent_workers=True, prefetch_factor=4) if nw else {}
                dl = DataLoader(ds, batch_size=args.batch, shuffle=True, pin_memory=dev is not None, drop_last=True, **kw)
                n_img, n_bytes, t0 = 0, 0, None
                done = False
                while not done:
                    for x, y in dl:
                        if t0 is None:                  # first batch = worker start-up: not timed
                            t0 = time.perf_counter()
                            continue
                        if dev is not None:
                            x, y = x.to(dev, non_blocking=True), y.to(dev, non_blocking=True)
                        n_img += x.shape[0]
                        n_bytes += x.numel() * x.element_size() + y.numel() * y.element_size()
                        if time.perf_counter() - t0 > args.seconds:
                            done = True
                            break
                if dev is not None:
                    torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                del dl
                r = {'format': fmt, 'workers': nw, 'images_per_sec': round(n_img / dt, 1),
                     'h2d_MB_per_image': round(n_bytes / max(n_img, 1) / 1e6, 3),
                     'covers_step': n_img / dt >= args.step_images_per_sec}
                out['results'].append(r)
                print(r, file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
