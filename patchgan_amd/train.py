"""``patchgan_train`` -- the reference's training CLI (patchgan/train.py:13-127) on the MI355X path.

Same flags (-c/--config_file, -b/--batch_size, --dataloader_workers, -n/--n_epochs, -d/--device, --summary) and the
same YAML keys.  Both config schemas are accepted: the one train.py v0.2.2 reads (``dataset.{train_data,validation_data}``
or ``dataset.{data,train_val_split}``, nested ``model_params.{generator,discriminator}``) and the older flat one that
``examples/train_coco.yaml`` and infer.py still use (top-level ``train_data`` / ``validation_data``,
``model_params.{gen_filts,disc_filts,n_disc_layers,activation,use_dropout,final_activation}``).

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N -m patchgan_amd.train ...``; each rank
takes a distinct shard of every epoch (DistributedSampler) and gradients are all-reduced over RCCL.
"""
import argparse
import os

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader, random_split

from .disc import Discriminator
from .io import COCOStuffDataset, load_plugin_dataset
from .trainer import Trainer
from .unet import UNet


def parse_config(config):
    """Normalise either YAML schema into (dataset_params, train_paths, val_paths, split, generator_cfg, disc_cfg)."""
    dataset_params = dict(config['dataset'])
    for key in ('train_data', 'validation_data', 'data', 'train_val_split'):      # legacy: paths at top level
        if key not in dataset_params and key in config:
            dataset_params[key] = config[key]
    if ('train_data' in dataset_params) and ('validation_data' in dataset_params):
        train_paths, val_paths, split = dataset_params['train_data'], dataset_params['validation_data'], None
    elif ('data' in dataset_params) and ('train_val_split' in dataset_params):
        train_paths, val_paths, split = dataset_params['data'], None, dataset_params['train_val_split']
    else:
        raise AttributeError("Please provide either the training and validation data paths or a train/val split!")
    if 'labels' not in dataset_params and isinstance(train_paths, dict) and 'labels' in train_paths:
        dataset_params['labels'] = train_paths['labels']                          # legacy: labels next to the paths
    mp = config['model_params']
    if 'generator' in mp:
        gen_cfg, disc_cfg = dict(mp['generator']), dict(mp['discriminator'])
    else:                                                                         # flat legacy schema (infer.py:127-132)
        gen_cfg = {'filters': mp['gen_filts'], 'activation': mp['activation'],
                   'use_dropout': mp.get('use_dropout', True), 'final_activation': mp.get('final_activation', 'sigmoid')}
        disc_cfg = {'filters': mp['disc_filts'], 'n_layers': mp['n_disc_layers'], 'norm': mp.get('disc_norm', False)}
    return dataset_params, train_paths, val_paths, split, gen_cfg, disc_cfg


def print_summary(name, module):
    n = sum(p.numel() for p in module.parameters())
    print(f"{name}: {n:,} parameters in {len(list(module.parameters()))} tensors")
    for k, v in module.state_dict().items():
        print(f"  {k:48s} {tuple(v.shape)}")


def patchgan_train(argv=None):
    parser = argparse.ArgumentParser(prog='PatchGAN', description='Train the PatchGAN architecture')
    parser.add_argument('-c', '--config_file', required=True, type=str, help='Location of the config YAML file')
    parser.add_argument('-b', '--batch_size', default=16, type=int, help='Number of images per batch (per GPU)')
    parser.add_argument('--dataloader_workers', default=4, type=int,
                        help='Number of workers to use with dataloader (set to 0 to disable multithreading)')
    parser.add_argument('-n', '--n_epochs', required=True, type=int, help='Number of epochs to train the model')
    parser.add_argument('-d', '--device', default='auto', help='Device to use to train the model (CUDA=GPU)')
    parser.add_argument('--summary', default=True, action='store_true', help="Print summary of the models")
    args = parser.parse_args(argv)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.device == 'cpu' or not torch.cuda.is_available():
        raise RuntimeError("patchgan_amd trains on a HIP device only (MI355X); no GPU is visible / --device cpu requested")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # bounded rendezvous / collective timeout: a rank whose peer died fails within minutes instead of holding its GPU for
        # the backend's 10-30 min default (torch.distributed.run then tears the job down)
        dist.init_process_group('nccl', device_id=device,
                                timeout=datetime.timedelta(seconds=float(os.environ.get('PATCHGAN_DIST_TIMEOUT_S', '600'))))

    with open(args.config_file, 'r') as infile:
        config = yaml.safe_load(infile)
    dataset_params, train_paths, val_paths, split, gen_cfg, disc_cfg = parse_config(config)

    size = dataset_params.get('size', 256)
    augmentation = dataset_params.get('augmentation', 'randomcrop')
    dataset_kwargs = {}
    if dataset_params['type'] == 'COCOStuff':
        Dataset = COCOStuffDataset
        in_channels = 3
        labels = dataset_params.get('labels', [1])
        out_channels = len(labels)
        dataset_kwargs['labels'] = labels
        if dataset_params.get('device_pipeline', False):      # extension: `/255.` + one-hot on the GPU (io.py docstring)
            dataset_kwargs['device_pipeline'] = True
    else:
        Dataset = load_plugin_dataset(dataset_params['type'])
        in_channels = dataset_params.get('in_channels', 3)
        out_channels = dataset_params.get('out_channels', 1)

    if split is None:
        train_datagen = Dataset(train_paths['images'], train_paths['masks'], size=size, augmentation=augmentation, **dataset_kwargs)
        val_datagen = Dataset(val_paths['images'], val_paths['masks'], size=size, augmentation=augmentation, **dataset_kwargs)
    else:
        datagen = Dataset(train_paths['images'], train_paths['masks'], size=size, augmentation=augmentation, **dataset_kwargs)
        train_datagen, val_datagen = random_split(datagen, split, generator=torch.Generator().manual_seed(0))

    dloader_kwargs = {}
    if args.dataloader_workers > 0:
        dloader_kwargs['num_workers'] = args.dataloader_workers
        dloader_kwargs['persistent_workers'] = True
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        train_data = DataLoader(train_datagen, batch_size=args.batch_size, pin_memory=True, drop_last=True,
                                sampler=DistributedSampler(train_datagen, shuffle=True), **dloader_kwargs)
        val_data = DataLoader(val_datagen, batch_size=args.batch_size, pin_memory=True, drop_last=True,
                              sampler=DistributedSampler(val_datagen, shuffle=False), **dloader_kwargs)
    else:
        train_data = DataLoader(train_datagen, batch_size=args.batch_size, shuffle=True, pin_memory=True, **dloader_kwargs)
        val_data = DataLoader(val_datagen, batch_size=args.batch_size, shuffle=True, pin_memory=True, **dloader_kwargs)

    generator = UNet(in_channels, out_channels, gen_cfg['filters'], use_dropout=gen_cfg.get('use_dropout', True),
                     activation=gen_cfg['activation'], final_act=gen_cfg.get('final_activation', 'sigmoid')).to(device)
    discriminator = Discriminator(in_channels + out_channels, disc_cfg['filters'], norm=disc_cfg.get('norm', False),
                                  n_layers=disc_cfg['n_layers']).to(device)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(generator.flat, 0)          # every rank starts from rank 0's initial weights
        dist.broadcast(discriminator.flat, 0)
    if args.summary and local_rank == 0:
        print_summary('generator', generator)
        print_summary('discriminator', discriminator)

    checkpoint_path = config.get('checkpoint_path', './checkpoints/')
    trainer = Trainer(generator, discriminator, savefolder=checkpoint_path)
    trainer.gc_freeze = True      # this process is the training job: keep generation-2 collections out of the step loop
    # per kind of step the trainer times two warm steps and decides: a launch-bound one is replayed from a captured hipGraph (one GPU,
    # use_dropout: False); a device-bound one stays launch by launch on two streams -- also with use_dropout: True (the default,
    # reference train.py:92), in the validation loop and under data parallelism
    trainer.graph = 'auto'
    if dataset_kwargs.get('device_pipeline', False):
        trainer.label_values = [int(v) for v in np.sort(dataset_kwargs['labels'])]
    if config.get('load_last_checkpoint', False):
        trainer.load_last_checkpoint()
    elif config.get('transfer_learn', {}).get('generator_checkpoint', None) is not None:
        gen_checkpoint = config['transfer_learn']['generator_checkpoint']
        dsc_checkpoint = config['transfer_learn']['discriminator_checkpoint']
        generator.load_transfer_data(torch.load(gen_checkpoint, map_location=device))
        discriminator.load_transfer_data(torch.load(dsc_checkpoint, map_location=device))

    train_params = config['train_params']
    trainer.loss_type = train_params['loss_type']
    trainer.seg_alpha = train_params['seg_alpha']
    return trainer.train(train_data, val_data, args.n_epochs,
                         dsc_learning_rate=train_params['disc_learning_rate'],
                         gen_learning_rate=train_params['gen_learning_rate'],
                         lr_decay=train_params.get('decay_rate', None),
                         save_freq=train_params.get('save_freq', 10))


if __name__ == '__main__':
    patchgan_train()
