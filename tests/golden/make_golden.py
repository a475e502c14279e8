#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING THE REFERENCE (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (/root/reference, ramanakumars/patchGAN v0.2.2) is imported, driven
on CPU with seeded inputs, and only DATA (inputs' seeds, initial weights, loss
curves, probes) is written to ``tests/golden/*.npz``.  No reference source or
bytecode is copied.  The GPU box has no /root/reference: tests there read the
committed ``.npz`` files only.
"""
import io
import os
import sys
import tempfile
import contextlib

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

# name, kwargs
CONFIGS = {
    # cfg2-shaped (leakyrelu/sigmoid/tversky/no-norm D) at nf=ndf=4
    'a_lrelu_tversky': dict(in_nc=3, out_nc=1, nf=4, ndf=4, n_layers=3, norm=False, activation='leakyrelu',
                            final_act='sigmoid', loss_type='tversky', B=2, size=256),
    # tanh + normed D + multi-class weighted BCE with sigmoid head
    'b_tanh_wbce_norm': dict(in_nc=3, out_nc=3, nf=4, ndf=4, n_layers=3, norm=True, activation='tanh',
                             final_act='sigmoid', loss_type='weighted_bce', B=2, size=256),
    # relu + MAE + 5-layer D (COCO-yaml-like: nf 2x ndf, n_layers 5)
    'c_relu_mae_l5': dict(in_nc=3, out_nc=1, nf=8, ndf=4, n_layers=5, norm=False, activation='relu',
                          final_act='sigmoid', loss_type='MAE', B=2, size=256),
    # softmax head, multi-class tversky, single-sample batch, normed 5-layer D
    'd_softmax_tversky': dict(in_nc=3, out_nc=4, nf=4, ndf=8, n_layers=5, norm=True, activation='leakyrelu',
                              final_act='softmax', loss_type='tversky', B=1, size=256),
    # single-class weighted_bce (weight = ones branch), 1-channel input
    'e_wbce_c1': dict(in_nc=1, out_nc=1, nf=4, ndf=4, n_layers=3, norm=False, activation='leakyrelu',
                      final_act='sigmoid', loss_type='weighted_bce', B=2, size=256),
}
# The BENCHMARK configurations at full width, driven through the reference itself (round 5).  Initial weights are NOT stored
# (41.8 M parameters): they are torch's default init under MODEL_SEED, which the build's modules reproduce bit for bit
# (tests/test_cabi_cpu.py::test_default_init_equals_torch_module_init); the weight probes 'g0/...' pin that claim per tensor.
WIDE_CONFIGS = {
    # cfg2 = BASELINE.json configs[1], the headline: 256x256x3 -> 1 mask, bs 16, nf = ndf = 64
    'w_cfg2': dict(in_nc=3, out_nc=1, nf=64, ndf=64, n_layers=3, norm=False, activation='leakyrelu',
                   final_act='sigmoid', loss_type='tversky', B=16, size=256, steps=10),
    # cfg1 = examples/train_coco.yaml:13-28 hyper-parameters at 256x256 (64x64 cannot run: SURVEY section 5), bs 4
    'w_cfg1': dict(in_nc=3, out_nc=7, nf=32, ndf=16, n_layers=5, norm=False, activation='relu',
                   final_act='sigmoid', loss_type='weighted_bce', B=4, size=256, steps=10),
    # cfg4's shape: 512x512x3 -> 4 classes (softmax head, weighted BCE over C > 1), nf = ndf = 64, B = 2 of the 8 per GPU
    'w_cfg4': dict(in_nc=3, out_nc=4, nf=64, ndf=64, n_layers=3, norm=False, activation='leakyrelu',
                   final_act='softmax', loss_type='weighted_bce', B=2, size=512, steps=4),
}
NSTEPS = 10
MODEL_SEED = 1234
DATA_SEED = 7
LR = 1e-3
NSAMP = 64


def make_inputs(cfg):
    """Same recipe as SURVEY 8(d): x ~ U[0,1), y = (U > 0.7)."""
    g = torch.Generator().manual_seed(DATA_SEED)
    x = torch.rand(cfg['B'], cfg['in_nc'], cfg['size'], cfg['size'], generator=g)
    y = (torch.rand(cfg['B'], cfg['out_nc'], cfg['size'], cfg['size'], generator=g) > 0.7).float()
    return x, y


def probe(t, nsamp=NSAMP):
    """(sum, abs-sum, strided samples) of a tensor, float64."""
    f = t.detach().double().flatten()
    n = f.numel()
    idx = (torch.arange(nsamp, dtype=torch.int64) * 2654435761 % n)
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def run_config(name, cfg, store_weights=True):
    cfg = dict(cfg)
    nsteps = cfg.pop('steps', NSTEPS)
    sys.path.insert(0, REF)
    from patchgan import UNet, Discriminator, Trainer
    torch.manual_seed(MODEL_SEED)
    g = UNet(cfg['in_nc'], cfg['out_nc'], cfg['nf'], use_dropout=False,
             activation=cfg['activation'], final_act=cfg['final_act'])
    d = Discriminator(cfg['in_nc'] + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=cfg['norm'])
    out = {}
    for k, v in g.state_dict().items():
        out['g0/' + k] = v.detach().numpy().copy() if store_weights else probe(v)
    for k, v in d.state_dict().items():
        out['d0/' + k] = v.detach().numpy().copy() if store_weights else probe(v)
    with contextlib.redirect_stdout(io.StringIO()):
        t = Trainer(g, d, tempfile.mkdtemp(), device='cpu')
    t.loss_type = cfg['loss_type']
    t.seg_alpha = 200
    t.gen_optimizer = torch.optim.Adam(g.parameters(), lr=LR, betas=(0.9, 0.999))
    t.disc_optimizer = torch.optim.Adam(d.parameters(), lr=LR, betas=(0.9, 0.999))
    x, y = make_inputs(cfg)

    # forward probes at the initial weights (train mode; dropout off)
    g.train()
    d.train()
    acts = {}
    hooks = []
    for i, blk in enumerate(g.encoder):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(f'enc{i}', o)))
    for i, blk in enumerate(g.decoder):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(f'dec{i}', o)))
    with torch.no_grad():
        gen0, hid0 = g(x, return_hidden=True)
        dfake0 = d(torch.cat((x, gen0), 1))
        dreal0 = d(torch.cat((x, y), 1))
    for h in hooks:
        h.remove()
    for k, v in acts.items():
        out['fwd/' + k] = probe(v)
    out['fwd/hidden'] = probe(hid0)
    out['fwd/disc_fake'] = probe(dfake0)
    out['fwd/disc_real'] = probe(dreal0)
    out['fwd/disc_shape'] = np.array(dfake0.shape)

    # eval-mode batch (train=False) at initial weights
    ev = t.batch(x, y, train=False)
    out['eval_losses'] = np.array([ev[k] for k in ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']])

    # 10 training steps; gradient probes after step 1
    curve = []
    for s in range(nsteps):
        l = t.batch(x, y, train=True)
        curve.append([l[k] for k in ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']])
        if s == 0:
            for k, p in g.named_parameters():
                out['ggrad1/' + k] = probe(p.grad)
            for k, p in d.named_parameters():
                out['dgrad1/' + k] = probe(p.grad)
            for k, v in g.state_dict().items():
                out['g1/' + k] = probe(v)
            for k, v in d.state_dict().items():
                out['d1/' + k] = probe(v)
    out['losses'] = np.array(curve, dtype=np.float64)
    for k, v in g.state_dict().items():
        out['g10/' + k] = probe(v)
    for k, v in d.state_dict().items():
        out['d10/' + k] = probe(v)
    with torch.no_grad():
        out['gen_img10'] = probe(g(x))
    out['cfg_keys'] = np.array(list(cfg.keys()))
    out['cfg_vals'] = np.array([str(v) for v in cfg.values()])
    out['meta'] = np.array([MODEL_SEED, DATA_SEED, nsteps, NSAMP])
    out['threads'] = np.array([torch.get_num_threads()])
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(name, 'loss[0]', curve[0][0], 'loss[-1]', curve[-1][0], flush=True)


def run_train_driver():
    """Trainer.train (trainer.py:117-279): LR schedule + per-epoch loss means + resume."""
    sys.path.insert(0, REF)
    from patchgan import UNet, Discriminator, Trainer
    cfg = CONFIGS['a_lrelu_tversky']
    x, y = make_inputs(cfg)
    data = [(x[:1], y[:1]), (x[1:], y[1:])]
    out = {}
    lrs = []
    for epochs in range(1, 7):
        torch.manual_seed(MODEL_SEED)
        g = UNet(3, 1, 4, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
        d = Discriminator(4, 4, n_layers=3)
        tmp = tempfile.mkdtemp()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            t = Trainer(g, d, tmp, device='cpu')
            G_ep, D_ep = t.train(data, data[:1], epochs, gen_learning_rate=1e-3, dsc_learning_rate=2e-3,
                                 lr_decay=0.9, decay_freq=2, save_freq=3)
        lrs.append([t.gen_optimizer.param_groups[0]['lr'], t.disc_optimizer.param_groups[0]['lr']])
        if epochs == 6:
            out['G_loss_ep'] = np.array(G_ep)
            out['D_loss_ep'] = np.array(D_ep)
            out['ckpt_files'] = np.array(sorted(os.listdir(tmp)))
            # resume: load_last_checkpoint -> start = 7; LR = lr*decay^((start-1)/decay_freq)
            torch.manual_seed(MODEL_SEED)
            g2 = UNet(3, 1, 4, use_dropout=False, activation='leakyrelu', final_act='sigmoid')
            d2 = Discriminator(4, 4, n_layers=3)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                t2 = Trainer(g2, d2, tmp, device='cpu')
                t2.load_last_checkpoint()
                G2, D2 = t2.train(data, data[:1], 7, gen_learning_rate=1e-3, dsc_learning_rate=2e-3,
                                  lr_decay=0.9, decay_freq=2, save_freq=3)
            out['resume_start'] = np.array([t2.start])
            out['resume_G_loss_ep'] = np.array(G2)
            out['resume_D_loss_ep'] = np.array(D2)
            out['resume_lr'] = np.array([t2.gen_optimizer.param_groups[0]['lr'],
                                         t2.disc_optimizer.param_groups[0]['lr']])
    out['lr_after_epochs'] = np.array(lrs)
    np.savez_compressed(os.path.join(HERE, 'train_driver.npz'), **out)
    print('train_driver', out['G_loss_ep'], out['lr_after_epochs'][:, 0], out['resume_start'], flush=True)


def run_infer_tiles():
    """n_crop / build_mask of the reference (infer.py:14-68).  patchgan.infer imports torchinfo and torchvision at module
    scope (absent here); the two functions are pure numpy/torch, so empty placeholder modules are registered for the
    import only."""
    import types
    for name in ('torchinfo', 'torchvision', 'torchvision.io', 'torchvision.transforms'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.summary = m.read_image = m.ImageReadMode = None
            sys.modules[name] = m
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].io = sys.modules['torchvision.io']
    sys.path.insert(0, REF)
    from patchgan.infer import n_crop, build_mask
    out = {}
    g = torch.Generator().manual_seed(3)
    for tag, (c, h, w, size, overlap, thr) in {'sq1024': (3, 1024, 1024, 256, 0.9, 0.0), 'sq600': (2, 600, 600, 256, 0.9, 0.4),
                                               'sq300_1c': (1, 300, 300, 256, 0.5, 0.5)}.items():
        img = torch.rand(c, h, w, generator=g)
        crops = n_crop(img, size, overlap)
        out[f'{tag}/ncrops'] = np.array(crops.shape)
        out[f'{tag}/crop_probe'] = probe(crops)
        masks = torch.rand(crops.shape[0], c, size, size, generator=g).numpy()
        m = build_mask(masks, size, (h, w), thr, overlap)
        out[f'{tag}/mask_shape'] = np.array(m.shape)
        out[f'{tag}/mask_probe'] = probe(torch.as_tensor(np.ascontiguousarray(m)))
        out[f'{tag}/params'] = np.array([c, h, w, size, overlap, thr], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'infer_tiles.npz'), **out)
    print('infer_tiles', {k: v.tolist() for k, v in out.items() if k.endswith('ncrops')}, flush=True)


def run_io_onehot():
    """The reference dataset's per-item arithmetic (io.py:38-58): `/ 255.`, the uint8 `+ 1` on the label map (255 wraps to 0)
    and the one-hot loop over np.sort(labels), driven on given decoded tensors.  patchgan.io imports torchvision (absent here) at
    module scope for read_image / transforms only: an empty placeholder module is registered for the import, and read_image is
    pointed at the tensors below, so everything that computes is the reference's own __getitem__."""
    import types
    for name in ('torchvision', 'torchvision.io', 'torchvision.transforms'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.read_image = m.ImageReadMode = None
            sys.modules[name] = m
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].io = sys.modules['torchvision.io']
    sys.path.insert(0, REF)
    import patchgan.io as rio
    g = torch.Generator().manual_seed(11)
    H, W = 40, 56
    img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 9, (1, H, W), dtype=torch.uint8, generator=g)
    lab[0, 0, :9] = 255                     # COCO-stuff "unlabeled": 255 + 1 wraps to 0 in uint8
    lab[0, 1, :5] = 254
    store = {'i.jpg': img, 'm.png': lab}

    class Mode:
        RGB, GRAY = 'RGB', 'GRAY'
    rio.ImageReadMode = Mode
    rio.read_image = lambda path, mode: store[path].clone()
    labels = [6, 0, 3, 255]                 # unsorted on purpose: the reference sorts them
    ds = object.__new__(rio.COCOStuffDataset)
    ds.images, ds.masks, ds.labels, ds.size, ds.augmentation = np.asarray(['i.jpg']), np.asarray(['m.png']), np.sort(labels), H, None
    x, y = ds[0]
    out = {'img_u8': img.numpy(), 'lab_u8': lab.numpy(), 'labels': np.asarray(labels), 'x': x.numpy(), 'y': y.numpy()}
    np.savez_compressed(os.path.join(HERE, 'io_onehot.npz'), **out)
    print('io_onehot', x.shape, y.shape, y.sum((1, 2)).tolist(), flush=True)


if __name__ == '__main__':
    torch.set_num_threads(8)
    which = sys.argv[1:] or (list(CONFIGS) + list(WIDE_CONFIGS) + ['train_driver', 'infer_tiles', 'io_onehot'])
    for name in which:
        if name == 'io_onehot':
            run_io_onehot()
        elif name == 'infer_tiles':
            run_infer_tiles()
        elif name == 'train_driver':
            run_train_driver()
        elif name in WIDE_CONFIGS:
            run_config(name, WIDE_CONFIGS[name], store_weights=False)
        else:
            run_config(name, CONFIGS[name])
