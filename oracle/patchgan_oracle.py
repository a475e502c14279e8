"""CPU oracle for the patchGAN G+D training step.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch-CPU fp32 ops, NCHW, functional --
no nn.Module, no torch.optim) of the reference's hot path.  It exists so that
the HIP path can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``patchgan_amd`` never does.

Parity status: PINNED.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned by fixtures generated in the build
container by importing the reference itself (``tests/golden/make_golden.py``
-> ``tests/golden/*.npz``) and checked by ``tests/test_oracle_golden.py``.

A second, independent statement of the bottom-level arithmetic (conv 4x4, conv-transpose 4x4, InstanceNorm, conv
weight gradient) lives in ``oracle/conv_ref.c`` (plain C loops, double accumulation; built by ``make -C oracle``) and
is held against this file by ``tests/test_oracle_c_cpu.py``.

Where the arithmetic lives: third-party PyTorch (reference ``setup.py:35``,
``torch>=1.13.0``, unpinned; this image has torch 2.10.0).  The reference's call
sites are cited per function below as ``file:line`` into /root/reference.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# activations (reference unet.py:12-17, 42-51; disc.py:20,29,39,46)
# ----------------------------------------------------------------------------


def apply_act(x, name):
    if name is None or name == 'none':
        return x
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'relu':
        return torch.relu(x)
    if name == 'leakyrelu':
        return F.leaky_relu(x, 0.2)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'softmax':
        return torch.softmax(x, dim=1)
    raise ValueError(f"unknown activation {name!r}")


def instance_norm(x):
    """nn.InstanceNorm2d defaults: eps 1e-5, biased variance, no affine, no
    running stats -> always instance statistics (reference unet.py:77,
    disc.py:8).  Raises ValueError at 1x1 like torch does."""
    if x.shape[2] * x.shape[3] <= 1:
        raise ValueError("Expected more than 1 spatial element when training, got input size "
                         f"{tuple(x.shape)}")
    return F.instance_norm(x, eps=1e-5)


# ----------------------------------------------------------------------------
# model plans
# ----------------------------------------------------------------------------


def unet_filters(nf):
    """reference unet.py:84"""
    return [nf, nf * 2, nf * 4, nf * 8, nf * 8, nf * 8, nf * 8]


def unet_weight_shapes(input_nc, output_nc, nf):
    """state_dict keys/shapes of the reference UNet (unet.py:88-107):
    Conv2d weight [Cout,Cin,4,4]; ConvTranspose2d weight [Cin,Cout,4,4]; no biases."""
    filts = unet_filters(nf)
    shapes = {}
    prev = input_nc
    for i, f in enumerate(filts):
        shapes[f'encoder.{i}.model.DownConv{i}.weight'] = (f, prev, 4, 4)
        prev = f
    for i, f in enumerate(filts[:-1][::-1]):
        cin = prev if i == 0 else prev * 2
        shapes[f'decoder.{i}.model.UpConv{i}.weight'] = (cin, f, 4, 4)
        prev = f
    shapes['decoder.6.model.UpConv6.weight'] = (nf * 2, output_nc, 4, 4)
    return shapes


def disc_plan(input_nc, ndf, n_layers, norm):
    """Layer list of the reference Discriminator (disc.py:19-46).
    Returns [(key_index, cin, cout, stride, has_bias, act, has_norm)]; key_index
    is the nn.Sequential index of the conv (state_dict key ``model.{idx}``)."""
    plan = []
    idx = 0
    plan.append((idx, input_nc, ndf, 2, True, 'leakyrelu', False))
    idx += 2
    mult = 1
    for n in range(1, n_layers):
        prev, mult = mult, min(2 ** n, 8)
        plan.append((idx, ndf * prev, ndf * mult, 2, False, 'tanh', norm))
        idx += 3 if norm else 2
    prev, mult = mult, min(2 ** n_layers, 8)
    plan.append((idx, ndf * prev, ndf * mult, 1, False, 'tanh', norm))
    idx += 3 if norm else 2
    plan.append((idx, ndf * mult, 1, 1, True, 'sigmoid', False))
    return plan


def disc_weight_shapes(input_nc, ndf, n_layers, norm):
    shapes = {}
    for (idx, cin, cout, _s, bias, _a, _n) in disc_plan(input_nc, ndf, n_layers, norm):
        shapes[f'model.{idx}.weight'] = (cout, cin, 4, 4)
        if bias:
            shapes[f'model.{idx}.bias'] = (cout,)
    return shapes


def default_init(shapes, generator=None):
    """torch default init for Conv2d / ConvTranspose2d (the reference's
    ``weights_init`` is a no-op, trainer.py:327-343): kaiming_uniform_(a=sqrt(5))
    == U(+-1/sqrt(fan_in)) with fan_in = shape[1]*16; bias U(+-1/sqrt(fan_in)) where
    fan_in is that of the layer's weight.  NOT bit-identical to nn.Module
    construction order; used only where a fixture supplies no weights."""
    out = {}
    fan = {}
    for k, shp in shapes.items():
        if k.endswith('.weight'):
            bound = 1.0 / math.sqrt(shp[1] * 16)
            fan[k[:-7]] = bound
            out[k] = (torch.rand(shp, generator=generator) * 2 - 1) * bound
    for k, shp in shapes.items():
        if k.endswith('.bias'):
            out[k] = (torch.rand(shp, generator=generator) * 2 - 1) * fan[k[:-5]]
    return out


# ----------------------------------------------------------------------------
# forward passes
# ----------------------------------------------------------------------------


def unet_forward(w, x, activation='tanh', final_act='softmax', dropout_masks=None,
                 return_hidden=False, probes=None):
    """reference unet.py:112-134 (+ blocks unet.py:8-72).

    ``dropout_masks``: optional dict {'enc{i}'|'dec{i}': keep-mask tensor of the
    block's output shape}; where given, output = act_out * mask / 0.8
    (nn.Dropout(0.2) training semantics, unet.py:28,65).  None = dropout off."""
    dm = dropout_masks or {}
    skips = []
    h = x
    for i in range(7):
        h = F.conv2d(h, w[f'encoder.{i}.model.DownConv{i}.weight'], None, stride=2, padding=1)  # unet.py:19
        h = instance_norm(h)                                                                    # unet.py:20
        h = apply_act(h, activation)
        if f'enc{i}' in dm:
            h = h * dm[f'enc{i}'] / 0.8
        if probes is not None:
            probes[f'enc{i}'] = h
        skips.append(h)
    hidden = skips[-1]
    skips = skips[::-1]                                                                         # unet.py:121
    for i in range(7):
        inp = hidden if i == 0 else torch.cat([h, skips[i]], dim=1)                             # unet.py:127
        h = F.conv_transpose2d(inp, w[f'decoder.{i}.model.UpConv{i}.weight'], None, stride=2, padding=1)
        if 1 <= i <= 5:                                                                         # unet.py:100-102
            h = instance_norm(h)
        h = apply_act(h, final_act if i == 6 else activation)
        if f'dec{i}' in dm:
            h = h * dm[f'dec{i}'] / 0.8
        if probes is not None:
            probes[f'dec{i}'] = h
    if return_hidden:
        return h, hidden
    return h


def disc_forward(w, x, n_layers=3, norm=False, probes=None):
    """reference disc.py:19-51: Conv(+bias) -> LeakyReLU; [Conv -> Tanh -> (IN)] x; Conv(+bias) -> Sigmoid."""
    input_nc = x.shape[1]
    ndf = w['model.0.weight'].shape[0]
    h = x
    for li, (idx, _cin, _cout, stride, bias, act, has_norm) in enumerate(disc_plan(input_nc, ndf, n_layers, norm)):
        b = w[f'model.{idx}.bias'] if bias else None
        h = F.conv2d(h, w[f'model.{idx}.weight'], b, stride=stride, padding=1)
        h = apply_act(h, act)
        if has_norm:
            h = instance_norm(h)
        if probes is not None:
            probes[f'd{li}'] = h
    return h


# ----------------------------------------------------------------------------
# losses (reference losses.py, trainer.py:71-85,101-103)
# ----------------------------------------------------------------------------


def fc_tversky(y_true, y_pred, beta, gamma=0.75):
    """reference losses.py:18-31 (batch_mean=True)"""
    tp = torch.sum(y_true * y_pred, dim=(1, 2, 3))
    fn = torch.sum((1. - y_pred) * y_true, dim=(1, 2, 3))
    fp = torch.sum(y_pred * (1. - y_true), dim=(1, 2, 3))
    tv = (tp + 1) / (tp + beta * fn + (1. - beta) * fp + 1)
    return torch.pow(torch.mean(1 - tv), gamma)


def tversky(y_true, y_pred, beta):
    """reference losses.py:5-15 (batch_mean=True; not used by the Trainer)"""
    tp = torch.sum(y_true * y_pred, dim=(1, 2, 3))
    fn = torch.sum((1. - y_pred) * y_true, dim=(1, 2, 3))
    fp = torch.sum(y_pred * (1. - y_true), dim=(1, 2, 3))
    return torch.mean(1. - tp / (tp + beta * fn + (1. - beta) * fp))


def mae_loss(a, b):
    """reference losses.py:34-35"""
    return torch.mean(torch.abs(a - b))


def bce(p, t, weight=None):
    """nn.BCELoss() / F.binary_cross_entropy, reduction='mean', log clamped at -100
    (reference losses.py:39, trainer.py:80)."""
    return F.binary_cross_entropy(p, t, weight=weight)


def seg_loss(loss_type, gen_img, target, seg_alpha=200, beta=0.75, gamma=0.75):
    """reference trainer.py:71-82"""
    if loss_type == 'tversky':
        return fc_tversky(target, gen_img, beta=beta, gamma=gamma) * seg_alpha
    if loss_type == 'weighted_bce':
        if gen_img.shape[1] > 1:
            weight = 1 - torch.sum(target, dim=(2, 3), keepdim=True) / torch.sum(target)
        else:
            weight = torch.ones_like(target)
        return bce(gen_img, target, weight=weight) * seg_alpha
    if loss_type == 'MAE':
        return mae_loss(gen_img, target) * seg_alpha
    raise ValueError(loss_type)


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: eps 1e-8, wd 0, no amsgrad; trainer.py:169-172)
# ----------------------------------------------------------------------------


def adam_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One in-place Adam update, restating torch's single-tensor CPU path:
    m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class OracleTrainer:
    """Functional restatement of reference ``Trainer.batch`` (trainer.py:50-115)
    plus the Adam set-up of ``Trainer.train`` (trainer.py:169-172)."""

    keys = ['gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc']

    def __init__(self, gw, dw, activation='tanh', final_act='sigmoid', n_layers=3, norm=False,
                 loss_type='tversky', seg_alpha=200, gen_lr=1e-3, dsc_lr=1e-3,
                 tversky_beta=0.75, tversky_gamma=0.75, dtype=torch.float32):
        # dtype=torch.float64 runs the same algorithm in double: the yardstick for fp32 rounding noise
        self.dtype = dtype
        self.gw = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in gw.items()}
        self.dw = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in dw.items()}
        self.activation, self.final_act = activation, final_act
        self.n_layers, self.norm = n_layers, norm
        self.loss_type, self.seg_alpha = loss_type, seg_alpha
        self.beta, self.gamma = tversky_beta, tversky_gamma
        self.gen_lr, self.dsc_lr = gen_lr, dsc_lr
        self.t_g = 0
        self.t_d = 0
        self.gm = {k: torch.zeros_like(v) for k, v in self.gw.items()}
        self.gv = {k: torch.zeros_like(v) for k, v in self.gw.items()}
        self.dm = {k: torch.zeros_like(v) for k, v in self.dw.items()}
        self.dv = {k: torch.zeros_like(v) for k, v in self.dw.items()}
        self.last = {}

    def G(self, x, dropout_masks=None, probes=None):
        return unet_forward(self.gw, x, self.activation, self.final_act, dropout_masks, probes=probes)

    def D(self, x, probes=None):
        return disc_forward(self.dw, x, self.n_layers, self.norm, probes=probes)

    def batch(self, x, y, train=False, dropout_masks=None):
        x = x.to(self.dtype)
        y = y.to(self.dtype)
        gen_img = self.G(x, dropout_masks)                                      # trainer.py:63
        disc_fake = self.D(torch.cat((x, gen_img), 1))                          # trainer.py:65-66
        ones = torch.ones_like(disc_fake)
        zeros = torch.zeros_like(disc_fake)
        gen_loss_seg = seg_loss(self.loss_type, gen_img, y, self.seg_alpha, self.beta, self.gamma)
        gen_loss_disc = bce(disc_fake, ones)                                    # trainer.py:84
        gen_loss = gen_loss_seg + gen_loss_disc
        if train:
            # generator.zero_grad(); backward; Adam (trainer.py:87-90).  D's grads from
            # this backward are discarded by discriminator.zero_grad() (trainer.py:94).
            names = list(self.gw)
            grads = torch.autograd.grad(gen_loss, [self.gw[k] for k in names])
            self.last['g_grads'] = dict(zip(names, grads))
            self.t_g += 1
            with torch.no_grad():
                for k, g in zip(names, grads):
                    adam_update(self.gw[k], g, self.gm[k], self.gv[k], self.t_g, self.gen_lr)
        disc_real = self.D(torch.cat((x, y), 1))                                # trainer.py:96-97
        disc_fake2 = self.D(torch.cat((x, gen_img.detach()), 1))                # trainer.py:98-99
        loss_real = bce(disc_real, ones)
        loss_fake = bce(disc_fake2, zeros)
        disc_loss = (loss_fake + loss_real) / 2.                                # trainer.py:103
        if train:
            names = list(self.dw)
            grads = torch.autograd.grad(disc_loss, [self.dw[k] for k in names])
            self.last['d_grads'] = dict(zip(names, grads))
            self.t_d += 1
            with torch.no_grad():
                for k, g in zip(names, grads):
                    adam_update(self.dw[k], g, self.dm[k], self.dv[k], self.t_d, self.dsc_lr)
        self.last['gen_img'] = gen_img.detach()
        vals = [gen_loss.item(), gen_loss.item(), gen_loss_disc.item(),
                loss_real.item(), loss_fake.item(), disc_loss.item()]
        return dict(zip(self.keys, vals))


# ----------------------------------------------------------------------------
# learning-rate schedule (trainer.py:155-160,180-181,266-270)
# ----------------------------------------------------------------------------


def resume_lr(lr, lr_decay, start, decay_freq):
    """reference trainer.py:155-157"""
    return lr * (lr_decay) ** ((start - 1) / decay_freq)


def exponential_lr_sequence(lr, lr_decay, epochs, decay_freq=5, start=1):
    """LR printed at the top of each epoch by the reference (trainer.py:192-200,266-270)."""
    cur = resume_lr(lr, lr_decay, start, decay_freq)
    seq = []
    for epoch in range(start, epochs + 1):
        seq.append(cur)
        if epoch % decay_freq == 0:
            cur = cur * lr_decay
    return seq


# ----------------------------------------------------------------------------
# tiled inference helpers (reference infer.py:14-68) -- "next" row f1
# ----------------------------------------------------------------------------


def n_crop(image, size, overlap):
    """reference infer.py:14-34, INCLUDING its ``j * ncropsy + i`` indexing
    (exact for square images only)."""
    import numpy as np
    c, height, width = image.shape
    eff = int(overlap * size)
    ncy = int(np.ceil(height / eff))
    ncx = int(np.ceil(width / eff))
    crops = torch.zeros((ncx * ncy, c, size, size))
    for j in range(ncy):
        for i in range(ncx):
            sy = j * eff
            sx = i * eff
            sy -= max(sy + size - height, 0)
            sx -= max(sx + size - width, 0)
            crops[j * ncy + i, :] = image[:, sy:sy + size, sx:sx + size]
    return crops


def build_mask(masks, crop_size, image_size, threshold, overlap):
    """reference infer.py:37-68 (float64 overlap-average, threshold, argmax if C>1)."""
    import numpy as np
    n, c, height, width = masks.shape
    ih, iw = image_size
    mask = np.zeros((c, ih, iw))
    count = np.zeros((c, ih, iw))
    eff = int(overlap * crop_size)
    ncy = int(np.ceil(ih / eff))
    ncx = int(np.ceil(iw / eff))
    for j in range(ncy):
        for i in range(ncx):
            sy = j * eff
            sx = i * eff
            sy -= max(sy + crop_size - ih, 0)
            sx -= max(sx + crop_size - iw, 0)
            mask[:, sy:sy + crop_size, sx:sx + crop_size] += masks[j * ncy + i, :]
            count[:, sy:sy + crop_size, sx:sx + crop_size] += 1
    mask = mask / count
    if threshold > 0:
        mask[mask >= threshold] = 1
        mask[mask < threshold] = 0
    if c > 1:
        return np.argmax(mask, axis=0)
    return mask[0]
