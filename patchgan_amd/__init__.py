"""patchgan_amd -- MI355X-native (gfx950) implementation of the patchGAN G+D training hot path behind the
reference's Python surface: ``from patchgan_amd import UNet, Discriminator, Trainer, __version__``."""
from .unet import UNet
from .disc import Discriminator
from .trainer import Trainer
from .version import __version__

__all__ = ['UNet', 'Discriminator', 'Trainer', '__version__']
