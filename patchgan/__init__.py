"""Drop-in alias: ``from patchgan import UNet, Discriminator, Trainer, __version__`` resolves to the MI355X path."""
from patchgan_amd import UNet, Discriminator, Trainer, __version__

__all__ = ['UNet', 'Discriminator', 'Trainer', '__version__']
