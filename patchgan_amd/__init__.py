"""patchgan_amd -- MI355X-native (gfx950) implementation of the patchGAN G+D training hot path behind the
reference's Python surface: ``from patchgan_amd import UNet, Discriminator, Trainer, __version__``."""
import os as _os

# The step uses up to five HIP streams at once (compute, the second stream of a two-stream step, the collectives' stream, RCCL's own,
# torch's copy stream).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in order: with more
# streams than queues two of them share a queue and run strictly one after the other -- measured under data parallelism: the compute
# stream stood still behind the discriminator's backward pass on the second stream (9.07 instead of 8.51 ms per step at cfg2,
# tools/step_phases.py).  Read when the runtime initialises (the first HIP call), so it is set here, at import; an explicit setting wins.
# If the embedding process has already initialised HIP (torch.cuda used before this import) the setting comes too late and is silently
# ignored by the runtime: remembered here, and the trainer warns once when a data-parallel step would need the queues (HWQ_LATE).
HWQ_LATE = False
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    try:
        import torch as _torch
        HWQ_LATE = bool(_torch.cuda.is_initialized())
    except Exception:
        HWQ_LATE = False
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .unet import UNet
from .disc import Discriminator
from .trainer import Trainer
from .version import __version__

__all__ = ['UNet', 'Discriminator', 'Trainer', '__version__']
