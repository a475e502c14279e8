from patchgan_amd.trainer import *  # noqa: F401,F403
from patchgan_amd.trainer import Trainer, weights_init  # noqa: F401
