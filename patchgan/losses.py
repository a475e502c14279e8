from patchgan_amd.losses import *  # noqa: F401,F403
from patchgan_amd.losses import tversky, fc_tversky, MAE_loss, bce_loss  # noqa: F401
