from patchgan_amd.unet import *  # noqa: F401,F403
