from patchgan_amd.train import *  # noqa: F401,F403
from patchgan_amd.train import patchgan_train  # noqa: F401
