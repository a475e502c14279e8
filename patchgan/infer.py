from patchgan_amd.infer import *  # noqa: F401,F403
from patchgan_amd.infer import patchgan_infer, n_crop, build_mask  # noqa: F401
