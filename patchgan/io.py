from patchgan_amd.io import *  # noqa: F401,F403
