from patchgan_amd.disc import *  # noqa: F401,F403
