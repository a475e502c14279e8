from patchgan_amd.transfer import *  # noqa: F401,F403
from patchgan_amd.transfer import Transferable, InvalidCheckpointError  # noqa: F401
