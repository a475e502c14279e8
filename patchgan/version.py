"""Alias of patchgan_amd.version (the reference's setup.py reads patchgan/version.py)."""
from patchgan_amd.version import __version__  # noqa: F401
