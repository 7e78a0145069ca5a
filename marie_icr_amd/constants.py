"""Where the model zoo lives — the two names of marie/constants.py:92-97 the hot path reads.

``MARIE_DEFAULT_MOUNT`` is the reference's own environment variable; without it the reference falls back to the parent of
its package directory, here the parent of this package."""
import os as _os

__root_dir__ = _os.path.dirname(_os.path.abspath(__file__))
__default_mount_point__ = _os.environ.get("MARIE_DEFAULT_MOUNT", _os.path.abspath(_os.path.join(__root_dir__, "..")))
__model_path__ = _os.path.join(__default_mount_point__, "model_zoo")
__config_dir__ = _os.path.join(__default_mount_point__, "config")
