"""Convenience alias: `import ivf_amd` puts the (hyphen-named, hence not directly
importable) package directory `interpreting-video-features_amd/` on sys.path and
re-exports its modules under the reference's names."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "interpreting-video-features_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import grad_cam_videos  # noqa: E402,F401
import ivf_engine  # noqa: E402,F401
import ivf_find_masks  # noqa: E402,F401
import ivf_lib  # noqa: E402,F401
import ivf_recipe  # noqa: E402,F401
import ivf_search  # noqa: E402,F401
import ivf_shard  # noqa: E402,F401
import mask  # noqa: E402,F401
import models  # noqa: E402,F401
