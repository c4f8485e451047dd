"""Drop-in for the reference's top-level utils.py."""
from dino_tracker_amd.utils import (add_config_paths, bilinear_interpolate_video, get_dino_features_video,  # noqa: F401
                                    save_dino_embed_video)
