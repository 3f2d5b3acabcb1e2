"""Synthetic scene helpers for the tests (thin re-export of pix2pose_amd.synthetic)."""
from pix2pose_amd.synthetic import (LM_K, OBJ_PARAM, decoder_map, make_scene, pose_error, project,  # noqa: F401
                                    random_rotation, render_ellipsoid_nocs, render_nocs_at, stage2_box)
