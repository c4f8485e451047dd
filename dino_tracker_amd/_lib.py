"""ctypes binding of libdtk.so (include/dtk.h).  No torch types cross the boundary: device pointers, sizes, a stream.

The hot path has NO CPU fallback: `lib()` raises if the shared library has not been built, and every wrapper in
ops.py raises if a tensor does not live on a GPU.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdtk.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class Geom(ctypes.Structure):
    """struct dtk_geom (include/dtk.h)."""
    _fields_ = [("T", ctypes.c_int32), ("C", ctypes.c_int32), ("ph", ctypes.c_int32), ("pw", ctypes.c_int32),
                ("video_h", ctypes.c_int32), ("video_w", ctypes.c_int32), ("patch", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("radius", ctypes.c_float)]

    @property
    def HW(self) -> int:
        return self.ph * self.pw


def make_geom(T: int, C: int, video_h: int, video_w: int, patch: int = 14, stride: int = 7, radius: float = 35.0) -> Geom:
    ph = 1 + (video_h - patch) // stride
    pw = 1 + (video_w - patch) // stride
    return Geom(T, C, ph, pw, video_h, video_w, patch, stride, radius)


class VitLayer(ctypes.Structure):
    """struct dtk_vit_layer."""
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b",
                                        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2",
                                        "qkv_w_lo", "proj_w_lo", "fc1_w_lo", "fc2_w_lo")] + [("w_scale", ctypes.c_float)]


class VitModel(ctypes.Structure):
    """struct dtk_vit_model."""
    _fields_ = [("D", ctypes.c_int32), ("heads", ctypes.c_int32), ("depth", ctypes.c_int32), ("patch", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("flags", ctypes.c_int32),
                ("patch_w", c_void_p), ("patch_b", c_void_p),
                ("cls_pos", c_void_p), ("pos", c_void_p), ("mean_std", c_void_p), ("layers", ctypes.POINTER(VitLayer)),
                ("frame_batch", ctypes.c_int32), ("overflow", c_void_p),
                ("tap_out", c_void_p), ("tap_mask", ctypes.c_uint64), ("tap_scale", ctypes.c_float)]


VIT_TILED_GEMMS, VIT_BF16, VIT_CHECK_RANGE, VIT_ATTENTION_V2, VIT_GEMM_WS_V1, VIT_ATTENTION_V4, VIT_GEMM_WIDE_V1, VIT_NO_LN_FUSION = 1, 2, 4, 8, 16, 32, 64, 128  # dtk_vit_model.flags
OPERAND_F16, OPERAND_BF16, OPERAND_ATTENTION_V2, OPERAND_ATTENTION_V4 = 0, 1, 0x100, 0x2000
OPERAND_ATTENTION_V5, OPERAND_ATTENTION_V5_INPHASE = 0x200, 0x400   # stand-alone stage only: the round-5 experiment kernel (vit_attention5.h)


class TrackOpts(ctypes.Structure):
    """struct dtk_track_opts."""
    _fields_ = [("method", ctypes.c_int32), ("normalized", ctypes.c_int32), ("round_sources", ctypes.c_int32),
                ("tier", ctypes.c_int32), ("emb_rows", ctypes.c_int32)]


ADAM_MAX_TENSORS, ADAM_MAX_GROUPS = 32, 4


class AdamArgs(ctypes.Structure):
    """struct dtk_adam_args (include/dtk.h): the tensors of one optimiser step, by value."""
    _fields_ = [("param", ctypes.c_void_p * ADAM_MAX_TENSORS), ("grad", ctypes.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", ctypes.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", ctypes.c_void_p * ADAM_MAX_TENSORS),
                ("numel", ctypes.c_int64 * ADAM_MAX_TENSORS), ("group", ctypes.c_int32 * ADAM_MAX_TENSORS),
                ("step", ctypes.c_int32 * ADAM_MAX_TENSORS), ("n_tensors", ctypes.c_int32), ("lr", ctypes.c_double * ADAM_MAX_GROUPS),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double)]


class TrackStats(ctypes.Structure):
    """struct dtk_track_stats (host side, filled by dtk_track)."""
    _fields_ = [("sources", ctypes.c_int32), ("whole_map_tier", ctypes.c_int32), ("exact_tier", ctypes.c_int32),
                ("syncs", ctypes.c_int32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# name -> (restype, argtypes); must list every symbol include/dtk.h declares (tests/test_abi.py checks this)
SIGNATURES = {
    "dtk_version": (c_int, []),
    "dtk_last_error": (ctypes.c_char_p, []),
    "dtk_profile_enable": (c_int, [c_int]),
    "dtk_profile_collect": (c_int, []),
    "dtk_profile_name": (ctypes.c_char_p, [c_int]),
    "dtk_profile_ms": (ctypes.c_double, [c_int]),
    "dtk_profile_launches": (ctypes.c_longlong, [c_int]),
    "dtk_pack_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtk_unpack_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtk_feature_norms": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtk_vit_workspace_bytes": (c_size_t, [ctypes.POINTER(VitModel), c_int, c_int, c_int]),
    "dtk_vit_forward": (c_int, [ctypes.POINTER(VitModel), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "dtk_vit_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtk_vit_attention_split": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtk_delta_dino_packed_floats": (c_size_t, [c_int, c_int]),
    "dtk_delta_dino_pack": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                    c_void_p, c_void_p]),
    "dtk_delta_dino_workspace_bytes": (c_size_t, [ctypes.POINTER(Geom)]),
    "dtk_delta_dino_refine": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                      c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dtk_delta_dino_refine_mode": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                           c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dtk_sample_points": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dtk_sample_grid": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dtk_normalized_conv2d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "dtk_corr_maps": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_void_p]),
    "dtk_head_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_head_forward": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dtk_head_forward_train": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dtk_conv_split_weight_halves": (c_size_t, [c_int, c_int]),
    "dtk_conv_split_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dtk_conv_split_input": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_conv_split_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p]),
    "dtk_conv_split_output": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dtk_conv_wgrad_split_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dtk_conv_wgrad_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_int, c_void_p, c_size_t, c_void_p]),
    "dtk_gemm_nt_f32_indexed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int64, c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_contrastive_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dtk_contrastive_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_contrastive_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dtk_emb_reg_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dtk_sample_bilinear_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    "dtk_sample_bilinear_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    "dtk_pow2_scale": (c_int, [c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p]),
    "dtk_adam_step": (c_int, [ctypes.POINTER(AdamArgs), c_void_p]),
    "dtk_adam_scalars": (c_int, [ctypes.POINTER(AdamArgs), ctypes.POINTER(ctypes.c_float)]),
    "dtk_adam_step_dev": (c_int, [ctypes.POINTER(AdamArgs), c_void_p, c_void_p]),
    "dtk_emb_reg_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtk_corr_window_backward": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p]),
    "dtk_head_backward": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p]),
    "dtk_track_workspace_bytes": (c_size_t, [ctypes.POINTER(Geom), c_int, ctypes.POINTER(TrackOpts)]),
    "dtk_track": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_int, c_void_p, ctypes.POINTER(TrackOpts), ctypes.POINTER(TrackStats),
                          c_void_p, c_size_t, c_void_p]),
    "dtk_argmax_cells": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dtk_bb_nms_workspace_bytes": (c_size_t, [ctypes.POINTER(Geom), c_int]),
    "dtk_bb_nms": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                           c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dtk_feat_f16_bytes": (c_size_t, [ctypes.POINTER(Geom)]),
    "dtk_make_feat_f16": (c_int, [ctypes.POINTER(Geom), c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_traj_cos_sims": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtk_build_anchor_sources": (c_int, [c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtk_occlusion": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int,
                              c_int, c_void_p]),
    "dtk_tapvid_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                  c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtk_blurpool_forward": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_void_p]),
    "dtk_blurpool_backward": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_void_p]),
    "dtk_gemm_nt_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dtk_im2col": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                           ctypes.c_int64, c_void_p]),
    "dtk_col2im": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtk_transpose_f32": (c_int, [c_void_p, c_void_p, ctypes.c_int64, ctypes.c_int64, c_int, c_void_p]),
    "dtk_resample2d_forward": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "dtk_resample2d_backward": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "dtk_batchnorm_workspace_bytes": (c_size_t, [c_int]),
    "dtk_batchnorm_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                            c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                                            c_void_p]),
    "dtk_batchnorm_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
}

_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C dino_tracker_amd/csrc). There is no CPU fallback for the hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libdtk error {rc}: {lib().dtk_last_error().decode()}")
