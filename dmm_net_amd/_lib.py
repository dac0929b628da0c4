"""Loader for libdmm_match.so (the gfx950 HIP library) through its C ABI (include/dmm_match.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C dmm_net_amd/csrc``) and
loaded with ctypes.  There is NO fallback: if the library is missing or a call fails, the product
raises -- nothing here ever routes to a CPU path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdmm_match.so")        # no environment override: see use_library()
CSRC = os.path.join(_HERE, "csrc")

DMM_OK = 0
DTYPE_F32, DTYPE_F16, DTYPE_BF16, DTYPE_PACKED1 = 0, 1, 2, 3
MAX_TEMPLATES = 32
MAX_PROPOSALS = 256
FRAME_TABLE = -(1 << 63)          # DMM_FRAME_TABLE: sp_b sentinel, masks_p = device table of per-frame base pointers

# every symbol include/dmm_match.h declares
SYMBOLS = (
    "dmm_abi_version", "dmm_status_string", "dmm_last_hip_error", "dmm_build_info", "dmm_launch_count",
    "dmm_set_option", "dmm_get_option", "dmm_reset_options",
    "dmm_iou_counts", "dmm_iou_counts_dual", "dmm_feature_normalize_f32", "dmm_cosine_f32", "dmm_cosine_features_f32", "dmm_feature_sim_bwd_f32", "dmm_relax_match_f32", "dmm_relax_solve_f32",
    "dmm_relax_bwd_workspace_bytes", "dmm_relax_match_bwd_f32",
    "dmm_mask_mix", "dmm_mask_mix_to", "dmm_mask_mix_shared_to", "dmm_mask_mix_shared_frames", "dmm_mask_mix_bwd", "dmm_workspace_bytes", "dmm_match_forward", "dmm_match_forward_ws", "dmm_roialign4_mean_fwd", "dmm_roialign4_mean_bwd",
    "dmm_iou_counts_frames", "dmm_iou_counts_dual_frames", "dmm_mask_mix_frames", "dmm_mask_mix_bwd_frames",
    "dmm_bias_act_bf16", "dmm_paste_masks_f32", "dmm_nms_f32", "dmm_pack_words", "dmm_pack_masks", "dmm_mask_boxes_f32", "dmm_merge_labels_f32", "dmm_ragged_pad",
    "dmm_workspace_bytes_packed", "dmm_match_forward_packed", "dmm_proposal_boxes_f32", "dmm_nms_slots_f32",
    "dmm_paste_kept_f32", "dmm_step_select_i32", "dmm_step_advance", "dmm_commit_masks_f32", "dmm_roialign4_mean_nhwc_fwd",
    "dmm_conv1x1_bf16", "dmm_im2col3x3_bf16", "dmm_bias_relu_maxpool_bf16", "dmm_relax_any_scratch_bytes", "dmm_relax_match_any_f32", "dmm_relax_match_f16s", "dmm_match_solve_packed", "dmm_step_finish_f32",
    "dmm_matching_loss_f32", "dmm_match_train_tape_bytes", "dmm_match_train_forward_workspace_bytes", "dmm_match_train_forward",
    "dmm_match_train_backward_workspace_bytes", "dmm_match_train_backward",
    "dmm_bn_stats_bf16", "dmm_bn_apply_bf16", "dmm_bn_bwd_reduce_bf16", "dmm_bn_bwd_dx_bf16",
    "dmm_bn_stats_grouped_bf16", "dmm_bn_apply_grouped_bf16", "dmm_bn_bwd_reduce_grouped_bf16", "dmm_bn_bwd_dx_grouped_bf16",
    "dmm_graph_nodes_to_kernels", "dmm_wprep3x3_bf16", "dmm_cast_many_bf16", "dmm_subsample2_bf16", "dmm_upsample2_zero_bf16", "dmm_wgrad_workspace_bytes", "dmm_wgrad_bf16", "dmm_wgrad3x3_bf16",
)

_lib = None


class DmmError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into dmm_net_amd/libdmm_match.so (hipcc cross-compiles)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise DmmError("build did not produce " + LIB_PATH)
    return LIB_PATH


def use_library(path: str) -> None:
    """Load ANOTHER build of the library instead of the in-tree one (A/B builds in tools/: cached instead of non-temporal
    plane loads, other tile sizes).  An explicit call BEFORE the first ``load()``: no environment variable can change which
    library production loads."""
    global LIB_PATH
    if _lib is not None:
        raise DmmError("use_library() after the library has been loaded")
    LIB_PATH = os.path.abspath(path)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the HIP extension is mandatory; there is no CPU fallback)")
    import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    L = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise DmmError(f"{LIB_PATH} does not export {s}")
    c_int, c_float, c_i64, vp, sz = ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    L.dmm_abi_version.restype = c_int
    L.dmm_status_string.restype = ctypes.c_char_p
    L.dmm_status_string.argtypes = [c_int]
    L.dmm_last_hip_error.restype = c_int
    L.dmm_build_info.restype = ctypes.c_char_p
    L.dmm_launch_count.restype = ctypes.c_longlong
    L.dmm_set_option.argtypes = [c_int, c_int]
    L.dmm_set_option.restype = c_int
    L.dmm_get_option.argtypes = [c_int]
    L.dmm_get_option.restype = c_int
    L.dmm_reset_options.restype = c_int
    L.dmm_iou_counts.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, vp, vp,
                                 vp, vp, vp, vp]
    L.dmm_iou_counts_dual.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_i64,
                                      c_i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_iou_counts_frames.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, vp, vp, vp, vp, vp,
                                        vp]
    L.dmm_iou_counts_dual_frames.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64,
                                             c_i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_mask_mix_frames.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, vp, vp, vp, c_i64, c_i64,
                                      vp]
    L.dmm_mask_mix_bwd_frames.argtypes = [vp, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_i64, vp, vp, vp, vp]
    L.dmm_feature_normalize_f32.argtypes = [vp, c_i64, c_int, vp, vp, vp]
    L.dmm_cosine_f32.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, vp, vp, vp]
    L.dmm_feature_sim_bwd_f32.argtypes = [vp, vp, vp, vp, c_float, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp,
                                          vp, vp, vp]
    L.dmm_feature_sim_bwd_f32.restype = c_int
    L.dmm_cosine_features_f32.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, vp]
    L.dmm_cosine_features_f32.restype = c_int
    L.dmm_relax_match_f32.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, c_float, c_int, c_int, c_float,
                                      c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_relax_match_f16s.argtypes = L.dmm_relax_match_f32.argtypes
    L.dmm_relax_match_any_f32.argtypes = L.dmm_relax_match_f32.argtypes[:-1] + [vp, sz, vp]
    L.dmm_relax_match_any_f32.restype = c_int
    L.dmm_relax_any_scratch_bytes.argtypes = [c_int, c_int, c_int]
    L.dmm_relax_any_scratch_bytes.restype = sz
    L.dmm_relax_match_f16s.restype = c_int
    L.dmm_relax_solve_f32.argtypes = [vp, c_int, c_int, c_int, vp, vp, c_int, c_int, c_float, vp, vp, vp, vp, vp]
    L.dmm_relax_bwd_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.dmm_relax_bwd_workspace_bytes.restype = sz
    L.dmm_relax_match_bwd_f32.argtypes = [vp, vp, c_int, c_int, c_int, vp, vp, c_int, c_int, c_float, c_int, vp, vp, vp,
                                          vp, vp, sz, vp]
    L.dmm_roialign4_mean_fwd.argtypes = [vp, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, vp]
    L.dmm_conv1x1_bf16.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, c_int, vp, vp, sz, vp]
    L.dmm_conv1x1_bf16.restype = c_int
    L.dmm_im2col3x3_bf16.argtypes = [vp, c_int, c_int, c_int, c_int, c_int, vp, vp]
    L.dmm_im2col3x3_bf16.restype = c_int
    L.dmm_bias_relu_maxpool_bf16.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, vp]
    L.dmm_bias_relu_maxpool_bf16.restype = c_int
    L.dmm_roialign4_mean_nhwc_fwd.argtypes = [vp, c_int, c_int, c_int, vp, vp, vp, vp, c_int, vp, vp]
    L.dmm_roialign4_mean_nhwc_fwd.restype = c_int
    L.dmm_roialign4_mean_bwd.argtypes = [vp, c_int, c_int, vp, vp, vp, vp, c_int, vp, vp]
    L.dmm_paste_masks_f32.argtypes = [vp, c_int, c_int, vp, c_int, c_int, c_float, c_int, vp, c_i64, vp, vp, vp]
    L.dmm_pack_words.argtypes = [c_int]
    L.dmm_pack_words.restype = c_i64
    L.dmm_pack_masks.argtypes = [vp, c_int, c_i64, c_int, c_i64, vp, c_i64, vp]
    L.dmm_nms_f32.argtypes = [vp, vp, vp, c_int, c_int, c_float, c_int, vp, vp, vp]
    L.dmm_mask_boxes_f32.argtypes = [vp, c_int, c_int, c_int, c_i64, c_float, vp, vp, vp]
    L.dmm_merge_labels_f32.argtypes = [vp, c_int, c_int, c_int, c_i64, c_i64, vp, vp, vp]
    L.dmm_ragged_pad.argtypes = [vp, vp, c_int, c_int, c_i64, vp, vp]
    L.dmm_mask_mix.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, vp, vp, vp, c_i64,
                               c_i64, vp]
    L.dmm_bias_act_bf16.argtypes = [vp, vp, vp, c_i64, c_int, c_int, vp]
    L.dmm_bias_act_bf16.restype = c_int
    L.dmm_bn_stats_bf16.argtypes = [vp, c_i64, c_int, vp, vp]
    L.dmm_bn_apply_bf16.argtypes = [vp, vp, c_i64, c_int, vp, vp, vp, vp, vp, c_float, c_float, c_int, vp, vp, vp]
    L.dmm_bn_bwd_reduce_bf16.argtypes = [vp, vp, vp, c_i64, c_int, vp, vp, vp, c_int, vp, vp]
    L.dmm_bn_bwd_dx_bf16.argtypes = [vp, vp, vp, c_i64, c_int, vp, vp, vp, vp, c_int, vp, vp, vp, vp, vp]
    L.dmm_bn_stats_grouped_bf16.argtypes = [vp, c_i64, c_int, c_int, vp, vp]
    L.dmm_bn_apply_grouped_bf16.argtypes = [vp, vp, c_i64, c_int, c_int, vp, vp, vp, vp, vp, c_float, c_float, c_int, vp, vp, vp]
    L.dmm_bn_bwd_reduce_grouped_bf16.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, vp, vp, vp, c_int, vp, vp]
    L.dmm_bn_bwd_dx_grouped_bf16.argtypes = [vp, vp, vp, vp, c_i64, c_int, c_int, vp, vp, vp, vp, c_int, vp, vp, vp, vp, vp]
    L.dmm_graph_nodes_to_kernels.argtypes = [vp, c_int, vp, vp, vp]
    L.dmm_wgrad_bf16.argtypes = [vp, vp, c_i64, c_int, c_int, c_i64, c_i64, vp, vp, sz, vp]
    L.dmm_wgrad3x3_bf16.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, sz, vp]
    L.dmm_wprep3x3_bf16.argtypes = [vp, c_int, c_i64, vp]
    L.dmm_wprep3x3_bf16.restype = c_int
    L.dmm_cast_many_bf16.argtypes = [vp, c_int, c_i64, vp]
    L.dmm_cast_many_bf16.restype = c_int
    L.dmm_subsample2_bf16.argtypes = [vp, c_int, c_int, c_int, c_int, vp, vp]
    L.dmm_subsample2_bf16.restype = c_int
    L.dmm_upsample2_zero_bf16.argtypes = [vp, c_int, c_int, c_int, c_int, vp, vp]
    L.dmm_upsample2_zero_bf16.restype = c_int
    L.dmm_wgrad_workspace_bytes.argtypes = [c_i64, c_int, c_int]
    L.dmm_wgrad_workspace_bytes.restype = sz
    L.dmm_wgrad_bf16.restype = L.dmm_wgrad3x3_bf16.restype = c_int
    for f in ("dmm_bn_stats_bf16", "dmm_bn_apply_bf16", "dmm_bn_bwd_reduce_bf16", "dmm_bn_bwd_dx_bf16",
              "dmm_bn_stats_grouped_bf16", "dmm_bn_apply_grouped_bf16", "dmm_bn_bwd_reduce_grouped_bf16",
              "dmm_bn_bwd_dx_grouped_bf16", "dmm_graph_nodes_to_kernels"):
        getattr(L, f).restype = c_int
    L.dmm_mask_mix_to.argtypes = [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, vp, vp, vp, c_int, c_i64,
                                  c_i64, vp]
    L.dmm_mask_mix_to.restype = c_int
    L.dmm_mask_mix_shared_to.argtypes = L.dmm_mask_mix_to.argtypes
    L.dmm_mask_mix_shared_to.restype = c_int
    L.dmm_mask_mix_shared_frames.argtypes = L.dmm_mask_mix_frames.argtypes
    L.dmm_mask_mix_shared_frames.restype = c_int
    L.dmm_mask_mix_bwd.argtypes = [vp, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, vp, vp, vp, vp]
    L.dmm_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    L.dmm_workspace_bytes.restype = sz
    L.dmm_match_forward.argtypes = [vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64,
                                    c_i64, c_i64, vp, vp, c_float, c_int, c_int, c_float, c_int, vp, vp, vp, vp,
                                    vp, vp, vp, vp, sz, vp]
    L.dmm_match_forward_ws.argtypes = [vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64,
                                       c_i64, c_i64, vp, vp, c_float, c_int, c_int, c_float, c_int, vp, vp, vp, vp,
                                       vp, vp, vp, vp, sz, ctypes.POINTER(c_int), vp]
    L.dmm_workspace_bytes_packed.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.dmm_workspace_bytes_packed.restype = sz
    L.dmm_match_forward_packed.argtypes = [vp, vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64,
                                           c_i64, c_i64, c_i64, c_i64, vp, vp, c_float, c_int, c_int, c_float, c_int, vp,
                                           vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.dmm_proposal_boxes_f32.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_float, c_int, vp, vp, vp]
    L.dmm_nms_slots_f32.argtypes = [vp, vp, vp, c_int, c_int, c_float, c_int, vp, vp, vp, vp]
    L.dmm_paste_kept_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp,
                                     c_i64, vp, vp, vp, vp, vp]
    L.dmm_match_solve_packed.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, c_float, c_int,
                                         c_int, c_float, c_int, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.dmm_match_solve_packed.restype = c_int
    L.dmm_step_finish_f32.argtypes = [vp, c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_step_finish_f32.restype = c_int
    L.dmm_step_select_i32.argtypes = [vp, vp, c_int, vp, vp]
    L.dmm_step_advance.argtypes = [vp, vp]
    L.dmm_commit_masks_f32.argtypes = [vp, vp, vp, c_int, c_i64, vp]
    L.dmm_matching_loss_f32.argtypes = [vp, vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, vp]
    L.dmm_matching_loss_f32.restype = c_int
    L.dmm_match_train_forward_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    L.dmm_match_train_forward_workspace_bytes.restype = sz
    L.dmm_match_train_tape_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.dmm_match_train_tape_bytes.restype = sz
    L.dmm_match_train_forward.argtypes = [vp, vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64,
                                          c_i64, c_i64, c_i64, c_i64, vp, vp, c_float, c_int, c_int, c_float, c_int, vp, vp,
                                          vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, ctypes.POINTER(c_int), vp]
    L.dmm_match_train_forward.restype = c_int
    L.dmm_match_train_backward_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int]
    L.dmm_match_train_backward_workspace_bytes.restype = sz
    L.dmm_match_train_backward.argtypes = [vp, c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int,
                                           c_int, c_i64, c_i64, vp, vp, c_float, c_int, c_int, c_float, c_int, vp, vp, vp,
                                           sz, vp, vp, c_int, vp]
    L.dmm_match_train_backward.restype = c_int
    for f in ("dmm_match_forward_packed", "dmm_proposal_boxes_f32", "dmm_nms_slots_f32", "dmm_paste_kept_f32",
              "dmm_step_select_i32", "dmm_step_advance", "dmm_commit_masks_f32"):
        getattr(L, f).restype = c_int
    for f in ("dmm_iou_counts_frames", "dmm_iou_counts_dual_frames", "dmm_mask_mix_frames", "dmm_mask_mix_bwd_frames",
              "dmm_iou_counts", "dmm_iou_counts_dual", "dmm_feature_normalize_f32", "dmm_cosine_f32", "dmm_relax_match_f32", "dmm_relax_solve_f32",
              "dmm_mask_mix", "dmm_match_forward", "dmm_match_forward_ws", "dmm_relax_match_bwd_f32", "dmm_roialign4_mean_fwd",
              "dmm_roialign4_mean_bwd", "dmm_mask_mix_bwd", "dmm_paste_masks_f32", "dmm_nms_f32", "dmm_pack_masks",
              "dmm_mask_boxes_f32", "dmm_merge_labels_f32", "dmm_ragged_pad"):
        getattr(L, f).restype = c_int
    if L.dmm_abi_version() != 2:
        raise DmmError("libdmm_match.so ABI version mismatch")
    # libdmm_match.so needs libhipblaslt.so.1 / libamdhip64.so.7 by SONAME; torch (imported above) has already mapped its
    # bundled copies under the same sonames, so the loader binds to those -- ONE HIP runtime and ONE hipBLASLt per process.
    try:
        with open("/proc/self/maps") as f:
            lt = {ln.split()[-1] for ln in f if "libhipblaslt.so" in ln}
        if len(lt) > 1:
            import warnings
            warnings.warn(f"two hipBLASLt copies are mapped ({sorted(lt)}): import torch before loading {LIB_PATH}")
    except OSError:
        pass
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != DMM_OK:
        L = load()
        msg = L.dmm_status_string(rc).decode()
        extra = f" (hipError {L.dmm_last_hip_error()})" if rc == 3 else ""
        raise DmmError(f"{what}: {msg}{extra}")


# include/dmm_match.h (0): dispatch options, set through the ABI (the library reads no environment variable)
OPTIONS = {name: k for k, name in enumerate((
    "COST_KERNEL", "COST_TINY_FRAMES", "SOLVER_KERNEL", "FORCE_WIDE", "COSINE_KERNEL", "COST_WGS", "COST_SMALL_WGS",
    "COST_TL_WGS", "COST_XCD", "MIX_XCD", "MIX_WGS", "MIX_STEPQ", "MIX_ALIGN", "MIX_NT", "SOLVER_HELPER_MAX", "NMS_WAVE",
    "COS_ROWS_MIN_N", "GEMM_TUNE", "PACK_VARIANT", "SMALL_FUSED", "MIX_SHARED", "MIX_SHARED_STEPS", "FEAT_BWD_FRAME",
    "MIX_SHARED_LOCKSTEP"))}


def set_option(name: str, value: int):
    check(load().dmm_set_option(OPTIONS[name], int(value)), f"dmm_set_option({name}, {value})")


def get_option(name: str) -> int:
    return int(load().dmm_get_option(OPTIONS[name]))


_OPTIONS_LOCK = None


class options:
    """``with _lib.options(COST_KERNEL=1, COST_TINY_FRAMES=0): ...`` -- pin dispatch options for a block (tests, A/B
    timing) and put the previous values back.  The options are PROCESS-WIDE: the block holds a re-entrant lock, so two
    threads pinning different values run their blocks one after the other instead of seeing each other's settings
    (ADVICE r4); threads that do not pin anything are not held up -- and see whatever is pinned at the moment."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        global _OPTIONS_LOCK
        if _OPTIONS_LOCK is None:
            import threading
            _OPTIONS_LOCK = threading.RLock()
        _OPTIONS_LOCK.acquire()
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        try:
            for k, v in self.old.items():
                set_option(k, v)
        finally:
            _OPTIONS_LOCK.release()
        return False


def small_to_device(values, dtype, device):
    """A short host list as a device tensor WITHOUT stalling the host: ``torch.tensor(values, device=cuda)`` copies
    from pageable memory, which the HIP runtime completes synchronously -- the host waits until the stream has drained
    to the copy (0.2 ms per call in the frame loop, 6 calls per frame step: the host could not run ahead of the GPU).
    Pinned staging + non_blocking: the caching host allocator keeps the pinned block until the copy's event has passed."""
    import torch
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def small_to_device_many(specs, device, memo=None):
    """Several short host lists in ONE pinned staging buffer and ONE copy: ``specs`` = [(values, torch dtype), ...] ->
    one device tensor per entry (views of one allocation, each 8-byte aligned).  The per-video driver hands five such
    tables to the layer per call (proposal / template counts, three pointer tables): five pinned allocations and five
    copies were ~70 us of host time per call (round 5, ``tools/dropin_trace.py model cprofile``).
    ``memo`` (a dict the caller keeps): when the VALUES are the same as in the caller's previous call -- the frames of a clip
    keep their counts, and the caching allocator hands the same addresses to same-sized tensors step after step -- the
    previous device tables are returned as they are (read-only by contract): no staging, no copy (~35 us of host time)."""
    import numpy as np
    import torch
    if memo is not None:
        key = (str(device), tuple((tuple(v), d) for v, d in specs))
        if memo.get("key") == key:
            return memo["out"]
    np_of = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32}
    parts, spans, off = [], [], 0
    for values, dtype in specs:
        a = np.asarray(values, dtype=np_of[dtype])
        nbytes = a.size * a.itemsize
        pad = (-nbytes) % 8
        parts.append(a.view(np.uint8))
        if pad:
            parts.append(np.zeros(pad, dtype=np.uint8))
        spans.append((off, nbytes, dtype))
        off += nbytes + pad
    host = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8))
    if torch.device(device).type == "cuda":
        buf = host.pin_memory().to(device, non_blocking=True)
    else:
        buf = host.to(device)
    out = [buf[o:o + n].view(dt) for (o, n, dt) in spans]
    if memo is not None:
        memo["key"], memo["out"] = key, out
    return out


_NULL_CTX = None


def device_guard(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already the current one: the launchers run under a
    device guard so that the HIP runtime targets the tensors' GPU, but pushing and popping the guard costs several
    microseconds per op -- with ~20 ops per frame step in a host-bound loop -- and almost every call is on the
    current device already."""
    global _NULL_CTX
    import contextlib
    import torch
    if _NULL_CTX is None:
        _NULL_CTX = contextlib.nullcontext()
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NULL_CTX
    return torch.cuda.device(dev)
