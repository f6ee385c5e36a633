"""ctypes binding of the C ABI declared in include/veto_amd.h (libveto_amd.so).

There is no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

# PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
# mapped BEFORE libveto_amd.so so that the library's NEEDED libamdhip64.so.7 resolves to that same copy;
# the other order maps two runtimes into the process and the second one finds no device.
import torch  # noqa: F401
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32, c_int64,
                    c_size_t, c_void_p)

_LIB = None

EXPORTS = [
    "veto_last_error", "veto_version", "veto_create", "veto_destroy", "veto_num_weights",
    "veto_weight_info", "veto_load_weights", "veto_workspace_bytes", "veto_forward", "veto_forward_saturation",
    "veto_enumerate_pairs", "veto_profile_enable", "veto_profile_collect", "veto_profile_entry",
    "veto_profile_reset", "veto_debug_gemm", "veto_debug_gemm_workspace_bytes", "veto_debug_gemm_forms", "veto_debug_ffn", "veto_debug_ffn_workspace_bytes", "veto_debug_outproj", "veto_debug_outproj_workspace_bytes", "veto_debug_layer_tail", "veto_debug_layer_tail_workspace_bytes", "veto_debug_qkv_attn", "veto_debug_qkv_attn_workspace_bytes",
    "veto_postprocess", "veto_postprocess_workspace_bytes", "veto_postprocess_meet", "veto_postprocess_vote",
    "veto_train_workspace_bytes", "veto_grad_floats", "veto_weight_offset", "veto_forward_train", "veto_backward",
    "veto_debug_attention_backward", "veto_debug_layernorm_backward", "veto_debug_layernorm_backward_workspace_bytes",
    "veto_debug_gelu_backward", "veto_debug_column_sums",
    "veto_debug_wgrad", "veto_debug_wgrad_workspace_bytes", "veto_ce_loss", "veto_ce_loss_workspace_bytes", "veto_meet_sample",
    "veto_roi_pool", "veto_roi_pool_backward", "veto_sgg_eval", "veto_sgg_eval_workspace_bytes",
]

VETO_PRECISE, VETO_FAST, VETO_MIXED = 0, 1, 2


class VetoConfig(Structure):
    _fields_ = [(n, c_int32) for n in (
        "struct_size", "dim", "layers", "heads", "patch", "channels", "resolution", "num_obj_cls",
        "embed_dim", "num_out", "precision", "device", "max_chunk_pairs")]


class VetoInputs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("n_obj", c_int32), ("n_pair", c_int32), ("n_img", c_int32),
        ("roi_rgb", c_void_p), ("roi_depth", c_void_p), ("boxes", c_void_p),
        ("box_mode", c_int32), ("reserved0", c_int32),
        ("obj_labels", c_void_p), ("obj_logits", c_void_p), ("rel_pairs", c_void_p),
        ("img_obj_offset", c_void_p), ("img_pair_offset", c_void_p), ("bn_batch_stats", c_void_p),
    ]


class VetoDebugOutputs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("reserved0", c_int32),
        ("subj_inds", c_void_p), ("obj_inds", c_void_p), ("tokens", c_void_p), ("cls", c_void_p),
    ]


class VetoSaturation(Structure):
    _fields_ = [(n, c_int64) for n in ("elements", "f16_saturated", "value_saturated", "resid_saturated")]


SATURATION_SITES = ("qkv_in", "attn_out", "ffn_in", "hidden")       # enum veto_saturation_site


class VetoPostArgs(Structure):
    _fields_ = [(n, c_int32) for n in ("struct_size", "n_img", "n_obj", "n_pair", "n_rel_cls", "n_obj_cls",
                                       "max_pairs_per_image", "reserved0")] + \
               [(n, c_void_p) for n in ("rel_logits", "obj_logits", "rel_pairs", "img_obj_offset", "img_pair_offset",
                                        "obj_scores", "obj_pred", "rel_prob_sorted", "rel_pairs_sorted",
                                        "rel_labels_sorted", "triple_sorted")]


class VetoPostMeetArgs(Structure):
    _fields_ = [(n, c_int32) for n in ("struct_size", "n_obj", "n_pair", "n_groups", "n_rel_cls", "n_obj_cls")] + \
               [(n, c_void_p) for n in ("group_logits", "group_widths", "incre_idx_list", "obj_logits", "rel_pairs",
                                        "obj_scores", "obj_pred", "rel_prob_sorted", "rel_pairs_sorted",
                                        "rel_labels_sorted", "triple_sorted")]


class VetoPostVoteArgs(Structure):
    _fields_ = [(n, c_int32) for n in ("struct_size", "n_obj", "n_pair", "n_groups", "n_rel_cls", "n_obj_cls",
                                       "voting", "reserved0")] + \
               [(n, c_void_p) for n in ("expert_logits", "group_widths", "incre_idx_list", "obj_logits", "rel_pairs",
                                        "obj_scores", "obj_pred", "rel_prob_sorted", "rel_pairs_sorted",
                                        "rel_labels_sorted", "triple_sorted", "kept_count")]


class VetoRoiPoolArgs(Structure):
    _fields_ = [(n, c_int32) for n in ("struct_size", "n_levels", "n_img", "n_roi", "channels", "depth_channels",
                                       "pooled", "sampling_ratio")] + \
               [("level_feat", c_void_p * 4), ("level_h", c_int32 * 4), ("level_w", c_int32 * 4),
                ("level_scale", ctypes.c_float * 4), ("depth_feat", c_void_p), ("depth_h", c_int32),
                ("depth_w", c_int32), ("rois", c_void_p), ("out_rgb", c_void_p), ("out_depth", c_void_p),
                ("out_levels", c_void_p)]


class VetoSggEvalArgs(Structure):
    _fields_ = [("struct_size", c_int32), ("n_img", c_int32), ("n_rel_cls", c_int32), ("n_zeroshot", c_int32),
                ("iou_thres", ctypes.c_float), ("reserved0", c_int32)] + \
               [(n, c_void_p) for n in ("gt_offset", "obj_offset", "pair_offset", "gt_rels", "gt_classes", "gt_boxes",
                                        "pred_pairs", "rel_scores", "pred_classes", "pred_boxes", "obj_scores", "zeroshot",
                                        "gc_rank", "ng_rank", "acc_rank", "zeroshot_flag", "ng_rows", "ng_cols", "ng_count",
                                        "metrics")]


class VetoTrainOpts(Structure):
    _fields_ = [("struct_size", c_int32), ("p_pos", ctypes.c_float), ("p_emb", ctypes.c_float), ("p_attn", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("d_roi_rgb", c_void_p), ("d_roi_depth", c_void_p)]


class VetoError(RuntimeError):
    pass


def library_path():
    # VETO_AMD_LIB: A/B a differently built copy of the same library (tools/ only; never a fallback)
    return os.environ.get("VETO_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                          "libveto_amd.so")


def load_library():
    """Loads libveto_amd.so (building it with hipcc first if it is absent). Raises if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        from .build import build_native
        build_native()
    lib = ctypes.CDLL(path)
    lib.veto_last_error.restype = c_char_p
    lib.veto_version.restype = c_char_p
    lib.veto_create.argtypes = [POINTER(VetoConfig), POINTER(c_void_p)]
    lib.veto_destroy.argtypes = [c_void_p]
    lib.veto_num_weights.argtypes = [c_void_p]
    lib.veto_weight_info.argtypes = [c_void_p, c_int, POINTER(c_char_p), POINTER(c_size_t)]
    lib.veto_load_weights.argtypes = [c_void_p, c_char_p, c_void_p, c_size_t, c_void_p]
    lib.veto_workspace_bytes.argtypes = [c_void_p, c_int32, c_int32]
    lib.veto_workspace_bytes.restype = c_size_t
    lib.veto_forward.argtypes = [c_void_p, c_void_p, POINTER(VetoInputs), c_void_p, c_size_t, c_void_p,
                                 POINTER(VetoDebugOutputs)]
    lib.veto_forward_saturation.argtypes = [c_void_p, c_void_p, POINTER(VetoInputs), c_void_p, c_size_t, c_void_p,
                                            POINTER(VetoSaturation), c_int32]
    lib.veto_enumerate_pairs.argtypes = [c_void_p, c_int32, c_void_p]
    lib.veto_profile_enable.argtypes = [c_void_p, c_int32]
    lib.veto_profile_collect.argtypes = [c_void_p]
    lib.veto_profile_entry.argtypes = [c_void_p, c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_int64),
                                       POINTER(c_double), POINTER(c_double)]
    lib.veto_profile_reset.argtypes = [c_void_p]
    lib.veto_debug_gemm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p, c_size_t]
    lib.veto_debug_gemm_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    lib.veto_debug_gemm_workspace_bytes.restype = c_size_t
    lib.veto_debug_gemm_forms.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p, c_size_t]
    lib.veto_debug_ffn.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                   c_int32, POINTER(ctypes.c_float), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
    lib.veto_debug_outproj.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                       POINTER(ctypes.c_float), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
    lib.veto_debug_outproj_workspace_bytes.argtypes = [c_int32]
    lib.veto_debug_outproj_workspace_bytes.restype = c_size_t
    lib.veto_debug_layer_tail.argtypes = [c_void_p] + [c_void_p] * 10 + [c_int32, c_int32, c_int32, POINTER(ctypes.c_float), c_void_p, c_size_t,
                                                                         c_void_p, c_void_p, c_void_p]
    lib.veto_debug_layer_tail_workspace_bytes.argtypes = [c_int32]
    lib.veto_debug_layer_tail_workspace_bytes.restype = c_size_t
    lib.veto_debug_qkv_attn.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(ctypes.c_float), c_void_p,
                                        c_size_t, c_void_p]
    lib.veto_debug_qkv_attn_workspace_bytes.argtypes = [c_int32]
    lib.veto_debug_qkv_attn_workspace_bytes.restype = c_size_t
    lib.veto_debug_ffn_workspace_bytes.argtypes = [c_int32]
    lib.veto_debug_ffn_workspace_bytes.restype = c_size_t
    lib.veto_postprocess_workspace_bytes.argtypes = [c_int32, c_int32]
    lib.veto_postprocess_workspace_bytes.restype = c_size_t
    lib.veto_postprocess.argtypes = [c_void_p, POINTER(VetoPostArgs), c_void_p, c_size_t]
    lib.veto_postprocess_meet.argtypes = [c_void_p, POINTER(VetoPostMeetArgs), c_void_p, c_size_t]
    lib.veto_postprocess_vote.argtypes = [c_void_p, POINTER(VetoPostVoteArgs), c_void_p, c_size_t]
    lib.veto_train_workspace_bytes.argtypes = [c_void_p, c_int32, c_int32]
    lib.veto_train_workspace_bytes.restype = c_size_t
    lib.veto_grad_floats.argtypes = [c_void_p]
    lib.veto_grad_floats.restype = c_size_t
    lib.veto_weight_offset.argtypes = [c_void_p, c_int, POINTER(c_size_t)]
    lib.veto_forward_train.argtypes = [c_void_p, c_void_p, POINTER(VetoInputs), POINTER(VetoTrainOpts), c_void_p, c_size_t, c_void_p]
    lib.veto_backward.argtypes = [c_void_p, c_void_p, POINTER(VetoInputs), POINTER(VetoTrainOpts), c_void_p, c_size_t, c_void_p, c_void_p]
    lib.veto_debug_attention_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32]
    lib.veto_debug_layernorm_backward_workspace_bytes.argtypes = [c_int32]
    lib.veto_debug_layernorm_backward_workspace_bytes.restype = c_size_t
    lib.veto_debug_layernorm_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                                  c_void_p, c_size_t]
    lib.veto_debug_gelu_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]
    lib.veto_debug_column_sums.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t]
    lib.veto_ce_loss_workspace_bytes.argtypes = [c_int32]
    lib.veto_ce_loss_workspace_bytes.restype = c_size_t
    lib.veto_ce_loss.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                 c_void_p, c_size_t]
    lib.veto_meet_sample.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.veto_debug_wgrad_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32]
    lib.veto_debug_wgrad_workspace_bytes.restype = c_size_t
    lib.veto_debug_wgrad.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t]
    lib.veto_roi_pool.argtypes = [c_void_p, POINTER(VetoRoiPoolArgs)]
    lib.veto_roi_pool_backward.argtypes = [c_void_p, POINTER(VetoRoiPoolArgs), c_void_p, c_void_p, POINTER(c_void_p), c_void_p]
    lib.veto_sgg_eval_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32]
    lib.veto_sgg_eval_workspace_bytes.restype = c_size_t
    lib.veto_sgg_eval.argtypes = [c_void_p, POINTER(VetoSggEvalArgs), c_int32, c_int32, c_void_p, c_size_t]
    _LIB = lib
    return lib


def check(rc):
    if rc < 0:
        raise VetoError("veto_amd native call failed (%d): %s" % (rc, load_library().veto_last_error().decode()))
    return rc


class Engine:
    """Owns one veto_handle_t. Thin: every method maps 1:1 onto a C-ABI call."""

    def __init__(self, layers, heads, num_obj_cls, num_out, precision=VETO_PRECISE, device=0, dim=576,
                 embed_dim=200, max_chunk_pairs=0):
        self.lib = load_library()
        cfg = VetoConfig(ctypes.sizeof(VetoConfig), dim, layers, heads, 2, 256, 8, num_obj_cls, embed_dim,
                         num_out, precision, device, max_chunk_pairs)
        self.cfg = cfg
        h = c_void_p()
        check(self.lib.veto_create(byref(cfg), byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.veto_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weight_specs(self):
        out = []
        for i in range(check(self.lib.veto_num_weights(self.handle))):
            name, numel = c_char_p(), c_size_t()
            check(self.lib.veto_weight_info(self.handle, i, byref(name), byref(numel)))
            out.append((name.value.decode(), numel.value))
        return out

    def load_weight(self, name, ptr, numel, stream=0):
        check(self.lib.veto_load_weights(self.handle, name.encode(), c_void_p(ptr), numel, c_void_p(stream)))

    def workspace_bytes(self, n_obj, n_pair):
        return self.lib.veto_workspace_bytes(self.handle, n_obj, n_pair)

    def forward(self, stream, inputs, workspace_ptr, workspace_bytes, out_ptr, dbg=None):
        check(self.lib.veto_forward(self.handle, c_void_p(stream), byref(inputs), c_void_p(workspace_ptr),
                                    workspace_bytes, c_void_p(out_ptr), byref(dbg) if dbg is not None else None))

    def forward_saturation(self, stream, inputs, workspace_ptr, workspace_bytes, out_ptr):
        """veto_forward_saturation: the forward in its launch-per-stage form plus, per layer and operand site, the count of mixed-row
        elements that sit at the clamp values.  Returns [{site: dict(elements, f16_saturated, value_saturated, resid_saturated)}]
        with one entry per layer (synchronises the stream)."""
        n = self.cfg.layers * len(SATURATION_SITES)
        counts = (VetoSaturation * n)()
        check(self.lib.veto_forward_saturation(self.handle, c_void_p(stream), byref(inputs), c_void_p(workspace_ptr), workspace_bytes,
                                               c_void_p(out_ptr), counts, n))
        out = []
        for layer in range(self.cfg.layers):
            out.append({site: {f: int(getattr(counts[layer * len(SATURATION_SITES) + i], f)) for f, _ in VetoSaturation._fields_}
                        for i, site in enumerate(SATURATION_SITES)})
        return out

    def weight_offsets(self):
        """{weight name: (offset, numel)} in floats into the flat gradient buffer of veto_backward."""
        if getattr(self, "_offsets", None) is None:
            out = {}
            for i in range(check(self.lib.veto_num_weights(self.handle))):
                name, numel, off = c_char_p(), c_size_t(), c_size_t()
                check(self.lib.veto_weight_info(self.handle, i, byref(name), byref(numel)))
                check(self.lib.veto_weight_offset(self.handle, i, byref(off)))
                out[name.value.decode()] = (off.value, numel.value)
            self._offsets = out
        return self._offsets

    def profile_enable(self, on):
        check(self.lib.veto_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.veto_profile_reset(self.handle))

    def profile(self):
        """Returns {kernel: dict(total_ms, launches, flops_per_launch, bytes_per_launch)}."""
        n = check(self.lib.veto_profile_collect(self.handle))
        out = {}
        for i in range(n):
            name, ms, cnt, fl, by = c_char_p(), c_double(), c_int64(), c_double(), c_double()
            check(self.lib.veto_profile_entry(self.handle, i, byref(name), byref(ms), byref(cnt), byref(fl), byref(by)))
            if cnt.value:
                out[name.value.decode()] = dict(total_ms=ms.value, launches=cnt.value,
                                                flops_per_launch=fl.value, bytes_per_launch=by.value)
        return out
