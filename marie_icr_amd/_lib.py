"""ctypes binding of libmarie_hip.so (the C ABI declared in include/marie_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, the product path raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C marie_icr_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
# MARIE_HIP_LIB selects another build of the same library (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get("MARIE_HIP_LIB") or os.path.join(_HERE, "libmarie_hip.so")

PREC_F16 = 0
PREC_F32 = 1

# every symbol include/marie_hip.h declares: (name, restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_SIGNATURES = (
    ("mhip_init", _i, [_i, C.POINTER(_vp)]),
    ("mhip_destroy", _i, [_vp]),
    ("mhip_last_error", C.c_char_p, [_vp]),
    ("mhip_set_stream", _i, [_vp, _vp]),
    ("mhip_synchronize", _i, [_vp]),
    ("mhip_device_info", _i, [_vp, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_sz)]),
    ("mhip_memcpy_dev", _i, [_vp, _vp, _vp, _sz]),
    ("mhip_gate_create", _i, [_vp, C.POINTER(_vp)]),
    ("mhip_gate_destroy", _i, [_vp]),
    ("mhip_gate_signal", _i, [_vp, _vp]),
    ("mhip_gate_count", C.c_longlong, [_vp]),
    ("mhip_gate_open", _i, [_vp, _i]),
    ("mhip_gate_wait", _i, [_vp, _vp, C.c_longlong, _i]),
    ("mhip_profile_enable", _i, [_vp, _i]),
    ("mhip_profile_reset", _i, [_vp]),
    ("mhip_profile_read", _i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("mhip_profile_flops", _i, [_vp, _i, C.POINTER(C.c_double)]),
    ("mhip_kernel_count", _i, []),
    ("mhip_kernel_name", C.c_char_p, [_i]),
    ("mhip_conv2d_nhwc", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_icr_create", _i, [_vp, _i, _i, C.POINTER(_vp)]),
    ("mhip_icr_destroy", _i, [_vp]),
    ("mhip_icr_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_icr_finalize", _i, [_vp]),
    ("mhip_icr_alloc_arena", _i, [_vp]),
    ("mhip_icr_arena", _i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_icr_steps", _i, []),
    ("mhip_icr_forward", _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("mhip_icr_forward_host", _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("mhip_craft_create", _i, [_vp, _i, C.POINTER(_vp)]),
    ("mhip_craft_destroy", _i, [_vp]),
    ("mhip_craft_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_craft_finalize", _i, [_vp]),
    ("mhip_craft_alloc_arena", _i, [_vp]),
    ("mhip_craft_arena", _i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_craft_geometry", _i, [_i, _i, _i, C.c_double, C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(_i),
                                 C.POINTER(_i), C.POINTER(_i)]),
    ("mhip_craft_workspace_bytes", _sz, [_vp, _i, _i, _i, C.c_double]),
    ("mhip_craft_kernel_flops", C.c_double, [_vp, _i, _i, _i, _i, C.c_double]),
    ("mhip_craft_forward", _i, [_vp, _vp, _i, _i, _i, C.c_double, _vp]),
    ("mhip_craft_detect", _i, [_vp, _vp, _i, _i, _i, C.c_double, C.c_float, C.c_float, C.c_float, _vp, _i,
                               C.POINTER(_i), _vp, C.POINTER(C.c_double)]),
    ("mhip_craft_detect_host", _i, [_vp, _vp, _i, _i, _i, C.c_double, C.c_float, C.c_float, C.c_float, _vp, _i,
                                    C.POINTER(_i), _vp, C.POINTER(C.c_double)]),
    ("mhip_crop_batch", _i, [_vp, _vp, _vp, _i, _i, _vp]),
    ("mhip_pil_resize_rgb_host", _i, [_vp, _vp, _i, _i, _vp, _i, _i, _i]),
    ("mhip_vit_create", _i, [_vp, _i, _vp, C.POINTER(_vp)]),
    ("mhip_vit_destroy", _i, [_vp]),
    ("mhip_vit_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_vit_finalize", _i, [_vp]),
    ("mhip_vit_alloc_arena", _i, [_vp]),
    ("mhip_vit_arena", _i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_vit_forward_host", _i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_dit_default_config", _i, [_i, _vp]),
    ("mhip_dit_resized_shape", _i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("mhip_dit_create", _i, [_vp, _i, _vp, C.POINTER(_vp)]),
    ("mhip_dit_destroy", _i, [_vp]),
    ("mhip_dit_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_dit_finalize", _i, [_vp]),
    ("mhip_dit_alloc_arena", _i, [_vp]),
    ("mhip_dit_arena", _i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_dit_workspace_bytes", _sz, [_vp, _i, _i, _i]),
    ("mhip_dit_detect", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    ("mhip_dit_detect_host", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    ("mhip_dit_debug_host", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_dit_debug_taps_host", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_rpn_proposals_host", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.c_float, _vp, _vp, _vp]),
    ("mhip_roi_align_host", _i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    ("mhip_det_final_host", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, _i, _vp, _vp, _vp]),
    ("mhip_blackout_bboxes", _i, [_vp, _vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    ("mhip_content_extents", _i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    ("mhip_overlay_create", _i, [_vp, _i, _i, C.POINTER(_vp)]),
    ("mhip_overlay_destroy", _i, [_vp]),
    ("mhip_overlay_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_overlay_finalize", _i, [_vp]),
    ("mhip_overlay_padded_shape", _i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    ("mhip_overlay_forward", _i, [_vp, _vp, _i, _i, _vp]),
    ("mhip_overlay_forward_host", _i, [_vp, _vp, _i, _i, _vp, _vp]),
    ("mhip_overlay_blend", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("mhip_trocr_default_config", _i, [_i, _vp]),
    ("mhip_trocr_max_len", _i, [_vp]),
    ("mhip_trocr_create", _i, [_vp, _i, _vp, C.POINTER(_vp)]),
    ("mhip_trocr_destroy", _i, [_vp]),
    ("mhip_trocr_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_trocr_finalize", _i, [_vp]),
    ("mhip_trocr_alloc_arena", _i, [_vp]),
    ("mhip_trocr_arena", _i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_trocr_workspace_bytes", _sz, [_vp, _i]),
    ("mhip_trocr_set_decode_gate", _i, [_vp, _vp]),
    ("mhip_trocr_generate", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    ("mhip_trocr_generate_host", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_trocr_generate_fragments", _i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    ("mhip_trocr_encode_begin", _i, [_vp, _i]),
    ("mhip_trocr_encode_fragments", _i, [_vp, _vp, _vp, _i, _i]),
    ("mhip_trocr_encoded", _i, [_vp]),
    ("mhip_trocr_decode", _i, [_vp, _vp, _vp, _vp]),
    ("mhip_trocr_generate_trace_host", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    ("mhip_cross_attention_host", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("mhip_max_page_size", _i, [_i, _i, _i, _i, C.c_double, C.POINTER(_i), C.POINTER(_i)]),
    ("mhip_resize_area_u8", _i, [_vp, _vp, _i, _i, _i, C.c_size_t, _vp, _i, _i]),
    ("mhip_resize_area_u8_host", _i, [_vp, _vp, _i, _i, _i, _vp, _i, _i]),
    ("mhip_resize_cubic_u8", _i, [_vp, _vp, _i, _i, _i, C.c_size_t, _vp, _i, _i]),
    ("mhip_resize_cubic_u8_host", _i, [_vp, _vp, _i, _i, _i, _vp, _i, _i]),
    ("mhip_merge_boxes", _i, [_vp, _i, _vp, C.POINTER(_i)]),
    ("mhip_line_merge", _i, [_vp, _i, _vp, C.POINTER(_i)]),
    ("mhip_find_line_numbers", _i, [_vp, _i, _vp, _i, _vp]),
    ("mhip_lines_from_bboxes", _i, [_vp, _i, _i, _i, _vp, _i, C.POINTER(_i)]),
    ("mhip_crnn_forward_crops", _i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_crnn_forward_fragments_host", _i, [_vp, _vp, _sz, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_crnn_create", _i, [_vp, _i, _i, C.POINTER(_vp)]),
    ("mhip_crnn_destroy", _i, [_vp]),
    ("mhip_crnn_set_tensor", _i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    ("mhip_crnn_finalize", _i, [_vp]),
    ("mhip_crnn_alloc_arena", _i, [_vp]),
    ("mhip_crnn_arena", _i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    ("mhip_crnn_seq_len", _i, [_i]),
    ("mhip_crnn_forward", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_crnn_forward_host", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mhip_crnn_workspace_bytes", _sz, [_vp, _i, _i]),
    ("mhip_crnn_kernel_flops", C.c_double, [_vp, _i, _i, _i]),
)
EXPORTED_SYMBOLS = tuple(s[0] for s in _SIGNATURES)



class ConvDesc(C.Structure):
    """mirror of ``mhip_conv_desc`` (include/marie_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "Cin", "KH", "KW", "pad", "N", "pool", "relu", "out_f32",
                                         "dil", "Cin1", "ldc", "pad_cols_writable")]


class VitConfig(C.Structure):
    """mirror of mhip_vit_config (include/marie_hip.h)"""
    _fields_ = [("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("patch", C.c_int), ("pos_h", C.c_int),
                ("pos_w", C.c_int), ("layer_scale", C.c_int), ("qkv_bias", C.c_int), ("final_norm", C.c_int),
                ("fpn", C.c_int), ("taps", C.c_int * 4), ("ln_eps", C.c_float)]


class DitConfig(C.Structure):
    """mirror of mhip_dit_config (include/marie_hip.h)"""
    _fields_ = [("model", C.c_int), ("min_size_test", C.c_int), ("max_size_test", C.c_int),
                ("detections_per_image", C.c_int), ("anchor_sizes", C.c_float * 5), ("aspect_ratios", C.c_float * 3),
                ("rpn_nms_thresh", C.c_float), ("score_thresh", C.c_float), ("nms_thresh", C.c_float)]


class TrocrConfig(C.Structure):
    """mirror of mhip_trocr_config (include/marie_hip.h)"""
    _fields_ = [("enc_dim", C.c_int), ("enc_depth", C.c_int), ("enc_heads", C.c_int), ("dec_dim", C.c_int),
                ("dec_layers", C.c_int), ("dec_heads", C.c_int), ("dec_ffn", C.c_int), ("vocab", C.c_int),
                ("max_positions", C.c_int), ("beam", C.c_int), ("max_len_b", C.c_int), ("min_len", C.c_int),
                ("pad", C.c_int), ("eos", C.c_int), ("embed_scale", C.c_float), ("img_size", C.c_int)]


class CropDesc(C.Structure):
    """mirror of ``mhip_crop_desc`` (include/marie_hip.h)"""
    _fields_ = [("src_offset", C.c_uint64), ("h", C.c_int32), ("w", C.c_int32), ("row_stride", C.c_int32),
                ("channels", C.c_int32)]


POOL_NONE, POOL_2x2, POOL_2x1 = 0, 1, 2

_lib = None


class MarieHipError(RuntimeError):
    pass


def load():
    """dlopen libmarie_hip.so and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.  If libmarie_hip.so were loaded first it would bring in
    # the system copy and a later `import torch` a second one: two HIP runtimes in one process, which intermittently
    # leaves torch with "No HIP GPUs are available".  Loading torch first makes both share torch's runtime (same
    # SONAME).  Without torch installed the system runtime is used alone.
    import importlib.util
    import sys

    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MarieHipError(
            f"{LIB_PATH} is missing — the HIP extension is not built; there is no CPU fallback "
            "(run __graft_entry__.build() or `make -C marie_icr_amd/csrc`)")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc: int, what: str):
    if rc != 0:
        msg = load().mhip_last_error(ctx_handle)
        raise MarieHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class Context:
    """One per (process, GPU).  reference counterpart: the `.to(device)` plumbing of
    marie/document/craft_ocr_processor.py:142-146."""

    def __init__(self, device_id: int = 0):
        lib = load()
        h = C.c_void_p()
        rc = lib.mhip_init(int(device_id), C.byref(h))
        if rc != 0 or not h.value:
            raise MarieHipError(f"mhip_init(device={device_id}) failed ({rc}): no usable HIP device")
        self.h = h
        self.lib = lib
        self.device_id = int(device_id)
        self._children = weakref.WeakSet()   # models created on this context; closed before the context is

    def adopt(self, child):
        self._children.add(child)

    def set_stream(self, stream_handle: int | None):
        check(self.h, self.lib.mhip_set_stream(self.h, C.c_void_p(stream_handle or 0)), "mhip_set_stream")

    def synchronize(self):
        check(self.h, self.lib.mhip_synchronize(self.h), "mhip_synchronize")

    def device_info(self):
        arch = C.create_string_buffer(64)
        cu = C.c_int()
        hbm = C.c_size_t()
        check(self.h, self.lib.mhip_device_info(self.h, arch, 64, C.byref(cu), C.byref(hbm)), "mhip_device_info")
        return {"arch": arch.value.decode(), "cu_count": cu.value, "hbm_bytes": hbm.value}

    def memcpy_dev(self, dst_ptr: int, src_ptr: int, nbytes: int):
        check(self.h, self.lib.mhip_memcpy_dev(self.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), int(nbytes)),
              "mhip_memcpy_dev")

    def conv2d_nhwc(self, precision: int, desc: "ConvDesc", in_ptr: int, w_ptr: int, scale_ptr: int, bias_ptr: int,
                    out_ptr: int, in2_ptr: int = 0):
        """Enqueue one NHWC conv/GEMM on the ctx stream; pointers are HBM addresses."""
        check(self.h, self.lib.mhip_conv2d_nhwc(self.h, int(precision), C.byref(desc), C.c_void_p(in_ptr),
                                                C.c_void_p(in2_ptr or 0), C.c_void_p(w_ptr), C.c_void_p(scale_ptr or 0),
                                                C.c_void_p(bias_ptr or 0), C.c_void_p(out_ptr)),
              "mhip_conv2d_nhwc")

    def profile_enable(self, on: bool):
        check(self.h, self.lib.mhip_profile_enable(self.h, 1 if on else 0), "mhip_profile_enable")

    def profile_reset(self):
        check(self.h, self.lib.mhip_profile_reset(self.h), "mhip_profile_reset")

    def profile_read(self):
        out = {}
        for k in range(self.lib.mhip_kernel_count()):
            ms = C.c_double()
            n = C.c_int64()
            check(self.h, self.lib.mhip_profile_read(self.h, k, C.byref(ms), C.byref(n)), "mhip_profile_read")
            fl = C.c_double()
            check(self.h, self.lib.mhip_profile_flops(self.h, k, C.byref(fl)), "mhip_profile_flops")
            out[self.lib.mhip_kernel_name(k).decode()] = {"id": k, "total_ms": ms.value, "launches": n.value,
                                                          "flops": fl.value}
        return out

    def make_gate(self) -> "PhaseGate":
        return PhaseGate(self)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            for child in list(getattr(self, "_children", ())):
                child.close()
            self.lib.mhip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PhaseGate:
    """``mhip_gate`` (include/marie_hip.h): one thread says where a phase starts in its stream (``signal``), another makes its
    own stream wait for that point (``wait``).  ``open()`` lets every waiter through — the error path."""

    def __init__(self, ctx: Context):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_gate_create(ctx.h, C.byref(h)), "mhip_gate_create")
        self.h = h
        ctx.adopt(self)

    def signal(self, ctx: Context):
        check(ctx.h, self.lib.mhip_gate_signal(self.h, ctx.h), "mhip_gate_signal")

    def count(self) -> int:
        return int(self.lib.mhip_gate_count(self.h))

    def open(self, opened: bool = True):
        self.lib.mhip_gate_open(self.h, 1 if opened else 0)

    def wait(self, ctx: Context, seq: int, timeout_ms: int = 60000) -> bool:
        rc = self.lib.mhip_gate_wait(self.h, ctx.h, int(seq), int(timeout_ms))
        if rc < 0:
            check(ctx.h, rc, "mhip_gate_wait")
        return rc == 1

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_gate_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
