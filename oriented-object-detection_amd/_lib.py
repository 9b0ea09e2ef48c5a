"""ctypes binding of libobbhip.so (include/obbhip.h).  There is NO fallback: if the HIP library is missing or a
call fails, this raises -- the product path never routes through a CPU implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libobbhip.so")
if os.environ.get("OBB_LIB"):  # diagnostic builds of the same library (tools/stamp_conv.sh): never silently
    import sys
    SO_PATH = os.environ["OBB_LIB"]
    sys.stderr.write("oriented_object_detection_amd: OBB_LIB is set -- loading the DIAGNOSTIC library %s instead of the product build\n" % SO_PATH)
_lib = None

c_dp, c_fp, c_ip, c_lp, c_bp = (C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64), C.POINTER(C.c_uint8))
_V = C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/obbhip.h one to one
SIGNATURES = {
    "obb_version": [],
    "obb_ctx_create": [C.c_int, C.POINTER(_V)],
    "obb_ctx_destroy": [_V],
    "obb_last_error": [_V],
    "obb_set_option": [_V, C.c_char_p, C.c_int64],
    "obb_poly_iou_pairs": [_V, _V, _V, C.c_int64, _V, _V],
    "obb_poly_iou_matrix": [_V, _V, _V, C.c_int64, _V, _V, C.c_int64, _V, _V],
    "obb_points_in_quads": [_V, _V, _V, C.c_int64, _V, _V, C.c_int64, _V, _V],
    "obb_build_multich": [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_sort_desc_stable": [_V, _V, C.c_int64, _V, _V],
    "obb_nms_mask": [_V, _V, _V, C.c_int64, C.c_double, _V, _V],
    "obb_nms_reduce": [_V, _V, C.c_int64, _V, _V, _V],
    "obb_merge_detections": [_V, _V, _V, _V, C.c_int64, C.c_double, _V, _V, _V, _V],
    "obb_merge_segments": [_V, _V, _V, _V, _V, C.c_int32, C.c_int64, C.c_double, _V, _V, _V],
    "obb_consensus": [_V, _V, _V, _V, c_lp, C.c_int32, C.c_double, C.c_double, C.c_double, _V, _V, _V],
    "obb_tile_grid": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_ip, C.c_int64, c_lp],
    "obb_tile_postprocess": [_V, _V, _V, _V, C.c_int64, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V, _V, _V],
    "obb_tile_survivors": [_V, _V, _V, C.c_int32, C.c_int32, _V, _V, _V, C.c_int32, C.c_int32, C.c_double, _V, _V, _V, _V],
    "obb_select_kept": [_V, _V, _V, C.c_int64, _V, _V, _V, _V, _V, _V, _V, _V, _V, _V],
    "obb_records_to_dets": [_V, _V, C.c_int64, _V, C.c_int32, _V, _V, _V, _V, _V],
    "obb_gather_compact": [_V, _V, C.c_int32, C.c_int32, _V, _V, _V],
    "obb_rotated_tal_assign": [_V, _V, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _V, _V, _V, _V, _V, _V],
    "obb_conv_dgrad_bf16": [_V, _V, c_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_conv_wgrad_bf16": [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_conv_packed_elems": [_V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)],
    "obb_conv_pack_bf16": [_V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_conv_fwd_bf16": [_V, _V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_silu_bf16": [_V, _V, _V, C.c_int64, _V],
    "obb_silu_bwd_bf16": [_V, _V, _V, _V, C.c_int64, _V],
    "obb_bias_grad_bf16": [_V, _V, C.c_int64, C.c_int32, _V, _V],
    "obb_gather_tiles": [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, C.c_int32, C.c_int32, _V, _V],
    "obb_letterbox": [_V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _V,
                      C.c_int32, C.c_int32, _V],
    "obb_model_load": [_V, _V, C.c_size_t],
    "obb_model_unload": [_V, C.c_int32],
    "obb_model_info": [_V, C.c_int32, C.c_int32, c_ip, c_ip, c_ip, c_ip],
    "obb_forward": [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_forward_gate": [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V, _V],
    "obb_debug_plan": [_V, C.c_int32, C.c_int32, C.c_char_p, C.c_int64, c_lp],
    "obb_debug_activation": [_V, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, _V, C.c_int64, c_lp, c_ip, _V],
    "obb_decode_nms": [_V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, _V, _V, _V],
    "obb_decode_nms_gate": [_V, _V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, _V, _V, _V],
    "obb_decode_nms_full": [_V, _V, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, _V, _V, _V],
    "obb_decode": [_V, _V, C.c_int32, C.c_int32, C.c_int32, _V, _V],
    "obb_probiou_nms": [_V, _V, _V, C.c_int64, C.c_float, _V, _V, _V],
    "obb_results": [_V, _V, _V, C.c_int64, _V, _V, _V],
    "obb_probiou_loss": [_V, _V, _V, _V, C.c_int64, C.c_float, _V, _V, _V],
    "obb_tile_labels": [_V, _V, C.c_int64, _V, C.c_int32, C.c_double, _V, _V, _V],
    "obb_dfl_loss": [_V, _V, _V, _V, C.c_int64, C.c_int32, C.c_float, _V, _V, _V],
    "obb_bce_loss": [_V, _V, _V, C.c_int64, C.c_float, _V, _V, _V],
    "obb_sgd_step": [_V, _V, _V, _V, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _V],
    "obb_adamw_step": [_V, _V, _V, _V, _V, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _V],
}
_RESTYPE = {"obb_last_error": C.c_char_p}


class ObbHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: the HIP extension has not been built. Run `python -c \"import __graft_entry__ as g; "
                "g.build()\"` (needs hipcc). There is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime; it must be the one in the process (tensors, streams and this library share it).  Loading
        # libobbhip.so first would pull in the system's libamdhip64 instead, and a process with both sees no device through the second.
        import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        for name, argt in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
            fn.argtypes = argt
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().obb_last_error(ctx)
        raise ObbHipError(f"libobbhip error {rc}: {msg.decode() if msg else '?'}")
