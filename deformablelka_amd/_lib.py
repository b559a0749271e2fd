"""ctypes binding of ``libdlka_hip.so`` (C-ABI in ``include/dlka.h``).

The product path has exactly one backend: the hipcc-built gfx950 library.  If it is missing the import of any
operator fails loudly with a build hint — there is no CPU or PyTorch fallback.

``_set_backend_for_tests`` exists only so that the CPU test-suite can drive the *host* logic (autograd plumbing,
module shapes, error behaviour) through ``tests/emu``'s host-compiled build of the very same kernel sources.
It is never selected automatically.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libdlka_hip.so")

DLKA_F32, DLKA_BF16, DLKA_F64 = 0, 1, 2
LKA3D_SYNAPSE, LKA3D_ACDC = 0, 1   # dlka_lka3d_variant (include/dlka.h)


class ConvGeom(ctypes.Structure):
    """``dlka_conv_geom`` (include/dlka.h)."""
    _fields_ = [(n, c_int32) for n in (
        "B", "C", "D", "H", "W", "Cout", "kd", "kh", "kw", "sd", "sh", "sw", "pd", "ph", "pw", "dd", "dh", "dw",
        "group", "deformable_group", "im2col_step")]


LKA3D_FIELDS = ("proj_1_w", "proj_1_b", "conv0_w", "conv0_b", "conv_spatial_w", "conv_spatial_b", "offset_w", "offset_b",
                "deform_w", "deform_b", "conv1_w", "conv1_b", "proj_2_w", "proj_2_b")
LKA2D_FIELDS = ("proj_1_w", "proj_1_b", "conv0_offset_w", "conv0_offset_b", "conv0_w", "conv_spatial_offset_w",
                "conv_spatial_offset_b", "conv_spatial_w", "conv1_w", "conv1_b", "proj_2_w", "proj_2_b")


class Lka3dPtrs(ctypes.Structure):
    """``dlka_lka3d_params`` / ``dlka_lka3d_grads`` (same field order)."""
    _fields_ = [(n, c_void_p) for n in LKA3D_FIELDS]


TBLOCK3D_FIELDS = ("norm_w", "norm_b", "gamma", "pos_embed", "conv51_conv1_w", "conv51_conv2_w", "conv51_norm1_w", "conv51_norm1_b",
                   "conv51_norm2_w", "conv51_norm2_b", "conv8_w", "conv8_b")


class TBlock3dPtrs(ctypes.Structure):
    """``dlka_tblock3d_params`` / ``dlka_tblock3d_grads``."""
    _fields_ = [(n, c_void_p) for n in TBLOCK3D_FIELDS]


class Lka2dPtrs(ctypes.Structure):
    """``dlka_lka2d_params`` / ``dlka_lka2d_grads``."""
    _fields_ = [(n, c_void_p) for n in LKA2D_FIELDS]


# name -> (restype, argtypes); every symbol include/dlka.h declares
_G = POINTER(ConvGeom)
SIGNATURES = {
    "dlka_abi_version": (c_int, []),
    "dlka_status_string": (c_char_p, [c_int]),
    "dlka_conv_out_size": (c_int, [c_int] * 5),
    "dlka_deform_conv3d_forward_workspace": (c_size_t, [_G, c_int]),
    "dlka_deform_conv3d_forward": (c_int, [c_void_p] * 6 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_conv3d_backward_workspace": (c_size_t, [_G, c_int]),
    "dlka_deform_conv3d_backward": (c_int, [c_void_p] * 9 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_conv3d_sample_index": (c_int, [c_void_p] * 3 + [_G, c_int, c_void_p]),
    "dlka_deform_conv3d_sample_index_path": (c_int, [c_void_p] * 3 + [_G, c_int, c_int, c_void_p]),
    "dlka_deform_conv2d_forward_workspace": (c_size_t, [_G, c_int]),
    "dlka_deform_conv2d_forward": (c_int, [c_void_p] * 6 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_conv2d_backward_workspace": (c_size_t, [_G, c_int]),
    "dlka_deform_conv2d_backward": (c_int, [c_void_p] * 9 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_conv3d_forward_workspace": (c_size_t, [_G, c_int]),
    "dlka_conv3d_forward": (c_int, [c_void_p] * 5 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_conv3d_backward_workspace": (c_size_t, [_G, c_int]),
    "dlka_conv3d_backward": (c_int, [c_void_p] * 7 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_gelu_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "dlka_gelu_backward": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "dlka_mul_forward": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "dlka_mul_backward": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_void_p]),
    "dlka_add_forward": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_void_p]),
    "dlka_lka3d_saved_bytes": (c_size_t, [c_int] * 6),
    "dlka_lka3d_workspace_bytes": (c_size_t, [c_int] * 6),
    "dlka_lka3d_attention_forward": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]
                                     + [c_int] * 6 + [c_void_p]),
    "dlka_lka3d_attention_backward": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                              POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 6 + [c_void_p]),
    "dlka_lka2d_saved_bytes": (c_size_t, [c_int] * 5),
    "dlka_lka2d_workspace_bytes": (c_size_t, [c_int] * 5),
    "dlka_lka2d_attention_forward": (c_int, [c_void_p, POINTER(Lka2dPtrs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]
                                     + [c_int] * 5 + [c_void_p]),
    "dlka_lka2d_attention_backward": (c_int, [c_void_p, POINTER(Lka2dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                              POINTER(Lka2dPtrs), c_void_p, c_size_t] + [c_int] * 5 + [c_void_p]),
    "dlka_lka2d_force_general": (c_int, [c_int]),
    "dlka_lka2d_saved_offsets": (c_int, [c_int] * 5 + [POINTER(c_size_t), POINTER(c_int)]),
    "dlka_conv3d_cl_workspace": (c_size_t, [_G, c_int, c_int]),
    "dlka_conv3d_forward_cl": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_size_t, _G, c_int, c_void_p]),
    "dlka_conv3d_backward_cl": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 4 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_conv3d_cl_workspace": (c_size_t, [_G, c_int, c_int]),
    "dlka_deform_conv3d_forward_cl": (c_int, [c_void_p] * 6 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_conv3d_backward_cl": (c_int, [c_void_p] * 9 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_ncdhw_to_ndhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dlka_ndhwc_to_ncdhw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dlka_lka3d_tokens_supported": (c_int, [c_int] * 6),
    "dlka_lka3d_tokens_saved_bytes": (c_size_t, [c_int] * 6),
    "dlka_lka3d_tokens_workspace_bytes": (c_size_t, [c_int] * 6),
    "dlka_lka3d_attention_tokens_forward": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]
                                            + [c_int] * 6 + [c_void_p]),
    "dlka_lka3d_attention_tokens_forward_prepared": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]
                                                     + [c_int] * 6 + [c_void_p]),
    "dlka_lka3d_tokens_prepare_plan_bytes": (c_size_t, [c_int]),
    "dlka_lka3d_tokens_prepare_plan": (c_int, [c_int, POINTER(Lka3dPtrs), POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int), c_int, c_void_p, c_size_t]),
    "dlka_lka3d_tokens_prepare_run": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "dlka_lka3d_tokens_prepare_run_range": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dlka_lka3d_tokens_partials_bytes_v": (c_size_t, [c_int] * 7),
    "dlka_wgrad_finalize_plan_bytes": (c_size_t, [c_int]),
    "dlka_wgrad_finalize_plan_init": (c_int, [c_void_p, c_size_t, c_int]),
    "dlka_lka3d_attention_tokens_backward_deferred_v": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                                               POINTER(Lka3dPtrs), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_int,
                                                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dlka_lka3d_attention_tokens_backward_phase_v": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                                            POINTER(Lka3dPtrs), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_int, c_int,
                                                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dlka_wgrad_finalize_plan_seal": (c_int, [c_void_p]),
    "dlka_wgrad_finalize_run_slot": (c_int, [c_void_p, c_int, c_void_p]),
    "dlka_wgrad_finalize_run": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dlka_lka3d_attention_tokens_backward": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                                     POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 6 + [c_void_p]),
    "dlka_layernorm_tokens_forward": (c_int, [c_void_p, c_int] + [c_void_p] * 6 + [c_int] * 3 + [ctypes.c_float, c_int, c_void_p]),
    "dlka_layernorm_tokens_backward": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p]),
    "dlka_scale_residual_forward": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p]),
    "dlka_scale_residual_backward": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "dlka_batchnorm_cl_forward": (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 2 + [c_int64, c_int, ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    "dlka_batchnorm_cl_backward": (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 5 + [c_int64, c_int, ctypes.c_float, c_int, c_void_p]),
    "dlka_channel_scale": (c_int, [c_void_p] * 3 + [c_int, c_int64, c_int, c_int, c_void_p]),
    "dlka_batchnorm_planar_forward": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int64, ctypes.c_float, c_void_p]),
    "dlka_batchnorm_planar_backward": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int64, c_void_p]),
    "dlka_pointwise_planar_forward": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64, c_void_p]),
    "dlka_pointwise_planar_backward": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int64, c_void_p]),
    "dlka_tblock3d_supported": (c_int, [c_int] * 6),
    "dlka_tblock3d_saved_bytes": (c_size_t, [c_int] * 6),
    "dlka_tblock3d_workspace_bytes": (c_size_t, [c_int] * 6),
    "dlka_tblock3d_forward": (c_int, [c_void_p, c_int, POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p, c_size_t] + [c_int] * 5 + [ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    "dlka_tblock3d_backward": (c_int, [POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                       POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 6 + [c_void_p]),
    "dlka_deform_conv2d_sample_index_path": (c_int, [c_void_p] * 3 + [_G, c_int, c_int, c_void_p]),
    "dlka_deform_dwconv2d_cl_workspace": (c_size_t, [_G, c_int, c_int]),
    "dlka_deform_dwconv2d_forward_cl": (c_int, [c_void_p] * 5 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_deform_dwconv2d_backward_cl": (c_int, [c_void_p] * 8 + [c_size_t, _G, c_int, c_void_p]),
    "dlka_lka3d_force_wgrad_gather": (c_int, [c_int]),
    "dlka_fork_stats": (c_int, [c_int, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64)]),
    "dlka_env_refresh": (None, []),
    "dlka_dwconv_lds_launch_count": (ctypes.c_long, []),
    "dlka_conv_brick_launch_count": (ctypes.c_long, []),
    "dlka_dwpair_launch_count": (ctypes.c_long, []),
    "dlka_dwconv_2p_launch_count": (ctypes.c_long, []),
    "dlka_conv_kw_launch_count": (ctypes.c_long, []),
    "dlka_lka3d_tokens_supported_v": (c_int, [c_int] * 7),
    "dlka_lka3d_tokens_saved_bytes_v": (c_size_t, [c_int] * 7),
    "dlka_lka3d_tokens_workspace_bytes_v": (c_size_t, [c_int] * 7),
    "dlka_lka3d_attention_tokens_forward_v": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]
                                              + [c_int] * 7 + [c_void_p]),
    "dlka_lka3d_attention_tokens_backward_v": (c_int, [c_void_p, POINTER(Lka3dPtrs), c_void_p, c_void_p, c_size_t, c_void_p,
                                                       POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 7 + [c_void_p]),
    "dlka_lka3d_tokens_saved_offsets_v": (c_int, [c_int] * 7 + [POINTER(c_size_t)]),
    "dlka_tblock3d_saved_offsets_v": (c_int, [c_int] * 7 + [POINTER(c_size_t)]),
    "dlka_tblock3d_saved_activations_v": (c_int, [c_int] * 7 + [POINTER(c_size_t)]),
    "dlka_tblock3d_supported_v": (c_int, [c_int] * 7),
    "dlka_tblock3d_saved_bytes_v": (c_size_t, [c_int] * 7),
    "dlka_tblock3d_workspace_bytes_v": (c_size_t, [c_int] * 7),
    "dlka_tblock3d_forward_v": (c_int, [c_void_p, c_int, POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p, c_size_t] + [c_int] * 5 + [ctypes.c_float, ctypes.c_float, c_int, c_int, c_void_p]),
    "dlka_tblock3d_backward_v": (c_int, [POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                         POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 7 + [c_void_p]),
    "dlka_tblock3d_backward_phase_v": (c_int, [POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                         POINTER(TBlock3dPtrs), POINTER(Lka3dPtrs), c_void_p, c_size_t] + [c_int] * 8 + [c_void_p]),
    "dlka_trace_start": (c_int, [c_int, c_void_p]),
    "dlka_trace_mark": (c_int, [c_void_p]),
    "dlka_trace_stop": (c_int, []),
    "dlka_trace_count": (c_int, []),
    "dlka_trace_get": (c_int, [c_int, ctypes.c_char_p, c_size_t, POINTER(ctypes.c_float)]),
}


def bind(cdll: ctypes.CDLL) -> ctypes.CDLL:
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if cdll.dlka_abi_version() != 1:
        raise RuntimeError("libdlka ABI version mismatch")
    return cdll


_lib = None
_test_backend = False


def get_lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"deformablelka_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C deformablelka_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        _lib = bind(ctypes.CDLL(LIB_PATH))
    return _lib


def _set_backend_for_tests(cdll) -> None:
    """TEST HOOK: route calls to an explicitly provided library (tests/emu host build). Allows CPU tensors."""
    global _lib, _test_backend
    _lib = bind(cdll) if cdll is not None else None
    _test_backend = cdll is not None


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = get_lib().dlka_status_string(rc).decode()
        raise RuntimeError(f"{what}: {msg} (dlka status {rc})")


def dtype_code(t: torch.Tensor, allow_f64: bool = False) -> int:
    """allow_f64: the general NCDHW operators (D3D.deform_conv_forward / backward, the torchvision-style 2-D op, the plain conv3d) also take float64, as the
    reference's op does (AT_DISPATCH_FLOATING_TYPES: float, double — deform_conv_cuda.cu:96,233); the fused blocks and channels-last fast paths do not."""
    if t.dtype == torch.float32:
        return DLKA_F32
    if t.dtype == torch.bfloat16:
        return DLKA_BF16
    if t.dtype == torch.float64 and allow_f64:
        return DLKA_F64
    raise RuntimeError(f"deformablelka_amd supports float32 and bfloat16 tensors (float64: the general deform_conv / conv3d operators only), got {t.dtype}")


def require_device(*tensors) -> None:
    for t in tensors:
        if t is None:
            continue
        if _test_backend:
            if t.device.type != "cpu":
                raise RuntimeError("test backend wants CPU tensors")
        elif not t.is_cuda:
            # reference: AT_ASSERTM(input.type().is_cuda(), "input must be a CUDA tensor") deform_conv_cuda.cu:44-47
            raise RuntimeError("deformablelka_amd: tensors must live on an AMD GPU (cuda/HIP device); "
                               "the operator is not implemented on the CPU")


def stream_ptr(t: torch.Tensor):
    if _test_backend:
        return None
    # The library launches on the CURRENT device (HIP's rule) with the handles of that device (include/dlka.h: fork contexts are per device): a tensor of
    # another device is a caller error that must not reach a launch.  nn.DataParallel replicas and autograd's device threads run with their device current.
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"deformablelka_amd: tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()} — "
                           f"call under torch.cuda.device({t.device.index})")
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def scratch(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    """Workspace from the caching allocator (stream-ordered, graph-capture safe)."""
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=like.device)
