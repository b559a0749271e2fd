"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Loader for ``oracle/_ref/D3D.so``: the reference's own pybind11 module ``D3D`` (3D/dcn/src/vision.cpp:4-7) built
unmodified from /root/reference by ``oracle/ref.mk`` (hipcc, gfx950).  It runs only on a GPU (its CPU branch is
``AT_ERROR``, 3D/dcn/src/deform_conv.h:46,90), so it is used by ``-m gpu`` tests and by
``tests/golden/make_ref_golden.py`` which records its outputs as committed fixtures for the CPU suite.
"""
import importlib.util
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "D3D.so")
REFERENCE_SRC = "/root/reference/3D/dcn/src"
_mod = None


def build() -> str:
    """Compile oracle/_ref/D3D.so (needs /root/reference: this container only — the GPU box uses the prebuilt file)."""
    if not os.path.isdir(REFERENCE_SRC):
        raise RuntimeError(f"{REFERENCE_SRC} is absent: oracle/_ref can only be built where the reference is mounted")
    subprocess.check_call(["make", "-C", _HERE, "-f", "ref.mk", "-s", "-j3"])
    return SO


def available() -> bool:
    return os.path.exists(SO)


def D3D():
    """The reference's native module.  ``import torch`` first (it resolves libtorch/libc10 for the extension)."""
    global _mod
    if _mod is None:
        import torch  # noqa: F401
        if not available():
            raise RuntimeError(f"{SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                               "/root/reference exists")
        spec = importlib.util.spec_from_file_location("D3D", SO)
        _mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mod)
    return _mod


def deform_conv3d_forward(input, weight, bias, offset, stride, padding, dilation, group=1, deformable_groups=1, im2col_step=64):
    """``D3D.deform_conv_forward`` with the argument order of 3D/dcn/functions/deform_conv_func.py:26-33."""
    k = tuple(weight.shape[2:5])
    return D3D().deform_conv_forward(input, weight, bias, offset, *k, *stride, *padding, *dilation, group, deformable_groups,
                                     im2col_step)


def deform_conv3d_backward(input, weight, bias, offset, grad_output, stride, padding, dilation, group=1, deformable_groups=1,
                           im2col_step=64):
    """``D3D.deform_conv_backward`` (deform_conv_func.py:43-51) -> [grad_input, grad_offset, grad_weight, grad_bias]."""
    k = tuple(weight.shape[2:5])
    return D3D().deform_conv_backward(input, weight, bias, offset, grad_output, *k, *stride, *padding, *dilation, group,
                                      deformable_groups, im2col_step)
