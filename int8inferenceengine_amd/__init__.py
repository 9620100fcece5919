"""int8inferenceengine_amd -- MI355X-native INT8 inference hot path behind the
reference's `i8ie` Python surface.

Importing this package makes the reference's two import names resolvable:

    import int8inferenceengine_amd      # once
    import i8ie                         # drop-in: i8ie.tensor / Conv2d / Linear / Module / quantize ...
    import _CXX_i8ie                    # the rebuilt pybind11 extension (HIP backend via include/i8ie_hip.h)

The extension is built in-tree by `python -m int8inferenceengine_amd.build`
(or `__graft_entry__.build()`).  There is no CPU fallback: if the extension or
libi8ie_hip.so is missing the import fails loudly.
"""
import os
import sys

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
if _PKG_DIR not in sys.path:
    sys.path.insert(0, _PKG_DIR)

# One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so and
# libtorch_hip.so asks for it by the unversioned name, so if libi8ie_hip.so pulled
# in /opt/rocm's libamdhip64.so.7 first, a later `import torch` would load a second
# runtime and GPU initialisation would fail ("no ROCm-capable device").  Loading
# torch first makes libi8ie_hip.so resolve libamdhip64.so.7 to torch's copy (same
# SONAME), which also lets the two share device pointers, streams and RCCL.
# Set I8IE_NO_TORCH_PRELOAD=1 for a torch-free process.
if "torch" not in sys.modules and not os.environ.get("I8IE_NO_TORCH_PRELOAD"):
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def native_library_path():
    """Path of the C-ABI shared library (for ctypes / other FFI users)."""
    return os.path.join(_PKG_DIR, "libi8ie_hip.so")


def _require_native():
    import glob

    if not os.path.exists(native_library_path()) or not glob.glob(os.path.join(_PKG_DIR, "_CXX_i8ie*.so")):
        raise ImportError(
            "int8inferenceengine_amd: native extension not built; run "
            "`python -m int8inferenceengine_amd.build` (needs hipcc, gfx950 target)")


__all__ = ["native_library_path"]
