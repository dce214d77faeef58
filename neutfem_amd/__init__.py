"""neutfem_amd -- MI355X (gfx950) native hot path of jujuC31/NeutFEM.

What lives here (and nothing else):
  csrc/          hand-written HIP kernels + the C ABI (include/neutfem_hip.h) + the pybind11 host module
  lib/           libneutfem_hip.so            (built in-tree by __graft_entry__.build())
  neutfem/       _neutfem_eigen.*.so          drop-in for the reference's Python module
  capi.py        ctypes binding of the C ABI (tests, bench)
  cases.py       benchmark input generators (IAEA-3D resampled meshes, synthetic checkerboard)
  shims/         stand-ins for plotting packages the reference *drivers* import (seaborn)

`install_compat()` puts `neutfem/` (namespace package, exactly like the reference's layout,
Makefile:24,51) on sys.path so `import neutfem._neutfem_eigen` resolves to the HIP-backed module.
There is no CPU fallback anywhere in this package.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def install_compat(with_shims=False):
    # NEUTFEM_MODULE_DIR: directory holding another build of neutfem/_neutfem_eigen (the ASan/UBSan build of `make test-asan`)
    alt = os.environ.get("NEUTFEM_MODULE_DIR")
    if alt and alt not in sys.path:
        sys.path.insert(0, alt)
    if _HERE not in sys.path:
        sys.path.insert(1 if alt else 0, _HERE)
    if with_shims:
        shim = os.path.join(_HERE, "shims")
        if shim not in sys.path:
            sys.path.append(shim)


def lib_path():
    """libneutfem_hip.so of this tree; NEUTFEM_HIP_LIB names another build of the same C ABI (A/B runs of two kernel versions in
    profiles/tools/: the file must exist, there is no fallback)"""
    return os.environ.get("NEUTFEM_HIP_LIB") or os.path.join(_HERE, "lib", "libneutfem_hip.so")
