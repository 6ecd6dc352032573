"""No-GPU checks of the C-ABI boundary: the library loads and exports every symbol include/actmi.h declares."""
import os

from actmi import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = L.declared_symbols(os.path.join(ROOT, "include", "actmi.h"))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.actmi_version() == 110


def test_struct_sizes_match_header_layout():
    import ctypes as C
    # actmi_config: 16 int32 + 1 float
    assert C.sizeof(L.ActmiConfig) == 21 * 4          # struct_size, 16 ints, kl_weight, vq / vq_class / vq_dim
    # descriptors: natural alignment, no packing pragmas on either side
    assert C.sizeof(L.GemmDesc) % 8 == 0 and C.sizeof(L.AttnDesc) % 8 == 0


def test_null_arguments_are_rejected_not_crashing():
    lib = L.load()
    assert lib.actmi_create(None, None) != 0
    assert lib.actmi_destroy(None) == 0
    assert lib.actmi_op_gemm(None, None) != 0
