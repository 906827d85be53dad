"""CPU: the C-ABI library loads and exports every symbol include/gla.h declares; no compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gla.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gla_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from gnn_applied_linear_algebra_amd import _lib
    names = _declared()
    assert len(names) >= 25
    handle = _lib.lib()                                   # raises if the .so is missing
    for name in names:
        assert hasattr(handle, name), f"{name} declared in gla.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and gla.h disagree"
    assert handle.gla_version() >= 100
    assert handle.gla_error_string(-2).decode() == "edge index out of range"


def test_argument_validation_without_gpu():
    """Pure host-side argument checks of the C ABI (return codes, no device work)."""
    import ctypes
    from gnn_applied_linear_algebra_amd import _lib
    h = _lib.lib()
    assert h.gla_laplacian2d_nnz(4096, 4096, 0, 4096 * 4096) == 5 * 4096 * 4096 - 4 * 4096
    assert h.gla_laplacian2d_nnz(5, 3, 0, 15) == 5 * 15 - 2 * 5 - 2 * 3
    assert h.gla_laplacian2d_nnz(5, 3, 4, 11) == sum(5 - (i == 0) - (i == 4) - (j == 0) - (j == 2)
                                                     for k in range(4, 11) for i, j in [(k % 5, k // 5)])
    assert h.gla_laplacian2d_nnz(0, 3, 0, 0) == -1
    out = ctypes.c_void_p()
    assert h.gla_csr_create_from_coo(-1, 1, 0, None, None, None, ctypes.byref(out)) == -1
    assert h.gla_csr_create_from_coo(4, 4, 3, None, None, None, ctypes.byref(out)) == -1
    assert h.gla_csr_create_from_coo(2 ** 31, 4, 0, None, None, None, ctypes.byref(out)) == -3
    assert h.gla_spmm(None, 1, 1, None, 1, None, 1, 0, None) == -1
    assert h.gla_set_option(b"no_such_option", 1) == -1
    assert h.gla_dot_work_bytes(10, 16, 1) == 1024 * 16 * 8


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (a product path through the oracle voids parity)."""
    pkg = os.path.join(ROOT, "gnn_applied_linear_algebra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "reference_port" not in src and "c_oracle" not in src and "gla_oracle" not in src, f


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gnn_applied_linear_algebra_amd as gla
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(gla._lib.GlaError):
        gla.JacobiGNN.JacobiGNN()(1, torch.ones(2, 3, dtype=torch.float64), ei, torch.ones(2, 2, dtype=torch.float64),
                                  torch.tensor([0.7]))
    with pytest.raises(gla._lib.GlaError):
        gla.UtilsGNN.laplacianfun_torch(4)


def test_header_is_valid_c(tmp_path):
    """include/gla.h is a C header (the boundary is a C ABI): a C11 translation unit that includes it and takes the
    address of every declared entry point compiles with gcc -std=c11 -Wall -Werror."""
    import re
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "gla.h")).read()
    names = sorted(set(re.findall(r"\b(gla_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 40
    src = tmp_path / "use_gla.c"
    src.write_text('#include "gla.h"\n#include <stddef.h>\n'
                   + "typedef void (*fn_t)(void);\n"
                   + "fn_t table[] = {" + ", ".join(f"(fn_t){n}" for n in names) + "};\n"
                   + "size_t n_entries(void) { return sizeof(table) / sizeof(table[0]); }\n")
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", "-c", str(src),
                    "-I" + os.path.join(root, "include"), "-o", str(tmp_path / "use_gla.o")], check=True)


def test_csr_info_struct_layout_matches_the_header(tmp_path):
    """gla_csr_info_t as the C compiler lays it out (include/gla.h) == the ctypes mirror the Python host passes to gla_csr_info:
    same size, same offset for every field (fields are appended round by round; a field added on one side only would shift
    everything the library writes)."""
    import ctypes
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from gnn_applied_linear_algebra_amd._lib import CsrInfo
    fields = [f[0] for f in CsrInfo._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include "gla.h"\n#include <stddef.h>\n#include <stdio.h>\nint main(void) {\n'
                   + '  printf("%zu\\n", sizeof(gla_csr_info_t));\n'
                   + "".join(f'  printf("%zu\\n", offsetof(gla_csr_info_t, {f}));\n' for f in fields)
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", "-Wall", "-Werror", str(src), "-I" + os.path.join(root, "include"), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(CsrInfo), (out[0], ctypes.sizeof(CsrInfo))
    assert out[1:] == [getattr(CsrInfo, f).offset for f in fields]
