"""CPU: the C-ABI library builds, loads and exports every symbol include/*.h declares."""

import os


def test_library_exports_header_symbols():
    from muygpys_amd import _lib, build

    path = build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    names = _lib.exported_names_from_header()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/muygpys_hip.h but not exported"
    assert b"gfx950" in lib.mgp_version()
    assert lib.mgp_max_nn_count(4, 1) >= 100
    assert lib.mgp_max_nn_count(8, 1) >= 64


def test_bad_arguments_are_rejected_without_a_gpu():
    from muygpys_amd import _lib

    lib = _lib.load()
    # null pointers / bad sizes are refused before any HIP call is made
    rc = lib.mgp_posterior_f32(None, None, 4, None, None, 10, 5, None, 1, 0, 0.0, None, 2, 0, None, 1,
                               None, None, None, None, None)
    assert rc == -1
    rc = lib.mgp_kernel_apply_f64(None, 10, 99, 1.0, None, None)
    assert rc == -1
