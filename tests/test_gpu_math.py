"""GPU: the mm facade of the hip backend follows the reference's dtype rules
(_src/math/numpy.py:92-104): float constructors -> ftype, iarray/arange -> int64, assign copies."""

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_facade_dtypes_and_device():
    import muygpys_amd._src.math as mm

    assert mm.ndarray is torch.Tensor and mm.itype is torch.int64
    for t in (mm.ones((2, 3)), mm.zeros(4), mm.eye(3), mm.full((2,), 1.5), mm.linspace(0, 1, 5), mm.array([1, 2])):
        assert t.is_cuda and t.dtype == mm.ftype
    assert mm.arange(5).dtype == torch.int64 and mm.iarray([1.0, 2.0]).dtype == torch.int64
    x = mm.zeros((3, 3))
    y = mm.assign(x, 1.0, 0, slice(None))
    assert float(x.sum()) == 0.0 and float(y.sum()) == 3.0, "assign must copy"
    assert float(mm.sum(mm.ones((2, 3)), axis=1)[0]) == 3.0
    assert mm.parameter(2.5) == 2.5
    assert len(mm._NAMES) == 49
    for name in mm._NAMES:
        assert hasattr(mm, name), name
