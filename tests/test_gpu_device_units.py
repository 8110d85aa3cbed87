"""GPU-tier unit tests of the three device-only pieces the host emulation replaces by plain C++ (tests/emu):
the Newton pivot square root on v_rsq_f64, the inline-asm LDS batch-read helpers, and the fp64 MFMA tile with the
lane layout the blocked Cholesky feeds it.  Each is run in isolation (tests/gpu_unit/cmpc_device_unit.hip) against
values computed on the host."""
import ctypes

import numpy as np
import pytest
import torch

import build as _b

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def unit():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    lib = ctypes.CDLL(_b.build_device_unit())
    vp = ctypes.c_void_p
    lib.unit_pivot_sqrt.argtypes = [vp, vp, vp, ctypes.c_int]
    lib.unit_lds_helpers.argtypes = [vp]
    lib.unit_mfma_tile.argtypes = [vp, vp, vp]
    return lib


def test_pivot_sqrt_matches_ieee_sqrt(unit):
    """Pivots range from the 1e-8 acceptance floor to barrier-inflated 1e+15: sqrt to 1 ulp, 1/sqrt to 2 ulp."""
    rng = np.random.default_rng(0)
    p = np.concatenate([10.0 ** rng.uniform(-8, 15, size=200000), [1e-8, 1.0, 4.0, 2.0 ** 52, 1e15, 3e-7]])
    d_p = torch.from_numpy(p).to("cuda:0")
    d_s, d_i = torch.empty_like(d_p), torch.empty_like(d_p)
    assert unit.unit_pivot_sqrt(d_p.data_ptr(), d_s.data_ptr(), d_i.data_ptr(), p.size) == 0
    s, inv = d_s.cpu().numpy(), d_i.cpu().numpy()
    ref = np.sqrt(p)
    assert (np.abs(s - ref) <= np.spacing(ref)).all()
    assert (np.abs(inv - 1.0 / ref) <= 2 * np.spacing(1.0 / ref)).all()
    assert np.abs(s * inv - 1.0).max() < 1e-15


def test_lds_batch_read_helpers(unit):
    bad = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    assert unit.unit_lds_helpers(bad.data_ptr()) == 0
    assert int(bad.item()) == 0


def test_mfma_f64_tile_layout(unit):
    rng = np.random.default_rng(1)
    A, B = rng.normal(size=(16, 16)), rng.normal(size=(16, 16))
    d_a, d_b = torch.from_numpy(A).to("cuda:0"), torch.from_numpy(B).to("cuda:0")
    d_d = torch.zeros((16, 16), dtype=torch.float64, device="cuda:0")
    assert unit.unit_mfma_tile(d_a.data_ptr(), d_b.data_ptr(), d_d.data_ptr()) == 0
    want = A @ B.T
    assert np.abs(d_d.cpu().numpy() - want).max() < 1e-13
