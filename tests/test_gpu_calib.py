"""shf_calib_matrix_pipe (include/shf_hip.h, measurement): the matrix pipe's sustained rate depends on the operands --
constant operands reach the nominal fp16 MFMA peak, random ones run into the power limit, zeros give some of it back."""
import ctypes as C

import pytest


@pytest.mark.gpu
def test_matrix_pipe_calibration_orders_by_operand_toggling():
    from smallhardface_amd import _lib
    L = _lib.load()

    def rate(bf, zero8, const):
        v = C.c_double(0.0)
        # (long enough for the power management to settle: ~10 ms launches, 3 settling + 6 timed)
        _lib.check(L.shf_calib_matrix_pipe(bf, zero8, const, 20000, 6, C.byref(v)))
        return v.value
    const, rnd, half, allz = rate(0, 0, 1), rate(0, 0, 0), rate(0, 4, 0), rate(0, 8, 0)
    assert 1800.0 < const < 2600.0                 # 2.5 PFLOP/s dense fp16 at 2.4 GHz (MI355X_MICROARCH.md)
    # the order of the rows (a few % of slack: boxes and their thermal state differ), and the size of the effect
    assert rnd < 1.03 * half and half < 1.03 * allz and allz < 1.03 * const and rnd < 0.9 * const, (const, rnd, half, allz)
    assert rate(1, 0, 0) > 0.0                     # bf16 instantiation runs
    assert L.shf_calib_matrix_pipe(0, 9, 0, 10, 1, C.byref(C.c_double())) != 0    # bad arguments are refused
    assert "calib_matrix_pipe" in _lib.last_error()
