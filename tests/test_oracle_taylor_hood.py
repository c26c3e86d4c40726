"""Pins of the mixed blocks of the P2-P1 Taylor-Hood Stokes operator (hyteg_amd/host/taylorhood.hpp): the closed-form element
matrices the host layer hands to the P2 kernel -- div (P2 -> P1) padded with zero edge rows, divT (P1 -> P2) padded with zero edge
columns -- against the reference's FEniCS forms compiled in place (oracle/_ref: p2_to_p1_tet_div_tet.h, p1_to_p2_tet_divt_tet.h,
as src/mixed_operator/P2ToP1ConstantOperator.hpp:90-97 and P1ToP2ConstantOperator.hpp use them).  No GPU needed."""
import numpy as np
import pytest

from conftest import SKEW_TET
from oracle import p1_oracle as po

TETS = [np.asarray(SKEW_TET, dtype=np.float64).reshape(12), np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0]),
        np.array([0.1, 0.2, 0.0, 1.0, 0.1, 0.3, 0.2, 1.1, 0.0, 0.3, 0.2, 0.9])]


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_mixed_forms_are_the_reference_fenics_forms(which, k):
    from hyteg_amd import host

    ref = po.ref_fenics()
    if ref is None or not hasattr(ref, "ref_p2_to_p1_tet_div"):
        pytest.skip("oracle/_ref was not built (the reference is not mounted)")
    host.lib()
    for co in TETS:
        M = host.taylor_hood_form_element_matrix(which, k, co)
        R = po.ref_taylor_hood_block(ref, co, which, k)
        assert np.abs(R).max() > 1e-3
        assert np.abs(M - R).max() < 1e-14
        # the padding: div has no edge rows, divT no edge columns
        assert np.all(M[4:, :] == 0.0) if which == 0 else np.all(M[:, 4:] == 0.0)


def test_div_and_divt_are_transposes():
    from hyteg_amd import host

    host.lib()
    for co in TETS:
        for k in range(3):
            assert np.abs(host.taylor_hood_form_element_matrix(0, k, co) - host.taylor_hood_form_element_matrix(1, k, co).T).max() < 1e-15
