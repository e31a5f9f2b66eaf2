"""CPU: the C oracle against the golden vectors captured from the reference (tier T1: tree logic)."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P


@pytest.mark.parametrize("name", P.T1_NAMES)
def test_oracle_matches_reference_tree(name):
    case, z = P.load_case(name)
    out = P.run_case(O.OracleEngine, case, z)
    # visit counts, parents, flags: bit-exact.  float64 statistics: the golden env is numpy/libm float64, the
    # oracle's sin/cos are azg_math.h's (<= 1 ulp apart), hence a 1e-12 tolerance rather than equality.
    P.compare_rows(out, z, float_tol=1e-12)
