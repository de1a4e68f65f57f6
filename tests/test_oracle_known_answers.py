"""Pins the CPU oracle on the reference-run known answers recorded in BASELINE.md section 2 / SURVEY.md 4, 8(c)
(tests/golden/baseline_known_answers.json).  CPU only."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "baseline_known_answers.json")))


def serial_sum(a):
    s = 0.0
    for x in a.ravel().tolist():
        s += x
    return s


def test_weno_ideal_weights(oracle):
    w = oracle.weno5_ideal_weights()
    ref = KA["weno5_ideal_weights"]
    assert w[0] == ref["idl_L"] and w[1] == ref["idl_C"] and w[2] == ref["idl_R"] and w[3] == ref["idl_H"]


def test_weno_step_stencil_falls_back_to_left_stencil(oracle):
    coefs, gll = oracle.weno5(KA["weno5_step_stencil"]["stencil"])
    ref = KA["weno5_step_stencil"]["coefs_approx"]
    assert coefs[0] == 1.0
    for c, r in zip(coefs[1:], ref[1:]):
        assert abs(c - r) <= 0.06 * abs(r)            # recorded to two significant digits
    assert gll[0] == 1.0 and gll[1] == 1.0


def _cell_averages(poly, x0=-2.5):
    """cell averages of a polynomial (coefficients low->high) over 5 unit cells centred at -2..2"""
    P = np.polynomial.Polynomial(poly).integ()
    return np.array([P(x0 + i + 1) - P(x0 + i) for i in range(5)])


def test_weno_polynomial_exactness(oracle):
    rng = np.random.default_rng(3)
    for deg in (0, 1, 2):
        for _ in range(20):
            poly = rng.normal(size=deg + 1)
            s = _cell_averages(poly)
            _, gll = oracle.weno5(s)
            P = np.polynomial.Polynomial(poly)
            err = max(abs(gll[0] - P(-0.5)), abs(gll[1] - P(0.5))) / max(1.0, np.max(np.abs(s)))
            assert err <= 4 * KA["weno5_polynomial_exactness"]["degree_le_2_max_error"]
    # degree 3 and 4 at O(1) amplitude are deliberately NOT reproduced (the limiter engages)
    for deg, order in ((3, KA["weno5_polynomial_exactness"]["degree_3_error_order"]), (4, KA["weno5_polynomial_exactness"]["degree_4_error_order"])):
        poly = np.zeros(deg + 1); poly[deg] = 1.0
        s = _cell_averages(poly)
        _, gll = oracle.weno5(s)
        P = np.polynomial.Polynomial(poly)
        err = max(abs(gll[0] - P(-0.5)), abs(gll[1] - P(0.5)))
        assert 0.2 * order <= err <= 5 * order


def test_cfl_time_steps(oracle):
    p, _ = oracle.make_params(32, 32, 16, 1, 16000., 16000., 20000.)
    assert oracle.lib().mwo_compute_time_step(p) == KA["dt_32x32x16"]
    p, _ = oracle.make_params(200, 200, 50, 1, 1.0e5, 1.0e5, 2.0e4)
    assert oracle.lib().mwo_compute_time_step(p) == KA["dt_config1_200x200x50"]


def test_supercell_32x32x16_three_steps_bitwise(oracle):
    ref = KA["supercell_32x32x16_3steps"]
    dyc, f = oracle.supercell_setup(32, 32, 16, 1, 16000., 16000., 20000.)
    dt = dyc.compute_time_step()
    assert serial_sum(f.rho_d) == ref["density_dry_sum_init"]
    for _ in range(3):
        dyc.time_step(f, dt)
    assert float(f.wvel.max()) == ref["wvel_max"]
    assert float(f.wvel.min()) == ref["wvel_min"]
    assert float(f.temp.max()) == ref["temp_max"]
    assert serial_sum(f.rho_d) == ref["density_dry_sum_after"]


def test_supercell_initial_perturbations_vanish(oracle):
    """rho' and (rho theta)' are exactly 0 after init_supercell."""
    dyc, f = oracle.supercell_setup(12, 12, 10, 1, 6000., 6000., 20000., perturb=False)
    hy = dyc.hy()
    rho = f.rho_d + f.tracers[0]
    assert np.array_equal(rho[:, 0, 0, 0], rho[:, 3, 5, 0])
    # rho == hy_dens_cells exactly wherever the sum rho_d + rho_v rounds back (mass is carried by rho_d = rho - rho_v)
    assert np.max(np.abs(rho[:, 0, 0, 0] - hy["hy_dens_cells"][:, 0])) <= 2e-16 * hy["hy_dens_cells"].max()
    assert np.all(f.vvel == 0) and np.all(f.wvel == 0)


def test_fp_literals_are_double_rounding_safe():
    """Every `x_fp` literal of the reference goes long double -> double (main_header.h:61-63).  The HIP kernels use plain
    double literals; this checks both routes give the same bits for every literal the oracle uses."""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "oracle", "mw_oracle.cpp")).read()
    lits = sorted(set(re.findall(r"FP\(([0-9.eE+-]+)\)", src)))
    assert len(lits) > 40
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    for l in lits:
        assert float(np.longdouble(l)) == float(l), l


def test_weno3_limiter_properties(oracle):
    """WenoLimiter<3> of the -DMW_ORD=3 oracle build: ideal weights (1, 1, 500)/502, exact for polynomials of degree <= 1 ... the
    high-order candidate reproduces a parabola's edge values when the data are smooth, and a step keeps the edges between the
    neighbouring cell values."""
    import numpy as np
    O3 = oracle.with_order(3)
    assert O3.lib().mwo_order() == 3 and oracle.lib().mwo_order() == 5
    coefs, gll = O3.weno5(np.array([2.0, 2.0, 2.0]))
    assert np.allclose(coefs, [2.0, 0.0, 0.0]) and np.allclose(gll, [2.0, 2.0])
    # cell averages of a straight line: the edge values are exact
    coefs, gll = O3.weno5(np.array([1.0, 2.0, 3.0]))
    assert np.allclose(gll, [1.5, 2.5], atol=1e-12)
    # a step: no new extrema
    coefs, gll = O3.weno5(np.array([0.0, 0.0, 1.0]))
    assert -1e-12 <= gll[0] <= 1.0 and -1e-12 <= gll[1] <= 1.0


def test_random_temperature_perturbation_of_the_oracle(oracle):
    """perturb_temperature(random = true), perturb_temperature.h:25-39, with splitmix64 for yakl::Random (pinned by definition):
    lowest nz/4 levels only, |noise| <= 3 K fading linearly with height, reproducible, different on another rank."""
    import numpy as np
    dyc, f = oracle.supercell_setup(10, 8, 16, 2, 5000., 4000., 20000., perturb=False)
    a = np.zeros_like(f.temp); b = np.zeros_like(f.temp); c = np.zeros_like(f.temp)
    oracle.perturb_temperature(dyc.p, a, thermal=False, random=True)
    oracle.perturb_temperature(dyc.p, b, thermal=False, random=True)
    oracle.perturb_temperature(dyc.p, c, thermal=False, random=True, myrank=2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.all(a[4:] == 0)
    for k in range(4):
        lim = 3.0 * (4 - k) / 4
        assert np.abs(a[k]).max() <= lim and np.abs(a[k]).max() > 0.5 * lim
    # the first draw, by hand: key 0 through the splitmix64 finaliser
    z = (0 + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    z = z ^ (z >> 31)
    u01 = (z >> 11) / 9007199254740992.0
    assert a.reshape(16, -1)[0, 0] == (u01 * 2.0 - 1.0) * 3.0 * 1.0


@pytest.mark.parametrize("order", [7, 9])
def test_weno_7_and_9_limiter_properties(oracle, order):
    """WenoLimiter<7> / <9> of the -DMW_ORD=7 / 9 oracle builds.  Independent of the reference: constants, cell averages of a
    polynomial of degree <= 2 are reproduced exactly at both edges (all four candidates agree, whatever the weights), smooth data give
    the full-order edge values (the high-order candidate carries ~1 of the weight), a step creates no new extremum."""
    import numpy as np
    from numpy.polynomial import polynomial as P
    Oo = oracle.with_order(order)
    assert Oo.lib().mwo_order() == order
    h = (order - 1) // 2
    xs = np.arange(-h, h + 1, dtype=np.float64)

    def averages(c):                                            # cell averages of sum c_m x^m over the unit cells centred at xs
        ci = P.polyint(c)
        return P.polyval(xs + 0.5, ci) - P.polyval(xs - 0.5, ci)

    coefs, gll = Oo.weno5(np.full(order, 3.0))
    assert np.allclose(coefs, [3.0] + [0.0] * (order - 1), atol=1e-14) and np.allclose(gll, [3.0, 3.0], atol=1e-14)
    c = np.array([0.7, -1.3, 0.4])
    coefs, gll = Oo.weno5(averages(c))
    assert np.allclose(gll, [P.polyval(-0.5, c), P.polyval(0.5, c)], rtol=0, atol=1e-12)
    c = np.array([1.0, 0.3, -0.02, 0.003, 5e-4, -2e-4, 1e-4, 3e-5, -1e-5][:order])            # smooth: the curvature terms are small
    coefs, gll = Oo.weno5(averages(c))
    assert np.allclose(gll, [P.polyval(-0.5, c), P.polyval(0.5, c)], rtol=0, atol=1e-7)     # limited: next to, not at, full order
    low = Oo.weno5(averages(np.concatenate([c[:3], np.zeros(order - 3)])))[1]                # what a parabola alone would give
    assert np.max(np.abs(low - [P.polyval(-0.5, c), P.polyval(0.5, c)])) > 1e-4
    s = np.zeros(order); s[h + 1:] = 1.0
    coefs, gll = Oo.weno5(s)
    assert -1e-10 <= gll[0] <= 1.0 + 1e-10 and -1e-10 <= gll[1] <= 1.0 + 1e-10


def test_weno_7_9_tables_equal_the_reference_literals():
    """tools/gen_weno_tables.py derives the 155 fit / total-variation constants and the 7- and 9-point Gauss-Lobatto rules; where the
    reference tree is at hand (this container, not the GPU box) every one of them -- and the order of the terms -- is compared with the
    reference's `_fp` literals as doubles."""
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/model/modules/helpers"):
        pytest.skip("reference tree not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_weno_tables.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "DIFFERENT" not in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("identical") == 8
