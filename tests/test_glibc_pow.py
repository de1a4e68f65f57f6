"""The strict kernel path's pow (miniweatherml_amd/csrc/mw_glibc_pow.h): glibc's algorithm and tables restated so that the device
produces the BITS of the host libm's pow -- which is what the CPU oracle, and a reference built with the YAKL serial backend, computes
`pow` with (dynamics_euler_stratified_wenofv.h:401, :1935, :2009).

CPU part (no GPU): the product header compiled for the host (oracle/pow_check.cpp) against std::pow of the running libm, bit for
bit, on millions of arguments; the generated table header against the installed libm.  GPU part: the device routine against the
same libm."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAMMA = 1003.0 / 716.0


def argument_sets(n, seed):
    rng = np.random.default_rng(seed)
    return {
        "(rho theta)^gamma (:401)": (rng.uniform(20.0, 500.0, n), np.full(n, GAMMA)),
        "(p/C0)^(1/gamma) (:2009)": (rng.uniform(1e-5, 1e-1, n), np.full(n, 1.0 / GAMMA)),
        "other gammas": (rng.uniform(1.0, 400.0, n), rng.uniform(1.2, 1.7, n)),
        "exner^(cp/R) (:1113)": (rng.uniform(0.2, 1.0, n), np.full(n, 1003.0 / 287.0)),
        "cos^2 (perturb_temperature.h:63)": (rng.uniform(1e-8, 1.0, n), np.full(n, 2.0)),
        "general": (np.exp(rng.uniform(-30.0, 30.0, n)), rng.uniform(-9.0, 9.0, n)),
        "wide x": (np.exp(rng.uniform(-700.0, 700.0, n)), rng.uniform(-1.5, 1.5, n)),
        "wide y": (rng.uniform(0.5, 2.0, n), np.exp(rng.uniform(-50.0, 45.0, n)) * rng.choice([-1.0, 1.0], n)),
        "near one": (1.0 + rng.uniform(-1e-3, 1e-3, n), rng.uniform(-1e6, 1e6, n)),
    }


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_host_build_of_the_restatement_is_bit_identical_to_libm(oracle):
    libm_pow, restated = oracle.powcheck()
    total = 0
    for name, (x, y) in argument_sets(1_000_000, 11).items():
        ref = libm_pow(x, y)
        got, main = restated(x, y)
        bad = (bits(ref) != bits(got)) & main
        assert not bad.any(), "%s: %d of %d main-path results differ from libm, e.g. pow(%r, %r) = %r vs %r" % (
            name, int(bad.sum()), int(main.sum()), x[bad][0], y[bad][0], ref[bad][0], got[bad][0])
        total += int(main.sum())
        if name.split()[0] in ("(rho", "(p/C0)^(1/gamma)", "other", "exner^(cp/R)", "cos^2", "general"):
            assert main.all(), name                                # everything the dycore produces is on the restated path
    assert total > 7_000_000


def test_host_build_of_the_exp_restatement_is_bit_identical_to_libm(oracle):
    """glibc's exp (the strict Kessler path's saturation vapour pressure, microphysics_kessler.h:304): same table, same polynomial."""
    libm_exp, restated = oracle.expcheck()
    rng = np.random.default_rng(21)
    for name, x in (("17.27 (T - 273) / (T - 36)", rng.uniform(-45.0, 12.0, 2_000_000)), ("wide", rng.uniform(-720.0, 720.0, 2_000_000)),
                    ("small", rng.uniform(-1e-3, 1e-3, 1_000_000) * rng.uniform(0.0, 1.0, 1_000_000))):
        ref = libm_exp(x)
        got, main = restated(x)
        bad = (bits(ref) != bits(got)) & main
        assert not bad.any(), "%s: %d results differ from libm" % (name, int(bad.sum()))
        if name != "wide":
            assert main.mean() > 0.999, name
    _, main = restated(np.array([0.0, 1e-300, 600.0, -800.0, np.inf, np.nan]))
    assert not main.any()


def test_host_build_of_the_cos_restatement_is_bit_identical_to_libm(oracle):
    """glibc's cos on [0, pi/2] (the cosine bells of the initial states) and on the rest of its two table / Taylor paths."""
    libm_cos, restated = oracle.expcheck("cos")
    rng = np.random.default_rng(31)
    for name, x, full in (("[0, pi/2]", rng.uniform(0.0, np.pi / 2, 3_000_000), True),
                          ("towards pi/2", np.pi / 2 - rng.uniform(0.0, 0.2, 1_000_000) * rng.uniform(0.0, 1.0, 1_000_000), True),
                          ("[-2.42, 2.42]", rng.uniform(-2.42, 2.42, 3_000_000), False), ("wide", rng.uniform(-60.0, 60.0, 500_000), False)):
        ref = libm_cos(x)
        got, main = restated(x)
        bad = (bits(ref) != bits(got)) & main
        assert not bad.any(), "%s: %d results differ from libm" % (name, int(bad.sum()))
        if full:
            assert main.mean() > 0.9999, name


def test_arguments_outside_the_main_path_are_declined(oracle):
    _, restated = oracle.powcheck()
    x = np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 1e-310, 2.0, 2.0, 2.0, 2.0, 1e300, 1e-300])
    y = np.array([2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 0.0, 1e-30, 1e30, np.nan, 2.0, 2.0])
    _, main = restated(x, y)
    assert not main.any()                                          # the caller then takes the device library's pow


def test_generated_tables_are_the_installed_libms(tmp_path):
    """mw_glibc_pow_tables.h is data read from a libm: regenerate it from the libm of THIS machine and compare (a box with another
    glibc would need its own tables -- and this test says so instead of a silent last-bit difference)."""
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no " + libm)
    hdr = os.path.join(ROOT, "miniweatherml_amd", "csrc", "mw_glibc_pow_tables.h")
    before = open(hdr).read()
    env = dict(os.environ, MW_GP_OUT=str(tmp_path / "tables.h"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_glibc_pow_tables.py"), libm], env=env, stdout=subprocess.DEVNULL)
    strip = lambda t: "\n".join(ln for ln in t.splitlines() if not ln.startswith("// GENERATED"))      # noqa: E731
    assert strip(open(str(tmp_path / "tables.h")).read()) == strip(before)
    assert open(hdr).read() == before


@pytest.mark.gpu
def test_device_routine_is_bit_identical_to_libm(mw, oracle):
    import torch
    from miniweatherml_amd import capi
    libm_pow, _ = oracle.powcheck()
    for name, (x, y) in argument_sets(400_000, 12).items():
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        out = torch.empty_like(xd)
        main = torch.empty(xd.numel(), dtype=torch.uint8, device="cuda")
        capi.check(capi.lib().mw_strict_pow(xd.numel(), xd.data_ptr(), yd.data_ptr(), out.data_ptr(), main.data_ptr(), None))
        torch.cuda.synchronize()
        m = main.cpu().numpy().astype(bool)
        ref, got = libm_pow(x, y), out.cpu().numpy()
        bad = (bits(ref) != bits(got)) & m
        assert not bad.any(), "%s: %d device results differ from libm" % (name, int(bad.sum()))
        # off the main path the device library's pow is used: within an ulp or two of libm, same special values
        o = ~m & np.isfinite(ref) & (np.abs(ref) > 1e-290)            # (normal results: a subnormal has fewer bits to agree on)
        if o.any():
            assert np.max(np.abs(got[o] - ref[o]) / np.abs(ref[o])) < 1e-15, name
