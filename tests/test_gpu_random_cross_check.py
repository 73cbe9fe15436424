"""A seeded, bounded slice of the randomised kernel cross-checks on every `pytest -m gpu` run: 90 cases of the
resident-row-panel kernels, 30 + 30 of the panels-inside-a-tile kernels in fp32 / fp64, 20 of the 256-row workgroups on block pairs, 40 of rbf with a large exponent scale (grid planes, round 5) -- shapes, chunk lengths, bands, shards, plane kinds,
group forms (incl. the 256-row workgroups), degrees, both real types, both variants -- each against the float64 product and the generic kernel
(tests/cross_check.py).  The long runs stay developer tools (tests/tools/narrow_stress.py, wide_stress.py: same generator, other seeds).
Reference counterpart: the kernel tests every backend runs in every build, /root/reference/tests/backends/generic_csvm_tests.hpp:372-493."""

import pytest

import cross_check

pytestmark = pytest.mark.gpu

SEED = 20261003
CASES = ([cross_check.narrow_case(SEED, i) for i in range(90)] + [cross_check.wide_case(SEED, i, False) for i in range(30)]
         + [cross_check.wide_case(SEED, i, True) for i in range(30)] + [cross_check.pair_case(SEED, i) for i in range(20)] + [cross_check.grid_case(SEED, i) for i in range(40)]
         + [cross_check.grid_pair_case(SEED, i) for i in range(16)] + [cross_check.rect_case(SEED, i) for i in range(24)])  # (round 6: the 256-row grid-plane kernel, the rectangular predict kernel)


@pytest.mark.parametrize("case", CASES, ids=[f"{c['family']}-{i}" for i, c in enumerate(CASES)])
def test_random_case_agrees_with_the_float64_product(case):
    res = cross_check.run_rect_case(case) if case["family"] == "rect" else cross_check.run_case(case)
    assert res["ok"], f"{cross_check.describe(case)} -> (gram mode, symmetric) ({res['gram_mode']}, {res['symmetric']}): {res['err']:.2f} eps from float64 (generic kernel: {res['err_generic']:.2f})"
