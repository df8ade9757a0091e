"""A seeded subset of the randomised GPU-vs-oracle frames (tests/fuzz_cases.py; tools/fuzz_parity.py runs thousands:
round 1 ran 9,560 mesh scenes, 26,300 analytic scenes and 4,000 ray batches with no mismatch)."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_scenes_match_the_oracle(rt, orc):
    from fuzz_cases import random_case
    failures = []
    for seed in range(48):
        ok, desc = random_case(rt, seed)
        if not ok:
            failures.append(desc)
    assert not failures, "\n".join(failures)


def test_random_analytic_scenes_match_the_oracle(rt, orc):
    from fuzz_cases import random_analytic_case
    failures = []
    for seed in range(64):
        ok, desc = random_analytic_case(rt, seed)
        if not ok:
            failures.append(desc)
    assert not failures, "\n".join(failures)


def test_random_ray_batches_match_the_oracle(rt, orc):
    from fuzz_cases import random_rays_case
    failures = []
    for seed in range(24):
        ok, desc = random_rays_case(rt, seed)
        if not ok:
            failures.append(desc)
    assert not failures, "\n".join(failures)


def test_random_grazing_shadow_scenes_match_the_oracle(rt, orc):
    """Exact shadows with the light in the plane of large floor faces (hit points off the ray): fuzz_cases.random_grazing_case."""
    from fuzz_cases import random_grazing_case
    failures = []
    for seed in range(32):
        ok, desc = random_grazing_case(rt, seed)
        if not ok:
            failures.append(desc)
    assert not failures, "\n".join(failures)


def test_random_scenes_match_the_oracle_in_burst_form(rt, orc, switch):
    """The walker's form for leaves of tens of faces (whole chunks tested wave-wide; trace_kernels.hip "burst") forced on
    for scenes of every leaf size, hierarchy-less meshes (one leaf of many chunks) included: the same frames."""
    from fuzz_cases import random_case, random_grazing_case, random_rays_case
    switch("BIG_LEAVES", "1")
    failures = []
    for family, n in ((random_case, 32), (random_rays_case, 12), (random_grazing_case, 16)):
        for seed in range(100, 100 + n):
            ok, desc = family(rt, seed)
            if not ok:
                failures.append(desc)
    assert not failures, "\n".join(failures)
