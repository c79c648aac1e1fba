import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (Tier-1 oracle; this container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/src")
    skip_ref = pytest.mark.skip(reason="/root/reference not mounted")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


NORTH_STAR = 1e-3  # BASELINE.json north_star: "latents matching reference within 1e-3 rel" (fp16)


def north_star(report, case, values, guard, cause):
    """Parity hygiene (round-4 verdict): a figure above the north star's 1e-3 is a MISS, whatever regression guard it sits
    under.  `values` = {name: measured rel-L2}; every value must stay under `guard` (the regression bar: a float or a dict per
    name) — that part is asserted — and if any value is >= 1e-3 the case is written to the parity report as a KNOWN MISS
    (measured, north-star bar, guard, cause) and the test ends as `xfail`, not as a green dot."""
    for k, v in values.items():
        g = guard[k] if isinstance(guard, dict) else guard
        assert v < g, f"{case}: {k} = {v:.2e} is above its regression guard {g:.1e}"
    above = {k: v for k, v in values.items() if v >= NORTH_STAR}
    if above:
        gs = guard if not isinstance(guard, dict) else max(guard[k] for k in above)
        line = (f"KNOWN MISS | {case} | " + ", ".join(f"{k} {v:.2e}" for k, v in above.items()) +
                f" | north-star bar 1e-3 | regression guard {gs:.1e} | {cause}")
        report(line)
        pytest.xfail(line)
