"""A few fixed seeds of the differential fuzzing tool (tools/fuzz_parity.py) as part of the GPU suite: random scenes and
random settings, library against oracle.  The tool itself runs for minutes on thousands of seeds; these keep it working and
keep its checks in the regression suite."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_soup_seeds():
    """Unstructured triangle soups: every acceleration structure, node layout, leaf size and kernel form; seed 591 is the
    scene on which two instance bounds shared a plane (hits went missing before the instance pass kept tied entries in order)."""
    fz = _tool()
    failures = [f for seed in (0, 1, 2, 3, 5, 37, 364, 591) for f in fz.one_seed(seed)]
    assert not failures, failures


def test_zoo_seeds():
    """Instanced primitives with every material feature, mirrors, objects scaled to nothing, texture coordinates outside
    the unit square; frames of every program, sample and tile splits, the entry point that allocates nothing."""
    fz = _tool()
    failures = [f for seed in range(60) for f in fz.one_zoo_seed(seed)]
    assert not failures, failures
