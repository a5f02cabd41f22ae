"""Random scenes against the oracle (the cases of tools/soak_gpu.py, a fixed handful of seeds): BVHs of every mix of
primitive kinds under instance chains, and lit rooms around one or two trees with plain, metal, glass, checkered and
perlin materials and constant media -- the combinations that pick the different forms of the tail kernel, of the shade
launches and of the wide walk."""
import importlib.util
import os

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _soak():
    spec = importlib.util.spec_from_file_location("soak_gpu", os.path.join(ROOT, "tools", "soak_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", range(10))
def test_random_tree_ray_batches_match_the_oracle(seed):
    """hit, primitive, t, point and normal bit for bit (one seed per mix of kinds in soak_gpu.KINDS)."""
    _soak().batch_case(seed)


@pytest.mark.parametrize("seed", range(12))
def test_random_room_films(seed):
    """The film does not depend on the schedule, is the in-order sum of its samples bit for bit, and agrees with the
    oracle's (seeds 3, 7 and 11 hold constant media; 0, 3, 6, 9 plain materials only; 2, 5, 8, 11 a perlin texture)."""
    _soak().render_case(seed)


def test_random_room_film_above_a_million_paths():
    """One batch of more than a million paths runs the per-bounce launches (fewer go straight to the tail kernel)."""
    _soak().render_case(7, width=520, spp=4)
