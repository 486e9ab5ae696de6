"""The committed oracle renders (tests/golden/renders.npz, written by
tests/golden/make_renders.py): the oracle must still produce them (CPU), and the GPU
library must produce them bit for bit (GPU)."""
import os

import numpy as np
import pytest

from raytracing_amd import make_params
from tests.golden import make_renders

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "renders.npz"))


@pytest.mark.parametrize("name", sorted(make_renders.CASES))
def test_oracle_reproduces_golden(oracle, tables, name):
    got = make_renders.render_case(oracle, tables, name)
    for key, val in got.items():
        want = GOLD[name + "/" + key]
        if key == "accum":
            assert np.array_equal(val.view(np.uint32), want.view(np.uint32))
        else:
            assert np.array_equal(val, want), key


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_renders.CASES))
def test_gpu_reproduces_golden(renderer, name):
    flat, cam, w, h, spp = make_renders.CASES[name]()
    renderer.set_scene(flat)
    renderer.set_camera(cam)
    renderer.begin_frame(w, h)
    renderer.render(make_params(w, h, spp=spp))
    assert np.array_equal(renderer.read_accum().view(np.uint32), GOLD[name + "/accum"].view(np.uint32))
    assert np.array_equal(renderer.read_primary_ids(), GOLD[name + "/ids"])
    assert np.array_equal(renderer.read_packed(), GOLD[name + "/packed"])
    assert renderer.ray_count() == int(GOLD[name + "/segments"])
