"""The short division and square-root sequences of the render kernel against `/` and sqrtf,
on the GPU, over random operands of the ranges the kernel uses them in (rt_check_arith)."""
import pytest

from raytracing_amd import Renderer, load_tables

pytestmark = pytest.mark.gpu


def test_short_sequences_give_the_bits_of_division_and_sqrt():
    with Renderer(0, load_tables()) as r:
        total = 0
        for seed in range(4):
            bad_div, bad_sqrt = r.check_arith(1 << 28, seed)     # 2.7e8 quotients and roots each
            assert (bad_div, bad_sqrt) == (0, 0), (seed, bad_div, bad_sqrt)
            total += 1 << 28
        assert total >= 10 ** 9
