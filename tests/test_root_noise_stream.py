"""CPU: the batched root noise of the drop-in search comes out of the GLOBAL NumPy stream exactly as B consecutive
reference searches would draw it (mcts/mcts.py:27 `np.random.dirichlet`, mcts/mcts.py:211 `np.random.gumbel`,
mcts/action_selection.py:69 `np.random.choice` = one `random_sample`), and leaves the stream at the same position.
The native generator restates NumPy 2.2.6's legacy MT19937 / standard_gamma; NumPy itself is the reference here."""
import numpy as np
import pytest

from muzero_amd import engine


@pytest.mark.parametrize('alpha,A,B,with_u', [(0.25, 2, 1, True), (0.2, 9, 300, True), (0.25, 18, 257, False),
                                              (0.03, 5, 64, True), (0.97, 64, 33, True)])
def test_dirichlet_rows_and_uniforms_match_numpy(alpha, A, B, with_u):
    for seed in (0, 7, 123456):
        np.random.seed(seed)
        np.random.random_sample(seed % 5)            # some position inside the 624-word block
        want_d, want_u = np.empty((B, A)), np.empty(B)
        for b in range(B):
            want_d[b] = np.random.dirichlet([alpha] * A)
            if with_u:
                want_u[b] = np.random.random_sample()
        after = np.random.random_sample(3)
        np.random.seed(seed)
        np.random.random_sample(seed % 5)
        got_d, got_u = engine.legacy_root_noise(alpha, A, B, with_u)
        np.testing.assert_array_equal(got_d, want_d)
        if with_u:
            np.testing.assert_array_equal(got_u, want_u)
        else:
            assert got_u is None
        np.testing.assert_array_equal(np.random.random_sample(3), after)   # the stream went on from the same place


def test_crosses_many_state_refills_and_keeps_a_cached_gaussian():
    np.random.seed(5)
    np.random.standard_normal()                      # leaves a cached Gaussian in the legacy state
    state = np.random.get_state()
    want = np.stack([np.random.dirichlet([0.25] * 18) for _ in range(5000)])
    want_next = np.random.standard_normal(2)
    np.random.set_state(state)
    got, _ = engine.legacy_root_noise(0.25, 18, 5000, False)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.random.standard_normal(2), want_next)


def test_shapes_of_one_and_above_fall_back_to_numpy():
    np.random.seed(3)
    want = np.stack([np.random.dirichlet([1.5] * 4) for _ in range(10)])
    np.random.seed(3)
    got, _ = engine.legacy_root_noise(1.5, 4, 10, False)
    np.testing.assert_array_equal(got, want)


def test_gumbel_rows_are_the_consecutive_draws():
    np.random.seed(11)
    want = np.stack([np.random.gumbel(size=9) for _ in range(1000)])
    np.random.seed(11)
    np.testing.assert_array_equal(np.random.gumbel(size=(1000, 9)), want)


@pytest.mark.parametrize('n', [16384, 20000, 16384 * 9, 300001])
def test_native_gumbel_draws_are_numpys(n):
    """np.random.gumbel(size=(B, A)) of MCTS.search_batch in one native call: the values, bit for bit, and the stream
    position afterwards (the next draws agree), from a state in the middle of a block of the generator and with a
    cached Gaussian pending."""
    np.random.seed(77)
    np.random.standard_normal(3)      # leaves a cached Gaussian
    np.random.random_sample(101)      # a position inside the 624-word block
    state = np.random.get_state()
    want = np.random.gumbel(size=n)
    after = np.random.random_sample(5), np.random.standard_normal(1)
    np.random.set_state(state)
    got = engine.legacy_gumbel((n,))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.random.random_sample(5), after[0])
    np.testing.assert_array_equal(np.random.standard_normal(1), after[1])


def test_native_gumbel_small_draws_and_shapes():
    np.random.seed(5)
    want = np.random.gumbel(size=(7, 9))
    np.random.seed(5)
    np.testing.assert_array_equal(engine.legacy_gumbel((7, 9)), want)       # small: NumPy itself
    np.random.seed(6)
    want = np.random.gumbel(size=(4096, 9))
    np.random.seed(6)
    got = engine.legacy_gumbel((4096, 9))
    assert got.shape == (4096, 9)
    np.testing.assert_array_equal(got, want)
