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


@pytest.mark.parametrize('words_before', [1, 623, 624])
def test_native_gumbel_from_odd_positions_and_block_ends(words_before):
    """A value is made of two consecutive words of the generator: from an odd position inside a block the pairs straddle
    the block boundaries, and a block that ends behind a value's first word hands that word over to the next one."""
    np.random.seed(3)
    for _ in range(words_before):
        np.random.bytes(4)            # one word each
    state = np.random.get_state()
    want = np.random.gumbel(size=20001)
    after = np.random.random_sample(3)
    np.random.set_state(state)
    got = engine.legacy_gumbel((20001,))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.random.random_sample(3), after)


def test_native_gumbel_in_a_forked_child_and_from_several_threads():
    """The helper threads of the draws live as long as the process; a forked child has none of them and must get its
    own, and callers on several threads take turns."""
    import os, threading
    np.random.seed(1)
    state = np.random.get_state()
    want = np.random.gumbel(size=(16384, 9))
    np.random.set_state(state)
    np.testing.assert_array_equal(engine.legacy_gumbel((16384, 9)), want)
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            np.random.set_state(state)
            code = 0 if (engine.legacy_gumbel((16384, 9)) == want).all() else 4
        finally:
            os._exit(code)
    assert os.WEXITSTATUS(os.waitpid(pid, 0)[1]) == 0
    import ctypes as C
    results = []

    def call():
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos, out = C.c_int32(int(state[2])), np.empty(16384 * 9)
        rc = engine.lib().mz_legacy_gumbel(key.ctypes.data_as(C.c_void_p), C.byref(pos), C.c_int64(out.size),
                                           out.ctypes.data_as(C.c_void_p))
        results.append(rc == 0 and (out.reshape(16384, 9) == want).all())
    threads = [threading.Thread(target=call) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == [True] * 4
