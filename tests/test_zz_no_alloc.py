"""shp_ntt_fr_noalloc (the entry point that allocates nothing) against shp_ntt_fr, through the C ABI.

Two calls in a row share the library's static ping-pong array, the second with a smaller input: its result must
not depend on what the first call left behind.  Bit-exact (integer arithmetic)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rand_fr(oracle, n, seed):
    """n valid Montgomery limb vectors (uniform values below 2^253 < r)"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)
    return raw


def _both(shp, ctx, a, log_n, batch, inverse):
    want = shp.to_host(ctx.ntt(shp.to_device(a), log_n, batch, inverse=inverse)).copy()
    got = shp.to_host(ctx.ntt_noalloc(shp.to_device(a), log_n, batch, inverse=inverse)).copy()
    return got, want


def test_noalloc_ntt_equals_ntt_two_calls_second_smaller(shp, oracle, gpu_ctx):
    # first call: 2^20 elements, two passes of 1024-element tiles (fills the whole static array)
    a = rand_fr(oracle, 1 << 20, 9001)
    got, want = _both(shp, gpu_ctx, a, 20, 1, False)
    assert np.array_equal(got, want), "2^20 forward"
    assert np.array_equal(want.reshape(-1, 4), oracle.ntt(a, 20)), "shp_ntt_fr itself against the oracle"
    # second call, smaller, two passes, batched, on top of the first call's leftovers
    b = rand_fr(oracle, 3 << 14, 9002)
    got, want = _both(shp, gpu_ctx, b, 14, 3, False)
    assert np.array_equal(got, want), "3 x 2^14 forward after the large call"
    got, want = _both(shp, gpu_ctx, b, 14, 3, True)
    assert np.array_equal(got, want), "3 x 2^14 inverse"
    # every pass structure the entry point has (log_n <= 20): a single pass (no ping-pong) and two passes, with both tile
    # sizes; inverse undoes forward
    for log_n, batch in ((1, 5), (5, 2), (11, 4), (12, 1), (13, 7), (18, 2), (19, 2), (20, 1)):
        c = rand_fr(oracle, batch << log_n, 9100 + log_n)
        got, want = _both(shp, gpu_ctx, c, log_n, batch, False)
        assert np.array_equal(got, want), f"log_n={log_n} batch={batch}"
        t = shp.to_device(c)
        gpu_ctx.ntt_noalloc(t, log_n, batch)
        gpu_ctx.ntt_noalloc(t, log_n, batch, inverse=True)
        assert np.array_equal(shp.to_host(t).reshape(-1, 4), c), f"round trip log_n={log_n}"


def test_noalloc_ntt_rejects_sizes_beyond_the_static_array(shp, oracle, gpu_ctx):
    t = shp.to_device(rand_fr(oracle, 1 << 12, 1))
    for log_n, batch in ((21, 1), (20, 2), (12, 257), (0, 1)):
        rc = shp.lib().shp_ntt_fr_noalloc(gpu_ctx.h, C.c_void_p(t.data_ptr()), C.c_uint32(log_n), C.c_uint32(batch),
                                          C.c_int32(0), None)
        assert rc == 1, f"log_n={log_n} batch={batch}: expected SHP_ERR_INVALID_ARG, got {rc}"
        assert b"ntt_noalloc" in shp.lib().shp_last_error()
    assert np.array_equal(shp.to_host(t).reshape(-1, 4), rand_fr(oracle, 1 << 12, 1)), "a rejected call must not touch the data"
