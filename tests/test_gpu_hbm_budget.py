"""GPU: MSM tables are sized against the HBM that is really free.  Several SRSs of different k live in one context; under
a cap on the process's device memory the multiples tables step their window down and finally give way to the bucket
method, and every variant commits to the same points as the oracle (bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GIB = 1 << 30
TAU = 0x1234567890ABCDEF1234567


def used_bytes():
    import torch
    free, total = torch.cuda.mem_get_info(0)
    return total - free


def check_commits(shp, oracle, params, seed):
    """commit (monomial) and commit_lagrange of one random polynomial == the oracle's MSM over the same points"""
    n = params.n
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    t = shp.to_device(a)
    g = params.get_g(shp.BASIS_MONOMIAL)
    gl = params.get_g(shp.BASIS_LAGRANGE)
    assert np.array_equal(params.commit(t)[0], oracle.msm(g, a))
    assert np.array_equal(params.commit_lagrange(t)[0], oracle.msm(gl, a))


def test_three_srs_of_different_k_and_window_step_down(shp, oracle, gpu_ctx):
    held = []
    try:
        gpu_ctx.set_hbm_limit(0)
        # 1. no cap: the widest window the per-table budget allows
        p11 = shp.Params.synthetic(gpu_ctx, 11, TAU)
        held.append(p11)
        # (the Lagrange basis carries every commitment of a proof: it is built first and may take c = 15; the coefficient
        # basis only serves commit() callers and stays within 8 GiB)
        d, c, nbytes = p11.msm_info(shp.BASIS_LAGRANGE)
        assert d and c == 15 and nbytes > 33 * GIB
        d, c, nbytes = p11.msm_info(shp.BASIS_MONOMIAL)
        assert d and c == 12 and 5 * GIB < nbytes < 8 * GIB
        check_commits(shp, oracle, p11, 1)
        # 2. a cap that leaves 6 GiB of headroom plus room for two c = 12 tables of a 2^10 SRS: both bases stay direct with
        #    narrower windows (the second basis sees what the first one took)
        c12 = 1024 * 22 * 2048 * 64
        gpu_ctx.set_hbm_limit(used_bytes() + 6 * GIB + 2 * c12 + (64 << 20))
        p10 = shp.Params.synthetic(gpu_ctx, 10, TAU)
        held.append(p10)
        d_g, c_g, _ = p10.msm_info(shp.BASIS_MONOMIAL)
        d_l, c_l, _ = p10.msm_info(shp.BASIS_LAGRANGE)
        assert d_g and d_l and 8 <= min(c_g, c_l) and max(c_g, c_l) <= 12, (c_g, c_l)
        check_commits(shp, oracle, p10, 2)
        # 3. nothing to spare: the bucket method, no table beyond the base rows
        gpu_ctx.set_hbm_limit(used_bytes() + 1 * GIB)
        p9 = shp.Params.synthetic(gpu_ctx, 9, TAU)
        held.append(p9)
        d9, _, bytes9 = p9.msm_info(shp.BASIS_LAGRANGE)
        assert not d9 and bytes9 < GIB
        check_commits(shp, oracle, p9, 3)
        # the three SRSs stay usable side by side
        check_commits(shp, oracle, p11, 4)
        check_commits(shp, oracle, p10, 5)
    finally:
        gpu_ctx.set_hbm_limit(0)
        for p in held:
            p.close()
