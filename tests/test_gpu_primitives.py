"""GPU: the HIP kernels behind the C ABI against the CPU oracle, bit for bit (integer work)."""
import json
import random

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def rand_fr(oracle, n, seed):
    """n pseudo-random Montgomery-form elements (uniform 254-bit values reduced by the oracle's sub)"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)          # < 2^253 < r : valid limb vectors (any residue is a valid Montgomery value)
    return raw


@pytest.mark.parametrize("log_n", [1, 2, 4, 7, 11, 12, 13, 16, 18])
def test_ntt_matches_oracle(shp, oracle, gpu_ctx, log_n):
    n = 1 << log_n
    batch = 3 if log_n <= 13 else 1
    a = rand_fr(oracle, n * batch, 100 + log_n)
    t = shp.to_device(a)
    gpu_ctx.ntt(t, log_n, batch)
    got = shp.to_host(t).reshape(batch, n, 4)
    for b in range(batch):
        assert np.array_equal(got[b], oracle.ntt(a.reshape(batch, n, 4)[b], log_n)), f"forward log_n={log_n} batch {b}"
    gpu_ctx.ntt(t, log_n, batch, inverse=True)
    assert np.array_equal(shp.to_host(t).reshape(-1, 4), a), "inverse(forward(x)) != x"
    t2 = shp.to_device(a)
    gpu_ctx.ntt(t2, log_n, batch, inverse=True)
    assert np.array_equal(shp.to_host(t2).reshape(batch, n, 4)[0], oracle.ntt(a.reshape(batch, n, 4)[0], log_n, inverse=True))


@pytest.mark.parametrize("log_n", [19, 20, 22])
def test_ntt_three_pass_matches_oracle(shp, oracle, gpu_ctx, log_n):
    n = 1 << log_n
    a = rand_fr(oracle, n, 7 + log_n)
    t = shp.to_device(a)
    gpu_ctx.ntt(t, log_n)
    assert np.array_equal(shp.to_host(t).reshape(n, 4), oracle.ntt(a, log_n))
    gpu_ctx.ntt(t, log_n, inverse=True)
    assert np.array_equal(shp.to_host(t).reshape(n, 4), a)


def test_ntt_2_24_roundtrip_and_linearity(shp, oracle, gpu_ctx):
    """full BASELINE size: size-independent properties (round trip, linearity, delta -> constant)"""
    import torch
    log_n, n = 24, 1 << 24
    a = rand_fr(oracle, n, 424242)
    ta = shp.to_device(a)
    gpu_ctx.ntt(ta, log_n)
    fa = shp.to_host(ta).reshape(n, 4).copy()
    gpu_ctx.ntt(ta, log_n, inverse=True)
    assert np.array_equal(shp.to_host(ta).reshape(n, 4), a)
    # NTT(e_1) = powers of omega: check a few entries against the oracle's root of unity
    e = np.zeros((n, 4), dtype=np.uint64)
    e[1] = oracle.fr_array([1])[0]
    te = shp.to_device(e)
    gpu_ctx.ntt(te, log_n)
    fe = shp.to_host(te).reshape(n, 4)
    w = oracle.fr_ints(oracle.root_of_unity(24))[0]
    for i in (0, 1, 2, 12345, n - 1):
        assert oracle.fr_ints(fe[i:i + 1])[0] == pow(w, i, R)
    # spot-check 4 outputs of the random transform with a direct sum over a sparse support instead:
    # linearity: NTT(a + e_1) = NTT(a) + NTT(e_1) on sampled positions
    a2 = a.copy()
    s = (oracle.fr_ints(a[1:2])[0] + 1) % R
    a2[1] = oracle.fr_array([s])[0]
    t2 = shp.to_device(a2)
    gpu_ctx.ntt(t2, log_n)
    f2 = shp.to_host(t2).reshape(n, 4)
    for i in (0, 5, 99999, n - 7):
        assert oracle.fr_ints(f2[i:i + 1])[0] == (oracle.fr_ints(fa[i:i + 1])[0] + pow(w, i, R)) % R
    del torch


def test_coset_extension_matches_oracle(shp, oracle, gpu_ctx):
    k, ek, batch = 13, 16, 2
    a = rand_fr(oracle, batch << k, 5)
    ext = gpu_ctx.coeff_to_extended(shp.to_device(a), k, ek, batch)
    got = shp.to_host(ext).reshape(batch, 1 << ek, 4)
    for b in range(batch):
        assert np.array_equal(got[b], oracle.coeff_to_extended(a.reshape(batch, 1 << k, 4)[b], k, ek))
    gpu_ctx.extended_to_coeff(ext, ek, batch)
    back = shp.to_host(ext).reshape(batch, 1 << ek, 4)
    for b in range(batch):
        assert np.array_equal(back[b], oracle.extended_to_coeff(got[b], ek))
        assert np.array_equal(back[b][: 1 << k], a.reshape(batch, 1 << k, 4)[b])


def test_poseidon2_matches_reference_vectors_and_oracle(shp, oracle, gpu_ctx):
    g = json.load(open(golden_path("poseidon2_t8.json")))
    ins = [[int(x, 16) for x in v["inputs"]] for v in g["vectors"]]
    flat = oracle.fr_array([x for row in ins for x in row])
    out = shp.to_host(gpu_ctx.poseidon2_hash(shp.to_device(flat), len(ins), 7)).reshape(-1, 4)
    assert oracle.fr_ints(out) == [int(v["hash"], 16) for v in g["vectors"]]
    # other arities and a big batch against the oracle
    for arity in (1, 2, 3, 6):
        n = 1000
        a = rand_fr(oracle, n * arity, 31 + arity)
        got = shp.to_host(gpu_ctx.poseidon2_hash(shp.to_device(a), n, arity)).reshape(n, 4)
        for i in (0, 1, 17, n - 1):
            assert np.array_equal(got[i], oracle.poseidon2_hash(a.reshape(n, arity, 4)[i]))
    st = rand_fr(oracle, 8 * 130, 77)
    perm = shp.to_host(gpu_ctx.poseidon2_permute(shp.to_device(st), 130)).reshape(130, 8, 4)
    for i in (0, 64, 129):
        assert np.array_equal(perm[i], oracle.poseidon2_permute(st.reshape(130, 8, 4)[i]))


def test_msm_real_srs_commit_equals_commit_lagrange(shp, oracle, gpu_ctx):
    """crates/powers-of-tau/lib.rs:247-264 through the HIP path, plus equality with the oracle"""
    raw_bytes = open(golden_path("ppot_0080_11_raw"), "rb").read()
    params = shp.Params.read_raw(gpu_ctx, raw_bytes)
    osrs = oracle.srs_from_halo2_raw(raw_bytes)
    assert params.k == 11
    assert np.array_equal(params.get_g(), osrs.g)
    a = oracle.fr_array(list(range(2048)))
    ta = shp.to_device(a)
    c_l = params.commit_lagrange(ta)[0]
    tb = shp.to_device(a)
    gpu_ctx.ntt(tb, 11, inverse=True)
    c_m = params.commit(tb)[0]
    assert np.array_equal(c_l, c_m)
    assert np.array_equal(c_l, oracle.msm(osrs.g_lagrange, a))
    params.close()


def test_srs_from_ptau_on_gpu_equals_raw(shp, oracle, gpu_ctx):
    """crates/powers-of-tau/lib.rs:266-281 through the HIP path; g_lagrange from the GPU group FFT"""
    p1 = shp.Params.read_ptau(gpu_ctx, open(golden_path("ppot_0080_11.head.ptau"), "rb").read())
    osrs = oracle.srs_from_halo2_raw(open(golden_path("ppot_0080_11_raw"), "rb").read())
    assert np.array_equal(p1.get_g(shp.BASIS_MONOMIAL), osrs.g)
    assert np.array_equal(p1.get_g(shp.BASIS_LAGRANGE), osrs.g_lagrange)
    assert p1.g2() == (osrs.g2, osrs.s_g2)
    p1.close()
    with pytest.raises(shp.ShpError):
        shp.Params.read_ptau(gpu_ctx, b"\x00" * 100)


def test_srs_downsize_real_srs(shp, oracle, gpu_ctx):
    """Params::downsize (reference shielder-cli pk.rs:110; generate_keys_with_min_k): first 2^k powers of the real
    k=11 SRS, g_lagrange re-derived on the GPU == the oracle's group iFFT, G2 part kept; commit == commit_lagrange
    (crates/powers-of-tau/lib.rs:247-264) under the downsized Params; k above the SRS's k is an argument error"""
    raw_bytes = open(golden_path("ppot_0080_11_raw"), "rb").read()
    full = shp.Params.read_raw(gpu_ctx, raw_bytes)
    osrs = oracle.srs_from_halo2_raw(raw_bytes)
    for k in (11, 9, 1):
        small = full.downsize(k)
        assert small.k == k
        assert np.array_equal(small.get_g(shp.BASIS_MONOMIAL), osrs.g[: 1 << k])
        want_gl = osrs.g_lagrange if k == 11 else oracle.g1_lagrange_from_monomial(osrs.g[: 1 << k], k)
        assert np.array_equal(small.get_g(shp.BASIS_LAGRANGE), want_gl)
        assert small.g2() == (osrs.g2, osrs.s_g2)
        a = rand_fr(oracle, 1 << k, 31 + k)
        c_l = small.commit_lagrange(shp.to_device(a))[0]
        tb = shp.to_device(a)
        gpu_ctx.ntt(tb, k, inverse=True)
        assert np.array_equal(c_l, small.commit(tb)[0])
        assert np.array_equal(c_l, oracle.msm(want_gl, a))
        small.close()
    with pytest.raises(shp.ShpError) as e:
        full.downsize(12)
    assert e.value.code == 1                             # SHP_ERR_INVALID_ARG
    full.close()


def test_srs_synthetic_matches_oracle(shp, oracle, gpu_ctx):
    tau = 0x1234567890ABCDEF1234567890ABCDEF55
    p = shp.Params.synthetic(gpu_ctx, 9, tau)
    o = oracle.srs_synthetic(9, tau)
    assert np.array_equal(p.get_g(shp.BASIS_MONOMIAL), o.g)
    assert np.array_equal(p.get_g(shp.BASIS_LAGRANGE), o.g_lagrange)
    p.close()


@pytest.mark.parametrize("n,window", [(1, 0), (5, 0), (100, 7), (2048, 0), (2048, 12), (8192, 0), (8192, 12)])
def test_msm_matches_oracle(shp, oracle, gpu_ctx, n, window):
    osrs = oracle.srs_synthetic(13, 0xDEADBEEFCAFE1234567)
    bases = osrs.g[:n]
    m = shp.Msm(gpu_ctx, bases, window)
    batch = 3
    s = rand_fr(oracle, n * batch, 900 + n)
    got = m.run(shp.to_device(s), n, batch)
    for b in range(batch):
        assert np.array_equal(got[b], oracle.msm(bases, s.reshape(batch, n, 4)[b])), f"n={n} batch {b}"
    m.close()


def test_msm_edge_cases(shp, oracle, gpu_ctx):
    """witness-like columns: zeros, ones, tiny values, r-1, one heavy bucket; identity bases"""
    n = 8192
    osrs = oracle.srs_synthetic(13, 0xDEADBEEFCAFE1234567)
    bases = osrs.g.copy()
    bases[5] = 0                       # an identity base
    m = shp.Msm(gpu_ctx, bases)
    rnd = random.Random(1)
    cols = [
        [0] * n,
        [1] * n,
        [R - 1] * n,
        [rnd.choice([0, 1]) for _ in range(n)],
        [rnd.randrange(256) for _ in range(n)],
        [0] * (n - 1) + [12345],
        [(1 << 253) + i for i in range(n)],
    ]
    s = oracle.fr_array([x for c in cols for x in c])
    got = m.run(shp.to_device(s), n, len(cols))
    for i, c in enumerate(cols):
        assert np.array_equal(got[i], oracle.msm(bases, s.reshape(len(cols), n, 4)[i])), f"column {i}"
    assert not got[0].any()
    # fewer scalars than bases
    got2 = m.run(shp.to_device(s[:1000]), 1000, 1)
    assert np.array_equal(got2[0], oracle.msm(bases[:1000], s[:1000]))
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("window", [10, 13, 16])
def test_msm_two_level_scatter_matches_oracle(shp, oracle, gpu_ctx, window, monkeypatch):
    """the bucket method's two-level scatter (bins of 256 buckets, then buckets), forced on inputs far smaller than the
    sizes that select it by themselves: random scalars, columns whose digits fall into one or two bins only (a slice of
    the second level then spans many bins or a single bucket), the empty column, fewer scalars than bases"""
    monkeypatch.setenv("SHP_MSM_TWO_LEVEL", "1")
    n = 8192
    osrs = oracle.srs_synthetic(13, 0xDEADBEEFCAFE1234567)
    bases = osrs.g.copy()
    bases[7] = 0
    m = shp.Msm(gpu_ctx, bases, window)
    rnd = random.Random(window)
    cols = [
        [rnd.randrange(R) for _ in range(n)],
        [0] * n,
        [1] * n,
        [rnd.randrange(4) for _ in range(n)],
        [R - 1 - rnd.randrange(3) for _ in range(n)],
        [0] * (n - 1) + [98765],
        [(rnd.randrange(1 << 20) << 200) + rnd.randrange(2) for _ in range(n)],
    ]
    s = oracle.fr_array([x for c in cols for x in c])
    got = m.run(shp.to_device(s), n, len(cols))
    for i, c in enumerate(cols):
        assert np.array_equal(got[i], oracle.msm(bases, s.reshape(len(cols), n, 4)[i])), f"window {window} column {i}"
    got2 = m.run(shp.to_device(s[:777]), 777, 1)
    assert np.array_equal(got2[0], oracle.msm(bases[:777], s[:777]))
    monkeypatch.setenv("SHP_MSM_TWO_LEVEL", "0")
    got3 = m.run(shp.to_device(s[:n]), n, 1)
    assert np.array_equal(got3[0], got[0])
    m.close()


# BASELINE configs 4 and 5 at every size they name (2^20, 2^22, 2^24) against the CPU port: the 2^24 MSM costs the port
# about 8 s and 1 GB, the 2^24 NTT 2 s.  MICRO_LOG_NS narrows the list (e.g. MICRO_LOG_NS=20 for a quick run).
_MICRO = [int(x) for x in __import__("os").environ.get("MICRO_LOG_NS", "20,22,24").split(",")]


@pytest.mark.parametrize("log_n", _MICRO)
def test_msm_2_20_matches_oracle(shp, oracle, gpu_ctx, log_n):
    """BASELINE config 4 against the multi-threaded oracle (a few seconds of CPU at 2^20)"""
    import torch
    n = 1 << log_n
    osrs = oracle.srs_synthetic(13, 987654321)
    reps = n // osrs.n
    # distinct-enough bases: tile the 8192 synthetic points (the MSM does not care about repeats)
    bases = np.tile(osrs.g, (reps, 1))
    s = rand_fr(oracle, n, 2020)
    import time
    m = shp.Msm(gpu_ctx, bases)
    ts = shp.to_device(s)
    got = m.run(ts, n, 1)[0]
    adds, nbytes = m.stats()
    assert nbytes == 96 * n and adds > 0
    t0 = time.perf_counter()
    for _ in range(3):
        m.run(ts, n, 1)
    gpu_s = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter(); want = oracle.msm(bases, s); cpu_s = time.perf_counter() - t0
    assert np.array_equal(got, want)
    _record(f"config4_msm_2_{log_n}", {"n": n, "gpu_ms": 1e3 * gpu_s, "cpu_port_ms": 1e3 * cpu_s, "cpu_threads": oracle.num_threads(),
                                  "speedup": cpu_s / gpu_s, "alg_GBps": 96 * n / gpu_s / 1e9})
    m.close()
    del torch


def _record(name, obj):
    """GPU-vs-CPU-port timings of the standalone configs, kept with the run's other outputs"""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "micro_vs_cpu.jsonl"), "a") as f:
            f.write(json.dumps({"config": name, **obj}) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("log_n", _MICRO)
def test_ntt_2_20_vs_cpu_port(shp, oracle, gpu_ctx, log_n):
    """BASELINE config 5: parity with the oracle and the GPU / CPU-port timing"""
    import time
    import torch
    n = 1 << log_n
    a = rand_fr(oracle, n, 55)
    t = shp.to_device(a)
    gpu_ctx.ntt(t, log_n)
    got = shp.to_host(t).reshape(n, 4).copy()
    # GPU timing first: the OpenMP oracle leaves its threads spinning for a while afterwards; a few untimed transforms
    # bring an idle GPU up to its clocks (the first ~20 ms of work after idle run several times slower)
    for _ in range(20):
        gpu_ctx.ntt(t, log_n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gpu_ctx.ntt(t, log_n)
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter(); want = oracle.ntt(a, log_n); cpu_s = time.perf_counter() - t0
    assert np.array_equal(got, want)
    _record(f"config5_ntt_2_{log_n}", {"n": n, "gpu_ms": 1e3 * gpu_s, "cpu_port_ms": 1e3 * cpu_s, "cpu_threads": oracle.num_threads(),
                                  "speedup": cpu_s / gpu_s, "alg_GBps": 64 * n / gpu_s / 1e9})


def test_merkle_tree_matches_contract_semantics(shp, oracle, gpu_ctx):
    """contracts/MerkleTree.sol through the GPU: levels = Poseidon2 of 7 children (unset = 0), getMerklePath layout
    (siblings-with-ancestor per level, then root), against the oracle's hash"""
    H = lambda xs: oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(xs)))[0]
    for n_leaves, height in ((1, 2), (100, 4), (343, 3)):
        rnd = random.Random(n_leaves)
        leaves = [rnd.randrange(R) for _ in range(n_leaves)]
        t = shp.MerkleTree(gpu_ctx, shp.to_device(oracle.fr_array(leaves)), n_leaves, height)
        levels = [leaves]
        for _ in range(height):
            cur = levels[-1]
            levels.append([H((cur[7 * j: 7 * j + 7] + [0] * 7)[:7]) for j in range((len(cur) + 6) // 7)])
        assert len(levels[-1]) == 1 and t.root() == levels[-1][0]
        qs = sorted({0, n_leaves - 1, n_leaves // 2, min(7, n_leaves - 1)})
        for (path, root), q in zip(t.paths(qs), qs):
            assert root == levels[-1][0]
            node = q
            for l in range(height):
                want = (levels[l][(node // 7) * 7: (node // 7) * 7 + 7] + [0] * 7)[:7]
                assert path[l] == want and path[l][node % 7] == levels[l][node]
                node //= 7
            # the relation the withdraw circuit proves: each level's hash sits in the next level
            for l in range(height - 1):
                assert H(path[l]) in path[l + 1]
            assert H(path[height - 1]) == root
        with pytest.raises(shp.ShpError):
            t.paths([n_leaves])
        t.close()


def test_msm_direct_tree_special_cases(shp, oracle, gpu_ctx):
    """the reduction trees of the direct-table MSM (six lanes an addition, complete formulas: no exceptional cases to
    branch on): equal partial sums (doubling), opposite partial sums (cancel to the identity, at the first passes and at
    the last one), identities scattered among the lanes"""
    n = 4096
    osrs = oracle.srs_synthetic(12, 0xDEADBEEFCAFE1234567)
    bases = osrs.g.copy()
    bases[1::2] = bases[0::2]                      # every base twice: lanes 2i and 2i+1 hold the same point
    m = shp.Msm(gpu_ctx, bases)
    rnd = random.Random(5)
    same = [0] * n
    opp = [0] * n
    mixed = [0] * n
    for i in range(0, n, 2):
        s = rnd.randrange(1, R)
        same[i] = same[i + 1] = s                  # P + P at the first tree level
        opp[i], opp[i + 1] = s, R - s              # P + (-P)
        mixed[i] = s if i % 8 == 0 else 0
        mixed[i + 1] = s if i % 16 == 0 else 0
    cols = [same, opp, mixed, [5] * n, [R - 5 if i % 2 else 5 for i in range(n)]]
    s_arr = oracle.fr_array([x for c in cols for x in c])
    got = m.run(shp.to_device(s_arr), n, len(cols))
    for i, c in enumerate(cols):
        assert np.array_equal(got[i], oracle.msm(bases, s_arr.reshape(len(cols), n, 4)[i])), f"column {i}"
    assert not got[1].any() and not got[4].any()   # everything cancels
    m.close()


def test_msm_direct_more_partial_sums_than_threads(shp, oracle, gpu_ctx, monkeypatch):
    """2^17 bases through a direct table (35 GB at the budget set here): 512 row blocks, so the workgroup that merges an
    MSM's partial sums holds two of them per thread before its tree starts (one lane's complete addition); columns of
    random scalars, of one repeated scalar, and one whose second half cancels the first"""
    monkeypatch.setenv("SHP_MSM_TABLE_GB", "40")
    n = 1 << 17
    osrs = oracle.srs_synthetic(13, 24680)
    bases = np.tile(osrs.g, (n // osrs.n, 1))
    m = shp.Msm(gpu_ctx, bases)
    rnd = random.Random(17)
    half = [rnd.randrange(1, R) for _ in range(n // 2)]
    cols = [None, [7] * n, half + [R - x for x in half]]      # bases repeat with period 8192: column 2 sums to the identity
    s = np.concatenate([rand_fr(oracle, n, 4711), oracle.fr_array(cols[1]), oracle.fr_array(cols[2])])
    got = m.run(shp.to_device(s), n, 3)
    for i in range(3):
        assert np.array_equal(got[i], oracle.msm(bases, s.reshape(3, n, 4)[i])), f"column {i}"
    assert not got[2].any()
    m.close()
