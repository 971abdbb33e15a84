"""CPU: the oracle against every golden vector the reference holds for the path."""
import json
import random

import numpy as np

from conftest import golden_path

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def test_keccak_kats(oracle):
    for v in json.load(open(golden_path("keccak_kat.json")))["vectors"]:
        assert oracle.keccak256(bytes.fromhex(v["input_hex"])).hex() == v["keccak256"]
    # empty string, Keccak (not SHA-3) padding
    assert oracle.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    # multi-block
    assert oracle.keccak256(b"a" * 200).hex() != oracle.keccak256(b"a" * 199).hex()


def test_field_arithmetic_vs_bigints(oracle):
    rnd = random.Random(7)
    xs = [rnd.randrange(R) for _ in range(64)] + [0, 1, R - 1]
    ys = [rnd.randrange(R) for _ in range(64)] + [R - 1, R - 1, R - 1]
    a, b = oracle.fr_array(xs), oracle.fr_array(ys)
    assert oracle.fr_ints(a) == xs
    inv = oracle.fr_ints(oracle.batch_invert(a))
    for x, i in zip(xs, inv):
        assert (x * i) % R == (1 if x else 0)
    # eval_poly = Horner
    x0 = rnd.randrange(R)
    got = oracle.fr_ints(oracle.eval_poly(a, oracle.fr_array([x0])[0]))[0]
    assert got == sum(c * pow(x0, i, R) for i, c in enumerate(xs)) % R
    del b


def test_domain_constants(oracle):
    c = json.load(open(golden_path("constants.json")))
    assert oracle.fr_ints(oracle.fr_delta())[0] == int(c["delta"], 16)
    assert oracle.fr_ints(oracle.root_of_unity(11))[0] == int(c["omega_11"], 16)
    assert oracle.fr_ints(oracle.fr_zeta())[0] == int(c["zeta"], 16)
    w28 = oracle.fr_ints(oracle.root_of_unity(28))[0]
    assert pow(w28, 1 << 28, R) == 1 and pow(w28, 1 << 27, R) != 1
    assert pow(int(c["zeta"], 16), 3, R) == 1


def test_poseidon2_reference_vectors(oracle):
    g = json.load(open(golden_path("poseidon2_t8.json")))
    for v in g["vectors"]:
        ins = [int(x, 16) for x in v["inputs"]]
        got = oracle.fr_ints(oracle.poseidon2_hash(oracle.fr_array(ins)))[0]
        assert got == int(v["hash"], 16)


def test_ntt_small_vs_naive_and_roundtrip(oracle):
    rnd = random.Random(3)
    for k in (1, 2, 5):
        n = 1 << k
        xs = [rnd.randrange(R) for _ in range(n)]
        w = oracle.fr_ints(oracle.root_of_unity(k))[0]
        naive = [sum(xs[j] * pow(w, i * j, R) for j in range(n)) % R for i in range(n)]
        assert oracle.fr_ints(oracle.ntt(oracle.fr_array(xs), k)) == naive
    xs = [rnd.randrange(R) for _ in range(1 << 10)]
    a = oracle.fr_array(xs)
    assert oracle.fr_ints(oracle.ntt(oracle.ntt(a, 10), 10, inverse=True)) == xs


def test_coset_extension_is_evaluation_on_zeta_coset(oracle):
    rnd = random.Random(5)
    k, ek = 3, 5
    cs = [rnd.randrange(R) for _ in range(1 << k)]
    ext = oracle.fr_ints(oracle.coeff_to_extended(oracle.fr_array(cs), k, ek))
    zeta = oracle.fr_ints(oracle.fr_zeta())[0]
    we = oracle.fr_ints(oracle.root_of_unity(ek))[0]
    for i in range(1 << ek):
        x = zeta * pow(we, i, R) % R
        assert ext[i] == sum(c * pow(x, j, R) for j, c in enumerate(cs)) % R
    back = oracle.fr_ints(oracle.extended_to_coeff(oracle.fr_array(ext), ek))
    assert back[: 1 << k] == cs and all(v == 0 for v in back[1 << k:])


def test_srs_raw_equals_ptau_and_g_lagrange(oracle):
    """crates/powers-of-tau/lib.rs:266-281 (raw_equals_perpetual) + g_lagrange reproduction"""
    raw = oracle.srs_from_halo2_raw(open(golden_path("ppot_0080_11_raw"), "rb").read())
    pt = oracle.srs_from_ptau(open(golden_path("ppot_0080_11.head.ptau"), "rb").read())
    assert raw.k == pt.k == 11
    assert np.array_equal(raw.g, pt.g)
    assert raw.g2 == pt.g2 and raw.s_g2 == pt.s_g2
    assert np.array_equal(raw.g_lagrange, pt.g_lagrange)      # derived by group FFT == stored
    c = json.load(open(golden_path("constants.json")))
    assert oracle.g1_affine_ints(raw.g[0:1])[0] == (1, 2)
    assert list(oracle.g1_affine_ints(raw.g[1:2])[0]) == [int(v, 16) for v in c["srs_k11_g1"]]
    assert list(oracle.g1_affine_ints(raw.g_lagrange[0:1])[0]) == [int(v, 16) for v in c["srs_k11_g_lagrange0"]]
    for p in oracle.g1_affine_ints(raw.g[:16]):
        assert (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0


def test_commit_equals_commit_lagrange(oracle):
    """crates/powers-of-tau/lib.rs:247-264 (test_commit_lagrange)"""
    raw = oracle.srs_from_halo2_raw(open(golden_path("ppot_0080_11_raw"), "rb").read())
    a = oracle.fr_array(list(range(2048)))
    b = oracle.ntt(a, 11, inverse=True)
    c1 = oracle.msm(raw.g, b)
    c2 = oracle.msm(raw.g_lagrange, a)
    assert np.array_equal(c1, c2)
    assert np.array_equal(c1, oracle.msm(raw.g, b, serial=True))
    assert oracle.g1_on_curve(c1)


def test_msm_small_vs_scalar_mul(oracle):
    rnd = random.Random(11)
    raw = oracle.srs_from_halo2_raw(open(golden_path("ppot_0080_11_raw"), "rb").read())
    for n in (1, 3, 5, 40):
        xs = [rnd.randrange(R) for _ in range(n)]
        acc = None
        for i, x in enumerate(xs):
            t = oracle.g1_mul(raw.g[i], oracle.fr_array([x])[0])
            acc = t if acc is None else oracle.g1_add(acc, t)
        assert np.array_equal(oracle.msm(raw.g[:n], oracle.fr_array(xs)), acc)
    # edge cases: zero scalars, empty
    z = oracle.msm(raw.g[:8], oracle.fr_array([0] * 8))
    assert not z.any()


def test_synthetic_srs_consistent(oracle):
    tau = 0x1234567890abcdef1234567890abcdef
    s = oracle.srs_synthetic(6, tau)
    assert oracle.g1_affine_ints(s.g[0:1])[0] == (1, 2)
    assert np.array_equal(s.g_lagrange, oracle.g1_lagrange_from_monomial(s.g, 6))
    a = oracle.fr_array([3 * i + 1 for i in range(64)])
    assert np.array_equal(oracle.msm(s.g, oracle.ntt(a, 6, inverse=True)), oracle.msm(s.g_lagrange, a))


def test_chacha20_rfc8439_block(oracle):
    # RFC 8439 section 2.3.2 uses a 32-bit counter + 96-bit nonce; with nonce = 0 and counter = 1 the
    # layouts coincide: check against the zero-key, counter-0 keystream of the original ChaCha20.
    ks = oracle.chacha20_block(bytes(32), 0)
    assert ks.hex().startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")
    d = oracle.fr_ints(oracle.fr_draw(bytes(32), 0).reshape(1, 4))[0]
    assert d == int.from_bytes(ks, "little") % R
