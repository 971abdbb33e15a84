"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE, never product code).

Field elements cross as numpy uint64 arrays of shape (..., 4): little-endian limbs in
Montgomery form (R = 2^256), the representation of the reference's SRS files.
G1 affine points are (..., 8) uint64 = x limbs | y limbs; Jacobian (..., 12).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R = 1 << 256


def build(force=False):
    so = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _DIR, "-j8"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


# ---------------------------------------------------------------- conversions (python bigints)
def to_mont(x, p=R_MOD):
    return (x % p) * MONT_R % p


def from_mont(x, p=R_MOD):
    return x * pow(MONT_R, -1, p) % p


def int_to_limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def fr_array(ints, p=R_MOD):
    """list of python ints (canonical) -> (n,4) uint64 Montgomery array"""
    a = np.zeros((len(ints), 4), dtype=np.uint64)
    for i, x in enumerate(ints):
        a[i] = int_to_limbs(to_mont(x, p))
    return a


def fr_ints(arr, p=R_MOD):
    """(n,4) uint64 Montgomery array -> list of canonical python ints"""
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [from_mont(limbs_to_int(row), p) for row in arr]


def g1_affine_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 8)
    return [(from_mont(limbs_to_int(r[:4]), Q_MOD), from_mont(limbs_to_int(r[4:]), Q_MOD)) for r in arr]


def g1_affine_array(pts):
    a = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        a[i, :4] = int_to_limbs(to_mont(x, Q_MOD))
        a[i, 4:] = int_to_limbs(to_mont(y, Q_MOD))
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---------------------------------------------------------------- hashes
def keccak256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_keccak256(C.c_char_p(data), C.c_size_t(len(data)), out)
    return out.raw


def chacha20_block(key: bytes, counter: int) -> bytes:
    out = C.create_string_buffer(64)
    lib().orc_chacha20_block(C.c_char_p(key), C.c_uint64(counter), out)
    return out.raw


def fr_draw(seed: bytes, counter: int):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_draw(C.c_char_p(seed), C.c_uint64(counter), _p(out))
    return out


def poseidon2_permute(state):
    s = _c(state).copy().reshape(8, 4)
    lib().orc_poseidon2_permute(_p(s))
    return s


def poseidon2_hash(inputs):
    a = _c(inputs).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_poseidon2_hash(_p(a), C.c_int(a.shape[0]), _p(out))
    return out


# ---------------------------------------------------------------- poly
def root_of_unity(k):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_root_of_unity(C.c_int(k), _p(out))
    return out


def fr_delta():
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_delta(_p(out))
    return out


def fr_zeta():
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_zeta(_p(out))
    return out


def ntt(a, k, inverse=False):
    """natural-order NTT over the 2^k domain; inverse includes the 1/n scaling"""
    a = _c(a).copy().reshape(1 << k, 4)
    (lib().orc_intt if inverse else lib().orc_fntt)(_p(a), C.c_int(k))
    return a


def coeff_to_extended(coeff, k, ext_k):
    coeff = _c(coeff).reshape(1 << k, 4)
    out = np.zeros((1 << ext_k, 4), dtype=np.uint64)
    lib().orc_coeff_to_extended(_p(coeff), C.c_int(k), C.c_int(ext_k), _p(out))
    return out


def extended_to_coeff(ext, ext_k):
    a = _c(ext).copy().reshape(1 << ext_k, 4)
    lib().orc_extended_to_coeff(_p(a), C.c_int(ext_k))
    return a


def eval_poly(coeff, x):
    coeff = _c(coeff).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_eval_poly(_p(coeff), C.c_size_t(coeff.shape[0]), _p(_c(x)), _p(out))
    return out


def batch_invert(a):
    a = _c(a).copy().reshape(-1, 4)
    lib().orc_batch_invert(_p(a), C.c_size_t(a.shape[0]))
    return a


def num_threads():
    return int(lib().orc_num_threads())


# ---------------------------------------------------------------- curve / msm
def msm(bases, scalars, serial=False):
    """returns the affine result as an (8,) uint64 array"""
    bases = _c(bases).reshape(-1, 8)
    scalars = _c(scalars).reshape(-1, 4)
    assert bases.shape[0] >= scalars.shape[0]
    j = np.zeros(12, dtype=np.uint64)
    f = lib().orc_msm_serial if serial else lib().orc_msm
    f(_p(bases), _p(scalars), C.c_size_t(scalars.shape[0]), _p(j))
    out = np.zeros(8, dtype=np.uint64)
    lib().g1j_to_affine(_p(out), _p(j))
    return out


def g1_on_curve(pt):
    return bool(lib().g1a_on_curve(_p(_c(pt))))


def g1_mul(pt, scalar):
    j = np.zeros(12, dtype=np.uint64)
    lib().g1j_from_affine_export(_p(j), _p(_c(pt)))
    lib().g1j_mul(_p(j), _p(j), _p(_c(scalar)))
    out = np.zeros(8, dtype=np.uint64)
    lib().g1j_to_affine(_p(out), _p(j))
    return out


def g1_add(p, q):
    a = np.zeros(12, dtype=np.uint64)
    b = np.zeros(12, dtype=np.uint64)
    lib().g1j_from_affine_export(_p(a), _p(_c(p)))
    lib().g1j_from_affine_export(_p(b), _p(_c(q)))
    lib().g1j_add(_p(a), _p(a), _p(b))
    out = np.zeros(8, dtype=np.uint64)
    lib().g1j_to_affine(_p(out), _p(a))
    return out


def g1_lagrange_from_monomial(g, k):
    g = _c(g).reshape(1 << k, 8)
    out = np.zeros_like(g)
    lib().orc_g1_lagrange_from_monomial(_p(g), C.c_int(k), _p(out))
    return out


# ---------------------------------------------------------------- SRS
class _Srs(C.Structure):
    _fields_ = [("k", C.c_uint32), ("n", C.c_size_t), ("g", C.c_void_p), ("g_lagrange", C.c_void_p),
                ("g2", C.c_uint8 * 128), ("s_g2", C.c_uint8 * 128)]


class Srs:
    def __init__(self, k, g, g_lagrange, g2, s_g2):
        self.k, self.n = k, 1 << k
        self.g, self.g_lagrange, self.g2, self.s_g2 = g, g_lagrange, g2, s_g2


def _take(s):
    n = s.n
    g = np.ctypeslib.as_array(C.cast(s.g, C.POINTER(C.c_uint64)), shape=(n, 8)).copy()
    gl = np.ctypeslib.as_array(C.cast(s.g_lagrange, C.POINTER(C.c_uint64)), shape=(n, 8)).copy()
    out = Srs(int(s.k), g, gl, bytes(s.g2), bytes(s.s_g2))
    lib().orc_srs_free(C.byref(s))
    return out


def srs_from_ptau(data: bytes) -> Srs:
    s = _Srs()
    rc = lib().orc_srs_from_ptau(C.c_char_p(data), C.c_size_t(len(data)), C.byref(s))
    if rc != 0:
        raise ValueError(f"ptau parse failed rc={rc}")
    return _take(s)


def srs_from_halo2_raw(data: bytes) -> Srs:
    s = _Srs()
    rc = lib().orc_srs_from_halo2_raw(C.c_char_p(data), C.c_size_t(len(data)), C.byref(s))
    if rc != 0:
        raise ValueError(f"raw parse failed rc={rc}")
    return _take(s)


def srs_synthetic(k: int, tau: int) -> Srs:
    s = _Srs()
    t = fr_array([tau])
    lib().orc_srs_synthetic(C.c_uint32(k), _p(t), C.byref(s))
    return _take(s)
