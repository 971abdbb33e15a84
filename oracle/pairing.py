"""BN254 optimal-ate pairing in plain Python integers -- TEST INFRASTRUCTURE (oracle), never shipped.

Used by oracle/plonk.py:verify for proofs over a REAL SRS (no trapdoor): the final check of the
reference's verifier, e(lhs, g2) * e(W', -s g2) = 1
(/root/reference/crates/halo2-verifier/templates/Halo2Verifier.sol:204-219, 553-559: the EVM
ecPairing precompile 0x08; src/lib/codegen/pcs/bdfg21.rs:316-414 builds lhs / rhs).

Restated from the published algorithm (Vercauteren's optimal ate pairing on BN curves as used by
the alt_bn128 precompile, EIP-197): Miller loop over 6u+2, u = 4965661367192848881, two Frobenius
correction steps, final exponentiation (p^12 - 1) / r.  Tower: Fq2 = Fq[i]/(i^2 + 1),
Fq12 = Fq2[w]/(w^6 - xi), xi = 9 + i; G2 lives on the D-type twist y^2 = x^3 + 3/xi and is
untwisted by (x, y) -> (x w^2, y w^3).

Pinned by tests/test_oracle_fixtures.py: bilinearity, non-degeneracy, and the relation the
reference's own k = 11 SRS file must satisfy, e(g[1], g2) = e(g[0], s_g2).
"""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
U = 4965661367192848881
ATE = 6 * U + 2
XI = (9, 1)


# ---- Fq2 ---------------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)
def f2_scale(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_conj(a): return (a[0], (-a[1]) % P)


def f2_inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


F2_ZERO, F2_ONE = (0, 0), (1, 0)

# ---- Fq12 = Fq2[w] / (w^6 - xi): tuples of six Fq2 coefficients ---------------------------
F12_ONE = (F2_ONE,) + (F2_ZERO,) * 5


def f12_mul(a, b):
    c = [F2_ZERO] * 11
    for i in range(6):
        ai = a[i]
        if ai == F2_ZERO:
            continue
        for j in range(6):
            bj = b[j]
            if bj == F2_ZERO:
                continue
            c[i + j] = f2_add(c[i + j], f2_mul(ai, bj))
    return tuple(f2_add(c[k], f2_mul(XI, c[k + 6])) if k < 5 else c[k] for k in range(6))


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


# ---- curve arithmetic (affine, None = point at infinity) ---------------------------------
def g1_on_curve(pt):
    return pt is None or (pt[1] * pt[1] - pt[0] ** 3 - 3) % P == 0


B2 = f2_mul((3, 0), f2_inv(XI))


def g2_on_curve(q):
    return q is None or f2_sub(f2_sqr(q[1]), f2_add(f2_mul(f2_sqr(q[0]), q[0]), B2)) == F2_ZERO


def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if f2_add(a[1], b[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_scale(f2_sqr(a[0]), 3), f2_inv(f2_scale(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(a[0], x)), a[1]))


def g2_mul(q, k):
    r = None
    while k:
        if k & 1:
            r = g2_add(r, q)
        q = g2_add(q, q)
        k >>= 1
    return r


def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def g1_mul(pt, k):
    r = None
    while k:
        if k & 1:
            r = g1_add(r, pt)
        pt = g1_add(pt, pt)
        k >>= 1
    return r


def g1_neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


# Frobenius on the twist: pi(x, y) = (conj(x) g12, conj(y) g13), pi^2(x, y) = (x g22, y g23)
G12 = f2_pow(XI, (P - 1) // 3)
G13 = f2_pow(XI, (P - 1) // 2)
G22 = f2_pow(XI, (P * P - 1) // 3)
G23 = f2_pow(XI, (P * P - 1) // 2)


def _line(t, q, p1):
    """line through the untwisted t, q (tangent when t == q) evaluated at p1 in G1, and t + q.
    l = yP - lam xP w + (lam x_t - y_t) w^3"""
    if t[0] == q[0] and t[1] == q[1]:
        lam = f2_mul(f2_scale(f2_sqr(t[0]), 3), f2_inv(f2_scale(t[1], 2)))
    else:
        lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), t[0]), q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])
    l = ((p1[1], 0), f2_scale(f2_neg(lam), p1[0]), F2_ZERO, f2_sub(f2_mul(lam, t[0]), t[1]), F2_ZERO, F2_ZERO)
    return l, (x3, y3)


def miller(p1, q):
    """Miller function f_{6u+2,Q}(P) times the two Frobenius lines (before the final exponentiation)"""
    if p1 is None or q is None:
        return F12_ONE
    f, t = F12_ONE, q
    for bit in bin(ATE)[3:]:
        l, t = _line(t, t, p1)
        f = f12_mul(f12_mul(f, f), l)
        if bit == "1":
            l, t = _line(t, q, p1)
            f = f12_mul(f, l)
    q1 = (f2_mul(f2_conj(q[0]), G12), f2_mul(f2_conj(q[1]), G13))
    q2n = (f2_mul(q[0], G22), f2_neg(f2_mul(q[1], G23)))
    l, t = _line(t, q1, p1)
    f = f12_mul(f, l)
    l, t = _line(t, q2n, p1)
    return f12_mul(f, l)


FINAL_EXP = (P ** 12 - 1) // R


def pairing(p1, q):
    return f12_pow(miller(p1, q), FINAL_EXP)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with one final exponentiation (what the EVM precompile answers)"""
    f = F12_ONE
    for p1, q in pairs:
        f = f12_mul(f, miller(p1, q))
    return f12_pow(f, FINAL_EXP) == F12_ONE


def g2_from_bytes(b: bytes):
    """128 bytes canonical little-endian x.c0 | x.c1 | y.c0 | y.c1 (oracle/srs.c export)"""
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
    q = ((v[0], v[1]), (v[2], v[3]))
    if not g2_on_curve(q):
        raise ValueError("G2 point not on the twist")
    return q


def check_kzg(lhs, rhs, g2_bytes: bytes, s_g2_bytes: bytes) -> bool:
    """e(lhs, g2) * e(rhs, -s g2) == 1 (Halo2Verifier.sol:553-559); lhs / rhs = (x, y) ints or None"""
    g2, s_g2 = g2_from_bytes(g2_bytes), g2_from_bytes(s_g2_bytes)
    if not (g1_on_curve(lhs) and g1_on_curve(rhs)):
        return False
    neg_s = (s_g2[0], f2_neg(s_g2[1]))
    return pairing_product_is_one([(lhs, g2), (rhs, neg_s)])
