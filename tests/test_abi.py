"""CPU: the C-ABI library loads and exports every symbol include/shielder_hip.h declares; host-only
entry points behave; compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "shielder_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(shp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(shp):
    lib = shp.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/shielder_hip.h but not exported"


def test_field_front_door(shp):
    """crates/type-conversions/lib.rs:142-157, 209-213: 41 -> [0x29, 0, ...]; non-canonical rejected"""
    le = (41).to_bytes(32, "little")
    limbs = shp.fr_from_le32(le)
    assert shp.fr_to_le32(limbs) == le and le[0] == 0x29
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    with pytest.raises(shp.ShpError) as e:
        shp.fr_from_le32(r.to_bytes(32, "little"))
    assert e.value.code == 4
    assert shp.fr_to_le32(shp.fr_from_le32((r - 1).to_bytes(32, "little"))) == (r - 1).to_bytes(32, "little")


def test_front_door_matches_oracle(shp, oracle):
    import numpy as np
    for v in (0, 1, 41, 2**200 + 12345):
        a = shp.fr_from_le32(v.to_bytes(32, "little"))
        assert np.array_equal(a, oracle.fr_array([v])[0])


def test_no_cpu_fallback(shp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(shp.ShpError) as e:
        shp.Context(0)
    assert e.value.code == 2


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "zkos-monorepo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace(
                    "never imports `oracle/`", ""), f


def test_encode_calldata_matches_abi_encoding(shp, oracle):
    """halo2_verifier::encode_calldata (crates/halo2-verifier/src/lib/verifier_contract.rs:6-20): selector of
    verifyProof(bytes,uint256[]) and the head/tail ABI layout, against an independent encoding"""
    import pytest
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    sel = oracle.keccak256(b"verifyProof(bytes,uint256[])")[:4]
    assert sel.hex() == "1e8e1e13"
    for plen, inst in ((0, []), (1, [0]), (64, [1, 2, 3]), (3456, [R - 1, 5, 2**200 + 7, 0, 1, 2, 3, 4])):
        proof = bytes((i * 7 + 3) & 0xFF for i in range(plen))
        pad = (plen + 31) // 32 * 32
        want = (sel + (64).to_bytes(32, "big") + (96 + pad).to_bytes(32, "big") + plen.to_bytes(32, "big") + proof
                + bytes(pad - plen) + len(inst).to_bytes(32, "big") + b"".join(x.to_bytes(32, "big") for x in inst))
        assert shp.encode_calldata(proof, inst) == want
    with pytest.raises(shp.ShpError) as e:
        shp.encode_calldata(b"\x01", [R])
    assert e.value.code == 4


def test_secret_derivation_reference_kats(shp):
    """crates/shielder-account/src/secrets.rs:75-106: the reference's two known-answer tests, through the library"""
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    before = 0x375a07a9503d15a291307e33ad0c297c9768fea4712947172ad09f2df34d8015
    assert shp.derive_nullifier(15, 0xFF) == before % R != before
    before = 0xf4b3b097dfb3da737872bdf8b59a3b3723345dc147a0b8229608db69cfef6499
    assert shp.derive_id(16, 26, 45) == before % R != before
    # arbitrary 256-bit inputs are accepted (U256, not a field element)
    assert 0 <= shp.derive_id(2**256 - 1, 2**64 - 1, 2**32 - 1) < R


def test_plan_min_k_and_relayout_on_the_host():
    """shp_plan_min_k / shp_plan_retarget (no GPU): a circuit laid out for 2^11 rows that fits 2^9 is found to need k = 9, and
    its re-layout IS the plan the builder emits for k = 9, byte for byte (and back)"""
    import importlib
    shp = importlib.import_module("zkos-monorepo_amd")
    C = importlib.import_module("zkos-monorepo_amd.circuits")
    kw = dict(height=1, range_bits=4, range_chunks=8)
    p11, _ = C.build_withdraw_plan(k=11, **kw)
    p9, _ = C.build_withdraw_plan(k=9, **kw)
    assert shp.plan_min_k(p11) == 9 and shp.plan_min_k(p9) == 9
    assert shp.plan_retarget(p11, 9) == p9 and shp.plan_retarget(p9, 11) == p11 and shp.plan_retarget(p9, 9) == p9
    with pytest.raises(shp.ShpError) as e:
        shp.plan_retarget(p11, 8)
    assert e.value.code == shp.SHP_ERR_INVALID_ARG
    with pytest.raises(shp.ShpError) as e:
        shp.plan_min_k(p11[:-1])
    assert e.value.code == shp.SHP_ERR_FORMAT


def test_blinding_stream_is_chacha20_wide_reduction(oracle):
    """shp_blinding_stream (host): draw i = from_u512(ChaCha20(seed, block i)), the oracle's stream"""
    import importlib
    import numpy as np
    shp = importlib.import_module("zkos-monorepo_amd")
    PL = importlib.import_module("oracle.plonk")
    seed = bytes(range(32))
    assert np.array_equal(shp.blinding_stream(seed, 0, 300), PL.draws(seed, 0, 300))
    assert np.array_equal(shp.blinding_stream(seed, 2**33 + 5, 3), PL.draws(seed, 2**33 + 5, 3))
