"""CPU: the TEE wire protocol mirror (zkos-monorepo_amd/serve.py) against the reference's message shapes
(tee/crates/shielder-prover-common/src/protocol.rs, tee/crates/shielder-prover-tee/src/circuits/withdraw.rs),
with a fake backend -- no GPU.  The GPU end-to-end is tests/test_gpu_prover.py::test_serve_end_to_end."""
import asyncio
import base64
import importlib
import json
import struct

import pytest

S = importlib.import_module("zkos-monorepo_amd.serve")
R = S.R_MOD


def le(x):
    return list(int(x).to_bytes(32, "little"))


def make_inputs(height, bump=0):
    vals = {"id": 5 + bump, "nullifier_old": 6, "account_balance_old": 10**18, "token_address": 0xABCD, "value": 10**17,
            "nullifier_new": 8, "commitment": 9, "mac_salt": 10}
    d = {k: le(v) for k, v in vals.items()}
    d["path"] = [b for i in range(7 * height) for b in le(100 + i)]
    return d


def make_request(inputs, ctype="Withdraw"):
    payload = {"circuit_type": ctype, "user_public_key": [1, 2, 3], "circuit_inputs": list(json.dumps(inputs).encode())}
    return {"GenerateProof": {"payload": list(json.dumps(payload).encode())}}


class FakeBackend:
    n_height = 2

    def __init__(self):
        self.calls = []

    def prove(self, wit, seed):
        self.calls.append((wit, seed))
        return b"PROOF" + bytes([len(self.calls)]), list(range(1, 9))


def test_witness_vector_order_and_validation():
    h = 2
    wit = S.withdraw_witness_bytes(make_inputs(h), h)
    ints = [int.from_bytes(wit[i:i + 32], "little") for i in range(0, len(wit), 32)]
    # argument order of WithdrawCircuit::prove: id, nullifier_old, balance, token, path..., value, nullifier_new, commitment, mac_salt
    assert ints == [5, 6, 10**18, 0xABCD] + [100 + i for i in range(14)] + [10**17, 8, 9, 10]
    bad = make_inputs(h); bad["id"] = le(0)[:-1]
    with pytest.raises(S.ProtocolError):
        S.withdraw_witness_bytes(bad, h)
    bad = make_inputs(h); bad["value"] = list(R.to_bytes(32, "little"))        # vec_to_f panics on non-canonical input
    with pytest.raises(S.ProtocolError):
        S.withdraw_witness_bytes(bad, h)
    bad = make_inputs(h); del bad["mac_salt"]
    with pytest.raises(S.ProtocolError):
        S.withdraw_witness_bytes(bad, h)
    bad = make_inputs(h); bad["path"] = bad["path"][:-32]
    with pytest.raises(S.ProtocolError):
        S.withdraw_witness_bytes(bad, h)


def test_pub_inputs_json_shape():
    out = json.loads(S.withdraw_pub_inputs_json([1, 2, 3, 4, 5, 6, 7, R - 1]))
    assert list(out) == ["merkle_root", "h_nullifier_old", "h_note_new", "withdrawal_value", "token_address", "commitment",
                         "mac_salt", "mac_commitment"]
    assert out["merkle_root"] == le(1) and out["mac_commitment"] == le(R - 1)


def test_server_round_trips_over_tcp():
    be, dep = FakeBackend(), FakeBackend()

    async def run():
        server = S.Server({"Withdraw": be, "Deposit": dep}, seed_source=lambda n: bytes([7]) * n)
        srv = await server.serve(port=0)
        port = srv.sockets[0].getsockname()[1]
        async with srv:
            assert await S.request("127.0.0.1", port, "Ping") == "Pong"
            pk = await S.request("127.0.0.1", port, "TeePublicKey")
            assert set(pk["TeePublicKey"]) == {"public_key", "attestation_document"}
            # two requests on ONE connection, frames are 4-byte big-endian length + JSON
            reader, writer = await asyncio.open_connection("127.0.0.1", port)
            for bump in (0, 1):
                body = json.dumps(make_request(make_inputs(2, bump))).encode()
                writer.write(struct.pack(">I", len(body)) + body)
                await writer.drain()
                (n,) = struct.unpack(">I", await reader.readexactly(4))
                resp = json.loads(await reader.readexactly(n))
                ep = resp["EncryptedProof"]
                assert base64.b64decode(ep["proof"]) == b"PROOF" + bytes([bump + 1])
                assert json.loads(base64.b64decode(ep["pub_inputs"]))["h_nullifier_old"] == le(2)
            writer.close()
            # the Deposit circuit has its own argument order (value, commitment, nullifier_new) and field names
            r = await S.request("127.0.0.1", port, make_request(make_inputs(2), "Deposit"))
            assert set(json.loads(base64.b64decode(r["EncryptedProof"]["pub_inputs"]))) == set(S.DEPOSIT_PUB_FIELDS)
            # protocol errors drop the connection (the reference returns Err from handle_request)
            for bad in ({"GenerateProof": {"payload": list(b"not json")}}, make_request(make_inputs(2), "NewAccount"), "Nope"):
                with pytest.raises(asyncio.IncompleteReadError):
                    await S.request("127.0.0.1", port, bad)
        return server.served

    assert asyncio.run(run()) == 3
    dw = dep.calls[0][0]
    tail = [int.from_bytes(dw[i:i + 32], "little") for i in range(32 * 18, 32 * 22, 32)]
    assert tail == [10**17, 9, 8, 10]                      # value, commitment, nullifier_new, mac_salt
    assert len(be.calls) == 2 and be.calls[0][1] == bytes([7]) * 32 and len(be.calls[0][0]) == 32 * (8 + 14)


def test_native_wire_codec_matches_the_python_mirror():
    """shp_tee_decode_request / shp_tee_encode_proof_response (csrc/wire.hip, host code: runs without a GPU) against the
    Python mirror of the same messages: same witness vector for Withdraw and Deposit (argument orders differ), unit
    variants, whitespace, unknown fields ignored as serde does, and every malformed frame rejected with SHP_ERR_FORMAT"""
    B = importlib.import_module("zkos-monorepo_amd.binding")
    for ctype, fields in (("Withdraw", S.WITHDRAW_INPUT_FIELDS), ("Deposit", S.DEPOSIT_INPUT_FIELDS)):
        for height in (1, 2, 13):
            inputs = make_inputs(height, bump=height)
            body = json.dumps(make_request(inputs, ctype)).encode()
            kind, c, wit = B.tee_decode_request(body, height)
            assert (kind, c) == (B.TEE_GENERATE_PROOF, ctype)
            assert wit == S.withdraw_witness_bytes(inputs, height, fields)
    # NewAccount: nine scalar arguments, no path (height is irrelevant)
    na = {f: le(1000 + i) for i, f in enumerate(S.NEW_ACCOUNT_INPUT_FIELDS)}
    kind, c, wit = B.tee_decode_request(json.dumps(make_request(na, "NewAccount")).encode(), 13)
    assert (kind, c) == (B.TEE_GENERATE_PROOF, "NewAccount")
    assert wit == S.withdraw_witness_bytes(na, 13, S.NEW_ACCOUNT_INPUT_FIELDS) and len(wit) == 9 * 32
    d = json.loads(B.tee_encode_proof_response("NewAccount", b"\x01\x02", list(range(13))))
    assert base64.b64decode(d["EncryptedProof"]["pub_inputs"]) == S.withdraw_pub_inputs_json(list(range(13)), S.NEW_ACCOUNT_PUB_FIELDS)
    # pretty-printed JSON and an extra field
    inputs = make_inputs(2)
    payload = {"circuit_type": "Withdraw", "extra": {"a": [1, "x", None]}, "user_public_key": [], "circuit_inputs":
               list(json.dumps(dict(inputs, unknown=[1, 2]), indent=1).encode())}
    body = json.dumps({"GenerateProof": {"payload": list(json.dumps(payload, indent=2).encode())}}, indent=3).encode()
    assert B.tee_decode_request(body, 2)[2] == S.withdraw_witness_bytes(inputs, 2)
    # ignored fields: every JSON value form, nested up to the depth limit, repeated unknown keys
    ok_extra = b'{"u":[-0.5e+3,true,false,null,"s\\n\\"",{},[],{"k":[{"z":0}]}],"u":1,"v":' + b"[" * 128 + b"]" * 128 + b","
    body = json.dumps(make_request(inputs, "Withdraw")).encode().replace(b'{"payload"', ok_extra + b'"payload"', 1)
    assert B.tee_decode_request(body, 2)[2] == S.withdraw_witness_bytes(inputs, 2)
    assert B.tee_decode_request(b'"Ping"', 2) == (B.TEE_PING, None, None)
    assert B.tee_decode_request(b'  "TeePublicKey"\n', 2) == (B.TEE_PUBLIC_KEY, None, None)

    def frame(inp, ctype="Withdraw", mutate=None):
        req = make_request(inp, ctype)
        return (mutate(req) if mutate else json.dumps(req)).encode()

    bad = []
    nc = make_inputs(2); nc["value"] = le(R)                                        # not canonical (vec_to_f panics)
    bad.append(frame(nc))
    short = make_inputs(2); short["path"] = short["path"][:-1]
    bad.append(frame(short))
    missing = make_inputs(2); del missing["mac_salt"]
    bad.append(frame(missing))
    big = make_inputs(2); big["id"] = [256] + big["id"][1:]
    bad.append(frame(big))
    bad.append(frame(make_inputs(2), "NewAccount"))                                 # no key for it
    bad.append(frame(make_inputs(2), "Transfer"))
    bad.append(frame(make_inputs(2)) + b"x")                                        # trailing bytes
    bad.append(frame(make_inputs(2))[:-3])                                          # truncated
    bad.append(b'"Pong"')
    bad.append(b'{"GenerateProof":{}}')
    bad.append(b'{"Generate":{"payload":[]}}')
    bad.append(b"")
    # an ignored field nested beyond serde_json's depth limit (128): refused, and never by exhausting the stack --
    # 100 000 levels are a 200 KB frame, far below the 8 MB frame cap
    for depth in (129, 100000):
        for o, c in (("[", "]"), ('{"a":', "}")):
            bad.append(b'{"GenerateProof":{"x":' + o.encode() * depth + b"1" + c.encode() * depth + b',"payload":[]}}')
    bad.append(b'{"GenerateProof":{"x":' + b"[" * 100000)                         # deep and truncated
    dup = json.dumps(make_request(make_inputs(2), "Withdraw"))
    bad.append(dup.replace('{"GenerateProof":', '{"GenerateProof":{"payload":[]},"GenerateProof":', 1).encode())   # variant twice
    bad.append(dup.replace('"payload":', '"payload":[],"payload":', 1).encode())                                    # field twice
    twice = make_inputs(2)
    inner_dup = json.dumps(twice).replace('"id":', '"id":' + json.dumps(twice["id"]) + ',"id":', 1)
    pl = {"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(inner_dup.encode())}
    bad.append(json.dumps({"GenerateProof": {"payload": list(json.dumps(pl).encode())}}).encode())                 # witness field twice
    pl2 = json.dumps({"circuit_type": "Withdraw", "user_public_key": [], "circuit_inputs": list(json.dumps(twice).encode())})
    bad.append(json.dumps({"GenerateProof": {"payload": list(pl2.replace('"circuit_type":', '"circuit_type":"Deposit","circuit_type":', 1).encode())}}).encode())
    for junk in (b"nul", b"tru", b"True", b"01", b"1.", b"-", b"1e", b"+1", b".5", b"0x10", b"NaN", b"'a'", b'"\\q"', b'"ab"'):
        bad.append(b'{"GenerateProof":{"x":' + junk + b',"payload":[]}}')                                          # not JSON values
    for b in bad:
        with pytest.raises(B.ShpError) as e:
            B.tee_decode_request(b, 2)
        assert e.value.code == 4, b[:60]
    # response: byte-identical pub_inputs JSON, proof round trip
    for ctype, pub in (("Withdraw", S.WITHDRAW_PUB_FIELDS), ("Deposit", S.DEPOSIT_PUB_FIELDS)):
        inst = [R - 1 - i for i in range(8)]
        for plen in (0, 1, 2, 3, 3456):
            proof = bytes((7 * i) & 255 for i in range(plen))
            d = json.loads(B.tee_encode_proof_response(ctype, proof, inst))
            assert base64.b64decode(d["EncryptedProof"]["proof"]) == proof
            assert base64.b64decode(d["EncryptedProof"]["pub_inputs"]) == S.withdraw_pub_inputs_json(inst, pub)


def test_native_wire_codec_fuzz():
    """random damage to valid frames: the native parser never accepts what the Python mirror of the protocol rejects (and
    returns the same witness when both accept); it may be stricter (it refuses number forms and escapes serde would take
    but no client produces)"""
    import random
    B = importlib.import_module("zkos-monorepo_amd.binding")
    rnd = random.Random(2024)
    height = 2
    base = {c: json.dumps(make_request(make_inputs(height, bump=i), c)).encode() for i, c in enumerate(("Withdraw", "Deposit"))}
    base["NewAccount"] = json.dumps(make_request({f: le(77 + i) for i, f in enumerate(S.NEW_ACCOUNT_INPUT_FIELDS)}, "NewAccount")).encode()

    def mirror(body):
        try:
            req = S.loads_strict(body)
            if not isinstance(req, dict) or set(req) != {"GenerateProof"}:
                return None
            ctype, inputs = S.decode_payload(S._bytes_field(req["GenerateProof"].get("payload"), "payload"))
            if not isinstance(inputs, dict):
                return None
            return ctype, S.withdraw_witness_bytes(inputs, height, S.INPUT_FIELDS[ctype])
        except (S.ProtocolError, ValueError, KeyError, TypeError, AttributeError, RecursionError):
            return None

    accepted = stricter = 0
    for it in range(1500):
        body = bytearray(base[rnd.choice(list(base))])
        for _ in range(rnd.choice((1, 1, 1, 2, 3))):
            kind = rnd.randrange(5)
            pos = rnd.randrange(len(body))
            if kind == 0:
                body[pos] = rnd.choice(b'0123456789,[]{}":x -')
            elif kind == 1:
                del body[pos:pos + rnd.randrange(1, 6)]
            elif kind == 2:
                body[pos:pos] = bytes(rnd.choice(b'0123456789,[]{}":') for _ in range(rnd.randrange(1, 4)))
            elif kind == 3:
                body = body[:pos]
            else:
                body[pos] ^= 1 << rnd.randrange(8)
        body = bytes(body)
        want = mirror(body)
        try:
            kind_, ctype, wit = B.tee_decode_request(body, height)
            got = (ctype, wit) if kind_ == B.TEE_GENERATE_PROOF else None
        except B.ShpError as e:
            assert e.code == 4
            got = None
        if got is not None:
            accepted += 1
            assert want is not None and got == want, (it, body[:80])
        elif want is not None:
            stricter += 1
    assert accepted > 20 and stricter <= 15, (accepted, stricter)


def test_python_mirror_refuses_what_serde_refuses():
    """the Python path of the server (used when the native codec is unavailable) agrees with the native one on duplicate
    fields and on nesting deep enough to exhaust a recursive parser"""
    with pytest.raises(S.ProtocolError):
        S.loads_strict(b'{"a":1,"a":2}')
    with pytest.raises(S.ProtocolError):
        S.loads_strict(b"[" * 100000 + b"]" * 100000)
    with pytest.raises(S.ProtocolError):
        S.loads_strict(b'{"a":')
    assert S.loads_strict(b'{"a":{"a":1},"b":[{"a":2}]}') == {"a": {"a": 1}, "b": [{"a": 2}]}


def test_native_server_answers_health_requests_and_drops_bad_peers():
    """the library's own TCP front (shp_server, csrc/server.hip) without a pool -- no GPU: Ping and TeePublicKey are
    answered in the reference's encodings, several requests per connection; a malformed frame, a GenerateProof for a
    circuit type without a key and an oversized length prefix each end the connection (the reference drops the client on
    any error, tee/crates/shielder-prover-tee/src/server.rs:70-73); the server stops cleanly with a client still connected"""
    import socket
    shp = importlib.import_module("zkos-monorepo_amd")
    srv = shp.NativeServer({}, height=2)
    try:
        assert srv.port > 0

        def connect():
            s = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
            return s

        def send(s, body: bytes):
            s.sendall(struct.pack(">I", len(body)) + body)

        def recv(s):
            hdr = b""
            while len(hdr) < 4:
                chunk = s.recv(4 - len(hdr))
                if not chunk:
                    return None                                   # the server closed the connection
                hdr += chunk
            (n,) = struct.unpack(">I", hdr)
            body = b""
            while len(body) < n:
                chunk = s.recv(n - len(body))
                assert chunk
                body += chunk
            return body

        c = connect()
        for _ in range(3):                                        # a connection serves requests until the peer leaves
            send(c, b'"Ping"')
            assert json.loads(recv(c)) == "Pong"
        send(c, b'"TeePublicKey"')
        assert json.loads(recv(c)) == {"TeePublicKey": {"public_key": "", "attestation_document": ""}}
        c.close()
        for bad in (b'{"Nope":1}', b"[[[[", S.encode_frame(make_request(make_inputs(2)))[4:]):     # the last: no Withdraw key loaded
            c = connect()
            send(c, bad)
            assert recv(c) is None
            c.close()
        c = connect()
        c.sendall(struct.pack(">I", 9 * 1024 * 1024))             # beyond LengthDelimitedCodec's 8 MiB
        assert recv(c) is None
        c.close()
        served, failed, conns = srv.stats()
        assert (served, failed, conns) == (0, 4, 5)
        idle = connect()                                          # still connected when the server stops
        send(idle, b'"Ping"')
        assert json.loads(recv(idle)) == "Pong"
    finally:
        srv.close()
    assert recv(idle) is None
    idle.close()


def _frame_recv(s):
    hdr = b""
    while len(hdr) < 4:
        chunk = s.recv(4 - len(hdr))
        if not chunk:
            return None
        hdr += chunk
    (n,) = struct.unpack(">I", hdr)
    body = b""
    while len(body) < n:
        chunk = s.recv(n - len(body))
        assert chunk
        body += chunk
    return body


def test_native_server_drops_stalled_and_idle_peers():
    """no thread of the native front waits without a bound (the reference bounds its requests with the task pool's spawn
    and run timeouts, tee/crates/shielder-prover-server/src/main.rs:42-45): a peer that sends three of the four length
    bytes, one that announces a frame and sends half of it, and one that goes silent between requests are each dropped
    after their timeout, and counted; a peer that keeps talking is not"""
    import socket
    import time
    shp = importlib.import_module("zkos-monorepo_amd")
    srv = shp.NativeServer({}, height=2, recv_timeout_ms=300, idle_timeout_ms=1200)
    try:
        t0 = time.monotonic()
        a = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        a.sendall(b"\x00\x00\x00")                                # three of four length bytes
        b = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        b.sendall(struct.pack(">I", 100) + b'"Pi')                # a frame of 100 bytes, three of them sent
        good = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        good.sendall(struct.pack(">I", 6) + b'"Ping"')
        assert json.loads(_frame_recv(good)) == "Pong"
        assert _frame_recv(a) is None and _frame_recv(b) is None  # closed by the server
        stalled = time.monotonic() - t0
        assert 0.25 < stalled < 3.0, stalled
        assert srv.timeouts() == (0, 0, 2)
        # the talking peer is still served after the stalled ones were dropped, then dropped itself once idle
        good.sendall(struct.pack(">I", 6) + b'"Ping"')
        assert json.loads(_frame_recv(good)) == "Pong"
        t1 = time.monotonic()
        assert _frame_recv(good) is None
        assert 1.0 < time.monotonic() - t1 < 4.0
        assert srv.timeouts() == (0, 0, 2)                        # idling out is not an error
        served, failed, conns = srv.stats()
        assert (served, failed, conns) == (0, 2, 3)
        for s_ in (a, b, good):
            s_.close()
    finally:
        srv.close()


def _http(port, method, path, body=None, headers=None, raw=None):
    import http.client
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
    c.request(method, path, body=body, headers=headers or {})
    r = c.getresponse()
    data = r.read()
    c.close()
    return r.status, dict((k.lower(), v) for k, v in r.getheaders()), data


def test_native_server_http_front():
    """the reference's host-side HTTP server (tee/crates/shielder-prover-server/src/main.rs:47-59) on the same listener:
    GET /health and /public_key answer the Response as JSON, POST /proof takes {"payload": base64}; without a key for the
    circuit type the request fails like a refused enclave request (500, error.rs:35-38); the extractor's refusals
    (content type, body syntax), the body limit (413), unknown routes and methods are answered, not dropped"""
    import http.client
    shp = importlib.import_module("zkos-monorepo_amd")
    srv = shp.NativeServer({}, height=2, max_http_body=64 * 1024)
    try:
        st, hd, body = _http(srv.port, "GET", "/health")
        assert st == 200 and json.loads(body) == "Pong" and hd["content-type"] == "application/json"
        assert hd["access-control-allow-origin"] == "*"           # CorsLayer::permissive
        st, _, body = _http(srv.port, "GET", "/public_key")
        assert st == 200 and json.loads(body) == {"TeePublicKey": {"public_key": "", "attestation_document": ""}}
        payload = json.dumps({"circuit_type": "Withdraw", "user_public_key": [1], "circuit_inputs":
                              list(json.dumps(make_inputs(2)).encode())}).encode()
        req = json.dumps({"payload": base64.b64encode(payload).decode()}).encode()
        js = {"Content-Type": "application/json"}
        st, _, body = _http(srv.port, "POST", "/proof", req, js)
        assert st == 500 and b"TEE Proving Server error" in body and b"no proving key" in body
        st, _, body = _http(srv.port, "POST", "/proof", req)      # no content type: the Json extractor refuses
        assert st == 415
        st, _, body = _http(srv.port, "POST", "/proof", b'{"payload": 5}', js)
        assert st == 422
        st, _, body = _http(srv.port, "POST", "/proof", b'{"payload": "@@@="}', js)
        assert st == 422 and b"base64" in body
        bad_payload = json.dumps({"payload": base64.b64encode(b'{"circuit_type":"Nope"}').decode()}).encode()
        st, _, body = _http(srv.port, "POST", "/proof", bad_payload, js)
        assert st == 500
        st, _, _ = _http(srv.port, "POST", "/proof", b"x" * (64 * 1024 + 1), js)
        assert st == 413
        assert _http(srv.port, "GET", "/nope")[0] == 404
        assert _http(srv.port, "GET", "/proof")[0] == 405
        assert _http(srv.port, "OPTIONS", "/proof")[0] == 200
        # several requests on one keep-alive connection
        c = http.client.HTTPConnection("127.0.0.1", srv.port, timeout=10)
        for _ in range(3):
            c.request("GET", "/health")
            r = c.getresponse()
            assert r.status == 200 and json.loads(r.read()) == "Pong"
        c.close()
        # both fronts side by side
        import socket
        f = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        f.sendall(struct.pack(">I", 6) + b'"Ping"')
        assert json.loads(_frame_recv(f)) == "Pong"
        f.close()
    finally:
        srv.close()


def test_http_body_and_payload_decoders():
    """shp_http_decode_proof_body + shp_tee_decode_payload == shp_tee_decode_request on the same payload"""
    shp = importlib.import_module("zkos-monorepo_amd")
    inputs = make_inputs(2, bump=3)
    req = make_request(inputs, "Deposit")
    payload = bytes(req["GenerateProof"]["payload"])
    kind, ctype, wit = shp.tee_decode_request(json.dumps(req).encode(), 2)
    body = json.dumps({"ignored": [1, {"a": None}], "payload": base64.b64encode(payload).decode()}).encode()
    assert shp.http_decode_proof_body(body) == payload
    assert shp.tee_decode_payload(payload, 2) == (ctype, wit) and ctype == "Deposit"
    for n in range(0, 7):                                         # every padding length
        raw = bytes(range(n))
        assert shp.http_decode_proof_body(json.dumps({"payload": base64.b64encode(raw).decode()}).encode()) == raw
    for bad in (b'{}', b'{"payload":"QQ"}', b'{"payload":"QQ=A"}', b'{"payload":"QR=="}', b'{"payload":"A","payload":"A"}', b'[1]'):
        with pytest.raises(shp.ShpError):
            shp.http_decode_proof_body(bad)
