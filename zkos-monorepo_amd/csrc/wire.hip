// TEE wire protocol, native side: decode of a GenerateProof request frame and encode of the EncryptedProof response.
//
// The reference's messages are serde_json values whose byte vectors are JSON arrays of numbers, nested three deep
// (Request.payload -> Payload.circuit_inputs -> *ProveInputsBytes fields: /root/reference/tee/crates/
// shielder-prover-common/src/protocol.rs:14-68, tee/crates/shielder-prover-tee/src/circuits/withdraw.rs:64-76,
// deposit.rs:61-72): a Withdraw request is a 170 KB frame for 3 KB of witness.  Decoding it with a general JSON library
// on the serving thread costs more than the proof costs the GPU (Python: ~4 ms per request against 0.8 ms per proof),
// so the front door is a single-pass parser for exactly this schema: no DOM, byte arrays straight into buffers.
// Host code only; the .hip suffix is for the build.
#include <string.h>

#include <string>
#include <vector>

#include "common.hpp"
#include "field.hip.hpp"

namespace {

struct Cur {
    const uint8_t* p;
    const uint8_t* e;
};

[[noreturn]] void bad(const char* what) { throw shp::Error(SHP_ERR_FORMAT, std::string("tee request: ") + what); }

void ws(Cur& c) {
    while (c.p < c.e && (*c.p == ' ' || *c.p == '\n' || *c.p == '\r' || *c.p == '\t')) c.p++;
}
bool peek(Cur& c, char ch) {
    ws(c);
    return c.p < c.e && *c.p == (uint8_t)ch;
}
void expect(Cur& c, char ch) {
    if (!peek(c, ch)) bad("unexpected character");
    c.p++;
}
std::string str(Cur& c) {
    expect(c, '"');
    std::string s;
    while (c.p < c.e && *c.p != '"') {
        if (*c.p == '\\') {                                  // field names and enum tags never need escapes
            if (c.p + 1 >= c.e) bad("truncated escape");
            const char x = (char)c.p[1];
            if (x == 'u') bad("unicode escapes are not part of this protocol");
            if (x != '"' && x != '\\' && x != '/' && x != 'n' && x != 't' && x != 'r' && x != 'b' && x != 'f') bad("invalid escape");
            s.push_back(x == 'n' ? '\n' : x == 't' ? '\t' : x == 'r' ? '\r' : x == 'b' ? '\b' : x == 'f' ? '\f' : x);
            c.p += 2;
        } else if (*c.p < 0x20) {
            bad("control character in a string");
        } else {
            s.push_back((char)*c.p++);
        }
    }
    if (c.p >= c.e) bad("unterminated string");
    c.p++;
    return s;
}
// [n, n, ...] with 0 <= n <= 255
void bytes_array(Cur& c, std::vector<uint8_t>& out) {
    out.clear();
    expect(c, '[');
    if (peek(c, ']')) { c.p++; return; }
    for (;;) {
        ws(c);
        if (c.p >= c.e || *c.p < '0' || *c.p > '9') bad("byte array: expected a number");
        if (*c.p == '0' && c.p + 1 < c.e && c.p[1] >= '0' && c.p[1] <= '9') bad("byte array: leading zero");
        uint32_t v = 0;
        while (c.p < c.e && *c.p >= '0' && *c.p <= '9') {
            v = v * 10 + (*c.p++ - '0');
            if (v > 255) bad("byte array: value above 255");
        }
        out.push_back((uint8_t)v);
        ws(c);
        if (c.p < c.e && *c.p == ',') { c.p++; continue; }
        expect(c, ']');
        return;
    }
}
// JSON number grammar (RFC 8259): -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
void number(Cur& c) {
    auto digits = [&] {
        if (c.p >= c.e || *c.p < '0' || *c.p > '9') bad("malformed number");
        while (c.p < c.e && *c.p >= '0' && *c.p <= '9') c.p++;
    };
    if (c.p < c.e && *c.p == '-') c.p++;
    if (c.p < c.e && *c.p == '0') c.p++;
    else digits();
    if (c.p < c.e && *c.p == '.') { c.p++; digits(); }
    if (c.p < c.e && (*c.p == 'e' || *c.p == 'E')) {
        c.p++;
        if (c.p < c.e && (*c.p == '+' || *c.p == '-')) c.p++;
        digits();
    }
}
// Any JSON value of an ignored field.  Iterative: the open brackets are kept on an explicit stack of at most
// MAX_DEPTH entries (serde_json refuses deeper documents at 128), so that no frame can exhaust the thread's stack.
constexpr size_t MAX_DEPTH = 128;
void skip(Cur& c) {
    char open[MAX_DEPTH];
    size_t depth = 0;
    for (;;) {
        // one value starts here
        ws(c);
        if (c.p >= c.e) bad("truncated value");
        bool closed_one = true;                                  // a scalar, or an empty container, is complete
        if (*c.p == '"') {
            (void)str(c);
        } else if (*c.p == '[' || *c.p == '{') {
            if (depth == MAX_DEPTH) bad("nesting too deep");
            const char o = (char)*c.p++;
            if (peek(c, o == '[' ? ']' : '}')) {
                c.p++;
            } else {
                open[depth++] = o;
                if (o == '{') { (void)str(c); expect(c, ':'); }
                closed_one = false;                              // its first member's value comes next
            }
        } else if (*c.p == '-' || (*c.p >= '0' && *c.p <= '9')) {
            number(c);
        } else {
            const size_t left = (size_t)(c.e - c.p);
            if (left >= 4 && memcmp(c.p, "true", 4) == 0) c.p += 4;
            else if (left >= 4 && memcmp(c.p, "null", 4) == 0) c.p += 4;
            else if (left >= 5 && memcmp(c.p, "false", 5) == 0) c.p += 5;
            else bad("unexpected token");
        }
        if (!closed_one) continue;
        // a value is complete: either another member follows in the innermost open container, or it closes
        for (;;) {
            if (depth == 0) return;
            ws(c);
            if (c.p < c.e && *c.p == ',') {
                c.p++;
                if (open[depth - 1] == '{') { (void)str(c); expect(c, ':'); }
                break;                                           // next member's value
            }
            expect(c, open[depth - 1] == '[' ? ']' : '}');
            depth--;
        }
    }
}
template <class F>
void object(Cur& c, F on_key) {                               // on_key(key, cur) consumes the value
    expect(c, '{');
    if (peek(c, '}')) { c.p++; return; }
    for (;;) {
        std::string k = str(c);
        expect(c, ':');
        on_key(k, c);
        ws(c);
        if (c.p < c.e && *c.p == ',') { c.p++; continue; }
        expect(c, '}');
        return;
    }
}
void end_of_input(Cur& c) {
    ws(c);
    if (c.p != c.e) bad("trailing bytes");
}

// argument order of WithdrawCircuit::prove / DepositCircuit::prove (crates/shielder_bindings/src/circuits/
// withdraw.rs:88-114, deposit.rs:88-114) = field order of the TEE's *ProveInputsBytes
const char* const WITHDRAW_FIELDS[9] = {"id", "nullifier_old", "account_balance_old", "token_address", "path", "value",
                                        "nullifier_new", "commitment", "mac_salt"};
const char* const DEPOSIT_FIELDS[9] = {"id", "nullifier_old", "account_balance_old", "token_address", "path", "value",
                                       "commitment", "nullifier_new", "mac_salt"};
// *PubInputsBytes field order (tee/.../circuits/withdraw.rs:12-23, deposit.rs:13-22)
const char* const WITHDRAW_PUB[8] = {"merkle_root", "h_nullifier_old", "h_note_new", "withdrawal_value", "token_address",
                                     "commitment", "mac_salt", "mac_commitment"};
const char* const DEPOSIT_PUB[8] = {"merkle_root", "h_nullifier_old", "h_note_new", "value", "commitment", "token_address",
                                    "mac_salt", "mac_commitment"};
// NewAccountProveInputsBytes / NewAccountPubInputsBytes (tee/.../circuits/new_account.rs:86-97, 13-28): no Merkle path
const char* const NEW_ACCOUNT_FIELDS[9] = {"id", "nullifier", "initial_deposit", "commitment", "token_address", "encryption_salt",
                                           "mac_salt", "anonymity_revoker_public_key_x", "anonymity_revoker_public_key_y"};
const char* const NEW_ACCOUNT_PUB[13] = {"hashed_note", "prenullifier", "initial_deposit", "commitment", "token_address",
                                         "anonymity_revoker_public_key_x", "anonymity_revoker_public_key_y",
                                         "sym_key_encryption_1_x", "sym_key_encryption_1_y", "sym_key_encryption_2_x",
                                         "sym_key_encryption_2_y", "mac_salt", "mac_commitment"};

struct CircuitShape {
    const char* const* fields;
    int n_fields, path_index;          // path_index < 0: no path argument
    const char* const* pub;
    int n_pub;
};
const CircuitShape& shape_of(int32_t circuit) {
    static const CircuitShape W{WITHDRAW_FIELDS, 9, 4, WITHDRAW_PUB, 8}, D{DEPOSIT_FIELDS, 9, 4, DEPOSIT_PUB, 8},
        N{NEW_ACCOUNT_FIELDS, 9, -1, NEW_ACCOUNT_PUB, 13};
    return circuit == SHP_TEE_WITHDRAW ? W : circuit == SHP_TEE_DEPOSIT ? D : N;
}

void b64(const uint8_t* d, size_t n, std::string& out) {
    static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        uint32_t v = (d[i] << 16) | (d[i + 1] << 8) | d[i + 2];
        out.push_back(T[v >> 18]); out.push_back(T[(v >> 12) & 63]); out.push_back(T[(v >> 6) & 63]); out.push_back(T[v & 63]);
    }
    if (i + 1 == n) {
        uint32_t v = d[i] << 16;
        out.push_back(T[v >> 18]); out.push_back(T[(v >> 12) & 63]); out += "==";
    } else if (i + 2 == n) {
        uint32_t v = (d[i] << 16) | (d[i + 1] << 8);
        out.push_back(T[v >> 18]); out.push_back(T[(v >> 12) & 63]); out.push_back(T[(v >> 6) & 63]); out.push_back('=');
    }
}


// standard base64 with padding, as the reference's base64_serialization writes it (general_purpose::STANDARD)
void b64_decode(const std::string& in, std::vector<uint8_t>& out) {
    auto val = [](char ch) -> int {
        if (ch >= 'A' && ch <= 'Z') return ch - 'A';
        if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
        if (ch >= '0' && ch <= '9') return ch - '0' + 52;
        if (ch == '+') return 62;
        if (ch == '/') return 63;
        return -1;
    };
    out.clear();
    if (in.size() % 4) bad("base64: length is not a multiple of four");
    out.reserve(in.size() / 4 * 3);
    for (size_t i = 0; i < in.size(); i += 4) {
        const bool last = i + 4 == in.size();
        const int pad = last ? (in[i + 3] == '=' ? (in[i + 2] == '=' ? 2 : 1) : 0) : 0;
        int v[4] = {val(in[i]), val(in[i + 1]), pad >= 2 ? 0 : val(in[i + 2]), pad >= 1 ? 0 : val(in[i + 3])};
        if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[3] < 0) bad("base64: invalid character");
        const uint32_t w = (uint32_t)v[0] << 18 | (uint32_t)v[1] << 12 | (uint32_t)v[2] << 6 | (uint32_t)v[3];
        if ((pad == 2 && (w & 0xFFFF)) || (pad == 1 && (w & 0xFF))) bad("base64: non-zero trailing bits");
        out.push_back((uint8_t)(w >> 16));
        if (pad < 2) out.push_back((uint8_t)(w >> 8));
        if (pad < 1) out.push_back((uint8_t)w);
    }
}

// Payload { circuit_type, user_public_key, circuit_inputs } (protocol.rs:14-20) -> the circuit's prove() arguments
void decode_payload(const std::vector<uint8_t>& payload, uint32_t height, int32_t* circuit, uint8_t* witness_le32,
                    size_t witness_cap, size_t* n_inputs) {
    std::vector<uint8_t> inner, field;
    auto once = [](bool& flag) {
        if (flag) bad("duplicate field");
        flag = true;
    };
    std::string ctype;
    bool have_inputs = false, have_key = false, have_type = false;
    Cur pc{payload.data(), payload.data() + payload.size()};
    object(pc, [&](const std::string& k, Cur& cc) {
        if (k == "circuit_type") { once(have_type); ctype = str(cc); }
        else if (k == "user_public_key") { once(have_key); bytes_array(cc, field); }
        else if (k == "circuit_inputs") { once(have_inputs); bytes_array(cc, inner); }
        else skip(cc);
    });
    end_of_input(pc);
    if (!have_inputs || !have_key) bad("payload: missing fields");
    if (ctype == "Withdraw") *circuit = SHP_TEE_WITHDRAW;
    else if (ctype == "Deposit") *circuit = SHP_TEE_DEPOSIT;
    else if (ctype == "NewAccount") *circuit = SHP_TEE_NEW_ACCOUNT;
    else bad("payload: bad circuit_type");
    const CircuitShape& sh = shape_of(*circuit);
    // witness layout: the circuit's prove() arguments in order, 32 bytes each, the path height x 7 x 32
    const size_t path_bytes = (size_t)32 * 7 * height;
    size_t offset[16], total = 0;
    for (int f = 0; f < sh.n_fields; f++) { offset[f] = total; total += f == sh.path_index ? path_bytes : 32; }
    SHP_REQUIRE(witness_le32 && witness_cap >= total, SHP_ERR_INVALID_ARG, "tee_decode_request: witness buffer too small");
    bool seen[16] = {false};
    Cur ic{inner.data(), inner.data() + inner.size()};
    object(ic, [&](const std::string& k, Cur& cc) {
        int f = -1;
        for (int i = 0; i < sh.n_fields; i++)
            if (k == sh.fields[i]) f = i;
        if (f < 0) { skip(cc); return; }
        if (seen[f]) bad("duplicate field");
        bytes_array(cc, field);
        const size_t want = f == sh.path_index ? path_bytes : 32;
        if (field.size() != want) bad("circuit_inputs: field of the wrong length");
        for (size_t o = 0; o < want; o += 32) {          // vec_to_f panics on a non-canonical element (utils.rs:33)
            shp::Fr raw;
            memcpy(raw.l, field.data() + o, 32);
            if (raw.geq_p()) bad("circuit_inputs: not a canonical field element");
        }
        memcpy(witness_le32 + offset[f], field.data(), want);
        seen[f] = true;
    });
    end_of_input(ic);
    for (int f = 0; f < sh.n_fields; f++)
        if (!seen[f]) bad("circuit_inputs: missing field");
    *n_inputs = total / 32;
}

}  // namespace

extern "C" {

int32_t shp_tee_decode_request(const uint8_t* json, size_t len, uint32_t height, int32_t* kind, int32_t* circuit,
                               uint8_t* witness_le32, size_t witness_cap, size_t* n_inputs) {
    return shp::guarded([&] {
        SHP_REQUIRE(json && kind && circuit && n_inputs, SHP_ERR_INVALID_ARG, "tee_decode_request: null argument");
        *kind = -1; *circuit = -1; *n_inputs = 0;
        Cur c{json, json + len};
        ws(c);
        if (c.p < c.e && *c.p == '"') {                       // unit variants are bare strings (serde external tagging)
            const std::string v = str(c);
            end_of_input(c);
            if (v == "Ping") *kind = SHP_TEE_PING;
            else if (v == "TeePublicKey") *kind = SHP_TEE_PUBLIC_KEY;
            else bad("unknown request");
            return;
        }
        // serde rejects a field that appears twice ("duplicate field") and an externally tagged enum with more
        // than one key; ignored (unknown) fields may repeat
        std::vector<uint8_t> payload;
        bool have_payload = false, have_variant = false;
        auto once = [](bool& flag) {
            if (flag) bad("duplicate field");
            flag = true;
        };
        object(c, [&](const std::string& k, Cur& cc) {
            if (k != "GenerateProof") bad("unknown request");
            once(have_variant);
            object(cc, [&](const std::string& k2, Cur& c2) {
                if (k2 == "payload") { once(have_payload); bytes_array(c2, payload); }
                else skip(c2);
            });
        });
        end_of_input(c);
        if (!have_payload) bad("GenerateProof without payload");
        *kind = SHP_TEE_GENERATE_PROOF;
        decode_payload(payload, height, circuit, witness_le32, witness_cap, n_inputs);
    });
}

int32_t shp_tee_encode_proof_response(int32_t circuit, const uint8_t* proof, size_t proof_len, const uint8_t* instances_le32,
                                      size_t n_instances, uint8_t* out, size_t cap, size_t* out_len) {
    return shp::guarded([&] {
        SHP_REQUIRE(proof && instances_le32 && out_len, SHP_ERR_INVALID_ARG, "tee_encode_proof_response: null argument");
        SHP_REQUIRE(circuit >= SHP_TEE_NEW_ACCOUNT && circuit <= SHP_TEE_WITHDRAW && (int)n_instances == shape_of(circuit).n_pub,
                    SHP_ERR_INVALID_ARG, "tee_encode_proof_response: unknown circuit or wrong instance count");
        const char* const* names = shape_of(circuit).pub;
        std::string pub = "{";                               // serde_json::to_vec(&*PubInputsBytes) (server.rs:192)
        for (size_t i = 0; i < n_instances; i++) {
            if (i) pub.push_back(',');
            pub.push_back('"');
            pub += names[i];
            pub += "\":[";
            for (int b = 0; b < 32; b++) {
                if (b) pub.push_back(',');
                pub += std::to_string((unsigned)instances_le32[32 * i + b]);
            }
            pub.push_back(']');
        }
        pub.push_back('}');
        std::string body = "{\"EncryptedProof\":{\"proof\":\"";
        b64(proof, proof_len, body);
        body += "\",\"pub_inputs\":\"";
        b64((const uint8_t*)pub.data(), pub.size(), body);
        body += "\"}}";
        *out_len = body.size();
        SHP_REQUIRE(out && cap >= body.size(), SHP_ERR_INVALID_ARG, "tee_encode_proof_response: buffer too small");
        memcpy(out, body.data(), body.size());
    });
}

int32_t shp_tee_decode_payload(const uint8_t* json, size_t len, uint32_t height, int32_t* circuit, uint8_t* witness_le32,
                               size_t witness_cap, size_t* n_inputs) {
    return shp::guarded([&] {
        SHP_REQUIRE(json && circuit && n_inputs, SHP_ERR_INVALID_ARG, "tee_decode_payload: null argument");
        *circuit = -1; *n_inputs = 0;
        const std::vector<uint8_t> payload(json, json + len);
        decode_payload(payload, height, circuit, witness_le32, witness_cap, n_inputs);
    });
}

int32_t shp_http_decode_proof_body(const uint8_t* json, size_t len, uint8_t* payload_out, size_t cap, size_t* payload_len) {
    return shp::guarded([&] {
        SHP_REQUIRE(json && payload_len, SHP_ERR_INVALID_ARG, "http_decode_proof_body: null argument");
        Cur c{json, json + len};
        std::string b64text;
        bool have = false;
        object(c, [&](const std::string& k, Cur& cc) {
            if (k == "payload") {
                if (have) bad("duplicate field");
                have = true;
                b64text = str(cc);
            } else {
                skip(cc);
            }
        });
        end_of_input(c);
        if (!have) bad("missing field `payload`");
        std::vector<uint8_t> raw;
        b64_decode(b64text, raw);
        *payload_len = raw.size();
        if (!payload_out) return;
        SHP_REQUIRE(cap >= raw.size(), SHP_ERR_INVALID_ARG, "http_decode_proof_body: buffer too small");
        memcpy(payload_out, raw.data(), raw.size());
    });
}

}  // extern "C"
