// The serve loop, native: one TCP listener in front of the proving pools of a whole node.
//
// Two fronts of the reference share it, told apart by the first bytes of a connection:
//   * the enclave server's wire protocol (/root/reference/tee/crates/shielder-prover-tee/src/server.rs:69-90): per
//     connection a sequence of length-delimited serde_json requests (tokio_util's LengthDelimitedCodec, a 4-byte
//     big-endian length, tee/crates/shielder-prover-common/src/vsock.rs:69-102), Ping -> Pong, TeePublicKey,
//     GenerateProof -> EncryptedProof; any error ends the connection (server.rs:70-73);
//   * the host-side HTTP front (/root/reference/tee/crates/shielder-prover-server/src/main.rs:47-59): GET /health,
//     GET /public_key, POST /proof {"payload": base64} -> Json(Response); a request that finds no free slot within the
//     spawn timeout or does not finish within the run timeout is answered 504 "Cannot schedule more tasks"
//     (main.rs:42-45, error.rs:30-34), any other failure 500 (error.rs:26-41), a body above maximum_request_size 413
//     (command_line_args.rs:30-32).
// Behind both: any number of proving pools per circuit type -- one per GPU of the node, each on its own device -- and
// every request goes to the pool with the fewest unfinished jobs.  Frames are decoded and answered by the library's
// own codec (wire.hip) and proofs come from the pools (pool.hip), so a request never passes through an interpreter
// and a host written in any language gets the whole serve path from the C ABI.  ECIES and attestation are out of
// scope: payloads travel in the clear and TeePublicKey answers with empty fields.
// One thread per connection.  No thread waits without a bound: idle peers, half-sent frames, a full queue and a
// stuck proof each have a timeout.
// Host code only; the .hip suffix is for the build.
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <string.h>
#include <strings.h>
#include <sys/random.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/shielder_hip.h"
#include "common.hpp"

namespace {
using Clock = std::chrono::steady_clock;
constexpr size_t MAX_FRAME = 8u * 1024 * 1024;          // LengthDelimitedCodec's default max_frame_length
constexpr size_t MAX_WITNESS_INPUTS = 4096;
constexpr size_t MAX_HTTP_HEAD = 16 * 1024;
constexpr uint32_t MAX_POOLS = 64;

int64_t ms_until(Clock::time_point t) {
    return std::chrono::duration_cast<std::chrono::milliseconds>(t - Clock::now()).count();
}

enum class Io { Ok, Closed, Timeout, Error };

// n bytes, all of them before `deadline`
Io read_exact(int fd, uint8_t* p, size_t n, Clock::time_point deadline) {
    while (n) {
        const int64_t left = ms_until(deadline);
        if (left <= 0) return Io::Timeout;
        pollfd pf{fd, POLLIN, 0};
        const int pr = ::poll(&pf, 1, (int)std::min<int64_t>(left, 1000));
        if (pr < 0) {
            if (errno == EINTR) continue;
            return Io::Error;
        }
        if (pr == 0) continue;
        const ssize_t r = ::recv(fd, p, n, 0);
        if (r == 0) return Io::Closed;
        if (r < 0) {
            if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue;
            return Io::Error;
        }
        p += r;
        n -= (size_t)r;
    }
    return Io::Ok;
}
bool write_all(int fd, const uint8_t* p, size_t n) {    // bounded by the socket's SO_SNDTIMEO
    while (n) {
        const ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}
bool write_frame(int fd, const uint8_t* body, size_t n) {
    std::vector<uint8_t> f(4 + n);
    f[0] = (uint8_t)(n >> 24); f[1] = (uint8_t)(n >> 16); f[2] = (uint8_t)(n >> 8); f[3] = (uint8_t)n;
    memcpy(f.data() + 4, body, n);
    return write_all(fd, f.data(), f.size());
}
bool write_http(int fd, int status, const char* reason, const char* content_type, const std::string& body, bool keep_alive) {
    std::string h = "HTTP/1.1 " + std::to_string(status) + " " + reason + "\r\ncontent-type: " + content_type +
                    "\r\ncontent-length: " + std::to_string(body.size()) +
                    "\r\naccess-control-allow-origin: *\r\nvary: origin, access-control-request-method, access-control-request-headers"
                    "\r\nconnection: " + (keep_alive ? "keep-alive" : "close") + "\r\n\r\n";
    h += body;
    return write_all(fd, (const uint8_t*)h.data(), h.size());
}

const char PONG[] = "\"Pong\"";
const char TEE_KEY[] = "{\"TeePublicKey\":{\"public_key\":\"\",\"attestation_document\":\"\"}}";
}  // namespace

struct shp_server {
    int listen_fd = -1;
    uint16_t port = 0;
    uint32_t height = 0;
    shp_server_options opt{};
    std::vector<shp_pool*> pools[3];                    // by CircuitType (SHP_TEE_NEW_ACCOUNT / _DEPOSIT / _WITHDRAW)
    std::atomic<uint32_t> rr{0};                        // tie-break between equally loaded pools
    std::thread acceptor;
    struct Conn {
        std::thread th;
        int fd = -1;                                    // -1 once the connection's thread has closed it
        std::atomic<bool> done{false};
    };
    std::mutex mu;                                      // conns and their fds
    std::vector<std::unique_ptr<Conn>> conns;
    std::atomic<bool> stopping{false};
    std::atomic<uint64_t> served{0}, errors{0}, connections{0};
    std::atomic<uint64_t> spawn_timeouts{0}, run_timeouts{0}, recv_timeouts{0};

    enum class Outcome { Ok, SpawnTimeout, RunTimeout, Failed };

    // One proof through the least-loaded pool of the circuit type.  The reference's bounded task pool makes a request
    // wait for a slot for at most the spawn timeout and run for at most the run timeout (main.rs:42-45).
    Outcome prove(int32_t circuit, const uint8_t* wit, size_t n_inputs, std::vector<uint8_t>& proof, size_t& plen,
                  std::vector<uint8_t>& inst, uint32_t& n_inst, std::string& why) {
        const std::vector<shp_pool*>* cand = circuit >= 0 && circuit < 3 ? &pools[circuit] : nullptr;
        if (!cand || cand->empty()) { why = "no proving key loaded for this circuit type"; return Outcome::Failed; }
        uint32_t want_inputs = 0, proof_len = 0;
        if (shp_pool_info((*cand)[0], &want_inputs, &n_inst, &proof_len) != SHP_OK || n_inputs != want_inputs) {
            why = "witness of another shape than the circuit's";
            return Outcome::Failed;
        }
        uint8_t seed[32];                               // the reference draws from rand::thread_rng() (withdraw.rs:112)
        if (getrandom(seed, sizeof seed, 0) != (ssize_t)sizeof seed) { why = "getrandom failed"; return Outcome::Failed; }
        const auto spawn_deadline = Clock::now() + std::chrono::milliseconds(opt.spawn_timeout_ms);
        shp_pool* pool = nullptr;
        uint64_t ticket = 0;
        const uint32_t start = rr.fetch_add(1);
        for (;;) {
            // pools in the order of their unfinished jobs (ties: round robin); the first with queue room takes the job
            std::vector<std::pair<uint64_t, shp_pool*>> order;
            for (size_t i = 0; i < cand->size(); i++) {
                shp_pool* q = (*cand)[(start + i) % cand->size()];
                uint64_t load = 0;
                (void)shp_pool_outstanding(q, &load);
                order.emplace_back(load, q);
            }
            std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            int32_t rc = SHP_ERR_BUSY;
            for (auto& o : order) {
                rc = shp_pool_submit(o.second, wit, n_inputs, seed, &ticket);
                if (rc != SHP_ERR_BUSY) { pool = o.second; break; }
            }
            if (rc == SHP_OK) break;
            if (rc != SHP_ERR_BUSY) { why = shp_last_error(); return Outcome::Failed; }
            if (stopping.load()) { why = "server is stopping"; return Outcome::Failed; }
            if (ms_until(spawn_deadline) <= 0) {
                spawn_timeouts++;
                why = "every proving queue stayed full for the spawn timeout";
                return Outcome::SpawnTimeout;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
        proof.resize(proof_len);
        inst.resize((size_t)n_inst * 32);
        plen = 0;
        const int32_t rc = shp_pool_wait_for(pool, ticket, opt.run_timeout_ms, proof.data(), proof.size(), &plen, inst.data(), inst.size());
        if (rc == SHP_ERR_TIMEOUT) {
            run_timeouts++;
            why = "the proof did not finish within the run timeout";
            return Outcome::RunTimeout;
        }
        if (rc != SHP_OK) { why = shp_last_error(); return Outcome::Failed; }   // e.g. a witness that does not satisfy the lookup
        return Outcome::Ok;
    }

    // ---- the enclave protocol: requests in sequence until the peer leaves or one fails (server.rs:75-90)
    void serve_frames(int fd, uint8_t first[4]) {
        std::vector<uint8_t> body, wit(MAX_WITNESS_INPUTS * 32), proof, inst, out;
        uint8_t hdr[4];
        memcpy(hdr, first, 4);
        for (bool have_hdr = true;; have_hdr = false) {
            if (!have_hdr) {
                // between requests the peer may stay idle for the idle timeout; once a frame has begun, all of it
                // must arrive within the receive timeout
                const Io r1 = read_exact(fd, hdr, 1, Clock::now() + std::chrono::milliseconds(opt.idle_timeout_ms));
                if (r1 != Io::Ok) break;                // an idle peer that is dropped is no error
                const Io r3 = read_exact(fd, hdr + 1, 3, Clock::now() + std::chrono::milliseconds(opt.recv_timeout_ms));
                if (r3 != Io::Ok) { if (r3 == Io::Timeout) { recv_timeouts++; errors++; } break; }
            }
            const auto frame_deadline = Clock::now() + std::chrono::milliseconds(opt.recv_timeout_ms);
            const size_t n = ((size_t)hdr[0] << 24) | ((size_t)hdr[1] << 16) | ((size_t)hdr[2] << 8) | hdr[3];
            if (n > MAX_FRAME) { errors++; break; }
            body.resize(n);
            const Io rb = read_exact(fd, body.data(), n, frame_deadline);
            if (rb != Io::Ok) { if (rb == Io::Timeout) { recv_timeouts++; errors++; } break; }
            int32_t kind = -1, circuit = -1;
            size_t n_inputs = 0;
            if (shp_tee_decode_request(body.data(), n, height, &kind, &circuit, wit.data(), wit.size(), &n_inputs) != SHP_OK) {
                errors++;
                break;
            }
            if (kind == SHP_TEE_PING) {
                if (!write_frame(fd, (const uint8_t*)PONG, sizeof PONG - 1)) break;
                continue;
            }
            if (kind == SHP_TEE_PUBLIC_KEY) {
                if (!write_frame(fd, (const uint8_t*)TEE_KEY, sizeof TEE_KEY - 1)) break;
                continue;
            }
            size_t plen = 0;
            uint32_t n_inst = 0;
            std::string why;
            if (prove(circuit, wit.data(), n_inputs, proof, plen, inst, n_inst, why) != Outcome::Ok) { errors++; break; }
            // base64 of the proof and of the public-input JSON (at most four characters per byte of each instance)
            size_t out_len = 0;
            out.resize(2 * plen + (size_t)n_inst * 400 + 256);
            if (shp_tee_encode_proof_response(circuit, proof.data(), plen, inst.data(), n_inst, out.data(), out.size(), &out_len) !=
                SHP_OK) {
                errors++;
                break;
            }
            if (!write_frame(fd, out.data(), out_len)) break;
            served++;
        }
    }

    // ---- the HTTP front (prover-server main.rs:47-59): requests in sequence on a keep-alive connection
    void serve_http(int fd, uint8_t first[4]) {
        std::string buf((const char*)first, 4);
        std::vector<uint8_t> wit(MAX_WITNESS_INPUTS * 32), payload, proof, inst, out;
        for (bool first_request = true;; first_request = false) {
            // head: up to the blank line
            size_t head_end;
            auto deadline = Clock::now() + std::chrono::milliseconds(first_request || !buf.empty() ? opt.recv_timeout_ms : opt.idle_timeout_ms);
            bool begun = !buf.empty();
            while ((head_end = buf.find("\r\n\r\n")) == std::string::npos) {
                if (buf.size() > MAX_HTTP_HEAD) { (void)write_http(fd, 431, "Request Header Fields Too Large", "text/plain", "", false); errors++; return; }
                uint8_t ch[2048];
                pollfd pf{fd, POLLIN, 0};
                const int64_t left = ms_until(deadline);
                if (left <= 0) { if (begun) { recv_timeouts++; errors++; } return; }
                const int pr = ::poll(&pf, 1, (int)std::min<int64_t>(left, 1000));
                if (pr < 0 && errno != EINTR) return;
                if (pr <= 0) continue;
                const ssize_t r = ::recv(fd, ch, sizeof ch, 0);
                if (r == 0) return;
                if (r < 0) { if (errno == EINTR || errno == EAGAIN) continue; return; }
                if (!begun) { begun = true; deadline = Clock::now() + std::chrono::milliseconds(opt.recv_timeout_ms); }
                buf.append((const char*)ch, (size_t)r);
            }
            const std::string head = buf.substr(0, head_end);
            buf.erase(0, head_end + 4);
            // request line and the three headers that matter
            const size_t eol = head.find("\r\n");
            const std::string line = head.substr(0, eol);
            const size_t sp1 = line.find(' '), sp2 = line.rfind(' ');
            if (sp1 == std::string::npos || sp2 == sp1) { (void)write_http(fd, 400, "Bad Request", "text/plain", "", false); errors++; return; }
            const std::string method = line.substr(0, sp1);
            std::string path = line.substr(sp1 + 1, sp2 - sp1 - 1);
            if (path.find('?') != std::string::npos) path.erase(path.find('?'));
            bool keep = line.substr(sp2 + 1) == "HTTP/1.1", json_type = false, has_len = false;
            size_t content_len = 0;
            for (size_t p = eol == std::string::npos ? head.size() : eol + 2; p < head.size();) {
                size_t e = head.find("\r\n", p);
                if (e == std::string::npos) e = head.size();
                const std::string h = head.substr(p, e - p);
                p = e + 2;
                const size_t colon = h.find(':');
                if (colon == std::string::npos) continue;
                std::string name = h.substr(0, colon), val = h.substr(colon + 1);
                while (!val.empty() && (val[0] == ' ' || val[0] == '\t')) val.erase(0, 1);
                while (!val.empty() && (val.back() == ' ' || val.back() == '\t')) val.pop_back();
                if (strcasecmp(name.c_str(), "content-length") == 0) {
                    if (val.empty() || val.size() > 12 || val.find_first_not_of("0123456789") != std::string::npos) {
                        (void)write_http(fd, 400, "Bad Request", "text/plain", "", false); errors++; return;
                    }
                    content_len = (size_t)strtoull(val.c_str(), nullptr, 10);
                    has_len = true;
                } else if (strcasecmp(name.c_str(), "content-type") == 0) {
                    json_type = strncasecmp(val.c_str(), "application/json", 16) == 0;
                } else if (strcasecmp(name.c_str(), "connection") == 0) {
                    if (strcasecmp(val.c_str(), "close") == 0) keep = false;
                    else if (strcasecmp(val.c_str(), "keep-alive") == 0) keep = true;
                } else if (strcasecmp(name.c_str(), "transfer-encoding") == 0) {
                    (void)write_http(fd, 501, "Not Implemented", "text/plain", "chunked bodies are not supported", false); errors++; return;
                }
            }
            if (has_len && content_len > opt.max_http_body) {     // axum's DefaultBodyLimit (main.rs:57)
                (void)write_http(fd, 413, "Payload Too Large", "text/plain", "length limit exceeded", false);
                errors++;
                return;
            }
            if (buf.size() < content_len) {
                const size_t have = buf.size();
                buf.resize(content_len);
                const Io rb = read_exact(fd, (uint8_t*)&buf[have], content_len - have, Clock::now() + std::chrono::milliseconds(opt.recv_timeout_ms));
                if (rb != Io::Ok) { if (rb == Io::Timeout) { recv_timeouts++; errors++; } return; }
            }
            const std::string body = buf.substr(0, content_len);
            buf.erase(0, content_len);

            bool ok = true;
            if (method == "OPTIONS") {                   // CorsLayer::permissive() answers the preflight (main.rs:58)
                std::string h = "HTTP/1.1 200 OK\r\naccess-control-allow-origin: *\r\naccess-control-allow-methods: *\r\n"
                                "access-control-allow-headers: *\r\ncontent-length: 0\r\nconnection: ";
                h += keep ? "keep-alive\r\n\r\n" : "close\r\n\r\n";
                ok = write_all(fd, (const uint8_t*)h.data(), h.size());
            } else if (path == "/health" || path == "/public_key") {
                if (method != "GET") ok = write_http(fd, 405, "Method Not Allowed", "text/plain", "", keep);
                else ok = write_http(fd, 200, "OK", "application/json", path == "/health" ? PONG : TEE_KEY, keep);
            } else if (path == "/proof") {
                if (method != "POST") {
                    ok = write_http(fd, 405, "Method Not Allowed", "text/plain", "", keep);
                } else if (!json_type) {                 // axum's Json extractor: MissingJsonContentType
                    ok = write_http(fd, 415, "Unsupported Media Type", "text/plain", "Expected request with `Content-Type: application/json`", keep);
                } else {
                    ok = http_proof(fd, body, keep, wit, payload, proof, inst, out);
                }
            } else {
                ok = write_http(fd, 404, "Not Found", "text/plain", "", keep);
            }
            if (!ok || !keep) return;
        }
    }

    bool http_proof(int fd, const std::string& body, bool keep, std::vector<uint8_t>& wit, std::vector<uint8_t>& payload,
                    std::vector<uint8_t>& proof, std::vector<uint8_t>& inst, std::vector<uint8_t>& out) {
        // GenerateProofPayload { payload: base64 } (handlers/mod.rs:35-41)
        size_t pay_len = 0;
        payload.resize(body.size());
        if (shp_http_decode_proof_body((const uint8_t*)body.data(), body.size(), payload.data(), payload.size(), &pay_len) != SHP_OK) {
            errors++;                                    // JsonSyntaxError / JsonDataError of the extractor
            return write_http(fd, 422, "Unprocessable Entity", "text/plain",
                              std::string("Failed to deserialize the JSON body into the target type: ") + shp_last_error(), keep);
        }
        int32_t circuit = -1;
        size_t n_inputs = 0;
        if (shp_tee_decode_payload(payload.data(), pay_len, height, &circuit, wit.data(), wit.size(), &n_inputs) != SHP_OK) {
            errors++;                                    // the enclave refuses the payload: ProvingServerError -> 500
            return write_http(fd, 500, "Internal Server Error", "text/plain", std::string("TEE Proving Server error: ") + shp_last_error(), keep);
        }
        size_t plen = 0;
        uint32_t n_inst = 0;
        std::string why;
        const Outcome oc = prove(circuit, wit.data(), n_inputs, proof, plen, inst, n_inst, why);
        if (oc == Outcome::SpawnTimeout || oc == Outcome::RunTimeout) {
            errors++;                                    // ShielderProverServerError::TaskPool -> GATEWAY_TIMEOUT (error.rs:30-34)
            return write_http(fd, 504, "Gateway Timeout", "text/plain", "Cannot schedule more tasks: " + why, keep);
        }
        if (oc != Outcome::Ok) {
            errors++;
            return write_http(fd, 500, "Internal Server Error", "text/plain", "TEE Proving Server error: " + why, keep);
        }
        size_t out_len = 0;
        out.resize(2 * plen + (size_t)n_inst * 400 + 256);
        if (shp_tee_encode_proof_response(circuit, proof.data(), plen, inst.data(), n_inst, out.data(), out.size(), &out_len) != SHP_OK) {
            errors++;
            return write_http(fd, 500, "Internal Server Error", "text/plain", std::string("TEE Proving Server error: ") + shp_last_error(), keep);
        }
        served++;
        return write_http(fd, 200, "OK", "application/json", std::string((const char*)out.data(), out_len), keep);
    }

    void serve_connection(Conn* self) {
        const int fd = self->fd;
        try {
            // the first four bytes decide the front: a frame length is at most 8 MiB, so its first byte is zero,
            // while an HTTP request line starts with a method name
            uint8_t first[4];
            const Io r1 = read_exact(fd, first, 1, Clock::now() + std::chrono::milliseconds(opt.idle_timeout_ms));
            if (r1 == Io::Ok) {
                const Io r3 = read_exact(fd, first + 1, 3, Clock::now() + std::chrono::milliseconds(opt.recv_timeout_ms));
                if (r3 == Io::Timeout) { recv_timeouts++; errors++; }
                if (r3 == Io::Ok) {
                    if (first[0] >= 'A' && first[0] <= 'Z') serve_http(fd, first);
                    else serve_frames(fd, first);
                }
            }
        } catch (...) {                                  // e.g. std::bad_alloc for a frame buffer: this connection only
            errors++;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ::close(fd);
            self->fd = -1;
        }
        self->done.store(true);
    }

    void accept_loop() {
        for (;;) {
            const int fd = ::accept(listen_fd, nullptr, nullptr);
            if (stopping.load()) {
                if (fd >= 0) ::close(fd);
                return;
            }
            if (fd < 0) {
                // only shp_server_stop ends the listener.  A peer that went away, or a process or system that has run
                // out of descriptors or buffers for a moment, must not: wait briefly where waiting helps, and go on
                const int e = errno;
                if (e == EMFILE || e == ENFILE || e == ENOBUFS || e == ENOMEM) {
                    errors++;
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));
                } else if (e != EINTR && e != ECONNABORTED && e != EAGAIN && e != EPROTO && e != EPERM) {
                    errors++;
                    std::this_thread::sleep_for(std::chrono::milliseconds(5));
                }
                continue;
            }
            int one = 1;
            (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            timeval tv;
            tv.tv_sec = opt.recv_timeout_ms / 1000;
            tv.tv_usec = (opt.recv_timeout_ms % 1000) * 1000;
            (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);     // a peer that stops reading cannot hold a thread
            connections++;
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < conns.size();) {     // connections that have ended: join and forget
                if (conns[i]->done.load()) {
                    conns[i]->th.join();
                    conns[i] = std::move(conns.back());
                    conns.pop_back();
                } else {
                    i++;
                }
            }
            if (conns.size() >= opt.max_connections) {
                ::close(fd);
                errors++;
                continue;
            }
            try {
                conns.emplace_back(new Conn);
                Conn* c = conns.back().get();
                c->fd = fd;
                c->th = std::thread([this, c] { serve_connection(c); });
            } catch (...) {                              // no memory or no thread for this peer: turn it away
                if (!conns.empty() && !conns.back()->th.joinable() && conns.back()->fd == fd) conns.pop_back();
                ::close(fd);
                errors++;
            }
        }
    }
};

extern "C" {

int32_t shp_server_start_ex(shp_pool* const* pools, const int32_t* circuits, uint32_t n_pools, uint32_t merkle_height,
                            const char* bind_addr, uint16_t port, const shp_server_options* options, shp_server** out) {
    return shp::guarded([&] {
        SHP_REQUIRE(out && (n_pools == 0 || (pools && circuits)) && n_pools <= MAX_POOLS, SHP_ERR_INVALID_ARG, "server_start: bad arguments");
        std::unique_ptr<shp_server> s(new shp_server);
        s->height = merkle_height;
        shp_server_options& o = s->opt;
        if (options) {
            SHP_REQUIRE(options->struct_size >= 8 && options->struct_size <= sizeof o, SHP_ERR_INVALID_ARG,
                        "server_start: options.struct_size must be sizeof(shp_server_options)");
            memcpy(&o, options, options->struct_size);
        }
        // the reference's defaults (command_line_args.rs:23-40): spawn 5 s, run 60 s, bodies up to 100 KiB
        if (!o.spawn_timeout_ms) o.spawn_timeout_ms = 5000;
        if (!o.run_timeout_ms) o.run_timeout_ms = 60000;
        if (!o.recv_timeout_ms) o.recv_timeout_ms = 5000;
        if (!o.idle_timeout_ms) o.idle_timeout_ms = 60000;
        if (!o.max_connections) o.max_connections = 1024;
        if (!o.max_http_body) o.max_http_body = 100 * 1024;
        for (uint32_t i = 0; i < n_pools; i++) {
            SHP_REQUIRE(pools[i] && circuits[i] >= 0 && circuits[i] < 3, SHP_ERR_INVALID_ARG, "server_start: null pool or unknown circuit type");
            auto& v = s->pools[circuits[i]];
            if (!v.empty()) {                            // pools of one circuit type are interchangeable: same key shape
                uint32_t a[3], b[3];
                SHP_REQUIRE(shp_pool_info(v[0], &a[0], &a[1], &a[2]) == SHP_OK && shp_pool_info(pools[i], &b[0], &b[1], &b[2]) == SHP_OK &&
                                memcmp(a, b, sizeof a) == 0,
                            SHP_ERR_INVALID_ARG, "server_start: pools of one circuit type must hold keys of the same shape");
            }
            v.push_back(pools[i]);
        }
        sockaddr_in addr;
        memset(&addr, 0, sizeof addr);
        addr.sin_family = AF_INET;
        addr.sin_port = htons(port);
        SHP_REQUIRE(inet_pton(AF_INET, bind_addr ? bind_addr : "127.0.0.1", &addr.sin_addr) == 1, SHP_ERR_INVALID_ARG,
                    "server_start: bind address is not a dotted IPv4 address");
        const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        SHP_REQUIRE(fd >= 0, SHP_ERR_INTERNAL, std::string("server_start: socket: ") + strerror(errno));
        int one = 1;
        (void)setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (::bind(fd, (const sockaddr*)&addr, sizeof addr) != 0 || ::listen(fd, 256) != 0) {
            const std::string why = strerror(errno);
            ::close(fd);
            throw shp::Error(SHP_ERR_INTERNAL, "server_start: bind / listen: " + why);
        }
        socklen_t len = sizeof addr;
        (void)getsockname(fd, (sockaddr*)&addr, &len);
        s->listen_fd = fd;
        s->port = ntohs(addr.sin_port);
        shp_server* raw = s.release();
        raw->acceptor = std::thread([raw] { raw->accept_loop(); });
        *out = raw;
    });
}

int32_t shp_server_start(shp_pool* const* pools, const int32_t* circuits, uint32_t n_pools, uint32_t merkle_height,
                         const char* bind_addr, uint16_t port, shp_server** out) {
    return shp_server_start_ex(pools, circuits, n_pools, merkle_height, bind_addr, port, nullptr, out);
}

int32_t shp_server_port(const shp_server* s, uint16_t* port) {
    return shp::guarded([&] {
        SHP_REQUIRE(s && port, SHP_ERR_INVALID_ARG, "server_port: null argument");
        *port = s->port;
    });
}

int32_t shp_server_stats(const shp_server* s, uint64_t* served, uint64_t* failed, uint64_t* connections) {
    return shp::guarded([&] {
        SHP_REQUIRE(s, SHP_ERR_INVALID_ARG, "server_stats: null server");
        if (served) *served = s->served.load();
        if (failed) *failed = s->errors.load();
        if (connections) *connections = s->connections.load();
    });
}

int32_t shp_server_timeouts(const shp_server* s, uint64_t* spawn, uint64_t* run, uint64_t* receive) {
    return shp::guarded([&] {
        SHP_REQUIRE(s, SHP_ERR_INVALID_ARG, "server_timeouts: null server");
        if (spawn) *spawn = s->spawn_timeouts.load();
        if (run) *run = s->run_timeouts.load();
        if (receive) *receive = s->recv_timeouts.load();
    });
}

int32_t shp_server_stop(shp_server* s) {
    return shp::guarded([&] {
        if (!s) return;
        s->stopping.store(true);
        ::shutdown(s->listen_fd, SHUT_RDWR);            // wakes accept()
        if (s->acceptor.joinable()) s->acceptor.join();
        ::close(s->listen_fd);
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (auto& c : s->conns)
                if (c->fd >= 0) ::shutdown(c->fd, SHUT_RDWR);     // wakes poll() / recv(); the connection's thread closes its socket
        }
        for (auto& c : s->conns)                        // the acceptor has ended: nobody else touches the list
            if (c->th.joinable()) c->th.join();
        delete s;
    });
}

}  // extern "C"
