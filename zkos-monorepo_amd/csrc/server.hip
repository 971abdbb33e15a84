// The serve loop, native: a TCP listener in front of the proving pools, speaking the reference's TEE wire protocol.
//
// The reference's enclave server accepts connections and answers, per connection, a sequence of length-delimited
// serde_json requests until the peer leaves or a request fails (/root/reference/tee/crates/shielder-prover-tee/src/
// server.rs:69-90: Ping -> Pong, TeePublicKey, GenerateProof -> EncryptedProof; framing = tokio_util's
// LengthDelimitedCodec, a 4-byte big-endian length, tee/crates/shielder-prover-common/src/vsock.rs:69-102; any error ends
// the connection, server.rs:70-73).  This file is that loop over TCP with one thread per connection: frames are decoded
// and answered by the library's own codec (wire.hip) and proofs come from the pools (pool.hip), so a request never
// passes through an interpreter and a host written in any language gets the whole serve path from the C ABI (measured
// with 48 lanes: 2290 requests/s, the pool's own rate; serve.py's asyncio loop in front of the same pool: 2210).  ECIES
// and attestation are out of scope, as in serve.py: payloads travel in the clear and TeePublicKey answers with empty
// fields.
// Host code only; the .hip suffix is for the build.
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <string.h>
#include <sys/random.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <memory>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/shielder_hip.h"
#include "common.hpp"

namespace {
constexpr size_t MAX_FRAME = 8u * 1024 * 1024;          // LengthDelimitedCodec's default max_frame_length
constexpr size_t MAX_WITNESS_INPUTS = 4096;
constexpr size_t MAX_CONNECTIONS = 1024;                // a thread each: further peers are turned away until one leaves

bool read_exact(int fd, uint8_t* p, size_t n) {
    while (n) {
        const ssize_t r = ::recv(fd, p, n, 0);
        if (r == 0) return false;                       // the peer closed
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}
bool write_all(int fd, const uint8_t* p, size_t n) {
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
}  // namespace

struct shp_server {
    int listen_fd = -1;
    uint16_t port = 0;
    uint32_t height = 0;
    shp_pool* pools[3] = {nullptr, nullptr, nullptr};   // by CircuitType (SHP_TEE_NEW_ACCOUNT / _DEPOSIT / _WITHDRAW)
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

    // one connection: requests in sequence until the peer leaves or one fails (server.rs:75-90)
    void serve_connection(Conn* self) {
        const int fd = self->fd;
        std::vector<uint8_t> body, wit(MAX_WITNESS_INPUTS * 32), proof, inst, out;
        for (;;) {
            uint8_t hdr[4];
            if (!read_exact(fd, hdr, 4)) break;
            const size_t n = ((size_t)hdr[0] << 24) | ((size_t)hdr[1] << 16) | ((size_t)hdr[2] << 8) | hdr[3];
            if (n > MAX_FRAME) { errors++; break; }
            body.resize(n);
            if (!read_exact(fd, body.data(), n)) break;
            int32_t kind = -1, circuit = -1;
            size_t n_inputs = 0;
            if (shp_tee_decode_request(body.data(), n, height, &kind, &circuit, wit.data(), wit.size(), &n_inputs) != SHP_OK) {
                errors++;
                break;
            }
            if (kind == SHP_TEE_PING) {
                static const char pong[] = "\"Pong\"";
                if (!write_frame(fd, (const uint8_t*)pong, sizeof pong - 1)) break;
                continue;
            }
            if (kind == SHP_TEE_PUBLIC_KEY) {
                static const char pkr[] = "{\"TeePublicKey\":{\"public_key\":\"\",\"attestation_document\":\"\"}}";
                if (!write_frame(fd, (const uint8_t*)pkr, sizeof pkr - 1)) break;
                continue;
            }
            shp_pool* pool = circuit >= 0 && circuit < 3 ? pools[circuit] : nullptr;
            uint32_t want_inputs = 0, n_inst = 0, proof_len = 0;
            if (!pool || shp_pool_info(pool, &want_inputs, &n_inst, &proof_len) != SHP_OK || n_inputs != want_inputs) {
                errors++;                               // no key for this circuit type, or a witness of another shape
                break;
            }
            uint8_t seed[32];                           // the reference draws from rand::thread_rng() (withdraw.rs:112)
            if (getrandom(seed, sizeof seed, 0) != (ssize_t)sizeof seed) { errors++; break; }
            uint64_t ticket = 0;
            int32_t rc;
            // a full queue is waited out: the reference's bounded task pool makes the request wait for a slot, too
            while ((rc = shp_pool_submit(pool, wit.data(), n_inputs, seed, &ticket)) == SHP_ERR_BUSY && !stopping.load())
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            if (rc != SHP_OK) { errors++; break; }
            proof.resize(proof_len);
            inst.resize((size_t)n_inst * 32);
            size_t plen = 0;
            if (shp_pool_wait(pool, ticket, proof.data(), proof.size(), &plen, inst.data(), inst.size()) != SHP_OK) {
                errors++;                               // e.g. a witness that does not satisfy the lookup
                break;
            }
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
            if (fd < 0) {
                if (errno == EINTR) continue;
                return;                                 // the listener was shut down (shp_server_stop)
            }
            if (stopping.load()) { ::close(fd); return; }
            int one = 1;
            (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
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
            if (conns.size() >= MAX_CONNECTIONS) {
                ::close(fd);
                errors++;
                continue;
            }
            conns.emplace_back(new Conn);
            Conn* c = conns.back().get();
            c->fd = fd;
            c->th = std::thread([this, c] { serve_connection(c); });
        }
    }
};

extern "C" {

int32_t shp_server_start(shp_pool* const* pools, const int32_t* circuits, uint32_t n_pools, uint32_t merkle_height,
                         const char* bind_addr, uint16_t port, shp_server** out) {
    return shp::guarded([&] {
        SHP_REQUIRE(out && (n_pools == 0 || (pools && circuits)) && n_pools <= 3, SHP_ERR_INVALID_ARG, "server_start: bad arguments");
        std::unique_ptr<shp_server> s(new shp_server);
        s->height = merkle_height;
        for (uint32_t i = 0; i < n_pools; i++) {
            SHP_REQUIRE(pools[i] && circuits[i] >= 0 && circuits[i] < 3 && !s->pools[circuits[i]], SHP_ERR_INVALID_ARG,
                        "server_start: a pool per circuit type, each type once");
            s->pools[circuits[i]] = pools[i];
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
                if (c->fd >= 0) ::shutdown(c->fd, SHUT_RDWR);     // wakes recv(); the connection's thread closes its socket
        }
        for (auto& c : s->conns)                        // the acceptor has ended: nobody else touches the list
            if (c->th.joinable()) c->th.join();
        delete s;
    });
}

}  // extern "C"
