// Proving pool: N proving lanes fed from one bounded FIFO queue by N worker threads.
//
// The native counterpart of the reference's "whole node" serve path: one tokio task per vsock
// connection proving on the worker thread (/root/reference/tee/crates/shielder-prover-tee/src/main.rs:13-20,
// server.rs:157-195) behind a bounded task pool (tee/crates/shielder-prover-server/src/main.rs:42-45,
// command_line_args.rs:23-28: cap 100, spawn timeout -> HTTP 504).  Here the proving key stays
// resident in HBM (the TEE re-decodes params.bin / pk.bin on every request, server.rs:170-173), a
// lane = one HIP stream + workspace, and shp_prove_batch is the batch entry point:
// `count` independent proofs sharded over the lanes of this GPU.  No CPU fallback anywhere.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <string.h>
#include <thread>

#include "common.hpp"

namespace {

struct Job {
    uint64_t ticket = 0;
    std::vector<uint8_t> witness;
    uint8_t seed[32];
    size_t n_inputs = 0;
    // a job whose columns the caller synthesised (shp_pool_submit_assigned): [num_advice][rows] Montgomery limbs, the public
    // inputs, optionally the pre-drawn randomness instead of the seed
    bool assigned = false;
    size_t rows = 0;
    std::vector<uint64_t> advice, inst_mont, draws;
    // result
    bool done = false;
    bool abandoned = false;      // its waiter gave up (shp_pool_wait_for timed out): a lane that has not started it skips it
    int32_t status = SHP_OK;
    std::string error;
    std::vector<uint8_t> proof, instances;
};

}  // namespace

struct shp_pool {
    shp_pk* pk = nullptr;
    int device = 0;
    uint32_t proof_len = 0, n_inputs = 0, n_instances = 0;
    uint32_t num_advice = 0, n_rows = 0, usable = 0;
    bool phased = false;
    uint64_t n_draws = 0;
    size_t queue_cap = 0;
    std::vector<shp_prover*> lanes;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Job>> queue;
    std::map<uint64_t, std::shared_ptr<Job>> jobs;       // submitted and not yet collected
    uint64_t next_ticket = 1;
    bool stopping = false;
    std::atomic<uint64_t> proofs_done{0};
    std::atomic<uint64_t> outstanding{0};                // jobs accepted and not finished: queued or on a lane
    shp::MsmCombiner* comb = nullptr;      // SHP_POOL_COMBINE=1: one MSM launch serves every lane waiting at a commitment

    void worker(shp_prover* lane) {
        (void)hipSetDevice(device);
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stopping || !queue.empty(); });
                if (queue.empty()) return;               // stopping and drained
                j = queue.front();
                queue.pop_front();
                if (j->abandoned) {                      // nobody will collect it: the lane goes to the next job
                    j->done = true;
                    outstanding--;
                    continue;
                }
            }
            j->proof.resize(proof_len);
            j->instances.resize((size_t)n_instances * 32);
            size_t len = 0;
            int32_t rc;
            if (j->assigned) {
                rc = shp_prove_assigned_rows(lane, j->advice.data(), num_advice, j->rows, j->inst_mont.data(), n_instances,
                                             j->draws.empty() ? nullptr : j->draws.data(), j->draws.size() / 4, j->seed,
                                             j->proof.data(), j->proof.size(), &len);
                for (uint32_t i = 0; rc == SHP_OK && i < n_instances; i++)     // handed back as every other job's are
                    rc = shp_fr_to_le32(j->inst_mont.data() + 4 * i, j->instances.data() + 32 * i);
                std::vector<uint64_t>().swap(j->advice);                        // 0.3 - 2.6 MB per job: not kept until the wait
            } else {
                rc = shp_prove(lane, j->witness.data(), j->n_inputs, j->seed, j->proof.data(), j->proof.size(), &len,
                               j->instances.data(), j->instances.size());
            }
            std::string err = rc == SHP_OK ? std::string() : std::string(shp_last_error());
            {
                std::lock_guard<std::mutex> lk(mu);
                j->status = rc;
                j->error = err;
                j->proof.resize(rc == SHP_OK ? len : 0);
                j->done = true;
            }
            proofs_done++;
            outstanding--;
            cv_done.notify_all();
        }
    }
};

extern "C" {

int32_t shp_pool_create(shp_pk* pk, uint32_t lanes, uint32_t queue_cap, shp_pool** out) {
    return shp::guarded([&] {
        SHP_REQUIRE(pk && out && lanes >= 1 && lanes <= 64, SHP_ERR_INVALID_ARG, "pool_create: bad arguments");
        std::unique_ptr<shp_pool> p(new shp_pool);
        p->pk = pk;
        // the key's device, not the calling thread's: an FFI thread that never called hipSetDevice sits on device 0
        p->device = shp_pk_device(pk);
        SHP_HIP(hipSetDevice(p->device));
        uint32_t k = 0;
        int32_t rc = shp_pk_info(pk, &k, &p->n_inputs, &p->n_instances, &p->proof_len);
        SHP_REQUIRE(rc == SHP_OK, rc, "pool_create: pk_info failed");
        shp_pk_shape(pk, &p->num_advice, &p->n_rows, &p->usable, &p->phased, &p->n_draws);
        p->queue_cap = queue_cap ? queue_cap : 100;      // the reference's task-pool bound
        // pools above this many lanes schedule them for throughput (SHP_POOL_THROUGHPUT_ABOVE, default 8: measured on the
        // serve path, latency-mode lanes win up to 8 lanes -- 6 lanes 1013 vs 934 proofs/s, 8 lanes 1134 vs 1101 -- draw
        // level at 12 and lose at 16, 1218 vs 1304)
        const uint32_t tp_above = getenv("SHP_POOL_THROUGHPUT_ABOVE") ? (uint32_t)atoi(getenv("SHP_POOL_THROUGHPUT_ABOVE")) : 8;
        for (uint32_t i = 0; i < lanes; i++) {
            shp_prover* pr = nullptr;
            rc = shp_prover_create(pk, &pr);
            if (rc != SHP_OK) {
                for (auto* q : p->lanes) shp_prover_destroy(q);
                throw shp::Error(rc, std::string("pool_create: ") + shp_last_error());
            }
            if (lanes > tp_above) (void)shp_prover_mode(pr, 1);   // many lanes: one host thread each, long MSM chains
            else if (lanes > 1) (void)shp_prover_witness_threads(pr, std::max<uint32_t>(1, 8 / lanes));   // a few lanes share the helper threads' cores
            p->lanes.push_back(pr);
        }
        // opt-in (SHP_POOL_COMBINE=1): on one MI355X the pool is not limited by the size of its MSM launches --
        // 16 lanes lose 19 % behind the combiner's serial launches, 48 lanes only draw level
        const char* ce = getenv("SHP_POOL_COMBINE");
        if (lanes > 1 && ce && ce[0] == '1') {
            try {
                p->comb = shp::msm_combiner_create(p->device);
                for (size_t i = 0; i < p->lanes.size(); i++) shp_prover_attach_combiner(p->lanes[i], p->comb, i == 0);
            } catch (...) {
                for (auto* q : p->lanes) shp_prover_destroy(q);
                shp::msm_combiner_destroy(p->comb);
                throw;
            }
        }
        shp_pool* raw = p.get();
        for (auto* pr : p->lanes) p->workers.emplace_back([raw, pr] { raw->worker(pr); });
        *out = p.release();
    });
}

int32_t shp_pool_destroy(shp_pool* p) {
    return shp::guarded([&] {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->stopping = true;
        }
        p->cv_work.notify_all();
        for (auto& t : p->workers) t.join();
        for (auto* pr : p->lanes) shp_prover_destroy(pr);
        shp::msm_combiner_destroy(p->comb);
        delete p;
    });
}

int32_t shp_pool_lanes(const shp_pool* p, uint32_t* lanes) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && lanes, SHP_ERR_INVALID_ARG, "pool_lanes: null argument");
        *lanes = (uint32_t)p->lanes.size();
    });
}

int32_t shp_pool_lane(shp_pool* p, uint32_t index, shp_prover** lane) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && lane && index < p->lanes.size(), SHP_ERR_INVALID_ARG, "pool_lane: bad argument");
        *lane = p->lanes[index];
    });
}

int32_t shp_pool_submit(shp_pool* p, const uint8_t* witness_le32, size_t n_inputs, const uint8_t seed[32], uint64_t* ticket) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && witness_le32 && seed && ticket, SHP_ERR_INVALID_ARG, "pool_submit: null argument");
        SHP_REQUIRE(n_inputs == p->n_inputs, SHP_ERR_INVALID_ARG, "pool_submit: wrong number of witness inputs");
        auto j = std::make_shared<Job>();
        j->witness.assign(witness_le32, witness_le32 + n_inputs * 32);
        memcpy(j->seed, seed, 32);
        j->n_inputs = n_inputs;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            SHP_REQUIRE(!p->stopping, SHP_ERR_INVALID_ARG, "pool_submit: pool is shutting down");
            SHP_REQUIRE(p->queue.size() < p->queue_cap, SHP_ERR_BUSY, "pool_submit: queue full");
            j->ticket = p->next_ticket++;
            p->queue.push_back(j);
            p->jobs[j->ticket] = j;
            p->outstanding++;
            *ticket = j->ticket;
        }
        p->cv_work.notify_one();
    });
}

int32_t shp_pool_submit_assigned(shp_pool* p, const uint64_t* advice, size_t n_columns, size_t n_rows, size_t assigned_rows,
                                 const uint64_t* instances, size_t n_instances, const uint64_t* randomness, size_t n_draws,
                                 const uint8_t* seed, uint64_t* ticket) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && advice && ticket && (instances || !n_instances), SHP_ERR_INVALID_ARG, "pool_submit_assigned: null argument");
        SHP_REQUIRE(!p->phased, SHP_ERR_INVALID_ARG,
                    "pool_submit_assigned: a circuit with several phases is synthesised phase by phase (shp_prove_assigned_phased)");
        SHP_REQUIRE(n_columns == p->num_advice && n_rows == p->n_rows && n_instances == p->n_instances, SHP_ERR_INVALID_ARG,
                    "pool_submit_assigned: the advice block must be num_advice columns of 2^k rows, with the circuit's public inputs");
        SHP_REQUIRE(assigned_rows <= p->usable, SHP_ERR_INVALID_ARG, "pool_submit_assigned: more assigned rows than usable rows");
        SHP_REQUIRE((randomness != nullptr) != (seed != nullptr), SHP_ERR_INVALID_ARG, "pool_submit_assigned: pass a seed or a randomness buffer");
        SHP_REQUIRE(!randomness || n_draws == p->n_draws, SHP_ERR_INVALID_ARG, "pool_submit_assigned: randomness buffer of the wrong length");
        auto j = std::make_shared<Job>();
        j->assigned = true;
        j->rows = assigned_rows ? assigned_rows : p->usable;
        // the job owns a compact copy (the rows that are read, column after column): the caller's block is free on return
        j->advice.resize((size_t)n_columns * j->rows * 4);
        for (size_t c = 0; c < n_columns; c++)
            memcpy(j->advice.data() + c * j->rows * 4, advice + c * n_rows * 4, j->rows * 32);
        j->inst_mont.assign(instances, instances + n_instances * 4);
        if (randomness) j->draws.assign(randomness, randomness + n_draws * 4);
        else memcpy(j->seed, seed, 32);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            SHP_REQUIRE(!p->stopping, SHP_ERR_INVALID_ARG, "pool_submit_assigned: pool is shutting down");
            SHP_REQUIRE(p->queue.size() < p->queue_cap, SHP_ERR_BUSY, "pool_submit_assigned: queue full");
            j->ticket = p->next_ticket++;
            p->queue.push_back(j);
            p->jobs[j->ticket] = j;
            p->outstanding++;
            *ticket = j->ticket;
        }
        p->cv_work.notify_one();
    });
}

static void pool_wait_impl(shp_pool* p, uint64_t ticket, int64_t timeout_ms, uint8_t* proof_out, size_t proof_cap,
                           size_t* proof_len, uint8_t* instances_le32, size_t instances_cap) {
    SHP_REQUIRE(p && proof_out && proof_len, SHP_ERR_INVALID_ARG, "pool_wait: null argument");
    std::shared_ptr<Job> j;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        auto it = p->jobs.find(ticket);
        SHP_REQUIRE(it != p->jobs.end(), SHP_ERR_INVALID_ARG, "pool_wait: unknown ticket");
        j = it->second;
        if (timeout_ms < 0) {
            p->cv_done.wait(lk, [&] { return j->done; });
        } else if (!p->cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return j->done; })) {
            // the reference's run timeout drops the task's future (tokio_task_pool with_run_timeout,
            // /root/reference/tee/crates/shielder-prover-server/src/main.rs:42-45): the ticket is gone, a proof that
            // is already on a lane still finishes there and is discarded
            j->abandoned = true;
            p->jobs.erase(it);
            throw shp::Error(SHP_ERR_TIMEOUT, "pool_wait_for: the proof did not finish in time; the ticket is abandoned");
        }
        p->jobs.erase(it);
    }
    if (j->status != SHP_OK) throw shp::Error(j->status, j->error);
    SHP_REQUIRE(j->proof.size() <= proof_cap, SHP_ERR_INVALID_ARG, "pool_wait: proof buffer too small");
    memcpy(proof_out, j->proof.data(), j->proof.size());
    *proof_len = j->proof.size();
    if (instances_le32) {
        SHP_REQUIRE(j->instances.size() <= instances_cap, SHP_ERR_INVALID_ARG, "pool_wait: instance buffer too small");
        memcpy(instances_le32, j->instances.data(), j->instances.size());
    }
}

int32_t shp_pool_wait(shp_pool* p, uint64_t ticket, uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                      uint8_t* instances_le32, size_t instances_cap) {
    return shp::guarded([&] { pool_wait_impl(p, ticket, -1, proof_out, proof_cap, proof_len, instances_le32, instances_cap); });
}

int32_t shp_pool_wait_for(shp_pool* p, uint64_t ticket, uint32_t timeout_ms, uint8_t* proof_out, size_t proof_cap,
                          size_t* proof_len, uint8_t* instances_le32, size_t instances_cap) {
    return shp::guarded(
        [&] { pool_wait_impl(p, ticket, (int64_t)timeout_ms, proof_out, proof_cap, proof_len, instances_le32, instances_cap); });
}

int32_t shp_pool_outstanding(const shp_pool* p, uint64_t* jobs) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && jobs, SHP_ERR_INVALID_ARG, "pool_outstanding: null argument");
        *jobs = p->outstanding.load();
    });
}

int32_t shp_prove_batch(shp_pool* p, const uint8_t* witnesses_le32, size_t n_inputs, size_t count, const uint8_t* seeds,
                        uint8_t* proofs_out, size_t proof_stride, size_t* proof_lens, uint8_t* instances_le32,
                        size_t instances_stride) {
    return shp::guarded([&] {
        SHP_REQUIRE(p && witnesses_le32 && seeds && proofs_out && proof_lens, SHP_ERR_INVALID_ARG, "prove_batch: null argument");
        SHP_REQUIRE(n_inputs == p->n_inputs && proof_stride >= p->proof_len, SHP_ERR_INVALID_ARG, "prove_batch: bad sizes");
        SHP_REQUIRE(!instances_le32 || instances_stride >= (size_t)p->n_instances * 32, SHP_ERR_INVALID_ARG,
                    "prove_batch: instance stride too small");
        // jobs bypass the bounded queue's cap: the batch is one request
        std::vector<std::shared_ptr<Job>> js(count);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            SHP_REQUIRE(!p->stopping, SHP_ERR_INVALID_ARG, "prove_batch: pool is shutting down");
            for (size_t i = 0; i < count; i++) {
                auto j = std::make_shared<Job>();
                j->witness.assign(witnesses_le32 + i * n_inputs * 32, witnesses_le32 + (i + 1) * n_inputs * 32);
                memcpy(j->seed, seeds + 32 * i, 32);
                j->n_inputs = n_inputs;
                j->ticket = p->next_ticket++;
                p->queue.push_back(j);
                p->outstanding++;
                js[i] = j;
            }
        }
        p->cv_work.notify_all();
        int32_t first_err = SHP_OK;
        std::string first_msg;
        for (size_t i = 0; i < count; i++) {
            {
                std::unique_lock<std::mutex> lk(p->mu);
                p->cv_done.wait(lk, [&] { return js[i]->done; });
            }
            if (js[i]->status != SHP_OK) {
                if (first_err == SHP_OK) { first_err = js[i]->status; first_msg = "proof " + std::to_string(i) + ": " + js[i]->error; }
                proof_lens[i] = 0;
                continue;
            }
            memcpy(proofs_out + i * proof_stride, js[i]->proof.data(), js[i]->proof.size());
            proof_lens[i] = js[i]->proof.size();
            if (instances_le32) memcpy(instances_le32 + i * instances_stride, js[i]->instances.data(), js[i]->instances.size());
        }
        if (first_err != SHP_OK) throw shp::Error(first_err, first_msg);
    });
}

int32_t shp_pool_info(const shp_pool* p, uint32_t* n_witness_inputs, uint32_t* n_instances, uint32_t* proof_len) {
    return shp::guarded([&] {
        SHP_REQUIRE(p, SHP_ERR_INVALID_ARG, "pool_info: null pool");
        if (n_witness_inputs) *n_witness_inputs = p->n_inputs;
        if (n_instances) *n_instances = p->n_instances;
        if (proof_len) *proof_len = p->proof_len;
    });
}

int32_t shp_pool_stats(const shp_pool* p, uint64_t* proofs_done, uint64_t* queued) {
    return shp::guarded([&] {
        SHP_REQUIRE(p, SHP_ERR_INVALID_ARG, "pool_stats: null pool");
        if (proofs_done) *proofs_done = p->proofs_done.load();
        if (queued) {
            std::lock_guard<std::mutex> lk(const_cast<shp_pool*>(p)->mu);
            *queued = p->queue.size();
        }
    });
}

}  // extern "C"
