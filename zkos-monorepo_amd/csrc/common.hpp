// Shared host-side plumbing of libshielder_hip: status codes, last-error slot, HIP call
// checking, a grow-only device workspace.  No torch types, no oracle code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string>
#include <stdexcept>
#include <vector>

#include "../../include/shielder_hip.h"

namespace shp {

void set_last_error(const std::string& msg);

struct Error : std::runtime_error {
    int32_t code;
    int hip_code;       // the hipError_t behind an SHP_ERR_HIP (0 otherwise)
    Error(int32_t c, const std::string& m, int hip = 0) : std::runtime_error(m), code(c), hip_code(hip) {}
};

#define SHP_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            throw ::shp::Error(SHP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e), (int)_e); \
    } while (0)

#define SHP_REQUIRE(cond, code, msg)                          \
    do {                                                      \
        if (!(cond)) throw ::shp::Error((code), (msg));       \
    } while (0)

// every extern "C" entry point runs its body through this: nothing unwinds across the ABI
template <class F>
static inline int32_t guarded(F&& f) {
    try {
        f();
        return SHP_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return SHP_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown exception");
        return SHP_ERR_INTERNAL;
    }
}

// grow-only device buffer (workspace that survives across calls; never freed in a launch path)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) SHP_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 2);
        SHP_HIP(hipMalloc(&p, want));
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// growable page-locked host staging buffer: asynchronous copies to or from pageable memory go through the
// runtime's own staging path, which serialises the streams of concurrent proving lanes
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) SHP_HIP(hipHostFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 2) + 256;
        SHP_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace shp

// stream waits with the time spent in them accounted per thread (SHP_PHASE_TRACE prints host time = total - waits)
namespace shp {
inline double& wait_us_accum() {
    static thread_local double v = 0;
    return v;
}
// A lone proof's host thread does not wait for the STREAM after a kernel whose results it needs, but for the results:
// the kernel's last workgroup stores them into pinned host memory and then, behind a system-scope fence, a stamp (a
// launch sequence number) per result.  The host spins on the stamps and goes on as soon as they are there -- the kernel's
// own end-of-kernel release, its completion signal and the runtime's wait on it (5 to 10 us a round trip, nine round trips
// a proof) pass while the host already hashes and launches.  Every 2048 spins the stream is asked whether it has drained:
// a drained stream has made every store visible (and reports a failed kernel), so the wait ends there whatever the stamps say.
inline void wait_stamps(const uint32_t* stamps, uint32_t count, uint32_t seq, hipStream_t st) {
    const volatile uint32_t* sp = stamps;
    uint32_t spins = 0;
    for (uint32_t i = 0; i < count;) {
        if (sp[i] == seq) { i++; continue; }
        if ((++spins & 2047u) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) SHP_HIP(q);
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
__device__ __forceinline__ void publish_stamp(uint32_t* stamp, uint32_t seq) {
    __threadfence_system();
    __hip_atomic_store(stamp, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline void sync_stream(hipStream_t st) {
    static const bool trace = getenv("SHP_PHASE_TRACE") != nullptr;
    if (!trace) {
        SHP_HIP(hipStreamSynchronize(st));
        return;
    }
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    SHP_HIP(hipStreamSynchronize(st));
    clock_gettime(CLOCK_MONOTONIC, &b);
    wait_us_accum() += (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
}
}  // namespace shp

// Cross-lane MSM combiner of a proving pool (implemented in msm.hip, owned by the pool): opaque here
namespace shp {
struct MsmCombiner;
MsmCombiner* msm_combiner_create(int device);
void msm_combiner_destroy(MsmCombiner* c);
}  // namespace shp
// internal (not part of the C ABI): the GPU a proving key lives on
int shp_pk_device(const shp_pk* pk);
// internal (not part of the C ABI): commitments of this lane go through the pool's combiner; `profile_owner` lanes
// report the combiner's kernel timers with their own (shp_prover_profile_read)
void shp_prover_attach_combiner(shp_prover* p, shp::MsmCombiner* c, bool profile_owner);
int32_t shp_prove_assigned_rows(shp_prover* p, const uint64_t* advice, size_t n_columns, size_t rows, const uint64_t* instances,
                                size_t n_instances, const uint64_t* randomness, size_t n_draws, const uint8_t* seed,
                                uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
void shp_pk_shape(const shp_pk* pk, uint32_t* num_advice, uint32_t* n, uint32_t* usable, bool* phased, uint64_t* n_draws);
