// Shared between prover.hip (which owns shp_pk) and verify.hip (the native verifier).
#pragma once
#include <utility>
#include <vector>

#include "engine.hpp"
#include "plan.hpp"

struct shp_pk;

// read-only view of the verifying-key material of a proving key
struct VkView {
    const shp::PlanData* plan = nullptr;
    const uint8_t* vk_digest = nullptr;
    const std::vector<shp::G1Affine>* fixed_comm = nullptr;
    const std::vector<shp::G1Affine>* sigma_comm = nullptr;
    shp::Fr omega, delta;
    const std::vector<int32_t>* superset = nullptr;
    // rotation sets as (rots, [(kind, idx)]) -- kinds: 0 adv, 1 pz, 2 lz, 3 lpi, 4 lpt, 5 fix, 6 sig, 7 h, 8 rand
    std::vector<std::pair<std::vector<int32_t>, std::vector<std::pair<int, uint32_t>>>> sets;
    const uint8_t* g2 = nullptr;
    const uint8_t* s_g2 = nullptr;
    bool has_g2 = false;
};
void shp_pk_vk_view(const shp_pk* pk, VkView& out);                                   // prover.hip
// g2 = the standard generator, s_g2 = tau * g2 for an SRS with a known trapdoor   (verify.hip)
void shp_synthetic_g2(const shp::Fr& tau_mont, uint8_t g2_out[128], uint8_t s_g2_out[128]);
