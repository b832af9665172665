// c_api_solver.hip — extern "C" surface of the solver (see include/phyx_amd.h for the contract).
#include "handles.h"

#include <algorithm>

extern "C" {

int phx_solver_create(phx_solver** out, int device)
{
    PHX_REQUIRE(out, "null out");
    *out = nullptr;
    PHX_TRY(phx::use_device(device));
    phx_solver* s = new (std::nothrow) phx_solver(device);
    PHX_REQUIRE(s, "out of host memory");
    int st = s->impl.init();
    if (st != PHX_OK) { delete s; return st; }
    *out = s;
    return PHX_OK;
}

void phx_solver_destroy(phx_solver* s) { delete s; }

int phx_solver_solve(phx_solver* s, phx_rigid_body* bodies, int32_t nb, const phx_contact_point* cps, int32_t ncp,
                     phx_contact_joint* joints, int32_t nj, const phx_config* cfg)
{
    PHX_REQUIRE(s && cfg, "null handle / config");
    return s->impl.solve_host(bodies, nb, cps, ncp, joints, nj, *cfg);
}

int phx_solver_solve_device(phx_solver* s, void* d_bodies, int32_t nb, const void* d_cps, int32_t ncp,
                            void* d_joints, int32_t nj, const phx_config* cfg)
{
    PHX_REQUIRE(s && cfg, "null handle / config");
    return s->impl.solve_device(d_bodies, nb, d_cps, ncp, d_joints, nj, *cfg);
}

int phx_solver_solve_resident(phx_solver* s, const phx_body_view* bodies, int32_t nb, const void* d_cps, int32_t ncp,
                              void* d_joints, int32_t nj, const phx_config* cfg)
{
    PHX_REQUIRE(s && cfg && bodies, "null handle / config / view");
    const phx::BodyView v{static_cast<float4*>(bodies->vel), static_cast<float4*>(bodies->dvel), static_cast<float4*>(bodies->mpos)};
    return s->impl.solve_resident(v, nb, d_cps, ncp, d_joints, nj, *cfg);
}

int phx_bodies_to_view(int device, const void* d_bodies, int32_t n, const phx_body_view* out, void* stream)
{
    PHX_REQUIRE(n >= 0 && out && (n == 0 || (d_bodies && out->vel && out->dvel && out->mpos)), "bad arguments");
    PHX_TRY(phx::use_device(device));
    const phx::BodyView v{static_cast<float4*>(out->vel), static_cast<float4*>(out->dvel), static_cast<float4*>(out->mpos)};
    if (n) hipLaunchKernelGGL(phx::k_bodies_to_view, dim3(std::max(1, std::min(phx::div_up(n, 256), 2048))), dim3(256), 0, static_cast<hipStream_t>(stream),
                              static_cast<const phx_rigid_body*>(d_bodies), n, v);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int phx_view_to_bodies(int device, const phx_body_view* in, int32_t n, void* d_bodies, void* stream)
{
    PHX_REQUIRE(n >= 0 && in && (n == 0 || (d_bodies && in->vel && in->dvel && in->mpos)), "bad arguments");
    PHX_TRY(phx::use_device(device));
    const phx::BodyView v{static_cast<float4*>(in->vel), static_cast<float4*>(in->dvel), static_cast<float4*>(in->mpos)};
    if (n) hipLaunchKernelGGL(phx::k_view_to_bodies, dim3(std::max(1, std::min(phx::div_up(n, 256), 2048))), dim3(256), 0, static_cast<hipStream_t>(stream),
                              v, n, static_cast<phx_rigid_body*>(d_bodies), (const unsigned long long*)nullptr, 0ull);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int phx_solver_synchronize(phx_solver* s)
{
    PHX_REQUIRE(s, "null handle");
    PHX_TRY(s->impl.synchronize());
    // the internal waits poll a device-written mailbox (common.h Readback); the public call also drains the stream, so that
    // work the caller orders on OTHER streams afterwards is behind everything queued here
    PHX_HIP(hipStreamSynchronize(s->impl.stream()));
    return PHX_OK;
}

int phx_solver_get_stats(phx_solver* s, phx_solve_stats* out)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_stats(out);
}

int phx_solver_get_schedule(phx_solver* s, int32_t* order, int32_t order_cap, int32_t* offsets, int32_t offsets_cap, int32_t* ncolours)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_schedule(order, order_cap, offsets, offsets_cap, ncolours);
}

int phx_solver_set_body_state_bits(phx_solver* s, int32_t bits)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.set_body_state_bits(bits);
}

int phx_solver_set_shard(phx_solver* s, int32_t shard, int32_t count)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.set_shard(shard, count);
}

int phx_solver_set_schedule_reuse(phx_solver* s, int32_t on)
{
    PHX_REQUIRE(s, "null handle");
    s->impl.set_schedule_reuse(on != 0);
    return PHX_OK;
}

int phx_solver_set_trace(phx_solver* s, int32_t on)
{
    PHX_REQUIRE(s, "null handle");
    s->impl.set_trace(on);
    return PHX_OK;
}

int phx_solver_get_island_trace(phx_solver* s, uint64_t* out, int32_t cap_groups, int32_t* groups)
{
    PHX_REQUIRE(s, "null handle");
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "trace words");
    return s->impl.get_island_trace(reinterpret_cast<unsigned long long*>(out), cap_groups, groups);
}

int phx_solver_get_wave_trace(phx_solver* s, uint64_t* out, int32_t cap_words, int32_t* waves_per_group)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_wave_trace(reinterpret_cast<unsigned long long*>(out), cap_words, waves_per_group);
}

int phx_solver_get_groups(phx_solver* s, int32_t* offsets, int32_t cap, int32_t* count, int32_t* lds_count)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_groups(offsets, cap, count, lds_count);
}

int phx_solver_get_lanes(phx_solver* s, int32_t* leader_slot, int32_t* lane, int32_t cap, int32_t* count)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_lanes(leader_slot, lane, cap, count);
}

int phx_solver_get_partition(phx_solver* s, int32_t* interior_classes, int32_t* parts, int32_t* sweep_launches)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_partition(interior_classes, parts, sweep_launches);
}

int phx_solver_get_refreshed(phx_solver* s, int32_t joint, float out30[30])
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_refreshed(joint, out30);
}

int phx_solver_bench_stage(phx_solver* s, const void* d_bodies, int32_t nb, const void* d_joints, int32_t nj, int32_t steps)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.bench_stage(d_bodies, nb, d_joints, nj, steps);
}

int phx_solver_bench(phx_solver* s, const void* d_bodies, int32_t nb, const void* d_cps, int32_t ncp, const void* d_joints, int32_t nj,
                     const phx_config* cfg, int32_t warmup, int32_t steps, phx_bench_result* out)
{
    PHX_REQUIRE(s && cfg, "null handle / config");
    return s->impl.bench(d_bodies, nb, d_cps, ncp, d_joints, nj, *cfg, warmup, steps, out);
}

int phx_solver_bench_hooked(phx_solver* s, const void* d_bodies, int32_t nb, const void* d_cps, int32_t ncp, const void* d_joints, int32_t nj,
                            const phx_config* cfg, int32_t warmup, int32_t steps, phx_step_hook hook, void* user, phx_bench_result* out)
{
    PHX_REQUIRE(s && cfg, "null handle / config");
    return s->impl.bench(d_bodies, nb, d_cps, ncp, d_joints, nj, *cfg, warmup, steps, out, hook, user);
}

int phx_solver_bench_checksum(phx_solver* s, uint64_t* out)
{
    PHX_REQUIRE(s && out, "null handle / output");
    unsigned long long v = 0;
    const int st = s->impl.bench_checksum(&v);
    *out = v;
    return st;
}

void* phx_solver_stream(phx_solver* s) { return s ? (void*)s->impl.stream() : nullptr; }

uint64_t phx_schedule_priority(uint32_t priority_id, uint32_t joint_index, uint32_t lower_body) { return phx::colour_priority(priority_id, joint_index, lower_body); }

int phx_schedule_colours(const int32_t* b1, const int32_t* b2, int32_t nj, const uint8_t* is_static, int32_t nb, const int32_t* priority_ids,
                         int32_t* order, int32_t* offsets, int32_t offsets_cap, int32_t* ncolours)
{
    PHX_REQUIRE(nj >= 0 && nb >= 0 && (nj == 0 || (b1 && b2 && order)) && (nb == 0 || is_static) && offsets && ncolours, "bad arguments");
    for (int j = 0; j < nj; ++j) PHX_REQUIRE((unsigned)b1[j] < (unsigned)nb && (unsigned)b2[j] < (unsigned)nb, "body index out of range");
    phx::Schedule s;
    phx::build_colour_schedule(b1, b2, nj, is_static, nb, s, priority_ids);
    *ncolours = (int)s.colour_offsets.size() - 1;
    if ((int)s.colour_offsets.size() > offsets_cap) { phx::set_error("colour_offsets too small"); return PHX_ERR_CAPACITY; }
    std::copy(s.order.begin(), s.order.end(), order);
    std::copy(s.colour_offsets.begin(), s.colour_offsets.end(), offsets);
    return PHX_OK;
}

int phx_schedule_groups(const int32_t* b1, const int32_t* b2, int32_t nj, const uint8_t* is_static, int32_t nb, const int32_t* priority_ids,
                        int32_t lanes, int32_t body_cap, int32_t* order, int32_t* offsets, int32_t offsets_cap, int32_t* ncolours,
                        int32_t* group_offsets, int32_t* group_first_colour, int32_t groups_cap, int32_t* lds_groups,
                        int32_t* unit_lane, int32_t* unit_leader_slot)
{
    PHX_REQUIRE(nj >= 0 && nb >= 0 && (nj == 0 || (b1 && b2 && order && unit_lane && unit_leader_slot)) && (nb == 0 || is_static) && offsets && ncolours
                && group_offsets && group_first_colour && lds_groups && lanes > 0 && body_cap > 0, "bad arguments");
    for (int j = 0; j < nj; ++j) PHX_REQUIRE((unsigned)b1[j] < (unsigned)nb && (unsigned)b2[j] < (unsigned)nb, "body index out of range");
    phx::Schedule s;
    phx::LdsCaps caps;
    caps.max_units = lanes; caps.max_joints = 2 * lanes; caps.max_bodies = body_cap; caps.max_colours = 64;
    phx::build_island_schedule(b1, b2, nj, is_static, nb, caps, s, nullptr, priority_ids);
    *ncolours = (int)s.colour_offsets.size() - 1;
    *lds_groups = s.lds_groups;
    if ((int)s.colour_offsets.size() > offsets_cap) { phx::set_error("colour_offsets too small"); return PHX_ERR_CAPACITY; }
    if (s.ngroups() + 1 > groups_cap) { phx::set_error("group arrays too small"); return PHX_ERR_CAPACITY; }
    std::copy(s.order.begin(), s.order.end(), order);
    std::copy(s.colour_offsets.begin(), s.colour_offsets.end(), offsets);
    std::copy(s.group_offsets.begin(), s.group_offsets.end(), group_offsets);
    std::copy(s.group_first_colour.begin(), s.group_first_colour.end(), group_first_colour);
    std::copy(s.unit_lane.begin(), s.unit_lane.end(), unit_lane);
    std::copy(s.unit_leader.begin(), s.unit_leader.end(), unit_leader_slot);
    return (int)s.unit_leader.size();
}

int phx_schedule_islands(const int32_t* b1, const int32_t* b2, int32_t nj, const uint8_t* is_static, int32_t nb,
                         int32_t* joint_island, int32_t* island_size, int32_t island_cap)
{
    PHX_REQUIRE(nj >= 0 && nb >= 0 && (nj == 0 || (b1 && b2 && joint_island)) && (nb == 0 || is_static), "bad arguments");
    for (int j = 0; j < nj; ++j) PHX_REQUIRE((unsigned)b1[j] < (unsigned)nb && (unsigned)b2[j] < (unsigned)nb, "body index out of range");
    std::vector<int> ji, sz;
    phx::gather_islands(b1, b2, nj, is_static, nb, ji, sz);
    if ((int)sz.size() > island_cap) { phx::set_error("island_size too small"); return PHX_ERR_CAPACITY; }
    std::copy(ji.begin(), ji.end(), joint_island);
    if (island_size) std::copy(sz.begin(), sz.end(), island_size);
    return (int)sz.size();
}

} // extern "C"
