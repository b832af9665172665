// c_api_solver.hip — extern "C" surface of the solver (see include/phyx_amd.h for the contract).
#include "solver.h"

struct phx_solver { phx::DeviceSolver impl; explicit phx_solver(int d) : impl(d) {} };

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

int phx_solver_synchronize(phx_solver* s)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.synchronize();
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

int phx_solver_get_refreshed(phx_solver* s, int32_t joint, float out30[30])
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.get_refreshed(joint, out30);
}

int phx_solver_bench(phx_solver* s, const void* d_bodies, int32_t nb, const void* d_cps, int32_t ncp, const void* d_joints, int32_t nj,
                     const phx_config* cfg, int32_t warmup, int32_t steps, phx_bench_result* out)
{
    PHX_REQUIRE(s && cfg, "null handle / config");
    return s->impl.bench(d_bodies, nb, d_cps, ncp, d_joints, nj, *cfg, warmup, steps, out);
}

} // extern "C"
