/*
 * drop_in.c — the C ABI driven from plain C, the way the reference's own loop drives World/Solver
 * (ref: src/main.cpp:88-103 resetWorld + :353-364 stats line, src/World.cpp:19-37 Update).
 *
 *   gcc -std=c11 -O2 -Iinclude examples/drop_in.c -Lphyx_amd -lphyx_amd -Wl,-rpath,$PWD/phyx_amd -o drop_in
 *   ./drop_in [columns rows steps]
 *
 * Part 1 steps a stack scene with phx_world_* (the whole step on the device).
 * Part 2 is the drop-in call a maintainer would make from World::Update line 34: the host arrays of that world go
 * through phx_solver_solve exactly as `bodies.data / contactPoints.data / contactJoints.data` would.
 * Exit status: 0 ok, 3 no usable device (there is no CPU fallback), 1 any other failure.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "phyx_amd.h"

#define TRY(call)                                                                      \
    do {                                                                               \
        int st_ = (call);                                                              \
        if (st_ != PHX_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, phx_last_error());           \
            return st_ == PHX_ERR_NO_DEVICE ? 3 : 1;                                   \
        }                                                                              \
    } while (0)

int main(int argc, char** argv)
{
    const int columns = argc > 1 ? atoi(argv[1]) : 20, rows = argc > 2 ? atoi(argv[2]) : 50, steps = argc > 3 ? atoi(argv[3]) : 10;
    if (phx_abi_version() != PHX_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }

    /* ---- part 1: the device-resident World (ref: main.cpp:88-103 builds the same kind of scene) */
    phx_world* world = NULL;
    TRY(phx_world_create(&world, 0));
    TRY(phx_world_set_gravity(world, -200.0f));
    int ground = phx_world_add_body(world, 0.0f, 0.0f, 0.0f, 15.0f * (float)columns, 10.0f);
    if (ground < 0) { fprintf(stderr, "add_body: %s\n", phx_last_error()); return 1; }
    TRY(phx_world_set_body_static(world, ground));
    for (int c = 0; c < columns; ++c)
        for (int r = 0; r < rows; ++r)
            if (phx_world_add_body(world, 15.0f * (float)c - 7.5f * (float)(columns - 1), 15.0f + 10.0f * (float)r, 0.0f, 5.0f, 5.0f) < 0) return 1;
    const phx_config cfg = { PHX_SOLVE_AVX2, PHX_ISLAND_MULTIPLE, 15, 15 };      /* ref: Configuration.h:20-23 */
    for (int s = 0; s < steps; ++s) TRY(phx_world_update(world, 1.0f / 60.0f, &cfg));
    int32_t nb = 0, nm = 0, ncp = 0, nj = 0;
    TRY(phx_world_counts(world, &nb, &nm, &ncp, &nj));
    phx_solve_stats ss;
    TRY(phx_world_get_solve_stats(world, &ss));
    printf("world: %d bodies %d manifolds %d joints, islands %d (max %d joints), %d colours, %d impulse sweeps\n",
           nb, nm, nj, ss.island_count, ss.island_max_size, ss.colour_count, ss.impulse_iterations);

    /* ---- part 2: the solver as a drop-in for Solver::SolveJoints (ref: World.cpp:34) on host arrays */
    TRY(phx_world_pre_solve(world, 1.0f / 60.0f));       /* everything of World::Update before the solver call */
    TRY(phx_world_counts(world, &nb, &nm, &ncp, &nj));
    phx_rigid_body* bodies = malloc(sizeof *bodies * (size_t)(nb > 0 ? nb : 1));
    phx_contact_point* cps = malloc(sizeof *cps * (size_t)(ncp > 0 ? ncp : 1));
    phx_contact_joint* joints = malloc(sizeof *joints * (size_t)(nj > 0 ? nj : 1));
    if (!bodies || !cps || !joints) return 1;
    TRY(phx_world_get_bodies(world, bodies, nb));
    TRY(phx_world_get_contact_points(world, cps, ncp));
    TRY(phx_world_get_joints(world, joints, nj));
    phx_solver* solver = NULL;
    TRY(phx_solver_create(&solver, 0));
    TRY(phx_solver_solve(solver, bodies, nb, cps, ncp, joints, nj, &cfg));
    TRY(phx_solver_get_stats(solver, &ss));
    double vy = 0.0;
    for (int i = 0; i < nb; ++i) vy += bodies[i].velocity.y;
    printf("drop-in solve: %d joints, islands %d, %d impulse sweeps, %lld joint visits, mean vy %.6f\n",
           nj, ss.island_count, ss.impulse_iterations, (long long)ss.joint_visits, nb ? vy / nb : 0.0);
    /* the same step finished by the world must agree with the drop-in call bit for bit (same inputs, same schedule) */
    TRY(phx_world_finish_step(world, 0.0f, &cfg));       /* dt = 0: solve, positions untouched */
    phx_rigid_body* again = malloc(sizeof *again * (size_t)(nb > 0 ? nb : 1));
    if (!again) return 1;
    TRY(phx_world_get_bodies(world, again, nb));
    int same = 1;
    for (int i = 0; i < nb && same; ++i)
        same = memcmp(&bodies[i].velocity, &again[i].velocity, sizeof bodies[i].velocity) == 0 &&
               memcmp(&bodies[i].angular_velocity, &again[i].angular_velocity, sizeof(float)) == 0;
    printf("drop-in call vs world step: %s\n", same ? "identical" : "DIFFERENT");
    free(again); free(bodies); free(cps); free(joints);
    phx_solver_destroy(solver);
    phx_world_destroy(world);
    return same ? 0 : 1;
}
