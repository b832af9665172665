/*
 * sharded.c — BASELINE config 3 from plain C: one process per GPU, every rank steps a replica of the same world, solves its
 * own share of the islands (schedule groups) and meets the others in ONE RCCL all-gather per step (the per-step barrier of the
 * north star; counterpart of the reference merging its islands' bodies after the parallel island loop,
 * ref: src/Solver.cpp:86-91, 482-494, 527-547).  No Python anywhere: the transport is the library's own (phx_comm_*).
 *
 *   gcc -std=c11 -O2 -Iinclude examples/sharded.c -Lphyx_amd -lphyx_amd -Wl,-rpath,$PWD/phyx_amd -o sharded
 *   ./sharded [columns rows steps]                       one rank (the collective still runs, over one rank)
 *   PHX_RANK=r PHX_NRANKS=n PHX_ID_FILE=/tmp/id ./sharded ...    rank r of n, one process per GPU (device = r unless
 *                                                        PHX_DEVICE says otherwise); rank 0 writes the communicator id to
 *                                                        the file, the others wait for it — any launcher will do
 *
 * Rank 0 also steps an UNSHARDED world beside the sharded one and compares every body after every step: the sharded step must
 * reproduce it bit for bit.  A coda does the other sharding mode: every rank steps a world of its own x-slab of whole columns and the
 * ranks re-slab once (phx_world_reslab) over the same communicator.  Exit status: 0 ok, 3 no usable device / no RCCL (there is no CPU fallback), 1 any other failure.
 */
#define _DEFAULT_SOURCE      /* usleep */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "phyx_amd.h"

#define TRY(call)                                                                      \
    do {                                                                               \
        int st_ = (call);                                                              \
        if (st_ != PHX_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, phx_last_error());           \
            return st_ == PHX_ERR_NO_DEVICE ? 3 : 1;                                   \
        }                                                                              \
    } while (0)

static int env_int(const char* name, int fallback) { const char* v = getenv(name); return v && *v ? atoi(v) : fallback; }

static int build_scene(phx_world* w, int columns, int rows)
{
    if (phx_world_set_gravity(w, -200.0f) != PHX_OK) return 1;
    const int ground = phx_world_add_body(w, 0.0f, 0.0f, 0.0f, 15.0f * (float)columns, 10.0f);
    if (ground < 0 || phx_world_set_body_static(w, ground) != PHX_OK) return 1;
    for (int c = 0; c < columns; ++c)
        for (int r = 0; r < rows; ++r)
            if (phx_world_add_body(w, ((float)c - (float)columns / 2.0f) * 15.0f, 15.0f + 10.0f * (float)r, 0.0f, 5.0f, 5.0f) < 0) return 1;
    return 0;
}

int main(int argc, char** argv)
{
    const int columns = argc > 1 ? atoi(argv[1]) : 16, rows = argc > 2 ? atoi(argv[2]) : 30, steps = argc > 3 ? atoi(argv[3]) : 10;
    const int rank = env_int("PHX_RANK", 0), nranks = env_int("PHX_NRANKS", 1), device = env_int("PHX_DEVICE", nranks > 1 ? rank : 0);
    const char* id_file = getenv("PHX_ID_FILE");
    if (phx_abi_version() != PHX_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    if (nranks > 1 && !id_file) { fprintf(stderr, "PHX_ID_FILE is needed for more than one rank\n"); return 1; }

    /* ---- rendezvous: rank 0 makes the id, everybody else reads it */
    unsigned char id[PHX_COMM_ID_BYTES];
    if (rank == 0) {
        TRY(phx_comm_unique_id(id));
        if (id_file) {
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot write %s\n", tmp); return 1; }
            fclose(f);
            if (rename(tmp, id_file) != 0) { fprintf(stderr, "cannot publish %s\n", id_file); return 1; }
        }
    } else {
        FILE* f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no communicator id in %s\n", rank, id_file); return 1; }
        fclose(f);
    }
    phx_comm* comm = NULL;
    TRY(phx_comm_create(&comm, id, rank, nranks, device));

    /* ---- every rank: the same world, sharded by rank */
    phx_world* world = NULL;
    TRY(phx_world_create(&world, device));
    if (build_scene(world, columns, rows)) { fprintf(stderr, "scene: %s\n", phx_last_error()); return 1; }
    TRY(phx_world_set_comm(world, comm));
    /* rank 0: the unsharded twin */
    phx_world* twin = NULL;
    if (rank == 0) {
        TRY(phx_world_create(&twin, device));
        if (build_scene(twin, columns, rows)) { fprintf(stderr, "scene: %s\n", phx_last_error()); return 1; }
    }
    const phx_config cfg = {PHX_SOLVE_AVX2, PHX_ISLAND_MULTIPLE, 15, 15};       /* ref: main.cpp:348 iteration counts */
    const int nb = columns * rows + 1;
    phx_rigid_body* a = (phx_rigid_body*)malloc((size_t)nb * sizeof *a);
    phx_rigid_body* b = (phx_rigid_body*)malloc((size_t)nb * sizeof *b);
    if (!a || !b) return 1;
    int differing_steps = 0;
    for (int s = 0; s < steps; ++s) {
        TRY(phx_world_step_sharded(world, 1.0f / 60.0f, &cfg));
        if (twin) {
            TRY(phx_world_update(twin, 1.0f / 60.0f, &cfg));
            TRY(phx_world_get_bodies(world, a, nb));
            TRY(phx_world_get_bodies(twin, b, nb));
            if (memcmp(a, b, (size_t)nb * sizeof *a) != 0) ++differing_steps;
        }
    }
    TRY(phx_world_check_exchange(world));
    TRY(phx_comm_barrier(comm, phx_world_stream(world)));
    int32_t async_error = 0;
    TRY(phx_comm_async_error(comm, &async_error));
    if (rank == 0) {
        int32_t nbodies = 0, nm = 0, ncp = 0, nj = 0;
        phx_solve_stats st;
        TRY(phx_world_counts(world, &nbodies, &nm, &ncp, &nj));
        TRY(phx_world_get_solve_stats(world, &st));
        printf("rank 0 of %d: %d bodies, %d manifolds, %d joints, %d island groups, %d steps; RCCL async error %d\n", nranks, nbodies, nm, nj,
               st.lds_islands, steps, async_error);
        printf("sharded world vs unsharded world: %s\n", differing_steps ? "DIFFERENT" : "identical after every step");
    }
    /* ---- ownership sharding (slab mode): this rank's x-slab of the same scene — whole columns = whole islands, plus the
     *      static ground — as a world of its own; the ranks meet only to RE-SLAB (phx_world_reslab: intervals first, states only if
     *      somebody changes owner), here once, over the same communicator, collectives on device buffers */
    {
        const int first = (int)((long long)columns * rank / nranks), last = (int)((long long)columns * (rank + 1) / nranks);
        phx_world* slab = NULL;
        TRY(phx_world_create(&slab, device));
        TRY(phx_world_set_gravity(slab, -200.0f));
        int64_t* scene_index = (int64_t*)malloc((size_t)nb * sizeof *scene_index);
        if (!scene_index) return 1;
        int32_t mine = 0;
        const int ground = phx_world_add_body(slab, 0.0f, 0.0f, 0.0f, 15.0f * (float)columns, 10.0f);
        if (ground < 0 || phx_world_set_body_static(slab, ground) != PHX_OK) return 1;
        scene_index[mine++] = 0;
        for (int c = first; c < last; ++c)
            for (int r = 0; r < rows; ++r) {
                if (phx_world_add_body(slab, ((float)c - (float)columns / 2.0f) * 15.0f, 15.0f + 10.0f * (float)r, 0.0f, 5.0f, 5.0f) < 0) return 1;
                scene_index[mine++] = 1 + (int64_t)c * rows + r;
            }
        for (int s = 0; s < 3; ++s) TRY(phx_world_update(slab, 1.0f / 60.0f, &cfg));
        const phx_slab_transport transport = {rank, nranks, comm, NULL, NULL, NULL};
        double bounds[2] = {0.0, 0.0};
        int32_t moved = 0;
        const int32_t before = mine;
        TRY(phx_world_reslab(slab, &transport, scene_index, nb, &mine, nb, 1.0, bounds, &moved));
        for (int s = 0; s < 2; ++s) TRY(phx_world_update(slab, 1.0f / 60.0f, &cfg));
        float extent[2] = {0.f, 0.f};
        TRY(phx_world_x_extent(slab, extent));
        const int inside = mine <= 1 || ((double)extent[0] > bounds[0] && (double)extent[1] < bounds[1]);
        printf("rank %d of %d, slab mode: %d bodies before the re-slab, %d after (%s), slab (%g, %g), bodies within it: %s\n", rank, nranks, before, mine,
               moved ? "bodies changed owner" : "nobody moved: the world was kept", bounds[0], bounds[1], inside ? "yes" : "NO");
        free(scene_index);
        phx_world_destroy(slab);
        if (!inside) return 1;
    }
    free(a); free(b);
    if (twin) phx_world_destroy(twin);
    phx_world_destroy(world);
    phx_comm_destroy(comm);
    return differing_steps || async_error ? 1 : 0;
}
