// reslab.h — the hand-over of an ownership-sharded world's bodies between ranks (reslab.hip)
#pragma once

#include "common.h"
#include "comm.h"

#include <vector>

namespace phx {

// one rank's world as phx_world_set_state takes it, plus the scene index of every body
struct SlabState {
    std::vector<long long> global_index;
    std::vector<phx_rigid_body> bodies;
    std::vector<phx_manifold> manifolds;
    std::vector<phx_contact_point> cps;          // two slots per manifold
    std::vector<phx_contact_joint> joints;
};

// how the ranks talk: the library's RCCL communicator on device buffers, or two host callbacks of the caller (include/phyx_amd.h
// phx_slab_transport); one rank needs neither
struct SlabTransport {
    int rank = 0, size = 1;
    Comm* comm = nullptr;
    int (*gather_fn)(void*, const void*, void*, size_t) = nullptr;
    int (*max_fn)(void*, long long*) = nullptr;
    void* user = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<unsigned char> d_send, d_recv;
    DevBuf<int> d_word;
    int all_gather_var(const std::vector<unsigned char>& mine, std::vector<std::vector<unsigned char>>& all);
    int reduce_max(long long* value);
};

void slab_cuts(const double* lo, const double* hi, int n, int nranks, double margin, int* owner, double* bounds);
int reslab_intervals(const SlabState& st, std::vector<long long>& gi, std::vector<double>& lo, std::vector<double>& hi);
void reslab_plan(std::vector<long long>& gi, std::vector<double>& lo, std::vector<double>& hi, int nranks, double margin, std::vector<int>& owner, std::vector<double>& bounds);
int reslab(SlabTransport& tp, SlabState& st, int scene_size, double margin, double bounds[2], int* moved);

} // namespace phx
