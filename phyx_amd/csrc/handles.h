// handles.h — the opaque handle types of the C ABI (include/phyx_amd.h) wrap the C++ objects by value.
#pragma once

#include "broadphase.h"
#include "solver.h"

struct phx_solver {
    phx::DeviceSolver impl;
    explicit phx_solver(int device) : impl(device) {}
};

struct phx_broadphase {
    phx::DeviceBroadphase impl;
    explicit phx_broadphase(int device) : impl(device) {}
};
