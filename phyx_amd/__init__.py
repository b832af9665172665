"""phyx_amd — MI355X-native sequential-impulse contact solver + sweep-and-prune broadphase.

Host-side mirror of the reference's World / Solver / Collider / Configuration interface
(ref: src/World.h, src/Solver.h, src/Collider.h, src/Configuration.h) over the C ABI of
libphyx_amd.so (include/phyx_amd.h).  All compute runs in hand-written HIP kernels for gfx950.
"""
from .api import (Configuration, Solver, Collider, World, Comm, SOLVE_SCALAR, SOLVE_SSE2, SOLVE_AVX2,  # noqa: F401
                  ISLAND_SINGLE, ISLAND_MULTIPLE, ISLAND_SINGLE_SLOPPY, ISLAND_MULTIPLE_SLOPPY,
                  rigid_body_dtype, contact_point_dtype, manifold_dtype, contact_joint_dtype,
                  broadphase_entry_dtype, sort_entry_dtype, device_count, device_info, DeviceArray, DeviceBuffer, exchange_layout, schedule_colours, schedule_groups, schedule_islands, schedule_priority)
from ._lib import PhxError  # noqa: F401
from . import scenes  # noqa: F401
