"""Shared helpers for the parity tests: scene states produced with the oracle world."""
import numpy as np

from oracle import binding as ob
from phyx_amd import scenes


def oracle_world(scene, gravity=-200.0):
    w = ob.OracleWorld(gravity)
    w.add_scene(scene)
    return w


def presolve_state(scene, warm_steps, iters=15, gravity=-200.0):
    """Run `warm_steps` full oracle steps, then everything of the next step that precedes SolveJoints.
    Returns copies of (bodies, contact_points, joints) = the solver's inputs (warm-start impulses included)."""
    w = oracle_world(scene, gravity)
    for _ in range(warm_steps):
        w.update(contact_iters=iters, penetration_iters=iters)
    w.pre_solve()
    return w.bodies().copy(), w.contact_points().copy(), w.joints().copy()


def is_static(bodies):
    return ((bodies["inv_mass"] == 0) & (bodies["inv_inertia"] == 0)).astype(np.uint8)


SMALL_SCENES = {
    "stack2x10": (lambda: scenes.stack(2, 10), 2),
    "stack10x100": (lambda: scenes.stack(10, 100), 3),
    "tilted60": (lambda: scenes.tilted(60), 25),
    "falling600": (lambda: scenes.falling(600, width=90.0, ymax=300.0), 45),
}
