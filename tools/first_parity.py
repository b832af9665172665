"""Scratch driver (not a test): first on-device parity + timing check of the solver C-ABI."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob
import importlib.util as _u; _sp=_u.spec_from_file_location("scenes", os.path.join(os.path.dirname(__file__),"..","phyx_amd","scenes.py")); scenes=_u.module_from_spec(_sp); _sp.loader.exec_module(scenes)

lib = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "phyx_amd", "libphyx_amd.so"))
lib.phx_last_error.restype = C.c_char_p
VP=C.c_void_p
lib.phx_solver_create.argtypes=[C.POINTER(VP), C.c_int]
lib.phx_solver_solve.argtypes=[VP,VP,C.c_int,VP,C.c_int,VP,C.c_int,VP]
lib.phx_solver_get_stats.argtypes=[VP,VP]
lib.phx_solver_get_schedule.argtypes=[VP,VP,C.c_int,VP,C.c_int,VP]
class Cfg(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("d", C.c_int32)]
class Stats(C.Structure):
    _fields_ = [("island_count", C.c_int32), ("island_max_size", C.c_int32), ("colour_count", C.c_int32),
                ("impulse_iterations", C.c_int32), ("displacement_iterations", C.c_int32), ("lds_islands", C.c_int32),
                ("recoloured", C.c_int32), ("reserved", C.c_int32), ("device_ms", C.c_double)]
def ck(st):
    if st < 0: raise RuntimeError("status %d: %s" % (st, lib.phx_last_error()))
print("devices", lib.phx_device_count())
name = C.create_string_buffer(128); cu = C.c_int(); lds = C.c_int(); hbm = C.c_int64()
ck(lib.phx_device_info(0, name, 128, C.byref(cu), C.byref(lds), C.byref(hbm))); print(name.value, cu.value, lds.value, hbm.value)
s = C.c_void_p(); ck(lib.phx_solver_create(C.byref(s), 0))

def run(scene, warm, iters, label):
    w = ob.OracleWorld(); w.add_scene(scene)
    for _ in range(warm): w.update(contact_iters=iters, penetration_iters=iters)
    w.pre_solve()
    bodies = w.bodies().copy(); cps = w.contact_points().copy(); joints = w.joints().copy()
    nb, ncp, nj = len(bodies), len(cps), len(joints)
    gb, gj = bodies.copy(), joints.copy()
    cfg = Cfg(0, 0, iters, iters)
    t = time.time()
    ck(lib.phx_solver_solve(s, gb.ctypes.data, nb, cps.ctypes.data, ncp, gj.ctypes.data, nj, C.byref(cfg)))
    t_first = time.time() - t
    st = Stats(); ck(lib.phx_solver_get_stats(s, C.byref(st)))
    order = np.zeros(nj, np.int32); offs = np.zeros(70000, np.int32); nc = C.c_int32()
    ck(lib.phx_solver_get_schedule(s, order.ctypes.data, nj, offs.ctypes.data, len(offs), C.byref(nc)))
    offs = offs[:nc.value + 1]
    ob_b, ob_j = bodies.copy(), joints.copy()
    t = time.time()
    ost = ob.solver_solve_ordered(ob_b, cps, ob_j, order, offs, iters, iters, ob.STAG_COLOUR_SYNC)
    t_or = time.time() - t
    sq_b, sq_j = bodies.copy(), joints.copy()
    sst = ob.solver_solve_ordered(sq_b, cps, sq_j, order, offs, iters, iters, ob.STAG_SEQUENTIAL)
    same_b = gb.tobytes() == ob_b.tobytes(); same_j = gj.tobytes() == ob_j.tobytes()
    dv = np.abs(gb["velocity"]["x"] - ob_b["velocity"]["x"]).max(), np.abs(gb["velocity"]["y"] - ob_b["velocity"]["y"]).max()
    dj = np.abs(gj["normal_acc"] - ob_j["normal_acc"]).max() if nj else 0
    # second call: schedule reuse + timing
    gb2, gj2 = bodies.copy(), joints.copy()
    t = time.time(); ck(lib.phx_solver_solve(s, gb2.ctypes.data, nb, cps.ctypes.data, ncp, gj2.ctypes.data, nj, C.byref(cfg))); t2 = time.time() - t
    st2 = Stats(); ck(lib.phx_solver_get_stats(s, C.byref(st2)))
    print(f"[{label}] nb={nb} nj={nj} colours={st.colour_count} islands={st.island_count}/{st.island_max_size} "
          f"iters gpu={st.impulse_iterations}/{st.displacement_iterations} oracle={ost.impulse_iterations}/{ost.displacement_iterations} "
          f"BITEXACT bodies={same_b} joints={same_j} max|dv|={dv} max|dacc|={dj} "
          f"seq-vs-coloursync bodies_equal={sq_b.tobytes()==ob_b.tobytes()} stag_events={sst.stag_events} "
          f"first_call={t_first*1e3:.1f}ms second={t2*1e3:.1f}ms dev_ms={st2.device_ms:.3f} recoloured2={st2.recoloured} rerun_same={gb2.tobytes()==gb.tobytes()} oracle={t_or*1e3:.1f}ms")

run(scenes.stack(2, 10), 2, 20, "stack2x10")
run(scenes.stack(10, 100), 3, 20, "stack10x100")
run(scenes.tilted(60), 30, 15, "tilted60")
run(scenes.falling(2000, width=150.0, ymax=400.0), 40, 15, "falling2000")
run(scenes.stack(1000, 200), 3, 20, "stack1000x200")
