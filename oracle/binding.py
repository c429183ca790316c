"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package (towr_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtowr_oracle.so")
_lib = None

SETS_HOT_PATH = 1 | 2 | 8 | 16  # terrain, dynamic, rangeofmotion, force
SETS_TOWR_DEFAULT = 63          # + splineacc-base-lin/-ang (4) and swing-* (32): parameters.cc:55-60
SET_BASE_ROM = 128              # BaseMotionConstraint ("baseMotion"), only when a caller pushes Parameters::BaseRom
SET_TOTAL_TIME = 64             # OptimizePhaseDurations(): ee-schedule_e variables, PhaseSplines, totalduration-e
ROBOTS = {"monoped": 0, "biped": 1, "hyq": 2, "anymal": 3, "go1": 4}
TERRAINS = {"flat": 0, "block": 1, "stairs": 2, "gap": 3, "slope": 4, "chimney": 5, "chimney_lr": 6, "csv": 7,
            "grid_map": 8}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build():
    """Compile the oracle with g++ (a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def use_library(path):
    """Switch to another build of the same oracle (bench.py's cpu_baseline leg times the -O3 -march=native
    build made on the box; the checker build stays the default)."""
    global _lib, _LIB_PATH
    _LIB_PATH = path
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _dp, _ip, C.c_double, C.c_double,
                                 C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                 _dp, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_n_vars", "orc_n_rows", "orc_n_var_sets", "orc_n_con_sets"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        for f in ("orc_var_set_name", "orc_con_set_name"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
            getattr(L, f).restype = C.c_char_p
        for f in ("orc_var_set_size", "orc_con_set_rows"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
            getattr(L, f).restype = C.c_int
        L.orc_initial_guess.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_variable_bounds.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.orc_eval.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip, _dp]
        L.orc_eval.restype = C.c_int
        L.orc_bounds.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_sample_trajectory.argtypes = [C.c_void_p, _dp, C.c_double, _dp, C.c_int]
        L.orc_sample_trajectory.restype = C.c_int
        L.orc_contact_plan.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, _dp, C.c_int]
        L.orc_contact_plan.restype = C.c_int
        L.orc_initial_guess_samples.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp]
        L.orc_planes_world_xy.argtypes = [_dp, _dp, _ip, C.c_int, _dp]
        L.orc_planes_world_xy.restype = None
        L.orc_nearest_plane.argtypes = [_dp, _ip, C.c_int, C.c_double, C.c_double]
        L.orc_nearest_plane.restype = C.c_int
        L.orc_initial_guess_samples.restype = None
        L.orc_time_callbacks.argtypes = [C.c_void_p, _dp, C.c_int]
        L.orc_time_callbacks.restype = C.c_double
        L.orc_gait.argtypes = [C.c_int, C.c_int, C.c_double, _ip, _ip, _dp, C.c_int]
        L.orc_gait.restype = C.c_int
        L.orc_set_grid_map.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_set_grid_map.restype = C.c_int
        L.orc_terrain_probe.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp]
        L.orc_hermite_dpos_dT.argtypes = [C.c_double] * 6
        L.orc_hermite_dpos_dT.restype = C.c_double
        L.orc_euler_probe.argtypes = [_dp, C.c_double, C.c_double, _dp]
        L.orc_hermite_weights.argtypes = [C.c_double, C.c_double, _dp]
        L.orc_terrain_height.argtypes = [C.c_int, C.c_double, C.c_double]
        L.orc_terrain_height.restype = C.c_double
        L.orc_terrain_basis.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        L.orc_terrain_dbasis.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def gait(n_ee, combo, t_total):
    """Reference GaitGenerator: returns (list of per-ee phase duration arrays, contact_at_start list)."""
    n_ph = np.zeros(n_ee, dtype=np.int32)
    con = np.zeros(n_ee, dtype=np.int32)
    out = np.zeros(256)
    w = lib().orc_gait(n_ee, combo, float(t_total), _i(n_ph), _i(con), _d(out), out.size)
    if w < 0:
        raise RuntimeError("orc_gait failed")
    res, o = [], 0
    for ee in range(n_ee):
        res.append(out[o:o + n_ph[ee]].copy())
        o += n_ph[ee]
    return res, [int(c) for c in con]


class OracleProblem:
    def __init__(self, robot, terrain, phase_durations, contact_at_start, dt_dynamic=0.1, dt_rom=0.08,
                 duration_base_poly=0.1, polys_per_swing=2, polys_per_stance_force=3, force_limit=1000.0,
                 constraint_sets=SETS_HOT_PATH, dt_base_motion=None, base_z_init=0.0, grid=None, grid_map=None):
        robot = ROBOTS[robot] if isinstance(robot, str) else robot
        terrain = TERRAINS[terrain] if isinstance(terrain, str) else terrain
        n_ee = len(phase_durations)
        n_ph = np.array([len(p) for p in phase_durations], dtype=np.int32)
        pd = np.concatenate([np.asarray(p, dtype=np.float64) for p in phase_durations])
        con = np.array(contact_at_start, dtype=np.int32)
        self._h = lib().orc_create(robot, terrain, n_ee, _i(n_ph), _d(pd), _i(con), dt_dynamic, dt_rom,
                                   duration_base_poly, polys_per_swing, polys_per_stance_force, force_limit,
                                   int(constraint_sets),
                                   float(duration_base_poly / 4.0 if dt_base_motion is None else dt_base_motion),
                                   float(base_z_init),
                                   *self._grid_args(grid))
        if not self._h:
            raise RuntimeError("orc_create failed")
        L = lib()
        if grid_map is not None:   # (elevation[size_x, size_y] float32, resolution, (pos_x, pos_y)) of the `Grid` terrain
            elev, res, pos = grid_map
            self._gm = np.asfortranarray(elev, dtype=np.float32)   # grid_map's column-major storage
            if L.orc_set_grid_map(self._h, self._gm.ctypes.data_as(C.POINTER(C.c_float)), self._gm.shape[0],
                                  self._gm.shape[1], float(res), float(pos[0]), float(pos[1])) != 0:
                raise RuntimeError("orc_set_grid_map failed")
        self.n_ee = n_ee
        self.n = L.orc_n_vars(self._h)
        self.m = L.orc_n_rows(self._h)
        self.var_sets = [(L.orc_var_set_name(self._h, i).decode(), L.orc_var_set_size(self._h, i))
                         for i in range(L.orc_n_var_sets(self._h))]
        self.con_sets = [(L.orc_con_set_name(self._h, i).decode(), L.orc_con_set_rows(self._h, i))
                         for i in range(L.orc_n_con_sets(self._h))]
        self.nnz = L.orc_eval(self._h, _d(np.zeros(self.n)), None, None, None, None)

    def _grid_args(self, grid):
        if grid is None:
            return None, 0, 0
        self._grid = np.ascontiguousarray(grid, dtype=np.float64)   # grid[y_cell, x_cell] (HeightMapFromCSV)
        return _d(self._grid), self._grid.shape[0], self._grid.shape[1]

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter exit)
            lib().orc_destroy(self._h)
            self._h = None

    def initial_guess(self, base_lin0, base_ang0, base_lin1, base_ang1, ee_pos0):
        x = np.zeros(self.n)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (base_lin0, base_ang0, base_lin1, base_ang1)]
        ee = np.ascontiguousarray(ee_pos0, dtype=np.float64).reshape(-1)
        lib().orc_initial_guess(self._h, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(ee), _d(x))
        return x

    def variable_bounds(self, init_base, final_base, ee_pos0):
        """init_base / final_base: 12 doubles {lin p, lin v, ang p, ang v}."""
        a = np.ascontiguousarray(init_base, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(final_base, dtype=np.float64).reshape(-1)
        ee = np.ascontiguousarray(ee_pos0, dtype=np.float64).reshape(-1)
        assert a.size == 12 and b.size == 12 and ee.size == 3 * self.n_ee
        lo, up = np.zeros(self.n), np.zeros(self.n)
        lib().orc_variable_bounds(self._h, _d(a), _d(b), _d(ee), _d(lo), _d(up))
        return lo, up

    def eval(self, x):
        """Returns g, row_ptr, col_idx, vals (CSR, explicit zeros kept)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.n
        g = np.zeros(self.m)
        rp = np.zeros(self.m + 1, dtype=np.int32)
        ci = np.zeros(self.nnz, dtype=np.int32)
        va = np.zeros(self.nnz)
        nnz = lib().orc_eval(self._h, _d(x), _d(g), _i(rp), _i(ci), _d(va))
        assert nnz == self.nnz, "Jacobian pattern must not depend on x"
        return g, rp, ci, va

    def values(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.m)
        lib().orc_eval(self._h, _d(x), _d(g), None, None, None)
        return g

    def sample_trajectory(self, x, dt=0.01):
        """fpowr::GetTrajectory: (n_samples, 20 + 13 n_ee) array, see towr_oracle.cc for the record layout."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = lib().orc_sample_trajectory(self._h, _d(x), float(dt), None, 0)
        out = np.zeros((n, 20 + 13 * self.n_ee))
        lib().orc_sample_trajectory(self._h, _d(x), float(dt), _d(out), n)
        return out

    def contact_plan(self, x, dt=0.01, time_horizon=2.0):
        """fpowr::ExtractFootstepPlan: (n_steps, 2 + 4 n_ee) array [t, duration, contact flags, ee positions]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = lib().orc_contact_plan(self._h, _d(x), float(dt), float(time_horizon), None, 0)
        out = np.zeros((n, 2 + 4 * self.n_ee))
        lib().orc_contact_plan(self._h, _d(x), float(dt), float(time_horizon), _d(out), n)
        return out

    def initial_guess_samples(self, x, times):
        """fpowr::ExtractInitialGuess at `times`: (n, 49) array [t | state 12 | controls 36]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        times = np.ascontiguousarray(times, dtype=np.float64)
        out = np.zeros((len(times), 49))
        lib().orc_initial_guess_samples(self._h, _d(x), _d(times), len(times), _d(out))
        return out

    def terrain_probe(self, x, y):
        """(height, dh/dx, dh/dy) of the problem's terrain."""
        o = np.zeros(3)
        lib().orc_terrain_probe(self._h, float(x), float(y), _d(o))
        return o

    def bounds(self):
        lo, up = np.zeros(self.m), np.zeros(self.m)
        lib().orc_bounds(self._h, _d(lo), _d(up))
        return lo, up

    def time_callbacks(self, x, iters):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().orc_time_callbacks(self._h, _d(x), int(iters))


def hermite_weights(t, T):
    w = np.zeros(12)
    lib().orc_hermite_weights(t, T, _d(w))
    return w.reshape(3, 4)


def hermite_dpos_dT(t, T, p0, v0, p1, v1):
    return lib().orc_hermite_dpos_dT(t, T, p0, v0, p1, v1)


def euler_probe(nodes12, T, t):
    """EulerConverter on one base-ang polynomial; dict of M, Mdot, R, omega, omega_dot and their node derivatives."""
    n = np.ascontiguousarray(nodes12, dtype=np.float64)
    o = np.zeros(429)
    lib().orc_euler_probe(_d(n), float(T), float(t), _d(o))
    d = o[33:]
    return dict(M=o[0:9].reshape(3, 3), Mdot=o[9:18].reshape(3, 3), R=o[18:27].reshape(3, 3), omega=o[27:30],
                omega_dot=o[30:33], dM=d[0:108].reshape(3, 3, 12), dMdot=d[108:216].reshape(3, 3, 12),
                dR=d[216:324].reshape(3, 3, 12), domega=d[324:360].reshape(3, 12), domega_dot=d[360:396].reshape(3, 12))


def terrain_height(terrain, x, y):
    return lib().orc_terrain_height(TERRAINS[terrain], x, y)


def terrain_basis(terrain, which, x, y):
    o = np.zeros(3)
    lib().orc_terrain_basis(TERRAINS[terrain], which, x, y, _d(o))
    return o


def terrain_dbasis(terrain, which, dim, x, y):
    o = np.zeros(3)
    lib().orc_terrain_dbasis(TERRAINS[terrain], which, dim, x, y, _d(o))
    return o


def planes_world_xy(regions, local_xy, start):
    """fpowr::PlanarRegionsToPolygons: regions (n, 7) [position xyz, orientation xyzw], boundary points local_xy (m, 2),
    start (n + 1) -> world xy (m, 2)."""
    regions = np.ascontiguousarray(regions, dtype=np.float64)
    local_xy = np.ascontiguousarray(local_xy, dtype=np.float64)
    start = np.ascontiguousarray(start, dtype=np.int32)
    out = np.zeros_like(local_xy)
    lib().orc_planes_world_xy(_d(regions), _d(local_xy), start.ctypes.data_as(C.POINTER(C.c_int)), len(regions), _d(out))
    return out


def nearest_plane(world_xy, start, px, py):
    """fpowr::NearestPlaneLookup::GetNearestPlaneIndex."""
    world_xy = np.ascontiguousarray(world_xy, dtype=np.float64)
    start = np.ascontiguousarray(start, dtype=np.int32)
    return lib().orc_nearest_plane(_d(world_xy), start.ctypes.data_as(C.POINTER(C.c_int)), len(start) - 1, float(px), float(py))
