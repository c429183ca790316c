"""towr_amd -- MI355X-native evaluation of towr's NLP constraint/Jacobian callback.

Thin ctypes layer over libtowr_amd.so (C ABI in include/towr_amd.h).  There is no CPU
fallback: every evaluation runs the HIP kernels; importing on a box without the built
library raises, evaluating without a GPU raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWR_AMD_LIB") or os.path.join(_HERE, "libtowr_amd.so")   # (override: diagnostic builds)

MAX_EE, MAX_PHASES, NAME_LEN = 4, 32, 40
ROBOTS = {"monoped": 0, "biped": 1, "hyq": 2, "anymal": 3, "go1": 4}
TERRAINS = {"flat": 0, "block": 1, "stairs": 2, "gap": 3, "slope": 4, "chimney": 5, "chimney_lr": 6, "csv": 7,
            "grid_map": 8}
EVAL_VALUES, EVAL_JACOBIAN, EVAL_BOTH, EVAL_CHECK = 1, 2, 3, 4
SET_TERRAIN, SET_DYNAMIC, SET_BASE_ACC, SET_ROM, SET_FORCE, SET_SWING, SET_TOTAL_TIME = 1, 2, 4, 8, 16, 32, 64
SET_BASE_ROM = 128
SETS_HOT_PATH, SETS_TOWR_DEFAULT, SETS_ALL, SETS_EVERY = 27, 63, 127, 255  # TWR_SETS_* of include/towr_amd.h
FAMILIES = ("terrain", "dynamic", "splineacc", "rangeofmotion", "force", "swing", "totalduration", "baseMotion")  # twr_batch_score
SUPPORTS_OPTIMISED_TIMINGS = True  # TWR_SET_TOTAL_TIME has a device path


class Model(C.Structure):
    _fields_ = [("n_ee", C.c_int32), ("terrain_id", C.c_int32), ("mass", C.c_double), ("inertia", C.c_double * 6),
                ("nominal_stance", (C.c_double * 3) * MAX_EE), ("max_dev", C.c_double * 3), ("gravity", C.c_double),
                ("friction", C.c_double), ("force_limit", C.c_double), ("flat_height", C.c_double)]


class Schedule(C.Structure):
    _fields_ = [("n_ee", C.c_int32), ("n_phases", C.c_int32 * MAX_EE), ("in_contact_at_start", C.c_int32 * MAX_EE),
                ("phase_durations", (C.c_double * MAX_PHASES) * MAX_EE)]

    def durations(self):
        return [list(self.phase_durations[e][:self.n_phases[e]]) for e in range(self.n_ee)]

    def contact(self):
        return [int(self.in_contact_at_start[e]) for e in range(self.n_ee)]


class Params(C.Structure):
    _fields_ = [("dt_dynamic", C.c_double), ("dt_rom", C.c_double), ("duration_base_poly", C.c_double),
                ("polys_per_swing", C.c_int32), ("polys_per_stance_force", C.c_int32),
                ("constraint_sets", C.c_int32), ("reserved_", C.c_int32),
                ("dt_base_motion", C.c_double), ("base_z_init", C.c_double)]


class Sizes(C.Structure):
    _fields_ = [("n_vars", C.c_int32), ("n_rows", C.c_int32), ("nnz", C.c_int32), ("n_var_sets", C.c_int32),
                ("n_con_sets", C.c_int32), ("k_dynamic", C.c_int32), ("k_rom", C.c_int32)]


class SetInfo(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("offset", C.c_int32), ("size", C.c_int32),
                ("nnz_offset", C.c_int32), ("nnz", C.c_int32)]


_lib = None
_dp = C.POINTER(C.c_double)


def build():
    """Compile libtowr_amd.so for gfx950 with hipcc (in-tree)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if
    libtowr_amd.so pulled in /opt/rocm's copy first, a later `import torch` would bring up a second
    runtime and see no GPU.  So when torch is installed, its runtime is loaded first (without
    importing torch) and libtowr_amd.so binds to it through the shared SONAME."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libtowr_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C towr_amd/csrc`")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.twr_last_error.restype = C.c_char_p
        L.twr_model_preset.argtypes = [C.c_int, C.c_int, C.POINTER(Model)]
        L.twr_params_default.argtypes = [C.POINTER(Params)]
        L.twr_gait_combo.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(Schedule)]
        L.twr_structure_create.argtypes = [C.POINTER(Model), C.POINTER(Schedule), C.POINTER(Params),
                                           C.POINTER(C.c_void_p)]
        L.twr_structure_destroy.argtypes = [C.c_void_p]
        L.twr_terrain_grid_create.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.twr_terrain_grid_map_create.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                  C.POINTER(C.c_void_p)]
        L.twr_terrain_grid_destroy.argtypes = [C.c_void_p]
        L.twr_structure_create_with_grid.argtypes = [C.POINTER(Model), C.POINTER(Schedule), C.POINTER(Params), C.c_void_p,
                                                     C.POINTER(C.c_void_p)]
        L.twr_structure_destroy.restype = None
        L.twr_structure_create_many.argtypes = [C.POINTER(Model), C.POINTER(Schedule), C.POINTER(Params), C.c_int, C.c_int,
                                                C.POINTER(C.c_void_p)]
        L.twr_structure_create_many_with_grid.argtypes = [C.POINTER(Model), C.POINTER(Schedule), C.POINTER(Params), C.c_int,
                                                          C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.twr_candidate_bytes.argtypes = [C.POINTER(Model), C.POINTER(Schedule), C.POINTER(Params), C.c_int, C.c_int,
                                          C.POINTER(C.c_int64)]
        L.twr_shard_bounds.argtypes = [_dp, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.twr_terrain_grid_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            _dp, _dp, _dp, C.POINTER(C.c_void_p)]
        L.twr_structure_sizes.argtypes = [C.c_void_p, C.POINTER(Sizes)]
        L.twr_structure_var_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(SetInfo)]
        L.twr_structure_con_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(SetInfo)]
        L.twr_structure_row_ptr.argtypes = [C.c_void_p]
        L.twr_structure_row_ptr.restype = C.POINTER(C.c_int32)
        L.twr_structure_col_idx.argtypes = [C.c_void_p]
        L.twr_structure_col_idx.restype = C.POINTER(C.c_int32)
        L.twr_structure_bounds.argtypes = [C.c_void_p, _dp, _dp]
        L.twr_structure_initial_guess.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]
        L.twr_structure_variable_bounds.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.twr_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]
        L.twr_batch_destroy.argtypes = [C.c_void_p]
        L.twr_batch_destroy.restype = None
        L.twr_batch_num_problems.argtypes = [C.c_void_p]
        L.twr_batch_table_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.twr_batch_streaming_stores.argtypes = [C.c_void_p]
        L.twr_batch_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.twr_batch_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.twr_batch_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
        L.twr_batch_eval_host.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int]
        L.twr_structure_sample_count.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int32)]
        L.twr_batch_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]
        L.twr_batch_initial_guess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.twr_batch_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.twr_batch_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.twr_batch_score_best.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p]
        L.twr_structure_contact_steps_max.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.twr_structure_values_items.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
        L.twr_planes_create.argtypes = [_dp, _dp, C.POINTER(C.c_int32), C.c_int32, C.c_int, C.POINTER(C.c_void_p)]
        L.twr_planes_destroy.argtypes = [C.c_void_p]
        L.twr_planes_destroy.restype = None
        L.twr_planes_world_xy.argtypes = [C.c_void_p, _dp]
        L.twr_batch_contact_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.twr_batch_contact_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_int32, C.c_void_p,
                                             C.c_void_p]
        L.twr_batch_host_buffers.argtypes = [C.c_void_p, C.POINTER(_dp), C.POINTER(_dp), C.POINTER(_dp)]
        L.twr_batch_profile_begin.argtypes = [C.c_void_p, C.c_int]
        L.twr_batch_profile_end.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int)]
        _lib = L
    return _lib


class TowrError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise TowrError("towr_amd error %d: %s" % (rc, lib().twr_last_error().decode()))


def _d(a):
    return a.ctypes.data_as(_dp)


def model_preset(robot, terrain):
    """RobotModel(robot) + HeightMap::MakeTerrain(terrain) as the POD model blob."""
    m = Model()
    r = ROBOTS[robot] if isinstance(robot, str) else robot
    t = TERRAINS[terrain] if isinstance(terrain, str) else terrain
    _check(lib().twr_model_preset(r, t, C.byref(m)))
    return m


def params_default(**kw):
    p = Params()
    _check(lib().twr_params_default(C.byref(p)))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def gait_combo(n_ee, combo, t_total, swing_scale=1.0):
    s = Schedule()
    _check(lib().twr_gait_combo(n_ee, combo, float(t_total), float(swing_scale), C.byref(s)))
    return s


def schedule(phase_durations, contact_at_start):
    s = Schedule()
    s.n_ee = len(phase_durations)
    for e, pd in enumerate(phase_durations):
        s.n_phases[e] = len(pd)
        s.in_contact_at_start[e] = int(contact_at_start[e])
        for i, d in enumerate(pd):
            s.phase_durations[e][i] = float(d)
    return s


class TerrainGrid:
    """Gridded terrain of HeightMapFromCSV: heights[y_cell, x_cell] (0.17 m cells)."""

    def __init__(self, heights):
        a = np.ascontiguousarray(heights, dtype=np.float64)
        assert a.ndim == 2
        self._h = C.c_void_p()
        _check(lib().twr_terrain_grid_create(_d(a), a.shape[0], a.shape[1], C.byref(self._h)))
        self.heights = a

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().twr_terrain_grid_destroy(self._h)
            self._h = None


class GridMap(TerrainGrid):
    """The "elevation" layer of a ROS grid_map for the `Grid` terrain (TWR_TERRAIN_GRID_MAP):
    elevation[i, j] float32 with i along -x and j along -y (grid_map's index convention, start index (0,0)),
    cell size `resolution`, map centre `position`."""

    def __init__(self, elevation, resolution, position=(0.0, 0.0)):
        a = np.asfortranarray(elevation, dtype=np.float32)   # grid_map's Eigen::MatrixXf is column-major
        assert a.ndim == 2
        self._h = C.c_void_p()
        _check(lib().twr_terrain_grid_map_create(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1],
                                                 float(resolution), float(position[0]), float(position[1]),
                                                 C.byref(self._h)))
        self.elevation, self.resolution, self.position = a, float(resolution), (float(position[0]), float(position[1]))
        self.heights = None


class Planes:
    """The planar regions of a terrain message as world polygons on a device (fpowr PlanarRegionsToPolygons /
    NearestPlaneLookup): regions (n, 7) [position xyz | orientation xyzw], boundaries = one (k, 2) array of local
    outer-boundary points per region."""

    def __init__(self, regions, boundaries, device=0):
        regions = np.ascontiguousarray(regions, dtype=np.float64).reshape(-1, 7)
        assert len(regions) == len(boundaries)
        pts = [np.asarray(b, dtype=np.float64).reshape(-1, 2) for b in boundaries]
        self.start = np.concatenate([[0], np.cumsum([len(b) for b in pts])]).astype(np.int32)
        xy = np.ascontiguousarray(np.concatenate(pts) if pts else np.zeros((0, 2)))
        self._h = C.c_void_p()
        _check(lib().twr_planes_create(_d(regions), _d(xy), self.start.ctypes.data_as(C.POINTER(C.c_int32)), len(regions),
                                       int(device), C.byref(self._h)))
        self.world_xy = np.zeros_like(xy)
        if len(xy):
            _check(lib().twr_planes_world_xy(self._h, _d(self.world_xy)))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().twr_planes_destroy(self._h)
            self._h = None


class Structure:
    """x-independent part of one candidate (index maps, time tables, CSR pattern).  The set tables and the CSR
    pattern are fetched from the library on first use (a sweep builds a thousand of these)."""

    def __init__(self, model, sched, params=None, grid=None, _handle=None):
        params = params or params_default()
        self._h = C.c_void_p()
        self.model, self.schedule, self.params, self.grid = model, sched, params, grid
        if _handle is not None:
            self._h = _handle
        elif grid is not None:
            _check(lib().twr_structure_create_with_grid(C.byref(model), C.byref(sched), C.byref(params), grid._h,
                                                        C.byref(self._h)))
        else:
            _check(lib().twr_structure_create(C.byref(model), C.byref(sched), C.byref(params), C.byref(self._h)))
        sz = Sizes()
        _check(lib().twr_structure_sizes(self._h, C.byref(sz)))
        self._sz = sz
        self.n, self.m, self.nnz = sz.n_vars, sz.n_rows, sz.nnz
        self.k_dynamic, self.k_rom = sz.k_dynamic, sz.k_rom
        self.n_ee = model.n_ee
        self._lazy = {}

    @classmethod
    def create_many(cls, model, scheds, params_list, threads=0, grid=None):
        """twr_structure_create_many[_with_grid]: the candidates of a sweep, built on `threads` host threads (0 = all);
        with a gridded terrain they all share `grid`."""
        n = len(scheds)
        assert n == len(params_list) and n > 0
        sa = (Schedule * n)(*scheds)
        pa = (Params * n)(*params_list)
        hs = (C.c_void_p * n)()
        _check(lib().twr_structure_create_many_with_grid(C.byref(model), sa, pa, n, int(threads),
                                                         grid._h if grid is not None else None, hs))
        return [cls(model, scheds[i], params_list[i], grid=grid, _handle=C.c_void_p(hs[i])) for i in range(n)]

    def _sets(self, fn, n):
        out = []
        for i in range(n):
            si = SetInfo()
            _check(fn(self._h, i, C.byref(si)))
            out.append(dict(name=si.name.decode(), offset=si.offset, size=si.size, nnz_offset=si.nnz_offset,
                            nnz=si.nnz))
        return out

    def _get(self, key, make):
        if key not in self._lazy:
            self._lazy[key] = make()
        return self._lazy[key]

    @property
    def var_sets(self):
        return self._get("var_sets", lambda: self._sets(lib().twr_structure_var_set, self._sz.n_var_sets))

    @property
    def con_sets(self):
        return self._get("con_sets", lambda: self._sets(lib().twr_structure_con_set, self._sz.n_con_sets))

    @property
    def row_ptr(self):
        return self._get("row_ptr", lambda: np.ctypeslib.as_array(lib().twr_structure_row_ptr(self._h),
                                                                  shape=(self.m + 1,)).copy())

    @property
    def col_idx(self):
        return self._get("col_idx", lambda: np.ctypeslib.as_array(lib().twr_structure_col_idx(self._h),
                                                                  shape=(self.nnz,)).copy())

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter exit)
            lib().twr_structure_destroy(self._h)
            self._h = None

    @property
    def algorithmic_bytes(self):
        """read x once, write g once, write the Jacobian values once."""
        return 8 * (self.n + self.m + self.nnz)

    def bounds(self):
        lo, up = np.zeros(self.m), np.zeros(self.m)
        _check(lib().twr_structure_bounds(self._h, _d(lo), _d(up)))
        return lo, up

    def initial_guess(self, base_lin0, base_ang0, base_lin1, base_ang1, ee_pos0):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (base_lin0, base_ang0, base_lin1, base_ang1)]
        ee = np.ascontiguousarray(ee_pos0, dtype=np.float64).reshape(-1)
        assert ee.size == 3 * self.n_ee
        x = np.zeros(self.n)
        _check(lib().twr_structure_initial_guess(self._h, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(ee), _d(x)))
        return x

    def contact_steps_max(self):
        """Upper bound of the footstep states twr_batch_contact_plan can produce for this structure."""
        n = C.c_int32()
        _check(lib().twr_structure_contact_steps_max(self._h, C.byref(n)))
        return n.value

    def values_items(self):
        """The work items of the values-only path (twr_structure_values_items): {"dynamic": [(k0, cnt, widest window)],
        "rom": [...], "dynamic_takes_rom": bool}; empty lists for a structure that keeps the Jacobian kernels' cut."""
        nd, nr, both = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().twr_structure_values_items(self._h, C.byref(nd), C.byref(nr), C.byref(both), None))
        items = np.zeros((nd.value + nr.value, 3), dtype=np.int32)
        _check(lib().twr_structure_values_items(self._h, C.byref(nd), C.byref(nr), C.byref(both), items.ctypes.data))
        rows = [tuple(int(v) for v in r) for r in items]
        return {"dynamic": rows[:nd.value], "rom": rows[nd.value:], "dynamic_takes_rom": bool(both.value)}

    def sample_count(self, dt=0.01):
        """Records fpowr::GetTrajectory produces for this structure at step dt."""
        n = C.c_int32()
        _check(lib().twr_structure_sample_count(self._h, float(dt), C.byref(n)))
        return n.value

    def variable_bounds(self, init_base, final_base, ee_pos0):
        """x_l, x_u of the reference's variable sets; base states = 12 doubles {lin p, lin v, ang p, ang v}."""
        a = np.ascontiguousarray(init_base, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(final_base, dtype=np.float64).reshape(-1)
        ee = np.ascontiguousarray(ee_pos0, dtype=np.float64).reshape(-1)
        assert a.size == 12 and b.size == 12 and ee.size == 3 * self.n_ee
        lo, up = np.zeros(self.n), np.zeros(self.n)
        _check(lib().twr_structure_variable_bounds(self._h, _d(a), _d(b), _d(ee), _d(lo), _d(up)))
        return lo, up


class Batch:
    """Device tables of a batch of candidates; problem p uses structures[struct_of_problem[p]]."""

    def __init__(self, structures, struct_of_problem=None, device=0):
        if struct_of_problem is None:
            struct_of_problem = list(range(len(structures)))
        self.structures = list(structures)
        self.struct_of_problem = np.ascontiguousarray(struct_of_problem, dtype=np.int32)
        hs = (C.c_void_p * len(structures))(*[s._h for s in structures])
        self._h = C.c_void_p()
        _check(lib().twr_batch_create(hs, len(structures), self.struct_of_problem.ctypes.data_as(C.POINTER(C.c_int32)),
                                      len(self.struct_of_problem), device, C.byref(self._h)))
        self.device = device
        self.n_problems = len(self.struct_of_problem)
        xo = np.zeros(self.n_problems + 1, dtype=np.int64)
        go, jo = xo.copy(), xo.copy()
        p64 = C.POINTER(C.c_int64)
        _check(lib().twr_batch_layout(self._h, xo.ctypes.data_as(p64), go.ctypes.data_as(p64), jo.ctypes.data_as(p64)))
        self.x_off, self.g_off, self.jac_off = xo, go, jo
        self.algorithmic_bytes = int(8 * (xo[-1] + go[-1] + jo[-1]))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().twr_batch_destroy(self._h)
            self._h = None

    def eval_device(self, d_x, d_g, d_jac, flags=EVAL_BOTH, stream=0):
        """Asynchronous launch on raw device pointers (ints), e.g. torch tensors' data_ptr()."""
        _check(lib().twr_batch_eval(self._h, C.c_void_p(d_x), C.c_void_p(d_g), C.c_void_p(d_jac), flags,
                                    C.c_void_p(stream)))

    def table_bytes(self):
        """Device bytes of the batch's tables: resident, layout tables of the dynamic set as built, and what is left of them
        after byte-identical tables of different structures were merged (twr_batch_table_bytes)."""
        r, a, d = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().twr_batch_table_bytes(self._h, C.byref(r), C.byref(a), C.byref(d)))
        return dict(resident=r.value, dyn_layout=a.value, dyn_layout_distinct=d.value)

    def streaming_stores(self):
        """True when the batch streams its Jacobian values out with non-temporal stores (twr_batch_streaming_stores)."""
        return bool(lib().twr_batch_streaming_stores(self._h))

    def status(self, stream=0):
        """Per-problem non-finite flags of the last eval_device(..., flags | EVAL_CHECK): bit 0 g, bit 1 jac."""
        st = np.zeros(self.n_problems, dtype=np.int32)
        _check(lib().twr_batch_status(self._h, st.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(stream)))
        return st

    def profile_begin(self, max_evals):
        """Record HIP events around each kernel of the next `max_evals` eval_device calls."""
        _check(lib().twr_batch_profile_begin(self._h, int(max_evals)))

    def profile_end(self):
        """Average duration [ms] of the dynamic / range-of-motion / node kernels, and the eval count."""
        ms = np.zeros(3)
        n = C.c_int(0)
        _check(lib().twr_batch_profile_end(self._h, _d(ms), C.byref(n)))
        return dict(dynamic=float(ms[0]), rangeofmotion=float(ms[1]), nodes=float(ms[2])), n.value

    def kernel_bytes(self):
        """Algorithmic bytes per launch of each kernel: it reads x once and writes its own rows of g
        and its own Jacobian values once (SURVEY 8d applied per kernel; x is counted for each kernel,
        so the three figures sum to algorithmic_bytes + 2*8*n per problem)."""
        out = dict(dynamic=0, rangeofmotion=0, nodes=0)
        for si in self.struct_of_problem:
            S = self.structures[si]
            for cs in S.con_sets:
                k = "dynamic" if cs["name"] == "dynamic" else ("rangeofmotion" if cs["name"].startswith("rangeofmotion")
                                                               else "nodes")
                out[k] += 8 * (cs["size"] + cs["nnz"])
            for k in out:
                out[k] += 8 * S.n
        return out

    def sample_device(self, d_x, dt, d_out, problem_stride, stream=0):
        """twr_batch_sample: d_out[p * problem_stride + sample * (20 + 13 n_ee) + field] (device pointers)."""
        _check(lib().twr_batch_sample(self._h, C.c_void_p(d_x), float(dt), C.c_void_p(d_out), int(problem_stride),
                                      C.c_void_p(stream)))

    def initial_guess_device(self, d_x, d_times, n_times, d_out, problem_stride, stream=0):
        """twr_batch_initial_guess (fpowr ExtractInitialGuess): d_out[p * problem_stride + 49 s + field], sample s at
        d_times[s]; record [t | state 12 | controls 36] (device pointers)."""
        _check(lib().twr_batch_initial_guess(self._h, C.c_void_p(d_x), C.c_void_p(d_times), int(n_times), C.c_void_p(d_out),
                                             int(problem_stride), C.c_void_p(stream)))

    def score_device(self, d_g, d_scores, stream=0):
        """twr_batch_score: d_scores[16 p + 2 f + {0: inf-norm, 1: 1-norm}] of the bound violation per family f."""
        _check(lib().twr_batch_score(self._h, C.c_void_p(d_g), C.c_void_p(d_scores), C.c_void_p(stream)))

    def score_best_device(self, d_g, d_scores, d_best, families=(0, 1, 3, 4), index_offset=0, stream=0):
        """twr_batch_score_best: score_device + best_device over this batch's candidates behind one call; d_best[0] is
        index_offset + the winner's index in the batch."""
        mask = 0
        for f in families:
            mask |= 1 << int(f)
        _check(lib().twr_batch_score_best(self._h, C.c_void_p(d_g), C.c_void_p(d_scores), mask, int(index_offset), C.c_void_p(d_best),
                                          C.c_void_p(stream)))

    def best_device(self, d_scores, n_candidates, d_best, families=(0, 1, 3, 4), stream=0):
        """twr_batch_best: device arg-min of the summed inf-norm violations of `families` (indices into FAMILIES) over a
        score table of n_candidates rows (this batch's, or the all-gathered one); d_best = 2 doubles [index, total]."""
        mask = 0
        for f in families:
            mask |= 1 << int(f)
        _check(lib().twr_batch_best(self._h, C.c_void_p(d_scores), int(n_candidates), mask, C.c_void_p(d_best), C.c_void_p(stream)))

    def contact_plan_device(self, d_x, dt, time_horizon, d_out, max_steps, d_counts, stream=0):
        """twr_batch_contact_plan (fpowr ExtractFootstepPlan without the plane lookup)."""
        _check(lib().twr_batch_contact_plan(self._h, C.c_void_p(d_x), float(dt), float(time_horizon), C.c_void_p(d_out),
                                            int(max_steps), C.c_void_p(d_counts), C.c_void_p(stream)))

    def contact_planes_device(self, planes, d_plan, d_counts, max_steps, d_plane_index, stream=0):
        """twr_batch_contact_planes: nearest planar region of every foot in contact of every footstep state of
        contact_plan_device(); d_plane_index int32[n_problems, max_steps, n_ee], -1 = in the air / no such state."""
        _check(lib().twr_batch_contact_planes(self._h, planes._h, C.c_void_p(d_plan), C.c_void_p(d_counts), int(max_steps),
                                              C.c_void_p(d_plane_index), C.c_void_p(stream)))

    def host_buffers(self):
        """Page-locked x / g / jac arrays owned by the batch (numpy views); eval_host_pinned() uses them."""
        px, pg, pj = _dp(), _dp(), _dp()
        _check(lib().twr_batch_host_buffers(self._h, C.byref(px), C.byref(pg), C.byref(pj)))
        as_np = lambda p, n: np.ctypeslib.as_array(p, shape=(int(n),))
        return as_np(px, self.x_off[-1]), as_np(pg, self.g_off[-1]), as_np(pj, self.jac_off[-1])

    def eval_host_pinned(self, flags=EVAL_BOTH):
        """Evaluate from / into the page-locked buffers of host_buffers() (fill x there first)."""
        px, pg, pj = _dp(), _dp(), _dp()
        _check(lib().twr_batch_host_buffers(self._h, C.byref(px), C.byref(pg), C.byref(pj)))
        _check(lib().twr_batch_eval_host(self._h, px, pg, pj, flags))

    def eval_host(self, x, flags=EVAL_BOTH):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.x_off[-1]
        g = np.zeros(self.g_off[-1])
        j = np.zeros(self.jac_off[-1])
        _check(lib().twr_batch_eval_host(self._h, _d(x), _d(g), _d(j), flags))
        return g, j
