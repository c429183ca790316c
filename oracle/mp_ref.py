"""Independent second implementation of the hot path (TEST INFRASTRUCTURE ONLY).

Purpose: pin the C++ oracle (oracle/towr_oracle.cc) with something that shares no
code with it.  This file evaluates the constraint values g(x) from the *math*
(reference lines cited inline) in 40-digit mpmath arithmetic and
obtains the Jacobian by high-precision central differences -- it contains no
analytic derivative at all, so a wrong closed form in the oracle (or in the
reference restatement) cannot hide.

Run as a script to (re)generate tests/golden/mp_*.npz:
    python -m oracle.mp_ref            # all cases, ~25 min on 8 cores (MP_REF_PROCS sets the worker count)
    python -m oracle.mp_ref c2_biped_walk_flat_k100 c3_anymal_trot_flat_k200 c5_anymal_walk_stairs_k200
                                       # the BASELINE-size cases only (round 4), ~12 min on 5 cores

Double-precision inputs that the reference computes in double (time grids,
polynomial durations, active segment, local time) are computed here in Python
floats (IEEE double) with the same operation order, then promoted exactly.

One deliberate exception, documented in DESIGN.md: the reference's
"derivative of the normalised terrain basis" (height_map.cc:80-91) is a
component-wise product, not a chain rule.  The entries it feeds (force rows w.r.t.
the stance foothold x/y) are therefore *not* true derivatives where the terrain
has curvature (Gap); those entries are flagged in the fixture (`quirk_mask`) and
checked by a hand known-answer test instead.
"""
import math
import os
import sys
from multiprocessing import Pool

import numpy as np
from mpmath import mp, mpf

mp.dps = 40

ROBOT = {  # models/examples/*.h, models/go1/go1_model.h
    "monoped": dict(n_ee=1, m=20.0, I=(1.2, 5.5, 6.0, 0.0, -0.2, -0.01), nom=[(0.0, 0.0, -0.58)], dev=(0.25, 0.15, 0.2)),
    "biped": dict(n_ee=2, m=20.0, I=(1.209, 5.583, 6.056, 0.005, -0.190, -0.012),
                  nom=[(0.0, 0.20, -0.65), (0.0, -0.20, -0.65)], dev=(0.25, 0.15, 0.15)),
    "hyq": dict(n_ee=4, m=83.0, I=(4.26, 8.97, 9.88, -0.0063, 0.193, 0.0126),
                nom=[(0.31, 0.29, -0.58), (0.31, -0.29, -0.58), (-0.31, 0.29, -0.58), (-0.31, -0.29, -0.58)],
                dev=(0.25, 0.20, 0.10)),
    "anymal": dict(n_ee=4, m=29.5, I=(0.946438, 1.94478, 2.01835, 0.000938112, -0.00595386, -0.00146328),
                   nom=[(0.34, 0.19, -0.42), (0.34, -0.19, -0.42), (-0.34, 0.19, -0.42), (-0.34, -0.19, -0.42)],
                   dev=(0.15, 0.1, 0.10)),
    "go1": dict(n_ee=4, m=12.84, I=(0.0168128557, 0.063009565, 0.0716547275, -0.0002296769, -0.0002945293, -0.0000418731),
                nom=[(0.1881, 0.04675 + 0.08, -0.3), (0.1881, -(0.04675 + 0.08), -0.3),
                     (-0.1881, 0.04675 + 0.08, -0.3), (-0.1881, -(0.04675 + 0.08), -0.3)], dev=(0.16, 0.12, 0.06)),
}
G = 9.80665  # dynamic_model.cc:37
MU = 0.5     # height_map.h:136


# ----------------------------------------------------------------- layout
def base_poly_durations(T, dt):  # parameters.cc:82-98
    out, left = [], T
    while left > 1e-10:
        out.append(dt if left > dt else left)
        left -= dt
    return out


def time_grid(T, dt):  # time_discretization_constraint.cc:37-50
    t, g = 0.0, [0.0]
    for _ in range(int(math.floor(T / dt))):
        t += dt
        g.append(t)
    g.append(T)
    return g


def locate(t, durs):  # spline.cc:48-78
    acc = 0.0
    seg = None
    for i, d in enumerate(durs):
        acc += d
        if acc >= t - 1e-10:
            seg = i
            break
    assert seg is not None
    tl = t
    for i in range(seg):
        tl -= durs[i]
    return seg, tl


class Layout:
    """Variable layout + node tables for one problem (nlp_formulation.cc:63-181,
    nodes_variables_all.cc:45-61, nodes_variables_phase_based.cc:38-298)."""

    def __init__(self, robot, phase_durations, contact_at_start, dt_dyn=0.1, dt_rom=0.08, dur_base=0.1,
                 polys_swing=2, polys_stance=3, optimize_timings=False, dt_base_motion=None):
        self.rb = ROBOT[robot]
        self.n_ee = self.rb["n_ee"]
        self.T = 0.0
        for d in phase_durations[0]:
            self.T += d
        self.base_durs = base_poly_durations(self.T, dur_base)
        nb = len(self.base_durs) + 1
        self.splines = {}  # name -> dict(durs, nodes=[{(deriv,dim): idx or None}], const=[bool])
        off = 0
        for name in ("base-lin", "base-ang"):
            nodes = []
            for k in range(nb):
                nodes.append({(dv, dm): off + 6 * k + 3 * dv + dm for dv in (0, 1) for dm in (0, 1, 2)})
            self.splines[name] = dict(durs=self.base_durs, nodes=nodes, off=off, size=6 * nb)
            off += 6 * nb
        for ee in range(self.n_ee):  # ee-motion: stance phases constant
            off = self._phase_based("ee-motion_%d" % ee, phase_durations[ee], bool(contact_at_start[ee]), polys_swing,
                                    off, motion=True)
        for ee in range(self.n_ee):  # ee-force: swing phases constant (zero)
            off = self._phase_based("ee-force_%d" % ee, phase_durations[ee], not bool(contact_at_start[ee]),
                                    polys_stance, off, motion=False)
        # ee-schedule<e> (phase_durations.cc:39-52): all phase durations but the last are variables
        self.optimize_timings = optimize_timings
        self.schedule = []
        for ee in range(self.n_ee):
            t_total = 0.0
            for d in phase_durations[ee]:
                t_total += d  # std::accumulate
            self.schedule.append(dict(off=off, size=len(phase_durations[ee]) - 1, t_total=t_total,
                                      initial=list(phase_durations[ee])))
            if optimize_timings:
                off += len(phase_durations[ee]) - 1
        self.n = off
        self.grid_dyn = time_grid(self.T, dt_dyn)
        self.grid_rom = time_grid(self.T, dt_rom)
        self.grid_bm = time_grid(self.T, dur_base / 4.0 if dt_base_motion is None else dt_base_motion)

    def _phase_based(self, name, phases, first_const, n_change, off, motion):
        polys = []  # (phase, is_const)
        const = first_const
        for ph in range(len(phases)):
            if const:
                polys.append((ph, True, 1))
            else:
                polys += [(ph, False, n_change)] * n_change
            const = not const
        durs = [phases[ph] / n for (ph, _, n) in polys]
        n_nodes = len(polys) + 1

        def node_const(i):
            adj = [0] if i == 0 else ([n_nodes - 2] if i == n_nodes - 1 else [i - 1, i])
            return any(polys[p][1] for p in adj)

        nodes = [dict() for _ in range(n_nodes)]
        consts = [node_const(i) for i in range(n_nodes)]
        idx = off
        i = 0
        while i < n_nodes:
            if not consts[i]:
                if motion:  # px vx py vy pz  (vz fixed 0)
                    for dm in (0, 1, 2):
                        nodes[i][(0, dm)] = idx
                        idx += 1
                        if dm != 2:
                            nodes[i][(1, dm)] = idx
                            idx += 1
                else:  # px vx py vy pz vz
                    for dm in (0, 1, 2):
                        nodes[i][(0, dm)] = idx
                        nodes[i][(1, dm)] = idx + 1
                        idx += 2
                i += 1
            else:
                if motion:  # one position shared by both nodes, velocities zero
                    for dm in (0, 1, 2):
                        nodes[i][(0, dm)] = idx
                        nodes[i + 1][(0, dm)] = idx
                        idx += 1
                i += 2
        self.splines[name] = dict(durs=durs, nodes=nodes, const=consts, polys=polys, off=off, size=idx - off,
                                  ee=int(name.split("_")[1]))
        return idx

    def poly_durations(self, name, x):
        """Polynomial durations of a spline: fixed (NodeSpline) or, with optimised timings, phase duration /
        polynomials in the phase with the phase durations taken from x and the last one filling up to the
        total time (PhaseSpline, phase_spline.cc:54-65; PhaseDurations::SetVariables, phase_durations.cc:77-103)."""
        s = self.splines[name]
        if not self.optimize_timings or "polys" not in s:
            return s["durs"]
        sc = self.schedule[s["ee"]]
        ph = [x[sc["off"] + i] for i in range(sc["size"])]
        ph.append(mpf(sc["t_total"]) - sum(ph))
        return [ph[p] / n for (p, _, n) in s["polys"]]


# ----------------------------------------------------------------- math
def hermite(n0, n1, T, t):
    """pos, vel, acc of the cubic through (p0,v0)->(p1,v1) over T at t (polynomial.cc:97-104,47-61)."""
    (p0, v0), (p1, v1) = n0, n1
    c = -(3 * (p0 - p1) + T * (2 * v0 + v1)) / T ** 2
    d = (2 * (p0 - p1) + T * (v0 + v1)) / T ** 3
    return (p0 + v0 * t + c * t ** 2 + d * t ** 3, v0 + 2 * c * t + 3 * d * t ** 2, 2 * c + 6 * d * t)


def spline_point(L, name, x, t):
    s = L.splines[name]
    durs = L.poly_durations(name, x)
    if durs is s["durs"]:
        seg, tl = locate(t, durs)
        T, tl = mpf(durs[seg]), mpf(tl)
    else:  # x-dependent durations: the active segment by the reference's double-precision rule, the rest exact
        seg, _ = locate(t, [float(d) for d in durs])
        T, tl = durs[seg], mpf(t) - sum(durs[:seg])
    out = [[None] * 3 for _ in range(3)]
    for dm in range(3):
        def val(node, dv):
            i = s["nodes"][node].get((dv, dm))
            return x[i] if i is not None else mpf(0)
        pva = hermite((val(seg, 0), val(seg, 1)), (val(seg + 1, 0), val(seg + 1, 1)), T, tl)
        for d in range(3):
            out[d][dm] = pva[d]
    return out  # [pos, vel, acc][dim]


def rot(e):  # euler_converter.cc:207-221 (ZYX)
    x, y, z = e
    sx, cx, sy, cy, sz, cz = mp.sin(x), mp.cos(x), mp.sin(y), mp.cos(y), mp.sin(z), mp.cos(z)
    return [[cy * cz, cz * sx * sy - cx * sz, sx * sz + cx * cz * sy],
            [cy * sz, cx * cz + sx * sy * sz, cx * sy * sz - cz * sx],
            [-sy, cy * sx, cx * cy]]


def omega_and_dot(e, ed, edd):  # euler_converter.cc:133-166, 58-83
    y, z = e[1], e[2]
    yd, zd = ed[1], ed[2]
    sy, cy, sz, cz = mp.sin(y), mp.cos(y), mp.sin(z), mp.cos(z)
    M = [[cy * cz, -sz, 0], [cy * sz, cz, 0], [-sy, 0, 1]]
    Md = [[-cz * sy * yd - cy * sz * zd, -cz * zd, 0], [cy * cz * zd - sy * sz * yd, -sz * zd, 0], [-cy * yd, 0, 0]]
    w = [sum(M[i][j] * ed[j] for j in range(3)) for i in range(3)]
    wd = [sum(Md[i][j] * ed[j] + M[i][j] * edd[j] for j in range(3)) for i in range(3)]
    return w, wd


def matvec(A, v):
    return [sum(A[i][j] * v[j] for j in range(3)) for i in range(3)]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


FLT_MAX = float(np.finfo(np.float32).max)


class GridMapTerrain:
    """The `Grid` height map fpowr hands the solver (towr/include/towr/terrain/grid_height_map.h:15-60,
    fpowr/src/footstep_plan_server.cc:155), written from that header -- NOT from oracle/towr_oracle.cc:
      GetHeight            = (double)(float) map_.atPosition("elevation", (x, y), INTER_LINEAR); out_of_range -> FLT_MAX  (:30-46)
      GetHeightDerivWrtX/Y = (float(h(+eps)) - float(h(-eps))) / (2 eps), the difference taken in FLOAT, eps = resolution / 6  (:23,48-60)
    The second height derivatives are not overridden (height_map.h: 0), so the reference's force rows carry explicit
    ZEROS in their foothold columns on this terrain.
    grid_map itself is third party and absent from /root/reference (unpinned distro package, Dockerfile); its half is
    restated here from the published grid_map_core algorithm (GridMap::atPosition / atPositionLinearInterpolated /
    getIndexFromPosition / getPositionFromIndex / checkIfPositionWithinMap), start index (0, 0):
      cell (i, j) centre = position + length / 2 - ((i, j) + 1/2) * resolution      (x falls with i, y falls with j)
      index of a position = trunc(-(p - length / 2 - position) / resolution), valid iff 0 <= length/2 - (p - position) < length
      bilinear over the cell of p and its neighbours TOWARDS p; a neighbour outside the buffer -> nearest cell; p outside
      the map -> std::out_of_range.
    elevation[i, j] is float32 (grid_map::Matrix = Eigen::MatrixXf)."""

    def __init__(self, elevation, resolution, position):
        self.e = np.asarray(elevation, dtype=np.float32)
        self.res = float(resolution)
        self.pos = (float(position[0]), float(position[1]))
        self.size = self.e.shape
        self.length = (self.size[0] * self.res, self.size[1] * self.res)
        self.eps = self.res / 6.0

    def _index(self, p):
        idx, inside = [], True
        for a in (0, 1):
            v = (p[a] - 0.5 * self.length[a] - self.pos[a]) / self.res
            idx.append(int(-v))                       # Eigen cast<int>: truncation towards zero
            t = -(p[a] - self.pos[a] - 0.5 * self.length[a])
            inside = inside and (0.0 <= t < self.length[a])
            inside = inside and (0 <= idx[a] < self.size[a])
        return idx, inside

    def _centre(self, idx):
        return [self.pos[a] + 0.5 * self.length[a] - (idx[a] + 0.5) * self.res for a in (0, 1)]

    def at_position(self, x, y):
        """float32 value of atPosition(INTER_LINEAR), or None for std::out_of_range."""
        p = (float(x), float(y))
        i0, inside = self._index(p)
        c0 = self._centre(i0)
        # the three other corners lie towards p: index - 1 where p >= centre (positions fall with the index)
        step = [-1 if p[a] >= c0[a] else +1 for a in (0, 1)]
        lo = [min(i0[a], i0[a] + step[a]) for a in (0, 1)]   # the corner with the LARGEST position has the smallest index
        corners = [(lo[0] + di, lo[1] + dj) for di in (0, 1) for dj in (0, 1)]
        n = self.size[0] * self.size[1]
        # grid_map's own range test is on the LINEAR index of each corner (column-major, i + j * size_x) against the buffer
        lin = [c[0] + c[1] * self.size[0] for c in corners]
        if all(0 <= l <= n for l in lin) and all(0 <= c[0] < self.size[0] and 0 <= c[1] < self.size[1] for c in corners) and inside:
            # weights from the corner with the largest position: u, v in [0, 1] measured DOWN from it
            top = self._centre(lo)
            u = (mpf(top[0]) - mpf(p[0])) / mpf(self.res)
            v = (mpf(top[1]) - mpf(p[1])) / mpf(self.res)
            f = lambda di, dj: mpf(float(self.e[lo[0] + di, lo[1] + dj]))
            val = f(0, 0) * (1 - u) * (1 - v) + f(1, 0) * u * (1 - v) + f(0, 1) * (1 - u) * v + f(1, 1) * u * v
            return np.float32(float(val))
        if inside:
            return self.e[i0[0], i0[1]]               # INTER_NEAREST fall-back
        return None

    def height(self, x, y):
        v = self.at_position(x, y)
        return np.float32(FLT_MAX) if v is None else np.float32(v)

    def h_hx_hy(self, x, y):
        x, y = float(x), float(y)
        h = float(self.height(x, y))
        hx = float(np.float32(self.height(x + self.eps, y) - self.height(x - self.eps, y))) / (2 * self.eps)
        hy = float(np.float32(self.height(x, y + self.eps) - self.height(x, y - self.eps))) / (2 * self.eps)
        return mpf(h), mpf(hx), mpf(hy)


def terrain_h(tid, x, y):
    """height and slopes (height_map_examples.{h,cc}); x,y are mpf, comparisons as in the reference."""
    if isinstance(tid, GridMapTerrain):
        return tid.h_hx_hy(x, y)
    z = mpf(0)
    if tid == "flat":
        return z, z, z
    if tid == "block":
        s0, ln, ht, eps = mpf(0.7), mpf(3.5), mpf(0.5), mpf(0.03)
        slope = mpf(0.5 / 0.03)
        h = z
        hx = z
        if s0 <= x <= mpf(0.7 + 0.03):
            h = slope * (x - s0)
            hx = slope
        if mpf(0.7 + 0.03) <= x <= mpf(0.7 + 3.5):
            h = ht
        return h, hx, z
    if tid == "stairs":
        h = z
        if x >= mpf(1.0):
            h = mpf(0.2)
        if x >= mpf(1.0 + 0.4):
            h = mpf(0.4)
        if x >= mpf(1.0 + 0.4 + 1.0):
            h = z
        return h, z, z  # quirk 5: slope not overridden
    if tid == "gap":
        gs, w, hh = 1.0, 0.5, 1.5
        xc = gs + w / 2.0
        ge = gs + w
        a = (4 * hh) / (w * w)
        b = -(8 * hh * xc) / (w * w)
        c = -(hh * (w - 2 * xc) * (w + 2 * xc)) / (w * w)
        if mpf(gs) <= x <= mpf(ge):
            return mpf(a) * x * x + mpf(b) * x + mpf(c), 2 * mpf(a) * x + mpf(b), z
        return z, z, z
    if tid == "slope":
        ss, up, dn, hc = 1.0, 1.0, 1.0, 0.7
        xd = ss + up
        xf = xd + dn
        sl = hc / up
        h, hx = z, z
        if x >= mpf(ss):
            h, hx = mpf(sl) * (x - mpf(ss)), mpf(sl)
        if x >= mpf(xd):
            h, hx = mpf(hc) - mpf(sl) * (x - mpf(xd)), -mpf(sl)
        if x >= mpf(xf):
            h, hx = z, z
        return h, hx, z
    if tid == "chimney":
        if mpf(1.0) <= x <= mpf(1.0 + 1.5):
            return mpf(3.0) * (y - mpf(0.5)), z, mpf(3.0)
        return z, z, z
    if tid == "chimney_lr":
        h, hy = z, z
        if mpf(0.5) <= x <= mpf(0.5 + 1.0):
            h, hy = mpf(2) * (y - mpf(0.5)), mpf(2)
        if mpf(0.5 + 1.0) <= x <= mpf(0.5 + 2 * 1.0):
            h, hy = -mpf(2) * (y + mpf(0.5)), -mpf(2)
        return h, z, hy
    raise ValueError(tid)


def unit(v):
    n = mp.sqrt(sum(c * c for c in v))
    return [c / n for c in v]


HOT_PATH, TOWR_DEFAULT = 1 | 2 | 8 | 16, 63


def spline_acc_rows(L, name, x):
    """splineacc-<name> (spline_acc_constraint.cc:49-65): acceleration at the end of polynomial j minus
    acceleration at the start of polynomial j+1, from the cubic through the node values."""
    s = L.splines[name]
    out = []
    for j in range(len(s["durs"]) - 1):
        for dm in range(3):
            def node(k):
                return (x[s["nodes"][k][(0, dm)]], x[s["nodes"][k][(1, dm)]])
            Tp, Tn = mpf(s["durs"][j]), mpf(s["durs"][j + 1])
            a_prev = hermite(node(j), node(j + 1), Tp, Tp)[2]
            a_next = hermite(node(j + 1), node(j + 2), Tn, mpf(0))[2]
            out.append(a_prev - a_next)
    return out


def swing_rows(L, ee, x, t_swing_avg=0.3):
    """swing-ee-motion_<ee> (swing_constraint.cc:58-84): every swing node sits at the xy midpoint of its
    neighbours with xy velocity = distance / t_swing_avg (swing_constraint.h:68)."""
    s = L.splines["ee-motion_%d" % ee]
    out = []
    for nid in range(len(s["nodes"])):
        if s["const"][nid]:
            continue
        for dm in (0, 1):
            prev = x[s["nodes"][nid - 1][(0, dm)]]
            nxt = x[s["nodes"][nid + 1][(0, dm)]]
            out.append(x[s["nodes"][nid][(0, dm)]] - (prev + nxt) / 2)
            out.append(x[s["nodes"][nid][(1, dm)]] - (nxt - prev) / mpf(t_swing_avg))
    return out


def constraints(L, terrain, x, fn_max=1000.0, sets=HOT_PATH, xt=None):
    """g(x) stacked in reference order (parameters.cc:55-60): terrain-*, dynamic, splineacc-base-{lin,ang},
    rangeofmotion-*, force-*, swing-*; `sets` is the bit mask of oracle/towr_oracle.h.  `xt`: the point whose footholds
    the TERRAIN is sampled at (default x; the gridded terrain is differentiated with the terrain frozen, see generate)."""
    parts = _constraint_parts(L, terrain, x, fn_max, xt)
    g = []
    if sets & 1:
        g += parts["terrain"]
    if sets & 2:
        g += parts["dynamic"]
    if sets & 4:
        g += spline_acc_rows(L, "base-lin", x) + spline_acc_rows(L, "base-ang", x)
    if sets & 8:
        g += parts["rom"]
    if sets & 16:
        g += parts["force"]
    if sets & 32:
        for ee in range(L.n_ee):
            g += swing_rows(L, ee, x)
    if sets & 128:  # baseMotion (base_motion_constraint.cc:60-66): rows AX..AZ base-ang position, LX..LZ base-lin
        for t in L.grid_bm:
            g += spline_point(L, "base-ang", x, t)[0] + spline_point(L, "base-lin", x, t)[0]
    if sets & 64:  # totalduration-<ee> (total_duration_constraint.cc:50-56): sum of the optimised durations
        for sc in L.schedule:
            g.append(sum(x[sc["off"] + i] for i in range(sc["size"])))
    return g


def _constraint_parts(L, terrain, x, fn_max=1000.0, xt=None):
    rb = L.rb
    xt = x if xt is None else xt
    parts = {}
    g = []
    # terrain_constraint.cc:57-70 : one row per ee-motion node id>=1
    for ee in range(L.n_ee):
        s = L.splines["ee-motion_%d" % ee]
        for nid in range(1, len(s["nodes"])):
            p = [x[s["nodes"][nid][(0, dm)]] for dm in range(3)]
            g.append(p[2] - terrain_h(terrain, xt[s["nodes"][nid][(0, 0)]], xt[s["nodes"][nid][(0, 1)]])[0])
    parts["terrain"], g = g, []
    # dynamic_constraint.cc:59-64 + single_rigid_body_dynamics.cc:76-101
    Ixx, Iyy, Izz, Ixy, Ixz, Iyz = [mpf(v) for v in rb["I"]]
    Ib = [[Ixx, -Ixy, -Ixz], [-Ixy, Iyy, -Iyz], [-Ixz, -Iyz, Izz]]
    m = mpf(rb["m"])
    for t in L.grid_dyn:
        c = spline_point(L, "base-lin", x, t)
        e = spline_point(L, "base-ang", x, t)
        R = rot(e[0])
        w, wd = omega_and_dot(e[0], e[1], e[2])
        Rt = [[R[j][i] for j in range(3)] for i in range(3)]
        Iw = lambda v: matvec(R, matvec(Ib, matvec(Rt, v)))
        tau = [mpf(0)] * 3
        fsum = [mpf(0)] * 3
        for ee in range(L.n_ee):
            f = spline_point(L, "ee-force_%d" % ee, x, t)[0]
            p = spline_point(L, "ee-motion_%d" % ee, x, t)[0]
            tau = [a + b for a, b in zip(tau, cross(f, [c[0][i] - p[i] for i in range(3)]))]
            fsum = [a + b for a, b in zip(fsum, f)]
        ang = [a + b - d for a, b, d in zip(Iw(wd), cross(w, Iw(w)), tau)]
        lin = [m * c[2][i] - fsum[i] for i in range(3)]
        lin[2] += m * mpf(G)
        g += ang + lin
    parts["dynamic"], g = g, []
    # range_of_motion_constraint.cc:58-69
    for ee in range(L.n_ee):
        for t in L.grid_rom:
            c = spline_point(L, "base-lin", x, t)[0]
            e = spline_point(L, "base-ang", x, t)[0]
            p = spline_point(L, "ee-motion_%d" % ee, x, t)[0]
            R = rot(e)
            d = [p[i] - c[i] for i in range(3)]
            g += [sum(R[j][i] * d[j] for j in range(3)) for i in range(3)]
    parts["rom"], g = g, []
    # force_constraint.cc:62-89
    for ee in range(L.n_ee):
        sf = L.splines["ee-force_%d" % ee]
        sm = L.splines["ee-motion_%d" % ee]
        for nid in range(len(sf["nodes"])):
            if sf["const"][nid]:
                continue
            adj = 0 if nid == 0 else nid - 1  # first adjacent polynomial -> its phase
            phase = sf["polys"][adj][0]
            start = next(i for i, pl in enumerate(sm["polys"]) if pl[0] == phase)
            px, py = xt[sm["nodes"][start][(0, 0)]], xt[sm["nodes"][start][(0, 1)]]
            _, hx, hy = terrain_h(terrain, px, py)
            n = unit([-hx, -hy, mpf(1)])
            t1 = unit([mpf(1), mpf(0), hx])
            t2 = unit([mpf(0), mpf(1), hy])
            f = [x[sf["nodes"][nid][(0, dm)]] for dm in range(3)]
            dot = lambda a, b: sum(u * v for u, v in zip(a, b))
            mu = mpf(MU)
            g.append(dot(f, n))
            g.append(dot(f, [a - mu * b for a, b in zip(t1, n)]))
            g.append(dot(f, [a + mu * b for a, b in zip(t1, n)]))
            g.append(dot(f, [a - mu * b for a, b in zip(t2, n)]))
            g.append(dot(f, [a + mu * b for a, b in zip(t2, n)]))
    parts["force"] = g
    return parts


# ----------------------------------------------------------------- golden cases
def _gait(n_ee, combo, T):
    # input construction only: uses the oracle's gait table through ctypes
    from oracle import binding as ob
    pd, con = ob.gait(n_ee, combo, T)
    return [list(map(float, p)) for p in pd], con


def _sweep_candidate(index):
    # input construction only: candidate `index` of the C5 enumeration through the product's host-side gait
    # tables (towr_amd.sweep; no device involved).  The fixture stores the resulting phase durations explicitly.
    import towr_amd as ta
    from towr_amd import sweep
    combo, T, scale = sweep.enumerate_candidates(index + 1)[index]
    sched = ta.gait_combo(4, combo, T, scale)
    return [list(map(float, p)) for p in sched.durations()], [int(c) for c in sched.contact()]


def cases():
    hop = ([[0.4, 0.2, 0.4, 0.2, 0.4, 0.2, 0.2]], [1])
    return {
        "hopper_flat": dict(robot="monoped", terrain="flat", phases=hop, seed=11),
        "biped_walk_flat": dict(robot="biped", terrain="flat", phases=_gait(2, 0, 2.0), seed=12),
        "anymal_trot_gap": dict(robot="anymal", terrain="gap", phases=_gait(4, 1, 2.0), seed=13),
        "anymal_walk_stairs": dict(robot="anymal", terrain="stairs", phases=_gait(4, 0, 2.4), seed=14),
        "hyq_gallop_slope": dict(robot="hyq", terrain="slope", phases=_gait(4, 4, 2.2), seed=15),
        "go1_pace_chimney": dict(robot="go1", terrain="chimney", phases=_gait(4, 2, 1.8), seed=16),
        "hyq_bound_chimney_lr": dict(robot="hyq", terrain="chimney_lr", phases=_gait(4, 3, 2.0), seed=23),
        # towr's whole default constraint list (adds splineacc-base-* and swing-*)
        "full_biped_walk_block": dict(robot="biped", terrain="block", phases=_gait(2, 0, 2.0), seed=17, sets=TOWR_DEFAULT),
        "full_anymal_trot_gap": dict(robot="anymal", terrain="gap", phases=_gait(4, 1, 2.0), seed=18, sets=TOWR_DEFAULT),
        # optimised phase durations (Parameters::OptimizePhaseDurations): ee-schedule variables, PhaseSplines
        "timings_hopper_flat": dict(robot="monoped", terrain="flat", phases=hop, seed=19, sets=TOWR_DEFAULT | 64),
        "timings_biped_walk_stairs": dict(robot="biped", terrain="stairs", phases=_gait(2, 0, 2.0), seed=20, sets=TOWR_DEFAULT | 64),
        "timings_anymal_trot_gap": dict(robot="anymal", terrain="gap", phases=_gait(4, 1, 2.0), seed=21, sets=TOWR_DEFAULT | 64),
        # every Parameters::ConstraintName at once (adds baseMotion)
        "every_biped_run_slope": dict(robot="biped", terrain="slope", phases=_gait(2, 1, 1.6), seed=22, sets=255),
        # BASELINE sizes (dt = T / (K - 1.5), so that floor(T/dt) + 2 = K time nodes; time_discretization_constraint.cc:41-49):
        # C2 biped K = 100, C3 ANYmal trot K = 200, and candidate 30 of the C5 enumeration (walk, T = 1.4, swing scale
        # 0.864) on Stairs at K = 200.  These stress the accumulated-time / eps junction rule (spline.cc:51-60) at
        # dt = T / 198.5, where the default-discretisation fixtures above cannot.
        "c2_biped_walk_flat_k100": dict(robot="biped", terrain="flat", phases=_gait(2, 0, 2.0), seed=24, k_nodes=100),
        "c3_anymal_trot_flat_k200": dict(robot="anymal", terrain="flat", phases=_gait(4, 1, 2.0), seed=25, k_nodes=200),
        "c5_anymal_walk_stairs_k200": dict(robot="anymal", terrain="stairs", phases=_sweep_candidate(30), seed=26, k_nodes=200),
        # (round 5) BASELINE C4 on the one analytic terrain with curvature, at K = 200: candidate 60 of the enumeration
        # (walk, T = 1.6, swing scale 0.928) on the Gap -- the non-chain-rule quirk of height_map.cc:80-91 at BASELINE size
        "c4_anymal_gap_k200": dict(robot="anymal", terrain="gap", phases=_sweep_candidate(60), seed=27, k_nodes=200),
        # (round 5) fpowr's production formulation (fpowr/src/footstep_plan_server.cc:147-200): Go1, `Grid` height map over a
        # grid_map elevation layer, gait combo C1, TIME_HORIZON = 2 s (:31), default discretisation, towr's default
        # constraint list (parameters.cc:55-60)
        "fpowr_go1_grid_c1": dict(robot="go1", terrain="grid_map", phases=_gait(4, 1, 2.0), seed=28, sets=TOWR_DEFAULT,
                                  grid_map=step_patch(seed=28)),
    }


def step_patch(seed, size=(100, 100), res=0.04, pos=(1.5, 0.0)):
    """A perception-style elevation layer (float32 [size_x, size_y], grid_map cell order: x falls with i, y with j): three
    8-cm steps across x with edges that wander a little in y, a gentle cross slope and +-3 mm of seeded sensor noise."""
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(size[0]), np.arange(size[1]), indexing="ij")
    cx = pos[0] + 0.5 * size[0] * res - (i + 0.5) * res
    cy = pos[1] + 0.5 * size[1] * res - (j + 0.5) * res
    h = 0.02 * cy + 0.003 * rng.uniform(-1, 1, size=size)
    for k, edge in enumerate((0.9, 1.4, 1.9)):
        h = h + 0.08 * (cx > edge + 0.05 * np.sin(2.0 * cy + k))
    return h.astype(np.float32), res, pos


def make_x(L, seed):
    """Deterministic, well-spread evaluation point: footholds are pushed along +x so that
    several of them land on the non-flat part of every example terrain."""
    rng = np.random.default_rng(seed)
    x = rng.normal(size=L.n) * 0.3
    for ee in range(L.n_ee):
        s = L.splines["ee-motion_%d" % ee]
        for i, nd in enumerate(s["nodes"]):
            x[nd[(0, 0)]] = 0.4 + 2.2 * i / (len(s["nodes"]) - 1) + 0.05 * rng.normal()
        sf = L.splines["ee-force_%d" % ee]
        for nd in sf["nodes"]:
            for k, i in nd.items():
                x[i] = rng.normal() * 80.0 + (200.0 if k == (0, 2) else 0.0)
    if L.optimize_timings:  # phase durations: the given ones +-10 %, never at a grid time, sum < total time
        for sc in L.schedule:
            for i in range(sc["size"]):
                x[sc["off"] + i] = sc["initial"][i] * (1.0 + 0.1 * rng.uniform(-1, 1))
            assert x[sc["off"]:sc["off"] + sc["size"]].sum() < sc["t_total"] - 0.05
    return x


_CTX = {}


def _col(j):
    L, terrain, x, sets = _CTX["L"], _CTX["terrain"], _CTX["x"], _CTX["sets"]
    h = mpf(10) ** (-15)
    xp = list(x)
    xm = list(x)
    xp[j] += h
    xm[j] -= h
    # a gridded terrain is rounded to float (grid_height_map.h:37-45): g is not differentiable through it, and the
    # reference does not pretend it is -- its terrain entries are DEFINED (central float differences over eps, zeros for
    # the basis derivatives).  So the terrain stays frozen at x here and generate() adds the defined entries.
    xt = x if isinstance(terrain, GridMapTerrain) else None
    gp = constraints(L, terrain, xp, sets=sets, xt=xt)
    gm = constraints(L, terrain, xm, sets=sets, xt=xt)
    col = [(a - b) / (2 * h) for a, b in zip(gp, gm)]
    return j, [(i, float(v)) for i, v in enumerate(col) if abs(v) > mpf(10) ** (-25)]


def generate(name, spec, outdir, procs=8):
    pd, con = spec["phases"]
    dts = {}
    if "k_nodes" in spec:
        T = 0.0
        for d in pd[0]:
            T += d
        dts = dict(dt_dyn=T / (spec["k_nodes"] - 1.5), dt_rom=T / (spec["k_nodes"] - 1.5))
    L = Layout(spec["robot"], pd, con, optimize_timings=bool(spec.get("sets", HOT_PATH) & 64), **dts)
    if "k_nodes" in spec:
        assert len(L.grid_dyn) == spec["k_nodes"] and len(L.grid_rom) == spec["k_nodes"]
    x64 = make_x(L, spec["seed"])
    x = [mpf(float(v)) for v in x64]
    sets = spec.get("sets", HOT_PATH)
    terrain, extra = spec["terrain"], {}
    if terrain == "grid_map":
        el, res, pos = spec["grid_map"]
        terrain = GridMapTerrain(el, res, pos)
        extra = dict(grid_elevation=terrain.e, grid_resolution=np.float64(res), grid_position=np.array(pos, dtype=np.float64))
    _CTX.update(L=L, terrain=terrain, x=x, sets=sets)
    g = constraints(L, terrain, x, sets=sets)
    with Pool(procs) as pool:
        cols = pool.map(_col, range(L.n), chunksize=4)
    rows, cidx, vals = [], [], []
    for j, ent in cols:
        for i, v in ent:
            rows.append(i)
            cidx.append(j)
            vals.append(v)
    if isinstance(terrain, GridMapTerrain):
        # terrain_constraint.cc:97-103: row of node id >= 1 carries -dh/dx, -dh/dy in the node's x / y columns, with the
        # reference's dh = Grid::GetHeightDerivWrtX/Y.  (The force rows' foothold columns are zeros in the reference:
        # nothing to add, the test reads a missing entry as 0.)
        assert sets & 1
        row = 0
        for ee in range(L.n_ee):
            s = L.splines["ee-motion_%d" % ee]
            for nid in range(1, len(s["nodes"])):
                cx, cy = s["nodes"][nid][(0, 0)], s["nodes"][nid][(0, 1)]
                for q in (x64[cx] - terrain.eps, x64[cx] + terrain.eps):   # every sample well inside the map
                    assert abs(q - terrain.pos[0]) < 0.5 * terrain.length[0] - 2 * terrain.res
                for q in (x64[cy] - terrain.eps, x64[cy] + terrain.eps):
                    assert abs(q - terrain.pos[1]) < 0.5 * terrain.length[1] - 2 * terrain.res
                _, hx, hy = terrain.h_hx_hy(x64[cx], x64[cy])
                for c, v in ((cx, -float(hx)), (cy, -float(hy))):
                    rows.append(row)
                    cidx.append(c)
                    vals.append(v)
                row += 1
    np.savez_compressed(
        os.path.join(outdir, "mp_%s.npz" % name), robot=spec["robot"], terrain=spec["terrain"],
        n_phases=np.array([len(p) for p in pd]), phase_durations=np.concatenate([np.array(p) for p in pd]),
        contact_at_start=np.array(con), constraint_sets=np.int32(sets), x=x64, g=np.array([float(v) for v in g]),
        jac_row=np.array(rows, dtype=np.int32), jac_col=np.array(cidx, dtype=np.int32), jac_val=np.array(vals),
        dt_dynamic=np.float64(dts.get("dt_dyn", 0.1)), dt_rom=np.float64(dts.get("dt_rom", 0.08)), **extra)
    print(name, "n", L.n, "m", len(g), "nnz(true)", len(vals), flush=True)


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[1:]
    for name, spec in cases().items():
        if only and name not in only:
            continue
        generate(name, spec, out, procs=int(os.environ.get("MP_REF_PROCS", "8")))
