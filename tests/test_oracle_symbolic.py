"""Pins the closed forms that BOTH the oracle and the mpmath twin inherit from the reference against the only ground
truth the reference itself holds: the authors' symbolic derivation records in towr/matlab/ (SURVEY.md section 8c,
item 2).  The recipes are restated here in sympy (not the .m text) and re-derived from scratch:

  * cubic_hermite_polynomial.m:18-76 -- a cubic p(t) = a + b t + c t^2 + d t^3 whose coefficients solve
    p(0) = p0, p'(0) = v0, p(T) = p1, p'(T) = v1; Jacobians of position, velocity and acceleration w.r.t.
    (p0, v0, p1, v1, T)                       -> polynomial.cc:140-257 (12 weights + GetDerivativeOfPosWrtDuration)
  * euler_converter.m:11-42 -- the matrix M that maps ZYX Euler rates to angular velocity (kindr convention), its
    time derivative, the rotation matrix R, and their derivatives w.r.t. a polynomial coefficient u on which the
    angles depend                              -> euler_converter.cc:133-304
  * gap_height_map.m:12-33 -- a parabola through (xc, -h) and (xc +- w/2, 0) -> height_map_examples.h:105-111

and compared numerically with the oracle at random points.  (The reference's M has the columns in roll, pitch, yaw order,
the .m file in yaw, pitch, roll order: the same map, columns permuted.)"""
import numpy as np
import pytest
import sympy as sp

from oracle import binding as ob


def _hermite():
    a, b, c, d, p0, v0, p1, v1, t, T = sp.symbols("a b c d p0 v0 p1 v1 t T")
    pos = d * t**3 + c * t**2 + b * t + a
    vel = sp.diff(pos, t)
    sol = sp.solve([pos.subs(t, 0) - p0, vel.subs(t, 0) - v0, pos.subs(t, T) - p1, vel.subs(t, T) - v1], [a, b, c, d], dict=True)[0]
    pos = pos.subs(sol)
    derivs = [pos, sp.diff(pos, t), sp.diff(pos, t, 2)]
    jac = [[sp.diff(f, u) for u in (p0, v0, p1, v1)] for f in derivs]
    dpos_dT = sp.diff(pos, T)
    args = (t, T, p0, v0, p1, v1)
    return sp.lambdify(args, jac, "numpy"), sp.lambdify(args, dpos_dT, "numpy")


def test_hermite_weights_and_duration_derivative_match_the_symbolic_derivation():
    jac, dpos_dT = _hermite()
    rng = np.random.default_rng(0)
    for _ in range(200):
        T = rng.uniform(0.05, 1.5)
        t = rng.uniform(0.0, T)
        p0, v0, p1, v1 = rng.normal(size=4) * [1.0, 3.0, 1.0, 3.0]
        want = np.array(jac(t, T, p0, v0, p1, v1), dtype=float)
        got = ob.hermite_weights(t, T)                       # d{p,v,a}/d{p0,v0,p1,v1}, polynomial.cc:140-234
        scale = np.abs(want).max(axis=1, keepdims=True)
        assert np.all(np.abs(got - want) <= 1e-11 * scale + 1e-13), (t, T, got, want)
        w = float(dpos_dT(t, T, p0, v0, p1, v1))              # polynomial.cc:236-257
        g = ob.hermite_dpos_dT(t, T, p0, v0, p1, v1)
        assert abs(g - w) <= 1e-10 * max(1.0, abs(w), abs(v0), abs(v1)), (t, T, g, w)


def _euler():
    """x, y, z (roll, pitch, yaw) each a cubic Hermite polynomial of its own four node values; u runs over all 12."""
    t, T = sp.symbols("t T")
    nodes = sp.symbols("n0:12")                     # NodesVariablesAll order: node0 {px py pz vx vy vz}, node1 {...}
    w = [2 * t**3 / T**3 - 3 * t**2 / T**2 + 1, t - 2 * t**2 / T + t**3 / T**2, 3 * t**2 / T**2 - 2 * t**3 / T**3, t**3 / T**2 - t**2 / T]
    ang = [w[0] * nodes[d] + w[1] * nodes[3 + d] + w[2] * nodes[6 + d] + w[3] * nodes[9 + d] for d in range(3)]
    x, y, z = ang
    # euler_converter.m:14-16 (columns yaw, pitch, roll) re-ordered to the reference's roll, pitch, yaw columns
    M = sp.Matrix([[sp.cos(y) * sp.cos(z), -sp.sin(z), 0], [sp.cos(y) * sp.sin(z), sp.cos(z), 0], [-sp.sin(y), 0, 1]])
    Md = sp.diff(M, t)
    R = sp.Matrix([[sp.cos(y) * sp.cos(z), sp.cos(z) * sp.sin(x) * sp.sin(y) - sp.cos(x) * sp.sin(z), sp.sin(x) * sp.sin(z) + sp.cos(x) * sp.cos(z) * sp.sin(y)],
                   [sp.cos(y) * sp.sin(z), sp.cos(x) * sp.cos(z) + sp.sin(x) * sp.sin(y) * sp.sin(z), sp.cos(x) * sp.sin(y) * sp.sin(z) - sp.cos(z) * sp.sin(x)],
                   [-sp.sin(y), sp.cos(y) * sp.sin(x), sp.cos(x) * sp.cos(y)]])
    rate = sp.Matrix([sp.diff(a, t) for a in ang])
    omega = M * rate
    omega_dot = sp.diff(omega, t)
    out = dict(M=M, Mdot=Md, R=R, omega=omega, omega_dot=omega_dot,
               dM=[[[sp.diff(M[r, c], u) for u in nodes] for c in range(3)] for r in range(3)],
               dMdot=[[[sp.diff(Md[r, c], u) for u in nodes] for c in range(3)] for r in range(3)],
               dR=[[[sp.diff(R[r, c], u) for u in nodes] for c in range(3)] for r in range(3)],
               domega=[[sp.diff(omega[r], u) for u in nodes] for r in range(3)],
               domega_dot=[[sp.diff(omega_dot[r], u) for u in nodes] for r in range(3)])
    return {k: sp.lambdify((t, T) + nodes, v, "numpy") for k, v in out.items()}


def test_euler_converter_matches_the_symbolic_derivation():
    f = _euler()
    rng = np.random.default_rng(1)
    for _ in range(40):
        T = rng.uniform(0.05, 0.8)
        t = rng.uniform(0.0, T)
        nodes = np.concatenate([rng.uniform(-1.2, 1.2, 3), rng.normal(size=3) * 2, rng.uniform(-1.2, 1.2, 3), rng.normal(size=3) * 2])
        got = ob.euler_probe(nodes, T, t)
        for key in ("M", "Mdot", "R", "omega", "omega_dot", "dM", "dMdot", "dR", "domega", "domega_dot"):
            want = np.array(f[key](t, T, *nodes), dtype=float).reshape(got[key].shape)
            scale = max(1.0, np.abs(want).max())
            assert np.abs(got[key] - want).max() <= 1e-10 * scale, (key, np.abs(got[key] - want).max(), scale)


def test_gap_parabola_matches_the_symbolic_derivation():
    x, a, b, c, h, w, xc = sp.symbols("x a b c h w xc")
    z = a * x**2 + b * x + c
    # gap_height_map.m:16-28: centre at -h, both edges at 0; height_map_examples.h:105-111 codes the solution
    # a = 4h/w^2, b = -8 h xc/w^2, c = -h (w - 2 xc)(w + 2 xc)/w^2 with h = 1.5, w = 0.5, gap start 1.0
    sol = sp.solve([z.subs(x, xc) + h, z.subs(x, xc - w / 2), z.subs(x, xc + w / 2)], [a, b, c], dict=True)[0]
    zf = sp.lambdify((x, h, w, xc), z.subs(sol), "numpy")
    dz = sp.lambdify((x, h, w, xc), sp.diff(z.subs(sol), x), "numpy")
    hh, ww, start = 1.5, 0.5, 1.0
    xs = np.linspace(start, start + ww, 41)
    got = np.array([ob.terrain_height("gap", xx, 0.0) for xx in xs])
    want = zf(xs, hh, ww, start + ww / 2)
    assert sp.simplify(sol[a] - 4 * h / w**2) == 0 and sp.simplify(sol[b] + 8 * h * xc / w**2) == 0
    assert sp.simplify(sol[c] + h * (w - 2 * xc) * (w + 2 * xc) / w**2) == 0
    assert np.abs(got - want).max() <= 1e-12 * hh
    assert ob.terrain_height("gap", start - 1e-9, 0.0) == 0.0 and ob.terrain_height("gap", start + ww + 1e-9, 0.0) == 0.0
    P = ob.OracleProblem("monoped", "gap", [[0.4, 0.2, 0.4]], [1])
    for xx in xs[1:-1]:
        assert abs(P.terrain_probe(xx, 0.0)[1] - dz(xx, hh, ww, start + ww / 2)) <= 1e-11 * hh / ww
