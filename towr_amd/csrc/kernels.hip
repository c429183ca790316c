// HIP kernels (gfx950 / CDNA4) for towr's NLP constraint + Jacobian callback.
//
// One fused launch evaluates every constraint set of every problem of the batch.
// A workgroup is one wavefront (64 lanes) and owns one *contiguous* slice of one
// problem's CSR value array:
//   kind 0  dynamic           : lanes = consecutive time nodes k       (6 rows each)
//   kind 1  rangeofmotion-ee  : lanes = consecutive time nodes k       (3 rows each)
//   kind 2  force-* / terrain-* sets of the problem: lanes = spline nodes
// Every lane computes its rows in registers (FP64, no MFMA: the work is 3x3 algebra),
// scatters the values into an LDS image of the slice at the CSR position they have in
// global memory, and the wave then streams the image out with 16-byte coalesced stores.
// x-independent index work (active polynomial, local time, node->column maps, CSR
// offsets) comes from the per-structure tables of device_tables.h.
//
// Math follows the reference line by line in meaning (citations per function); the
// arithmetic is re-associated (Hermite basis form, factored base-ang tile) and agrees
// with the reference formulas to rounding, tests/ pin that at <= 1e-9.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"

namespace twr {

#define TWR_DEV __device__ __forceinline__

template <typename T>
TWR_DEV const T* tbl(const char* blob, uint32_t off) {
  return reinterpret_cast<const T*>(blob + off);
}

// ---------------------------------------------------------------- cubic Hermite weights
// d{pos,vel,acc}/d{p0,v0,p1,v1} of CubicHermitePolynomial (src/polynomial.cc:140-234); iT = 1/T comes
// from the tables (one IEEE division on the host instead of one per lane and spline).
TWR_DEV void hermite_pos(double t, double iT, double w[4]) {
  const double iT2 = iT * iT, iT3 = iT2 * iT;
  const double t2 = t * t, t3 = t2 * t;
  w[0] = 2.0 * t3 * iT3 - 3.0 * t2 * iT2 + 1.0;
  w[1] = t - 2.0 * t2 * iT + t3 * iT2;
  w[2] = 3.0 * t2 * iT2 - 2.0 * t3 * iT3;
  w[3] = t3 * iT2 - t2 * iT;
}
TWR_DEV void hermite_all(double t, double iT, double wp[4], double wv[4], double wa[4]) {
  const double iT2 = iT * iT, iT3 = iT2 * iT;
  const double t2 = t * t, t3 = t2 * t;
  wp[0] = 2.0 * t3 * iT3 - 3.0 * t2 * iT2 + 1.0;
  wp[1] = t - 2.0 * t2 * iT + t3 * iT2;
  wp[2] = 3.0 * t2 * iT2 - 2.0 * t3 * iT3;
  wp[3] = t3 * iT2 - t2 * iT;
  wv[0] = 6.0 * t2 * iT3 - 6.0 * t * iT2;
  wv[1] = 3.0 * t2 * iT2 - 4.0 * t * iT + 1.0;
  wv[2] = 6.0 * t * iT2 - 6.0 * t2 * iT3;
  wv[3] = 3.0 * t2 * iT2 - 2.0 * t * iT;
  wa[0] = 12.0 * t * iT3 - 6.0 * iT2;
  wa[1] = 6.0 * t * iT2 - 4.0 * iT;
  wa[2] = 6.0 * iT2 - 12.0 * t * iT3;
  wa[3] = 6.0 * t * iT2 - 2.0 * iT;
}

// ---------------------------------------------------------------- ee polynomial record
struct PolyR {
  double iT;
  int xbase;
  uint32_t meta;   // nslots | cnt[0]<<8 | cnt[1]<<16 | cnt[2]<<24
  uint32_t c[6];   // 12 candidate descriptors (16 bit each)
  uint32_t shared;
  TWR_DEV int nslots() const { return meta & 0xFF; }
  TWR_DEV int cnt(int d) const { return (meta >> (8 * (d + 1))) & 0xFF; }
  TWR_DEV uint32_t cand(int i) const { return (c[i >> 1] >> (16 * (i & 1))) & 0xFFFFu; }
};
TWR_DEV int meta_nslots(uint32_t meta) { return meta & 0xFF; }
TWR_DEV int meta_cnt(uint32_t meta, int d) { return (meta >> (8 * (d + 1))) & 0xFF; }
TWR_DEV PolyR load_poly(const EePoly* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], d = q[2];
  PolyR r;
  r.iT = __hiloint2double((int)a.y, (int)a.x);
  r.xbase = (int)a.z;
  r.meta = a.w;
  r.c[0] = b.x; r.c[1] = b.y; r.c[2] = b.z; r.c[3] = b.w;
  r.c[4] = d.x; r.c[5] = d.y;
  r.shared = d.z & 0xFF;
  return r;
}
// Position of an ee spline at the lane's time: sum over the node values that are variables
// (Spline::GetPoint, src/spline.cc:80-93, in Hermite basis form).  A stance ee-motion
// polynomial keeps one shared position variable for both nodes: w_p1 is folded into w_p0.
TWR_DEV void ee_weights_and_point(const PolyR& P, double tl, const double* __restrict__ xp, double w[4], double out[3]) {
  hermite_pos(tl, P.iT, w);
  if (P.shared) w[0] += w[2];
  // all 12 loads are issued unconditionally (absent candidates read slot 0 and get weight 0): one
  // memory round trip instead of twelve dependent, branch-guarded ones
  double v[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const int sl = P.cand(c) & 0xF;
    v[c] = xp[P.xbase + (sl != 0xF ? sl : 0)];
  }
  out[0] = out[1] = out[2] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool valid = (P.cand(j * 3 + d) & 0xF) != 0xF;
      out[d] = fma(valid ? w[j] : 0.0, v[j * 3 + d], out[d]);
    }
}

// entry (r,d), r != d, of the cross-product matrix [v]x (single_rigid_body_dynamics.cc:46-57)
template <int R, int D>
TWR_DEV double crs(const double v[3]) {
  static_assert(R != D, "diagonal of a cross matrix is structurally absent");
  constexpr int o = 3 - R - D;
  constexpr bool pos = (R == 0 && D == 2) || (R == 1 && D == 0) || (R == 2 && D == 1);
  return pos ? v[o] : -v[o];
}
TWR_DEV void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// ZYX Euler rotation and its partial derivatives w.r.t. roll/pitch/yaw
// (euler_converter.cc:207-221 and the cell-wise derivatives of :241-268).
struct Rot {
  double R[3][3], Rx[3][3], Ry[3][3], Rz[3][3];
  double sx, cx, sy, cy, sz, cz;
};
TWR_DEV void rotation(const double e[3], Rot& o) {
  double sx, cx, sy, cy, sz, cz;
  sincos(e[0], &sx, &cx);
  sincos(e[1], &sy, &cy);
  sincos(e[2], &sz, &cz);
  o.sx = sx; o.cx = cx; o.sy = sy; o.cy = cy; o.sz = sz; o.cz = cz;
  o.R[0][0] = cy * cz; o.R[0][1] = cz * sx * sy - cx * sz; o.R[0][2] = sx * sz + cx * cz * sy;
  o.R[1][0] = cy * sz; o.R[1][1] = cx * cz + sx * sy * sz; o.R[1][2] = cx * sy * sz - cz * sx;
  o.R[2][0] = -sy;     o.R[2][1] = cy * sx;                o.R[2][2] = cx * cy;
  o.Rx[0][0] = 0.0; o.Rx[0][1] = cx * cz * sy + sx * sz;  o.Rx[0][2] = cx * sz - cz * sx * sy;
  o.Rx[1][0] = 0.0; o.Rx[1][1] = cx * sy * sz - cz * sx;  o.Rx[1][2] = -sx * sy * sz - cx * cz;
  o.Rx[2][0] = 0.0; o.Rx[2][1] = cx * cy;                 o.Rx[2][2] = -cy * sx;
  o.Ry[0][0] = -cz * sy; o.Ry[0][1] = cy * cz * sx; o.Ry[0][2] = cx * cy * cz;
  o.Ry[1][0] = -sy * sz; o.Ry[1][1] = cy * sx * sz; o.Ry[1][2] = cx * cy * sz;
  o.Ry[2][0] = -cy;      o.Ry[2][1] = -sx * sy;     o.Ry[2][2] = -cx * sy;
  o.Rz[0][0] = -cy * sz; o.Rz[0][1] = -sx * sy * sz - cx * cz; o.Rz[0][2] = cz * sx - cx * sy * sz;
  o.Rz[1][0] = cy * cz;  o.Rz[1][1] = cz * sx * sy - cx * sz;  o.Rz[1][2] = cx * cz * sy + sx * sz;
  o.Rz[2][0] = 0.0;      o.Rz[2][1] = 0.0;                     o.Rz[2][2] = 0.0;
}
TWR_DEV void matvec(const double A[3][3], const double v[3], double o[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2];
}
TWR_DEV void matTvec(const double A[3][3], const double v[3], double o[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[0][i] * v[0] + A[1][i] * v[1] + A[2][i] * v[2];
}
TWR_DEV void symmul(const double I[6], const double v[3], double o[3]) {  // I = (00,01,02,11,12,22)
  o[0] = I[0] * v[0] + I[1] * v[1] + I[2] * v[2];
  o[1] = I[1] * v[0] + I[3] * v[1] + I[4] * v[2];
  o[2] = I[2] * v[0] + I[4] * v[1] + I[5] * v[2];
}

// ---------------------------------------------------------------- dynamic (SRBD) item
// DynamicConstraint::{UpdateModel, UpdateConstraintAtInstance, UpdateJacobianAtInstance}
// (dynamic_constraint.cc:59-137) with SingleRigidBodyDynamics::{GetDynamicViolation,
// GetJacobianWrt{BaseLin,BaseAng,Force,EEPos}} (single_rigid_body_dynamics.cc:76-192) and the
// EulerConverter derivatives (euler_converter.cc:85-131,168-198,223-304) for one time node.
// The base-ang block is evaluated in factored form: with u = (p0,v0,p1,v1) of Euler dim d,
//   d g_ang / d u_j = A_d wP[j] + B_d wV[j] + C_d wA[j],
//   A_d = d g_ang/d e_d, B_d = d g_ang/d edot_d, C_d = d g_ang/d eddot_d  (3-vectors).
template <int NEE>
TWR_DEV void dyn_item(const DevStruct* __restrict__ S, const char* __restrict__ blob, const double* __restrict__ xp,
                      double* __restrict__ gp, double* __restrict__ stage, int soff, int trash, int k, bool want_g,
                      bool want_j) {
  const double tb = tbl<double>(blob, S->o_dyn_tl_base)[k];
  const int q = tbl<int32_t>(blob, S->o_dyn_base_poly)[k];
  const double iTb = tbl<double>(blob, S->o_base_iT)[q];
  double wP[4], wV[4], wA[4];
  hermite_all(tb, iTb, wP, wV, wA);
  const double* xl = xp + S->off_base_lin + 6 * q;  // [p0 v0 p1 v1] x 3, NodesVariablesAll order
  const double* xa = xp + S->off_base_ang + 6 * q;
  double c[3], cdd[3], e[3], ed[3], edd[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double l0 = xl[d], l1 = xl[3 + d], l2 = xl[6 + d], l3 = xl[9 + d];
    const double a0 = xa[d], a1 = xa[3 + d], a2 = xa[6 + d], a3 = xa[9 + d];
    c[d] = wP[0] * l0 + wP[1] * l1 + wP[2] * l2 + wP[3] * l3;
    cdd[d] = wA[0] * l0 + wA[1] * l1 + wA[2] * l2 + wA[3] * l3;
    e[d] = wP[0] * a0 + wP[1] * a1 + wP[2] * a2 + wP[3] * a3;
    ed[d] = wV[0] * a0 + wV[1] * a1 + wV[2] * a2 + wV[3] * a3;
    edd[d] = wA[0] * a0 + wA[1] * a1 + wA[2] * a2 + wA[3] * a3;
  }

  // --- pass A over the end-effectors: spline points, force / torque sums, slot counts.
  // Only f_i, r_i = c - p_i and the two count words per ee stay live; the polynomial records and
  // weights are re-read (L1 hits) in pass B, which keeps the lane under 256 VGPRs.
  double f[NEE][3], rv[NEE][3];
  uint32_t mmeta[NEE], fmeta[NEE];
  double F[3] = {0.0, 0.0, 0.0}, tau[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int ee = 0; ee < NEE; ++ee) {
    const int mq = tbl<int32_t>(blob, S->o_dyn_mpoly[ee])[k];
    const int fq = tbl<int32_t>(blob, S->o_dyn_fpoly[ee])[k];
    const double tm = tbl<double>(blob, S->o_dyn_tl_m[ee])[k];
    const double tf = tbl<double>(blob, S->o_dyn_tl_f[ee])[k];
    const PolyR MP = load_poly(tbl<EePoly>(blob, S->o_mpoly[ee]) + mq);
    const PolyR FP = load_poly(tbl<EePoly>(blob, S->o_fpoly[ee]) + fq);
    double wm[4], wf[4], p[3];
    ee_weights_and_point(MP, tm, xp, wm, p);
    ee_weights_and_point(FP, tf, xp, wf, f[ee]);
    mmeta[ee] = MP.meta;
    fmeta[ee] = FP.meta;
    double t3[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) rv[ee][d] = c[d] - p[d];
    cross3(f[ee], rv[ee], t3);  // f x (c - p), single_rigid_body_dynamics.cc:84-88
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      tau[d] += t3[d];
      F[d] += f[ee][d];
    }
  }

  // --- angular quantities (euler_converter.cc:58-83,133-166)
  Rot ro;
  rotation(e, ro);
  const double sy = ro.sy, cy = ro.cy, sz = ro.sz, cz = ro.cz;
  const double xd = ed[0], yd = ed[1], zd = ed[2];
  const double Mx[3] = {cy * cz, cy * sz, -sy}, My[3] = {-sz, cz, 0.0};
  const double Mdx[3] = {-cz * sy * yd - cy * sz * zd, cy * cz * zd - sy * sz * yd, -cy * yd};
  const double Mdy[3] = {-cz * zd, -sz * zd, 0.0};
  double om[3], omd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    om[i] = Mx[i] * xd + My[i] * yd;
    omd[i] = Mdx[i] * xd + Mdy[i] * yd + Mx[i] * edd[0] + My[i] * edd[1];
  }
  om[2] += zd;
  omd[2] += edd[2];
  double Ib[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) Ib[i] = S->Ib[i];
  // I_w v = R I_b R^T v  (single_rigid_body_dynamics.cc:91)
  auto Iw = [&](const double v[3], double o[3]) {
    double a[3], b[3];
    matTvec(ro.R, v, a);
    symmul(Ib, a, b);
    matvec(ro.R, b, o);
  };
  double Iw_wd[3], Iw_w[3];
  Iw(omd, Iw_wd);
  Iw(om, Iw_w);
  const double m = S->mass;

  if (want_g) {  // GetDynamicViolation, single_rigid_body_dynamics.cc:76-101
    double wxIw[3];
    cross3(om, Iw_w, wxIw);
    double* go = gp + S->row_dyn + 6 * k;
#pragma unroll
    for (int i = 0; i < 3; ++i) go[i] = Iw_wd[i] + wxIw[i] - tau[i];
    go[3] = m * cdd[0] - F[0];
    go[4] = m * cdd[1] - F[1];
    go[5] = m * cdd[2] - F[2] + m * S->gravity;
  }
  if (!want_j) return;

  // --- row layout of this time node inside the CSR slice
  int nma[3] = {0, 0, 0}, nfa[3] = {0, 0, 0}, nfl[3] = {0, 0, 0};
#pragma unroll
  for (int ee = 0; ee < NEE; ++ee)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      nma[r] += meta_nslots(mmeta[ee]) - meta_cnt(mmeta[ee], r);
      nfa[r] += meta_nslots(fmeta[ee]) - meta_cnt(fmeta[ee], r);
      nfl[r] += meta_cnt(fmeta[ee], r);
    }
  int rs[3], rl[3];
  rs[0] = soff;
  rs[1] = rs[0] + 20 + nma[0] + nfa[0];
  rs[2] = rs[1] + 20 + nma[1] + nfa[1];
  rl[0] = rs[2] + 20 + nma[2] + nfa[2];
  rl[1] = rl[0] + 4 + nfl[0];
  rl[2] = rl[1] + 4 + nfl[1];

  // --- base-lin block: ang rows -sum_i [f_i]x J_pos, lin rows m J_acc (:103-121)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    stage[rs[0] + 2 * j + 0] = -crs<0, 1>(F) * wP[j];
    stage[rs[0] + 2 * j + 1] = -crs<0, 2>(F) * wP[j];
    stage[rs[1] + 2 * j + 0] = -crs<1, 0>(F) * wP[j];
    stage[rs[1] + 2 * j + 1] = -crs<1, 2>(F) * wP[j];
    stage[rs[2] + 2 * j + 0] = -crs<2, 0>(F) * wP[j];
    stage[rs[2] + 2 * j + 1] = -crs<2, 1>(F) * wP[j];
#pragma unroll
    for (int d = 0; d < 3; ++d) stage[rl[d] + j] = m * wA[j];
  }

  // --- base-ang block (:123-165), factored
  {
    double RtWd[3], RtW[3], aWd[3], aW[3];
    matTvec(ro.R, omd, RtWd);
    matTvec(ro.R, om, RtW);
    symmul(Ib, RtWd, aWd);  // I_b R^T omega_dot  (v11)
    symmul(Ib, RtW, aW);    // I_b R^T omega      (v21)
    // d(I_w v)/d e_d = R_d I_b R^T v + R I_b R_d^T v   (jac11+jac12 / jac21+jac22)
    auto dIw = [&](const double Rd[3][3], const double v[3], const double av[3], double o[3]) {
      double t1[3], t2[3], t3[3], t4[3];
      matvec(Rd, av, t1);
      matTvec(Rd, v, t2);
      symmul(Ib, t2, t3);
      matvec(ro.R, t3, t4);
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = t1[i] + t4[i];
    };
    // columns of M and the partials of omega, omega_dot (euler_converter.cc:168-198,270-304)
    const double Mz[3] = {0.0, 0.0, 1.0};
    const double dMx_dy[3] = {-sy * cz, -sy * sz, -cy};
    const double dMx_dz[3] = {-cy * sz, cy * cz, 0.0};
    const double dMy_dz[3] = {-cz, -sz, 0.0};
    const double dMdx_dy[3] = {-cz * cy * yd + sy * sz * zd, -sy * cz * zd - cy * sz * yd, sy * yd};
    const double dMdx_dz[3] = {sz * sy * yd - cy * cz * zd, -cy * sz * zd - sy * cz * yd, 0.0};
    const double dMdy_dz[3] = {sz * zd, -cz * zd, 0.0};
    double A[3][3], B[3][3], C[3][3];  // [euler dim][row]
    Iw(Mx, C[0]);
    Iw(My, C[1]);
    Iw(Mz, C[2]);
    // B_d = I_w d(omega_dot)/d(edot_d) + M_d x (I_w omega) + omega x (I_w M_d)
    {
      double dwd[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        dwd[0][i] = Mdx[i];
        dwd[1][i] = Mdy[i] + xd * dMx_dy[i];
        dwd[2][i] = xd * dMx_dz[i] + yd * dMy_dz[i];
      }
      const double Mc[3][3] = {{Mx[0], Mx[1], Mx[2]}, {My[0], My[1], My[2]}, {Mz[0], Mz[1], Mz[2]}};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double t1[3], t2[3], t3[3];
        Iw(dwd[d], t1);
        cross3(Mc[d], Iw_w, t2);
        cross3(om, C[d], t3);
#pragma unroll
        for (int i = 0; i < 3; ++i) B[d][i] = t1[i] + t2[i] + t3[i];
      }
    }
    // A_d = dI_w(omega_dot) + I_w d(omega_dot) + d(omega) x I_w omega + omega x (dI_w(omega) + I_w d(omega))
    {
      double dw[3][3], dwd[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        dw[0][i] = 0.0;
        dwd[0][i] = 0.0;
        dw[1][i] = xd * dMx_dy[i];
        dwd[1][i] = xd * dMdx_dy[i] + edd[0] * dMx_dy[i];
        dw[2][i] = xd * dMx_dz[i] + yd * dMy_dz[i];
        dwd[2][i] = xd * dMdx_dz[i] + yd * dMdy_dz[i] + edd[0] * dMx_dz[i] + edd[1] * dMy_dz[i];
      }
      {  // roll: omega, omega_dot do not depend on it
        double t1[3], t2[3], t3[3];
        dIw(ro.Rx, omd, aWd, t1);
        dIw(ro.Rx, om, aW, t2);
        cross3(om, t2, t3);
#pragma unroll
        for (int i = 0; i < 3; ++i) A[0][i] = t1[i] + t3[i];
      }
#pragma unroll
      for (int d = 1; d < 3; ++d) {
        double t1[3], t2[3], t3[3], t4[3], t5[3], t6[3], t7[3];
        if (d == 1) {
          dIw(ro.Ry, omd, aWd, t1);
          dIw(ro.Ry, om, aW, t4);
        } else {
          dIw(ro.Rz, omd, aWd, t1);
          dIw(ro.Rz, om, aW, t4);
        }
        Iw(dwd[d], t2);
        cross3(dw[d], Iw_w, t3);
        Iw(dw[d], t5);
#pragma unroll
        for (int i = 0; i < 3; ++i) t6[i] = t4[i] + t5[i];
        cross3(om, t6, t7);
#pragma unroll
        for (int i = 0; i < 3; ++i) A[d][i] = t1[i] + t2[i] + t3[i] + t7[i];
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          stage[rs[r] + 8 + 3 * j + d] = A[d][r] * wP[j] + B[d][r] * wV[j] + C[d][r] * wA[j];
  }

  // --- pass B: ee-motion blocks [f]x J_p (:181-192) and ee-force blocks {[r]x J_f ; -J_f} (:167-179).
  // Candidates that are not variables write to the lane's trash slot instead of branching.
  int ms[3] = {rs[0] + 20, rs[1] + 20, rs[2] + 20};
  int fs[3] = {ms[0] + nma[0], ms[1] + nma[1], ms[2] + nma[2]};
  int ls[3] = {rl[0] + 4, rl[1] + 4, rl[2] + 4};
#pragma unroll
  for (int ee = 0; ee < NEE; ++ee) {
    const int mq = tbl<int32_t>(blob, S->o_dyn_mpoly[ee])[k];
    const int fq = tbl<int32_t>(blob, S->o_dyn_fpoly[ee])[k];
    const double tm = tbl<double>(blob, S->o_dyn_tl_m[ee])[k];
    const double tf = tbl<double>(blob, S->o_dyn_tl_f[ee])[k];
    const PolyR MP = load_poly(tbl<EePoly>(blob, S->o_mpoly[ee]) + mq);
    const PolyR FP = load_poly(tbl<EePoly>(blob, S->o_fpoly[ee]) + fq);
    double wm[4], wf[4];
    hermite_pos(tm, MP.iT, wm);
    if (MP.shared) wm[0] += wm[2];
    hermite_pos(tf, FP.iT, wf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#define TWR_EE_TILE(D, R1, R2)                                                             \
  {                                                                                        \
    const uint32_t cm = MP.cand(j * 3 + D);                                                \
    const bool vm = (cm & 0xF) != 0xF;                                                     \
    stage[vm ? ms[R1] + (int)((cm >> 4) & 0xF) : trash] = crs<R1, D>(f[ee]) * wm[j];       \
    stage[vm ? ms[R2] + (int)((cm >> 8) & 0xF) : trash] = crs<R2, D>(f[ee]) * wm[j];       \
    const uint32_t cf = FP.cand(j * 3 + D);                                                \
    const bool vf = (cf & 0xF) != 0xF;                                                     \
    stage[vf ? fs[R1] + (int)((cf >> 4) & 0xF) : trash] = crs<R1, D>(rv[ee]) * wf[j];      \
    stage[vf ? fs[R2] + (int)((cf >> 8) & 0xF) : trash] = crs<R2, D>(rv[ee]) * wf[j];      \
    stage[vf ? ls[D] + (int)((cf >> 12) & 0xF) : trash] = -wf[j];                          \
  }
      TWR_EE_TILE(0, 1, 2)
      TWR_EE_TILE(1, 2, 0)
      TWR_EE_TILE(2, 0, 1)
#undef TWR_EE_TILE
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ms[r] += meta_nslots(mmeta[ee]) - meta_cnt(mmeta[ee], r);
      fs[r] += meta_nslots(fmeta[ee]) - meta_cnt(fmeta[ee], r);
      ls[r] += meta_cnt(fmeta[ee], r);
    }
  }
}

// ---------------------------------------------------------------- range-of-motion item
// RangeOfMotionConstraint::{UpdateConstraintAtInstance, UpdateJacobianAtInstance}
// (range_of_motion_constraint.cc:58-109) for one (time node, ee).
TWR_DEV void rom_item(const DevStruct* __restrict__ S, const char* __restrict__ blob, const double* __restrict__ xp,
                      double* __restrict__ gp, double* __restrict__ stage, int soff, int trash, int k, int ee,
                      bool want_g, bool want_j) {
  const double tb = tbl<double>(blob, S->o_rom_tl_base)[k];
  const int q = tbl<int32_t>(blob, S->o_rom_base_poly)[k];
  const double iTb = tbl<double>(blob, S->o_base_iT)[q];
  double wP[4];
  hermite_pos(tb, iTb, wP);
  const double* xl = xp + S->off_base_lin + 6 * q;
  const double* xa = xp + S->off_base_ang + 6 * q;
  double c[3], e[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c[d] = wP[0] * xl[d] + wP[1] * xl[3 + d] + wP[2] * xl[6 + d] + wP[3] * xl[9 + d];
    e[d] = wP[0] * xa[d] + wP[1] * xa[3 + d] + wP[2] * xa[6 + d] + wP[3] * xa[9 + d];
  }
  const int mq = tbl<int32_t>(blob, S->o_rom_mpoly[ee])[k];
  const double tm = tbl<double>(blob, S->o_rom_tl_m[ee])[k];
  const PolyR MP = load_poly(tbl<EePoly>(blob, S->o_mpoly[ee]) + mq);
  double wm[4], p[3], v[3];
  ee_weights_and_point(MP, tm, xp, wm, p);
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = p[d] - c[d];
  Rot ro;
  rotation(e, ro);
  if (want_g) {
    double gv[3];
    matTvec(ro.R, v, gv);  // b_R_w (p - c)
    double* go = gp + S->row_rom[ee] + 3 * k;
    go[0] = gv[0]; go[1] = gv[1]; go[2] = gv[2];
  }
  if (!want_j) return;
  double ux[3], uy[3], uz[3];  // DerivOfRotVecMult(t, v, inverse=true): d(R^T v)/d e_d (euler_converter.cc:223-239)
  matTvec(ro.Rx, v, ux);
  matTvec(ro.Ry, v, uy);
  matTvec(ro.Rz, v, uz);
  const int nm = MP.nslots();
  const int rs[3] = {soff, soff + 20 + nm, soff + 44 + 2 * nm};
  const int mo[3] = {20, 24, 24};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int d = 0; d < 3; ++d) stage[rs[r] + 3 * j + d] = -ro.R[d][r] * wP[j];  // -R^T J_c
      if (r == 0) {  // row 0 of R^T v does not depend on roll
        stage[rs[0] + 12 + 2 * j + 0] = wP[j] * uy[0];
        stage[rs[0] + 12 + 2 * j + 1] = wP[j] * uz[0];
      } else {
        stage[rs[r] + 12 + 3 * j + 0] = wP[j] * ux[r];
        stage[rs[r] + 12 + 3 * j + 1] = wP[j] * uy[r];
        stage[rs[r] + 12 + 3 * j + 2] = wP[j] * uz[r];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {  // R^T J_p
        const uint32_t cd = MP.cand(j * 3 + d);
        const bool valid = (cd & 0xF) != 0xF;
        stage[valid ? rs[r] + mo[r] + (int)(cd & 0xF) : trash] = ro.R[d][r] * wm[j];
      }
    }
  }
}

// ---------------------------------------------------------------- terrain
// HeightMap example terrains: height, slopes, second derivative (height_map_examples.{h,cc}).
// Only Gap has curvature (GetHeightDerivWrtXX); Stairs overrides only GetHeight (zero slope).
struct Terr {
  double h, hx, hy, hxx;
};
TWR_DEV Terr terrain_eval(int id, double flat_height, double x, double y) {
  Terr t = {0.0, 0.0, 0.0, 0.0};
  switch (id) {
    case 0: t.h = flat_height; break;
    case 1: {  // Block (height_map_examples.cc:40-65)
      const double start = 0.7, len = 3.5, height = 0.5, eps = 0.03, slope = height / eps;
      if (start <= x && x <= start + eps) { t.h = slope * (x - start); t.hx = slope; }
      if (start + eps <= x && x <= start + len) t.h = height;
      break;
    }
    case 2: {  // Stairs (:69-84)
      if (x >= 1.0) t.h = 0.2;
      if (x >= 1.0 + 0.4) t.h = 0.4;
      if (x >= 1.0 + 0.4 + 1.0) t.h = 0.0;
      break;
    }
    case 3: {  // Gap (:88-120, height_map_examples.h:96-111)
      const double gs = 1.0, w = 0.5, hh = 1.5, xc = gs + w / 2.0, ge = gs + w;
      const double a = (4 * hh) / (w * w), b = -(8 * hh * xc) / (w * w), c = -(hh * (w - 2 * xc) * (w + 2 * xc)) / (w * w);
      if (gs <= x && x <= ge) { t.h = a * x * x + b * x + c; t.hx = 2 * a * x + b; t.hxx = 2 * a; }
      break;
    }
    case 4: {  // Slope (:124-157)
      const double ss = 1.0, up = 1.0, dn = 1.0, hc = 0.7, xd = ss + up, xf = xd + dn, sl = hc / up;
      if (x >= ss) { t.h = sl * (x - ss); t.hx = sl; }
      if (x >= xd) { t.h = hc - sl * (x - xd); t.hx = -sl; }
      if (x >= xf) { t.h = 0.0; t.hx = 0.0; }
      break;
    }
    case 5: {  // Chimney (:161-181)
      const double xs = 1.0, len = 1.5, ys = 0.5, sl = 3.0;
      if (xs <= x && x <= xs + len) { t.h = sl * (y - ys); t.hy = sl; }
      break;
    }
    case 6: {  // ChimneyLR (:185-211)
      const double xs = 0.5, len = 1.0, ys = 0.5, sl = 2, xe1 = xs + len, xe2 = xs + 2 * len;
      if (xs <= x && x <= xe1) { t.h = sl * (y - ys); t.hy = sl; }
      if (xe1 <= x && x <= xe2) { t.h = -sl * (y + ys); t.hy = -sl; }
      break;
    }
  }
  return t;
}

// ForceConstraint::{GetValues, FillJacobianBlock} (force_constraint.cc:62-171) for one stance force
// node; terrain basis and its "derivative" per height_map.cc:62-148 (component-wise product, as is).
TWR_DEV void force_item(const DevStruct* __restrict__ S, const ForceNode fn, const double* __restrict__ xp,
                        double* __restrict__ g5, double* __restrict__ st25, bool want_g, bool want_j) {
  const double f[3] = {xp[fn.fidx], xp[fn.fidx + 2], xp[fn.fidx + 4]};
  const double px = xp[fn.hidx], py = xp[fn.hidx + 1];
  const Terr t = terrain_eval(S->terrain_id, S->flat_height, px, py);
  const double mu = S->mu;
  const double vb[3][3] = {{-t.hx, -t.hy, 1.0}, {1.0, 0.0, t.hx}, {0.0, 1.0, t.hy}};  // n, t1, t2 (height_map.cc:93-139)
  double nb[3][3], sq[3], nr[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    sq[b] = vb[b][0] * vb[b][0] + vb[b][1] * vb[b][1] + vb[b][2] * vb[b][2];
    nr[b] = sqrt(sq[b]);
#pragma unroll
    for (int i = 0; i < 3; ++i) nb[b][i] = vb[b][i] / nr[b];
  }
  double rowv[5][3];  // the five pyramid directions
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rowv[0][i] = nb[0][i];
    rowv[1][i] = nb[1][i] - mu * nb[0][i];
    rowv[2][i] = nb[1][i] + mu * nb[0][i];
    rowv[3][i] = nb[2][i] - mu * nb[0][i];
    rowv[4][i] = nb[2][i] + mu * nb[0][i];
  }
  if (want_g) {
#pragma unroll
    for (int r = 0; r < 5; ++r) g5[r] = f[0] * rowv[r][0] + f[1] * rowv[r][1] + f[2] * rowv[r][2];
  }
  if (!want_j) return;
  // second derivatives: only h_xx can be non-zero (GetSecondDerivativeOfHeightWrt, height_map.cc:150-163)
  double drow[2][5];  // [dim][row]: f . d(direction)/d(foothold dim)
#pragma unroll
  for (int dim = 0; dim < 2; ++dim) {
    const double hx_d = dim == 0 ? t.hxx : 0.0;  // d(hx)/d(dim)
    const double hy_d = 0.0;                     // d(hy)/d(dim)
    const double dv[3][3] = {{-hx_d, -hy_d, 0.0}, {0.0, 0.0, hx_d}, {0.0, 0.0, hy_d}};
    double db[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double unit = i == dim ? 1.0 : 0.0;
        const double outer = (1.0 / sq[b]) * (nr[b] * unit - vb[b][dim] * nb[b][i]);
        db[b][i] = outer * dv[b][i];
      }
    double dr[5][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dr[0][i] = db[0][i];
      dr[1][i] = db[1][i] - mu * db[0][i];
      dr[2][i] = db[1][i] + mu * db[0][i];
      dr[3][i] = db[2][i] - mu * db[0][i];
      dr[4][i] = db[2][i] + mu * db[0][i];
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) drow[dim][r] = f[0] * dr[r][0] + f[1] * dr[r][1] + f[2] * dr[r][2];
  }
#pragma unroll
  for (int r = 0; r < 5; ++r) {  // columns: foothold x, y (ee-motion) then force px, py, pz
    st25[5 * r + 0] = drow[0][r];
    st25[5 * r + 1] = drow[1][r];
    st25[5 * r + 2] = rowv[r][0];
    st25[5 * r + 3] = rowv[r][1];
    st25[5 * r + 4] = rowv[r][2];
  }
}

// ---------------------------------------------------------------- LDS image -> global
// The image is placed at stage[par + i] with par = parity of the destination's 8-byte index,
// so that 16-byte aligned global pairs are 16-byte aligned LDS pairs.
TWR_DEV void copy_out(double* __restrict__ dst, const double* __restrict__ stage, int n, int par, int lane) {
  double* al = dst - par;  // 16-byte aligned
  const int total = n + par;
  const int npairs = total >> 1;
  for (int t = lane; t < npairs; t += 64) {
    if (t == 0 && par) {
      al[1] = stage[1];
    } else {
      const double2 v = *reinterpret_cast<const double2*>(stage + 2 * t);
      *reinterpret_cast<double2*>(al + 2 * t) = v;
    }
  }
  if ((total & 1) && lane == 0 && total - 1 >= par) al[total - 1] = stage[total - 1];
}

constexpr int kStageDoubles = 4936;  // ~39 KiB LDS image per workgroup: four workgroups share a CU's 160 KiB

template <int NEE>
__global__ __launch_bounds__(64) void eval_kernel(const ProbRec* __restrict__ probs, const Work* __restrict__ work,
                                                  const double* __restrict__ x, double* __restrict__ g,
                                                  double* __restrict__ jac, int flags) {
  __shared__ __attribute__((aligned(16))) double stage[kStageDoubles + 2 + 64];  // image + parity slack + trash slots
  const int trash = kStageDoubles + 2 + (int)threadIdx.x;
  const Work w = work[blockIdx.x];
  const ProbRec pr = probs[w.prob];
  const char* blob = reinterpret_cast<const char*>(pr.blob);
  const DevStruct* S = reinterpret_cast<const DevStruct*>(blob);
  const double* xp = x + pr.x_off;
  double* gp = g + pr.g_off;
  double* jp = jac + pr.j_off;
  const bool want_g = flags & 1, want_j = flags & 2;
  const int lane = threadIdx.x;

  if (w.kind == 0) {
    const int32_t* vo = tbl<int32_t>(blob, S->o_dyn_val_off);
    const int base = vo[w.k0], end = vo[w.k0 + w.cnt];
    double* dst = jp + S->nnz_dyn + base;
    const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
    if (lane < w.cnt) {
      const int k = w.k0 + lane;
      dyn_item<NEE>(S, blob, xp, gp, stage, par + vo[k] - base, trash, k, want_g, want_j);
    }
    if (want_j) {
      __syncthreads();
      copy_out(dst, stage, end - base, par, lane);
    }
  } else if (w.kind == 1) {
    const int ee = w.ee;
    const int32_t* vo = tbl<int32_t>(blob, S->o_rom_val_off[ee]);
    const int base = vo[w.k0], end = vo[w.k0 + w.cnt];
    double* dst = jp + S->nnz_rom[ee] + base;
    const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
    if (lane < w.cnt) {
      const int k = w.k0 + lane;
      rom_item(S, blob, xp, gp, stage, par + vo[k] - base, trash, k, ee, want_g, want_j);
    }
    if (want_j) {
      __syncthreads();
      copy_out(dst, stage, end - base, par, lane);
    }
  } else {
    // terrain-ee-motion_e (terrain_constraint.cc:57-108) then force-ee-force_e, 64 nodes at a time
    for (int ee = 0; ee < S->n_ee; ++ee) {
      const TerrainRow* rows = tbl<TerrainRow>(blob, S->o_terrain_rows[ee]);
      const int nr = S->n_terrain_rows[ee];
      for (int r0 = 0; r0 < nr; r0 += 64) {
        const int cnt = min(64, nr - r0);
        double* dst = jp + S->nnz_terrain[ee] + 3 * r0;
        const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        if (lane < cnt) {
          const TerrainRow tr = rows[r0 + lane];
          const double px = xp[tr.idx], py = xp[tr.idx + tr.stride], pz = xp[tr.idx + 2 * tr.stride];
          const Terr t = terrain_eval(S->terrain_id, S->flat_height, px, py);
          if (want_g) gp[S->row_terrain[ee] + r0 + lane] = pz - t.h;
          if (want_j) {
            stage[par + 3 * lane + 0] = -t.hx;
            stage[par + 3 * lane + 1] = -t.hy;
            stage[par + 3 * lane + 2] = 1.0;
          }
        }
        if (want_j) {
          __syncthreads();
          copy_out(dst, stage, 3 * cnt, par, lane);
          __syncthreads();
        }
      }
    }
    for (int ee = 0; ee < S->n_ee; ++ee) {
      const ForceNode* nodes = tbl<ForceNode>(blob, S->o_force_nodes[ee]);
      const int nn = S->n_force_nodes[ee];
      for (int i0 = 0; i0 < nn; i0 += 64) {
        const int cnt = min(64, nn - i0);
        double* dst = jp + S->nnz_force[ee] + 25 * i0;
        const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        if (lane < cnt)
          force_item(S, nodes[i0 + lane], xp, gp + S->row_force[ee] + 5 * (i0 + lane), stage + par + 25 * lane, want_g,
                     want_j);
        if (want_j) {
          __syncthreads();
          copy_out(dst, stage, 25 * cnt, par, lane);
          __syncthreads();
        }
      }
    }
  }
}

// host-side launcher (called from capi.cc)
hipError_t launch_eval(int n_ee, int n_work, const ProbRec* probs, const Work* work, const double* x, double* g,
                       double* jac, int flags, hipStream_t stream) {
  if (n_work <= 0) return hipSuccess;
  dim3 grid(n_work), block(64);
  switch (n_ee) {
    case 1: hipLaunchKernelGGL(eval_kernel<1>, grid, block, 0, stream, probs, work, x, g, jac, flags); break;
    case 2: hipLaunchKernelGGL(eval_kernel<2>, grid, block, 0, stream, probs, work, x, g, jac, flags); break;
    case 3: hipLaunchKernelGGL(eval_kernel<3>, grid, block, 0, stream, probs, work, x, g, jac, flags); break;
    case 4: hipLaunchKernelGGL(eval_kernel<4>, grid, block, 0, stream, probs, work, x, g, jac, flags); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

int stage_capacity_doubles() { return kStageDoubles; }

}  // namespace twr
