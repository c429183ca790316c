// HIP kernels (gfx950 / CDNA4) for towr's NLP constraint + Jacobian callback.
//
// Three launches per callback batch (dynamic, range of motion, node-based sets), one stream; batches of up to 8192
// rom slices run the same three bodies as block roles of ONE launch (eval_fused_kernel).
// A dyn/rom workgroup is one wavefront (64 lanes); it walks a strided list of work items, each one a
// *contiguous* slice of one problem's CSR value array:
//   dyn_kernel   dynamic           : <= 16 consecutive time nodes k, FOUR lanes per node (6 rows each)
//   rom_kernel   rangeofmotion-ee  : lanes = consecutive time nodes k               (3 rows each)
//   node_kernel  terrain-* / force-* / splineacc-base-* / swing-* (+ totalduration-*) of one problem:
//                four waves, one per family, lanes = spline nodes / rows (node_kernel2: the first two families only)
//   phase_locate_kernel + dyn_phase_kernel / rom_phase_kernel: the same math for problems with optimised phase
//                durations (x-dependent active polynomials, rows that hold all variables of every ee set): 16 resp. 4
//                lanes per time node, expanded rows assembled in LDS
// Every lane computes its rows in registers (FP64, no MFMA: the work is 3x3 algebra),
// scatters the values into an LDS image of the slice at the CSR position they have in
// global memory, and the wave then streams the image out with 16-byte coalesced stores.
// x-independent index work (active polynomial, local time, node->column maps, CSR
// offsets) comes from the per-lane records of device_tables.h.  The dyn/rom kernels are
// software pipelined over their work list: while slice i is computed and stored, the record of
// slice i+2 and the x values of slice i+1 are already in flight, so the only exposed memory
// latency per slice is the LDS round trip.
//
// Math follows the reference line by line in meaning (citations per function); the
// arithmetic is re-associated (Hermite basis form, factored base-ang tile) and agrees
// with the reference formulas to rounding, tests/ pin that at <= 1e-9.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <tuple>
#include <utility>

#include "device_tables.h"

namespace twr {

#define TWR_DEV __device__ __forceinline__
#ifdef TWR_DIAG_NOCONFLICT
// DIAGNOSTIC BUILD ONLY (make DIAG=-DTWR_DIAG_NOCONFLICT; wrong results on purpose): every scattered 8-byte store into the LDS
// image of dyn_kernel / rom_kernel goes to `8 * lane + a constant of the call site` instead of its CSR position -- the sixteen
// lanes of a ds_write_b64 group then hit sixteen different bank pairs (MI355X_MICROARCH.md: 4 x 16 contiguous lanes, bank =
// (a / 4) mod 32), with the same number of stores and no address arithmetic at all: an UPPER bound of what removing the bank
// conflicts of the image scatter can buy.
template <int N>
TWR_DEV void diag_put(double v) {
  const uint32_t a = (threadIdx.x & 63u) << 3;
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"((N % 30) * 512));
}
#define lds_put(l, o, ...) diag_put<__COUNTER__>(__VA_ARGS__)
#define TWR_ROM_PUT(idx, ...) diag_put<__COUNTER__>(__VA_ARGS__)
#else
#define TWR_ROM_PUT(idx, ...) stage[idx] = (__VA_ARGS__)
#endif
constexpr int TWR_MAX_PHASES_DEV = 32;  // = TWR_MAX_PHASES of include/towr_amd.h

template <typename T>
TWR_DEV const T* tbl(const char* blob, uint32_t off) {
  return reinterpret_cast<const T*>(blob + off);
}
// Device addresses stored as integers in the work items: tell the compiler they are global memory
// (a plain reinterpret_cast yields a generic pointer -> flat_load, which also ties up lgkmcnt).
#if defined(__HIP_DEVICE_COMPILE__)
#define TWR_GLOBAL __attribute__((address_space(1)))
#else
#define TWR_GLOBAL
#endif
template <typename T>
TWR_DEV const TWR_GLOBAL T* gptr(uint64_t addr) {
  return reinterpret_cast<const TWR_GLOBAL T*>(addr);
}
// Read-only tables reached through a wave-uniform address: the constant address space makes their loads
// scalar (s_load into SGPRs) instead of one vector load per lane.
#if defined(__HIP_DEVICE_COMPILE__)
#define TWR_CONST __attribute__((address_space(4)))
#else
#define TWR_CONST
#endif
template <typename T>
TWR_DEV const TWR_CONST T* cptr(uint64_t addr) {
  return reinterpret_cast<const TWR_CONST T*>(addr);
}

// Kernel launch that RETURNS its status (hipLaunchKernel) instead of leaving it in the thread's sticky error slot: the
// library neither reads nor clears an error the host application may have pending.
template <typename T, size_t... I>
static inline hipError_t twr_launch_impl(const void* k, dim3 g, dim3 b, size_t lds, hipStream_t s, T& vals, std::index_sequence<I...>) {
  void* ptrs[] = {static_cast<void*>(&std::get<I>(vals))...};
  return hipLaunchKernel(k, g, b, ptrs, lds, s);
}
template <typename... P, typename... A>
static inline hipError_t twr_launch(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A... args) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  std::tuple<P...> vals{static_cast<P>(args)...};
  return twr_launch_impl(reinterpret_cast<const void*>(kern), grid, block, lds, stream, vals, std::index_sequence_for<P...>{});
}
static inline hipError_t twr_first(hipError_t a, hipError_t b) { return a != hipSuccess ? a : b; }

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

// ---------------------------------------------------------------- candidate descriptors
TWR_DEV int meta_nslots(uint32_t meta) { return meta & 0xF; }
TWR_DEV bool meta_shared(uint32_t meta) { return (meta >> 16) & 1; }
// all 12 candidate loads are issued unconditionally (absent candidates read slot 0 and later get
// weight 0): one memory round trip, no branches (Spline::GetPoint needs at most these 12 values)
TWR_DEV void gather12(const double* __restrict__ xp, int xbase, uint64_t slots, double v[12]) {
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const int sl = (int)((slots >> (4 * c)) & 0xF);
    v[c] = xp[xbase + (sl != 0xF ? sl : 0)];
  }
}
// Position of an ee spline from its gathered node values (Spline::GetPoint, src/spline.cc:80-93, in
// Hermite basis form).  A stance ee-motion polynomial keeps one shared position variable for both
// nodes: w_p1 is folded into w_p0.
TWR_DEV void ee_point(uint64_t slots, bool shared, double tl, double iT, const double v[12], double w[4], double out[3]) {
  hermite_pos(tl, iT, w);
  if (shared) w[0] += w[2];
  out[0] = out[1] = out[2] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool valid = ((slots >> (4 * (j * 3 + d))) & 0xF) != 0xF;
      out[d] = fma(valid ? w[j] : 0.0, v[j * 3 + d], out[d]);
    }
}
TWR_DEV void gather12c(const double* __restrict__ xp, int xbase, const uint16_t cand[12], double v[12]) {
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const int sl = cand[c] & 0xF;
    v[c] = xp[xbase + (sl != 0xF ? sl : 0)];
  }
}
TWR_DEV uint64_t slots_of(const uint16_t cand[12]) {
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < 12; ++c) s |= (uint64_t)(cand[c] & 0xF) << (4 * c);
  return s;
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

// sin and cos of one Euler angle.  Cody-Waite reduction by pi/2 in three FMA steps and the fdlibm
// minimax kernels (k_sin.c / k_cos.c coefficients), ~35 FP64 instructions for both results and
// accurate to about 1 ulp for |x| < 1e5; anything larger (or non-finite) takes the ocml path.
// The reference calls libm sin/cos (euler_converter.cc:133-221); the difference is rounding level.
TWR_DEV void sincos_fast(double x, double* __restrict__ s, double* __restrict__ c) {
  if (!(fabs(x) < 1.0e5)) {
    sincos(x, s, c);
    return;
  }
  // (the constants pass through an empty asm: they are then materialised where they are used -- two s_mov each -- instead
  // of being hoisted out of the persistent loops, where the register allocator ends up spilling them to scratch)
  auto K = [](double c) {
    asm volatile("" : "+s"(c));
    return c;
  };
  const double n = rint(x * K(6.36619772367581382433e-01));
  double r = fma(-n, K(1.5707963267948966e+00), x);
  r = fma(-n, K(6.1232339957367574e-17), r);
  r = fma(-n, K(8.4784276603688985e-32), r);
  const double z = r * r;
  double ps = fma(z, K(1.58969099521155010221e-10), K(-2.50507602534068634195e-08));
  ps = fma(z, ps, K(2.75573137070700676789e-06));
  ps = fma(z, ps, K(-1.98412698298579493134e-04));
  ps = fma(z, ps, K(8.33333333332248946124e-03));
  ps = fma(z, ps, K(-1.66666666666666324348e-01));
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, K(-1.13596475577881948265e-11), K(2.08757232129817482790e-09));
  pc = fma(z, pc, K(-2.75573143513906633035e-07));
  pc = fma(z, pc, K(2.48015872894767294178e-05));
  pc = fma(z, pc, K(-1.38888888888741095749e-03));
  pc = fma(z, pc, K(4.16666666666666019037e-02));
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)n & 3;
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// ZYX Euler rotation base->world (euler_converter.cc:207-221).
struct Rot {
  double R[3][3];
  double sx, cx, sy, cy, sz, cz;
};
TWR_DEV void rotation(const double e[3], Rot& o) {
  double sx, cx, sy, cy, sz, cz;
  sincos_fast(e[0], &sx, &cx);
  sincos_fast(e[1], &sy, &cy);
  sincos_fast(e[2], &sz, &cz);
  o.sx = sx; o.cx = cx; o.sy = sy; o.cy = cy; o.sz = sz; o.cz = cz;
  o.R[0][0] = cy * cz; o.R[0][1] = cz * sx * sy - cx * sz; o.R[0][2] = sx * sz + cx * cz * sy;
  o.R[1][0] = cy * sz; o.R[1][1] = cx * cz + sx * sy * sz; o.R[1][2] = cx * sy * sz - cz * sx;
  o.R[2][0] = -sy;     o.R[2][1] = cy * sx;                o.R[2][2] = cx * cy;
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

// ---------------------------------------------------------------- quad helpers
// The dynamic kernel maps FOUR lanes to one time node (a "quad" = lanes 4i..4i+3); they exchange
// scalars with DPP quad permutes (no LDS, no memory).
template <int CTRL>
TWR_DEV double quad_perm(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
TWR_DEV double quad_sum(double v) {
  v += quad_perm<0xB1>(v);  // lane ^ 1
  v += quad_perm<0x4E>(v);  // lane ^ 2
  return v;
}
TWR_DEV double sel3(int i, double a, double b, double c) { return i == 0 ? a : (i == 1 ? b : c); }

// ---------------------------------------------------------------- optimised timings helpers
// Spline::GetSegmentID + GetLocalTime (src/spline.cc:48-78) on durations that depend on x: the same
// double-precision accumulation, eps and sequential subtraction as the reference (and the host Locate).
TWR_DEV int locate_segment(const double* __restrict__ d, int n, double t, double& t_local) {
  // Branch-free on purpose: with the reference's early `break` every iteration's load waits for the previous compare
  // (a chain of dependent LDS round trips); predicated, the loads of an unrolled block go out together and only the
  // additions are sequential.  Same additions and subtractions in the same order, so the same bits.
  const double eps = 1e-10;
  double acc = 0.0, tl = t;
  int id = n - 1;  // (the reference asserts that a segment is found)
  bool found = false;
#pragma unroll 8
  for (int i = 0; i < n; ++i) {
    const double di = d[i];
    acc += di;
    const bool hit = !found && acc >= t - eps;
    id = hit ? i : id;
    found = found || hit;
    if (!found && i < n - 1) tl -= di;   // t_local = t - sum of the durations BEFORE the segment, sequentially
  }
  t_local = tl;
  return id;
}
// node values (p0,v0,p1,v1)[dim] of the active polynomial from its gathered candidates: values that are
// not optimised are the constant 0, a stance ee-motion polynomial has p1 = p0
TWR_DEV void node_values(uint64_t slots, bool shared, const double v[12], double nv[4][3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool valid = ((slots >> (4 * (j * 3 + d))) & 0xF) != 0xF;
      nv[j][d] = valid ? v[j * 3 + d] : 0.0;
    }
  if (shared) {
#pragma unroll
    for (int d = 0; d < 3; ++d) nv[2][d] = nv[0][d];
  }
}
// ---------------------------------------------------------------- dynamic (SRBD) quad
// DynamicConstraint::{UpdateModel, UpdateConstraintAtInstance, UpdateJacobianAtInstance}
// (dynamic_constraint.cc:59-137) with SingleRigidBodyDynamics::{GetDynamicViolation,
// GetJacobianWrt{BaseLin,BaseAng,Force,EEPos}} (single_rigid_body_dynamics.cc:76-192) and the
// EulerConverter derivatives (euler_converter.cc:85-131,168-198,223-304) for one time node,
// computed by the four lanes of a quad:
//   role = lane & 3 evaluates end-effector `role` (splines, [f]x J_p and {[r]x J_f ; -J_f} blocks),
//   role 0..2 additionally produce Euler dimension `role` of the base-ang block, role 3 the
//   base-lin block and the six constraint values.
// The base-ang block is evaluated in factored form: with u = (p0,v0,p1,v1) of Euler dim d,
//   d g_ang / d u_j = A_d wP[j] + B_d wV[j] + C_d wA[j],
//   A_d = d g_ang/d e_d, B_d = d g_ang/d edot_d, C_d = d g_ang/d eddot_d  (3-vectors).
// Two kernels share this mapping: dyn_kernel (fixed timings: dyn2_front / dyn2_back, every index resolved on the
// host) and dyn_phase_kernel (optimised timings: pdyn_math / pdyn_puts behind the phase_locate_kernel pre-pass).

// ---------------------------------------------------------------- range-of-motion item
// RangeOfMotionConstraint::{UpdateConstraintAtInstance, UpdateJacobianAtInstance}
// (range_of_motion_constraint.cc:58-109) for one (time node, ee).
struct RomX {
  double bl[12], ba[12], m[12];
};
TWR_DEV uint64_t rom_slots(const RomRec& r) { return ((uint64_t)r.slots[1] << 32) | r.slots[0]; }
// What a lane of rom_kernel carries through the pipeline, as loaded (raw: nothing is computed from the records at issue
// time, so a prefetch never waits): its time node's record -- fetched THREE slices ahead, because it also says which segment
// record the lane reads -- and its segment's record, fetched two slices ahead.
TWR_DEV RomNode rom_load_node(const RomWork& w, int lane) { return gptr<RomNode>(w.nodes)[min(lane, w.cnt - 1)]; }
TWR_DEV RomSeg rom_load_seg(const RomWork& w, const RomNode& nd) { return gptr<RomSeg>(w.segs)[(nd.seg >> (3 * w.ee)) & 7u]; }
// the lane's view in the form the math is written for (fields of the optimised-timings record RomRec)
TWR_DEV RomRec rom_rec_of(const RomWork& w, const RomNode& nd, const RomSeg& sg, int lane) {
  RomRec r;
  r.tb = nd.tb; r.iTb = nd.iTb;
  r.tm = nd.t - sg.t0; r.iTm = sg.iTm;
  r.q6 = nd.q6;
  r.xbase = sg.xbase;
  r.voff = sg.voff0 + (min(lane, w.cnt - 1) - sg.kfirst) * sg.node_vals;   // relative to the slice's first value
  r.meta = sg.meta;
  r.slots[0] = sg.slots[0]; r.slots[1] = sg.slots[1];
  r.pad[0] = r.pad[1] = 0;
  return r;
}
TWR_DEV void rom_load_x(const RomWork& w, const RomNode& nd, const RomSeg& sg, const double* __restrict__ x, RomX& X) {
  const double* xp = x + w.x_off;
  const double* xl = xp + w.off_lin + nd.q6;
  const double* xa = xp + w.off_ang + nd.q6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    X.bl[i] = xl[i];
    X.ba[i] = xa[i];
  }
  gather12(xp, sg.xbase, ((uint64_t)sg.slots[1] << 32) | sg.slots[0], X.m);
}
TWR_DEV void rom_item(const RomWork& w, const RomRec& r, const RomX& X, double* __restrict__ gst, double* __restrict__ stage,
                      int par, int vbase, int trash, int lane, bool want_g, bool want_j) {
  const int soff = par + r.voff - vbase;
  double wP[4];
  hermite_pos(r.tb, r.iTb, wP);
  double c[3], e[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c[d] = wP[0] * X.bl[d] + wP[1] * X.bl[3 + d] + wP[2] * X.bl[6 + d] + wP[3] * X.bl[9 + d];
    e[d] = wP[0] * X.ba[d] + wP[1] * X.ba[3 + d] + wP[2] * X.ba[6 + d] + wP[3] * X.ba[9 + d];
  }
  const uint64_t slots = rom_slots(r);
  double wm[4], p[3], v[3];
  ee_point(slots, meta_shared(r.meta), r.tm, r.iTm, X.m, wm, p);
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = p[d] - c[d];
  Rot ro;
  rotation(e, ro);
  if (want_g) {
    double gv[3];
    matTvec(ro.R, v, gv);  // b_R_w (p - c)
    double* go = gst + 3 * lane;  // staged in LDS, written out coalesced with the Jacobian slice
    go[0] = gv[0]; go[1] = gv[1]; go[2] = gv[2];
  }
  if (!want_j) return;
  // DerivOfRotVecMult(t, v, inverse=true) = d(R^T v)/d e_d (euler_converter.cc:223-239).  With
  // dR/d e_d = [M_d]x R (M_d = rotation axis d of the ZYX sequence = column d of M) this is R^T (v x M_d).
  double ux[3], uy[3], uz[3];
  {
    const double Mx[3] = {ro.cy * ro.cz, ro.cy * ro.sz, -ro.sy}, My[3] = {-ro.sz, ro.cz, 0.0};
    double tx[3], ty[3];
    cross3(v, Mx, tx);
    cross3(v, My, ty);
    const double tz[3] = {v[1], -v[0], 0.0};  // v x e_z
    matTvec(ro.R, tx, ux);
    matTvec(ro.R, ty, uy);
    matTvec(ro.R, tz, uz);
  }
  const int nm = meta_nslots(r.meta);
  const int rs[3] = {soff, soff + 20 + nm, soff + 44 + 2 * nm};
  const int mo[3] = {20, 24, 24};
#pragma unroll
  for (int row = 0; row < 3; ++row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int d = 0; d < 3; ++d) TWR_ROM_PUT(rs[row] + 3 * j + d, -ro.R[d][row] * wP[j]);  // -R^T J_c
      if (row == 0) {  // row 0 of R^T v does not depend on roll
        TWR_ROM_PUT(rs[0] + 12 + 2 * j + 0, wP[j] * uy[0]);
        TWR_ROM_PUT(rs[0] + 12 + 2 * j + 1, wP[j] * uz[0]);
      } else {
        TWR_ROM_PUT(rs[row] + 12 + 3 * j + 0, wP[j] * ux[row]);
        TWR_ROM_PUT(rs[row] + 12 + 3 * j + 1, wP[j] * uy[row]);
        TWR_ROM_PUT(rs[row] + 12 + 3 * j + 2, wP[j] * uz[row]);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {  // R^T J_p
        const int sl = (int)((slots >> (4 * (j * 3 + d))) & 0xF);
        TWR_ROM_PUT(sl != 0xF ? rs[row] + mo[row] + sl : trash, ro.R[d][row] * wm[j]);
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
// HeightMapFromCSV (include/towr/terrain/height_map_from_csv.h:29-109): piecewise-constant heights per cell;
// the "slope" is the step to the next / previous cell smeared over eps on the lower side of the step, checked
// in the reference's order (next cell first); second derivatives are the base class's zeros.
TWR_DEV Terr grid_eval(const DevStruct* __restrict__ S, double x, double y) {
  Terr t = {0.0, 0.0, 0.0, 0.0};
  const double* __restrict__ grid = reinterpret_cast<const double*>(S->grid_ptr);
  const int rows = S->grid_rows, cols = S->grid_cols;
  const double res = S->grid_res, eps = S->grid_eps;
  const long xc = (long)(x / res), yc = (long)(y / res);   // truncation toward zero, like static_cast<size_t>
  if (xc < 0 || yc < 0 || xc >= cols || yc >= rows) return t;
  const double h0 = grid[yc * cols + xc];
  t.h = h0;
#pragma unroll
  for (int dim = 0; dim < 2; ++dim) {
    const double v = dim == 0 ? x : y;
    const long c = dim == 0 ? xc : yc, n = dim == 0 ? cols : rows;
    const long stride = dim == 0 ? 1 : cols;
    double d = 0.0;
    bool done = false;
    if (c + 1 < n) {
      const double diff_end = grid[yc * cols + xc + stride] - h0;
      const double v_end = (double)(c + 1) * res;
      if (diff_end > 0 && v <= v_end && v >= v_end - eps) { d = diff_end / eps; done = true; }
    }
    if (!done && c - 1 >= 0) {
      const double diff_start = h0 - grid[yc * cols + xc - stride];
      const double v_start = (double)c * res;
      if (diff_start < 0 && v >= v_start && v <= v_start + eps) d = diff_start / eps;
    }
    if (dim == 0) t.hx = d; else t.hy = d;
  }
  return t;
}
// Grid (include/towr/terrain/grid_height_map.h:15-60) over the float "elevation" layer of a ROS grid_map.
// gridmap_sample = grid_map's published GridMap::atPosition(layer, position, INTER_LINEAR) for a map with start
// index (0,0) (third party, absent and unpinned in the reference: restated from its published algorithm --
// atPositionLinearInterpolated, GridMapMath getIndexFromPosition / getPositionFromIndex / checkIfPositionWithinMap):
// cell (i,j) centre = map position + length/2 - (index + 1/2) res; index = trunc((position + length/2 - p) / res);
// bilinear over the 2x2 cells around p evaluated in double and rounded to FLOAT; if one of the four cells lies outside
// the map: the nearest cell when p is inside the map, otherwise std::out_of_range, which Grid::GetHeight turns into
// numeric_limits<float>::max() (:41-45).
TWR_DEV float gridmap_sample(const DevStruct* __restrict__ S, double x, double y) {
  const float* __restrict__ el = reinterpret_cast<const float*>(S->grid_ptr);
  const int sx = S->grid_rows, sy = S->grid_cols;
  const double res = S->grid_res, lx = sx * res, ly = sy * res, mx = S->grid_px, my = S->grid_py;
  const long i0 = (long)(-((x - 0.5 * lx - mx) / res)), j0 = (long)(-((y - 0.5 * ly - my) / res));
  const double tx = mx + 0.5 * lx - x, ty = my + 0.5 * ly - y;
  const bool inside = tx >= 0.0 && ty >= 0.0 && tx < lx && ty < ly;
  const double cx0 = mx + 0.5 * lx - 0.5 * res - res * (double)i0, cy0 = my + 0.5 * ly - 0.5 * res - res * (double)j0;
  const long ia = x >= cx0 ? i0 : i0 + 1, ja = y >= cy0 ? j0 : j0 + 1, ib = ia - 1, jb = ja - 1;
  if (ib >= 0 && jb >= 0 && ia < sx && ja < sy) {
    const double px = mx + 0.5 * lx - 0.5 * res - res * (double)ia, py = my + 0.5 * ly - 0.5 * res - res * (double)ja;
    const double rx = (x - px) / res, ry = (y - py) / res, fx = 1.0 - rx, fy = 1.0 - ry;
    const double f0 = el[ia + ja * (long)sx], f1 = el[ib + ja * (long)sx], f2 = el[ia + jb * (long)sx], f3 = el[ib + jb * (long)sx];
    // no FMA contraction: the reference's products and sums are separately rounded doubles
    return (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(__dmul_rn(f0, fx), fy), __dmul_rn(__dmul_rn(f1, rx), fy)),
                                      __dmul_rn(__dmul_rn(f2, fx), ry)),
                            __dmul_rn(__dmul_rn(f3, rx), ry));
  }
  if (inside && i0 >= 0 && j0 >= 0 && i0 < sx && j0 < sy) return el[i0 + j0 * (long)sx];
  return 3.402823466e+38f;  // numeric_limits<float>::max()
}
TWR_DEV Terr gridmap_eval(const DevStruct* __restrict__ S, double x, double y) {
  Terr t = {0.0, 0.0, 0.0, 0.0};
  const double eps = S->grid_eps;
  t.h = (double)gridmap_sample(S, x, y);
  // grid_height_map.h:48-60: the two heights are floats, their difference is a float, the quotient a double
  t.hx = (double)(gridmap_sample(S, x + eps, y) - gridmap_sample(S, x - eps, y)) / (2 * eps);
  t.hy = (double)(gridmap_sample(S, x, y + eps) - gridmap_sample(S, x, y - eps)) / (2 * eps);
  return t;
}
TWR_DEV Terr terrain_eval(const DevStruct* __restrict__ S, int id, double flat_height, double x, double y) {
  Terr t = {0.0, 0.0, 0.0, 0.0};
  switch (id) {
    case 7: return grid_eval(S, x, y);
    case 8: return gridmap_eval(S, x, y);
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
TWR_DEV void force_core(const DevStruct* __restrict__ S, const double f[3], double px, double py, double* __restrict__ g5,
                        double* __restrict__ st25, bool want_g, bool want_j);
TWR_DEV void force_item(const DevStruct* __restrict__ S, const ForceNode fn, const double* __restrict__ xp,
                        double* __restrict__ g5, double* __restrict__ st25, bool want_g, bool want_j) {
  const double f[3] = {xp[fn.fidx], xp[fn.fidx + 2], xp[fn.fidx + 4]};
  force_core(S, f, xp[fn.hidx], xp[fn.hidx + 1], g5, st25, want_g, want_j);
}
// (f: the node's force, px / py: the stance foothold; the single body of the force rows for node_kernel and node_chunk_kernel)
TWR_DEV void force_core(const DevStruct* __restrict__ S, const double f[3], double px, double py, double* __restrict__ g5,
                        double* __restrict__ st25, bool want_g, bool want_j) {
  const Terr t = terrain_eval(S, S->terrain_id, S->flat_height, px, py);
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
// Same, with a compile-time number of store instructions: iterations past the end re-store the
// last pair (idempotent).  A fixed count lets the compiler keep the stores in flight behind a
// counted s_waitcnt vmcnt(N) when the persistent loop next touches its prefetched loads (on gfx9
// loads and stores retire through one in-order counter; an unknown store count would force
// vmcnt(0), i.e. a full store drain per slice).
// NT: the 16-byte stores carry the non-temporal hint (global_store_dwordx4 ... nt): the values stream to HBM without
// displacing what the kernels RE-READ -- a sweep's per-candidate tables and x -- from the XCD's L2 and the Infinity Cache.
// Chosen per batch (twr_batch_create): worth 5-15 % of a step whose candidates all bring their own tables and
// whose output exceeds the Infinity Cache; it COSTS rom_kernel 10-15 % when one structure serves thousands of problems.
template <int NIT, int kBatch, bool NT = false>
TWR_DEV void copy_out_fixed(double* __restrict__ dst, const double* __restrict__ stage, int n, int par, int lane) {
  // byte offsets as unsigned 32-bit values: one VGPR addresses both the LDS read and the global store (wave-uniform
  // base + 32-bit offset), two VALU instructions per store instead of a 64-bit address computation
  char* al = reinterpret_cast<char*>(dst - par);  // 16-byte aligned
  const char* st = reinterpret_cast<const char*>(stage);
  const int total = n + par;
  const uint32_t last = (uint32_t)((total >> 1) - 1) * 16u;  // last complete pair; pairs [par, npairs) are complete
  const uint32_t first = (uint32_t)(par + lane) * 16u;
  // Branch-free on purpose: lanes past the end of the slice re-store its last pair (idempotent; predicating them off
  // makes the compiler sink every LDS read into its own `if`), and the count of vector-memory instructions stays a
  // compile-time constant (counted s_waitcnt for the prefetched loads).
  // kBatch = LDS reads in flight before the first store needs its data.  (In dyn_kernel, which sits at the VGPR limit,
  // the register allocator leaves the copy-out one quad of registers and the reads end up serialised with their stores
  // anyway; moving the copy-out ahead of the front half or pipelining it by hand changed nothing measurable.)
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += kBatch) {
    double2 v[kBatch];
    uint32_t off[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b)
      if (it0 + b < NIT) {
        off[b] = min(first + 1024u * (uint32_t)(it0 + b), last);
        v[b] = *reinterpret_cast<const double2*>(st + off[b]);
      }
#pragma unroll
    for (int b = 0; b < kBatch; ++b)
      if (it0 + b < NIT) {
        if (NT) {
          typedef double twr_d2 __attribute__((ext_vector_type(2)));
          const twr_d2 vv = {v[b].x, v[b].y};
          __builtin_nontemporal_store(vv, reinterpret_cast<twr_d2*>(al + off[b]));
        } else {
          *reinterpret_cast<double2*>(al + off[b]) = v[b];
        }
      }
  }
  // first and last value of the slice (they sit in half pairs when the slice starts / ends on an odd index): stored by
  // every lane, whatever the parity -- two more store instructions that are ALWAYS issued.  A store inside a divergent
  // `if` is a branch the compiler must assume not taken when it counts the stores behind a prefetched load, and one
  // uncounted store is enough to turn the next counted wait into a drain of the whole copy-out.
  // (A/B on one box, rom_kernel: these two as `if (lane == 0)` stores cost 4 %: 0.876 -> 0.914 ms)
  dst[0] = stage[par];
  dst[n - 1] = stage[total - 1];
}

// LDS image sizes (doubles).  dynamic: 16 time nodes per wave (~170 values each at 4 ee) -> 22.5 KiB,
// seven workgroups per CU; range of motion: 64 lanes x ~84 values -> 39 KiB, four per CU.
// (kRomStage, the rom image: device_tables.h)

// dynamic / range of motion: persistent workgroups, software pipelined over the strided work list.
// Loop body for slice i (its record -- and for rom its x values -- were prefetched):
//   C  compute slice i into the LDS image          (the ONLY call site of the math: every slice of a
//      batch goes through the same instruction sequence, so equal x give bit-identical results
//      wherever they sit in the batch)
//   A  issue the record loads of slice i+2 and (rom) the x loads of slice i+1
//   B  stream the image of slice i to HBM with a fixed number of store instructions
// One wave per workgroup: its LDS accesses are ordered, no barriers.
// dynamic kernel loop.  On gfx9 a wave's loads and stores retire through ONE in-order counter, so x
// loads issued right after a slice's stores are only seen once those stores have drained (measured: a
// quarter of this kernel).  The slice is therefore copied out one phase late:
//   front(i+1): x loads of slice i+1 (behind the stores of slice i-1, long gone) -> compact state
//   copy-out(i): image of slice i -> HBM
//   back(i+1):  Jacobian blocks of slice i+1 -> image        (single call site of each half)
#ifndef TWR_TU_ROM
// ---------------------------------------------------------------- dynamic, fixed timings (dyn_kernel)
// DynamicConstraint + SingleRigidBodyDynamics + EulerConverter for one time node on a quad of lanes (see above); every
// index is an LDS byte offset prepared on the host (device_tables.h DynNodeL / DynSel / DynPolyL / DynTile): the slice's part of x
// sits in LDS ("xs"), values that are not optimisation variables read its zero slot, and a Jacobian value is stored at
// node base + a 16-bit offset from the record.
constexpr int kDynX0 = kDynG0 + 96;        // xs: zero pair, then <= kDynXsCap staged doubles of x
constexpr int kDynLds = kDynX0 + 2 + kDynXsCap;   // 2560 doubles = 20480 B: eight workgroups per CU (the VGPR limit too)
static_assert(kDynLds * 8 <= 20480, "dyn_kernel: eight workgroups of 20 KB per CU");
#ifndef TWR_DYN_COPY_BATCH
#define TWR_DYN_COPY_BATCH 8
#endif
#ifndef TWR_DYN_WAVES
#define TWR_DYN_WAVES 2   // waves per SIMD dyn_kernel is compiled for (experiments: 3 with -DTWR_DYN_IMAGE=1362 -DTWR_DYN_XS=126)
#endif

constexpr int kDynCopyBatch = TWR_DYN_COPY_BATCH;   // LDS reads in flight before the first store of the copy-out
struct Dyn2Front {   // (the base-spline weights are recomputed in the back half: 48 registers less across the copy-out)
  double cdd[3], ed[3], edd[3];
  double wm[4], wf[4], f[3], rv[3], F[3], tau[3];
  double sx, cx, sy, cy, sz, cz;
};
TWR_DEV double lds_f64(const char* __restrict__ lds, uint32_t byte_off) {
  return *reinterpret_cast<const double*>(lds + byte_off);
}
#ifndef TWR_DIAG_NOCONFLICT
TWR_DEV void lds_put(char* __restrict__ lds, uint32_t byte_off, double v) { *reinterpret_cast<double*>(lds + byte_off) = v; }
#endif

// staged candidate c of a spline: xs[idx[c]], idx = twelve bytes in three dwords
TWR_DEV void gather12s(const char* __restrict__ xs, const uint32_t w[3], double v[12]) {
#pragma unroll
  for (int c = 0; c < 12; ++c) v[c] = lds_f64(xs, ((w[c >> 2] >> (8 * (c & 3))) & 0xFFu) << 3);
}
// What a lane needs of its records for the FRONT half of a slice (prefetched during the previous slice's back half) and
// for the tile stores of the BACK half (the codes, loaded while the previous slice is copied out).  Whole dwords: no
// sub-dword struct fields cross the loop back-edge (see the compiler note in DESIGN 6.0).
struct DynFrontRec {
  DynNode nd;
  double t0m, iTm, t0f, iTf;
  uint32_t relm[3], presm[3], relf[3], presf[3];
  uint32_t flagsm, flagsf;
  uint32_t tl[5];      // DynTile as five dwords: base_m[0..1] | base_m[2], base_f[0] | base_f[1..2] | base_f[3..4] | base_f[5], s_m, s_f
};
struct DynCodes {
  uint32_t m[6];       // DynPolyL::code of the ee-motion polynomial: [c][2]
  uint32_t f[9];       // of the ee-force polynomial: [c][3]
};
TWR_DEV uint32_t dyn2_load_sel(const DynWork& w, int lane) {
  const int kk = min(lane >> 2, w.cnt - 1);
  return gptr<uint32_t>(w.sel)[kk * 4 + (lane & 3)];
}
// The records of roles >= n_ee (DynSel::dm / df = 255): every value reads the zero slot, finite weights, codes 0.  Constants of
// the code object, the same for every structure.
__device__ const DynPolyT g_dyn_dummy_t = {0.0, 1.0};
__device__ const DynPolyL g_dyn_dummy_l = {{0}, {0}, 2u | 4u, {0}};
TWR_DEV const TWR_GLOBAL char* dyn2_poly_t_addr(const DynWork& w, uint32_t d) {   // d = DynSel::dm or df
  const uint64_t rec = w.poly_t + (uint64_t)d * sizeof(DynPolyT);
  return reinterpret_cast<const TWR_GLOBAL char*>(d == (uint32_t)kDynPolyDummy ? reinterpret_cast<uint64_t>(&g_dyn_dummy_t) : rec);
}
TWR_DEV const TWR_GLOBAL char* dyn2_poly_l_addr(const DynWork& w, uint32_t d) {
  const uint64_t rec = w.poly_l + (uint64_t)d * sizeof(DynPolyL);
  return reinterpret_cast<const TWR_GLOBAL char*>(d == (uint32_t)kDynPolyDummy ? reinterpret_cast<uint64_t>(&g_dyn_dummy_l) : rec);
}
TWR_DEV void dyn2_load_front(const DynWork& w, uint32_t sel, int lane, DynFrontRec& r) {
  const int kk = min(lane >> 2, w.cnt - 1);
  const DynNodeT nt = gptr<DynNodeT>(w.nodes_t)[kk];
  const uint4 nl = gptr<uint4>(w.nodes_l)[kk];
  static_assert(sizeof(DynNodeL) == sizeof(uint4), "DynNodeL as four dwords");
  r.nd.t = nt.t; r.nd.tb = nt.tb; r.nd.iTb = nt.iTb;
  r.nd.sb_lin = (uint16_t)nl.x; r.nd.sb_ang = (uint16_t)(nl.x >> 16);
  r.nd.nb = (uint16_t)nl.y;     r.nd.rs1 = (uint16_t)(nl.y >> 16);
  r.nd.rs2 = (uint16_t)nl.z;    r.nd.rl[0] = (uint16_t)(nl.z >> 16);
  r.nd.rl[1] = (uint16_t)nl.w;  r.nd.rl[2] = (uint16_t)(nl.w >> 16);
  const uint32_t im = (sel >> 16) & 0xFFu, jf = sel >> 24;
  const TWR_GLOBAL double* dm = reinterpret_cast<const TWR_GLOBAL double*>(dyn2_poly_t_addr(w, im));
  const TWR_GLOBAL double* df = reinterpret_cast<const TWR_GLOBAL double*>(dyn2_poly_t_addr(w, jf));
  const TWR_GLOBAL uint32_t* um = reinterpret_cast<const TWR_GLOBAL uint32_t*>(dyn2_poly_l_addr(w, im));
  const TWR_GLOBAL uint32_t* uf = reinterpret_cast<const TWR_GLOBAL uint32_t*>(dyn2_poly_l_addr(w, jf));
  const TWR_GLOBAL uint32_t* tl = reinterpret_cast<const TWR_GLOBAL uint32_t*>(w.tile + (uint64_t)(sel & 0xFFFFu) * sizeof(DynTile));
  static_assert(offsetof(DynPolyL, rel) == 0 && offsetof(DynPolyL, pres) == 12 && offsetof(DynPolyL, flags) == 24 && offsetof(DynPolyL, code) == 28,
                "DynPolyL dwords");
  r.t0m = dm[0]; r.iTm = dm[1];
  r.t0f = df[0]; r.iTf = df[1];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r.relm[i] = um[i]; r.presm[i] = um[3 + i];
    r.relf[i] = uf[i]; r.presf[i] = uf[3 + i];
  }
  r.flagsm = um[6];
  r.flagsf = uf[6];
#pragma unroll
  for (int i = 0; i < 5; ++i) r.tl[i] = tl[i];
}
TWR_DEV void dyn2_load_codes(const DynWork& w, uint32_t sel, DynCodes& c) {
  const TWR_GLOBAL uint32_t* um = reinterpret_cast<const TWR_GLOBAL uint32_t*>(dyn2_poly_l_addr(w, (sel >> 16) & 0xFFu));
  const TWR_GLOBAL uint32_t* uf = reinterpret_cast<const TWR_GLOBAL uint32_t*>(dyn2_poly_l_addr(w, sel >> 24));
#pragma unroll
  for (int i = 0; i < 6; ++i) c.m[i] = um[7 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.f[i] = uf[7 + i];
}
TWR_DEV void dyn2_front(const DynFrontRec& r, const char* __restrict__ xs, int lane, Dyn2Front& S) {
  const DynNode& nd = r.nd;
  const int role = lane & 3;
  double wP[4], wV[4], wA[4];
  hermite_all(nd.tb, nd.iTb, wP, wV, wA);
  double c[3], e[3];
  {  // base spline points (Spline::GetPoint in basis form): the active polynomial's [p0 v0 p1 v1] x 3
    double bl[12], ba[12];
    const double2* pl = reinterpret_cast<const double2*>(__builtin_assume_aligned(xs + nd.sb_lin, 16));
    const double2* pa = reinterpret_cast<const double2*>(__builtin_assume_aligned(xs + nd.sb_ang, 16));
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double2 a = pl[i], b = pa[i];
      bl[2 * i] = a.x; bl[2 * i + 1] = a.y;
      ba[2 * i] = b.x; ba[2 * i + 1] = b.y;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      c[d] = wP[0] * bl[d] + wP[1] * bl[3 + d] + wP[2] * bl[6 + d] + wP[3] * bl[9 + d];
      S.cdd[d] = wA[0] * bl[d] + wA[1] * bl[3 + d] + wA[2] * bl[6 + d] + wA[3] * bl[9 + d];
      e[d] = wP[0] * ba[d] + wP[1] * ba[3 + d] + wP[2] * ba[6 + d] + wP[3] * ba[9 + d];
      S.ed[d] = wV[0] * ba[d] + wV[1] * ba[3 + d] + wV[2] * ba[6 + d] + wV[3] * ba[9 + d];
      S.edd[d] = wA[0] * ba[d] + wA[1] * ba[3 + d] + wA[2] * ba[6 + d] + wA[3] * ba[9 + d];
    }
  }
  // staging indices of the twelve candidates of a spline, four at a time: (S4 + rel4) & pres4 -- a candidate that is not a
  // variable reads index 0, the zero slot (no selects in the spline evaluation)
  const uint32_t sm4 = ((r.tl[4] >> 16) & 0xFFu) * 0x01010101u, sf4 = (r.tl[4] >> 24) * 0x01010101u;
  // this lane's end-effector: weights and spline points
  {
    double vm[12], p[3];
    const uint32_t im[3] = {(sm4 + r.relm[0]) & r.presm[0], (sm4 + r.relm[1]) & r.presm[1], (sm4 + r.relm[2]) & r.presm[2]};
    gather12s(xs, im, vm);
    hermite_pos(nd.t - r.t0m, r.iTm, S.wm);
    S.wm[0] += (r.flagsm & 1) ? S.wm[2] : 0.0;   // stance polynomial: p1 is the same variable as p0
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      p[d] = S.wm[0] * vm[d] + S.wm[1] * vm[3 + d] + S.wm[2] * vm[6 + d] + S.wm[3] * vm[9 + d];
      S.rv[d] = c[d] - p[d];
    }
  }
  {
    double vf[12];
    const uint32_t jf[3] = {(sf4 + r.relf[0]) & r.presf[0], (sf4 + r.relf[1]) & r.presf[1], (sf4 + r.relf[2]) & r.presf[2]};
    gather12s(xs, jf, vf);
    hermite_pos(nd.t - r.t0f, r.iTf, S.wf);
#pragma unroll
    for (int d = 0; d < 3; ++d) S.f[d] = S.wf[0] * vf[d] + S.wf[1] * vf[3 + d] + S.wf[2] * vf[6 + d] + S.wf[3] * vf[9 + d];
  }
  // force and torque sums over the end-effectors (single_rigid_body_dynamics.cc:81-88); dummy roles carry f = 0
  double t3[3];
  cross3(S.f, S.rv, t3);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    S.F[d] = quad_sum(S.f[d]);
    S.tau[d] = quad_sum(t3[d]);
  }
  // rotation: lanes 0..2 of the quad evaluate one sincos each and broadcast it
  double my_s, my_c;
  sincos_fast(sel3(role, e[0], e[1], e[2]), &my_s, &my_c);
  S.sx = quad_perm<0x00>(my_s); S.cx = quad_perm<0x00>(my_c);
  S.sy = quad_perm<0x55>(my_s); S.cy = quad_perm<0x55>(my_c);
  S.sz = quad_perm<0xAA>(my_s); S.cz = quad_perm<0xAA>(my_c);
}

// Back half: Jacobian blocks and constraint values into the LDS image.  `img` is the byte address of the image
// (parity shift included), node base and row offsets come from DynNodeL, tile offsets from DynTile + DynPolyL::code.
TWR_DEV void dyn2_back(const DynWork& w, const DynFrontRec& rec, const DynCodes& cd, const Dyn2Front& S, double* __restrict__ gst,
                       char* __restrict__ img, int lane, bool want_g, bool want_j) {
  const DynNode& nd = rec.nd;
  const int kk = lane >> 2, role = lane & 3;
  if (kk >= w.cnt) return;
  const double (&cdd)[3] = S.cdd, (&ed)[3] = S.ed, (&edd)[3] = S.edd;
  const double (&wm)[4] = S.wm, (&wf)[4] = S.wf, (&f)[3] = S.f, (&rv)[3] = S.rv, (&F)[3] = S.F, (&tau)[3] = S.tau;
  const double sx = S.sx, cx = S.cx, sy = S.sy, cy = S.cy, sz = S.sz, cz = S.cz;
  char* nb = img + nd.nb;   // first value of this time node
  // --- ee-motion block [f]x J_p (:181-192) and ee-force block {[r]x J_f ; -J_f} (:167-179) of this lane's
  // end-effector, BEFORE the base blocks.  A value goes to  node base + tile start in its row (DynTile) + 8 * rank
  // (DynPolyL::code, one byte per value); see DynPolyL for where the values of candidates that are not variables end up.
  if (want_j) {
    char* rowm[3] = {nb + (rec.tl[0] & 0xFFFFu), nb + (rec.tl[0] >> 16), nb + (rec.tl[1] & 0xFFFFu)};
    char* rowf[6] = {nb + (rec.tl[1] >> 16), nb + (rec.tl[2] & 0xFFFFu), nb + (rec.tl[2] >> 16),
                     nb + (rec.tl[3] & 0xFFFFu), nb + (rec.tl[3] >> 16), nb + (rec.tl[4] & 0xFFFFu)};
    // a force node that is constant stores the values of the polynomial's other node once more (same slots, same values)
    const bool c0 = rec.flagsf & 2, c1 = rec.flagsf & 4;
    const double wfp[4] = {c0 ? wf[2] : wf[0], c0 ? wf[3] : wf[1], c1 ? wf[0] : wf[2], c1 ? wf[1] : wf[3]};
#define TWR_CODE_M(c, k) ((cd.m[(2 * (c) + (k)) >> 2] >> (8 * ((2 * (c) + (k)) & 3))) & 0xFFu)
#define TWR_CODE_F(c, k) ((cd.f[(3 * (c) + (k)) >> 2] >> (8 * ((3 * (c) + (k)) & 3))) & 0xFFu)
#define TWR_EE_TILE_M(J, D, R1, R2)                                              \
  {                                                                              \
    lds_put(rowm[R1], TWR_CODE_M((J) * 3 + D, 0), crs<R1, D>(f) * wm[J]);        \
    lds_put(rowm[R2], TWR_CODE_M((J) * 3 + D, 1), crs<R2, D>(f) * wm[J]);        \
  }
#define TWR_EE_TILE_F(J, D, R1, R2)                                              \
  {                                                                              \
    lds_put(rowf[R1], TWR_CODE_F((J) * 3 + D, 0), crs<R1, D>(rv) * wfp[J]);      \
    lds_put(rowf[R2], TWR_CODE_F((J) * 3 + D, 1), crs<R2, D>(rv) * wfp[J]);      \
    lds_put(rowf[3 + D], TWR_CODE_F((J) * 3 + D, 2), -wfp[J]);                   \
  }
#define TWR_EE_TILES(J)      \
  TWR_EE_TILE_M(J, 0, 1, 2)  \
  TWR_EE_TILE_M(J, 1, 2, 0)  \
  TWR_EE_TILE_M(J, 2, 0, 1)  \
  TWR_EE_TILE_F(J, 0, 1, 2)  \
  TWR_EE_TILE_F(J, 1, 2, 0)  \
  TWR_EE_TILE_F(J, 2, 0, 1)
    TWR_EE_TILES(3)   // (ee-motion: p0 LAST -- it overwrites what candidates without a variable left in its slots)
    TWR_EE_TILES(2)
    TWR_EE_TILES(1)
    TWR_EE_TILES(0)
#undef TWR_EE_TILES
#undef TWR_EE_TILE_F
#undef TWR_EE_TILE_M
#undef TWR_CODE_F
#undef TWR_CODE_M
  }
  // --- angular quantities (euler_converter.cc:58-83,133-166,207-221)
  double R[3][3];
  R[0][0] = cy * cz; R[0][1] = cz * sx * sy - cx * sz; R[0][2] = sx * sz + cx * cz * sy;
  R[1][0] = cy * sz; R[1][1] = cx * cz + sx * sy * sz; R[1][2] = cx * sy * sz - cz * sx;
  R[2][0] = -sy;     R[2][1] = cy * sx;                R[2][2] = cx * cy;
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
  const TWR_CONST DevStruct* H = cptr<DevStruct>(w.hdr);  // uniform per work item: scalar loads
  double Iw6[6];  // I_w = R I_b R^T (single_rigid_body_dynamics.cc:91), symmetric: (00,01,02,11,12,22)
  {
    double Ib[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Ib[i] = H->Ib[i];
    double T[3][3];  // R I_b
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      T[i][0] = R[i][0] * Ib[0] + R[i][1] * Ib[1] + R[i][2] * Ib[2];
      T[i][1] = R[i][0] * Ib[1] + R[i][1] * Ib[3] + R[i][2] * Ib[4];
      T[i][2] = R[i][0] * Ib[2] + R[i][1] * Ib[4] + R[i][2] * Ib[5];
    }
    Iw6[0] = T[0][0] * R[0][0] + T[0][1] * R[0][1] + T[0][2] * R[0][2];
    Iw6[1] = T[0][0] * R[1][0] + T[0][1] * R[1][1] + T[0][2] * R[1][2];
    Iw6[2] = T[0][0] * R[2][0] + T[0][1] * R[2][1] + T[0][2] * R[2][2];
    Iw6[3] = T[1][0] * R[1][0] + T[1][1] * R[1][1] + T[1][2] * R[1][2];
    Iw6[4] = T[1][0] * R[2][0] + T[1][1] * R[2][1] + T[1][2] * R[2][2];
    Iw6[5] = T[2][0] * R[2][0] + T[2][1] * R[2][1] + T[2][2] * R[2][2];
  }
  double Iw_wd[3], Iw_w[3];
  symmul(Iw6, omd, Iw_wd);
  symmul(Iw6, om, Iw_w);
  const double m = H->mass;
  double wP[4], wV[4], wA[4];
  {
    double tb = nd.tb;
    asm volatile("" : "+v"(tb));   // a fresh evaluation: do not carry the front half's twelve weights across the copy-out
    hermite_all(tb, nd.iTb, wP, wV, wA);
  }
  char* row[3] = {nb, nb + nd.rs1, nb + nd.rs2};

  if (role == 3) {
    if (want_g) {  // GetDynamicViolation, single_rigid_body_dynamics.cc:76-101
      double wxIw[3];
      cross3(om, Iw_w, wxIw);
      double* go = gst + 6 * kk;  // staged in LDS, written out coalesced with the Jacobian slice
#pragma unroll
      for (int i = 0; i < 3; ++i) go[i] = Iw_wd[i] + wxIw[i] - tau[i];
      go[3] = m * cdd[0] - F[0];
      go[4] = m * cdd[1] - F[1];
      go[5] = m * cdd[2] - F[2] + m * H->gravity;
    }
  }
  if (want_j) {  // base-lin block: ang rows -sum_i [f_i]x J_pos, lin rows m J_acc (:103-121).  Every lane of the quad stores
                 // the nine values of ONE node value j = role (all four roles busy instead of role 3 storing all 36)
    const double wPj = role == 0 ? wP[0] : (role == 1 ? wP[1] : (role == 2 ? wP[2] : wP[3]));
    const double wAj = role == 0 ? wA[0] : (role == 1 ? wA[1] : (role == 2 ? wA[2] : wA[3]));
    const uint32_t o2 = 16u * (uint32_t)role, o1 = 8u * (uint32_t)role;
    lds_put(row[0], o2, -crs<0, 1>(F) * wPj);
    lds_put(row[0], o2 + 8, -crs<0, 2>(F) * wPj);
    lds_put(row[1], o2, -crs<1, 0>(F) * wPj);
    lds_put(row[1], o2 + 8, -crs<1, 2>(F) * wPj);
    lds_put(row[2], o2, -crs<2, 0>(F) * wPj);
    lds_put(row[2], o2 + 8, -crs<2, 1>(F) * wPj);
    const double ma = m * wAj;
#pragma unroll
    for (int d = 0; d < 3; ++d) lds_put(nb + nd.rl[d], o1, ma);
  }
  if (role != 3 && want_j) {
    // --- base-ang block (:123-165), Euler dimension d = role, factored (derivation: see pdyn_math)
    const double dMx_dy[3] = {-sy * cz, -sy * sz, -cy};
    const double dMx_dz[3] = {-cy * sz, cy * cz, 0.0};
    const double dMy_dz[3] = {-cz, -sz, 0.0};
    const double dMdx_dy[3] = {-cz * cy * yd + sy * sz * zd, -sy * cz * zd - cy * sz * yd, sy * yd};
    const double dMdx_dz[3] = {sz * sy * yd - cy * cz * zd, -cy * sz * zd - sy * cz * yd, 0.0};
    const double dMdy_dz[3] = {sz * zd, -cz * zd, 0.0};
    double Md[3], dwd_ed[3], dw[3], dwd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Md[i] = sel3(role, Mx[i], My[i], i == 2 ? 1.0 : 0.0);  // column d of M (euler_converter.cc:133-148)
      dwd_ed[i] = sel3(role, Mdx[i], Mdy[i] + xd * dMx_dy[i], xd * dMx_dz[i] + yd * dMy_dz[i]);
      dw[i] = sel3(role, 0.0, xd * dMx_dy[i], xd * dMx_dz[i] + yd * dMy_dz[i]);
      dwd[i] = sel3(role, 0.0, xd * dMdx_dy[i] + edd[0] * dMx_dy[i],
                    xd * dMdx_dz[i] + yd * dMdy_dz[i] + edd[0] * dMx_dz[i] + edd[1] * dMy_dz[i]);
    }
    auto dIw = [&](const double v[3], const double Iwv[3], double o[3]) {
      double t1[3], t2[3], t3b[3];
      cross3(Md, Iwv, t1);
      cross3(v, Md, t2);
      symmul(Iw6, t2, t3b);
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = t1[i] + t3b[i];
    };
    double A[3], B[3], C[3];
    symmul(Iw6, Md, C);
    {
      double t1[3], t2[3], t3b[3];
      symmul(Iw6, dwd_ed, t1);
      cross3(Md, Iw_w, t2);
      cross3(om, C, t3b);
#pragma unroll
      for (int i = 0; i < 3; ++i) B[i] = t1[i] + t2[i] + t3b[i];
    }
    {
      double t1[3], t2[3], t3b[3], t4[3], t5[3], t6[3], t7[3];
      dIw(omd, Iw_wd, t1);
      symmul(Iw6, dwd, t2);
      cross3(dw, Iw_w, t3b);
      dIw(om, Iw_w, t4);
      symmul(Iw6, dw, t5);
#pragma unroll
      for (int i = 0; i < 3; ++i) t6[i] = t4[i] + t5[i];
      cross3(om, t6, t7);
#pragma unroll
      for (int i = 0; i < 3; ++i) A[i] = t1[i] + t2[i] + t3b[i] + t7[i];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_put(row[r], 8 * (8 + 3 * j + role), A[r] * wP[j] + B[r] * wV[j] + C[r] * wA[j]);
  }
}

// Loop of one persistent single-wave workgroup over its strided slices.  State at the top of iteration i: xs holds
// x(i); `xr` (registers) holds x(i+1), gathered one iteration ago; mapr holds the staging map of slice i+2;
// the front records of slice i and the record selectors (DynSel) of slices i and i+1 are in registers.
//   F  front(i): LDS reads of xs + the front records -> compact state
//   P  issue the loads of slice i's tile codes (needed by the back half only) and of slice i+1's front records (node,
//      polynomial and tile records, addressed through its selector, which arrived an iteration ago); they land during
//      the copy-out
//   O  copy-out(i-1): image -> HBM (one phase late: the x gathers of the previous iteration were issued before
//      these stores, so waiting for them never waits for a store)
//   B  back(i): Jacobian blocks -> image
//   S  xs <- xr (x(i+1));  then issue: selector of slice i+2, xr <- gather x(i+2), mapr <- map(i+3)
// (`stage`: kDynLds doubles of LDS owned by this wave; the wave takes slices i, i + stride, ...)
// WANT_G / WANT_J are compile-time, the first iteration copies the not-yet-filled image to a dump region instead of
// skipping the copy-out, and no store sits in a divergent branch: the number of vector-memory instructions between a
// prefetch and its use is then a constant on every path, and the compiler's counted waits stay counted.  (With run-time
// flags and `if (pending)` it had to assume the shortest path -- no stores at all -- and the wait for the put record in
// the middle of the loop drained the whole copy-out of the previous slice.)
// XC = 64-entry chunks of the staging map the slices of the batch use (2: every slice stages at most 128 doubles of x --
// true for all K = 200 problems, whose 12..15-node slices stage ~100 -- and reads the 256-byte form of its map; 4: the
// general 512-byte form).  A compile-time count: the x gather is XC loads per lane, the same on every path.
template <bool WANT_G, bool WANT_J, int XC, bool NT>
TWR_DEV void dyn_body(const DynWork* __restrict__ work, int n_work, const double* __restrict__ x, double* __restrict__ g,
                      double* __restrict__ jac, double* __restrict__ dump, double* stage, int lane, int i, int stride) {
  static_assert(XC == 2 || XC == 4, "staging map chunks");
  // (WANT_J false: no image -- the wave's LDS would be g + xs alone)
  constexpr int kG0 = WANT_J ? kDynG0 : 0, kX0 = kG0 + 96;
  double* gst = stage + kG0;
  char* xs = reinterpret_cast<char*>(stage + kX0);
  if (i >= n_work) return;
  constexpr int NIT = (kDynImage + 2 + 127) / 128;
  if (lane < 2) stage[kX0 + lane] = 0.0;   // the zero pair
  auto load_map = [&](const DynWork& w) {   // XC 16-bit x indices per lane
    uint2 m;
    if (XC == 4) m = gptr<uint2>(w.map)[lane];
    else m = make_uint2(gptr<uint32_t>(w.map)[lane], 0u);
    return m;
  };
  auto gather_x = [&](const DynWork& w, uint2 m, double xr[XC]) {
    const double* xp = x + w.x_off;
    xr[0] = xp[m.x & 0xFFFFu];
    xr[1] = xp[m.x >> 16];
    if (XC == 4) {
      xr[XC - 2] = xp[m.y & 0xFFFFu];
      xr[XC - 1] = xp[m.y >> 16];
    }
  };
  auto stage_x = [&](const double xr[XC]) {
#pragma unroll
    for (int c = 0; c < XC; ++c)
      if (64 * c + lane < kDynXsCap) stage[kX0 + 2 + 64 * c + lane] = xr[c];
  };
  // image -> HBM; the constraint values (6 per time node, contiguous in g) with clamped lanes instead of predicates
  auto copy_out = [&](double* pdst, double* pg, int nvals, int cnt) {
    const int ppar = (int)((reinterpret_cast<uintptr_t>(pdst) >> 3) & 1);
    // (NT: batches of two -- with the non-temporal builtin the LDS reads of a batch keep registers of their own, and eight of
    // them in flight put 12 VGPRs into scratch; the plain form ends up serialised through one quad whatever the batch is)
    if (WANT_J) copy_out_fixed<NIT, NT ? 2 : kDynCopyBatch, NT>(pdst, stage, nvals, ppar, lane);
    if (WANT_G) {
      const int last = 6 * cnt - 1;
      pg[min(lane, last)] = gst[min(lane, last)];
      pg[min(lane + 64, last)] = gst[min(lane + 64, last)];
    }
  };
  // Every prefetch is issued unconditionally, with the slice index clamped to this workgroup's last slice (a repeated load
  // of the last slice's records / x is harmless): no prefetched value goes through a phi or a conditional copy -- that is
  // what makes the compiler wait for a load right where it is issued (see rom_body) -- and the number of vector-memory
  // instructions per iteration is the same on every path.
  const int last = i + (n_work - 1 - i) / stride * stride;
  DynWork wp = work[i];   // slice whose image is waiting to be copied out
  // (nothing is pending in the first iteration: its copy-out goes to the dump region, same instruction count)
  double* pdst = dump;
  double* pg = dump + kDynImage + 2;
  DynWork w0 = wp, w1 = work[min(i + stride, last)], w2 = work[min(i + 2 * stride, last)];
  DynFrontRec fr0;
  double xr[XC];
  uint2 mapr = load_map(w0);
  uint32_t sel0 = dyn2_load_sel(w0, lane);
  dyn2_load_front(w0, sel0, lane, fr0);                  // (the only exposed record -> record dependency)
  gather_x(w0, mapr, xr);
  stage_x(xr);                                           // x(first slice): the only exposed gather
  mapr = load_map(w1);
  uint32_t sel1 = dyn2_load_sel(w1, lane);
  gather_x(w1, mapr, xr);
  mapr = load_map(w2);
  for (; i <= last; i += stride) {
    const DynWork w3 = work[min(i + 3 * stride, last)];
    double* dst = jac + w0.j_off;
    const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
    Dyn2Front S;
    dyn2_front(fr0, xs, lane, S);                                                            // F
    DynCodes cd;
    dyn2_load_codes(w0, sel0, cd);                                                           // P  (lands during the copy-out)
    DynFrontRec fr1;
    dyn2_load_front(w1, sel1, lane, fr1);                                                    //    front records of slice i+1
    // (the copy-out runs at raised wave priority: with two waves per SIMD a wave in its store phase then gets its LDS
    // reads and stores issued ahead of its neighbour's math, which keeps the store stream of the CU steadier -- A/B on one
    // box 0.602-0.608 -> 0.591-0.596 ms; raised priority around the prefetches as well, or in rom_kernel, which runs one
    // wave per SIMD, is neutral to slower)
    __builtin_amdgcn_s_setprio(3);
    copy_out(pdst, pg, wp.nvals, wp.cnt);                                                    // O
    __builtin_amdgcn_s_setprio(0);
    dyn2_back(w0, fr0, cd, S, gst, reinterpret_cast<char*>(stage + par), lane, WANT_G, WANT_J);   // B
    stage_x(xr);                                                                             // S
    const uint32_t sel2 = dyn2_load_sel(w2, lane);
    gather_x(w2, mapr, xr);
    mapr = load_map(w3);
    wp = w0;
    pdst = dst;
    pg = g + w0.g_off;
    w0 = w1; fr0 = fr1; sel0 = sel1;
    w1 = w2; sel1 = sel2;
    w2 = w3;
  }
  copy_out(pdst, pg, wp.nvals, wp.cnt);                 // last slice of this workgroup
}

template <bool WANT_G, bool WANT_J, int XC, bool NT>
__global__ __launch_bounds__(64, TWR_DYN_WAVES) void dyn_kernel(const DynWork* __restrict__ work, int n_work, const double* __restrict__ x,
                                                    double* __restrict__ g, double* __restrict__ jac, double* __restrict__ dump) {
  __shared__ __attribute__((aligned(16))) double stage[kDynLds];
  dyn_body<WANT_G, WANT_J, XC, NT>(work, n_work, x, g, jac, dump, stage, threadIdx.x, blockIdx.x, gridDim.x);
}

// Values only (TWR_EVAL_VALUES: Ipopt's eval_g, a planner's scoring step): problems with fixed timings take the one-lane-per-
// time-node kernels below (eval_values_kernel); kDynValuesLds is what node_body needs of a wave's LDS there.
constexpr int kDynValuesLds = 96 + 2 + kDynXsCap;

#endif  // !TWR_TU_ROM

// rom_kernel is compiled in its own translation unit (rom_tu.hip) with a different instruction scheduling
// strategy: it is store bound and gains 4-5 % from clause-oriented scheduling, the VALU-bound kernels lose.
constexpr int kRomLds = kRomStage + 2 + 64 + 192;   // doubles: image, per-lane trash slots, g
// NIT = store instructions of the copy-out: the launcher picks the smallest instantiation that covers the largest slice
// of the batch.  The stores past the end of a slice are re-stores of its last pair -- no HBM traffic, but requests all
// the same: C3's balanced 50-node slices need 34, and 34 instead of 38 is worth 4 % of the kernel (A/B on one box: 0.925
// -> 0.883 ms together with the balanced slices; five 40-node slices with 27 stores each: 0.965 ms -- large slices win).
// (The hand-scheduled form of dyn_phase_kernel -- asm loads into AGPRs, one counted wait behind the copy-out -- was built
// for this loop too and is SLOWER here, 0.98 vs 0.90 ms on one box: the compiler's clause-oriented schedule of the 28
// loads with scalar bases and immediate offsets beats 28 separate asm loads with per-lane 64-bit addresses.)
template <int NIT, bool WANT_G, bool WANT_J, bool NT>
TWR_DEV void rom_body(const RomWork* __restrict__ work, int n_work, const double* __restrict__ x, double* __restrict__ g,
                      double* __restrict__ jac, double* stage, int lane, int i, int stride) {
  static_assert(NIT * 128 <= kRomStage + 2 + 127, "copy-out longer than the image");
  const int trash = kRomStage + 2 + lane;
  double* gst = stage + (WANT_J ? kRomStage + 2 + 64 : 0);   // (values only: the wave's LDS is the 192 constraint values alone)
  if (i >= n_work) return;
  // Every prefetch is issued unconditionally, with the slice index clamped to this workgroup's last slice (a repeated load
  // of the last slice's records / x is harmless): no prefetched value goes through a phi or a conditional copy, which is
  // what makes the compiler wait for a load right where it is issued (round 4: a `r2 = r1; if (has2) r2 = load` form of
  // this loop waited with vmcnt(0) behind the record loads in every iteration -- +8 % with L2-resident tables, +17 % in a sweep).
  const int last = i + (n_work - 1 - i) / stride * stride;
  RomWork w0 = work[i], w1 = work[min(i + stride, last)], w2 = work[min(i + 2 * stride, last)];
  RomNode n0 = rom_load_node(w0, lane), n1 = rom_load_node(w1, lane), n2 = rom_load_node(w2, lane);
  RomSeg s0 = rom_load_seg(w0, n0), s1 = rom_load_seg(w1, n1);
  RomX X;
  rom_load_x(w0, n0, s0, x, X);
  for (; i <= last; i += stride) {
    const RomWork w3 = work[min(i + 3 * stride, last)];
    double* dst = jac + w0.j_off;
    const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
    if (lane < w0.cnt) rom_item(w0, rom_rec_of(w0, n0, s0, lane), X, gst, stage, par, 0, trash, lane, WANT_G, WANT_J);   // C
    const RomNode n3 = rom_load_node(w3, lane);         // A: node records three slices ahead, segment records two ahead
    const RomSeg s2 = rom_load_seg(w2, n2);             //    (records first: the wait for x retires them too)
    rom_load_x(w1, n1, s1, x, X);
    if (WANT_J) copy_out_fixed<NIT, 13, NT>(dst, stage, w0.nvals, par, lane);   // B
    if (WANT_G) {                                       //   3 constraint values per time node, contiguous in g: clamped
      double* go = g + w0.g_off;                        //   lanes instead of predicates (see copy_out_fixed)
      const int last_g = 3 * w0.cnt - 1;
#pragma unroll
      for (int t = 0; t < 3; ++t) go[min(lane + 64 * t, last_g)] = gst[min(lane + 64 * t, last_g)];
    }
    w0 = w1; n0 = n1; s0 = s1;
    w1 = w2; n1 = n2; s1 = s2;
    w2 = w3; n2 = n3;
  }
}

constexpr int kRomNitMax = (kRomStage + 2 + 127) / 128;   // 38
#ifdef TWR_TU_ROM
template <int NIT, bool WANT_G, bool WANT_J, bool NT>
__global__ __launch_bounds__(64, 1) void rom_kernel(const RomWork* __restrict__ work, int n_work, const double* __restrict__ x,
                                                    double* __restrict__ g, double* __restrict__ jac) {
  __shared__ __attribute__((aligned(16))) double stage[kRomLds];
  rom_body<NIT, WANT_G, WANT_J, NT>(work, n_work, x, g, jac, stage, threadIdx.x, blockIdx.x, gridDim.x);
}

// max_vals: Jacobian values of the largest slice of the batch
hipError_t launch_rom_kernel(int grid, hipStream_t stream, const RomWork* rom, int n_rom, int max_vals, const double* x, double* g,
                             double* jac, int flags, bool nt) {
  const bool wg = flags & 1, wj = flags & 2;
  const int need = (max_vals + 1 + 2 + 127) / 128;   // (+ parity shift, rounded up to whole store instructions)
#define TWR_ROM_LAUNCH(NIT)                                                                                                  \
  {                                                                                                                          \
    if (wg && wj && nt) return twr_launch(rom_kernel<NIT, true, true, true>, dim3(grid), dim3(64), 0, stream, rom, n_rom, x, g, jac);    \
    if (wg && wj) return twr_launch(rom_kernel<NIT, true, true, false>, dim3(grid), dim3(64), 0, stream, rom, n_rom, x, g, jac);         \
    if (wj && nt) return twr_launch(rom_kernel<NIT, false, true, true>, dim3(grid), dim3(64), 0, stream, rom, n_rom, x, g, jac);         \
    if (wj) return twr_launch(rom_kernel<NIT, false, true, false>, dim3(grid), dim3(64), 0, stream, rom, n_rom, x, g, jac);              \
    return twr_launch(rom_kernel<NIT, true, false, false>, dim3(grid), dim3(64), 0, stream, rom, n_rom, x, g, jac);                      \
  }
  if (need <= 26) TWR_ROM_LAUNCH(26)
  if (need <= 30) TWR_ROM_LAUNCH(30)
  if (need <= 34) TWR_ROM_LAUNCH(34)
  TWR_ROM_LAUNCH(kRomNitMax)
#undef TWR_ROM_LAUNCH
}
#else   // !TWR_TU_ROM
hipError_t launch_rom_kernel(int grid, hipStream_t stream, const RomWork* rom, int n_rom, int max_vals, const double* x, double* g,
                             double* jac, int flags, bool nt);

// all terrain-ee-motion_e (terrain_constraint.cc:57-108), force-ee-force_e, splineacc-base-* and
// swing-ee-motion_e sets of one problem.  One workgroup of four waves per problem, one wave per family
// (each with its own LDS image, no barriers), 64 spline nodes / rows at a time.
constexpr int kStageTerrain = 64 * 3 + 2, kStageForce = 64 * 25 + 2, kStageAcc = 64 * 6 + 2, kStageSwing = 64 * 12 + 2;
constexpr int kNodeStageOff[4] = {0, kStageTerrain, kStageTerrain + kStageForce, kStageTerrain + kStageForce + kStageAcc};
// one family (wave-uniform 0..3) of one problem; `stage`: the family's LDS image (kStage<Family> doubles)
TWR_DEV void node_body(const NodeWork& w, const double* __restrict__ x, double* __restrict__ g, double* __restrict__ jac,
                       int flags, double* stage, int family, int lane) {
  const char* blob = reinterpret_cast<const char*>(w.blob);
  const DevStruct* S = reinterpret_cast<const DevStruct*>(blob);
  const double* xp = x + w.x_off;
  double* gp = g + w.g_off;
  double* jp = jac + w.j_off;
  const bool want_g = flags & 1, want_j = flags & 2;
  if (family == 0) {
    // The first 64 rows sit at a fixed offset behind the header (device_tables.h, node head): their load does not wait for a
    // header field.  One chunk body, called for the head records and -- structures with more than 64 rows -- for the rest.
    auto chunk = [&](const TerrainRow tr, int r0, int nr) {
      const int cnt = min(64, nr - r0);
      double* dst = jp + S->nnz_terrain + 3 * r0;
      const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
      if (lane < cnt) {
        const double px = xp[tr.idx], py = xp[tr.idx + tr.stride], pz = xp[tr.idx + 2 * tr.stride];
        const Terr t = terrain_eval(S, S->terrain_id, S->flat_height, px, py);
        if (want_g) gp[S->row_terrain + r0 + lane] = pz - t.h;
        if (want_j) {
          stage[par + 3 * lane + 0] = -t.hx;
          stage[par + 3 * lane + 1] = -t.hy;
          stage[par + 3 * lane + 2] = 1.0;
        }
      }
      if (want_j) copy_out(dst, stage, 3 * cnt, par, lane);  // single wave: LDS accesses are ordered
    };
    const TerrainRow tr0 = tbl<TerrainRow>(blob, kNodeHeadTerrainOff)[lane];
    const int nr = S->n_terrain_rows;
    if (nr > 0) chunk(tr0, 0, nr);
    if (nr > 64) {
      const TerrainRow* rows = tbl<TerrainRow>(blob, S->o_terrain_rows);
      for (int r0 = 64; r0 < nr; r0 += 64) chunk(rows[min(r0 + lane, nr - 1)], r0, nr);
    }
  } else if (family == 1) {
    auto chunk = [&](const ForceNode fn, int i0, int nn) {
      const int cnt = min(64, nn - i0);
      double* dst = jp + S->nnz_force + 25 * i0;
      const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
      if (lane < cnt) force_item(S, fn, xp, gp + S->row_force + 5 * (i0 + lane), stage + par + 25 * lane, want_g, want_j);
      if (want_j) copy_out(dst, stage, 25 * cnt, par, lane);
    };
    const ForceNode fn0 = tbl<ForceNode>(blob, kNodeHeadForceOff)[lane];   // (fixed offset, see the terrain family)
    const int nn = S->n_force_nodes;
    if (nn > 0) chunk(fn0, 0, nn);
    if (nn > 64) {
      const ForceNode* nodes = tbl<ForceNode>(blob, S->o_force_nodes);
      for (int i0 = 64; i0 < nn; i0 += 64) chunk(nodes[min(i0 + lane, nn - 1)], i0, nn);
    }
  } else if (family == 2) {
    // splineacc-base-lin | splineacc-base-ang (spline_acc_constraint.cc:49-81): lane = row (set, junction j,
    // dim d); its six variables are {p,v}_d of base nodes j, j+1, j+2 and the six Jacobian values depend on
    // the polynomial durations only (AccJunction), so g = sum c_i x_i.
    const AccJunction* acc = tbl<AccJunction>(blob, S->o_acc);
    const int per_set = 3 * S->n_junctions, nr = 2 * per_set;
    for (int r0 = 0; r0 < nr; r0 += 64) {
      const int cnt = min(64, nr - r0);
      double* dst = jp + S->nnz_acc + 6 * r0;
      const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
      if (lane < cnt) {
        const int r = r0 + lane;
        const int which = r >= per_set, rem = r - which * per_set;
        const int j = rem / 3, d = rem - 3 * j;
        const AccJunction a = acc[j];
        const double* xb = xp + (which ? S->off_base_ang : 0) + 6 * j + d;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          v += a.c[i] * xb[3 * i];
          if (want_j) stage[par + 6 * lane + i] = a.c[i];
        }
        if (want_g) gp[S->row_acc + r] = v;
      }
      if (want_j) copy_out(dst, stage, 6 * cnt, par, lane);
    }
    // baseMotion (base_motion_constraint.cc:60-90): lane = time node; rows AX..AZ = base-ang position,
    // LX..LZ = base-lin position, every row holds the four position weights of its dimension
    const BaseNode* bm = tbl<BaseNode>(blob, S->o_bm);
    const int nb = S->n_bm_nodes;
    for (int k0 = 0; k0 < nb; k0 += 16) {  // 16 nodes x 24 values fit the family's 384-value image
      const int cnt = min(16, nb - k0);
      double* dst = jp + S->nnz_bm + 24 * k0;
      const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
      if (lane < cnt) {
        const BaseNode n = bm[k0 + lane];
        double wP[4];
        hermite_pos(n.t, n.iT, wP);
        const double* xa = xp + S->off_base_ang + n.q6;
        const double* xl = xp + n.q6;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (want_g) {
            double* g6 = gp + S->row_bm + 6 * (k0 + lane);
            g6[d] = wP[0] * xa[d] + wP[1] * xa[3 + d] + wP[2] * xa[6 + d] + wP[3] * xa[9 + d];
            g6[3 + d] = wP[0] * xl[d] + wP[1] * xl[3 + d] + wP[2] * xl[6 + d] + wP[3] * xl[9 + d];
          }
          if (want_j) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              stage[par + 24 * lane + 4 * d + j] = wP[j];
              stage[par + 24 * lane + 12 + 4 * d + j] = wP[j];
            }
          }
        }
      }
      if (want_j) copy_out(dst, stage, 24 * cnt, par, lane);
    }
  } else {
    // swing-ee-motion_e (swing_constraint.cc:58-121): lane = swing node, rows {x pos, x vel, y pos, y vel},
    // columns {previous node, this node, next node}
    const SwingNode* sw = tbl<SwingNode>(blob, S->o_swing_nodes);
    const int nn = S->n_swing_nodes;
    const double it = S->inv_t_swing;
    for (int i0 = 0; i0 < nn; i0 += 64) {
      const int cnt = min(64, nn - i0);
      double* dst = jp + S->nnz_swing + 12 * i0;
      const int par = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
      if (lane < cnt) {
        const SwingNode sn = sw[i0 + lane];
        const int pi[2] = {sn.prev_x, sn.prev_y}, ni[2] = {sn.next_x, sn.next_y};
#pragma unroll
        for (int dim = 0; dim < 2; ++dim) {
          const double prev = xp[pi[dim]], next = xp[ni[dim]];
          const double pos = xp[sn.cur + 2 * dim], vel = xp[sn.cur + 2 * dim + 1];
          const double distance = next - prev;
          if (want_g) {
            double* g4 = gp + S->row_swing + 4 * (i0 + lane);
            g4[2 * dim] = pos - (prev + 0.5 * distance);
            g4[2 * dim + 1] = vel - distance * it;
          }
          if (want_j) {
            double* st = stage + par + 12 * lane + 6 * dim;
            st[0] = -0.5; st[1] = 1.0; st[2] = -0.5;
            st[3] = it;   st[4] = 1.0; st[5] = -it;
          }
        }
      }
      if (want_j) copy_out(dst, stage, 12 * cnt, par, lane);
    }
    // totalduration-<e> (total_duration_constraint.cc:50-72): sum of the optimised phase durations
    if (S->timings) {
      const PhaseTables* PT = tbl<PhaseTables>(blob, S->o_phase);
      if (lane < S->n_ee) {
        int voff = 0;
        for (int e = 0; e < lane; ++e) voff += PT->n_phases[e] - 1;
        const int ns = PT->n_phases[lane] - 1;
        double sum = 0.0;
        for (int i = 0; i < ns; ++i) {
          sum += xp[PT->off_sched[lane] + i];
          if (want_j) jp[PT->nnz_total + voff + i] = 1.0;
        }
        if (want_g) gp[PT->row_total + lane] = sum;
      }
    }
  }
}

__global__ __launch_bounds__(256, 4) void node_kernel(const NodeWork* __restrict__ work, const double* __restrict__ x,
                                                   double* __restrict__ g, double* __restrict__ jac, int flags) {
  __shared__ __attribute__((aligned(16))) double stage_all[kStageTerrain + kStageForce + kStageAcc + kStageSwing];
  const int family = threadIdx.x >> 6;  // wave-uniform
  // (one call site per CONSTANT family: each wave's code holds its own family's body only)
  const NodeWork w = work[blockIdx.x];
  const int lane = threadIdx.x & 63;
  if (family == 0) node_body(w, x, g, jac, flags, stage_all + kNodeStageOff[0], 0, lane);
  else if (family == 1) node_body(w, x, g, jac, flags, stage_all + kNodeStageOff[1], 1, lane);
  else if (family == 2) node_body(w, x, g, jac, flags, stage_all + kNodeStageOff[2], 2, lane);
  else node_body(w, x, g, jac, flags, stage_all + kNodeStageOff[3], 3, lane);
}

// Batches whose problems carry terrain-* and force-* sets only (the hot-path sets): workgroups of those two families.
// The kernel is one chain of dependent loads per wave (work item -> header -> rows -> x) and lives on occupancy; with
// four-wave workgroups two waves of each would find no work, i.e. 8 busy waves per CU instead of 16
// (A/B on one box, 8192 C3 problems: 0.054 -> see DESIGN 6.0).
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4))) void node_kernel2(const NodeWork* __restrict__ work, const double* __restrict__ x,
                                                    double* __restrict__ g, double* __restrict__ jac, int flags) {
  __shared__ __attribute__((aligned(16))) double stage_all[kStageTerrain + kStageForce];
  const int family = threadIdx.x >> 6;  // wave-uniform
  // (two call sites with a CONSTANT family: the bodies of the other families are not compiled into this kernel, and the
  // register allocation is that of terrain / force alone)
  if (family == 0) node_body(work[blockIdx.x], x, g, jac, flags, stage_all + kNodeStageOff[0], 0, threadIdx.x & 63);
  else node_body(work[blockIdx.x], x, g, jac, flags, stage_all + kNodeStageOff[1], 1, threadIdx.x & 63);
}

// ---------------------------------------------------------------- node-based sets of large batches: persistent chunks
// node_kernel is one chain of dependent loads per wave (work item -> header -> records -> x) and lives on occupancy alone
// (PMC: waiting 0.80-0.85 of the wave cycles).  For batches of more than a few thousand problems the same rows are evaluated
// by PERSISTENT waves instead: one wave = one family, walking that family's chunk list (FamWork) with a stride, the
// records of the chunk two items ahead and its x values one item ahead already in flight while the current chunk is
// computed and streamed out.  Every prefetch is unconditional with a clamped index (no phi on in-flight values, see
// rom_body).  Same arithmetic as node_body, statement by statement, so the same bits.
template <int FAM>
struct FamIn {
  int32_t a[6];        // the lane's record: TerrainRow | ForceNode | (unused) | SwingNode
  double c[6];         // family 2: the junction's six coefficients
  double v[8];         // x values: 3 | 5 | 6 | 8
};
template <int FAM>
TWR_DEV void fam_load_rec(const FamWork& w, int lane, FamIn<FAM>& in) {
  const int i = min(lane, w.cnt - 1);
  if (FAM == 0 || FAM == 1) {
    const int2 r = gptr<int2>(w.table)[i];
    in.a[0] = r.x; in.a[1] = r.y;
  } else if (FAM == 2) {
    const int rem = (w.i0 + i) % w.aux0;
    const TWR_GLOBAL double* a = reinterpret_cast<const TWR_GLOBAL double*>(w.table) + 6 * (rem / 3);
#pragma unroll
    for (int q = 0; q < 6; ++q) in.c[q] = a[q];
  } else {
    const TWR_GLOBAL int32_t* r = reinterpret_cast<const TWR_GLOBAL int32_t*>(w.table) + 6 * i;
#pragma unroll
    for (int q = 0; q < 5; ++q) in.a[q] = r[q];
  }
}
template <int FAM>
TWR_DEV void fam_load_x(const FamWork& w, int lane, const double* __restrict__ x, FamIn<FAM>& in) {
  const double* xp = x + w.x_off;
  if (FAM == 0) {        // TerrainRow {idx, stride}
    in.v[0] = xp[in.a[0]]; in.v[1] = xp[in.a[0] + in.a[1]]; in.v[2] = xp[in.a[0] + 2 * in.a[1]];
  } else if (FAM == 1) { // ForceNode {fidx, hidx}
    in.v[0] = xp[in.a[0]]; in.v[1] = xp[in.a[0] + 2]; in.v[2] = xp[in.a[0] + 4];
    in.v[3] = xp[in.a[1]]; in.v[4] = xp[in.a[1] + 1];
  } else if (FAM == 2) {
    const int r = w.i0 + min(lane, w.cnt - 1);
    const int which = r >= w.aux0, rem = r - which * w.aux0;
    const int j = rem / 3, d = rem - 3 * j;
    const double* xb = xp + (which ? w.aux1 : 0) + 6 * j + d;
#pragma unroll
    for (int q = 0; q < 6; ++q) in.v[q] = xb[3 * q];
  } else {               // SwingNode {cur, prev_x, prev_y, next_x, next_y}
    in.v[0] = xp[in.a[1]]; in.v[1] = xp[in.a[3]]; in.v[2] = xp[in.a[0]]; in.v[3] = xp[in.a[0] + 1];
    in.v[4] = xp[in.a[2]]; in.v[5] = xp[in.a[4]]; in.v[6] = xp[in.a[0] + 2]; in.v[7] = xp[in.a[0] + 3];
  }
}
// Items per chunk, Jacobian values and constraint values per item, store instructions of the copy-out (128 doubles each:
// the largest chunk + parity shift), g store instructions (64 values each)
template <int FAM>
struct FamShape {
  static constexpr int kItems = FAM == 1 ? 32 : 64;
  static constexpr int kPer = FAM == 0 ? 3 : (FAM == 1 ? 25 : (FAM == 2 ? 6 : 12));
  static constexpr int kRows = FAM == 0 ? 1 : (FAM == 1 ? 5 : (FAM == 2 ? 1 : 4));
  static constexpr int kNit = (kItems * kPer + 1 + 2 + 127) / 128;
  static constexpr int kGit = (kItems * kRows + 63) / 64;
};
// The chunk's values go to the LDS image (Jacobian) and to `gst` (constraint values); fam_store then streams both out with a
// COMPILE-TIME number of store instructions and no store inside a divergent branch -- like rom_body / dyn_body, and for the
// same reason: with the run-time copy-out loop and the lanes' own g stores of the first form of this kernel the compiler
// could not count the stores behind the prefetched loads, and every chunk began with s_waitcnt vmcnt(0), i.e. with the
// drain of the previous chunk's stores (A/B on one box with the work items three chunks ahead as well: 0.078 -> 0.069-0.074 ms per
// 8192 problems with towr's default list).
template <int FAM, bool WANT_G, bool WANT_J>
TWR_DEV void fam_compute(const FamWork& w, const FamIn<FAM>& in, double* stage, double* gst, int par, int lane) {
  const DevStruct* S = reinterpret_cast<const DevStruct*>(w.blob);
  if (lane < w.cnt) {
    if (FAM == 0) {
      const Terr t = terrain_eval(S, S->terrain_id, S->flat_height, in.v[0], in.v[1]);
      if (WANT_G) gst[lane] = in.v[2] - t.h;
      if (WANT_J) {
        stage[par + 3 * lane + 0] = -t.hx;
        stage[par + 3 * lane + 1] = -t.hy;
        stage[par + 3 * lane + 2] = 1.0;
      }
    } else if (FAM == 1) {
      const double f[3] = {in.v[0], in.v[1], in.v[2]};
      force_core(S, f, in.v[3], in.v[4], gst + 5 * lane, stage + par + 25 * lane, WANT_G, WANT_J);
    } else if (FAM == 2) {
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        v += in.c[q] * in.v[q];
        if (WANT_J) stage[par + 6 * lane + q] = in.c[q];
      }
      if (WANT_G) gst[lane] = v;
    } else {
      const double it = w.inv_t_swing;
#pragma unroll
      for (int dim = 0; dim < 2; ++dim) {
        const double prev = in.v[4 * dim], next = in.v[4 * dim + 1], pos = in.v[4 * dim + 2], vel = in.v[4 * dim + 3];
        const double distance = next - prev;
        if (WANT_G) {
          gst[4 * lane + 2 * dim] = pos - (prev + 0.5 * distance);
          gst[4 * lane + 2 * dim + 1] = vel - distance * it;
        }
        if (WANT_J) {
          double* st = stage + par + 12 * lane + 6 * dim;
          st[0] = -0.5; st[1] = 1.0; st[2] = -0.5;
          st[3] = it;   st[4] = 1.0; st[5] = -it;
        }
      }
    }
  }
}
template <int FAM, bool WANT_G, bool WANT_J>
TWR_DEV void fam_store(const FamWork& w, double* __restrict__ g, double* __restrict__ jac, const double* stage, const double* gst,
                       int par, int lane) {
  typedef FamShape<FAM> Sh;
  // (non-temporal: A/B on one box, towr's default list, 8192 problems of one structure: 0.067 -> 0.063 ms)
  if (WANT_J) copy_out_fixed<Sh::kNit, Sh::kNit, true>(jac + w.j_off, stage, Sh::kPer * w.cnt, par, lane);   // single wave: LDS accesses are ordered
  if (WANT_G) {   // clamped lanes instead of predicates (see copy_out_fixed)
    double* gp = g + w.g_off;
    const int last_g = Sh::kRows * w.cnt - 1;
#pragma unroll
    for (int t = 0; t < Sh::kGit; ++t) gp[min(lane + 64 * t, last_g)] = gst[min(lane + 64 * t, last_g)];
  }
}
// A use of every loaded value of the prologue that the compiler can see: it then waits for the prologue's loads IN the prologue
// and the loop is entered with none in flight.  (Otherwise the one wait in front of a chunk's math has to serve both ways
// into the loop header -- from the prologue, where the x loads of the first chunk are the youngest loads, and from the
// latch, where a counted number of stores is behind them -- and becomes s_waitcnt vmcnt(0): the store drain again.)
template <int FAM>
TWR_DEV void fam_pin(FamIn<FAM>& in, bool with_x) {
  constexpr int kA = FAM == 2 ? 0 : (FAM == 3 ? 5 : 2), kC = FAM == 2 ? 6 : 0, kV = FAM == 0 ? 3 : (FAM == 1 ? 5 : (FAM == 2 ? 6 : 8));
#pragma unroll
  for (int q = 0; q < kA; ++q) asm volatile("" : "+v"(in.a[q]));
#pragma unroll
  for (int q = 0; q < kC; ++q) asm volatile("" : "+v"(in.c[q]));
  if (with_x) {
#pragma unroll
    for (int q = 0; q < kV; ++q) asm volatile("" : "+v"(in.v[q]));
  }
}
template <int FAM, bool WANT_G, bool WANT_J>
TWR_DEV void fam_body(const FamWork* __restrict__ work, int n_work, const double* __restrict__ x, double* __restrict__ g,
                      double* __restrict__ jac, double* stage, double* gst, int lane, int i, int stride) {
  if (i >= n_work) return;
  const int last = i + (n_work - 1 - i) / stride * stride;
  FamWork w0 = work[i], w1 = work[min(i + stride, last)], w2 = work[min(i + 2 * stride, last)];
  FamIn<FAM> in0, in1;
  fam_load_rec<FAM>(w0, lane, in0);
  fam_load_rec<FAM>(w1, lane, in1);
  fam_load_x<FAM>(w0, lane, x, in0);
  fam_pin<FAM>(in0, true);
  fam_pin<FAM>(in1, false);
  for (; i <= last; i += stride) {
    const FamWork w3 = work[min(i + 3 * stride, last)];   // work item three chunks ahead (the list is a cold stream: a scalar
                                                          // load that is used in the iteration it is issued in costs its HBM latency)
    FamIn<FAM> in2;
    fam_load_rec<FAM>(w2, lane, in2);          // records two chunks ahead
    fam_load_x<FAM>(w1, lane, x, in1);         // x one chunk ahead (its records arrived an iteration ago)
    const int par = (int)((reinterpret_cast<uintptr_t>(jac + w0.j_off) >> 3) & 1);
    fam_compute<FAM, WANT_G, WANT_J>(w0, in0, stage, gst, par, lane);
    fam_store<FAM, WANT_G, WANT_J>(w0, g, jac, stage, gst, par, lane);
    w0 = w1; in0 = in1;
    w1 = w2; in1 = in2;
    w2 = w3;
  }
}
// blocks [0, g0) walk the terrain chunks, the next g1 the force chunks, then splineacc, then swing (any count may be 0)
constexpr int kForceChunk = 32;                                   // force nodes per chunk (25 values each)
constexpr int kStageChunk = kForceChunk * 25 + 2;                // >= 64 x 12 + 2 (swing), 64 x 6 + 2, 64 x 3 + 2
constexpr int kStageChunkG = 64 * 4;                             // constraint values of a chunk: <= 64 x 4 (swing), 32 x 5 (force)
static_assert(kStageChunk >= kStageSwing && kStageChunk >= kStageAcc && kStageChunk >= kStageTerrain, "chunk image");
static_assert(FamShape<1>::kItems == kForceChunk && FamShape<0>::kNit * 128 <= kStageChunk + 127 && FamShape<1>::kNit * 128 <= kStageChunk + 127 &&
              FamShape<2>::kNit * 128 <= kStageChunk + 127 && FamShape<3>::kNit * 128 <= kStageChunk + 127, "chunk copy-out inside the image");
template <bool WANT_G, bool WANT_J>
__global__ __launch_bounds__(64, 4) void node_chunk_kernel(const FamWork* __restrict__ f0, int n0, int g0, const FamWork* __restrict__ f1,
                                                        int n1, int g1, const FamWork* __restrict__ f2, int n2, int g2,
                                                        const FamWork* __restrict__ f3, int n3, int g3, const double* __restrict__ x,
                                                        double* __restrict__ g, double* __restrict__ jac) {
  __shared__ __attribute__((aligned(16))) double stage[kStageChunk + kStageChunkG];
  double* gst = stage + kStageChunk;
  const int lane = threadIdx.x;
  int b = blockIdx.x;
  if (b < g0) return fam_body<0, WANT_G, WANT_J>(f0, n0, x, g, jac, stage, gst, lane, b, g0);
  b -= g0;
  if (b < g1) return fam_body<1, WANT_G, WANT_J>(f1, n1, x, g, jac, stage, gst, lane, b, g1);
  b -= g1;
  if (b < g2) return fam_body<2, WANT_G, WANT_J>(f2, n2, x, g, jac, stage, gst, lane, b, g2);
  b -= g2;
  fam_body<3, WANT_G, WANT_J>(f3, n3, x, g, jac, stage, gst, lane, b, g3);
}

// Small and mid-size batches: the three kernels above as ONE launch, so that their pipeline fills and drains overlap
// instead of adding up (three dependent launches cost ~14 us of a 19-us step at 32 candidates).  Workgroups of two waves
// and 40 KB: blocks [0, g_rom) take the rom role (wave 0 only; the image is 39 KB), the next g_dyn blocks the dyn role
// (both waves, one 20-KB half each), the rest the node role (two families per block) -- the residency per CU of each
// role is that of its own kernel, and blocks are dispatched in this order, so a later role starts as the earlier drains.
template <int ROM_NIT, bool WANT_G, bool WANT_J, int XC, bool NT>
__global__ __launch_bounds__(128, 2) void eval_fused_kernel(const RomWork* __restrict__ rom, int n_rom, int g_rom,
                                                            const DynWork* __restrict__ dyn, int n_dyn, int g_dyn,
                                                            const NodeWork* __restrict__ node, const double* __restrict__ x,
                                                            double* __restrict__ g, double* __restrict__ jac,
                                                            double* __restrict__ dump) {
  constexpr int kFusedLds = 2 * kDynLds >= kRomLds ? 2 * kDynLds : kRomLds;   // (the second case: experiment builds with small dyn images)
  static_assert(kDynLds >= kStageForce, "LDS of the fused kernel");
  __shared__ __attribute__((aligned(16))) double stage[kFusedLds];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int b = blockIdx.x;
  if (b < g_rom) {
    if (wave == 0) rom_body<ROM_NIT, WANT_G, WANT_J, NT>(rom, n_rom, x, g, jac, stage, lane, b, g_rom);
    return;
  }
  b -= g_rom;
  if (b < g_dyn) {
    // Which slices a block's two waves take: the work list is ordered so that position j wants XCD j % 8 (all slices of one
    // problem on one XCD, capi.cc), and blocks are dealt round-robin over the XCDs -- so block 8 q + r takes positions
    // 16 q + r and 16 q + 8 + r: every position a block ever visits is r modulo 8.  (With positions 2 b, 2 b + 1 the
    // slices of a problem ran on two XCDs, and its x and tables were fetched by both.)  Needs g_dyn to be a multiple
    // of 8 (the launcher rounds it down); tiny grids keep the plain mapping.
    const int first = (g_dyn & 7) == 0 ? ((b >> 3) << 4) + (wave << 3) + (b & 7) : 2 * b + wave;
    dyn_body<WANT_G, WANT_J, XC, NT>(dyn, n_dyn, x, g, jac, dump, stage + wave * kDynLds, lane, first, 2 * g_dyn);
    return;
  }
  b -= g_dyn;
  const int family = 2 * (b & 1) + wave;
  node_body(node[b >> 1], x, g, jac, (WANT_G ? 1 : 0) | (WANT_J ? 2 : 0), stage + wave * kDynLds, family, lane);
}

// ---------------------------------------------------------------- values only: one lane per time node
// TWR_EVAL_VALUES -- Ipopt's eval_g (every line-search trial point), a planner's scoring step -- for problems with fixed
// timings.  The values-only instantiations of the Jacobian bodies were bound by vector-instruction issue, not by memory
// (72 % / 55 % of the issue slots at 0.12 of the HBM roof).  The Jacobian kernels' cuts -- "rangeofmotion-e":
// one slice per end-effector, lane = time node; "dynamic": a quad of lanes per time node -- evaluate the base splines and the
// rotation (three sin / cos pairs) of a time node once per END-EFFECTOR resp. per LANE OF THE QUAD; without an image to
// assemble one lane can take a time node for ALL end-effectors (device_tables.h FlatNode / FlatPoly / FlatWork).
// What made the first form of this slower for "dynamic" (0.205 vs 0.142 ms per 8192 C3 problems) was the ~120 per-lane global
// gathers of x behind eight 40-byte polynomial records; here the wave copies the problem's x and the item's windows of
// polynomial records into LDS (coalesced, one round trip behind the work item) and a candidate is ONE LDS read at a byte
// offset the host prepared (zero pair for candidates that are not variables, p0's variable for p1 of a stance polynomial):
// no selects, no global gathers.
// LDS of a wave (dynamic size, the launcher knows the largest problem of the batch): [ polynomial windows: 2 kMaxEE x
// kFlatWindow records | zero pair | x ]
typedef uint32_t twr_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t twr_u2 __attribute__((ext_vector_type(2)));
struct FlatIn {            // what a lane fetches for an item: its time node and one polynomial record of the windows, as whole
  twr_u4 na, nb;           // dwords (no sub-dword struct fields across the loop back-edge): FlatNode = na | nb | nc
  twr_u2 nc;
  twr_u4 pa, pb, pc;       // FlatPoly
};
struct FlatNodeR {         // FlatNode unpacked (registers)
  double t, tb, iTb;
  int q6;
  uint32_t qm, qf;         // four bytes each
};
TWR_DEV FlatNodeR flat_node(const FlatIn& in) {
  static_assert(offsetof(FlatNode, q6) == 24 && offsetof(FlatNode, qm) == 28 && offsetof(FlatNode, qf) == 32, "FlatNode dwords");
  FlatNodeR n;
  n.t = __hiloint2double((int)in.na.y, (int)in.na.x);
  n.tb = __hiloint2double((int)in.na.w, (int)in.na.z);
  n.iTb = __hiloint2double((int)in.nb.y, (int)in.nb.x);
  n.q6 = (int)in.nb.z;
  n.qm = in.nb.w;
  n.qf = in.nc.x;
  return n;
}
// The work record of an item: wave-uniform, read with scalar loads (one round trip: everything an item needs that is the
// same for all its lanes is IN the record, nothing behind a pointer of it); a field is addressed by its dword.
struct FlatRec {
  const TWR_CONST uint32_t* p;
  TWR_DEV uint32_t u32(int k) const { return p[k]; }
  TWR_DEV int i32(int k) const { return (int)p[k]; }
  TWR_DEV uint64_t u64(int k) const { return ((uint64_t)p[k + 1] << 32) | p[k]; }
  TWR_DEV double f64(int k) const { return __hiloint2double((int)p[k + 1], (int)p[k]); }
};
TWR_DEV void flat_load(const FlatRec& w, int lane, FlatIn& in) {
  {
    const TWR_GLOBAL twr_u2* nd = reinterpret_cast<const TWR_GLOBAL twr_u2*>(gptr<FlatNode>(w.u64(kFwNodes)) + min(lane, w.i32(kFwCnt) - 1));   // (clamped:
    const twr_u2 a = nd[0], b = nd[1], c = nd[2], d = nd[3];   // every lane loads, the tail lanes store nothing; 8-byte aligned records)
    in.na = twr_u4{a.x, a.y, b.x, b.y};
    in.nb = twr_u4{c.x, c.y, d.x, d.y};
    in.nc = nd[4];
  }
}
TWR_DEV void flat_load_polys(const FlatRec& w, int lane, FlatIn& in) {
  // lane 8 s + j: record j of spline s's window (clamped to the window: no predicates)
  const int s = lane >> 3, j = lane & 7;
  const uint64_t st = lane < 32 ? w.u64(kFwStart) : w.u64(kFwStart + 2);
  const int first = (int)((st >> (16 * (s & 3))) & 0xFFFFu), cnt = (int)((w.u64(kFwCount) >> (8 * s)) & 0xFFu);
  const TWR_GLOBAL twr_u4* rec = reinterpret_cast<const TWR_GLOBAL twr_u4*>(gptr<FlatPoly>(w.u64(kFwPolys)) + first + min(j, max(cnt, 1) - 1));
  in.pa = rec[0];
  in.pb = rec[1];
  in.pc = rec[2];
}
TWR_DEV void flat_stage_polys(const FlatIn& in, char* lds, int lane) {
  twr_u4* dst = reinterpret_cast<twr_u4*>(lds + lane * (int)sizeof(FlatPoly));
  dst[0] = in.pa;
  dst[1] = in.pb;
  dst[2] = in.pc;
}
// x of a problem by the 256 threads of a group, NX doubles per thread (the launcher picks the smallest instantiation that
// covers the largest problem)
constexpr int kFlatGroup = 4;   // items (= waves) per workgroup
template <int NX>
TWR_DEV void flat_load_x(const double* __restrict__ xp, int n_x, int tid, double xr[NX]) {
#pragma unroll
  for (int j = 0; j < NX; ++j) xr[j] = xp[min(64 * kFlatGroup * j + tid, n_x - 1)];
}
template <int NX>
TWR_DEV void flat_stage_x(const double xr[NX], int n_x, double* xs, int tid) {
  if (tid < 2) xs[tid] = 0.0;
#pragma unroll
  for (int j = 0; j < NX; ++j) xs[2 + min(64 * kFlatGroup * j + tid, n_x - 1)] = xr[j];   // (clamped like the loads: the threads past
}                                                                                        // the end write the last variable once more)
// A polynomial's record in registers: from the wave's window in LDS, or -- items of a coarse grid, whose time nodes hardly
// share polynomials (FlatWork::gather) -- fetched by the lane itself from the structure's table.
struct FlatPolyR {
  double t0, iT;
  uint32_t o[6];
};
template <bool GATHER>
TWR_DEV FlatPolyR flat_poly(const FlatRec& w, const char* __restrict__ lds, int spline, int local) {
  FlatPolyR r;
  if (GATHER) {
    const int first = (int)((w.u64(kFwStart + 2 * (spline >> 2)) >> (16 * (spline & 3))) & 0xFFFFu);   // (uniform)
    const TWR_GLOBAL twr_u4* rec = reinterpret_cast<const TWR_GLOBAL twr_u4*>(gptr<FlatPoly>(w.u64(kFwPolys)) + first + local);
    const twr_u4 a = rec[0], b = rec[1];
    const twr_u2 c = reinterpret_cast<const TWR_GLOBAL twr_u2*>(rec)[4];
    r.t0 = __hiloint2double((int)a.y, (int)a.x);
    r.iT = __hiloint2double((int)a.w, (int)a.z);
    r.o[0] = b.x; r.o[1] = b.y; r.o[2] = b.z; r.o[3] = b.w; r.o[4] = c.x; r.o[5] = c.y;
  } else {
    const char* rec = lds + (spline * kFlatWindow + local) * (int)sizeof(FlatPoly);
    const double2 ti = *reinterpret_cast<const double2*>(rec);
    const uint4 oa = reinterpret_cast<const uint4*>(rec)[1];
    const uint2 ob = reinterpret_cast<const uint2*>(rec)[4];
    r.t0 = ti.x; r.iT = ti.y;
    r.o[0] = oa.x; r.o[1] = oa.y; r.o[2] = oa.z; r.o[3] = oa.w; r.o[4] = ob.x; r.o[5] = ob.y;
  }
  return r;
}
// Spline::GetPoint (spline.cc:80-93) of an ee spline in Hermite basis form, position only
TWR_DEV void flat_point(const char* __restrict__ xs, const FlatPolyR& r, double t, double p[3]) {
  double w[4], X[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) X[c] = lds_f64(xs, (c & 1) ? r.o[c >> 1] >> 16 : r.o[c >> 1] & 0xFFFFu);
  hermite_pos(t - r.t0, r.iT, w);
#pragma unroll
  for (int d = 0; d < 3; ++d) p[d] = w[0] * X[d] + w[1] * X[3 + d] + w[2] * X[6 + d] + w[3] * X[9 + d];
}
// Same formula as rom_item (RangeOfMotionConstraint::UpdateConstraintAtInstance, range_of_motion_constraint.cc:58-69).
// `lds`: the wave's polynomial windows; `xs`: the group's copy of x (zero pair first); `gs`: 192 doubles of the wave's own
template <bool GATHER>
TWR_DEV void flat_rom_math(const FlatRec& w, const FlatNodeR& n, double* __restrict__ g, const char* lds, const char* xs, double* gs, int lane) {
  const double* xv = reinterpret_cast<const double*>(xs) + 2;
  double* gp = g + (int64_t)w.u64(kFwG);
  const int n_ee = w.i32(kFwNee);
  // base splines: the twelve node values of the active polynomial are contiguous in x (nodes q, q + 1: p then v)
  double wP[4], c[3], e[3];
  hermite_pos(n.tb, n.iTb, wP);
  {
    const double* xl = xv + w.i32(kFwOffLin) + n.q6;
    const double* xa = xv + w.i32(kFwOffAng) + n.q6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      c[d] = wP[0] * xl[d] + wP[1] * xl[3 + d] + wP[2] * xl[6 + d] + wP[3] * xl[9 + d];
      e[d] = wP[0] * xa[d] + wP[1] * xa[3 + d] + wP[2] * xa[6 + d] + wP[3] * xa[9 + d];
    }
  }
  Rot ro;
  rotation(e, ro);
#pragma unroll
  for (int ee = 0; ee < kMaxEE; ++ee)
    if (ee < n_ee) {   // g = b_R_w (p_ee - c)
      double p[3], v[3], gv[3];
      flat_point(xs, flat_poly<GATHER>(w, lds, 2 * ee, (n.qm >> (8 * ee)) & 0xFFu), n.t, p);
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = p[d] - c[d];
      matTvec(ro.R, v, gv);
      // the 3 cnt values of this end-effector's rows are contiguous in g: through LDS, then whole lines (a lane storing its own
      // three values writes 16 + 8 of every 24 bytes per instruction)
      gs[3 * lane] = gv[0];
      gs[3 * lane + 1] = gv[1];
      gs[3 * lane + 2] = gv[2];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      double* go = gp + w.i32(kFwRowRom + ee) + 3 * w.i32(kFwK0);
      const int last_g = 3 * w.i32(kFwCnt) - 1;
#pragma unroll
      for (int t = 0; t < 3; ++t) go[min(lane + 64 * t, last_g)] = gs[min(lane + 64 * t, last_g)];
    }
}
// DynamicConstraint::UpdateConstraintAtInstance (dynamic_constraint.cc:59-77) with SingleRigidBodyDynamics::GetDynamicViolation
// (single_rigid_body_dynamics.cc:76-101) and the EulerConverter quantities (euler_converter.cc:58-83,133-166,207-221) of one time
// node on ONE lane -- the statements of dyn2_front / dyn2_back without the quad.
template <bool GATHER>
TWR_DEV void flat_dyn_math(const FlatRec& w, const FlatNodeR& n, double* __restrict__ g, const char* lds, const char* xs, double* gs_rom, int lane) {
  const double* xv = reinterpret_cast<const double*>(xs) + 2;
  const int n_ee = w.i32(kFwNee);
  const double* bl = xv + w.i32(kFwOffLin) + n.q6;   // base splines: the twelve node values of the active polynomial
  const double* ba = xv + w.i32(kFwOffAng) + n.q6;
  // base position and rotation first: with coinciding grids (FlatWork::with_rom) the lane evaluates "rangeofmotion-e" of its
  // time node along the way -- g = b_R_w (p_e - c), range_of_motion_constraint.cc:58-69 -- from the same base point, rotation and
  // end-effector positions
  const bool with_rom = w.i32(kFwWithRom) != 0;
  double* gp = g + (int64_t)w.u64(kFwG);
  const int last_rom = 3 * w.i32(kFwCnt) - 1;
  Rot ro;
  // force and torque sums over the end-effectors (single_rigid_body_dynamics.cc:81-88)
  double F[3] = {0.0, 0.0, 0.0}, tau[3] = {0.0, 0.0, 0.0};
  {
    double wP[4], c[3], e[3];
    hermite_pos(n.tb, n.iTb, wP);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      c[d] = wP[0] * bl[d] + wP[1] * bl[3 + d] + wP[2] * bl[6 + d] + wP[3] * bl[9 + d];
      e[d] = wP[0] * ba[d] + wP[1] * ba[3 + d] + wP[2] * ba[6 + d] + wP[3] * ba[9 + d];
    }
    rotation(e, ro);
#pragma unroll
    for (int ee = 0; ee < kMaxEE; ++ee)
      if (ee < n_ee) {
        double p[3], f[3], rv[3], t3[3];
        flat_point(xs, flat_poly<GATHER>(w, lds, 2 * ee, (n.qm >> (8 * ee)) & 0xFFu), n.t, p);
        flat_point(xs, flat_poly<GATHER>(w, lds, 2 * ee + 1, (n.qf >> (8 * ee)) & 0xFFu), n.t, f);
#pragma unroll
        for (int d = 0; d < 3; ++d) rv[d] = c[d] - p[d];
        cross3(f, rv, t3);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          F[d] += f[d];
          tau[d] += t3[d];
        }
        if (with_rom) {
          double v[3], gv[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) v[d] = p[d] - c[d];
          matTvec(ro.R, v, gv);
          gs_rom[3 * lane] = gv[0];
          gs_rom[3 * lane + 1] = gv[1];
          gs_rom[3 * lane + 2] = gv[2];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          double* go = gp + w.i32(kFwRowRom + ee) + 3 * w.i32(kFwK0);
#pragma unroll
          for (int t = 0; t < 3; ++t) go[min(lane + 64 * t, last_rom)] = gs_rom[min(lane + 64 * t, last_rom)];
        }
      }
  }
  double cdd[3], ed[3], edd[3];
  {
    double wP[4], wV[4], wA[4];
    hermite_all(n.tb, n.iTb, wP, wV, wA);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      cdd[d] = wA[0] * bl[d] + wA[1] * bl[3 + d] + wA[2] * bl[6 + d] + wA[3] * bl[9 + d];
      ed[d] = wV[0] * ba[d] + wV[1] * ba[3 + d] + wV[2] * ba[6 + d] + wV[3] * ba[9 + d];
      edd[d] = wA[0] * ba[d] + wA[1] * ba[3 + d] + wA[2] * ba[6 + d] + wA[3] * ba[9 + d];
    }
  }
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
  double Iw6[6];  // I_w = R I_b R^T (single_rigid_body_dynamics.cc:91), symmetric: (00,01,02,11,12,22)
  {
    const double (&R)[3][3] = ro.R;
    double Ib[6], Tm[3][3];
#pragma unroll
    for (int i = 0; i < 6; ++i) Ib[i] = w.f64(kFwIb + 2 * i);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Tm[i][0] = R[i][0] * Ib[0] + R[i][1] * Ib[1] + R[i][2] * Ib[2];
      Tm[i][1] = R[i][0] * Ib[1] + R[i][1] * Ib[3] + R[i][2] * Ib[4];
      Tm[i][2] = R[i][0] * Ib[2] + R[i][1] * Ib[4] + R[i][2] * Ib[5];
    }
    Iw6[0] = Tm[0][0] * R[0][0] + Tm[0][1] * R[0][1] + Tm[0][2] * R[0][2];
    Iw6[1] = Tm[0][0] * R[1][0] + Tm[0][1] * R[1][1] + Tm[0][2] * R[1][2];
    Iw6[2] = Tm[0][0] * R[2][0] + Tm[0][1] * R[2][1] + Tm[0][2] * R[2][2];
    Iw6[3] = Tm[1][0] * R[1][0] + Tm[1][1] * R[1][1] + Tm[1][2] * R[1][2];
    Iw6[4] = Tm[1][0] * R[2][0] + Tm[1][1] * R[2][1] + Tm[1][2] * R[2][2];
    Iw6[5] = Tm[2][0] * R[2][0] + Tm[2][1] * R[2][1] + Tm[2][2] * R[2][2];
  }
  double Iw_wd[3], Iw_w[3], wxIw[3];
  symmul(Iw6, omd, Iw_wd);
  symmul(Iw6, om, Iw_w);
  cross3(om, Iw_w, wxIw);
  const double m = w.f64(kFwMass);
  // the 6 cnt values of the item are contiguous in g: through LDS (the polynomial windows are no longer needed), then whole lines
  double* gs = reinterpret_cast<double*>(const_cast<char*>(lds)) + 6 * lane;
  static_assert(kFlatPolyLds >= 64 * 6 * 8, "constraint values of a dynamic item in the window region");
#pragma unroll
  for (int i = 0; i < 3; ++i) gs[i] = Iw_wd[i] + wxIw[i] - tau[i];
  gs[3] = m * cdd[0] - F[0];
  gs[4] = m * cdd[1] - F[1];
  gs[5] = m * cdd[2] - F[2] + m * w.f64(kFwGravity);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  {
    const double* gl = reinterpret_cast<const double*>(lds);
    double* go = gp + w.i32(kFwRowDyn) + 6 * w.i32(kFwK0);
    const int last_g = 6 * w.i32(kFwCnt) - 1;
#pragma unroll
    for (int t = 0; t < 6; ++t) go[min(lane + 64 * t, last_g)] = gl[min(lane + 64 * t, last_g)];
  }
}
// One launch per values-only evaluation.  Workgroups of four waves: the first n_groups take a GROUP each -- four consecutive work
// records, items of ONE problem (twr_batch_create pads a problem's list to whole groups with empty items, cnt = 0) --, so that
// the problem's x is fetched and copied to LDS once per group by all 256 threads and every wave evaluates its own item on it
// (K = 200 with coinciding grids: one group per problem); the remaining workgroups take the node families of one problem each,
// one family per wave.  Per item: fetch (the lane's time node, the thread's share of x, one polynomial record of the windows
// -- one round trip behind the work record), copy to LDS, evaluate.
// LDS (dynamic, the launcher knows the largest problem of the batch): [ zero pair | x ] [ per wave: polynomial windows | 192
// constraint values ].
// (A persistent form of this -- waves looping over strided items, the fetches of item i + 1 in flight during the math of item
// i, the work record read by one vector load and v_readlane so that no scalar load sat in the loop -- was built, is parity-
// green and is NOT faster: 0.084-0.092 / 0.067-0.072 ms against 0.068-0.073 / 0.066 ms per 8192 C3 problems as single-wave
// workgroups, at three waves per SIMD instead of four because of the prefetch registers)
constexpr int kFlatWaveLds = kFlatPolyLds + 192 * 8;   // bytes of a wave's own region
static_assert(kFlatWaveLds >= kDynValuesLds * 8, "node_body stages its constraint values in a wave's region");
inline size_t flat_x_bytes(int max_n_x) { return sizeof(double) * (size_t)(2 + ((max_n_x + 1) & ~1)); }
inline size_t flat_lds_bytes(int max_n_x) { return flat_x_bytes(max_n_x) + (size_t)kFlatGroup * kFlatWaveLds; }
template <int NX>
__global__ __launch_bounds__(64 * kFlatGroup, 4) void eval_values_kernel(const FlatWork* __restrict__ flat, int n_groups, int x_bytes, const NodeWork* __restrict__ node,
                                                                         int node_families, const double* __restrict__ x, double* __restrict__ g) {
  extern __shared__ __attribute__((aligned(16))) double flat_stage[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* xs = reinterpret_cast<char*>(flat_stage);
  char* mine = xs + x_bytes + wave * kFlatWaveLds;
  int b = blockIdx.x;
  if (b >= n_groups) {
    if (wave < node_families)   // (values only: the node image is not touched)
      node_body(node[b - n_groups], x, g, nullptr, 1, reinterpret_cast<double*>(mine), wave, lane);
    return;
  }
  FlatRec w;
  w.p = cptr<uint32_t>(reinterpret_cast<uint64_t>(flat + (kFlatGroup * b + wave)));
  const int cnt = w.i32(kFwCnt);
  FlatIn in;
  double xr[NX];
  const bool gather = w.i32(kFwGather) != 0;   // (uniform) a coarse grid: every lane fetches its own polynomial records
  if (cnt > 0) flat_load(w, lane, in);
  if (cnt > 0 && !gather) flat_load_polys(w, lane, in);
  flat_load_x<NX>(x + (int64_t)w.u64(kFwX), w.i32(kFwNx), tid, xr);   // (the empty items of a group carry the problem's x as well)
  flat_stage_x<NX>(xr, w.i32(kFwNx), reinterpret_cast<double*>(xs), tid);
  if (cnt > 0 && !gather) flat_stage_polys(in, mine, lane);
  __syncthreads();
  if (cnt <= 0) return;
  double* gs = reinterpret_cast<double*>(mine + kFlatPolyLds);
  const bool dynamic = w.i32(kFwDynamic) != 0;
  if (gather) {
    if (dynamic) flat_dyn_math<true>(w, flat_node(in), g, mine, xs, gs, lane);
    else flat_rom_math<true>(w, flat_node(in), g, mine, xs, gs, lane);
  } else {
    if (dynamic) flat_dyn_math<false>(w, flat_node(in), g, mine, xs, gs, lane);
    else flat_rom_math<false>(w, flat_node(in), g, mine, xs, gs, lane);
  }
}

// ---------------------------------------------------------------- optimised timings (PhaseSpline) kernels
// With Parameters::OptimizePhaseDurations the active polynomial of every ee spline depends on x and every
// Jacobian row of an ee spline holds all variables of its set (phase_spline.cc:44-51), most of them
// explicit zeros; the duration columns come on top (dynamic_constraint.cc:107-113,
// range_of_motion_constraint.cc:106-108).  Three kernels: phase_locate_kernel resolves what depends on x in the index
// work (durations -> active polynomials, local times, current phase) into per-(time node, ee) records;
// dyn_phase_kernel and rom_phase_kernel then evaluate the same math as the fixed-timing kernels with more lanes per
// time node, assemble the EXPANDED rows of a few time nodes in LDS (clear, store every value at its final position)
// and stream them out, so every byte of the Jacobian -- zeros included -- is written to HBM exactly once.
// PhaseDurations::SetVariables (phase_durations.cc:77-103) + ConvertPhaseToPolyDurations
// (nodes_variables_phase_based.cc:73-84) of one ee, spread over the lanes of a wave: loads in parallel, only the
// duration sum is serial.
TWR_DEV void phase_poly_durations_wave(const PhaseTables* PT, const char* blob, const double* __restrict__ xp, int e,
                                       double* ph, double* md, int lane, int nthreads = 64) {
  const int ns = PT->n_phases[e] - 1;
  if (lane < ns) ph[lane] = xp[PT->off_sched[e] + lane];
  __syncthreads();
  if (lane == 0) {
    double sum = 0.0;
    for (int i = 0; i < ns; ++i) sum += ph[i];
    ph[ns] = PT->t_total[e] - sum;
  }
  __syncthreads();
  const PhasePoly* mp = tbl<PhasePoly>(blob, PT->o_mpoly[e]);
  for (int q = lane; q < PT->n_mpoly[e]; q += nthreads) md[q] = ph[mp[q].phase] / mp[q].n_in_phase;
  __syncthreads();
}

// ---- dynamic with optimised timings (dyn_phase_kernel)
// Persistent single-wave workgroups, software pipelined like rom_kernel.  The rows of "dynamic" now hold ALL variables
// of every ee set (C3: 1280 values = 10 KB per time node, ~75 % explicit zeros), so a work item is a "pass" of only FOUR
// consecutive time nodes (40 KB of Jacobian values) and a time node gets SIXTEEN lanes:
//   lane = 16 n + 4 e + j:   n = time node of the pass, e = end-effector, j = node value (p0, v0, p1, v1)
// Lane (n, e, j) loads the three dimensions of node value j of the base splines and of ee e's active polynomials and owns
// their Jacobian columns.  Spline points are sums over j (DPP quad sums), force / torque totals are sums over e (DPP row
// rotations by 4 and 8 lanes); the SRBD algebra is evaluated by every lane (the same instructions for all of them);
// lanes e < 3 produce Euler dimension e of the base-ang block, lanes e = 3 the base-lin block, the duration columns of
// ee e are spread over its four lanes.  What depends on x in the index work -- active polynomials, local times, current
// phase of every (time node, ee) -- comes from the pre-pass (phase_locate_kernel -> DynLoc); everything else is a
// constant of the structure: the byte offset of every value inside its time node is read off the CSR pattern on the
// host (PhasePutM / PhasePutF per polynomial, PhaseEe for the duration columns).  Per pass the wave clears the LDS image
// of the four expanded nodes, every lane stores its ~25 values at their final positions, and the wave streams the image
// to HBM with 16-byte coalesced stores: every byte of the Jacobian is written exactly once.
struct PDynRec {       // per lane: what depends on the work item only -- DynShared {tb, iTb, q6} and the DynLoc record,
                       // kept as whole dwords (unpacked with shifts where used)
  double tb, iTb;      // base spline: local time in the active polynomial, 1 / duration
  int32_t q6;
  double tm, Tm, tf, Tf;
  int32_t xbase_m, xbase_f;
  uint32_t slots_m[2], slots_f[2];
  uint32_t imjf;       // im | jf << 16
  uint32_t misc;       // cur | flags << 8 | np_m << 16 | np_f << 24
};
TWR_DEV double pdyn_f64(float lo, float hi) { return __hiloint2double((int)__float_as_uint(hi), (int)__float_as_uint(lo)); }
struct PDynIn {        // per lane: what depends on the record (second stage of the pipeline)
  double bl[3], ba[3], m[3], f[3];   // node value j (3 dimensions) of base-lin / base-ang and of ee e's polynomials
};
struct PDynPut {       // per lane: where its values go (loaded at the top of the pass, used at its end), whole dwords
  uint32_t pm[4], pf[6];             // 16-bit put offsets of candidates (j, D): index 2 D + r resp. 3 D + r
  int32_t ns, dur_ang[3], dur_lin[3];
  TWR_DEV uint32_t m(int q) const { return (pm[q >> 1] >> (16 * (q & 1))) & 0xFFFFu; }
  TWR_DEV uint32_t f(int q) const { return (pf[q >> 1] >> (16 * (q & 1))) & 0xFFFFu; }
};
TWR_DEV double sel4(int i, const double v[4]) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }
// sum over the four quads of a 16-lane row (lanes l, l-4, l-8, l-12 mod 16): DPP row_ror
TWR_DEV double row_perm_ror4(double v) { return quad_perm<0x124>(v); }
TWR_DEV double row_perm_ror8(double v) { return quad_perm<0x128>(v); }
TWR_DEV double row4_sum(double v) {
  v += row_perm_ror4(v);
  v += row_perm_ror8(v);
  return v;
}
// --- asynchronous loads.  The software pipeline of dyn_phase_kernel is scheduled by hand: its vector loads are issued
// as asm into AGPRs (free at one wave per SIMD) and waited for with explicit counted s_waitcnt, always inside the
// iteration that issued them.  Left to the compiler the prefetched values are shuffled between register files right
// after the loads are issued (which waits for them on the spot), and every wait behind the copy-out drains its stores.
// The compiler sees none of these loads; a value is used only through the wait that follows its load (`+a` operands),
// and the compiler's own waits (scalar loads, LDS) are unaffected.
typedef float twr_v4f __attribute__((ext_vector_type(4)));
TWR_DEV void aload(double& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(d) : "v"(p) : "memory"); }
TWR_DEV void aload(twr_v4f& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(d) : "v"(p) : "memory"); }
TWR_DEV void aload(uint32_t& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=a"(d) : "v"(p) : "memory"); }
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt / lgkmcnt fields left at "no wait")
constexpr int vmcnt_imm(int n) { return (n & 0xF) | (0x7 << 4) | (0xF << 8) | ((n >> 4) << 14); }
struct ARec {          // record of a pass: DynShared {tb, iTb}, q6, DynLoc                 (6 loads)
  twr_v4f sh;
  uint32_t q6;
  twr_v4f lc[4];
};
struct AIn {           // x values of a pass                                               (12 loads)
  double v[12];
};
struct APut {          // put offsets of a pass: PhasePutM row j, PhasePutF row j, PhaseEe  (6 loads)
  twr_v4f pm;
  double pf[3];
  twr_v4f pe[2];
};
TWR_DEV void pdyn_issue_rec(uint64_t shared, uint64_t loc, int loc_stride, int cnt, int n_ee, int lane, ARec& a) {
  const int n = min(lane >> 4, cnt - 1), e = min((lane >> 2) & 3, n_ee - 1);   // (ee >= n_ee read ee n_ee-1 and are neutralised)
  const char* sh = reinterpret_cast<const char*>(shared) + sizeof(DynShared) * (size_t)n;
  const char* lc = reinterpret_cast<const char*>(loc) + (size_t)e * (size_t)loc_stride + sizeof(DynLoc) * (size_t)n;
  aload(a.sh, sh);
  aload(a.q6, sh + offsetof(DynShared, q6));
#pragma unroll
  for (int q = 0; q < 4; ++q) aload(a.lc[q], lc + 16 * q);
}
template <int N>
TWR_DEV void pdyn_wait_rec(ARec& a, PDynRec& r) {
  asm volatile("s_waitcnt %6" : "+a"(a.sh), "+a"(a.q6), "+a"(a.lc[0]), "+a"(a.lc[1]), "+a"(a.lc[2]), "+a"(a.lc[3]) : "n"(vmcnt_imm(N)) : "memory");
  static_assert(offsetof(DynLoc, xbase_m) == 32 && offsetof(DynLoc, slots_m) == 40 && offsetof(DynLoc, slots_f) == 48 &&
                offsetof(DynLoc, im) == 56 && offsetof(DynLoc, cur) == 60, "DynLoc dwords");
  r.tb = pdyn_f64(a.sh.x, a.sh.y);
  r.iTb = pdyn_f64(a.sh.z, a.sh.w);
  r.q6 = (int32_t)a.q6;
  r.tm = pdyn_f64(a.lc[0].x, a.lc[0].y);
  r.Tm = pdyn_f64(a.lc[0].z, a.lc[0].w);
  r.tf = pdyn_f64(a.lc[1].x, a.lc[1].y);
  r.Tf = pdyn_f64(a.lc[1].z, a.lc[1].w);
  r.xbase_m = (int32_t)__float_as_uint(a.lc[2].x);
  r.xbase_f = (int32_t)__float_as_uint(a.lc[2].y);
  r.slots_m[0] = __float_as_uint(a.lc[2].z);
  r.slots_m[1] = __float_as_uint(a.lc[2].w);
  r.slots_f[0] = __float_as_uint(a.lc[3].x);
  r.slots_f[1] = __float_as_uint(a.lc[3].y);
  r.imjf = __float_as_uint(a.lc[3].z);
  r.misc = __float_as_uint(a.lc[3].w);
}
TWR_DEV void pdyn_issue_in(const PDynWork& w, const PDynRec& r, const double* __restrict__ x, int lane, AIn& a) {
  const double* xp = x + w.x_off;
  const int j = lane & 3;
  const double* xl = xp + w.off_lin + r.q6 + 3 * j;
  const double* xa = xp + w.off_ang + r.q6 + 3 * j;
  const uint32_t sm = (uint32_t)((((uint64_t)r.slots_m[1] << 32) | r.slots_m[0]) >> (12 * j)) & 0xFFFu;
  const uint32_t sf = (uint32_t)((((uint64_t)r.slots_f[1] << 32) | r.slots_f[0]) >> (12 * j)) & 0xFFFu;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    aload(a.v[d], xl + d);
    aload(a.v[3 + d], xa + d);
    const uint32_t am = (sm >> (4 * d)) & 0xFu, af = (sf >> (4 * d)) & 0xFu;   // absent candidates read slot 0, weight 0
    aload(a.v[6 + d], xp + r.xbase_m + (am != 0xFu ? am : 0u));
    aload(a.v[9 + d], xp + r.xbase_f + (af != 0xFu ? af : 0u));
  }
}
template <int N>
TWR_DEV void pdyn_wait_in(AIn& a, PDynIn& in) {
  asm volatile("s_waitcnt %12" : "+a"(a.v[0]), "+a"(a.v[1]), "+a"(a.v[2]), "+a"(a.v[3]), "+a"(a.v[4]), "+a"(a.v[5]), "+a"(a.v[6]),
               "+a"(a.v[7]), "+a"(a.v[8]), "+a"(a.v[9]), "+a"(a.v[10]), "+a"(a.v[11]) : "n"(vmcnt_imm(N)) : "memory");
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    in.bl[d] = a.v[d];
    in.ba[d] = a.v[3 + d];
    in.m[d] = a.v[6 + d];
    in.f[d] = a.v[9 + d];
  }
}
TWR_DEV void pdyn_issue_put(const PDynWork& w, const PDynRec& r, int lane, APut& a) {
  const int ee = (lane >> 2) & 3, j = lane & 3;
  // an end-effector the robot does not have takes the structure's dummy records (every offset = its trash entry, no
  // duration columns): the select is on the index, nothing is done to the loaded values
  const bool has_ee = ee < w.n_ee;
  const uint32_t im = has_ee ? (r.imjf & 0xFFFFu) : (uint32_t)(w.n_mput + ee), jf = has_ee ? (r.imjf >> 16) : (uint32_t)(w.n_fput + ee);
  const char* pm = reinterpret_cast<const char*>(w.mput) + sizeof(PhasePutM) * (size_t)im + 16 * j;
  const char* pf = reinterpret_cast<const char*>(w.fput) + sizeof(PhasePutF) * (size_t)jf + 24 * j;
  const char* pe = reinterpret_cast<const char*>(w.ee) + sizeof(PhaseEe) * (size_t)ee;
  aload(a.pm, pm);
#pragma unroll
  for (int q = 0; q < 3; ++q) aload(a.pf[q], pf + 8 * q);
  aload(a.pe[0], pe);
  aload(a.pe[1], pe + 16);
}
template <int N>
TWR_DEV void pdyn_wait_put(APut& a, PDynPut& pu) {
  asm volatile("s_waitcnt %6" : "+a"(a.pm), "+a"(a.pf[0]), "+a"(a.pf[1]), "+a"(a.pf[2]), "+a"(a.pe[0]), "+a"(a.pe[1]) : "n"(vmcnt_imm(N)) : "memory");
  pu.pm[0] = __float_as_uint(a.pm.x); pu.pm[1] = __float_as_uint(a.pm.y); pu.pm[2] = __float_as_uint(a.pm.z); pu.pm[3] = __float_as_uint(a.pm.w);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    pu.pf[2 * q] = (uint32_t)__double2loint(a.pf[q]);
    pu.pf[2 * q + 1] = (uint32_t)__double2hiint(a.pf[q]);
  }
  pu.ns = (int32_t)__float_as_uint(a.pe[0].x);
  pu.dur_ang[0] = (int32_t)__float_as_uint(a.pe[0].y); pu.dur_ang[1] = (int32_t)__float_as_uint(a.pe[0].z); pu.dur_ang[2] = (int32_t)__float_as_uint(a.pe[0].w);
  pu.dur_lin[0] = (int32_t)__float_as_uint(a.pe[1].x); pu.dur_lin[1] = (int32_t)__float_as_uint(a.pe[1].y); pu.dur_lin[2] = (int32_t)__float_as_uint(a.pe[1].z);
}
// One pass, first half: the math of lane (n, e, j); its share of the constraint values goes straight to global memory.
struct PDynVals {      // what the put half needs
  double wPj, wVj, wAj, wmj, wfj, f[3], rv[3];
  double F[3], mass;                     // e = 3: base-lin block
  double A[3], B[3], C[3];               // e < 3: base-ang factors of Euler dimension e
  double ap[3], ac[3], fprev[3], fcur[3];   // duration columns: angular rows (previous phases / current phase), forces
};
TWR_DEV void pdyn_math(const PDynWork& w, const PDynRec& r, const PDynIn& in, double* __restrict__ g, int lane, bool want_g,
                       PDynVals& V) {
  const int n = lane >> 4, ee = (lane >> 2) & 3, j = lane & 3;
  const bool live = n < w.cnt, has_ee = ee < w.n_ee;
  double (&f)[3] = V.f, (&rv)[3] = V.rv, (&F)[3] = V.F, (&fprev)[3] = V.fprev, (&fcur)[3] = V.fcur;
  double &wPj = V.wPj, &wVj = V.wVj, &wAj = V.wAj, &wmj = V.wmj, &wfj = V.wfj;
  {
    double wP[4], wV[4], wA[4];
    hermite_all(r.tb, r.iTb, wP, wV, wA);
    wPj = sel4(j, wP); wVj = sel4(j, wV); wAj = sel4(j, wA);
  }
  // base spline points (Spline::GetPoint in basis form): this lane's node value times its weight, summed over j
  double c[3], e[3], cdd[3], ed[3], edd[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c[d] = quad_sum(wPj * in.bl[d]);
    cdd[d] = quad_sum(wAj * in.bl[d]);
    e[d] = quad_sum(wPj * in.ba[d]);
    ed[d] = quad_sum(wVj * in.ba[d]);
    edd[d] = quad_sum(wAj * in.ba[d]);
  }
  // --- end-effector e: positions, forces, and the duration columns (PhaseSpline::GetJacobianOfPosWrtDurations,
  // phase_spline.cc:67-93, phase_durations.cc:126-154, polynomial.cc:236-257: dx/dT_poly is linear in the node values,
  // so it is a sum over j like the point itself)
  const uint32_t sm = (uint32_t)((((uint64_t)r.slots_m[1] << 32) | r.slots_m[0]) >> (12 * j)) & 0xFFFu;
  const uint32_t sf = (uint32_t)((((uint64_t)r.slots_f[1] << 32) | r.slots_f[0]) >> (12 * j)) & 0xFFFu;
  const bool shared = (r.misc >> 9) & 1, in_last = (r.misc >> 8) & 1;
  double xprev[3], xcur[3];
  auto ee_spline = [&](double t, double T, uint32_t slots, const double v[3], bool fold, double inner, double prevp, double& wj,
                       double pt[3], double prev[3], double cur[3]) {
    const double iT = 1.0 / T, iT2 = iT * iT, t2 = t * t, t3 = t2 * t;
    double wp[4], wv[4], wa[4], ct[4];
    hermite_all(t, iT, wp, wv, wa);
    // d pos / d T_poly = ct . (x0, v0, x1, v1)   (polynomial.cc:236-257, collected by node value)
    ct[0] = 6.0 * t2 * iT2 * iT - 6.0 * t3 * iT2 * iT2;
    ct[1] = 2.0 * t2 * iT2 - 2.0 * t3 * iT2 * iT;
    ct[2] = -ct[0];
    ct[3] = t2 * iT2 - 2.0 * t3 * iT2 * iT;
    if (fold) {   // stance ee-motion polynomial: p1 is the same variable as p0
      wp[0] += wp[2];
      wv[0] += wv[2];
      ct[0] += ct[2];
    }
    wj = sel4(j, wp);
    const double wvj = sel4(j, wv), ctj = sel4(j, ct);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double nv = ((slots >> (4 * d)) & 0xFu) != 0xFu ? v[d] : 0.0;
      pt[d] = quad_sum(wj * nv);
      const double vel = quad_sum(wvj * nv), dxdT = quad_sum(ctj * nv);
      const double dph = inner * (dxdT - prevp * vel);
      cur[d] = dph;
      prev[d] = -vel - (in_last ? dph : 0.0);
    }
  };
  {
    double p[3];
    ee_spline(r.tm, r.Tm, sm, in.m, shared, 1.0 / (double)((r.misc >> 16) & 0xF), (double)((r.misc >> 20) & 0xF), wmj, p, xprev, xcur);
    ee_spline(r.tf, r.Tf, sf, in.f, false, 1.0 / (double)((r.misc >> 24) & 0xF), (double)(r.misc >> 28), wfj, f, fprev, fcur);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      rv[d] = c[d] - p[d];
      if (!has_ee) f[d] = 0.0;
    }
  }
  // force and torque sums over the end-effectors (single_rigid_body_dynamics.cc:81-88): over the four quads of the row
  double tau[3];
  {
    double t3[3];
    cross3(f, rv, t3);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      F[d] = row4_sum(f[d]);
      tau[d] = row4_sum(t3[d]);
    }
  }
  // --- rotation: lanes j = 0..2 of a quad evaluate one sincos each and broadcast it
  double my_s, my_c;
  sincos_fast(sel3(j, e[0], e[1], e[2]), &my_s, &my_c);
  const double sx = quad_perm<0x00>(my_s), cx = quad_perm<0x00>(my_c);
  const double sy = quad_perm<0x55>(my_s), cy = quad_perm<0x55>(my_c);
  const double sz = quad_perm<0xAA>(my_s), cz = quad_perm<0xAA>(my_c);
  // --- angular quantities (euler_converter.cc:58-83,133-166,207-221)
  double R[3][3];
  R[0][0] = cy * cz; R[0][1] = cz * sx * sy - cx * sz; R[0][2] = sx * sz + cx * cz * sy;
  R[1][0] = cy * sz; R[1][1] = cx * cz + sx * sy * sz; R[1][2] = cx * sy * sz - cz * sx;
  R[2][0] = -sy;     R[2][1] = cy * sx;                R[2][2] = cx * cy;
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
  const TWR_CONST DevStruct* H = cptr<DevStruct>(w.hdr);  // uniform per work item: scalar loads
  double Iw6[6];  // I_w = R I_b R^T (single_rigid_body_dynamics.cc:91), symmetric: (00,01,02,11,12,22)
  {
    double Ib[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Ib[i] = H->Ib[i];
    double Tm[3][3];  // R I_b
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Tm[i][0] = R[i][0] * Ib[0] + R[i][1] * Ib[1] + R[i][2] * Ib[2];
      Tm[i][1] = R[i][0] * Ib[1] + R[i][1] * Ib[3] + R[i][2] * Ib[4];
      Tm[i][2] = R[i][0] * Ib[2] + R[i][1] * Ib[4] + R[i][2] * Ib[5];
    }
    Iw6[0] = Tm[0][0] * R[0][0] + Tm[0][1] * R[0][1] + Tm[0][2] * R[0][2];
    Iw6[1] = Tm[0][0] * R[1][0] + Tm[0][1] * R[1][1] + Tm[0][2] * R[1][2];
    Iw6[2] = Tm[0][0] * R[2][0] + Tm[0][1] * R[2][1] + Tm[0][2] * R[2][2];
    Iw6[3] = Tm[1][0] * R[1][0] + Tm[1][1] * R[1][1] + Tm[1][2] * R[1][2];
    Iw6[4] = Tm[1][0] * R[2][0] + Tm[1][1] * R[2][1] + Tm[1][2] * R[2][2];
    Iw6[5] = Tm[2][0] * R[2][0] + Tm[2][1] * R[2][1] + Tm[2][2] * R[2][2];
  }
  double Iw_wd[3], Iw_w[3];
  symmul(Iw6, omd, Iw_wd);
  symmul(Iw6, om, Iw_w);
  const double m = H->mass;
  if (want_g && live && ee < 2 && j < 3) {  // GetDynamicViolation, single_rigid_body_dynamics.cc:76-101: one value per lane
    double wxIw[3];
    cross3(om, Iw_w, wxIw);
    const double ga = sel3(j, Iw_wd[0] + wxIw[0] - tau[0], Iw_wd[1] + wxIw[1] - tau[1], Iw_wd[2] + wxIw[2] - tau[2]);
    const double gl = sel3(j, m * cdd[0] - F[0], m * cdd[1] - F[1], m * cdd[2] - F[2] + m * H->gravity);
    g[w.g_off + 6 * n + 3 * ee + j] = ee == 0 ? ga : gl;
  }
  V.mass = m;
  {  // duration columns {[r]x J_f + [f]x J_p ; -J_f} of ee e (dynamic_constraint.cc:107-113)
    double t1[3], t2[3];
    cross3(rv, fprev, t1);
    cross3(f, xprev, t2);
#pragma unroll
    for (int i = 0; i < 3; ++i) V.ap[i] = t1[i] + t2[i];
    cross3(rv, fcur, t1);
    cross3(f, xcur, t2);
#pragma unroll
    for (int i = 0; i < 3; ++i) V.ac[i] = t1[i] + t2[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) V.A[i] = V.B[i] = V.C[i] = 0.0;
  if (ee < 3) {
    // --- base-ang block (:123-165), Euler dimension d = e, node value j, factored.
    // The columns of M are the rotation axes of the ZYX sequence, so dR/d e_d = [M_d]x R
    // (the cell-wise derivatives of euler_converter.cc:241-268 in closed form) and
    //   d(I_w v)/d e_d = R_d I_b R^T v + R I_b R_d^T v = M_d x (I_w v) + I_w (v x M_d)
    // (jac11+jac12 resp. jac21+jac22 of the reference) without ever forming R_d.
    const int role = ee;
    const double dMx_dy[3] = {-sy * cz, -sy * sz, -cy};
    const double dMx_dz[3] = {-cy * sz, cy * cz, 0.0};
    const double dMy_dz[3] = {-cz, -sz, 0.0};
    const double dMdx_dy[3] = {-cz * cy * yd + sy * sz * zd, -sy * cz * zd - cy * sz * yd, sy * yd};
    const double dMdx_dz[3] = {sz * sy * yd - cy * cz * zd, -cy * sz * zd - sy * cz * yd, 0.0};
    const double dMdy_dz[3] = {sz * zd, -cz * zd, 0.0};
    double Md[3], dwd_ed[3], dw[3], dwd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Md[i] = sel3(role, Mx[i], My[i], i == 2 ? 1.0 : 0.0);  // column d of M (euler_converter.cc:133-148)
      // d omega_dot / d edot_d
      dwd_ed[i] = sel3(role, Mdx[i], Mdy[i] + xd * dMx_dy[i], xd * dMx_dz[i] + yd * dMy_dz[i]);
      // d omega / d e_d , d omega_dot / d e_d   (roll: none)   (euler_converter.cc:168-198,270-304)
      dw[i] = sel3(role, 0.0, xd * dMx_dy[i], xd * dMx_dz[i] + yd * dMy_dz[i]);
      dwd[i] = sel3(role, 0.0, xd * dMdx_dy[i] + edd[0] * dMx_dy[i],
                    xd * dMdx_dz[i] + yd * dMdy_dz[i] + edd[0] * dMx_dz[i] + edd[1] * dMy_dz[i]);
    }
    auto dIw = [&](const double v[3], const double Iwv[3], double o[3]) {
      double t1[3], t2[3], t3b[3];
      cross3(Md, Iwv, t1);
      cross3(v, Md, t2);
      symmul(Iw6, t2, t3b);
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = t1[i] + t3b[i];
    };
    double (&A)[3] = V.A, (&B)[3] = V.B, (&C)[3] = V.C;
    symmul(Iw6, Md, C);
    {  // B_d = I_w d(omega_dot)/d(edot_d) + M_d x (I_w omega) + omega x (I_w M_d)
      double t1[3], t2[3], t3b[3];
      symmul(Iw6, dwd_ed, t1);
      cross3(Md, Iw_w, t2);
      cross3(om, C, t3b);
#pragma unroll
      for (int i = 0; i < 3; ++i) B[i] = t1[i] + t2[i] + t3b[i];
    }
    {  // A_d = dI_w(omega_dot) + I_w d(omega_dot) + d(omega) x I_w omega + omega x (dI_w(omega) + I_w d(omega))
      double t1[3], t2[3], t3b[3], t4[3], t5[3], t6[3], t7[3];
      dIw(omd, Iw_wd, t1);
      symmul(Iw6, dwd, t2);
      cross3(dw, Iw_w, t3b);
      dIw(om, Iw_w, t4);
      symmul(Iw6, dw, t5);
#pragma unroll
      for (int i = 0; i < 3; ++i) t6[i] = t4[i] + t5[i];
      cross3(om, t6, t7);
#pragma unroll
      for (int i = 0; i < 3; ++i) A[i] = t1[i] + t2[i] + t3b[i] + t7[i];
    }
  }
}
// One pass, second half: the Jacobian values of lane (n, e, j) into the LDS image (`img`: byte address of the image's
// first value).
TWR_DEV void pdyn_puts(const PDynWork& w, const PDynRec& r, const PDynPut& pu, const PDynVals& V, char* __restrict__ img, int lane) {
  const int n = lane >> 4, ee = (lane >> 2) & 3, j = lane & 3;
  if (n >= w.cnt) return;
  const double (&f)[3] = V.f, (&rv)[3] = V.rv, (&F)[3] = V.F;
  const double wPj = V.wPj, wVj = V.wVj, wAj = V.wAj, wmj = V.wmj, wfj = V.wfj, m = V.mass;
  char* nb = img + (size_t)n * (size_t)w.node_vals * 8;   // first value of this time node
  // --- ee-motion block [f]x J_p (:181-192) and ee-force block {[r]x J_f ; -J_f} (:167-179), candidates (j, D), BEFORE
  // the base blocks: a candidate that is not a variable carries the offset of a base-ang entry of row AX, which the
  // base-ang stores below overwrite (same wave, program order)
#define TWR_EE_TILE3(D, R1, R2)                           \
  {                                                       \
    lds_put(nb, pu.m(2 * D + 0), crs<R1, D>(f) * wmj);    \
    lds_put(nb, pu.m(2 * D + 1), crs<R2, D>(f) * wmj);    \
    lds_put(nb, pu.f(3 * D + 0), crs<R1, D>(rv) * wfj);   \
    lds_put(nb, pu.f(3 * D + 1), crs<R2, D>(rv) * wfj);   \
    lds_put(nb, pu.f(3 * D + 2), -wfj);                   \
  }
  TWR_EE_TILE3(0, 1, 2)
  TWR_EE_TILE3(1, 2, 0)
  TWR_EE_TILE3(2, 0, 1)
#undef TWR_EE_TILE3
  {  // duration columns of ee e, phases p = j, j + 4, ...: `prev` for the phases before the current one, `cur` for it,
     // zero (the cleared image) after it
    const int cur = (int)(r.misc & 0xFF), n_dur = min(cur + 1, pu.ns);
    for (int p = j; p < n_dur; p += 4) {
      const bool is_cur = p == cur;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        lds_put(nb, (uint32_t)(pu.dur_ang[q] + 8 * p), is_cur ? V.ac[q] : V.ap[q]);
        lds_put(nb, (uint32_t)(pu.dur_lin[q] + 8 * p), is_cur ? -V.fcur[q] : -V.fprev[q]);
      }
    }
  }
  char* row[3] = {nb, nb + w.row_off[0], nb + w.row_off[1]};
  if (ee == 3) {   // base-lin block, node value j: ang rows -sum_i [f_i]x J_pos, lin rows m J_acc (:103-121)
    const uint32_t rl0 = w.row_off[2], rl1 = w.row_off[3], rl2 = w.row_off[4];
    lds_put(row[0], 8 * (2 * j + 0), -crs<0, 1>(F) * wPj);
    lds_put(row[0], 8 * (2 * j + 1), -crs<0, 2>(F) * wPj);
    lds_put(row[1], 8 * (2 * j + 0), -crs<1, 0>(F) * wPj);
    lds_put(row[1], 8 * (2 * j + 1), -crs<1, 2>(F) * wPj);
    lds_put(row[2], 8 * (2 * j + 0), -crs<2, 0>(F) * wPj);
    lds_put(row[2], 8 * (2 * j + 1), -crs<2, 1>(F) * wPj);
    lds_put(nb + rl0, 8 * j, m * wAj);
    lds_put(nb + rl1, 8 * j, m * wAj);
    lds_put(nb + rl2, 8 * j, m * wAj);
  } else {         // base-ang block, Euler dimension e, node value j
#pragma unroll
    for (int q = 0; q < 3; ++q) lds_put(row[q], 8 * (8 + 3 * j + ee), V.A[q] * wPj + V.B[q] * wVj + V.C[q] * wAj);
  }
}
// zero [0, n) doubles of an LDS image (n rounded up to a pair): eight 16-byte stores per lane and round, lanes past the
// end re-clear the last pair
TWR_DEV void lds_clear(double* __restrict__ img, int n, int lane) {
  const double2 z = {0.0, 0.0};
  const int npairs = (n + 1) >> 1, nit = (npairs + 63) >> 6;
  for (int it0 = 0; it0 < nit; it0 += 8) {
#pragma unroll
    for (int b = 0; b < 8; ++b) reinterpret_cast<double2*>(img)[min(lane + 64 * (it0 + b), npairs - 1)] = z;
  }
}
// image -> HBM for a run-time length: batches of eight 16-byte LDS reads, then their eight stores; iterations past the
// end re-store the last complete pair (idempotent).  The image is NOT shifted by the parity of the destination (it fills
// the workgroup's LDS to the last byte): a 16-byte aligned global pair is read from an 8-byte aligned LDS address with
// ds_read2_b64.  The batch is staged in AGPRs (DS and vector-memory instructions take them as data operands; the kernel
// runs few waves per SIMD, so they are free), which leaves the arch VGPRs to the math -- left to the register
// allocator the eight reads end up serialised through one VGPR quad.  Written as asm for that reason; the waits are
// explicit (the compiler's own s_waitcnt counts stay conservative: unknown younger operations only make them stricter).
// NIT > 0: compile-time number of store instructions (the wait for the loads issued before them can then be counted);
// NIT = 0: run-time length (images larger than 40 KB).  At most two more stores follow (odd head / tail).
template <int NIT>
TWR_DEV void stream_out(double* __restrict__ dst, const double* __restrict__ stage, int n, int par, int lane) {
  char* al = reinterpret_cast<char*>(dst - par);  // 16-byte aligned; global pair t = image values 2t - par, 2t + 1 - par
  const int total = n + par;
  const uint32_t last = (uint32_t)((total >> 1) - 1) * 16u;
  const uint32_t first = (uint32_t)(par + lane) * 16u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)stage - 8u * (uint32_t)par;
  constexpr int kB = 8;
  auto batch = [&](int it0) {
    uint32_t off[kB];
    twr_v4f v[kB];
#pragma unroll
    for (int b = 0; b < kB; ++b) {
      off[b] = min(first + 1024u * (uint32_t)(it0 + b), last);
      asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=a"(v[b]) : "v"(lds0 + off[b]) : "memory");
    }
#pragma unroll
    for (int b = 0; b < kB; ++b) {
      switch (kB - 1 - b) {   // the b-th read has landed once at most kB-1-b younger ones are outstanding
        case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+a"(v[b])); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+a"(v[b])); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+a"(v[b])); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+a"(v[b])); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+a"(v[b])); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+a"(v[b])); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+a"(v[b])); break;
        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(v[b])); break;
      }
      asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off[b]), "a"(v[b]), "s"(al) : "memory");
    }
  };
  if constexpr (NIT > 0) {
    static_assert(NIT % kB == 0, "whole batches");
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += kB) batch(it0);
  } else {
    const int nit = ((total >> 1) - par + 63) >> 6;
    for (int it0 = 0; it0 < nit; it0 += kB) batch(it0);
  }
  if (par && lane == 0 && n > 0) dst[0] = stage[0];
  if ((total & 1) && lane == 0 && total - 1 > par) dst[n - 1] = stage[n - 1];
}

// LDS: the image of one pass (dynamic size).  State at the top of iteration i: the record and the x values of pass i have
// landed.  Vector-memory operations of one iteration, in issue order (loads by hand, see aload):
//   P  put offsets of pass i (6 loads)        R  record of pass i+1 (6 loads)
//   .. clear the image, math of pass i ..     G  constraint values (1 store, WANT_G)
//   wait P (younger: R, G)  .. puts ..        wait R (younger: G)
//   X  x values of pass i+1 (12 loads)        S  copy-out of pass i (NIT stores + at most 2)
//   wait X (younger: S)
template <int NIT, bool WANT_G, bool WANT_J>   // NIT: store instructions of the copy-out (40: images of up to 40 KB, C3 sizes; 0: run-time)
__global__ __launch_bounds__(64, 1) void dyn_phase_kernel(const PDynWork* __restrict__ work, int n_work,
                                                          const double* __restrict__ x, double* __restrict__ g,
                                                          double* __restrict__ jac) {
  extern __shared__ __attribute__((aligned(16))) double pdyn_lds[];
  const int lane = threadIdx.x;
  const int stride = gridDim.x;
  int i = blockIdx.x;
  if (i >= n_work) return;
  PDynWork w0 = work[i], w1 = w0;
  const int n_ee = w0.n_ee;   // (all problems of a batch share n_ee)
  PDynRec r0;
  PDynIn in;
  {
    ARec ar;
    AIn ai;
    pdyn_issue_rec(w0.shared, w0.loc, w0.loc_stride, w0.cnt, n_ee, lane, ar);
    pdyn_wait_rec<0>(ar, r0);
    pdyn_issue_in(w0, r0, x, lane, ai);
    pdyn_wait_in<0>(ai, in);
  }
  for (; i < n_work; i += stride) {
    const bool has1 = i + stride < n_work;   // (the last pass of a workgroup prefetches itself once more: harmless)
    if (has1) w1 = work[i + stride];
    APut ap;
    ARec ar;
    AIn ai;
    if (WANT_J) pdyn_issue_put(w0, r0, lane, ap);                                   // P
    pdyn_issue_rec(w1.shared, w1.loc, w1.loc_stride, w1.cnt, n_ee, lane, ar);                      // R
    const int nv = w0.cnt * w0.node_vals;
    if (WANT_J) lds_clear(pdyn_lds, nv, lane);
    PDynVals V;
    pdyn_math(w0, r0, in, g, lane, WANT_G, V);                                      // (G inside)
    if (WANT_J) {
      PDynPut pu;
      pdyn_wait_put<6 + (WANT_G ? 1 : 0)>(ap, pu);
      pdyn_puts(w0, r0, pu, V, reinterpret_cast<char*>(pdyn_lds), lane);
    }
    PDynRec r1;
    pdyn_wait_rec<(WANT_G ? 1 : 0)>(ar, r1);
    pdyn_issue_in(w1, r1, x, lane, ai);                                             // X
    if (WANT_J) {                                                                   // S
      double* dst = jac + w0.j_off;
      stream_out<NIT>(dst, pdyn_lds, nv, (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1), lane);
    }
    pdyn_wait_in<(WANT_J ? (NIT < 63 ? NIT : 63) : 0)>(ai, in);   // (NIT = 0: drains the copy-out)
    w0 = w1;
    r0 = r1;
  }
}

// Optimised timings, pre-pass: one workgroup per (problem, ee) resolves the x-dependent part of every time node of
// the dynamic and the rangeofmotion-<ee> grids -- phase and polynomial durations, active polynomials, local times,
// current phase (PhaseDurations::SetVariables, ConvertPhaseToPolyDurations, Spline::GetSegmentID / GetLocalTime) -- into
// DynLoc / RomRec records in the batch's scratch buffer.  256 threads, one per time node: a lookup is a chain of ~60
// dependent LDS reads (the reference's sequential accumulation, kept bit for bit), so the kernel lives on waves in flight
// (one wave per workgroup, four time nodes per lane: 0.094 ms per 2048 C3 problems; this form: see DESIGN 6.0).
constexpr int kLocateThreads = 256;
__global__ __launch_bounds__(kLocateThreads) void phase_locate_kernel(const LocWork* __restrict__ work, const double* __restrict__ x) {
  __shared__ double s_ph[TWR_MAX_PHASES_DEV], s_md[kMaxPhasePolys], s_fd[kMaxPhasePolys];
  // the records of one block of time nodes are staged in LDS and leave as one contiguous, fully coalesced stream
  // (a lane storing its own 64-byte record touches 32 lines per store instruction)
  __shared__ __attribute__((aligned(16))) char s_rec[kLocateThreads * 64];
  static_assert(sizeof(DynLoc) == 64 && sizeof(RomRec) == 64, "record staging");
  const LocWork lw = work[blockIdx.x];
  const char* blob = reinterpret_cast<const char*>(lw.blob);
  const DevStruct* H = reinterpret_cast<const DevStruct*>(blob);
  const PhaseTables* PT = tbl<PhaseTables>(blob, H->o_phase);
  const int lane = threadIdx.x, e = lw.ee;
  phase_poly_durations_wave(PT, blob, x + lw.x_off, e, s_ph, s_md, lane, kLocateThreads);
  const PhasePoly* mp = tbl<PhasePoly>(blob, PT->o_mpoly[e]);
  const int last_phase = PT->n_phases[e] - 1;
  auto flush = [&](char* dst, int n_recs) {   // s_rec[0 .. 64 n_recs) -> dst, 16 bytes per thread and round
    __syncthreads();
    for (int c = lane; c < 4 * n_recs; c += kLocateThreads)
      reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(s_rec)[c];
    __syncthreads();
  };
  if (lw.dyn_loc) {
    const PhasePoly* fp = tbl<PhasePoly>(blob, PT->o_fpoly[e]);
    for (int q = lane; q < PT->n_fpoly[e]; q += kLocateThreads) s_fd[q] = s_ph[fp[q].phase] / fp[q].n_in_phase;
    __syncthreads();
    const double* tg = tbl<double>(blob, PT->o_tdyn);
    const int K = PT->k_dyn;
    for (int k0 = 0; k0 < K; k0 += kLocateThreads) {
      const int k = k0 + lane;
      if (k < K) {
        const double t = tg[k];
        double tlm, tlf, tlp;
        const int qm = locate_segment(s_md, PT->n_mpoly[e], t, tlm);
        const int qf = locate_segment(s_fd, PT->n_fpoly[e], t, tlf);
        const int cur = locate_segment(s_ph, PT->n_phases[e], t, tlp);
        const PhasePoly pm = mp[qm], pf = fp[qf];
        DynLoc o;
        o.tm = tlm; o.Tm = s_md[qm];
        o.tf = tlf; o.Tf = s_fd[qf];
        o.xbase_m = pm.xbase; o.xbase_f = pf.xbase;
        const uint64_t sm = slots_of(pm.cand), sf = slots_of(pf.cand);
        o.slots_m[0] = (uint32_t)sm; o.slots_m[1] = (uint32_t)(sm >> 32);
        o.slots_f[0] = (uint32_t)sf; o.slots_f[1] = (uint32_t)(sf >> 32);
        o.im = (uint16_t)(PT->mput_base[e] + qm); o.jf = (uint16_t)(PT->fput_base[e] + qf);
        o.cur = (uint8_t)cur;
        o.flags = (uint8_t)((cur == last_phase ? 1 : 0) | (meta_shared(pm.meta) ? 2 : 0));
        o.np_m = (uint8_t)(pm.n_in_phase | (pm.poly_in_phase << 4));
        o.np_f = (uint8_t)(pf.n_in_phase | (pf.poly_in_phase << 4));
        reinterpret_cast<DynLoc*>(s_rec)[lane] = o;
      }
      flush(reinterpret_cast<char*>(lw.dyn_loc) + sizeof(DynLoc) * (size_t)k0, min(kLocateThreads, K - k0));
    }
  }
  if (!lw.recs) return;
  const double* tg = tbl<double>(blob, PT->o_trom);
  const RomRec* base = tbl<RomRec>(blob, PT->o_rom_recs[e]);  // base-spline part (tb, iTb, q6) is x-independent
  const int K = PT->k_rom;
  for (int k0 = 0; k0 < K; k0 += kLocateThreads) {
    const int k = k0 + lane;
    if (k < K) {
      const double t = tg[k];
      double tlm, tlp;
      const int qm = locate_segment(s_md, PT->n_mpoly[e], t, tlm);
      const int cur = locate_segment(s_ph, PT->n_phases[e], t, tlp);
      const PhasePoly pm = mp[qm];
      RomRec r = base[k];
      r.tm = tlm;
      r.iTm = 1.0 / s_md[qm];
      r.xbase = pm.xbase;
      r.meta = pm.meta;
      const uint64_t slots = slots_of(pm.cand);
      r.slots[0] = (uint32_t)slots;
      r.slots[1] = (uint32_t)(slots >> 32);
      r.voff = 0;
      r.pad[0] = (uint32_t)pm.base_all | ((uint32_t)cur << 16) | ((cur == last_phase ? 1u : 0u) << 24);
      r.pad[1] = (uint32_t)pm.n_in_phase | ((uint32_t)pm.poly_in_phase << 8);
      reinterpret_cast<RomRec*>(s_rec)[lane] = r;
    }
    flush(reinterpret_cast<char*>(lw.recs) + sizeof(RomRec) * (size_t)k0, min(kLocateThreads, K - k0));
  }
}

// rangeofmotion-<ee> with optimised timings (rom_phase_kernel): the same recipe as dyn_phase_kernel.  A pass is a run of
// up to SIXTEEN consecutive time nodes of one (problem, ee) -- their expanded rows [base-lin 12 | base-ang 12 (8) | all
// ee-motion_e variables | all durations] are one contiguous slice of the value array (C3: 182 values per node, 23 KB
// per pass) -- and a time node gets FOUR lanes:
//   lane = 4 n + j:   n = time node of the pass, j = node value (p0, v0, p1, v1)
// Lane (n, j) loads the three dimensions of node value j of the base splines and of the active ee-motion polynomial
// (9 values instead of 36) and owns their Jacobian columns; spline points are DPP quad sums, the rotation algebra is
// evaluated by all four lanes.  The wave clears the LDS image of the pass, every lane stores its ~26 values (+ its share
// of the duration columns) at their final positions, and the image goes to HBM with 16-byte coalesced stores.
// The x-dependent lookup comes from the pre-pass (RomRec); the pipeline is scheduled by hand like dyn_phase_kernel's.
struct ARomIn {        // x values of a pass (9 loads)
  double v[9];
};
struct RomPRec {       // the RomRec written by the pre-pass, whole dwords
  double tb, iTb, tm, iTm;
  int32_t q6, xbase;
  uint32_t slots[2], pad0, pad1;   // pad0 = base_all | current phase << 16 | in_last_phase << 24, pad1 = n_in_phase | poly_in_phase << 8
  uint32_t meta;
};
struct ARomRec {
  twr_v4f q[4];
};
TWR_DEV void romp_issue_rec(uint64_t recs, int cnt, int lane, ARomRec& a) {
  const char* p = reinterpret_cast<const char*>(recs) + sizeof(RomRec) * (size_t)min(lane >> 2, cnt - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) aload(a.q[q], p + 16 * q);
}
template <int N>
TWR_DEV void romp_wait_rec(ARomRec& a, RomPRec& r) {
  asm volatile("s_waitcnt %4" : "+a"(a.q[0]), "+a"(a.q[1]), "+a"(a.q[2]), "+a"(a.q[3]) : "n"(vmcnt_imm(N)) : "memory");
  static_assert(offsetof(RomRec, q6) == 32 && offsetof(RomRec, meta) == 44 && offsetof(RomRec, slots) == 48 && offsetof(RomRec, pad) == 56,
                "RomRec dwords");
  r.tb = pdyn_f64(a.q[0].x, a.q[0].y);
  r.iTb = pdyn_f64(a.q[0].z, a.q[0].w);
  r.tm = pdyn_f64(a.q[1].x, a.q[1].y);
  r.iTm = pdyn_f64(a.q[1].z, a.q[1].w);
  r.q6 = (int32_t)__float_as_uint(a.q[2].x);
  r.xbase = (int32_t)__float_as_uint(a.q[2].y);
  r.meta = __float_as_uint(a.q[2].w);
  r.slots[0] = __float_as_uint(a.q[3].x);
  r.slots[1] = __float_as_uint(a.q[3].y);
  r.pad0 = __float_as_uint(a.q[3].z);
  r.pad1 = __float_as_uint(a.q[3].w);
}
TWR_DEV void romp_issue_in(const RomPhaseWork& w, const RomPRec& r, const double* __restrict__ x, int lane, ARomIn& a) {
  const double* xp = x + w.x_off;
  const int j = lane & 3;
  const double* xl = xp + w.off_lin + r.q6 + 3 * j;
  const double* xa = xp + w.off_ang + r.q6 + 3 * j;
  const uint32_t sm = (uint32_t)((((uint64_t)r.slots[1] << 32) | r.slots[0]) >> (12 * j)) & 0xFFFu;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    aload(a.v[d], xl + d);
    aload(a.v[3 + d], xa + d);
    const uint32_t am = (sm >> (4 * d)) & 0xFu;   // an absent candidate reads slot 0, weight 0
    aload(a.v[6 + d], xp + r.xbase + (am != 0xFu ? am : 0u));
  }
}
template <int N>
TWR_DEV void romp_wait_in(ARomIn& a, double in[9]) {
  asm volatile("s_waitcnt %9" : "+a"(a.v[0]), "+a"(a.v[1]), "+a"(a.v[2]), "+a"(a.v[3]), "+a"(a.v[4]), "+a"(a.v[5]), "+a"(a.v[6]),
               "+a"(a.v[7]), "+a"(a.v[8]) : "n"(vmcnt_imm(N)) : "memory");
#pragma unroll
  for (int q = 0; q < 9; ++q) in[q] = a.v[q];
}
// RangeOfMotionConstraint::{UpdateConstraintAtInstance, UpdateJacobianAtInstance} (range_of_motion_constraint.cc:58-109)
// with the PhaseSpline Jacobians (phase_spline.cc:44-93) for lane (n, j); `img`: byte address of the image.
TWR_DEV void romp_pass(const RomPhaseWork& w, const RomPRec& r, const double in[9], char* __restrict__ img, double* __restrict__ g,
                       int lane, bool want_g, bool want_j) {
  const int n = lane >> 2, j = lane & 3;
  const bool live = n < w.cnt;
  double wPj;
  {
    double wP[4];
    hermite_pos(r.tb, r.iTb, wP);
    wPj = sel4(j, wP);
  }
  double c[3], e[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c[d] = quad_sum(wPj * in[d]);
    e[d] = quad_sum(wPj * in[3 + d]);
  }
  // ee-motion point and the duration columns of PhaseSpline::GetJacobianOfPosWrtDurations (phase_spline.cc:67-93,
  // phase_durations.cc:126-154, polynomial.cc:236-257; dx/dT_poly collected by node value: a sum over j)
  const uint32_t sm = (uint32_t)((((uint64_t)r.slots[1] << 32) | r.slots[0]) >> (12 * j)) & 0xFFFu;
  const bool shared = meta_shared(r.meta), in_last = (r.pad0 >> 24) & 1;
  double wmj, v[3], prev[3], cur[3];
  {
    const double t = r.tm, iT = r.iTm, iT2 = iT * iT, t2 = t * t, t3 = t2 * t;
    const double inner = 1.0 / (double)(r.pad1 & 0xFF), prevp = (double)((r.pad1 >> 8) & 0xFF);
    double wp[4], wv[4], wa[4], ct[4];
    hermite_all(t, iT, wp, wv, wa);
    ct[0] = 6.0 * t2 * iT2 * iT - 6.0 * t3 * iT2 * iT2;
    ct[1] = 2.0 * t2 * iT2 - 2.0 * t3 * iT2 * iT;
    ct[2] = -ct[0];
    ct[3] = t2 * iT2 - 2.0 * t3 * iT2 * iT;
    if (shared) {   // stance ee-motion polynomial: p1 is the same variable as p0
      wp[0] += wp[2];
      wv[0] += wv[2];
      ct[0] += ct[2];
    }
    wmj = sel4(j, wp);
    const double wvj = sel4(j, wv), ctj = sel4(j, ct);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double nv = ((sm >> (4 * d)) & 0xFu) != 0xFu ? in[6 + d] : 0.0;
      v[d] = quad_sum(wmj * nv) - c[d];
      const double vel = quad_sum(wvj * nv), dxdT = quad_sum(ctj * nv);
      const double dph = inner * (dxdT - prevp * vel);
      cur[d] = dph;
      prev[d] = -vel - (in_last ? dph : 0.0);
    }
  }
  // rotation: lanes j = 0..2 of the quad evaluate one sincos each and broadcast it (euler_converter.cc:207-221)
  Rot ro;
  {
    double my_s, my_c;
    sincos_fast(sel3(j, e[0], e[1], e[2]), &my_s, &my_c);
    const double sx = quad_perm<0x00>(my_s), cx = quad_perm<0x00>(my_c);
    const double sy = quad_perm<0x55>(my_s), cy = quad_perm<0x55>(my_c);
    const double sz = quad_perm<0xAA>(my_s), cz = quad_perm<0xAA>(my_c);
    ro.sx = sx; ro.cx = cx; ro.sy = sy; ro.cy = cy; ro.sz = sz; ro.cz = cz;
    ro.R[0][0] = cy * cz; ro.R[0][1] = cz * sx * sy - cx * sz; ro.R[0][2] = sx * sz + cx * cz * sy;
    ro.R[1][0] = cy * sz; ro.R[1][1] = cx * cz + sx * sy * sz; ro.R[1][2] = cx * sy * sz - cz * sx;
    ro.R[2][0] = -sy;     ro.R[2][1] = cy * sx;                ro.R[2][2] = cx * cy;
  }
  if (want_g && live && j < 3) {   // b_R_w (p - c): one value per lane
    double gv[3];
    matTvec(ro.R, v, gv);
    g[w.g_off + 3 * n + j] = sel3(j, gv[0], gv[1], gv[2]);
  }
  if (!want_j || !live) return;
  // DerivOfRotVecMult(t, v, inverse=true) = d(R^T v)/d e_d = R^T (v x M_d)  (euler_converter.cc:223-239, see rom_item)
  double ux[3], uy[3], uz[3];
  {
    const double Mx[3] = {ro.cy * ro.cz, ro.cy * ro.sz, -ro.sy}, My[3] = {-ro.sz, ro.cz, 0.0};
    double tx[3], ty[3];
    cross3(v, Mx, tx);
    cross3(v, My, ty);
    const double tz[3] = {v[1], -v[0], 0.0};  // v x e_z
    matTvec(ro.R, tx, ux);
    matTvec(ro.R, ty, uy);
    matTvec(ro.R, tz, uz);
  }
  const int base_all = r.pad0 & 0xFFFF, cur_ph = (r.pad0 >> 16) & 0xFF;
  const uint32_t len0 = 20u + (uint32_t)(w.msize + w.ns), len1 = len0 + 4u;
  char* nb = img + (size_t)n * (size_t)w.node_vals * 8;
  char* row[3] = {nb, nb + 8u * len0, nb + 8u * (len0 + len1)};
  const uint32_t mo[3] = {20u, 24u, 24u};
  // R^T J_p of candidates (j, D), BEFORE the base blocks: a candidate that is not a variable goes to the first value of
  // row 0, which the base-lin stores below overwrite (same wave, program order)
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const uint32_t sl = (sm >> (4 * d)) & 0xFu;
    const bool valid = sl != 0xFu;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      lds_put(valid ? row[q] + 8u * (mo[q] + (uint32_t)base_all + sl) : nb, 0, ro.R[d][q] * wmj);
  }
  {  // duration columns b_R_w * GetJacobianOfPosWrtDurations (range_of_motion_constraint.cc:106-108), phases p = j, j + 4, ..:
     // `prev` for the phases before the current one, `cur` for it, zero (the cleared image) after it
    double rp[3], rc[3];
    matTvec(ro.R, prev, rp);
    matTvec(ro.R, cur, rc);
    const int n_dur = min(cur_ph + 1, w.ns);
    for (int p = j; p < n_dur; p += 4) {
      const bool is_cur = p == cur_ph;
#pragma unroll
      for (int q = 0; q < 3; ++q) lds_put(row[q], 8u * (mo[q] + (uint32_t)w.msize + (uint32_t)p), is_cur ? rc[q] : rp[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int d = 0; d < 3; ++d) lds_put(row[q], 8 * (3 * j + d), -ro.R[d][q] * wPj);  // -R^T J_c
    if (q == 0) {  // row 0 of R^T v does not depend on roll
      lds_put(row[0], 8 * (12 + 2 * j + 0), wPj * uy[0]);
      lds_put(row[0], 8 * (12 + 2 * j + 1), wPj * uz[0]);
    } else {
      lds_put(row[q], 8 * (12 + 3 * j + 0), wPj * ux[q]);
      lds_put(row[q], 8 * (12 + 3 * j + 1), wPj * uy[q]);
      lds_put(row[q], 8 * (12 + 3 * j + 2), wPj * uz[q]);
    }
  }
}
// Vector-memory operations of one iteration, in issue order:  R record of pass i+1 (4 loads)  .. clear, math ..
//   G constraint values (1 store, WANT_G)  .. puts ..  wait R (younger: G)  X x values of pass i+1 (9 loads)
//   S copy-out of pass i (NIT stores + at most 2)   wait X (younger: S)
template <int NIT, bool WANT_G, bool WANT_J>
__global__ __launch_bounds__(64, 1) void rom_phase_kernel(const RomPhaseWork* __restrict__ work, int n_work,
                                                          const double* __restrict__ x, double* __restrict__ g,
                                                          double* __restrict__ jac) {
  extern __shared__ __attribute__((aligned(16))) double romp_lds[];
  const int lane = threadIdx.x;
  const int stride = gridDim.x;
  int i = blockIdx.x;
  if (i >= n_work) return;
  RomPhaseWork w0 = work[i], w1 = w0;
  RomPRec r0;
  double in[9];
  {
    ARomRec ar;
    ARomIn ai;
    romp_issue_rec(w0.recs, w0.cnt, lane, ar);
    romp_wait_rec<0>(ar, r0);
    romp_issue_in(w0, r0, x, lane, ai);
    romp_wait_in<0>(ai, in);
  }
  for (; i < n_work; i += stride) {
    const bool has1 = i + stride < n_work;   // (the last pass of a workgroup prefetches itself once more: harmless)
    if (has1) w1 = work[i + stride];
    ARomRec ar;
    ARomIn ai;
    romp_issue_rec(w1.recs, w1.cnt, lane, ar);                                       // R
    const int nv = w0.cnt * w0.node_vals;
    if (WANT_J) lds_clear(romp_lds, nv, lane);
    romp_pass(w0, r0, in, reinterpret_cast<char*>(romp_lds), g, lane, WANT_G, WANT_J);   // (G inside)
    RomPRec r1;
    romp_wait_rec<(WANT_G ? 1 : 0)>(ar, r1);
    romp_issue_in(w1, r1, x, lane, ai);                                              // X
    if (WANT_J) {                                                                    // S
      double* dst = jac + w0.j_off;
      __builtin_amdgcn_s_setprio(3);   // six workgroups per CU: as in dyn_kernel, the streaming wave goes first (-2 %)
      stream_out<NIT>(dst, romp_lds, nv, (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1), lane);
      __builtin_amdgcn_s_setprio(0);
    }
    romp_wait_in<(WANT_J ? (NIT < 63 ? NIT : 63) : 0)>(ai, in);   // (NIT = 0: drains the copy-out)
    w0 = w1;
    r0 = r1;
  }
}

// ---------------------------------------------------------------- trajectory sampling
// fpowr::GetTrajectory (fpowr/include/fpowr/footstep_plan_extractor.h:19-53) for a batch of solutions: lane =
// sample.  Durations (constants of the structure, or recomputed from x with optimised timings) go to LDS once
// per workgroup; every lane accumulates its sample time like the reference (t += dt), locates the active
// polynomial of every spline and evaluates position / velocity / acceleration in Hermite basis form.
TWR_DEV void sample_spline(const double* __restrict__ xp, const PolyDesc& pd, double tl, double T, double p[3], double v[3],
                           double a[3]) {
  double X[12], nv[4][3], wp[4], wv[4], wa[4];
  gather12c(xp, pd.xbase, pd.cand, X);
  node_values(slots_of(pd.cand), meta_shared(pd.meta), X, nv);
  hermite_all(tl, 1.0 / T, wp, wv, wa);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    p[d] = wp[0] * nv[0][d] + wp[1] * nv[1][d] + wp[2] * nv[2][d] + wp[3] * nv[3][d];
    v[d] = wv[0] * nv[0][d] + wv[1] * nv[1][d] + wv[2] * nv[2][d] + wv[3] * nv[3][d];
    a[d] = wa[0] * nv[0][d] + wa[1] * nv[1][d] + wa[2] * nv[2][d] + wa[3] * nv[3][d];
  }
}
// With `times` set the kernel is fpowr::ExtractInitialGuess (fpowr/include/fpowr/initial_guess_extractor.h:17-34)
// instead: sample s of every problem is taken at times[s] and the record is [ t | state: base-lin p, base-ang p (Euler
// angles), base-lin v, base-ang v (Euler rates) | controls (36): ee-motion acceleration of ee i at 3 i, twelve zeros
// ("joint torques"), ee-force of ee i at 24 + 3 i ] = 49 doubles.
__global__ __launch_bounds__(64) void sample_kernel(const SampleWork* __restrict__ work, const double* __restrict__ x,
                                                    double* __restrict__ out, double dt, const double* __restrict__ times) {
  __shared__ double s_bd[2 * kMaxPhasePolys], s_ph[kMaxEE][TWR_MAX_PHASES_DEV], s_md[kMaxEE][kMaxPhasePolys],
      s_fd[kMaxEE][kMaxPhasePolys];
  const SampleWork sw = work[blockIdx.x];
  const char* blob = reinterpret_cast<const char*>(sw.blob);
  const DevStruct* H = reinterpret_cast<const DevStruct*>(blob);
  const SampleTables* ST = tbl<SampleTables>(blob, H->o_sample);
  const double* xp = x + sw.x_off;
  const int lane = threadIdx.x, n_ee = H->n_ee;
  for (int q = lane; q < ST->n_base; q += 64) s_bd[q] = tbl<double>(blob, ST->o_bdur)[q];
  for (int e = 0; e < n_ee; ++e) {
    if (H->timings) {  // PhaseDurations::SetVariables + ConvertPhaseToPolyDurations (phase_durations.cc:77-103)
      const PhaseTables* PT = tbl<PhaseTables>(blob, H->o_phase);
      phase_poly_durations_wave(PT, blob, xp, e, s_ph[e], s_md[e], lane);
      const PhasePoly* fp = tbl<PhasePoly>(blob, PT->o_fpoly[e]);
      for (int q = lane; q < PT->n_fpoly[e]; q += 64) s_fd[e][q] = s_ph[e][fp[q].phase] / fp[q].n_in_phase;
    } else {
      for (int q = lane; q < ST->n_phases[e]; q += 64) s_ph[e][q] = tbl<double>(blob, ST->o_phdur[e])[q];
      for (int q = lane; q < ST->n_mpoly[e]; q += 64) s_md[e][q] = tbl<double>(blob, ST->o_mdur[e])[q];
      for (int q = lane; q < ST->n_fpoly[e]; q += 64) s_fd[e][q] = tbl<double>(blob, ST->o_fdur[e])[q];
    }
  }
  __syncthreads();
  if (lane >= sw.cnt) return;
  const int s_idx = sw.s0 + lane;
  double t = 0.0;
  if (times) t = times[s_idx];
  else for (int i = 0; i < s_idx; ++i) t += dt;   // the reference's accumulated sample time
  const int rec = times ? 49 : 20 + 13 * n_ee;
  double* o = out + sw.out_off + (int64_t)s_idx * rec;
  o[0] = t;
  // base-lin / base-ang (NodesVariablesAll: [p0 v0 p1 v1] x 3 of polynomial q at 6 q)
  double tl;
  const int qb = locate_segment(s_bd, ST->n_base, t, tl);
  double wp[4], wv[4], wa[4];
  hermite_all(tl, 1.0 / s_bd[qb], wp, wv, wa);
  const double* xl = xp + ST->off_lin + 6 * qb;
  const double* xa = xp + ST->off_ang + 6 * qb;
  double e3[3], ed[3], edd[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    o[1 + d] = wp[0] * xl[d] + wp[1] * xl[3 + d] + wp[2] * xl[6 + d] + wp[3] * xl[9 + d];
    o[4 + d] = wv[0] * xl[d] + wv[1] * xl[3 + d] + wv[2] * xl[6 + d] + wv[3] * xl[9 + d];
    o[7 + d] = wa[0] * xl[d] + wa[1] * xl[3 + d] + wa[2] * xl[6 + d] + wa[3] * xl[9 + d];
    e3[d] = wp[0] * xa[d] + wp[1] * xa[3 + d] + wp[2] * xa[6 + d] + wp[3] * xa[9 + d];
    ed[d] = wv[0] * xa[d] + wv[1] * xa[3 + d] + wv[2] * xa[6 + d] + wv[3] * xa[9 + d];
    edd[d] = wa[0] * xa[d] + wa[1] * xa[3 + d] + wa[2] * xa[6 + d] + wa[3] * xa[9 + d];
  }
  if (times) {   // ExtractInitialGuess record
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double lv = o[4 + d];
      o[4 + d] = e3[d];
      o[7 + d] = lv;
      o[10 + d] = ed[d];
    }
    for (int q = 13; q < 49; ++q) o[q] = 0.0;
    for (int e = 0; e < n_ee; ++e) {
      double tlm, tlf, p[3], v[3], a[3];
      const int qm = locate_segment(s_md[e], ST->n_mpoly[e], t, tlm);
      const int qf = locate_segment(s_fd[e], ST->n_fpoly[e], t, tlf);
      sample_spline(xp, tbl<PolyDesc>(blob, ST->o_mdesc[e])[qm], tlm, s_md[e][qm], p, v, a);
#pragma unroll
      for (int d = 0; d < 3; ++d) o[13 + 3 * e + d] = a[d];
      sample_spline(xp, tbl<PolyDesc>(blob, ST->o_fdesc[e])[qf], tlf, s_fd[e][qf], p, v, a);
#pragma unroll
      for (int d = 0; d < 3; ++d) o[37 + 3 * e + d] = p[d];
    }
    return;
  }
  Rot ro;
  rotation(e3, ro);
  {  // Eigen::Quaterniond(R) (euler_converter.cc:51-56; Eigen 3.3 Quaternion.h, quaternionbase_assign_impl)
    const double (&m)[3][3] = ro.R;
    double q[4];  // x y z w
    double tr = m[0][0] + m[1][1] + m[2][2];
    if (tr > 0) {
      tr = sqrt(tr + 1.0);
      q[3] = 0.5 * tr;
      tr = 0.5 / tr;
      q[0] = (m[2][1] - m[1][2]) * tr;
      q[1] = (m[0][2] - m[2][0]) * tr;
      q[2] = (m[1][0] - m[0][1]) * tr;
    } else {
      int i = 0;
      if (m[1][1] > m[0][0]) i = 1;
      if (m[2][2] > (i == 0 ? m[0][0] : m[1][1])) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      auto M = [&](int r, int c) { return sel3(r, sel3(c, m[0][0], m[0][1], m[0][2]), sel3(c, m[1][0], m[1][1], m[1][2]),
                                               sel3(c, m[2][0], m[2][1], m[2][2])); };
      tr = sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
      const double qi = 0.5 * tr;
      tr = 0.5 / tr;
      const double qw = (M(k, j) - M(j, k)) * tr, qj = (M(j, i) + M(i, j)) * tr, qk = (M(k, i) + M(i, k)) * tr;
      q[3] = qw;
      q[0] = i == 0 ? qi : (j == 0 ? qj : qk);
      q[1] = i == 1 ? qi : (j == 1 ? qj : qk);
      q[2] = i == 2 ? qi : (j == 2 ? qj : qk);
    }
    o[10] = q[3]; o[11] = q[0]; o[12] = q[1]; o[13] = q[2];
  }
  {  // omega = M edot, omega_dot = Mdot edot + M eddot (euler_converter.cc:58-83,133-166)
    const double sy = ro.sy, cy = ro.cy, sz = ro.sz, cz = ro.cz;
    const double xd = ed[0], yd = ed[1], zd = ed[2];
    const double Mx[3] = {cy * cz, cy * sz, -sy}, My[3] = {-sz, cz, 0.0};
    const double Mdx[3] = {-cz * sy * yd - cy * sz * zd, cy * cz * zd - sy * sz * yd, -cy * yd};
    const double Mdy[3] = {-cz * zd, -sz * zd, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      o[14 + i] = Mx[i] * xd + My[i] * yd + (i == 2 ? zd : 0.0);
      o[17 + i] = Mdx[i] * xd + Mdy[i] * yd + Mx[i] * edd[0] + My[i] * edd[1] + (i == 2 ? edd[2] : 0.0);
    }
  }
  for (int e = 0; e < n_ee; ++e) {
    double* oe = o + 20 + 13 * e;
    double tlp, tlm, tlf;
    const int phase = locate_segment(s_ph[e], ST->n_phases[e], t, tlp);   // PhaseDurations::IsContactPhase
    oe[0] = ((phase & 1) == 0) == (ST->contact0[e] != 0) ? 1.0 : 0.0;
    const int qm = locate_segment(s_md[e], ST->n_mpoly[e], t, tlm);
    const int qf = locate_segment(s_fd[e], ST->n_fpoly[e], t, tlf);
    const PolyDesc pm = tbl<PolyDesc>(blob, ST->o_mdesc[e])[qm];
    const PolyDesc pf = tbl<PolyDesc>(blob, ST->o_fdesc[e])[qf];
    double p[3], v[3], a[3];
    sample_spline(xp, pm, tlm, s_md[e][qm], p, v, a);
#pragma unroll
    for (int d = 0; d < 3; ++d) { oe[1 + d] = p[d]; oe[4 + d] = v[d]; oe[7 + d] = a[d]; }
    sample_spline(xp, pf, tlf, s_fd[e][qf], p, v, a);
#pragma unroll
    for (int d = 0; d < 3; ++d) oe[10 + d] = p[d];
  }
}
hipError_t launch_sample(const SampleWork* work, int n_work, const double* x, double* out, double dt, const double* times,
                         hipStream_t stream) {
  if (n_work <= 0) return hipSuccess;
  return twr_launch(sample_kernel, dim3(n_work), dim3(64), 0, stream, work, x, out, dt, times);
}

// ---------------------------------------------------------------- candidate scoring
// twr_batch_score: per problem and constraint family the inf- and 1-norm of the bound violation
// max(lower - g, g - upper, 0) over the family's rows (bounds of ConstraintSet::GetBounds, twr_structure_bounds).
// A sweep then returns 16 doubles per candidate instead of its Jacobian.
// (round 5: 256 threads per problem, row r = 256 k + thread, sixteen rows per thread; family slot and bounds of a row come
// from its 16-bit meta word and the structure's short table of distinct (lower, upper) pairs, device_tables.h ScoreTables.)
// Two dependent round trips: (1) the work item and its successor (row count = difference of their g offsets; the list
// carries an end entry), (2) the rows of g, the meta words of the rows AND the head of the score record (2 KB requested on spec:
// the pair table) -> LDS -- the record sits at a fixed offset of the blob (kScoreOff), no field of the header is needed.  The slot of a thread's rows never falls, so it keeps one
// running (max, sum) pair and hands it to its LDS cell when the slot moves on.  Round 4's kernel fetched a cold candidate's
// fields one s_load at a time and 16 bytes of bounds per row: 25 us for 128 candidates, 43 us for 1024.
TWR_DEV double nan_max(double a, double b) { return (a != a || b != b) ? (a != a ? a : b) : fmax(a, b); }
__device__ __forceinline__ void best_of(double& v, int& i, double ov, int oi) {
  if (ov < v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}
__global__ __launch_bounds__(256) void score_kernel(const NodeWork* __restrict__ work, const double* __restrict__ g,
                                                    double* __restrict__ scores) {
  constexpr int kHeadDwords = kScoreHeadBytes / 4, kTabDwords = (int)(sizeof(ScoreTables) / 4);
  static_assert(kHeadDwords == 512 && kTabDwords + 4 * kScoreMaxPairs <= kHeadDwords, "the whole record within its head: two loads per thread");
  __shared__ __attribute__((aligned(16))) int32_t s_tab[kHeadDwords];   // the record as in the blob
  __shared__ double red[256 * 16 + 16 * 16];   // cell (2 slot + {max, sum}, thread), then the partial results of the fold
  const int tid = threadIdx.x;
  const NodeWork w = work[blockIdx.x];
  const int n_rows = (int)(work[blockIdx.x + 1].g_off - w.g_off);
  if (n_rows <= 0) {   // (uniform) a structure whose constraint sets have no rows: nothing is violated, nothing is read
    if (tid < 16) scores[16 * (size_t)blockIdx.x + tid] = 0.0;
    return;
  }
  const double* gp = g + w.g_off;
  const char* blob = reinterpret_cast<const char*>(w.blob);
#pragma unroll
  for (int c = 0; c < 16; ++c) red[c * 256 + tid] = 0.0;   // (a cell is written at most once more; own cells only: no barrier)
  int cur = 0;
  double cm = 0.0, cs = 0.0;
  for (int base = 0; base < n_rows; base += 16 * 256) {   // (one trip for structures of up to 4096 rows)
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = gp[min(base + 256 * k + tid, n_rows - 1)];
    const int32_t* Tw = reinterpret_cast<const int32_t*>(tbl<ScoreTables>(blob, kScoreOff));   // (fixed offset: no header field)
    const uint16_t* meta = reinterpret_cast<const uint16_t*>(blob + kScoreOff + kScoreHeadBytes);
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = meta[min(base + 256 * k + tid, n_rows - 1)];
    if (base == 0) {
      const int32_t a = Tw[tid], c = Tw[tid + 256];
      s_tab[tid] = a;
      s_tab[tid + 256] = c;
      __syncthreads();
    }
    const double* s_pairs = reinterpret_cast<const double*>(s_tab + kTabDwords);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (base + 256 * k < n_rows) {   // uniform
        const bool live = base + 256 * k + tid < n_rows;
        const int slot = (int)(m[k] >> 12), q = (int)(m[k] & 0xFFFu);
        const double lo = s_pairs[2 * q], up = s_pairs[2 * q + 1], val = v[k];
        // NaN-propagating: a non-finite constraint value makes the family's scores NaN instead of hiding in a max()
        double viol = fmax(fmax(lo - val, val - up), 0.0);
        if (val != val) viol = val;
        if (live && slot != cur) {   // the slot moved on: hand over the running pair (rare: a few times per thread)
          red[(2 * cur) * 256 + tid] = cm;
          red[(2 * cur + 1) * 256 + tid] = cs;
          cm = cs = 0.0;
          cur = slot;
        }
        if (live) {
          cm = nan_max(cm, viol);
          cs += viol;
        }
      }
    }
  }
  red[(2 * cur) * 256 + tid] = cm;
  red[(2 * cur + 1) * 256 + tid] = cs;
  // 256 threads x 16 cells -> 16 values, transposed: thread 16 c + j folds column j of threads c, c + 16, ..., then sixteen
  // threads fold the sixteen partial results of their column and store it at its FAMILY's place
  __syncthreads();
  {
    const int j = tid & 15, c = tid >> 4;   // column j (even: an inf-norm, odd: a 1-norm)
    double acc = red[j * 256 + c];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      const double o = red[j * 256 + c + 16 * i];
      acc = (j & 1) ? acc + o : nan_max(acc, o);
    }
    red[256 * 16 + c * 16 + j] = acc;
  }
  __syncthreads();
  if (tid < 16) {
    const int slot = reinterpret_cast<const int8_t*>(s_tab + 2)[tid >> 1];   // ScoreTables::slot_of_family[family tid / 2]
    double acc = 0.0;
    if (slot >= 0) {
      const int j = 2 * slot + (tid & 1);
      acc = red[256 * 16 + j];
      for (int c = 1; c < 16; ++c) {
        const double o = red[256 * 16 + c * 16 + j];
        acc = (tid & 1) ? acc + o : nan_max(acc, o);
      }
    }
    scores[16 * (size_t)blockIdx.x + tid] = acc;
  }
}
hipError_t launch_score(const NodeWork* work, int n_problems, const double* g, double* scores, hipStream_t stream) {
  return twr_launch(score_kernel, dim3(n_problems), dim3(256), 0, stream, work, g, scores);
}

// twr_batch_best: the planner's decision on the device -- arg-min over a score table of the summed inf-norm violations of
// the chosen constraint families (the table may be longer than this batch: after an all-gather it holds the candidates of
// every rank).  Same rule as towr_amd.dist.best_candidate: families summed in ascending order, NaN loses, first index wins
// a tie.  One launch: every block reduces its stride of candidates, the last block to finish reduces the blocks' results
// (threadfence + counter, reset for the next call) and writes best[0] = index, best[1] = its total.
__global__ __launch_bounds__(256) void best_kernel(const double* __restrict__ scores, int n, unsigned families, double* __restrict__ partial,
                                                   unsigned* __restrict__ counter, double* __restrict__ best, double index_offset) {
  __shared__ double s_v[4];
  __shared__ int s_i[4];
  __shared__ bool s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double bv = __builtin_inf();
  int bi = 0x7fffffff;
  for (int c = blockIdx.x * 256 + tid; c < n; c += gridDim.x * 256) {
    const double* row = scores + 16 * (size_t)c;
    double t = 0.0;
    bool first = true;
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (families >> f & 1u) {
        t = first ? row[2 * f] : t + row[2 * f];
        first = false;
      }
    if (t != t) t = __builtin_inf();
    best_of(bv, bi, t, c);
  }
  auto block_reduce = [&]() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best_of(bv, bi, __shfl_xor(bv, o), __shfl_xor(bi, o));
    if (lane == 0) {
      s_v[wave] = bv;
      s_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0)
      for (int w = 1; w < 4; ++w) best_of(bv, bi, s_v[w], s_i[w]);
  };
  block_reduce();
  if (tid == 0) {
    partial[2 * blockIdx.x] = bv;
    partial[2 * blockIdx.x + 1] = (double)bi;
    __threadfence();
    s_last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  bv = __builtin_inf();
  bi = 0x7fffffff;
  for (int k = tid; k < (int)gridDim.x; k += 256)
    best_of(bv, bi, __builtin_nontemporal_load(partial + 2 * k), (int)__builtin_nontemporal_load(partial + 2 * k + 1));
  __syncthreads();
  block_reduce();
  if (tid == 0) {
    best[0] = index_offset + (double)(bi == 0x7fffffff ? 0 : bi);
    best[1] = bv;
    *counter = 0u;
  }
}
int best_max_blocks() { return 256; }
hipError_t launch_best(const double* scores, int n, unsigned families, double* partial, unsigned* counter, double* best, double index_offset,
                       hipStream_t stream) {
  int blocks = (n + 1023) / 1024;   // >= four candidates per thread before another block pays
  blocks = blocks < 1 ? 1 : (blocks > best_max_blocks() ? best_max_blocks() : blocks);
  return twr_launch(best_kernel, dim3(blocks), dim3(256), 0, stream, scores, n, families, partial, counter, best, index_offset);
}

// ---------------------------------------------------------------- contact plan
// fpowr::ExtractFootstepPlan (fpowr/include/fpowr/footstep_plan_extractor.h:69-133) minus the nearest-plane lookup
// (boost::geometry over ROS messages, left to the caller): the solution is sampled every dt like GetTrajectory
// (t accumulated), a footstep state is the first sample and every sample whose contact flags differ from the
// previous sample's (HasEndEffectorContactChanged, :55-67); its duration is the time to the next footstep state, the
// last one lasts until time_horizon (:121-128).  One wave per problem, lane = sample, 64 samples per round, footstep
// states compacted with a ballot.  Record: [ t | duration | contact per ee | ee-motion position (3) per ee ].
__global__ __launch_bounds__(64) void contact_plan_kernel(const NodeWork* __restrict__ work, const double* __restrict__ x,
                                                          double* __restrict__ out, int32_t* __restrict__ counts, double dt,
                                                          double time_horizon, int n_samples_max, int max_steps) {
  __shared__ double s_ph[kMaxEE][TWR_MAX_PHASES_DEV], s_md[kMaxEE][kMaxPhasePolys];
  const NodeWork w = work[blockIdx.x];
  const char* blob = reinterpret_cast<const char*>(w.blob);
  const DevStruct* H = reinterpret_cast<const DevStruct*>(blob);
  const SampleTables* ST = tbl<SampleTables>(blob, H->o_sample);
  const double* xp = x + w.x_off;
  const int lane = threadIdx.x, n_ee = H->n_ee;
  for (int e = 0; e < n_ee; ++e) {
    if (H->timings) {  // PhaseDurations::SetVariables + ConvertPhaseToPolyDurations (phase_durations.cc:77-103)
      phase_poly_durations_wave(tbl<PhaseTables>(blob, H->o_phase), blob, xp, e, s_ph[e], s_md[e], lane);
    } else {
      for (int q = lane; q < ST->n_phases[e]; q += 64) s_ph[e][q] = tbl<double>(blob, ST->o_phdur[e])[q];
      for (int q = lane; q < ST->n_mpoly[e]; q += 64) s_md[e][q] = tbl<double>(blob, ST->o_mdur[e])[q];
    }
  }
  __syncthreads();
  // GetTrajectory: while (t <= T + 1e-5) { ...; t += dt; }
  const double T = ST->t_total;
  const int rec = 2 + 4 * n_ee;
  double* o = out + (size_t)blockIdx.x * (size_t)max_steps * rec;
  int n_steps = 0;          // wave-uniform
  double t_carry = 0.0;     // t of the last footstep state of the previous round (wave-uniform)
  for (int s0 = 0; s0 < n_samples_max; s0 += 64) {
    const int s_idx = s0 + lane;
    double t = 0.0, tq = 0.0;   // this sample's and the previous sample's accumulated time (t += dt, like the reference)
    for (int i = 0; i < s_idx; ++i) {
      tq = t;
      t += dt;
    }
    const bool live = t <= T + 1e-5;
    unsigned mask = 0, mask_prev = 0;
    for (int e = 0; e < n_ee; ++e) {  // PhaseDurations::IsContactPhase (phase_durations.cc:118-124)
      double tl;
      const int ph = locate_segment(s_ph[e], ST->n_phases[e], t, tl);
      mask |= ((((ph & 1) == 0) == (ST->contact0[e] != 0)) ? 1u : 0u) << e;
      const int pq = locate_segment(s_ph[e], ST->n_phases[e], tq, tl);
      mask_prev |= ((((pq & 1) == 0) == (ST->contact0[e] != 0)) ? 1u : 0u) << e;
    }
    const bool step = live && (s_idx == 0 || mask != mask_prev);
    const uint64_t ball = __ballot(step);
    if (ball == 0) {
      if (!__any(live)) break;
      continue;
    }
    const uint64_t below = ball & ((1ull << lane) - 1ull);
    const uint64_t above = lane == 63 ? 0ull : (ball >> (lane + 1));
    const int my = n_steps + __popcll(below);
    // time of the footstep state before this one: the closest step lane below, or the previous round's last
    const double t_below = __shfl(t, below ? 63 - __clzll(below) : lane);
    const double t_prev = below ? t_below : t_carry;
    if (step && my < max_steps) {
      double* r = o + (size_t)my * rec;
      r[0] = t;
      if (!above) r[1] = time_horizon - t;   // last footstep state so far (:125-127); a later round may close it
      for (int e = 0; e < n_ee; ++e) {
        r[2 + e] = (mask >> e) & 1u ? 1.0 : 0.0;
        double tlm;
        const int qm = locate_segment(s_md[e], ST->n_mpoly[e], t, tlm);
        const PolyDesc pm = tbl<PolyDesc>(blob, ST->o_mdesc[e])[qm];
        double p[3], v[3], a[3];
        sample_spline(xp, pm, tlm, s_md[e][qm], p, v, a);
#pragma unroll
        for (int d = 0; d < 3; ++d) r[2 + n_ee + 3 * e + d] = p[d];
      }
    }
    // duration of the state before (:121-123) -- also when THIS state no longer fits the output (my == max_steps): the
    // last record that was kept still gets its duration
    if (step && my > 0 && my - 1 < max_steps) o[(size_t)(my - 1) * rec + 1] = t - t_prev;
    t_carry = __shfl(t, 63 - __clzll(ball));
    n_steps += __popcll(ball);
  }
  if (lane == 0) counts[blockIdx.x] = n_steps;
}
hipError_t launch_contact_plan(const NodeWork* work, int n_problems, const double* x, double* out, int32_t* counts, double dt,
                               double time_horizon, int n_samples_max, int max_steps, hipStream_t stream) {
  return twr_launch(contact_plan_kernel, dim3(n_problems), dim3(64), 0, stream, work, x, out, counts, dt, time_horizon,
                    n_samples_max, max_steps);
}

// ---------------------------------------------------------------- nearest plane of a footstep
// fpowr::NearestPlaneLookup::GetNearestPlaneIndex (nearest_plane_lookup.h:62-84) for every (problem, footstep state,
// end-effector) of a contact plan: boost::geometry::distance(point, polygon) restated from Boost 1.71 (winding
// strategy for covered_by, projected-point distance to the ring's consecutive points, no closing edge added; exact
// coordinate comparisons where boost allows a few ulp).  One thread per foot, polygons read from global memory.
TWR_DEV int ring_side(const double* __restrict__ xy, int n, double px, double py) {   // 1 inside, 0 boundary, -1 outside
  if (n < 4) return -1;
  int count = 0;
  for (int i = 0; i + 1 < n; ++i) {
    const double s1x = xy[2 * i], s1y = xy[2 * i + 1], s2x = xy[2 * i + 2], s2y = xy[2 * i + 3];
    const bool eq1 = s1x == px, eq2 = s2x == px;
    int c = 0;
    if (eq1 && eq2) {
      if ((s1y <= py && s2y >= py) || (s2y <= py && s1y >= py)) return 0;
    } else {
      c = eq1 ? (s2x > px ? 1 : -1) : eq2 ? (s1x > px ? -1 : 1) : (s1x < px && s2x > px) ? 2 : (s2x < px && s1x > px) ? -2 : 0;
    }
    if (c != 0) {
      int side;
      if (c == 1 || c == -1) {
        const double sey = eq1 ? s1y : s2y;
        side = py == sey ? 0 : (py < sey ? -c : c);
      } else {
        // (no FMA contraction: the sign of a tiny determinant must not depend on it)
        const double det = __dsub_rn(__dmul_rn(s2x - s1x, py - s1y), __dmul_rn(s2y - s1y, px - s1x));
        side = det > 0 ? 1 : (det < 0 ? -1 : 0);
      }
      if (side == 0) return 0;
      if (side * c > 0) count += c;
    }
  }
  return count == 0 ? -1 : 1;
}
TWR_DEV double ring_distance(const double* __restrict__ xy, int n, double px, double py) {
  if (n == 0) return 0.0;
  auto comparable = [&](double ax, double ay, double bx, double by) {
    const double vx = bx - ax, vy = by - ay, wx = px - ax, wy = py - ay;
    const double c1 = __dadd_rn(__dmul_rn(wx, vx), __dmul_rn(wy, vy));
    if (c1 <= 0) return __dadd_rn(__dmul_rn(wx, wx), __dmul_rn(wy, wy));
    const double c2 = __dadd_rn(__dmul_rn(vx, vx), __dmul_rn(vy, vy));
    if (c2 <= c1) return __dadd_rn(__dmul_rn(px - bx, px - bx), __dmul_rn(py - by, py - by));
    const double b = c1 / c2, qx = __dadd_rn(ax, __dmul_rn(b, vx)), qy = __dadd_rn(ay, __dmul_rn(b, vy));
    return __dadd_rn(__dmul_rn(px - qx, px - qx), __dmul_rn(py - qy, py - qy));
  };
  if (n == 1) return sqrt(comparable(xy[0], xy[1], xy[0], xy[1]));
  double best = comparable(xy[0], xy[1], xy[2], xy[3]);
  for (int i = 0; i + 1 < n; ++i) {
    const double c = comparable(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3]);
    if (c == 0.0) return 0.0;
    if (c < best) best = c;
  }
  return sqrt(best);
}
__global__ __launch_bounds__(256) void plane_kernel(const double* __restrict__ plan, const int32_t* __restrict__ counts,
                                                    const double* __restrict__ poly_xy, const int32_t* __restrict__ poly_start,
                                                    int n_polys, int n_problems, int max_steps, int n_ee,
                                                    int32_t* __restrict__ plane_index) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)n_problems * max_steps * n_ee) return;
  const int e = (int)(id % n_ee), s = (int)((id / n_ee) % max_steps), p = (int)(id / ((int64_t)n_ee * max_steps));
  const double* rec = plan + ((int64_t)p * max_steps + s) * (2 + 4 * n_ee);
  int nearest = -1;
  if (s < counts[p] && rec[2 + e] != 0.0) {
    const double px = rec[2 + n_ee + 3 * e], py = rec[2 + n_ee + 3 * e + 1];
    double min_distance = 1.7976931348623157e308;
    for (int i = 0; i < n_polys; ++i) {
      const double* xy = poly_xy + 2 * poly_start[i];
      const int n = poly_start[i + 1] - poly_start[i];
      const double distance = ring_side(xy, n, px, py) >= 0 ? 0.0 : ring_distance(xy, n, px, py);
      if (distance < min_distance) {
        min_distance = distance;
        nearest = i;
      }
    }
  }
  plane_index[id] = nearest;
}
hipError_t launch_planes(const double* plan, const int32_t* counts, const double* poly_xy, const int32_t* poly_start, int n_polys,
                         int n_problems, int max_steps, int n_ee, int32_t* plane_index, hipStream_t stream) {
  const int64_t n = (int64_t)n_problems * max_steps * n_ee;
  if (n <= 0) return hipSuccess;
  return twr_launch(plane_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, plan, counts, poly_xy, poly_start, n_polys,
                    n_problems, max_steps, n_ee, plane_index);
}

// ---------------------------------------------------------------- TWR_EVAL_CHECK
// Per-problem non-finite flags (failure detection): a separate pass over the outputs of one
// evaluation, off the hot path (it re-reads g and jac once).  status[p] |= 1 if a constraint value of problem p is
// NaN/Inf, |= 2 if a Jacobian value is.  Sixteen workgroups per problem, each scanning a contiguous share.
constexpr int kCheckParts = 16;
TWR_DEV bool non_finite(double v) { return (__double2hiint(v) & 0x7ff00000) == 0x7ff00000; }
__global__ __launch_bounds__(256) void check_kernel(const int64_t* __restrict__ g_off, const int64_t* __restrict__ j_off,
                                                    const double* __restrict__ g, const double* __restrict__ jac,
                                                    int32_t* __restrict__ status, int flags) {
  const int p = blockIdx.x / kCheckParts, part = blockIdx.x % kCheckParts;
  int bad = 0;
  if (flags & 1) {
    const int64_t a = g_off[p], n = g_off[p + 1] - a;
    const int64_t lo = a + n * part / kCheckParts, hi = a + n * (part + 1) / kCheckParts;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) bad |= non_finite(g[i]) ? 1 : 0;
  }
  if (flags & 2) {
    const int64_t a = j_off[p], n = j_off[p + 1] - a;
    const int64_t lo = a + n * part / kCheckParts, hi = a + n * (part + 1) / kCheckParts;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) bad |= non_finite(jac[i]) ? 2 : 0;
  }
  if (bad) atomicOr(&status[p], bad);
}
hipError_t launch_check(int n_problems, const int64_t* g_off, const int64_t* j_off, const double* g, const double* jac,
                        int32_t* status, int flags, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(status, 0, sizeof(int32_t) * (size_t)n_problems, stream);
  if (e != hipSuccess) return e;
  return twr_launch(check_kernel, dim3(n_problems * kCheckParts), dim3(256), 0, stream, g_off, j_off, g, jac, status, flags);
}

// host-side launcher (called from capi.cc): three launches on one stream.  The dyn/rom grids are
// persistent: as many workgroups as are resident at once.  Residency is LDS bound; the occupancy API
// over-reports it for the dynamic kernel (measured: 7 x 22.5 KB resident, an 8th starts a second
// round), so the per-CU counts are fixed here and can be overridden for experiments.
// (Running dyn and rom concurrently on two streams was measured and is slower than back to back.)
// Tuning knobs (include/towr_amd.h, "Tuning knobs"): read from the environment ONLY in builds with -DTWR_TUNING_KNOBS
// (make TUNING=1; what scripts/ab.py and the A/B notes of DESIGN 6 use); the default build has the measured optima compiled in.
static int env_int(const char* name, int dflt) {
#ifdef TWR_TUNING_KNOBS
  const char* e = getenv(name);
  if (!e) return dflt;
  int v = atoi(e);
  return v > 0 ? v : dflt;
#else
  (void)name;
  return dflt;
#endif
}
static int n_chunks_of(const int n_fam[4]) { return n_fam[0] + n_fam[1] + n_fam[2] + n_fam[3]; }
hipError_t launch_eval(int n_ee, int n_cu, const DynWork* dyn, int n_dyn, int dyn_map_chunks /* 2 or 4 */, const RomWork* rom, int n_rom,
                       int rom_max_vals, const NodeWork* node, int n_node, int node_families /* 2: terrain + force only; 4 */,
                       const FamWork* const fam[4], const int n_fam[4] /* chunk lists of node_chunk_kernel; all 0: none */,
                       const PDynWork* pdyn, int n_pdyn, int pdyn_img_cap, const LocWork* ploc, int n_ploc, const RomPhaseWork* prom, int n_prom, int prom_img_cap, const double* x,
                       double* g, double* jac, double* dump /* kDynDump doubles */, int flags, bool stream_nt /* non-temporal copy-out of
                       dyn / rom (copy_out_fixed) */, const FlatWork* flat /* values-only work items in groups of four; nullptr: none */, int n_flat, int flat_max_x /* variables of the largest problem */, hipStream_t stream, hipEvent_t* ev /* 4 events or nullptr */) {
#ifdef TWR_TUNING_KNOBS   // (read on every launch: an experiment can change them between evaluations of one process)
  const int dyn_bpc = env_int("TWR_DYN_BPC", 8), rom_bpc = env_int("TWR_ROM_BPC", 4);
#else
  static const int dyn_bpc = env_int("TWR_DYN_BPC", 8), rom_bpc = env_int("TWR_ROM_BPC", 4);
#endif
  dim3 block(64);
  hipError_t st = hipSuccess;
  if (n_ee < 1 || n_ee > 4) return hipErrorInvalidValue;
  // The fused launch is used while the rom role needs at most TEN rounds of its residency (rom_bpc workgroups per CU x
  // n_cu; round-3 re-tune on one box, ragged sweep, fused vs three launches: 320 / 400 / 512 candidates 75 / 98 / 126 vs
  // 83 / 104 / 128 us per step, 768 / 1024: 187 / 250 vs 183 / 235; round 4: 512 candidates 115 vs 124, 768 / 1024 equal).
  // Ten, not eight: the 512 candidates of a two-GPU shard of the C5 sweep are ~8500 rom slices (the enumeration is ragged:
  // 16.6 slices per candidate), 8.3 rounds on the 256 CUs of an MI355X.  The thresholds are in units of the device's
  // residency, not constants of one chip.
  auto launch_nodes = [&]() {   // the node-based sets (terrain-*, force-*, splineacc-*, swing-*, ...): last launch of every path
    const int n_chunks = n_fam[0] + n_fam[1] + n_fam[2] + n_fam[3];
    if (n_chunks > 0) {
      // persistent waves per CU for ALL families together, shared out by their chunk counts (by count, not by bytes: an iteration
      // costs about the same whatever the family -- weighting the force chunks 1.5 x ... 3 x was 4-15 % slower)
      static const int node_bpc = env_int("TWR_NODE_BPC", 16);
      const int res = node_bpc * n_cu;
      int gf[4];
      for (int f = 0; f < 4; ++f) {
        gf[f] = n_fam[f] == 0 ? 0 : (int)((long long)res * n_fam[f] / n_chunks);
        if (n_fam[f] > 0 && gf[f] < 1) gf[f] = 1;
        if (gf[f] > n_fam[f]) gf[f] = n_fam[f];
      }
  #define TWR_CHUNK_LAUNCH(WG, WJ)                                                                                                              \
    st = twr_first(st, twr_launch(node_chunk_kernel<WG, WJ>, dim3(gf[0] + gf[1] + gf[2] + gf[3]), dim3(64), 0, stream, fam[0], n_fam[0], gf[0],   \
                                  fam[1], n_fam[1], gf[1], fam[2], n_fam[2], gf[2], fam[3], n_fam[3], gf[3], x, g, jac))
      if ((flags & 3) == 3) TWR_CHUNK_LAUNCH(true, true);
      else if (flags & 2) TWR_CHUNK_LAUNCH(false, true);
      else TWR_CHUNK_LAUNCH(true, false);
  #undef TWR_CHUNK_LAUNCH
    } else if (n_node > 0 && node_families == 2)
      st = twr_first(st, twr_launch(node_kernel2, dim3(n_node), dim3(128), 0, stream, node, x, g, jac, flags));
    else if (n_node > 0)
      st = twr_first(st, twr_launch(node_kernel, dim3(n_node), dim3(256), 0, stream, node, x, g, jac, flags));
  };
  const int cap = rom_bpc * n_cu;
  // Values only (no Jacobian), every problem with fixed timings and at most kFlatXCap variables: "dynamic" and "rangeofmotion-*"
  // with one lane per time node (flat items), the node-based sets -- one launch (eval_values_kernel); with per-kernel events three.
  if (!(flags & 2) && (flags & 1) && flat && n_flat > 0 && n_pdyn == 0 && n_prom == 0 && n_ploc == 0) {
    const int nf = !ev && n_chunks_of(n_fam) == 0 ? node_families : 0;   // (large batches: the chunk kernel takes the node sets; with
                                                                          // per-kernel events they are a launch of their own)
    const size_t lds = flat_lds_bytes(flat_max_x);
    const int x_bytes = (int)flat_x_bytes(flat_max_x);
    const int nx = (flat_max_x + 64 * kFlatGroup - 1) / (64 * kFlatGroup);
    const int n_groups = n_flat / kFlatGroup;
    const dim3 vgrid(n_groups + (nf > 0 ? n_node : 0)), vblock(64 * kFlatGroup);
    if (ev) (void)hipEventRecord(ev[0], stream);
    if (nx <= 3) st = twr_first(st, twr_launch(eval_values_kernel<3>, vgrid, vblock, lds, stream, flat, n_groups, x_bytes, node, nf, x, g));
    else if (nx <= 5) st = twr_first(st, twr_launch(eval_values_kernel<5>, vgrid, vblock, lds, stream, flat, n_groups, x_bytes, node, nf, x, g));
    else st = twr_first(st, twr_launch(eval_values_kernel<8>, vgrid, vblock, lds, stream, flat, n_groups, x_bytes, node, nf, x, g));
    static_assert(kFlatXCap <= 8 * 64 * kFlatGroup, "largest instantiation of the values-only kernel");
    if (ev) {   // (the two flat families are one launch: the second interval is empty)
      (void)hipEventRecord(ev[1], stream);
      (void)hipEventRecord(ev[2], stream);
    }
    if (nf == 0) launch_nodes();
    if (ev) (void)hipEventRecord(ev[3], stream);
    return st;
  }
  // With non-temporal stores (sweep-like batches) the fused launch stays ahead for longer -- 768 / 896 / 1024 candidates of the
  // C5 sweep (12.7 / 14.9 / 17 thousand rom slices): 160-165 / 179 / 210 us as three launches, 148 / 166 / 202 us fused -- twenty
  // rounds there (the enumeration ends at 1040 candidates; nothing larger was measured).
  // (round 5, with the roles always co-resident: batches that share one structure, plain stores, 640 / 1024 / 2048 problems
  // 129.5 / 193.5 / 365 us as three launches, 120 / 188 / 376 us fused -- twenty rounds for both store policies)
  const int fused_max = env_int("TWR_FUSED_MAX_ROM", 20 * cap);
  if (!ev && n_pdyn == 0 && n_prom == 0 && n_rom > 0 && n_dyn > 0 && n_rom <= fused_max) {
    int g_rom = n_rom < cap ? n_rom : cap, g_dyn = (n_dyn + 1) / 2 < cap ? (n_dyn + 1) / 2 : cap;
    // When the two persistent roles do not fit the CUs together, the blocks of the later role only start as the earlier
    // ones retire, i.e. the roles run one after the other.  For up to 2560 rom slices (160 quadruped candidates of
    // K = 200) it pays to give each role half of the residency instead, so that the latency-bound dyn waves and the
    // store-bound rom waves overlap from the start (64 / 128 / 160 candidates: 18.7 / 30.1 / 35.2 -> 17.5 / 27.8 /
    // 31.6 us per step; from 200 candidates on the extra rounds cost more than the overlap gains).
    // TWR_FUSED_SPLIT = eighths of the residency given to rom (experiments; 8 = never split).
    static const int split_env = env_int("TWR_FUSED_SPLIT", 0);
    // (round 3: five eighths for rom up to 3200 slices -- 128 / 160 / 200 candidates 27.5 / 33.5 / 40.5 us against 27.7 / 33.9 /
    // 44.0 us with the round-2 rule "half each up to 2560"; from 256 candidates on unsplit was as good or better THEN.)
    // Round 5, re-measured with the round-4 kernels (split tables, nt stores; scripts/r05_split_exp.sh, one box, ragged
    // sweep, us per step unsplit -> split): 256 candidates 53.8 -> 47.2 (five eighths; 48.6 at four), 320: 69.3 -> 62.5 (four),
    // 384: 81.7 -> 74.5, 512: 100.9 -> 92.6, 768: 155.8 -> 145.9, 1024: 206.0 -> 194.1; six eighths loses everywhere.  The two
    // roles are ALWAYS co-resident now: the latency-bound dyn waves fill the holes of the store-bound rom stream.
    const int split = split_env > 0 ? split_env : (8 * n_rom <= 34 * cap ? 5 : 4);   // (4352 slices on 256 CUs)
    if (split < 8 && g_rom + g_dyn > cap) {
      const int r = cap * split / 8, d = cap - r;
      if (g_rom > r) g_rom = r;
      if (g_dyn > d) g_dyn = d;
    }
    {   // experiments (make TUNING=1): explicit grids of the two roles
      static const int e_rom = env_int("TWR_FUSED_GROM", 0), e_dyn = env_int("TWR_FUSED_GDYN", 0);
      if (e_rom > 0) g_rom = e_rom < n_rom ? e_rom : n_rom;
      if (e_dyn > 0) g_dyn = e_dyn < (n_dyn + 1) / 2 ? e_dyn : (n_dyn + 1) / 2;
    }
    if (g_dyn >= 8) g_dyn &= ~7;   // (the XCD-aware slice mapping of the dyn role, eval_fused_kernel)
    const dim3 fgrid(g_rom + g_dyn + 2 * n_node);
    const int need = (rom_max_vals + 1 + 2 + 127) / 128;   // copy-out length of the rom role (see launch_rom_kernel)
#define TWR_FUSED_ARGS fgrid, dim3(128), 0, stream, rom, n_rom, g_rom, dyn, n_dyn, g_dyn, node, x, g, jac, dump
#define TWR_FUSED_LAUNCH(NIT, XC)                                                                              \
  {                                                                                                            \
    if ((flags & 3) == 3 && stream_nt) return twr_launch(eval_fused_kernel<NIT, true, true, XC, true>, TWR_FUSED_ARGS);    \
    if ((flags & 3) == 3) return twr_launch(eval_fused_kernel<NIT, true, true, XC, false>, TWR_FUSED_ARGS);                \
    if ((flags & 2) && stream_nt) return twr_launch(eval_fused_kernel<NIT, false, true, XC, true>, TWR_FUSED_ARGS);        \
    if (flags & 2) return twr_launch(eval_fused_kernel<NIT, false, true, XC, false>, TWR_FUSED_ARGS);                      \
    return twr_launch(eval_fused_kernel<NIT, true, false, XC, false>, TWR_FUSED_ARGS);                                     \
  }
    if (dyn_map_chunks == 2) {
      if (need <= 34) TWR_FUSED_LAUNCH(34, 2)
      TWR_FUSED_LAUNCH(kRomNitMax, 2)
    }
    if (need <= 34) TWR_FUSED_LAUNCH(34, 4)
    TWR_FUSED_LAUNCH(kRomNitMax, 4)
#undef TWR_FUSED_LAUNCH
#undef TWR_FUSED_ARGS
  }
  if (ev) (void)hipEventRecord(ev[0], stream);
  if (n_dyn > 0) {
    const int res = dyn_bpc * n_cu;
    dim3 grid(n_dyn < res ? n_dyn : res);
#define TWR_DYN_ARGS grid, block, 0, stream, dyn, n_dyn, x, g, jac, dump
#define TWR_DYN_LAUNCH(XC)                                                                                                   \
  {                                                                                                                           \
    if ((flags & 3) == 3 && stream_nt) st = twr_first(st, twr_launch(dyn_kernel<true, true, XC, true>, TWR_DYN_ARGS));        \
    else if ((flags & 3) == 3) st = twr_first(st, twr_launch(dyn_kernel<true, true, XC, false>, TWR_DYN_ARGS));               \
    else if ((flags & 2) && stream_nt) st = twr_first(st, twr_launch(dyn_kernel<false, true, XC, true>, TWR_DYN_ARGS));       \
    else if (flags & 2) st = twr_first(st, twr_launch(dyn_kernel<false, true, XC, false>, TWR_DYN_ARGS));                     \
    else st = twr_first(st, twr_launch(dyn_kernel<true, false, XC, false>, TWR_DYN_ARGS));                                    \
  }
    if (dyn_map_chunks == 2) TWR_DYN_LAUNCH(2) else TWR_DYN_LAUNCH(4)
#undef TWR_DYN_LAUNCH
#undef TWR_DYN_ARGS
  }
  // optimised-timings problems: the pre-pass (segment lookup -> records), then the persistent kernels
  if (n_ploc > 0) st = twr_first(st, twr_launch(phase_locate_kernel, dim3(n_ploc), dim3(kLocateThreads), 0, stream, ploc, x));
  if (n_pdyn > 0) {
    // LDS per workgroup: the image of one pass; residency follows from it
    const size_t lds = sizeof(double) * (size_t)((pdyn_img_cap + 1) & ~1);
    static const int pdyn_bpc_env = env_int("TWR_PDYN_BPC", 0);
    int bpc = (int)((size_t)(160 * 1024) / lds);
    if (bpc > 4) bpc = 4;   // one wave per SIMD (the kernel uses the AGPR half of the register file as well)
    if (bpc < 1) bpc = 1;
    if (pdyn_bpc_env > 0) bpc = pdyn_bpc_env;
    const int res = bpc * n_cu;
    dim3 grid(n_pdyn < res ? n_pdyn : res);
    const bool wg = flags & 1, wj = flags & 2;
#define TWR_PDYN_LAUNCH(NIT, WG, WJ) st = twr_first(st, twr_launch(dyn_phase_kernel<NIT, WG, WJ>, grid, block, lds, stream, pdyn, n_pdyn, x, g, jac))
    if (pdyn_img_cap <= 40 * 128) {
      if (wg && wj) TWR_PDYN_LAUNCH(40, true, true);
      else if (wj) TWR_PDYN_LAUNCH(40, false, true);
      else TWR_PDYN_LAUNCH(40, true, false);
    } else {
      if (wg && wj) TWR_PDYN_LAUNCH(0, true, true);
      else if (wj) TWR_PDYN_LAUNCH(0, false, true);
      else TWR_PDYN_LAUNCH(0, true, false);
    }
#undef TWR_PDYN_LAUNCH
  }
  if (ev) (void)hipEventRecord(ev[1], stream);
  if (n_prom > 0) {
    const size_t lds = sizeof(double) * (size_t)((prom_img_cap + 1) & ~1);
    static const int prom_bpc_env = env_int("TWR_PROM_BPC", 0);
    int bpc = (int)((size_t)(160 * 1024) / lds);
    if (bpc > 8) bpc = 8;
    if (bpc < 1) bpc = 1;
    if (prom_bpc_env > 0) bpc = prom_bpc_env;
    const int res = bpc * n_cu;
    dim3 grid(n_prom < res ? n_prom : res);
    const bool wg = flags & 1, wj = flags & 2;
#define TWR_PROM_LAUNCH(NIT, WG, WJ) st = twr_first(st, twr_launch(rom_phase_kernel<NIT, WG, WJ>, grid, block, lds, stream, prom, n_prom, x, g, jac))
#define TWR_PROM_FLAGS(NIT)                           \
  {                                                   \
    if (wg && wj) TWR_PROM_LAUNCH(NIT, true, true);   \
    else if (wj) TWR_PROM_LAUNCH(NIT, false, true);   \
    else TWR_PROM_LAUNCH(NIT, true, false);           \
  }
    if (prom_img_cap <= 24 * 128) TWR_PROM_FLAGS(24)
    else if (prom_img_cap <= 32 * 128) TWR_PROM_FLAGS(32)
    else if (prom_img_cap <= 40 * 128) TWR_PROM_FLAGS(40)
    else {
      TWR_PROM_FLAGS(0)
    }
#undef TWR_PROM_FLAGS
#undef TWR_PROM_LAUNCH
  }
  if (n_rom > 0) {
    const int res = rom_bpc * n_cu;
    dim3 grid(n_rom < res ? n_rom : res);
    st = twr_first(st, launch_rom_kernel((int)grid.x, stream, rom, n_rom, rom_max_vals, x, g, jac, flags, stream_nt));
  }
  if (ev) (void)hipEventRecord(ev[2], stream);
  launch_nodes();
  if (ev) (void)hipEventRecord(ev[3], stream);
  return st;
}

// Images of more than 64 KB need the dynamic-LDS limit of the run-time-length instantiations raised; done when a batch
// is created (on its device, always to the whole 160 KB of a CU), not at launch: an evaluation must stay capturable
// in a hipGraph.
hipError_t prepare_phase_kernels(int pdyn_img_cap, int prom_img_cap) {
  const int whole = 160 * 1024;
  hipError_t st = hipSuccess;
  auto raise = [&](const void* fn) { st = twr_first(st, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, whole)); };
  if ((size_t)pdyn_img_cap * sizeof(double) > 64 * 1024) {
    raise(reinterpret_cast<const void*>(dyn_phase_kernel<0, true, true>));
    raise(reinterpret_cast<const void*>(dyn_phase_kernel<0, false, true>));
    raise(reinterpret_cast<const void*>(dyn_phase_kernel<0, true, false>));
  }
  if ((size_t)prom_img_cap * sizeof(double) > 64 * 1024) {
    raise(reinterpret_cast<const void*>(rom_phase_kernel<0, true, true>));
    raise(reinterpret_cast<const void*>(rom_phase_kernel<0, false, true>));
    raise(reinterpret_cast<const void*>(rom_phase_kernel<0, true, false>));
  }
  return st;
}
int dyn_dump_doubles() { return kDynImage + 2 + 96; }
int node_force_chunk() { return kForceChunk; }
#endif  // !TWR_TU_ROM

}  // namespace twr
