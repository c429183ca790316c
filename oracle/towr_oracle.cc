// TEST INFRASTRUCTURE ONLY -- CPU oracle, see towr_oracle.h for the contract.
//
// Restates, function by function, the reference path
//   ifopt::Problem::{EvaluateConstraints, EvalNonzerosOfJacobian}
//     -> towr::{Terrain,Dynamic,RangeOfMotion,Force}Constraint
//     -> SingleRigidBodyDynamics / EulerConverter / NodeSpline / CubicHermitePolynomial / HeightMap
// All "ref:" citations are file:line under /root/reference/towr/.
// It is deliberately reference-shaped (one Jacobian block per variable set, sparse
// temporaries per time node, std::pow based polynomials): it doubles as the
// "port" CPU baseline in bench.py.
#include "towr_oracle.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>

#include "sparse.h"

namespace orc {

enum Dx { kPos = 0, kVel = 1, kAcc = 2 };
enum { X = 0, Y = 1, Z = 2 };
enum { AX = 0, AY, AZ, LX, LY, LZ };
static const double kInf = 1e20;  // ifopt bounds "infinity"

struct Bound {
  double lo = 0, up = 0;
};

// ---------------------------------------------------------------- state.h:52-131
struct NodeVal {  // Node: position + velocity
  V3 p, v;
  double& at(int deriv, int dim) { return deriv == kPos ? p(dim) : v(dim); }
  double at(int deriv, int dim) const { return deriv == kPos ? p(dim) : v(dim); }
};
struct StateVal {  // State with pos, vel, acc
  V3 p, v, a;
  V3& at(int d) { return d == kPos ? p : (d == kVel ? v : a); }
};

// ------------------------------------------------------- polynomial.cc:38-257
struct CubicHermite {
  NodeVal n0, n1;
  double T = 0.0;
  V3 coeff[4];  // A,B,C,D

  // ref: polynomial.cc:97-104
  void UpdateCoeff() {
    coeff[0] = n0.p;
    coeff[1] = n0.v;
    coeff[2] = (-1.0 * (3.0 * (n0.p - n1.p) + T * (2.0 * n0.v + n1.v))) / std::pow(T, 2);
    coeff[3] = (2.0 * (n0.p - n1.p) + T * (n0.v + n1.v)) / std::pow(T, 3);
  }
  // ref: polynomial.cc:63-72
  static double DerivWrtCoeff(double t, int deriv, int c) {
    switch (deriv) {
      case kPos: return std::pow(t, c);
      case kVel: return c >= 1 ? c * std::pow(t, c - 1) : 0.0;
      case kAcc: return c >= 2 ? c * (c - 1) * std::pow(t, c - 2) : 0.0;
    }
    return 0.0;
  }
  // ref: polynomial.cc:47-61
  StateVal GetPoint(double t_local) const {
    StateVal out;
    for (int d : {kPos, kVel, kAcc})
      for (int c = 0; c < 4; ++c) out.at(d) = out.at(d) + DerivWrtCoeff(t_local, d, c) * coeff[c];
    return out;
  }
  // ref: polynomial.cc:140-186 (start node) -- node_value is kPos or kVel
  double DerivWrtStartNode(int dfdt, int node_value, double t) const {
    double t2 = std::pow(t, 2), t3 = std::pow(t, 3);
    double T2 = std::pow(T, 2), T3 = std::pow(T, 3);
    switch (dfdt) {
      case kPos: return node_value == kPos ? (2 * t3) / T3 - (3 * t2) / T2 + 1 : t - (2 * t2) / T + t3 / T2;
      case kVel: return node_value == kPos ? (6 * t2) / T3 - (6 * t) / T2 : (3 * t2) / T2 - (4 * t) / T + 1;
      case kAcc: return node_value == kPos ? (12 * t) / T3 - 6 / T2 : (6 * t) / T2 - 4 / T;
    }
    return 0.0;
  }
  // ref: polynomial.cc:236-257
  V3 GetDerivativeOfPosWrtDuration(double t) const {
    V3 x0 = n0.p, x1 = n1.p, v0 = n0.v, v1 = n1.v;
    double t2 = std::pow(t, 2), t3 = std::pow(t, 3);
    double T2 = std::pow(T, 2), T3 = std::pow(T, 3), T4 = std::pow(T, 4);
    return (t3 * (v0 + v1)) / T3 - (t2 * (2.0 * v0 + v1)) / T2 - (3 * t3 * (2.0 * x0 - 2.0 * x1 + T * v0 + T * v1)) / T4 +
           (2 * t2 * (3.0 * x0 - 3.0 * x1 + 2 * T * v0 + T * v1)) / T3;
  }
  // ref: polynomial.cc:188-234 (end node)
  double DerivWrtEndNode(int dfdt, int node_value, double t) const {
    double t2 = std::pow(t, 2), t3 = std::pow(t, 3);
    double T2 = std::pow(T, 2), T3 = std::pow(T, 3);
    switch (dfdt) {
      case kPos: return node_value == kPos ? (3 * t2) / T2 - (2 * t3) / T3 : t3 / T2 - t2 / T;
      case kVel: return node_value == kPos ? (6 * t) / T2 - (6 * t2) / T3 : (3 * t2) / T2 - (2 * t) / T;
      case kAcc: return node_value == kPos ? 6 / T2 - (12 * t) / T3 : (6 * t) / T2 - 2 / T;
    }
    return 0.0;
  }
};

// ------------------------------------------- nodes_variables*.cc (variable sets)
struct NVI {
  int id, deriv, dim;
  bool operator==(const NVI& o) const { return id == o.id && deriv == o.deriv && dim == o.dim; }
};
struct PolyInfo {  // ref: nodes_variables_phase_based.h:70-82
  int phase, poly_in_phase, n_polys_in_phase;
  bool is_constant;
};

struct NodesVars {
  std::string name;
  std::vector<NodeVal> nodes;
  std::vector<std::vector<NVI>> info;  // GetNodeValuesInfo(idx)
  std::vector<Bound> bounds;
  std::vector<PolyInfo> poly_info;  // phase-based sets only
  struct NodeSpline* observer = nullptr;

  int rows() const { return (int)info.size(); }
  int poly_count() const { return (int)nodes.size() - 1; }

  // ref: nodes_variables.cc:40-50 (linear search, first match)
  int GetOptIndex(const NVI& want) const {
    for (int idx = 0; idx < rows(); ++idx)
      for (const NVI& nvi : info[idx])
        if (nvi == want) return idx;
    return -1;
  }
  // ref: nodes_variables.cc:52-62
  void GetValues(double* x) const {
    for (int idx = 0; idx < rows(); ++idx)
      for (const NVI& nvi : info[idx]) x[idx] = nodes.at(nvi.id).at(nvi.deriv, nvi.dim);
  }
  void SetVariables(const double* x);  // ref: nodes_variables.cc:64-72 (below, needs spline)

  // ref: nodes_variables.cc:126-150
  void SetByLinearInterpolation(const V3& initial_val, const V3& final_val, double t_total) {
    V3 dp = final_val - initial_val;
    V3 average_velocity = dp / t_total;
    int num_nodes = (int)nodes.size();
    for (int idx = 0; idx < rows(); ++idx)
      for (const NVI& nvi : info[idx]) {
        if (nvi.deriv == kPos) {
          V3 pos = initial_val + (nvi.id / static_cast<double>(num_nodes - 1)) * dp;
          nodes.at(nvi.id).p(nvi.dim) = pos(nvi.dim);
        }
        if (nvi.deriv == kVel) nodes.at(nvi.id).v(nvi.dim) = average_velocity(nvi.dim);
      }
  }
  // ref: nodes_variables.cc:152-181
  void AddBounds(int node_id, int deriv, const std::vector<int>& dimensions, const V3& val) {
    for (int dim : dimensions) AddBound(NVI{node_id, deriv, dim}, val(dim));
  }
  void AddStartBound(int deriv, const std::vector<int>& dimensions, const V3& val) { AddBounds(0, deriv, dimensions, val); }
  void AddFinalBound(int deriv, const std::vector<int>& dimensions, const V3& val) {
    AddBounds((int)nodes.size() - 1, deriv, dimensions, val);
  }
  void AddBound(const NVI& want, double val) {
    for (int idx = 0; idx < rows(); ++idx)
      for (const NVI& nvi : info[idx])
        if (nvi == want) bounds.at(idx) = {val, val};
  }

  // --- phase based helpers, ref: nodes_variables_phase_based.cc:99-179
  std::vector<int> AdjacentPolyIds(int node_id) const {
    int last = (int)nodes.size() - 1;
    if (node_id == 0) return {0};
    if (node_id == last) return {last - 1};
    return {node_id - 1, node_id};
  }
  bool IsConstantNode(int node_id) const {
    bool c = false;
    for (int p : AdjacentPolyIds(node_id))
      if (poly_info.at(p).is_constant) c = true;
    return c;
  }
  std::vector<int> NonConstantNodes() const {
    std::vector<int> ids;
    for (int id = 0; id < (int)nodes.size(); ++id)
      if (!IsConstantNode(id)) ids.push_back(id);
    return ids;
  }
  int GetPhase(int node_id) const { return poly_info.at(AdjacentPolyIds(node_id).front()).phase; }
  int NodeIdAtStartOfPhase(int phase) const {
    for (int i = 0; i < (int)poly_info.size(); ++i)
      if (poly_info[i].phase == phase) return i;  // GetNodeId(poly, Start) == poly
    throw std::runtime_error("phase not found");
  }
  std::vector<double> PhaseToPolyDurations(const std::vector<double>& phase_durations) const {
    std::vector<double> d;  // ref: nodes_variables_phase_based.cc:73-84
    for (int i = 0; i < poly_count(); ++i)
      d.push_back(phase_durations.at(poly_info[i].phase) / poly_info[i].n_polys_in_phase);
    return d;
  }
};

// ref: nodes_variables_all.cc:34-61
static NodesVars MakeNodesAll(int n_nodes, const std::string& name) {
  NodesVars nv;
  nv.name = name;
  nv.nodes.assign(n_nodes, NodeVal());
  int n = n_nodes * 2 * 3;
  for (int idx = 0; idx < n; ++idx) {
    int per_node = 2 * 3;
    int internal = idx % per_node;  // p.x p.y p.z v.x v.y v.z
    NVI nvi;
    nvi.deriv = internal < 3 ? kPos : kVel;
    nvi.dim = internal % 3;
    nvi.id = idx / per_node;
    nv.info.push_back({nvi});
  }
  nv.bounds.assign(n, Bound{-kInf, kInf});
  return nv;
}

// ref: nodes_variables_phase_based.cc:38-58
static std::vector<PolyInfo> BuildPolyInfos(int phase_count, bool first_phase_constant, int n_polys_changing) {
  std::vector<PolyInfo> v;
  bool constant = first_phase_constant;
  for (int i = 0; i < phase_count; ++i) {
    if (constant)
      v.push_back({i, 0, 1, true});
    else
      for (int j = 0; j < n_polys_changing; ++j) v.push_back({i, j, n_polys_changing, false});
    constant = !constant;
  }
  return v;
}

// ref: nodes_variables_phase_based.cc:197-253
static NodesVars MakeNodesEEMotion(int phase_count, bool contact_at_start, const std::string& name, int n_polys_swing) {
  NodesVars nv;
  nv.name = name;
  nv.poly_info = BuildPolyInfos(phase_count, contact_at_start, n_polys_swing);
  nv.nodes.assign(nv.poly_info.size() + 1, NodeVal());
  for (int node_id = 0; node_id < (int)nv.nodes.size(); ++node_id) {
    if (!nv.IsConstantNode(node_id)) {  // swing node: px,vx,py,vy,pz (vz fixed 0)
      for (int dim = 0; dim < 3; ++dim) {
        nv.info.push_back({NVI{node_id, kPos, dim}});
        if (dim == Z)
          nv.nodes.at(node_id).v(Z) = 0.0;
        else
          nv.info.push_back({NVI{node_id, kVel, dim}});
      }
    } else {  // stance: one position variable shared by both nodes of the polynomial
      nv.nodes.at(node_id).v = V3();
      nv.nodes.at(node_id + 1).v = V3();
      for (int dim = 0; dim < 3; ++dim) nv.info.push_back({NVI{node_id, kPos, dim}, NVI{node_id + 1, kPos, dim}});
      node_id += 1;
    }
  }
  nv.bounds.assign(nv.info.size(), Bound{-kInf, kInf});
  return nv;
}

// ref: nodes_variables_phase_based.cc:255-298
static NodesVars MakeNodesEEForce(int phase_count, bool contact_at_start, const std::string& name, int n_polys_stance) {
  NodesVars nv;
  nv.name = name;
  nv.poly_info = BuildPolyInfos(phase_count, !contact_at_start, n_polys_stance);
  nv.nodes.assign(nv.poly_info.size() + 1, NodeVal());
  for (int id = 0; id < (int)nv.nodes.size(); ++id) {
    if (!nv.IsConstantNode(id)) {  // stance node: px,vx,py,vy,pz,vz
      for (int dim = 0; dim < 3; ++dim) {
        nv.info.push_back({NVI{id, kPos, dim}});
        nv.info.push_back({NVI{id, kVel, dim}});
      }
    } else {  // swing: force identically zero, not optimised
      nv.nodes.at(id) = NodeVal();
      nv.nodes.at(id + 1) = NodeVal();
      id += 1;
    }
  }
  nv.bounds.assign(nv.info.size(), Bound{-kInf, kInf});
  return nv;
}

// ------------------------------------------------ spline.cc / node_spline.cc
struct NodeSpline {
  NodesVars* nv = nullptr;
  std::vector<CubicHermite> polys;
  SpMat jac_wrt_nodes_structure;  // node_spline.h:108; empty for a NodeSpline, all variables for a PhaseSpline

  NodeSpline(NodesVars* nodes, const std::vector<double>& durations) : nv(nodes) {
    polys.assign(durations.size(), CubicHermite());
    for (size_t i = 0; i < durations.size(); ++i) polys[i].T = durations[i];
    UpdateNodes();
    jac_wrt_nodes_structure = SpMat(3, nv->rows());  // node_spline.cc:42
  }
  virtual ~NodeSpline() = default;
  // ref: node_spline.h:96-100 (only a PhaseSpline can answer)
  virtual SpMat GetJacobianOfPosWrtDurations(double) const { throw std::runtime_error("durations are fixed"); }
  // ref: node_spline.cc:45-54, nodes_variables.cc:93-100
  void UpdateNodes() {
    for (size_t i = 0; i < polys.size(); ++i) {
      polys[i].n0 = nv->nodes.at(i);
      polys[i].n1 = nv->nodes.at(i + 1);
    }
    for (auto& p : polys) p.UpdateCoeff();
  }
  // ref: spline.cc:108-116 (a fresh vector per call, like the reference)
  std::vector<double> GetPolyDurations() const {
    std::vector<double> d;
    for (const auto& p : polys) d.push_back(p.T);
    return d;
  }
  // ref: spline.cc:48-66
  static int GetSegmentID(double t_global, const std::vector<double>& durations) {
    double eps = 1e-10;
    double t = 0;
    int i = 0;
    for (double d : durations) {
      t += d;
      if (t >= t_global - eps) return i;  // at junctions returns the previous segment
      i++;
    }
    throw std::runtime_error("GetSegmentID: t beyond spline");
  }
  // ref: spline.cc:68-78
  static std::pair<int, double> GetLocalTime(double t_global, const std::vector<double>& durations) {
    int id = GetSegmentID(t_global, durations);
    double t_local = t_global;
    for (int i = 0; i < id; i++) t_local -= durations.at(i);
    return {id, t_local};
  }
  // ref: spline.cc:80-93
  StateVal GetPoint(double t_global) const {
    auto lt = GetLocalTime(t_global, GetPolyDurations());
    return polys.at(lt.first).GetPoint(lt.second);
  }
  // ref: spline.cc:95-99
  StateVal GetPoint(int poly_id, double t_local) const { return polys.at(poly_id).GetPoint(t_local); }
  // ref: node_spline.cc:62-70
  SpMat GetJacobianWrtNodes(double t_global, int dxdt) const {
    auto lt = GetLocalTime(t_global, GetPolyDurations());
    return GetJacobianWrtNodes(lt.first, lt.second, dxdt);
  }
  // ref: node_spline.cc:72-82
  SpMat GetJacobianWrtNodes(int poly_id, double t_local, int dxdt) const {
    SpMat jac = jac_wrt_nodes_structure;
    FillJacobianWrtNodes(poly_id, t_local, dxdt, jac, false);
    return jac;
  }
  // ref: node_spline.cc:84-112
  void FillJacobianWrtNodes(int poly_id, double t_local, int dxdt, SpMat& jac, bool fill_with_zeros) const {
    for (int idx = 0; idx < jac.c; ++idx)
      for (const NVI& nvi : nv->info[idx])
        for (int side : {0, 1}) {  // Start, End
          int node = poly_id + side;
          if (node == nvi.id) {
            double val = side == 0 ? polys.at(poly_id).DerivWrtStartNode(dxdt, nvi.deriv, t_local)
                                   : polys.at(poly_id).DerivWrtEndNode(dxdt, nvi.deriv, t_local);
            if (fill_with_zeros) val = 0.0;
            jac.coeffRef(nvi.dim, idx) += val;
          }
        }
  }
};

// ------------------------------------------------ phase_durations.cc / phase_spline.cc (optimised timings)
struct PhaseSpline;
// ref: phase_durations.cc:39-154
struct PhaseDurations {
  std::string name;
  std::vector<double> durations;
  double t_total = 0;
  Bound phase_duration_bounds;
  bool initial_contact_state = true;
  std::vector<PhaseSpline*> observers;

  PhaseDurations(int ee, const std::vector<double>& timings, bool is_first_phase_in_contact, double min_duration,
                 double max_duration) {
    name = "ee-schedule" + std::to_string(ee);  // variable_names.h:47,60-63
    durations = timings;
    t_total = std::accumulate(timings.begin(), timings.end(), 0.0);
    phase_duration_bounds = Bound{min_duration, max_duration};
    initial_contact_state = is_first_phase_in_contact;
  }
  int rows() const { return (int)durations.size() - 1; }  // the last phase fills up to the total time
  void GetValues(double* x) const {
    for (int i = 0; i < rows(); ++i) x[i] = durations.at(i);
  }
  void SetVariables(const double* x);  // below (needs PhaseSpline)
  // ref: phase_durations.cc:126-154.  Dense 3 x rows() matrix turned into a sparse one that keeps
  // every entry, zeros included (sparseView(1.0, -1.0)).
  SpMat GetJacobianOfPos(int current_phase, const V3& dx_dT, const V3& xd) const {
    int n = rows();
    std::vector<V3> col(n);
    bool in_last_phase = (current_phase == (int)durations.size() - 1);
    if (!in_last_phase) col.at(current_phase) = dx_dT;
    for (int phase = 0; phase < current_phase; ++phase) {
      col.at(phase) = -1.0 * xd;
      if (in_last_phase) col.at(phase) = col.at(phase) - dx_dT;
    }
    SpMat jac(3, n);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < n; ++j) jac.coeffRef(i, j) = col[j](i);
    return jac;
  }
};

// ref: phase_spline.cc:35-93
struct PhaseSpline : NodeSpline {
  PhaseDurations* phase_durations;
  PhaseSpline(NodesVars* nodes, PhaseDurations* pd)
      : NodeSpline(nodes, nodes->PhaseToPolyDurations(pd->durations)), phase_durations(pd) {
    pd->observers.push_back(this);
    UpdatePolynomialDurations();
    // "assume every global time can fall into every polynomial": the structure holds every variable
    for (int i = 0; i < nv->poly_count(); ++i) FillJacobianWrtNodes(i, 0.0, kPos, jac_wrt_nodes_structure, true);
  }
  void UpdatePolynomialDurations() {
    auto poly_durations = nv->PhaseToPolyDurations(phase_durations->durations);
    for (size_t i = 0; i < polys.size(); ++i) polys.at(i).T = poly_durations.at(i);
    for (auto& p : polys) p.UpdateCoeff();
  }
  SpMat GetJacobianOfPosWrtDurations(double t_global) const override {
    V3 dx_dT = GetDerivativeOfPosWrtPhaseDuration(t_global);
    V3 xd = GetPoint(t_global).v;
    int current_phase = GetSegmentID(t_global, phase_durations->durations);
    return phase_durations->GetJacobianOfPos(current_phase, dx_dT, xd);
  }
  V3 GetDerivativeOfPosWrtPhaseDuration(double t_global) const {
    auto lt = GetLocalTime(t_global, GetPolyDurations());
    int poly_id = lt.first;
    double t_local = lt.second;
    V3 vel = GetPoint(t_global).v;
    V3 dxdT = polys.at(poly_id).GetDerivativeOfPosWrtDuration(t_local);
    // ref: nodes_variables_phase_based.cc:86-98
    double inner_derivative = 1. / nv->poly_info.at(poly_id).n_polys_in_phase;
    double prev_polys_in_phase = nv->poly_info.at(poly_id).poly_in_phase;
    return inner_derivative * (dxdT - prev_polys_in_phase * vel);
  }
};

// ref: phase_durations.cc:77-103 (the assert t_total > sum is compiled out in Release; Eigen's x.sum()
// is restated as a sequential sum, which can differ from its vectorised reduction in the last bit)
void PhaseDurations::SetVariables(const double* x) {
  double sum = 0.0;
  for (int i = 0; i < rows(); ++i) {
    durations.at(i) = x[i];
    sum += x[i];
  }
  durations.back() = t_total - sum;
  for (PhaseSpline* spline : observers) spline->UpdatePolynomialDurations();
}

void NodesVars::SetVariables(const double* x) {
  for (int idx = 0; idx < rows(); ++idx)
    for (const NVI& nvi : info[idx]) nodes.at(nvi.id).at(nvi.deriv, nvi.dim) = x[idx];
  if (observer) observer->UpdateNodes();
}

// ------------------------------------------------------- euler_converter.cc
struct EulerConverter {
  const NodeSpline* euler = nullptr;
  int n() const { return euler->nv->rows(); }

  // ref: euler_converter.cc:133-148
  static SpMat GetM(const V3& xyz) {
    double z = xyz(Z), y = xyz(Y);
    SpMat M(3, 3);
    M.coeffRef(0, Y) = -sin(z); M.coeffRef(0, X) = cos(y) * cos(z);
    M.coeffRef(1, Y) = cos(z);  M.coeffRef(1, X) = cos(y) * sin(z);
    M.coeffRef(2, Z) = 1.0;     M.coeffRef(2, X) = -sin(y);
    return M;
  }
  // ref: euler_converter.cc:150-166
  static SpMat GetMdot(const V3& xyz, const V3& xyz_d) {
    double z = xyz(Z), zd = xyz_d(Z), y = xyz(Y), yd = xyz_d(Y);
    SpMat Md(3, 3);
    Md.coeffRef(0, Y) = -cos(z) * zd; Md.coeffRef(0, X) = -cos(z) * sin(y) * yd - cos(y) * sin(z) * zd;
    Md.coeffRef(1, Y) = -sin(z) * zd; Md.coeffRef(1, X) = cos(y) * cos(z) * zd - sin(y) * sin(z) * yd;
    Md.coeffRef(2, X) = -cos(y) * yd;
    return Md;
  }
  // ref: euler_converter.cc:207-221
  static M3 RotationDense(const V3& xyz) {
    double x = xyz(X), y = xyz(Y), z = xyz(Z);
    M3 M;
    M(0, 0) = cos(y) * cos(z); M(0, 1) = cos(z) * sin(x) * sin(y) - cos(x) * sin(z); M(0, 2) = sin(x) * sin(z) + cos(x) * cos(z) * sin(y);
    M(1, 0) = cos(y) * sin(z); M(1, 1) = cos(x) * cos(z) + sin(x) * sin(y) * sin(z); M(1, 2) = cos(x) * sin(y) * sin(z) - cos(z) * sin(x);
    M(2, 0) = -sin(y);         M(2, 1) = cos(y) * sin(x);                            M(2, 2) = cos(x) * cos(y);
    return M;
  }
  // sparse with every entry kept (sparseView(1.0,-1.0)), ref: euler_converter.cc:220
  SpMat GetRotationMatrixBaseToWorld(double t) const { return sparse_view_full(RotationDense(euler->GetPoint(t).p)); }
  // ref: euler_converter.cc:58-83
  V3 GetAngularVelocityInWorld(double t) const {
    StateVal ori = euler->GetPoint(t);
    return GetM(ori.p) * ori.v;
  }
  V3 GetAngularAccelerationInWorld(double t) const {
    StateVal ori = euler->GetPoint(t);
    return GetMdot(ori.p, ori.v) * ori.v + GetM(ori.p) * ori.a;
  }
  // ref: euler_converter.cc:306-310
  SpVec GetJac(double t, int deriv, int dim) const { return euler->GetJacobianWrtNodes(t, deriv).rows[dim]; }

  // ref: euler_converter.cc:168-198
  SpMat GetDerivMwrtNodes(double t, int ang_acc_dim) const {
    StateVal ori = euler->GetPoint(t);
    double z = ori.p(Z), y = ori.p(Y);
    SpVec jac_z = GetJac(t, kPos, Z), jac_y = GetJac(t, kPos, Y);
    SpMat jac(3, n());
    switch (ang_acc_dim) {
      case X:
        jac.rows[Y] = -cos(z) * jac_z;
        jac.rows[X] = -cos(z) * sin(y) * jac_y - cos(y) * sin(z) * jac_z;
        break;
      case Y:
        jac.rows[Y] = -sin(z) * jac_z;
        jac.rows[X] = cos(y) * cos(z) * jac_z - sin(y) * sin(z) * jac_y;
        break;
      case Z:
        jac.rows[X] = -cos(y) * jac_y;
        break;
    }
    return jac;
  }
  // ref: euler_converter.cc:270-304
  SpMat GetDerivMdotwrtNodes(double t, int ang_acc_dim) const {
    StateVal ori = euler->GetPoint(t);
    double z = ori.p(Z), zd = ori.v(Z), y = ori.p(Y), yd = ori.v(Y);
    SpVec jac_z = GetJac(t, kPos, Z), jac_y = GetJac(t, kPos, Y);
    SpVec jac_zd = GetJac(t, kVel, Z), jac_yd = GetJac(t, kVel, Y);
    SpMat jac(3, n());
    switch (ang_acc_dim) {
      case X:
        jac.rows[Y] = sin(z) * zd * jac_z - cos(z) * jac_zd;
        jac.rows[X] = sin(y) * sin(z) * yd * jac_z - cos(y) * sin(z) * jac_zd - cos(y) * cos(z) * yd * jac_y -
                      cos(y) * cos(z) * zd * jac_z - cos(z) * sin(y) * jac_yd + (sin(y) * sin(z)) * jac_y * zd;
        break;
      case Y:
        jac.rows[Y] = (-sin(z)) * jac_zd - cos(z) * zd * jac_z;
        jac.rows[X] = cos(y) * cos(z) * jac_zd - sin(y) * sin(z) * jac_yd - cos(y) * sin(z) * yd * jac_y -
                      cos(z) * sin(y) * yd * jac_z - (cos(z) * sin(y)) * jac_y * zd - cos(y) * sin(z) * zd * jac_z;
        break;
      case Z:
        jac.rows[X] = sin(y) * yd * jac_y - cos(y) * jac_yd;
        break;
    }
    return jac;
  }
  // ref: euler_converter.cc:85-102
  SpMat GetDerivOfAngVelWrtEulerNodes(double t) const {
    SpMat jac(3, n());
    StateVal ori = euler->GetPoint(t);
    SpVec vel = sparse_view_full_row(ori.v);
    SpMat dVel_du = euler->GetJacobianWrtNodes(t, kVel);
    for (int dim : {X, Y, Z}) {
      SpMat dM_du = GetDerivMwrtNodes(t, dim);
      jac.rows[dim] = vel * dM_du + GetM(ori.p).rows[dim] * dVel_du;
    }
    return jac;
  }
  // ref: euler_converter.cc:104-131
  SpMat GetDerivOfAngAccWrtEulerNodes(double t) const {
    SpMat jac(3, n());
    StateVal ori = euler->GetPoint(t);
    SpVec vel = sparse_view_full_row(ori.v);
    SpVec acc = sparse_view_full_row(ori.a);
    SpMat dVel_du = euler->GetJacobianWrtNodes(t, kVel);
    SpMat dAcc_du = euler->GetJacobianWrtNodes(t, kAcc);
    for (int dim : {X, Y, Z}) {
      SpMat dMdot_du = GetDerivMdotwrtNodes(t, dim);
      SpMat dM_du = GetDerivMwrtNodes(t, dim);
      jac.rows[dim] = vel * dMdot_du + GetMdot(ori.p, ori.v).rows[dim] * dVel_du + acc * dM_du +
                      GetM(ori.p).rows[dim] * dAcc_du;
    }
    return jac;
  }
  // ref: euler_converter.cc:241-268  (cell-wise derivative of R w.r.t. node values)
  void GetDerivativeOfRotationMatrixWrtNodes(double t, SpVec Rd[3][3]) const {
    StateVal ori = euler->GetPoint(t);
    double x = ori.p(X), y = ori.p(Y), z = ori.p(Z);
    SpVec jx = GetJac(t, kPos, X), jy = GetJac(t, kPos, Y), jz = GetJac(t, kPos, Z);
    Rd[X][X] = -cos(z) * sin(y) * jy - cos(y) * sin(z) * jz;
    Rd[X][Y] = sin(x) * sin(z) * jx - cos(x) * cos(z) * jz - sin(x) * sin(y) * sin(z) * jz + cos(x) * cos(z) * sin(y) * jx + cos(y) * cos(z) * sin(x) * jy;
    Rd[X][Z] = cos(x) * sin(z) * jx + cos(z) * sin(x) * jz - cos(z) * sin(x) * sin(y) * jx - cos(x) * sin(y) * sin(z) * jz + cos(x) * cos(y) * cos(z) * jy;
    Rd[Y][X] = cos(y) * cos(z) * jz - sin(y) * sin(z) * jy;
    Rd[Y][Y] = cos(x) * sin(y) * sin(z) * jx - cos(x) * sin(z) * jz - cos(z) * sin(x) * jx + cos(y) * sin(x) * sin(z) * jy + cos(z) * sin(x) * sin(y) * jz;
    Rd[Y][Z] = sin(x) * sin(z) * jz - cos(x) * cos(z) * jx - sin(x) * sin(y) * sin(z) * jx + cos(x) * cos(y) * sin(z) * jy + cos(x) * cos(z) * sin(y) * jz;
    Rd[Z][X] = -cos(y) * jy;
    Rd[Z][Y] = cos(x) * cos(y) * jx - sin(x) * sin(y) * jy;
    Rd[Z][Z] = -cos(y) * sin(x) * jx - cos(x) * sin(y) * jy;
  }
  // ref: euler_converter.cc:223-239
  SpMat DerivOfRotVecMult(double t, const V3& v, bool inverse) const {
    SpVec Rd[3][3];
    GetDerivativeOfRotationMatrixWrtNodes(t, Rd);
    SpMat jac(3, n());
    for (int row : {X, Y, Z})
      for (int col : {X, Y, Z}) {
        const SpVec& jr = inverse ? Rd[col][row] : Rd[row][col];
        jac.rows[row] = jac.rows[row] + v(col) * jr;
      }
    return jac;
  }
};

// ------------------------------- single_rigid_body_dynamics.cc / dynamic_model.cc
struct RobotConsts {
  int n_ee;
  double mass, Ixx, Iyy, Izz, Ixy, Ixz, Iyz;
  V3 nominal[4];
  V3 max_dev;
};
// ref: models/examples/{monoped,biped,hyq,anymal}_model.h, models/go1/go1_model.h, robot_model.cc:41-68
static RobotConsts MakeRobot(int robot) {
  RobotConsts r{};
  auto quad = [&](double xn, double yn, double zn) {
    r.nominal[0] = V3(xn, yn, zn);    // LF
    r.nominal[1] = V3(xn, -yn, zn);   // RF
    r.nominal[2] = V3(-xn, yn, zn);   // LH
    r.nominal[3] = V3(-xn, -yn, zn);  // RH
  };
  switch (robot) {
    case 0:  // Monoped
      r = {1, 20, 1.2, 5.5, 6.0, 0.0, -0.2, -0.01, {}, V3(0.25, 0.15, 0.2)};
      r.nominal[0] = V3(0.0, 0.0, -0.58);
      break;
    case 1:  // Biped
      r = {2, 20, 1.209, 5.583, 6.056, 0.005, -0.190, -0.012, {}, V3(0.25, 0.15, 0.15)};
      r.nominal[0] = V3(0.0, 0.20, -0.65);
      r.nominal[1] = V3(0.0, -0.20, -0.65);
      break;
    case 2:  // HyQ
      r = {4, 83, 4.26, 8.97, 9.88, -0.0063, 0.193, 0.0126, {}, V3(0.25, 0.20, 0.10)};
      quad(0.31, 0.29, -0.58);
      break;
    case 3:  // ANYmal
      r = {4, 29.5, 0.946438, 1.94478, 2.01835, 0.000938112, -0.00595386, -0.00146328, {}, V3(0.15, 0.1, 0.10)};
      quad(0.34, 0.19, -0.42);
      break;
    case 4:  // Go1
      r = {4, 12.84, 0.0168128557, 0.063009565, 0.0716547275, -0.0002296769, -0.0002945293, -0.0000418731, {}, V3(0.16, 0.12, 0.06)};
      quad(0.1881, 0.04675 + 0.08, -0.3);
      break;
    default: throw std::runtime_error("unknown robot");
  }
  return r;
}

// ref: single_rigid_body_dynamics.cc:46-57 (coeffRef => six structural entries whatever the value)
static SpMat Cross(const V3& in) {
  SpMat out(3, 3);
  out.coeffRef(0, 1) = -in(2); out.coeffRef(0, 2) = in(1);
  out.coeffRef(1, 0) = in(2);  out.coeffRef(1, 2) = -in(0);
  out.coeffRef(2, 0) = -in(1); out.coeffRef(2, 1) = in(0);
  return out;
}

struct SRBD {
  double m_, g_ = 9.80665;  // ref: dynamic_model.cc:34-38
  SpMat I_b;                // ref: single_rigid_body_dynamics.cc:36-44,73 (sparseView prunes exact zeros)
  V3 com_pos, com_acc, omega, omega_dot;
  M3 w_R_b;
  std::vector<V3> ee_force, ee_pos;

  SRBD(const RobotConsts& r) : m_(r.mass) {
    M3 I;
    I(0, 0) = r.Ixx;  I(0, 1) = -r.Ixy; I(0, 2) = -r.Ixz;
    I(1, 0) = -r.Ixy; I(1, 1) = r.Iyy;  I(1, 2) = -r.Iyz;
    I(2, 0) = -r.Ixz; I(2, 1) = -r.Iyz; I(2, 2) = r.Izz;
    I_b = sparse_view_pruned(I);
    ee_force.assign(r.n_ee, V3());
    ee_pos.assign(r.n_ee, V3());
  }
  SpMat Iw() const {  // ref: single_rigid_body_dynamics.cc:91,127
    return (sparse_view_pruned(w_R_b) * I_b) * sparse_view_pruned(w_R_b.transpose());
  }
  // ref: single_rigid_body_dynamics.cc:76-101
  void GetDynamicViolation(double acc[6]) const {
    V3 f_sum, tau_sum;
    for (size_t ee = 0; ee < ee_pos.size(); ++ee) {
      V3 f = ee_force[ee];
      tau_sum = tau_sum + cross(f, com_pos - ee_pos[ee]);
      f_sum = f_sum + f;
    }
    SpMat I_w = Iw();
    V3 ang = (I_w * omega_dot + Cross(omega) * (I_w * omega)) - tau_sum;
    V3 lin = (m_ * com_acc - f_sum) - V3(0.0, 0.0, -m_ * g_);
    for (int i = 0; i < 3; ++i) { acc[AX + i] = ang(i); acc[LX + i] = lin(i); }
  }
  static void SetMiddleRows(SpMat& dst, int r0, const SpMat& src) {
    for (int i = 0; i < src.r; ++i) dst.rows[r0 + i] = src.rows[i];
  }
  // ref: single_rigid_body_dynamics.cc:103-121
  SpMat GetJacobianWrtBaseLin(const SpMat& jac_pos, const SpMat& jac_acc) const {
    int n = jac_pos.c;
    SpMat jac_tau_sum(3, n);
    for (const V3& f : ee_force) jac_tau_sum = jac_tau_sum + Cross(f) * jac_pos;
    SpMat jac(6, n);
    SetMiddleRows(jac, AX, -jac_tau_sum);
    SetMiddleRows(jac, LX, m_ * jac_acc);
    return jac;
  }
  // ref: single_rigid_body_dynamics.cc:123-165
  SpMat GetJacobianWrtBaseAng(const EulerConverter& base_euler, double t) const {
    SpMat I_w = Iw();
    SpMat R = sparse_view_pruned(w_R_b);
    M3 Rt = w_R_b.transpose();
    // I_b*R^T as a dense 3x3 (sparse * dense), then times the vector
    auto IbRt_times = [&](const V3& v) {
      M3 P;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (auto& e : I_b.rows[i].e) s += e.second * Rt(e.first, j);
          P(i, j) = s;
        }
      return P * v;
    };
    V3 v11 = IbRt_times(omega_dot);
    SpMat jac11 = base_euler.DerivOfRotVecMult(t, v11, false);
    SpMat jac12 = (R * I_b) * base_euler.DerivOfRotVecMult(t, omega_dot, true);
    SpMat jac_ang_acc = base_euler.GetDerivOfAngAccWrtEulerNodes(t);
    SpMat jac13 = I_w * jac_ang_acc;
    SpMat jac1 = (jac11 + jac12) + jac13;

    V3 v21 = IbRt_times(omega);
    SpMat jac21 = base_euler.DerivOfRotVecMult(t, v21, false);
    SpMat jac22 = (R * I_b) * base_euler.DerivOfRotVecMult(t, omega, true);
    SpMat jac_ang_vel = base_euler.GetDerivOfAngVelWrtEulerNodes(t);
    SpMat jac23 = I_w * jac_ang_vel;
    SpMat jac2 = Cross(omega) * ((jac21 + jac22) + jac23) - Cross(I_w * omega) * jac_ang_vel;

    SpMat jac(6, jac_ang_vel.c);
    SetMiddleRows(jac, AX, jac1 + jac2);
    return jac;
  }
  // ref: single_rigid_body_dynamics.cc:167-179
  SpMat GetJacobianWrtForce(const SpMat& jac_force, int ee) const {
    V3 r = com_pos - ee_pos.at(ee);
    SpMat jac_tau = (-Cross(r)) * jac_force;
    SpMat jac(6, jac_force.c);
    SetMiddleRows(jac, AX, -jac_tau);
    SetMiddleRows(jac, LX, -jac_force);
    return jac;
  }
  // ref: single_rigid_body_dynamics.cc:181-192
  SpMat GetJacobianWrtEEPos(const SpMat& jac_ee_pos, int ee) const {
    V3 f = ee_force.at(ee);
    SpMat jac_tau = Cross(f) * (-jac_ee_pos);
    SpMat jac(6, jac_tau.c);
    SetMiddleRows(jac, AX, -jac_tau);
    return jac;
  }
};

// -------------------------------------- height_map.cc / height_map_examples.{h,cc}
struct HeightMap {
  int id;  // HeightMap::TerrainID; 7 = HeightMapFromCSV (include/towr/terrain/height_map_from_csv.h)
  double friction = 0.5;  // ref: height_map.h:136
  // HeightMapFromCSV: grid_(y_cell, x_cell), 0.17 m cells, eps = cell / 50 (height_map_from_csv.h:112-115)
  std::vector<double> grid;
  long grid_rows = 0, grid_cols = 0;
  double res = 0.17, eps = 0.17 / 50;
  explicit HeightMap(int id_) : id(id_) {}

  // ---- id 8: the fork's `Grid` height map (include/towr/terrain/grid_height_map.h:15-60), the terrain fpowr hands
  // the solver (fpowr/src/footstep_plan_server.cc:155): the "elevation" layer of a ROS grid_map sampled with
  // grid_map::InterpolationMethods::INTER_LINEAR into a FLOAT, out of range -> numeric_limits<float>::max()
  // (:33-46); slopes are central differences of those floats over eps = resolution / 6 (:25,48-60); second
  // derivatives are the base class's zeros.  grid_map itself is a third-party dependency that is neither in
  // /root/reference nor in this image and is not version-pinned by the reference (Dockerfile installs the
  // distro's ros-noetic-grid-map): its published algorithm (grid_map_core GridMap::atPosition,
  // atPositionLinearInterpolated, GridMapMath getIndexFromPosition / getPositionFromIndex /
  // checkIfPositionWithinMap) is restated here for a map with start index (0,0):
  //   cell (i,j) centre = map position + length/2 - (index + 1/2) resolution  (x falls with i, y falls with j)
  //   index of a position = trunc((map position + length/2 - position) / resolution)
  //   bilinear over the 2x2 cells around the position, origin cell = the one with the smaller x and y, evaluated
  //   in double and rounded to float; if one of the four cells is outside the map: nearest cell if the position
  //   is inside the map, else std::out_of_range.
  std::vector<float> gm;   // column-major like grid_map's Eigen::MatrixXf: gm[i + j * gm_sx]
  int gm_sx = 0, gm_sy = 0;
  double gm_res = 0, gm_px = 0, gm_py = 0, gm_eps = 0;
  bool GmIn(long i, long j) const { return i >= 0 && j >= 0 && i < gm_sx && j < gm_sy; }
  float GmAt(long i, long j) const { return gm[(size_t)i + (size_t)j * (size_t)gm_sx]; }
  bool GmAtPosition(double x, double y, float& value) const {   // false = std::out_of_range
    const double lx = gm_sx * gm_res, ly = gm_sy * gm_res;
    // getIndexFromPosition: indexVector = (position - length/2 - mapPosition) / resolution, index = -indexVector
    // converted to int (truncation toward zero)
    const double vx = (x - 0.5 * lx - gm_px) / gm_res, vy = (y - 0.5 * ly - gm_py) / gm_res;
    const long i0 = (long)(-vx), j0 = (long)(-vy);
    // checkIfPositionWithinMap: transformed = mapPosition + length/2 - position in [0, length)
    const double tx = gm_px + 0.5 * lx - x, ty = gm_py + 0.5 * ly - y;
    const bool inside = tx >= 0.0 && ty >= 0.0 && tx < lx && ty < ly;
    // atPositionLinearInterpolated
    const double cx0 = gm_px + 0.5 * lx - 0.5 * gm_res - gm_res * (double)i0;   // getPositionFromIndex
    const double cy0 = gm_py + 0.5 * ly - 0.5 * gm_res - gm_res * (double)j0;
    const long ia = x >= cx0 ? i0 : i0 + 1, ja = y >= cy0 ? j0 : j0 + 1;        // origin cell: smaller x, smaller y
    const long ib = ia - 1, jb = ja - 1;
    if (GmIn(ia, ja) && GmIn(ib, jb)) {
      const float f0 = GmAt(ia, ja), f1 = GmAt(ib, ja), f2 = GmAt(ia, jb), f3 = GmAt(ib, jb);
      const double px = gm_px + 0.5 * lx - 0.5 * gm_res - gm_res * (double)ia;
      const double py = gm_py + 0.5 * ly - 0.5 * gm_res - gm_res * (double)ja;
      const double rx = (x - px) / gm_res, ry = (y - py) / gm_res, fx = 1.0 - rx, fy = 1.0 - ry;
      value = (float)(f0 * fx * fy + f1 * rx * fy + f2 * fx * ry + f3 * rx * ry);
      return true;
    }
    if (inside && GmIn(i0, j0)) {   // INTER_NEAREST fallback
      value = GmAt(i0, j0);
      return true;
    }
    return false;
  }
  // ref: grid_height_map.h:29-46 (the value travels as float)
  float GridHeight(double x, double y) const {
    float h;
    if (!GmAtPosition(x, y, h)) h = std::numeric_limits<float>::max();
    return h;
  }
  // ref: grid_height_map.h:48-60: float difference, double quotient
  double GridDeriv(int dim, double x, double y) const {
    const float hp = dim == X ? GridHeight(x + gm_eps, y) : GridHeight(x, y + gm_eps);
    const float hm = dim == X ? GridHeight(x - gm_eps, y) : GridHeight(x, y - gm_eps);
    return (hp - hm) / (2 * gm_eps);
  }

  // static_cast<size_t>(x / res) of the reference: truncation toward zero; a negative quotient <= -1 wraps to
  // a huge size_t on x86-64 (formally undefined) and fails every range check -- restated as a signed cell
  // index that is "invalid" when negative.  (-1, 0) truncates to cell 0 like the reference.
  static long Cell(double v, double res_) { return (long)(v / res_); }
  bool CellValid(long xc, long yc) const { return xc >= 0 && yc >= 0 && xc < grid_cols && yc < grid_rows; }
  double At(long yc, long xc) const { return grid[(size_t)yc * (size_t)grid_cols + (size_t)xc]; }
  // ref: height_map_from_csv.h:29-37
  double CsvHeight(double x, double y) const {
    const long xc = Cell(x, res), yc = Cell(y, res);
    if (!CellValid(xc, yc)) return 0.0;
    return At(yc, xc);
  }
  // ref: height_map_from_csv.h:40-73 (dim X) and :76-109 (dim Y): a step is smeared over eps on its lower side
  double CsvDeriv(int dim, double x, double y) const {
    const long xc = Cell(x, res), yc = Cell(y, res);
    if (!CellValid(xc, yc)) return 0.0;
    const double v = dim == X ? x : y;
    const long c = dim == X ? xc : yc;
    const long nxc = dim == X ? xc + 1 : xc, nyc = dim == X ? yc : yc + 1;   // next cell
    if (CellValid(nxc, nyc)) {
      const double diff_end = At(nyc, nxc) - At(yc, xc);
      const double v_end = (double)(c + 1) * res;
      if ((diff_end > 0) && (v <= v_end) && (v >= v_end - eps)) return diff_end / eps;
    }
    const long pxc = dim == X ? xc - 1 : xc, pyc = dim == X ? yc : yc - 1;   // previous cell
    if (CellValid(pxc, pyc)) {
      const double diff_start = At(yc, xc) - At(pyc, pxc);
      const double v_start = (double)c * res;
      if ((diff_start < 0) && (v >= v_start) && (v <= v_start + eps)) return diff_start / eps;
    }
    return 0.0;
  }

  double GetHeight(double x, double y) const {
    switch (id) {
      case 7: return CsvHeight(x, y);
      case 8: return GridHeight(x, y);
      case 0: return 0.0;  // FlatGround(0.0)
      case 1: {            // Block, ref: height_map_examples.cc:40-53, .h:63-68
        const double block_start = 0.7, length = 3.5, height = 0.5, eps = 0.03, slope = height / eps;
        double h = 0.0;
        if (block_start <= x && x <= block_start + eps) h = slope * (x - block_start);
        if (block_start + eps <= x && x <= block_start + length) h = height;
        return h;
      }
      case 2: {  // Stairs, ref: height_map_examples.cc:69-84, .h:79-83
        const double first_step_start = 1.0, first_step_width = 0.4, h1 = 0.2, h2 = 0.4, width_top = 1.0;
        double h = 0.0;
        if (x >= first_step_start) h = h1;
        if (x >= first_step_start + first_step_width) h = h2;
        if (x >= first_step_start + first_step_width + width_top) h = 0.0;
        return h;
      }
      case 3: {  // Gap, ref: height_map_examples.cc:88-98
        GapC c;
        double h = 0.0;
        if (c.gap_start <= x && x <= c.gap_end_x) h = c.a * x * x + c.b * x + c.c;
        return h;
      }
      case 4: {  // Slope, ref: height_map_examples.cc:124-141, .h:123-130
        SlopeC s;
        double z = 0.0;
        if (x >= s.slope_start) z = s.slope * (x - s.slope_start);
        if (x >= s.x_down_start) z = s.height_center - s.slope * (x - s.x_down_start);
        if (x >= s.x_flat_start) z = 0.0;
        return z;
      }
      case 5: {  // Chimney, ref: height_map_examples.cc:161-170, .h:142-147
        const double x_start = 1.0, length = 1.5, y_start = 0.5, slope = 3.0, x_end = x_start + length;
        double z = 0.0;
        if (x_start <= x && x <= x_end) z = slope * (y - y_start);
        return z;
      }
      case 6: {  // ChimneyLR, ref: height_map_examples.cc:185-197, .h:159-165
        const double x_start = 0.5, length = 1.0, y_start = 0.5, slope = 2, x_end1 = x_start + length, x_end2 = x_start + 2 * length;
        double z = 0.0;
        if (x_start <= x && x <= x_end1) z = slope * (y - y_start);
        if (x_end1 <= x && x <= x_end2) z = -slope * (y + y_start);
        return z;
      }
    }
    throw std::runtime_error("unknown terrain");
  }
  struct GapC {  // ref: height_map_examples.h:96-111
    const double gap_start = 1.0, w = 0.5, h = 1.5;
    const double dx = w / 2.0, xc = gap_start + dx, gap_end_x = gap_start + w;
    const double a = (4 * h) / (w * w), b = -(8 * h * xc) / (w * w), c = -(h * (w - 2 * xc) * (w + 2 * xc)) / (w * w);
  };
  struct SlopeC {
    const double slope_start = 1.0, up_length = 1.0, down_length = 1.0, height_center = 0.7;
    const double x_down_start = slope_start + up_length, x_flat_start = x_down_start + down_length;
    const double slope = height_center / up_length;
  };
  double DerivX(double x, double y) const {
    switch (id) {
      case 7: return CsvDeriv(X, x, y);
      case 8: return GridDeriv(X, x, y);
      case 1: {  // ref: height_map_examples.cc:55-65
        const double block_start = 0.7, height = 0.5, eps = 0.03, slope = height / eps;
        return (block_start <= x && x <= block_start + eps) ? slope : 0.0;
      }
      case 3: {  // ref: height_map_examples.cc:100-109
        GapC c;
        return (c.gap_start <= x && x <= c.gap_end_x) ? 2 * c.a * x + c.b : 0.0;
      }
      case 4: {  // ref: height_map_examples.cc:143-157
        SlopeC s;
        double d = 0.0;
        if (x >= s.slope_start) d = s.slope;
        if (x >= s.x_down_start) d = -s.slope;
        if (x >= s.x_flat_start) d = 0.0;
        return d;
      }
      default: return 0.0;  // Flat, Stairs (quirk: only GetHeight overridden), Chimneys
    }
  }
  double DerivY(double x, double y) const {
    switch (id) {
      case 7: return CsvDeriv(Y, x, y);
      case 8: return GridDeriv(Y, x, y);
      case 5: {  // ref: height_map_examples.cc:172-181
        const double x_start = 1.0, length = 1.5, slope = 3.0, x_end = x_start + length;
        return (x_start <= x && x <= x_end) ? slope : 0.0;
      }
      case 6: {  // ref: height_map_examples.cc:199-211
        const double x_start = 0.5, length = 1.0, slope = 2, x_end1 = x_start + length, x_end2 = x_start + 2 * length;
        double d = 0.0;
        if (x_start <= x && x <= x_end1) d = slope;
        if (x_end1 <= x && x <= x_end2) d = -slope;
        return d;
      }
      default: return 0.0;
    }
  }
  double DerivXX(double x, double y) const {  // ref: height_map_examples.cc:111-120
    if (id == 3) {
      GapC c;
      return (c.gap_start <= x && x <= c.gap_end_x) ? 2 * c.a : 0.0;
    }
    return 0.0;
  }
  // ref: height_map.cc:52-60
  double GetDerivativeOfHeightWrt(int dim, double x, double y) const { return dim == X ? DerivX(x, y) : DerivY(x, y); }
  // ref: height_map.cc:150-163 (XY, YX, YY never overridden -> 0)
  double GetSecondDerivativeOfHeightWrt(int d1, int d2, double x, double y) const {
    if (d1 == X && d2 == X) return DerivXX(x, y);
    return 0.0;
  }
  // deriv < 0: basis requested, else derivative of the (non-normalised) basis w.r.t. dim `deriv`
  // ref: height_map.cc:93-139
  V3 GetBasis(int basis, double x, double y, int deriv = -1) const {
    bool req = deriv < 0;
    V3 b;
    switch (basis) {
      case 0:  // Normal
        for (int dim : {X, Y})
          b(dim) = req ? -GetDerivativeOfHeightWrt(dim, x, y) : -GetSecondDerivativeOfHeightWrt(dim, deriv, x, y);
        b(Z) = req ? 1.0 : 0.0;
        break;
      case 1:  // Tangent1
        b(X) = req ? 1.0 : 0.0;
        b(Y) = 0.0;
        b(Z) = req ? GetDerivativeOfHeightWrt(X, x, y) : GetSecondDerivativeOfHeightWrt(X, deriv, x, y);
        break;
      case 2:  // Tangent2
        b(X) = 0.0;
        b(Y) = req ? 1.0 : 0.0;
        b(Z) = req ? GetDerivativeOfHeightWrt(Y, x, y) : GetSecondDerivativeOfHeightWrt(Y, deriv, x, y);
        break;
    }
    return b;
  }
  static V3 normalized(const V3& v) { return v / std::sqrt(dot(v, v)); }
  // ref: height_map.cc:62-66
  V3 GetNormalizedBasis(int basis, double x, double y) const { return normalized(GetBasis(basis, x, y)); }
  // ref: height_map.cc:80-91,141-148 (component-wise product -- quirk 4 -- reproduced as is)
  V3 GetDerivativeOfNormalizedBasisWrt(int basis, int dim, double x, double y) const {
    V3 dv = GetBasis(basis, x, y, dim);
    V3 v = GetBasis(basis, x, y);
    double sq = dot(v, v), nrm = std::sqrt(sq);
    V3 unit;
    unit(dim) = 1.0;
    V3 outer = (1 / sq) * (nrm * unit - v(dim) * normalized(v));
    return V3(outer(0) * dv(0), outer(1) * dv(1), outer(2) * dv(2));
  }
};

// ------------------------------------------------------------ constraint sets
struct Vars {  // the ifopt "variables" composite
  std::vector<NodesVars*> sets;
  NodesVars* Get(const std::string& name) const {
    for (auto* s : sets)
      if (s->name == name) return s;
    throw std::runtime_error("variable set " + name + " not found");
  }
};

struct ConSet {
  std::string name;
  int rows = 0;
  virtual ~ConSet() = default;
  virtual void GetValues(double* g) const = 0;
  virtual void GetBounds(Bound* b) const = 0;
  virtual void FillJacobianBlock(const std::string& var_set, SpMat& jac) const = 0;
};

// ref: time_discretization_constraint.cc:37-50
static std::vector<double> MakeTimeGrid(double T, double dt) {
  double t = 0.0;
  std::vector<double> dts = {t};
  for (int i = 0; i < floor(T / dt); ++i) {
    t += dt;
    dts.push_back(t);
  }
  dts.push_back(T);
  return dts;
}

struct Splines {
  NodeSpline* base_linear;
  NodeSpline* base_angular;
  std::vector<NodeSpline*> ee_motion, ee_force;
};

// ref: dynamic_constraint.cc:37-137
struct DynamicConstraint : ConSet {
  mutable SRBD* model;
  std::vector<double> dts;
  NodeSpline* base_linear;
  EulerConverter base_angular;
  std::vector<NodeSpline*> ee_forces, ee_motion;

  DynamicConstraint(SRBD* m, double T, double dt, const Splines& s) : model(m) {
    name = "dynamic";
    dts = MakeTimeGrid(T, dt);
    base_linear = s.base_linear;
    base_angular.euler = s.base_angular;
    ee_forces = s.ee_force;
    ee_motion = s.ee_motion;
    rows = (int)dts.size() * 6;
  }
  void UpdateModel(double t) const {  // ref: dynamic_constraint.cc:119-137
    StateVal com = base_linear->GetPoint(t);
    SpMat R = base_angular.GetRotationMatrixBaseToWorld(t);
    M3 w_R_b;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) w_R_b(i, j) = R.coeff(i, j);
    model->omega = base_angular.GetAngularVelocityInWorld(t);
    model->omega_dot = base_angular.GetAngularAccelerationInWorld(t);
    for (size_t ee = 0; ee < ee_forces.size(); ++ee) {
      model->ee_force[ee] = ee_forces[ee]->GetPoint(t).p;
      model->ee_pos[ee] = ee_motion[ee]->GetPoint(t).p;
    }
    model->com_pos = com.p;
    model->com_acc = com.a;
    model->w_R_b = w_R_b;
  }
  void GetValues(double* g) const override {  // ref: time_discretization_constraint.cc:65-75, dynamic_constraint.cc:59-64
    int k = 0;
    for (double t : dts) {
      UpdateModel(t);
      model->GetDynamicViolation(g + 6 * k);
      k++;
    }
  }
  void GetBounds(Bound* b) const override {  // ref: dynamic_constraint.cc:66-71
    for (int i = 0; i < rows; ++i) b[i] = {0.0, 0.0};
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    int k = 0;
    for (double t : dts) UpdateJacobianAtInstance(t, k++, var_set, jac);
  }
  // ref: dynamic_constraint.cc:73-117
  void UpdateJacobianAtInstance(double t, int k, const std::string& var_set, SpMat& jac) const {
    UpdateModel(t);
    int n = jac.c;
    SpMat jac_model(6, n);
    if (var_set == "base-lin") {
      SpMat jp = base_linear->GetJacobianWrtNodes(t, kPos);
      SpMat ja = base_linear->GetJacobianWrtNodes(t, kAcc);
      jac_model = model->GetJacobianWrtBaseLin(jp, ja);
    }
    if (var_set == "base-ang") jac_model = model->GetJacobianWrtBaseAng(base_angular, t);
    for (int ee = 0; ee < (int)ee_forces.size(); ++ee) {
      if (var_set == "ee-force_" + std::to_string(ee)) {
        SpMat jf = ee_forces[ee]->GetJacobianWrtNodes(t, kPos);
        jac_model = model->GetJacobianWrtForce(jf, ee);
      }
      if (var_set == "ee-motion_" + std::to_string(ee)) {
        SpMat jp = ee_motion[ee]->GetJacobianWrtNodes(t, kPos);
        jac_model = model->GetJacobianWrtEEPos(jp, ee);
      }
      if (var_set == "ee-schedule" + std::to_string(ee)) {  // ref: dynamic_constraint.cc:107-113
        SpMat jac_f_dT = ee_forces[ee]->GetJacobianOfPosWrtDurations(t);
        jac_model = jac_model + model->GetJacobianWrtForce(jac_f_dT, ee);
        SpMat jac_x_dT = ee_motion[ee]->GetJacobianOfPosWrtDurations(t);
        jac_model = jac_model + model->GetJacobianWrtEEPos(jac_x_dT, ee);
      }
    }
    for (int i = 0; i < 6; ++i) jac.rows[6 * k + i] = jac_model.rows[i];
  }
};

// ref: range_of_motion_constraint.cc:35-109
struct RangeOfMotionConstraint : ConSet {
  std::vector<double> dts;
  NodeSpline* base_linear;
  EulerConverter base_angular;
  NodeSpline* ee_motion;
  V3 max_dev, nominal;
  int ee;

  RangeOfMotionConstraint(const RobotConsts& r, double T, double dt, int ee_, const Splines& s) : ee(ee_) {
    name = "rangeofmotion-" + std::to_string(ee);
    dts = MakeTimeGrid(T, dt);
    base_linear = s.base_linear;
    base_angular.euler = s.base_angular;
    ee_motion = s.ee_motion.at(ee);
    max_dev = r.max_dev;
    nominal = r.nominal[ee];
    rows = (int)dts.size() * 3;
  }
  static SpMat Transpose(const SpMat& A) {
    SpMat t(A.c, A.r);
    for (int i = 0; i < A.r; ++i)
      for (auto& e : A.rows[i].e) t.coeffRef(e.first, i) = e.second;
    return t;
  }
  void GetValues(double* g) const override {  // ref: range_of_motion_constraint.cc:58-69
    int k = 0;
    for (double t : dts) {
      V3 base_W = base_linear->GetPoint(t).p;
      V3 pos_ee_W = ee_motion->GetPoint(t).p;
      SpMat b_R_w = Transpose(base_angular.GetRotationMatrixBaseToWorld(t));
      V3 v = b_R_w * (pos_ee_W - base_W);
      for (int d = 0; d < 3; ++d) g[3 * k + d] = v(d);
      k++;
    }
  }
  void GetBounds(Bound* b) const override {  // ref: range_of_motion_constraint.cc:71-81
    for (size_t k = 0; k < dts.size(); ++k)
      for (int dim = 0; dim < 3; ++dim) {
        Bound bd;  // ifopt::Bounds() = (0,0)
        bd.lo += nominal(dim); bd.up += nominal(dim);
        bd.up += max_dev(dim);
        bd.lo -= max_dev(dim);
        b[3 * k + dim] = bd;
      }
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    int k = 0;
    for (double t : dts) {  // ref: range_of_motion_constraint.cc:83-109
      SpMat b_R_w = Transpose(base_angular.GetRotationMatrixBaseToWorld(t));
      int row_start = 3 * k;
      auto set_rows = [&](const SpMat& m) { for (int i = 0; i < 3; ++i) jac.rows[row_start + i] = m.rows[i]; };
      if (var_set == "base-lin") set_rows((-1.0 * b_R_w) * base_linear->GetJacobianWrtNodes(t, kPos));
      if (var_set == "base-ang") {
        V3 base_W = base_linear->GetPoint(t).p;
        V3 ee_pos_W = ee_motion->GetPoint(t).p;
        V3 r_W = ee_pos_W - base_W;
        set_rows(base_angular.DerivOfRotVecMult(t, r_W, true));
      }
      if (var_set == "ee-motion_" + std::to_string(ee)) set_rows(b_R_w * ee_motion->GetJacobianWrtNodes(t, kPos));
      if (var_set == "ee-schedule" + std::to_string(ee)) set_rows(b_R_w * ee_motion->GetJacobianOfPosWrtDurations(t));
      k++;
    }
  }
};

// ref: force_constraint.cc:37-171
struct ForceConstraint : ConSet {
  const HeightMap* terrain;
  double fn_max, mu;
  int ee;
  NodesVars *ee_force, *ee_motion;
  std::vector<int> pure_stance_force_node_ids;

  ForceConstraint(const HeightMap* t, double force_limit, int ee_, const Vars& x) : terrain(t), fn_max(force_limit), mu(t->friction), ee(ee_) {
    name = "force-ee-force_" + std::to_string(ee);
    ee_force = x.Get("ee-force_" + std::to_string(ee));
    ee_motion = x.Get("ee-motion_" + std::to_string(ee));
    pure_stance_force_node_ids = ee_force->NonConstantNodes();
    rows = (int)pure_stance_force_node_ids.size() * 5;
  }
  void GetValues(double* g) const override {  // ref: force_constraint.cc:62-89
    int row = 0;
    for (int f_node_id : pure_stance_force_node_ids) {
      int phase = ee_force->GetPhase(f_node_id);
      V3 p = ee_motion->nodes.at(ee_motion->NodeIdAtStartOfPhase(phase)).p;
      V3 n = terrain->GetNormalizedBasis(0, p(X), p(Y));
      V3 f = ee_force->nodes.at(f_node_id).p;
      g[row++] = dot(f, n);
      V3 t1 = terrain->GetNormalizedBasis(1, p(X), p(Y));
      g[row++] = dot(f, t1 - mu * n);
      g[row++] = dot(f, t1 + mu * n);
      V3 t2 = terrain->GetNormalizedBasis(2, p(X), p(Y));
      g[row++] = dot(f, t2 - mu * n);
      g[row++] = dot(f, t2 + mu * n);
    }
  }
  void GetBounds(Bound* b) const override {  // ref: force_constraint.cc:91-105
    int i = 0;
    for (size_t k = 0; k < pure_stance_force_node_ids.size(); ++k) {
      b[i++] = {0.0, fn_max};
      b[i++] = {-kInf, 0.0};
      b[i++] = {0.0, kInf};
      b[i++] = {-kInf, 0.0};
      b[i++] = {0.0, kInf};
    }
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    if (var_set == ee_force->name) {  // ref: force_constraint.cc:111-135
      int row = 0;
      for (int f_node_id : pure_stance_force_node_ids) {
        int phase = ee_force->GetPhase(f_node_id);
        V3 p = ee_motion->nodes.at(ee_motion->NodeIdAtStartOfPhase(phase)).p;
        V3 n = terrain->GetNormalizedBasis(0, p(X), p(Y));
        V3 t1 = terrain->GetNormalizedBasis(1, p(X), p(Y));
        V3 t2 = terrain->GetNormalizedBasis(2, p(X), p(Y));
        for (int dim : {X, Y, Z}) {
          int idx = ee_force->GetOptIndex(NVI{f_node_id, kPos, dim});
          int rr = row;
          jac.coeffRef(rr++, idx) = n(dim);
          jac.coeffRef(rr++, idx) = t1(dim) - mu * n(dim);
          jac.coeffRef(rr++, idx) = t1(dim) + mu * n(dim);
          jac.coeffRef(rr++, idx) = t2(dim) - mu * n(dim);
          jac.coeffRef(rr++, idx) = t2(dim) + mu * n(dim);
        }
        row += 5;
      }
    }
    if (var_set == ee_motion->name) {  // ref: force_constraint.cc:138-170
      int row = 0;
      for (int f_node_id : pure_stance_force_node_ids) {
        int phase = ee_force->GetPhase(f_node_id);
        int ee_node_id = ee_motion->NodeIdAtStartOfPhase(phase);
        V3 p = ee_motion->nodes.at(ee_node_id).p;
        V3 f = ee_force->nodes.at(f_node_id).p;
        for (int dim : {X, Y}) {
          V3 dn = terrain->GetDerivativeOfNormalizedBasisWrt(0, dim, p(X), p(Y));
          V3 dt1 = terrain->GetDerivativeOfNormalizedBasisWrt(1, dim, p(X), p(Y));
          V3 dt2 = terrain->GetDerivativeOfNormalizedBasisWrt(2, dim, p(X), p(Y));
          int idx = ee_motion->GetOptIndex(NVI{ee_node_id, kPos, dim});
          int rr = row;
          jac.coeffRef(rr++, idx) = dot(f, dn);
          jac.coeffRef(rr++, idx) = dot(f, dt1 - mu * dn);
          jac.coeffRef(rr++, idx) = dot(f, dt1 + mu * dn);
          jac.coeffRef(rr++, idx) = dot(f, dt2 - mu * dn);
          jac.coeffRef(rr++, idx) = dot(f, dt2 + mu * dn);
        }
        row += 5;
      }
    }
  }
};

// ref: terrain_constraint.cc:36-108
struct TerrainConstraint : ConSet {
  const HeightMap* terrain;
  NodesVars* ee_motion;
  std::vector<int> node_ids;
  TerrainConstraint(const HeightMap* t, const std::string& ee_motion_id, const Vars& x) : terrain(t) {
    name = "terrain-" + ee_motion_id;
    ee_motion = x.Get(ee_motion_id);
    for (int id = 1; id < (int)ee_motion->nodes.size(); ++id) node_ids.push_back(id);
    rows = (int)node_ids.size();
  }
  void GetValues(double* g) const override {
    int row = 0;
    for (int id : node_ids) {
      V3 p = ee_motion->nodes.at(id).p;
      g[row++] = p(Z) - terrain->GetHeight(p(X), p(Y));
    }
  }
  void GetBounds(Bound* b) const override {
    int row = 0;
    for (int id : node_ids) {
      b[row] = ee_motion->IsConstantNode(id) ? Bound{0.0, 0.0} : Bound{0.0, 1e20};
      row++;
    }
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    if (var_set != ee_motion->name) return;
    int row = 0;
    for (int id : node_ids) {
      int idx = ee_motion->GetOptIndex(NVI{id, kPos, Z});
      jac.coeffRef(row, idx) = 1.0;
      V3 p = ee_motion->nodes.at(id).p;
      for (int dim : {X, Y}) {
        int i2 = ee_motion->GetOptIndex(NVI{id, kPos, dim});
        jac.coeffRef(row, i2) = -terrain->GetDerivativeOfHeightWrt(dim, p(X), p(Y));
      }
      row++;
    }
  }
};

// ref: spline_acc_constraint.cc:34-88
struct SplineAccConstraint : ConSet {
  const NodeSpline* spline;
  std::string node_variables_id;
  int n_dim = 3, n_junctions = 0;
  std::vector<double> T;
  SplineAccConstraint(const NodeSpline* sp, const std::string& node_variable_name) : spline(sp) {
    name = "splineacc-" + node_variable_name;
    node_variables_id = node_variable_name;
    n_junctions = (int)spline->polys.size() - 1;
    T = spline->GetPolyDurations();
    rows = n_dim * n_junctions;
  }
  void GetValues(double* g) const override {
    for (int j = 0; j < n_junctions; ++j) {
      int p_prev = j;
      V3 acc_prev = spline->GetPoint(p_prev, T.at(p_prev)).a;
      int p_next = j + 1;
      V3 acc_next = spline->GetPoint(p_next, 0.0).a;
      V3 d = acc_prev - acc_next;
      for (int i = 0; i < n_dim; ++i) g[j * n_dim + i] = d(i);
    }
  }
  void GetBounds(Bound* b) const override {
    for (int i = 0; i < rows; ++i) b[i] = Bound{0.0, 0.0};
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    if (var_set != node_variables_id) return;
    for (int j = 0; j < n_junctions; ++j) {
      int p_prev = j;
      SpMat acc_prev = spline->GetJacobianWrtNodes(p_prev, T.at(p_prev), kAcc);
      int p_next = j + 1;
      SpMat acc_next = spline->GetJacobianWrtNodes(p_next, 0.0, kAcc);
      SpMat diff = acc_prev - acc_next;  // sparse difference: union pattern, cancelled entries stay
      for (int i = 0; i < n_dim; ++i) jac.rows[j * n_dim + i] = diff.rows[i];
    }
  }
};

// ref: swing_constraint.cc:35-121, swing_constraint.h:68
struct SwingConstraint : ConSet {
  NodesVars* ee_motion;
  std::vector<int> pure_swing_node_ids;
  double t_swing_avg = 0.3;
  SwingConstraint(const std::string& ee_motion_id, const Vars& x) {
    name = "swing-" + ee_motion_id;
    ee_motion = x.Get(ee_motion_id);
    pure_swing_node_ids = ee_motion->NonConstantNodes();
    // "assumes ... starting and ending in stance" (swing_constraint.cc:66): the reference would index
    // nodes.at(-1) / past the end otherwise; refuse such schedules at construction.
    for (int id : pure_swing_node_ids)
      if (id == 0 || id == (int)ee_motion->nodes.size() - 1) throw std::runtime_error("swing: schedule must start and end in stance");
    rows = (int)pure_swing_node_ids.size() * 2 * 2;  // Node::n_derivatives * k2D
  }
  void GetValues(double* g) const override {
    int row = 0;
    const auto& nodes = ee_motion->nodes;
    for (int node_id : pure_swing_node_ids) {
      const NodeVal& curr = nodes.at(node_id);
      const NodeVal& prev = nodes.at(node_id - 1);
      const NodeVal& next = nodes.at(node_id + 1);
      for (int dim : {X, Y}) {
        double distance = next.p(dim) - prev.p(dim);
        double center = prev.p(dim) + 0.5 * distance;
        double des_vel_center = distance / t_swing_avg;
        g[row++] = curr.p(dim) - center;
        g[row++] = curr.v(dim) - des_vel_center;
      }
    }
  }
  void GetBounds(Bound* b) const override {
    for (int i = 0; i < rows; ++i) b[i] = Bound{0.0, 0.0};
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    if (var_set != ee_motion->name) return;
    int row = 0;
    auto idx = [&](int id, int deriv, int dim) {
      int i = ee_motion->GetOptIndex(NVI{id, deriv, dim});
      if (i < 0) throw std::runtime_error("swing: node value is not an optimisation variable");
      return i;
    };
    for (int node_id : pure_swing_node_ids) {
      for (int dim : {X, Y}) {
        jac.coeffRef(row, idx(node_id, kPos, dim)) = 1.0;
        jac.coeffRef(row, idx(node_id + 1, kPos, dim)) = -0.5;
        jac.coeffRef(row, idx(node_id - 1, kPos, dim)) = -0.5;
        row++;
        jac.coeffRef(row, idx(node_id, kVel, dim)) = 1.0;
        jac.coeffRef(row, idx(node_id + 1, kPos, dim)) = -1.0 / t_swing_avg;
        jac.coeffRef(row, idx(node_id - 1, kPos, dim)) = +1.0 / t_swing_avg;
        row++;
      }
    }
  }
};

// ref: base_motion_constraint.cc:38-99.  z_init is base_linear_->GetPoint(0.0).p().z() at construction, i.e. the
// initial base height of the formulation (the spline then holds the initial guess); passed in explicitly.
struct BaseMotionConstraint : ConSet {
  std::vector<double> dts;
  NodeSpline* base_linear;
  NodeSpline* base_angular;
  Bound node_bounds[6];
  BaseMotionConstraint(double T, double dt, const Splines& s, double z_init) {
    name = "baseMotion";
    dts = MakeTimeGrid(T, dt);
    base_linear = s.base_linear;
    base_angular = s.base_angular;
    double dev_rad = 0.05;
    node_bounds[AX] = Bound{-dev_rad, dev_rad};
    node_bounds[AY] = Bound{-dev_rad, dev_rad};
    node_bounds[AZ] = Bound{-kInf, kInf};
    node_bounds[LX] = Bound{-kInf, kInf};
    node_bounds[LY] = Bound{-kInf, kInf};
    node_bounds[LZ] = Bound{z_init - 0.02, z_init + 0.1};
    rows = (int)dts.size() * 6;
  }
  void GetValues(double* g) const override {
    int k = 0;
    for (double t : dts) {
      V3 lin = base_linear->GetPoint(t).p, ang = base_angular->GetPoint(t).p;
      for (int i = 0; i < 3; ++i) {
        g[6 * k + LX + i] = lin(i);
        g[6 * k + AX + i] = ang(i);
      }
      k++;
    }
  }
  void GetBounds(Bound* b) const override {
    for (size_t k = 0; k < dts.size(); ++k)
      for (int dim = 0; dim < 6; ++dim) b[6 * k + dim] = node_bounds[dim];
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    int k = 0;
    for (double t : dts) {
      if (var_set == "base-ang") {
        SpMat j = base_angular->GetJacobianWrtNodes(t, kPos);
        for (int i = 0; i < 3; ++i) jac.rows[6 * k + AX + i] = j.rows[i];
      }
      if (var_set == "base-lin") {
        SpMat j = base_linear->GetJacobianWrtNodes(t, kPos);
        for (int i = 0; i < 3; ++i) jac.rows[6 * k + LX + i] = j.rows[i];
      }
      k++;
    }
  }
};

// ref: total_duration_constraint.cc:36-72
struct TotalDurationConstraint : ConSet {
  double T_total;
  int ee;
  const PhaseDurations* phase_durations;
  TotalDurationConstraint(double T, int ee_, const PhaseDurations* pd) : T_total(T), ee(ee_), phase_durations(pd) {
    name = "totalduration-" + std::to_string(ee);
    rows = 1;
  }
  void GetValues(double* g) const override {
    double sum = 0.0;  // phase_durations_->GetValues().sum(): excludes the last duration
    for (int i = 0; i < phase_durations->rows(); ++i) sum += phase_durations->durations.at(i);
    g[0] = sum;
  }
  void GetBounds(Bound* b) const override {
    double min_duration_last_phase = 0.2;
    b[0] = Bound{0.1, T_total - min_duration_last_phase};
  }
  void FillJacobianBlock(const std::string& var_set, SpMat& jac) const override {
    if (var_set == phase_durations->name)
      for (int col = 0; col < phase_durations->rows(); ++col) jac.coeffRef(0, col) = 1.0;
  }
};

// ------------------------------------------- parameters.cc / nlp_formulation.cc
// ref: parameters.cc:82-98
static std::vector<double> GetBasePolyDurations(double T, double dt) {
  std::vector<double> v;
  double t_left = T;
  double eps = 1e-10;
  while (t_left > eps) {
    double duration = t_left > dt ? dt : t_left;
    v.push_back(duration);
    t_left -= dt;
  }
  return v;
}

}  // namespace orc

using namespace orc;

struct orc_problem {
  RobotConsts robot;
  std::unique_ptr<HeightMap> terrain;
  std::unique_ptr<SRBD> model;
  int n_ee = 0;
  std::vector<std::vector<double>> phase_durations;
  std::vector<int> contact_at_start;
  double T = 0;
  double dt_dyn, dt_rom, dur_base;
  int polys_swing, polys_stance;

  std::vector<std::unique_ptr<NodesVars>> var_sets;  // order: nlp_formulation.cc:68-82
  std::vector<std::unique_ptr<PhaseDurations>> schedules;  // nlp_formulation.cc:183-198 (always built)
  bool optimize_timings = false;                           // Parameters::IsOptimizeTimings, parameters.cc:128-135
  std::vector<std::unique_ptr<NodeSpline>> splines;
  Vars vars;
  Splines sp;
  std::vector<std::unique_ptr<ConSet>> cons;  // order: parameters.cc:55-60 restricted to the hot path

  int n_vars() const {
    int n = 0;
    for (auto& v : var_sets) n += v->rows();
    if (optimize_timings)
      for (auto& s : schedules) n += s->rows();
    return n;
  }
  // the ifopt variable sets in order (nlp_formulation.cc:63-93): node sets, then ee-schedule_e if optimised
  int n_sets() const { return (int)var_sets.size() + (optimize_timings ? (int)schedules.size() : 0); }
  const std::string& set_name(int i) const {
    return i < (int)var_sets.size() ? var_sets[i]->name : schedules.at(i - var_sets.size())->name;
  }
  int set_rows(int i) const { return i < (int)var_sets.size() ? var_sets[i]->rows() : schedules.at(i - var_sets.size())->rows(); }
  int n_rows() const {
    int m = 0;
    for (auto& c : cons) m += c->rows;
    return m;
  }
  // ifopt Composite::SetVariables
  void SetVariables(const double* x) {
    int off = 0;
    for (auto& v : var_sets) {
      v->SetVariables(x + off);
      off += v->rows();
    }
    if (optimize_timings)
      for (auto& s : schedules) {
        s->SetVariables(x + off);
        off += s->rows();
      }
  }
};

extern "C" {

orc_problem* orc_create(int robot, int terrain, int n_ee, const int* n_phases, const double* phase_durations,
                        const int* in_contact_at_start, double dt_dynamic, double dt_rom, double duration_base_poly,
                        int polys_per_swing, int polys_per_stance_force, double force_limit, int constraint_sets,
                        double dt_base_motion, double base_z_init, const double* grid, int grid_rows, int grid_cols) {
  try {
    auto* P = new orc_problem();
    P->robot = MakeRobot(robot);
    if (P->robot.n_ee != n_ee) throw std::runtime_error("n_ee mismatch with robot");
    P->terrain.reset(new HeightMap(terrain));
    if (terrain == 7) {  // HeightMapFromCSV: the grid the reference reads with rapidcsv (height_map_from_csv.h:19-27)
      if (!grid || grid_rows < 1 || grid_cols < 1) throw std::runtime_error("terrain 7 needs a grid");
      P->terrain->grid.assign(grid, grid + (size_t)grid_rows * grid_cols);
      P->terrain->grid_rows = grid_rows;
      P->terrain->grid_cols = grid_cols;
    }
    P->model.reset(new SRBD(P->robot));
    P->n_ee = n_ee;
    const double* pd = phase_durations;
    for (int ee = 0; ee < n_ee; ++ee) {
      P->phase_durations.emplace_back(pd, pd + n_phases[ee]);
      pd += n_phases[ee];
      P->contact_at_start.push_back(in_contact_at_start[ee]);
    }
    // ref: parameters.cc:112-126
    P->T = std::accumulate(P->phase_durations[0].begin(), P->phase_durations[0].end(), 0.0);
    P->dt_dyn = dt_dynamic; P->dt_rom = dt_rom; P->dur_base = duration_base_poly;
    P->polys_swing = polys_per_swing; P->polys_stance = polys_per_stance_force;

    // variables, ref: nlp_formulation.cc:63-181
    auto base_durations = GetBasePolyDurations(P->T, P->dur_base);
    int n_nodes = (int)base_durations.size() + 1;
    P->var_sets.emplace_back(new NodesVars(MakeNodesAll(n_nodes, "base-lin")));
    P->var_sets.emplace_back(new NodesVars(MakeNodesAll(n_nodes, "base-ang")));
    for (int ee = 0; ee < n_ee; ++ee)
      P->var_sets.emplace_back(new NodesVars(MakeNodesEEMotion((int)P->phase_durations[ee].size(), P->contact_at_start[ee] != 0,
                                                               "ee-motion_" + std::to_string(ee), polys_per_swing)));
    for (int ee = 0; ee < n_ee; ++ee)
      P->var_sets.emplace_back(new NodesVars(MakeNodesEEForce((int)P->phase_durations[ee].size(), P->contact_at_start[ee] != 0,
                                                              "ee-force_" + std::to_string(ee), polys_per_stance_force)));
    for (auto& v : P->var_sets) P->vars.sets.push_back(v.get());

    // contact schedule, ref: nlp_formulation.cc:183-198, parameters.cc:52 (bound_phase_duration_)
    P->optimize_timings = (constraint_sets & ORC_SET_TOTAL_TIME) != 0;
    for (int ee = 0; ee < n_ee; ++ee)
      P->schedules.emplace_back(new PhaseDurations(ee, P->phase_durations[ee], P->contact_at_start[ee] != 0, 0.2, 1.0));
    // splines, ref: spline_holder.cc:35-61 (fixed timings -> NodeSpline, optimised -> PhaseSpline)
    int ee_of_spline = -1;
    auto add_spline = [&](NodesVars* nv, const std::vector<double>& d) {
      if (P->optimize_timings && ee_of_spline >= 0)
        P->splines.emplace_back(new PhaseSpline(nv, P->schedules[ee_of_spline].get()));
      else
        P->splines.emplace_back(new NodeSpline(nv, d));
      nv->observer = P->splines.back().get();
      return P->splines.back().get();
    };
    P->sp.base_linear = add_spline(P->var_sets[0].get(), base_durations);
    P->sp.base_angular = add_spline(P->var_sets[1].get(), base_durations);
    for (int ee = 0; ee < n_ee; ++ee) {
      NodesVars* m = P->var_sets[2 + ee].get();
      ee_of_spline = ee;
      P->sp.ee_motion.push_back(add_spline(m, m->PhaseToPolyDurations(P->phase_durations[ee])));
    }
    for (int ee = 0; ee < n_ee; ++ee) {
      NodesVars* f = P->var_sets[2 + n_ee + ee].get();
      ee_of_spline = ee;
      P->sp.ee_force.push_back(add_spline(f, f->PhaseToPolyDurations(P->phase_durations[ee])));
    }

    // constraints in reference order {Terrain, Dynamic, BaseAcc, EndeffectorRom, Force, Swing}
    // (parameters.cc:55-60, nlp_formulation.cc:183-331); constraint_sets selects which are built.
    if (constraint_sets & ORC_SET_TERRAIN)
      for (int ee = 0; ee < n_ee; ++ee)
        P->cons.emplace_back(new TerrainConstraint(P->terrain.get(), "ee-motion_" + std::to_string(ee), P->vars));
    if (constraint_sets & ORC_SET_DYNAMIC)
      P->cons.emplace_back(new DynamicConstraint(P->model.get(), P->T, dt_dynamic, P->sp));
    if (constraint_sets & ORC_SET_BASE_ACC) {
      P->cons.emplace_back(new SplineAccConstraint(P->sp.base_linear, "base-lin"));
      P->cons.emplace_back(new SplineAccConstraint(P->sp.base_angular, "base-ang"));
    }
    if (constraint_sets & ORC_SET_ROM)
      for (int ee = 0; ee < n_ee; ++ee)
        P->cons.emplace_back(new RangeOfMotionConstraint(P->robot, P->T, dt_rom, ee, P->sp));
    if (constraint_sets & ORC_SET_FORCE)
      for (int ee = 0; ee < n_ee; ++ee)
        P->cons.emplace_back(new ForceConstraint(P->terrain.get(), force_limit, ee, P->vars));
    if (constraint_sets & ORC_SET_SWING)
      for (int ee = 0; ee < n_ee; ++ee)
        P->cons.emplace_back(new SwingConstraint("ee-motion_" + std::to_string(ee), P->vars));
    if (constraint_sets & ORC_SET_BASE_ROM)  // not in the default list: a caller that pushes BaseRom (nlp_formulation.cc:229-235)
      P->cons.emplace_back(new BaseMotionConstraint(P->T, dt_base_motion, P->sp, base_z_init));
    if (constraint_sets & ORC_SET_TOTAL_TIME)  // Parameters::OptimizePhaseDurations appends TotalTime (parameters.cc:76-80)
      for (int ee = 0; ee < n_ee; ++ee)
        P->cons.emplace_back(new TotalDurationConstraint(P->T, ee, P->schedules[ee].get()));
    return P;
  } catch (const std::exception&) {
    return nullptr;
  }
}

void orc_destroy(orc_problem* p) { delete p; }
int orc_n_vars(const orc_problem* p) { return p->n_vars(); }
int orc_n_rows(const orc_problem* p) { return p->n_rows(); }
int orc_n_var_sets(const orc_problem* p) { return p->n_sets(); }
int orc_n_con_sets(const orc_problem* p) { return (int)p->cons.size(); }
const char* orc_var_set_name(const orc_problem* p, int i) { return p->set_name(i).c_str(); }
int orc_var_set_size(const orc_problem* p, int i) { return p->set_rows(i); }
const char* orc_con_set_name(const orc_problem* p, int i) { return p->cons.at(i)->name.c_str(); }
int orc_con_set_rows(const orc_problem* p, int i) { return p->cons.at(i)->rows; }

void orc_initial_guess(orc_problem* P, const double* bl0, const double* ba0, const double* bl1, const double* ba1,
                       const double* ee_pos0, double* x_out) {
  V3 lin0(bl0[0], bl0[1], bl0[2]), ang0(ba0[0], ba0[1], ba0[2]);
  V3 lin1(bl1[0], bl1[1], bl1[2]), ang1(ba1[0], ba1[1], ba1[2]);
  double T = P->T;
  // ref: nlp_formulation.cc:95-125
  {
    double x = lin1(0), y = lin1(1);
    double z = P->terrain->GetHeight(x, y) - P->robot.nominal[0](2);
    P->var_sets[0]->SetByLinearInterpolation(lin0, V3(x, y, z), T);
    P->var_sets[1]->SetByLinearInterpolation(ang0, ang1, T);
  }
  // ref: nlp_formulation.cc:127-156
  for (int ee = 0; ee < P->n_ee; ++ee) {
    double yaw = ang1(2);
    M3 w_R_b = EulerConverter::RotationDense(V3(0.0, 0.0, yaw));
    V3 final_ee = lin1 + w_R_b * P->robot.nominal[ee];
    double x = final_ee(0), y = final_ee(1);
    double z = P->terrain->GetHeight(x, y);
    V3 p0(ee_pos0[3 * ee], ee_pos0[3 * ee + 1], ee_pos0[3 * ee + 2]);
    P->var_sets[2 + ee]->SetByLinearInterpolation(p0, V3(x, y, z), T);
  }
  // ref: nlp_formulation.cc:158-181
  for (int ee = 0; ee < P->n_ee; ++ee) {
    V3 f_stance(0.0, 0.0, P->model->m_ * P->model->g_ / P->n_ee);
    P->var_sets[2 + P->n_ee + ee]->SetByLinearInterpolation(f_stance, f_stance, T);
  }
  int off = 0;
  for (auto& v : P->var_sets) {
    v->GetValues(x_out + off);
    off += v->rows();
  }
  if (P->optimize_timings)
    for (auto& sc : P->schedules) {  // PhaseDurations::GetValues: the given phase durations (all but the last)
      sc->durations = P->phase_durations[&sc - &P->schedules[0]];
      sc->GetValues(x_out + off);
      off += sc->rows();
    }
}

// Variable bounds as NlpFormulation::Make{Base,Endeffector,Force}Variables set them
// (nlp_formulation.cc:109-122,151, parameters.cc:65-69); base states are {lin p, lin v, ang p, ang v}.
void orc_variable_bounds(orc_problem* P, const double* init_base, const double* final_base, const double* ee_pos0,
                         double* lower, double* upper) {
  auto v3 = [](const double* p) { return V3(p[0], p[1], p[2]); };
  for (auto& v : P->var_sets) v->bounds.assign(v->rows(), Bound{-kInf, kInf});
  NodesVars* lin = P->var_sets[0].get();
  lin->AddStartBound(kPos, {X, Y, Z}, v3(init_base));
  lin->AddStartBound(kVel, {X, Y, Z}, v3(init_base + 3));
  lin->AddFinalBound(kPos, {X, Y}, v3(final_base));        // bounds_final_lin_pos_
  lin->AddFinalBound(kVel, {X, Y, Z}, v3(final_base + 3)); // bounds_final_lin_vel_
  NodesVars* ang = P->var_sets[1].get();
  ang->AddStartBound(kPos, {X, Y, Z}, v3(init_base + 6));
  ang->AddStartBound(kVel, {X, Y, Z}, v3(init_base + 9));
  ang->AddFinalBound(kPos, {X, Y, Z}, v3(final_base + 6));
  ang->AddFinalBound(kVel, {X, Y, Z}, v3(final_base + 9));
  for (int ee = 0; ee < P->n_ee; ++ee) P->var_sets[2 + ee]->AddStartBound(kPos, {X, Y, Z}, v3(ee_pos0 + 3 * ee));
  int off = 0;
  for (auto& v : P->var_sets)
    for (const Bound& b : v->bounds) {
      lower[off] = b.lo;
      upper[off] = b.up;
      ++off;
    }
  if (P->optimize_timings)
    for (auto& sc : P->schedules)  // PhaseDurations::GetBounds, phase_durations.cc:105-114
      for (int i = 0; i < sc->rows(); ++i) {
        lower[off] = sc->phase_duration_bounds.lo;
        upper[off] = sc->phase_duration_bounds.up;
        ++off;
      }
}

// ifopt::ConstraintSet::GetJacobian + Composite row stacking + Problem::EvalNonzerosOfJacobian
int orc_eval(orc_problem* P, const double* x, double* g, int* row_ptr, int* col_idx, double* vals) {
  P->SetVariables(x);
  int nnz = 0, row0 = 0;
  if (row_ptr) row_ptr[0] = 0;
  for (auto& c : P->cons) {
    if (g) c->GetValues(g + row0);
    std::vector<SpMat> blocks;
    for (int vs = 0; vs < P->n_sets(); ++vs) {
      SpMat jac(c->rows, P->set_rows(vs));
      c->FillJacobianBlock(P->set_name(vs), jac);
      blocks.push_back(std::move(jac));
    }
    for (int r = 0; r < c->rows; ++r) {
      int col0 = 0;
      for (size_t s = 0; s < blocks.size(); ++s) {
        for (auto& e : blocks[s].rows[r].e) {
          if (col_idx) col_idx[nnz] = col0 + e.first;
          if (vals) vals[nnz] = e.second;
          ++nnz;
        }
        col0 += blocks[s].c;
      }
      if (row_ptr) row_ptr[row0 + r + 1] = nnz;
    }
    row0 += c->rows;
  }
  return nnz;
}

// fpowr::GetTrajectory (fpowr/include/fpowr/footstep_plan_extractor.h:19-53): the solution sampled every dt
// while t <= T + 1e-5 (t accumulated).  One record per sample, end-effectors in towr order (the reference
// relabels them to xpp ids, fpowr_xpp_ee_map.h:87-117):
//   [ t | base lin p v a (9) | quaternion w x y z (4) | omega (3) | omega_dot (3) |
//     per ee: contact (0/1), ee-motion p v a (9), ee-force p (3) ]        = 20 + 13 n_ee doubles
// The quaternion is Eigen::Quaterniond(R) (euler_converter.cc:51-56): Eigen 3.3 Quaternion.h
// quaternionbase_assign_impl<Matrix3d> restated (third party, not in /root/reference).
// fpowr::ExtractFootstepPlan (fpowr/include/fpowr/footstep_plan_extractor.h:69-133) without the ROS / boost::geometry
// nearest-plane lookup: the trajectory sampled every dt (GetTrajectory, :19-53), a footstep state wherever it is the
// first state or HasEndEffectorContactChanged (:55-67) against the previous state; duration = time to the next
// footstep state, the last one lasts until time_horizon (:121-128).  Record per footstep state:
//   [ t_global | duration | contact flag per ee | ee-motion position (3) per ee ]  = 2 + 4 n_ee doubles
int orc_contact_plan(orc_problem* P, const double* x, double dt, double time_horizon, double* out, int max_steps) {
  const int n = orc_sample_trajectory(P, x, dt, nullptr, 0);
  const int rec = 20 + 13 * P->n_ee, srec = 2 + 4 * P->n_ee;
  std::vector<double> traj((size_t)n * rec);
  orc_sample_trajectory(P, x, dt, traj.data(), n);
  std::vector<int> steps;
  for (int i = 0; i < n; ++i) {
    bool changed = i == 0;
    for (int ee = 0; ee < P->n_ee && !changed; ++ee)
      changed = traj[(size_t)i * rec + 20 + 13 * ee] != traj[(size_t)(i - 1) * rec + 20 + 13 * ee];
    if (changed) steps.push_back(i);
  }
  for (size_t s = 0; s < steps.size() && out && (int)s < max_steps; ++s) {
    const double* st = traj.data() + (size_t)steps[s] * rec;
    double* o = out + s * srec;
    o[0] = st[0];
    o[1] = s + 1 < steps.size() ? traj[(size_t)steps[s + 1] * rec] - st[0] : time_horizon - st[0];
    for (int ee = 0; ee < P->n_ee; ++ee) {
      o[2 + ee] = st[20 + 13 * ee];
      for (int d = 0; d < 3; ++d) o[2 + P->n_ee + 3 * ee + d] = st[20 + 13 * ee + 1 + d];
    }
  }
  return (int)steps.size();
}

// ---- fpowr::NearestPlaneLookup (fpowr/include/fpowr/nearest_plane_lookup.h).  Third-party pieces that are absent
// from /root/reference and restated from their published sources: tf::Matrix3x3(tf::Quaternion) (ROS tf / Bullet
// LinearMath Matrix3x3::setRotation) and boost::geometry::distance(point, polygon) (Boost 1.71 of the Ubuntu 20.04
// image: strategy/cartesian/point_in_poly_winding.hpp, strategies/cartesian/distance_projected_point.hpp,
// algorithms/detail/distance/point_to_geometry.hpp).  The reference appends the boundary points to a default
// bg::model::polygon (clockwise, CLOSED) without bg::correct: boost then walks the consecutive points exactly as given
// -- no closing edge is added -- and so does this restatement.  Deviation: boost compares coordinates / the side
// determinant with a few-ulp tolerance (math::equals), here exactly; it only matters for a point within rounding of a
// polygon boundary, where both give a distance of (almost) zero.
// PlanarRegionsToPolygons (:20-49): boundary point (x, y, 0) of a region rotated by its orientation, shifted by its
// position; the polygon keeps the world x, y.
void orc_planes_world_xy(const double* regions /* n x [position xyz, orientation xyzw] */, const double* local_xy,
                         const int* start /* n + 1 */, int n, double* out_xy) {
  for (int r = 0; r < n; ++r) {
    const double* P = regions + 7 * r;
    const double x = P[3], y = P[4], z = P[5], w = P[6];
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;   // Matrix3x3::setRotation
    const double xs = x * s, ys = y * s, zs = z * s, wz = w * zs, xx = x * xs, xy = x * ys, yy = y * ys, zz = z * zs;
    const double R00 = 1.0 - (yy + zz), R01 = xy - wz, R10 = xy + wz, R11 = 1.0 - (xx + zz);
    for (int i = start[r]; i < start[r + 1]; ++i) {
      const double lx = local_xy[2 * i], ly = local_xy[2 * i + 1];
      out_xy[2 * i] = (R00 * lx + R01 * ly + 0.0) + P[0];       // tf: m_el[0].dot(v) with v.z = 0, then + position
      out_xy[2 * i + 1] = (R10 * lx + R11 * ly + 0.0) + P[1];
    }
  }
}
namespace {
// boost winding strategy over the ring's consecutive points: 1 inside, 0 on the boundary, -1 outside
int RingSide(const double* xy, int n, double px, double py) {
  if (n < 4) return -1;   // core_detail::closure::minimum_ring_size<closed>
  int count = 0;
  for (int i = 0; i + 1 < n; ++i) {
    const double s1x = xy[2 * i], s1y = xy[2 * i + 1], s2x = xy[2 * i + 2], s2y = xy[2 * i + 3];
    const bool eq1 = s1x == px, eq2 = s2x == px;
    int c;
    if (eq1 && eq2) {   // vertical segment through px: touch if py lies on it
      if ((s1y <= py && s2y >= py) || (s2y <= py && s1y >= py)) return 0;
      c = 0;
    } else {
      c = eq1 ? (s2x > px ? 1 : -1) : eq2 ? (s1x > px ? -1 : 1) : (s1x < px && s2x > px) ? 2 : (s2x < px && s1x > px) ? -2 : 0;
    }
    if (c != 0) {
      int side;
      if (c == 1 || c == -1) {
        const double sey = eq1 ? s1y : s2y;
        side = py == sey ? 0 : (py < sey ? -c : c);
      } else {
        const double det = (s2x - s1x) * (py - s1y) - (s2y - s1y) * (px - s1x);   // side_by_triangle: > 0 left
        side = det > 0 ? 1 : (det < 0 ? -1 : 0);
      }
      if (side == 0) return 0;
      if (side * c > 0) count += c;
    }
  }
  return count == 0 ? -1 : 1;
}
double RingDistance(const double* xy, int n, double px, double py) {   // point_to_range + projected_point
  if (n == 0) return 0.0;
  auto comparable = [&](double ax, double ay, double bx, double by) {
    const double vx = bx - ax, vy = by - ay, wx = px - ax, wy = py - ay;
    const double c1 = wx * vx + wy * vy;
    if (c1 <= 0) return wx * wx + wy * wy;
    const double c2 = vx * vx + vy * vy;
    if (c2 <= c1) return (px - bx) * (px - bx) + (py - by) * (py - by);
    const double b = c1 / c2, qx = ax + b * vx, qy = ay + b * vy;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
  };
  if (n == 1) return std::sqrt(comparable(xy[0], xy[1], xy[0], xy[1]));
  double best = comparable(xy[0], xy[1], xy[2], xy[3]);
  for (int i = 0; i + 1 < n; ++i) {
    const double c = comparable(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3]);
    if (c == 0.0) return 0.0;
    if (c < best) best = c;
  }
  return std::sqrt(best);
}
}  // namespace
// NearestPlaneLookup::GetNearestPlaneIndex (:62-84): first polygon with the smallest bg::distance, -1 without polygons
int orc_nearest_plane(const double* world_xy, const int* start, int n_polys, double px, double py) {
  double min_distance = std::numeric_limits<double>::max();
  int nearest = -1;
  for (int i = 0; i < n_polys; ++i) {
    const double* xy = world_xy + 2 * start[i];
    const int n = start[i + 1] - start[i];
    const double distance = RingSide(xy, n, px, py) >= 0 ? 0.0 : RingDistance(xy, n, px, py);
    if (distance < min_distance) {
      min_distance = distance;
      nearest = i;
    }
  }
  return nearest;
}

// fpowr::ExtractInitialGuess (fpowr/include/fpowr/initial_guess_extractor.h:17-34), one record per requested time:
//   [ t | state: base-lin p (3), base-ang p = Euler angles (3), base-lin v (3), base-ang v = Euler rates (3) |
//     controls (36): ee-motion acceleration of ee i at 3 i, twelve zeros ("joint torques"), ee-force of ee i at 24 + 3 i ]
void orc_initial_guess_samples(orc_problem* P, const double* x, const double* times, int n_times, double* out) {
  P->SetVariables(x);
  for (int s = 0; s < n_times; ++s) {
    const double t = times[s];
    double* o = out + (size_t)s * 49;
    for (int i = 0; i < 49; ++i) o[i] = 0.0;
    o[0] = t;
    StateVal lin = P->sp.base_linear->GetPoint(t), ang = P->sp.base_angular->GetPoint(t);
    for (int i = 0; i < 3; ++i) {
      o[1 + i] = lin.p(i);
      o[4 + i] = ang.p(i);
      o[7 + i] = lin.v(i);
      o[10 + i] = ang.v(i);
    }
    for (int ee = 0; ee < P->n_ee; ++ee) {
      StateVal mo = P->sp.ee_motion[ee]->GetPoint(t);
      V3 f = P->sp.ee_force[ee]->GetPoint(t).p;
      for (int i = 0; i < 3; ++i) {
        o[13 + 3 * ee + i] = mo.a(i);
        o[13 + 24 + 3 * ee + i] = f(i);
      }
    }
  }
}

int orc_sample_trajectory(orc_problem* P, const double* x, double dt, double* out, int max_samples) {
  P->SetVariables(x);
  EulerConverter base_angular;
  base_angular.euler = P->sp.base_angular;
  const int rec = 20 + 13 * P->n_ee;
  double T = 0.0;  // Spline::GetTotalTime: accumulate over the polynomial durations (spline.cc:118-123)
  for (double d : P->sp.base_linear->GetPolyDurations()) T += d;
  int n = 0;
  double t = 0.0;
  while (t <= T + 1e-5) {
    if (out && n < max_samples) {
      double* o = out + (size_t)n * rec;
      StateVal lin = P->sp.base_linear->GetPoint(t);
      *o++ = t;
      for (int i = 0; i < 3; ++i) *o++ = lin.p(i);
      for (int i = 0; i < 3; ++i) *o++ = lin.v(i);
      for (int i = 0; i < 3; ++i) *o++ = lin.a(i);
      M3 m = EulerConverter::RotationDense(P->sp.base_angular->GetPoint(t).p);
      double q[4];  // x y z w (Eigen coeffs order)
      double tr = m(0, 0) + m(1, 1) + m(2, 2);
      if (tr > 0) {
        tr = std::sqrt(tr + 1.0);
        q[3] = 0.5 * tr;
        tr = 0.5 / tr;
        q[0] = (m(2, 1) - m(1, 2)) * tr;
        q[1] = (m(0, 2) - m(2, 0)) * tr;
        q[2] = (m(1, 0) - m(0, 1)) * tr;
      } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        tr = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * tr;
        tr = 0.5 / tr;
        q[3] = (m(k, j) - m(j, k)) * tr;
        q[j] = (m(j, i) + m(i, j)) * tr;
        q[k] = (m(k, i) + m(i, k)) * tr;
      }
      *o++ = q[3]; *o++ = q[0]; *o++ = q[1]; *o++ = q[2];
      V3 w = base_angular.GetAngularVelocityInWorld(t), wd = base_angular.GetAngularAccelerationInWorld(t);
      for (int i = 0; i < 3; ++i) *o++ = w(i);
      for (int i = 0; i < 3; ++i) *o++ = wd(i);
      for (int ee = 0; ee < P->n_ee; ++ee) {
        // PhaseDurations::IsContactPhase (phase_durations.cc:118-124)
        const PhaseDurations* pd = P->schedules[ee].get();
        int phase_id = NodeSpline::GetSegmentID(t, pd->durations);
        bool contact = phase_id % 2 == 0 ? pd->initial_contact_state : !pd->initial_contact_state;
        *o++ = contact ? 1.0 : 0.0;
        StateVal mo = P->sp.ee_motion[ee]->GetPoint(t);
        for (int i = 0; i < 3; ++i) *o++ = mo.p(i);
        for (int i = 0; i < 3; ++i) *o++ = mo.v(i);
        for (int i = 0; i < 3; ++i) *o++ = mo.a(i);
        V3 f = P->sp.ee_force[ee]->GetPoint(t).p;
        for (int i = 0; i < 3; ++i) *o++ = f(i);
      }
    }
    ++n;
    t += dt;
  }
  return n;
}

void orc_bounds(orc_problem* P, double* lower, double* upper) {
  int row0 = 0;
  for (auto& c : P->cons) {
    std::vector<Bound> b(c->rows);
    c->GetBounds(b.data());
    for (int i = 0; i < c->rows; ++i) {
      lower[row0 + i] = b[i].lo;
      upper[row0 + i] = b[i].up;
    }
    row0 += c->rows;
  }
}

double orc_time_callbacks(orc_problem* P, const double* x, int iters) {
  int m = P->n_rows();
  int nnz = orc_eval(P, x, nullptr, nullptr, nullptr, nullptr);
  std::vector<double> g(m), vals(nnz);
  std::vector<int> rp(m + 1), ci(nnz);
  auto t0 = std::chrono::steady_clock::now();
  double sink = 0;
  for (int i = 0; i < iters; ++i) {
    orc_eval(P, x, g.data(), rp.data(), ci.data(), vals.data());
    sink += g[i % m] + vals[i % nnz];
  }
  auto t1 = std::chrono::steady_clock::now();
  if (sink == 12345.678) g[0] = 1;  // keep the loop alive
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------------------------------------------------------- gait tables
namespace {
using Contact = std::vector<bool>;
using GaitInfo = std::pair<std::vector<double>, std::vector<Contact>>;
enum Gaits { Stand = 0, Flight, Walk1, Walk2, Walk2E, Run2, Run2E, Run1, Run1E, Run3, Run3E, Hop1, Hop1E, Hop2, Hop3, Hop3E, Hop5, Hop5E };

// quadruped contact code: s[0] = hind pair, s[1] = front pair; 'P' left, 'b' right, 'B' both, 'I' none
// ee order LF,RF,LH,RH (endeffector_mappings.h:44), ref: quadruped_gait_generator.cc:39-74
Contact Q(const char* s) {
  Contact c(4, false);
  auto set = [&](char ch, int left, int right) {
    if (ch == 'P' || ch == 'B') c[left] = true;
    if (ch == 'b' || ch == 'B') c[right] = true;
  };
  set(s[0], 2, 3);
  set(s[1], 0, 1);
  return c;
}
GaitInfo RemoveTransition(const GaitInfo& g) {  // ref: gait_generator.cc:131-144
  GaitInfo n = g;
  n.first.pop_back();
  n.first.back() += g.first.back();
  n.second.pop_back();
  return n;
}
GaitInfo QuadGait(int gait) {  // ref: quadruped_gait_generator.cc:89-366
  switch (gait) {
    case Stand: return {{0.3}, {Q("BB")}};
    case Flight: return {{0.3}, {Q("Bb")}};
    case Walk1: return {{0.3, 0.2, 0.3, 0.2, 0.3, 0.2, 0.3, 0.2}, {Q("bB"), Q("BB"), Q("Bb"), Q("BB"), Q("PB"), Q("BB"), Q("BP"), Q("BB")}};
    case Walk2:
    case Walk2E: {
      GaitInfo g = {{0.25, 0.13, 0.25, 0.13, 0.25, 0.13, 0.25, 0.13}, {Q("bB"), Q("bb"), Q("Bb"), Q("Pb"), Q("PB"), Q("PP"), Q("BP"), Q("bP")}};
      return gait == Walk2 ? g : RemoveTransition(g);
    }
    case Run1: return {{0.3, 0.2, 0.3, 0.2}, {Q("bP"), Q("BB"), Q("Pb"), Q("BB")}};
    case Run2: return {{0.4, 0.1, 0.4, 0.1}, {Q("bP"), Q("II"), Q("Pb"), Q("II")}};
    case Run2E: return {{0.4}, {Q("bP")}};
    case Run3: return {{0.3, 0.1, 0.3, 0.1}, {Q("PP"), Q("II"), Q("bb"), Q("II")}};
    case Run3E: return {{0.3}, {Q("PP")}};
    case Hop1: return {{0.3, 0.1, 0.3, 0.1}, {Q("BI"), Q("II"), Q("IB"), Q("II")}};
    case Hop1E: return {{0.3}, {Q("BI")}};
    case Hop2: return {{0.3, 0.4, 0.3}, {Q("BB"), Q("II"), Q("BB")}};
    case Hop3:
    case Hop3E: {
      GaitInfo g = {{0.2, 0.3, 0.2, 0.2, 0.2, 0.3, 0.2, 0.2}, {Q("Bb"), Q("BI"), Q("BP"), Q("bP"), Q("bB"), Q("IB"), Q("PB"), Q("Pb")}};
      return gait == Hop3 ? g : RemoveTransition(g);
    }
    case Hop5: return {{0.1, 0.2, 0.1, 0.1, 0.2, 0.1}, {Q("Bb"), Q("BB"), Q("IP"), Q("Bb"), Q("BB"), Q("IP")}};
  }
  throw std::runtime_error("gait not implemented");
}
std::vector<int> QuadCombo(int combo) {  // ref: quadruped_gait_generator.cc:76-87
  switch (combo) {
    case 0: return {Stand, Walk2, Walk2, Walk2, Walk2E, Stand};
    case 1: return {Stand, Run2, Run2, Run2, Run2E, Stand};
    case 2: return {Stand, Run3, Run3, Run3, Run3E, Stand};
    case 3: return {Stand, Hop1, Hop1, Hop1, Hop1E, Stand};
    case 4: return {Stand, Hop3, Hop3, Hop3, Hop3E, Stand};
  }
  throw std::runtime_error("combo");
}
// biped: L = index 0, R = index 1; ref: biped_gait_generator.cc:39-228
GaitInfo BipedGait(int gait) {
  Contact I = {false, false}, b = {false, true}, P = {true, false}, B = {true, true};
  switch (gait) {
    case Stand: return {{0.2}, {B}};
    case Flight: return {{0.5}, {I}};
    case Walk1:
    case Walk2: return {{0.3, 0.05, 0.3, 0.05}, {b, B, P, B}};
    case Run1:
    case Run3: return {{0.15, 0.4, 0.15 + 0.15, 0.4, 0.15}, {b, I, P, I, b}};
    case Hop1: return {{0.15, 0.5, 0.15}, {B, I, B}};
    case Hop2: return {{0.15, 0.4, 0.15}, {b, I, b}};
    case Hop3: return {{0.2, 0.2, 0.2}, {P, I, P}};
    case Hop5: return {{0.2, 0.3, 0.2, 0.2}, {P, I, b, B}};
  }
  throw std::runtime_error("gait not implemented");
}
std::vector<int> BipedCombo(int combo) {  // ref: biped_gait_generator.cc:51-62
  switch (combo) {
    case 0: return {Stand, Walk1, Walk1, Walk1, Walk1, Stand};
    case 1: return {Stand, Run1, Run1, Run1, Run1, Stand};
    case 2: return {Stand, Hop1, Hop1, Hop1, Stand};
    case 3: return {Stand, Hop1, Hop2, Hop2, Stand};
    case 4: return {Stand, Hop5, Hop5, Hop5, Stand};
  }
  throw std::runtime_error("combo");
}
GaitInfo MonoGait(int gait) {  // ref: monoped_gait_generator.cc:50-120
  Contact o = {true}, x = {false};
  switch (gait) {
    case Stand: return {{0.5}, {o}};
    case Flight: return {{0.5}, {x}};
    case Hop1: return {{0.3, 0.3}, {o, x}};
    case Hop2: return {{0.2, 0.3}, {o, x}};
  }
  throw std::runtime_error("gait not implemented");
}
std::vector<int> MonoCombo(int combo) {  // ref: monoped_gait_generator.cc:37-48
  switch (combo) {
    case 0: return {Stand, Hop1, Hop1, Hop1, Hop1, Stand};
    case 1: return {Stand, Hop1, Hop1, Hop1, Stand};
    case 2: return {Stand, Hop1, Hop1, Hop1, Hop1, Stand};
    case 3: return {Stand, Hop2, Hop2, Hop2, Stand};
    case 4: return {Stand, Hop2, Hop2, Hop2, Hop2, Hop2, Stand};
  }
  throw std::runtime_error("combo");
}
}  // namespace

int orc_gait(int n_ee, int combo, double t_total, int* n_phases, int* contact_at_start, double* out, int out_cap) {
  try {
    std::vector<double> times;
    std::vector<Contact> contacts;
    std::vector<int> gaits = n_ee == 1 ? MonoCombo(combo) : n_ee == 2 ? BipedCombo(combo) : QuadCombo(combo);
    for (int g : gaits) {  // ref: gait_generator.cc:113-129
      GaitInfo info = n_ee == 1 ? MonoGait(g) : n_ee == 2 ? BipedGait(g) : QuadGait(g);
      times.insert(times.end(), info.first.begin(), info.first.end());
      contacts.insert(contacts.end(), info.second.begin(), info.second.end());
    }
    // ref: gait_generator.cc:76-105
    std::vector<double> acc(n_ee, 0.0);
    std::vector<std::vector<double>> foot(n_ee);
    for (size_t phase = 0; phase + 1 < contacts.size(); ++phase) {
      const Contact& curr = contacts[phase];
      const Contact& next = contacts[phase + 1];
      for (int ee = 0; ee < n_ee; ++ee) {
        acc[ee] += times[phase];
        if (curr[ee] != next[ee]) {
          foot[ee].push_back(acc[ee]);
          acc[ee] = 0.0;
        }
      }
    }
    for (int ee = 0; ee < n_ee; ++ee) foot[ee].push_back(acc[ee] + times.back());
    int w = 0;
    for (int ee = 0; ee < n_ee; ++ee) {
      // ref: gait_generator.cc:54-74 (normalise then scale)
      std::vector<double> v = foot[ee];
      double total = std::accumulate(v.begin(), v.end(), 0.0);
      for (double& d : v) d = d / total;
      n_phases[ee] = (int)v.size();
      contact_at_start[ee] = contacts.front()[ee] ? 1 : 0;
      for (double d : v) {
        if (w >= out_cap) return -1;
        out[w++] = d * t_total;
      }
    }
    return w;
  } catch (const std::exception&) {
    return -1;
  }
}

void orc_hermite_weights(double t, double T, double w[12]) {
  CubicHermite p;
  p.T = T;
  int i = 0;
  for (int d : {kPos, kVel, kAcc}) {
    w[i++] = p.DerivWrtStartNode(d, kPos, t);
    w[i++] = p.DerivWrtStartNode(d, kVel, t);
    w[i++] = p.DerivWrtEndNode(d, kPos, t);
    w[i++] = p.DerivWrtEndNode(d, kVel, t);
  }
}
int orc_set_grid_map(orc_problem* P, const float* elevation, int size_x, int size_y, double resolution, double pos_x,
                     double pos_y) {
  if (!P || !elevation || size_x < 1 || size_y < 1 || !(resolution > 0) || P->terrain->id != 8) return -1;
  HeightMap& t = *P->terrain;
  t.gm.assign(elevation, elevation + (size_t)size_x * size_y);
  t.gm_sx = size_x; t.gm_sy = size_y;
  t.gm_res = resolution; t.gm_px = pos_x; t.gm_py = pos_y;
  t.gm_eps = resolution / 6.0;   // ref: grid_height_map.h:25
  return 0;
}
void orc_terrain_probe(const orc_problem* P, double x, double y, double out[3]) {
  out[0] = P->terrain->GetHeight(x, y);
  out[1] = P->terrain->GetDerivativeOfHeightWrt(X, x, y);
  out[2] = P->terrain->GetDerivativeOfHeightWrt(Y, x, y);
}
// --- probes for tests/test_oracle_symbolic.py (closed forms re-derived from the recipes of towr/matlab/*.m)
// CubicHermitePolynomial::GetDerivativeOfPosWrtDuration (polynomial.cc:236-257) of one scalar polynomial
double orc_hermite_dpos_dT(double t, double T, double p0, double v0, double p1, double v1) {
  CubicHermite c;
  c.T = T;
  c.n0.p = V3(p0, 0, 0); c.n0.v = V3(v0, 0, 0);
  c.n1.p = V3(p1, 0, 0); c.n1.v = V3(v1, 0, 0);
  c.UpdateCoeff();
  return c.GetDerivativeOfPosWrtDuration(t)(0);
}
// EulerConverter on ONE polynomial of base-ang (NodesVariablesAll order: node0 {px py pz vx vy vz}, node1 {...}):
// out = M[9] | Mdot[9] | R[9] | omega[3] | omega_dot[3]
//     | dM[dim][c][12] (GetDerivMwrtNodes, 108) | dMdot[dim][c][12] (GetDerivMdotwrtNodes, 108) | dR[row][col][12] (108)
//     | d omega / du [3][12] | d omega_dot / du [3][12]            = 33 + 324 + 72 = 429 doubles
void orc_euler_probe(const double nodes[12], double T, double t, double* out) {
  NodesVars nv = MakeNodesAll(2, "base-ang");
  NodeSpline sp(&nv, std::vector<double>{T});
  nv.observer = &sp;
  nv.SetVariables(nodes);
  EulerConverter ec;
  ec.euler = &sp;
  StateVal ori = sp.GetPoint(t);
  SpMat M = EulerConverter::GetM(ori.p), Md = EulerConverter::GetMdot(ori.p, ori.v);
  M3 R = EulerConverter::RotationDense(ori.p);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      out[3 * r + c] = M.coeff(r, c);
      out[9 + 3 * r + c] = Md.coeff(r, c);
      out[18 + 3 * r + c] = R(r, c);
    }
  V3 w = ec.GetAngularVelocityInWorld(t), wd = ec.GetAngularAccelerationInWorld(t);
  for (int i = 0; i < 3; ++i) {
    out[27 + i] = w(i);
    out[30 + i] = wd(i);
  }
  double* o = out + 33;
  for (int dim = 0; dim < 3; ++dim) {
    SpMat a = ec.GetDerivMwrtNodes(t, dim), b = ec.GetDerivMdotwrtNodes(t, dim);
    for (int c = 0; c < 3; ++c)
      for (int u = 0; u < 12; ++u) {
        o[(dim * 3 + c) * 12 + u] = a.coeff(c, u);
        o[108 + (dim * 3 + c) * 12 + u] = b.coeff(c, u);
      }
  }
  SpVec Rd[3][3];
  ec.GetDerivativeOfRotationMatrixWrtNodes(t, Rd);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      for (int u = 0; u < 12; ++u) o[216 + (r * 3 + c) * 12 + u] = Rd[r][c].get(u);
  SpMat dw = ec.GetDerivOfAngVelWrtEulerNodes(t), dwd = ec.GetDerivOfAngAccWrtEulerNodes(t);
  for (int r = 0; r < 3; ++r)
    for (int u = 0; u < 12; ++u) {
      o[324 + r * 12 + u] = dw.coeff(r, u);
      o[360 + r * 12 + u] = dwd.coeff(r, u);
    }
}
double orc_terrain_height(int terrain, double x, double y) { return HeightMap(terrain).GetHeight(x, y); }
void orc_terrain_basis(int terrain, int which, double x, double y, double out[3]) {
  V3 v = HeightMap(terrain).GetNormalizedBasis(which, x, y);
  for (int i = 0; i < 3; ++i) out[i] = v(i);
}
void orc_terrain_dbasis(int terrain, int which, int dim, double x, double y, double out[3]) {
  V3 v = HeightMap(terrain).GetDerivativeOfNormalizedBasisWrt(which, dim, x, y);
  for (int i = 0; i < 3; ++i) out[i] = v(i);
}

}  // extern "C"
