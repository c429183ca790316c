// TEST INFRASTRUCTURE ONLY -- BASELINE config 1 ("plumbing"): the hopper problem of towr/test/hopper_example.cc:45-90
// (monoped, flat ground, phases {0.4,0.2,0.4,0.2,0.4,0.2,0.2}, in contact at start) driven through the ifopt surface
// the way Ipopt's adapter drives it: AddVariableSet / AddConstraintSet (-> LinkWithVariables), EvaluateConstraints(x),
// EvalNonzerosOfJacobian(x, values), GetBoundsOnConstraints -- with towr_amd's device sets behind
// ifopt::ConstraintSet (towr_amd/csrc/ifopt_adapter.h).  Compiled against tests/ifopt_stub (this image has neither
// ifopt nor Eigen).  The result must equal a direct twr_batch_eval_host call bit for bit, in ifopt's stacking order.
//   hopper_adapter_test --gpu <sets> [gridmap|flat] [poll|strict|push]
//                                  -> needs a GPU, exit 0 on success; `gridmap`: on the `Grid` terrain fpowr hands the
//                                     solver (a grid_map elevation layer) instead of flat ground; how the adapter learns
//                                     that x moved: polled once per sweep (default), polled on every request, or
//                                     pushed by the variable sets' observers (what towr_binding.h does with towr's
//                                     NodesObserver / PhaseDurationsObserver)
//   hopper_adapter_test --quadruped [poll|strict|push] [iterations]
//                                  -> needs a GPU: ANYmal flying trot, towr's default list (19 constraint sets x 10
//                                     variable sets), one JSON line with the host and device microseconds per Ipopt
//                                     iteration (eval_g + eval_jac_g on a new x); scripts/latency.py prints it
//   hopper_adapter_test --no-gpu   -> everything up to the device: expects TWR_ERR_NO_DEVICE to surface as an exception
#include <ifopt/problem.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../towr_amd/csrc/ifopt_adapter.h"

namespace {
// A towr NodesVariables stand-in with the reference's COST SHAPE and its observer channel (nothing of the reference is
// compiled; this mirrors what the cited lines do so that the adapter's host cost is measured against the real thing's):
//   GetValues   nodes_variables.cc:52-62   per index a BY-VALUE std::vector<NodeValueInfo> (GetNodeValuesInfo:
//                                          nodes_variables_phase_based.h:168-170 = std::map::at + a vector copy, i.e. one
//                                          heap allocation), then nodes_.at(id).at(deriv)(dim)
//   SetVariables            :64-72         the same walk, writing; then UpdateObservers (:74-79)
//   AddObserver             :81-85         raw observer pointers, never dropped (nodes_observer.h:52)
struct ChangeObserver {
  virtual ~ChangeObserver() = default;
  virtual void Changed() = 0;   // NodesObserver::UpdateNodes / PhaseDurationsObserver::UpdatePolynomialDurations
};
class PlainVariables : public ifopt::VariableSet {
 public:
  struct Info { int id, deriv, dim; };
  PlainVariables(const std::string& name, const double* x0, int n) : ifopt::VariableSet(n, name), nodes_((n + 5) / 6) {
    for (int i = 0; i < n; ++i) {
      index_to_info_[i] = {Info{i / 6, (i % 6) / 3, i % 3}};
      nodes_[i / 6][(i % 6) / 3][i % 3] = x0[i];
    }
  }
  std::vector<Info> GetNodeValuesInfo(int idx) const { return index_to_info_.at(idx); }
  VectorXd GetValues() const override {
    ++reads;
    VectorXd x(GetRows());
    for (int idx = 0; idx < GetRows(); ++idx)
      for (auto nvi : GetNodeValuesInfo(idx)) x[idx] = nodes_.at(nvi.id).at(nvi.deriv).at(nvi.dim);
    return x;
  }
  void SetVariables(const VectorXd& x) override {
    for (int idx = 0; idx < GetRows(); ++idx)
      for (auto nvi : GetNodeValuesInfo(idx)) nodes_.at(nvi.id).at(nvi.deriv).at(nvi.dim) = x[idx];
    for (auto* o : observers_) o->Changed();
  }
  void AddObserver(ChangeObserver* o) { observers_.push_back(o); }
  VecBound GetBounds() const override { return VecBound(static_cast<size_t>(GetRows()), ifopt::NoBound); }
  static long reads;   // GetValues calls on any set, by anybody

 private:
  std::map<int, std::vector<Info>> index_to_info_;
  std::vector<std::array<std::array<double, 3>, 2>> nodes_;   // [node][deriv][dim]
  std::vector<ChangeObserver*> observers_;
};
long PlainVariables::reads = 0;

// what towr_binding.h's NodesDirtyObserver / DurationsDirtyObserver are for towr's subjects
class DirtyFlag final : public ChangeObserver {
 public:
  DirtyFlag(PlainVariables* subject, towr_amd::DeviceProblem* problem, int var_set) : problem_(problem), var_set_(var_set) {
    subject->AddObserver(this);
  }
  void Changed() override { problem_->MarkDirty(var_set_); }

 private:
  towr_amd::DeviceProblem* problem_;
  int var_set_;
};
void RegisterObservers(towr_amd::DeviceProblem& problem, const std::vector<ifopt::Component::Ptr>& sets) {
  for (size_t i = 0; i < sets.size(); ++i)
    if (auto v = std::dynamic_pointer_cast<PlainVariables>(sets[i])) {
      problem.KeepAlive(std::make_shared<DirtyFlag>(v.get(), &problem, static_cast<int>(i)));
      problem.EnablePush(static_cast<int>(i));
    }
}
enum Mode { kPoll, kStrict, kPush };
Mode ParseMode(const char* s) {
  if (std::string(s) == "push") return kPush;
  if (std::string(s) == "strict") return kStrict;
  return kPoll;
}
const char* ModeName(Mode m) { return m == kPush ? "push" : m == kStrict ? "poll on every request" : "poll per sweep"; }
void Configure(towr_amd::DeviceProblem& p, Mode mode) {
  if (mode == kPush) p.SetLinkHook(RegisterObservers);
  if (mode == kStrict) p.set_polling(towr_amd::DeviceProblem::Polling::kEveryRequest);
}
int fails = 0;
void expect(bool ok, const char* what) {
  if (!ok) {
    std::fprintf(stderr, "FAILED: %s\n", what);
    ++fails;
  }
}
}  // namespace

// ANYmal, gait combo C1 over 2 s, flat ground, default discretisation, towr's default constraint list: the problem fpowr
// solves (footstep_plan_server.cc:147-200) up to the robot constants.  One Ipopt iteration = eval_g + eval_jac_g on a new x.
int Quadruped(Mode mode, int iterations) {
  twr_model model;
  twr_model_preset(TWR_ROBOT_ANYMAL, TWR_TERRAIN_FLAT, &model);
  twr_schedule sched;
  twr_gait_combo(4, 1, 2.0, 1.0, &sched);
  twr_params prm;
  twr_params_default(&prm);
  prm.constraint_sets = TWR_SETS_TOWR_DEFAULT;
  std::shared_ptr<towr_amd::DeviceProblem> dp;
  std::vector<ifopt::ConstraintSet::Ptr> device_sets;
  try {
    device_sets = towr_amd::MakeDeviceConstraints(model, sched, prm, 0, nullptr, &dp);
  } catch (const std::exception& e) {
    std::printf("MakeDeviceConstraints: %s\n", e.what());
    return 2;
  }
  Configure(*dp, mode);
  const twr_sizes sz = dp->sizes();
  const double lin0[3] = {0, 0, 0.42}, ang0[3] = {0, 0, 0}, lin1[3] = {1.0, 0, 0.42};
  const double ee0[12] = {0.34, 0.19, 0, 0.34, -0.19, 0, -0.34, 0.19, 0, -0.34, -0.19, 0};
  std::vector<double> x(sz.n_vars);
  twr_structure_initial_guess(dp->structure(), lin0, ang0, lin1, ang0, ee0, x.data());
  ifopt::Problem nlp;
  std::vector<std::shared_ptr<PlainVariables>> vars;
  for (const twr_set_info& v : dp->var_sets()) {
    vars.push_back(std::make_shared<PlainVariables>(v.name, x.data() + v.offset, v.size));
    nlp.AddVariableSet(vars.back());
  }
  for (auto& c : device_sets) nlp.AddConstraintSet(c);
  std::vector<double> vals(sz.nnz);
  auto step = [&](int k) {
    for (int i = 0; i < sz.n_vars; ++i) x[i] += 1e-4 * std::sin(0.1 * k + 0.37 * i);
    (void)nlp.EvaluateConstraints(x.data());
    nlp.EvalNonzerosOfJacobian(x.data(), vals.data());
  };
  for (int k = 0; k < 5; ++k) step(k);
  // what ONE read of all of x costs with these variable sets (the unit round 4's rule paid 209 times per iteration)
  const auto r0 = std::chrono::steady_clock::now();
  const int reps = 200;
  double sink = 0;
  for (int r = 0; r < reps; ++r)
    for (auto& v : vars) sink += v->GetValues()[0];
  const double read_all_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count() / reps * 1e6;
  const long reads0 = dp->variable_reads(), v0 = dp->value_evaluations(), j0 = dp->jacobian_evaluations();
  const double rs0 = dp->read_seconds(), es0 = dp->eval_seconds();
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iterations; ++k) step(5 + k);
  const double wall_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iterations * 1e6;
  const int requests = sz.n_con_sets * (1 + sz.n_var_sets);
  std::printf("{\"case\": \"anymal_c1_default_list\", \"mode\": \"%s\", \"n_con_sets\": %d, \"n_var_sets\": %d, \"n\": %d, \"nnz\": %d, "
              "\"requests_per_iteration\": %d, \"variable_set_reads_per_iteration\": %.1f, \"host_change_detection_us_per_iteration\": %.1f, "
              "\"device_eval_host_us_per_iteration\": %.1f, \"wall_us_per_iteration_incl_stub_sparse_fill\": %.1f, \"read_all_sets_once_us\": %.2f, "
              "\"round4_rule_us_per_iteration\": %.1f, \"value_evals\": %ld, \"jacobian_evals\": %ld, \"iterations\": %d}\n",
              mode == kPush ? "push" : mode == kStrict ? "poll_on_every_request" : "poll_per_sweep", sz.n_con_sets, sz.n_var_sets, sz.n_vars, sz.nnz,
              requests, double(dp->variable_reads() - reads0) / iterations, (dp->read_seconds() - rs0) / iterations * 1e6,
              (dp->eval_seconds() - es0) / iterations * 1e6, wall_us, read_all_us, read_all_us * requests, dp->value_evaluations() - v0,
              dp->jacobian_evaluations() - j0, iterations);
  const bool ok = dp->value_evaluations() - v0 == iterations && dp->jacobian_evaluations() - j0 == iterations && sink == sink;
  // reads per iteration: push = one per set and SetVariables (2 per iteration); polled = one per set and sweep (2 sweeps);
  // on every request = round 4's rule
  const double per_it = double(dp->variable_reads() - reads0) / iterations;
  const double bound = mode == kStrict ? double(requests + sz.n_var_sets) * sz.n_var_sets : (mode == kPoll ? 3.0 : 2.0) * sz.n_var_sets;
  if (!ok || per_it > bound) {
    std::fprintf(stderr, "FAILED: %g reads per iteration (bound %g), evaluations %ld / %ld\n", per_it, bound, dp->value_evaluations() - v0,
                 dp->jacobian_evaluations() - j0);
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--quadruped")
    return Quadruped(argc > 2 ? ParseMode(argv[2]) : kPoll, argc > 3 ? std::atoi(argv[3]) : 100);
  const bool no_gpu = argc > 1 && std::string(argv[1]) == "--no-gpu";
  const int sets = argc > 2 ? std::atoi(argv[2]) : TWR_SETS_TOWR_DEFAULT;
  const bool gridmap = argc > 3 && std::string(argv[3]) == "gridmap";
  const Mode mode = argc > 4 ? ParseMode(argv[4]) : kPoll;
  twr_model model;
  twr_model_preset(TWR_ROBOT_MONOPED, gridmap ? TWR_TERRAIN_GRID_MAP : TWR_TERRAIN_FLAT, &model);
  // a perception-style elevation layer: 64 x 40 cells of 4 cm around (0.6, 0), a ramp with two steps (float, column-major
  // [size_x][size_y] like grid_map's Eigen::MatrixXf)
  twr_terrain_grid* grid = nullptr;
  std::vector<float> elevation;
  if (gridmap) {
    const int sx = 64, sy = 40;
    elevation.resize(static_cast<size_t>(sx) * sy);
    for (int j = 0; j < sy; ++j)
      for (int i = 0; i < sx; ++i) {
        const double xw = 0.6 + 0.5 * sx * 0.04 - (i + 0.5) * 0.04;   // cell centre, grid_map convention
        elevation[i + static_cast<size_t>(j) * sx] = static_cast<float>(0.05 * xw + (xw > 0.4 ? 0.03 : 0.0) + (xw > 0.8 ? 0.04 : 0.0) + 0.002 * j);
      }
    if (twr_terrain_grid_map_create(elevation.data(), sx, sy, 0.04, 0.6, 0.0, &grid) != TWR_OK) {
      std::printf("twr_terrain_grid_map_create: %s\n", twr_last_error());
      return 2;
    }
  }
  twr_schedule sched;
  std::memset(&sched, 0, sizeof(sched));
  sched.n_ee = 1;
  sched.n_phases[0] = 7;
  sched.in_contact_at_start[0] = 1;
  const double ph[7] = {0.4, 0.2, 0.4, 0.2, 0.4, 0.2, 0.2};  // hopper_example.cc:67
  for (int i = 0; i < 7; ++i) sched.phase_durations[0][i] = ph[i];
  twr_params prm;
  twr_params_default(&prm);
  prm.constraint_sets = sets;  // towr's default constraints_ (parameters.cc:55-60), or + TotalTime

  std::vector<ifopt::ConstraintSet::Ptr> device_sets;
  std::shared_ptr<towr_amd::DeviceProblem> shared_problem;
  try {
    device_sets = towr_amd::MakeDeviceConstraints(model, sched, prm, /*device=*/0, grid, &shared_problem);
  } catch (const std::exception& e) {
    std::printf("MakeDeviceConstraints: %s\n", e.what());
    if (no_gpu) return std::strstr(e.what(), "no HIP device") ? 0 : 2;
    return 2;
  }
  Configure(*shared_problem, mode);
  if (no_gpu) {
    std::printf("a GPU is visible: --no-gpu has nothing to check\n");
    return 0;
  }

  // reference data straight from the C ABI (what the adapter must reproduce)
  twr_structure* S = nullptr;
  expect(twr_structure_create_with_grid(&model, &sched, &prm, grid, &S) == TWR_OK, "twr_structure_create_with_grid");
  twr_sizes sz;
  twr_structure_sizes(S, &sz);
  const double lin0[3] = {0, 0, 0.5}, ang0[3] = {0, 0, 0}, lin1[3] = {1.0, 0, 0.5}, ee0[3] = {0, 0, 0};  // hopper_example.cc:53-59
  std::vector<double> x(sz.n_vars);
  twr_structure_initial_guess(S, lin0, ang0, lin1, ang0, ee0, x.data());
  for (int i = 0; i < sz.n_vars; ++i) x[i] += 0.01 * std::sin(1.0 + 0.37 * i);  // off the trivial guess
  const twr_structure* list[1] = {S};
  const int32_t map[1] = {0};
  twr_batch* B = nullptr;
  expect(twr_batch_create(list, 1, map, 1, 0, &B) == TWR_OK, "twr_batch_create");
  std::vector<double> g_ref(sz.n_rows), j_ref(sz.nnz), lo(sz.n_rows), up(sz.n_rows);
  expect(twr_batch_eval_host(B, x.data(), g_ref.data(), j_ref.data(), TWR_EVAL_BOTH) == TWR_OK, "twr_batch_eval_host");
  twr_structure_bounds(S, lo.data(), up.data());

  // the ifopt problem: variable sets in the reference order (nlp_formulation.cc:63-93), then the device sets
  ifopt::Problem nlp;
  for (int i = 0; i < sz.n_var_sets; ++i) {
    twr_set_info v;
    twr_structure_var_set(S, i, &v);
    nlp.AddVariableSet(std::make_shared<PlainVariables>(v.name, x.data() + v.offset, v.size));
  }
  for (auto& c : device_sets) nlp.AddConstraintSet(c);
  expect(nlp.GetNumberOfOptimizationVariables() == sz.n_vars, "variable count");
  expect(nlp.GetNumberOfConstraints() == sz.n_rows, "constraint count");
  expect(static_cast<int>(device_sets.size()) == sz.n_con_sets, "one ifopt set per constraint set");
  for (int i = 0; i < sz.n_con_sets; ++i) {
    twr_set_info c;
    twr_structure_con_set(S, i, &c);
    expect(device_sets[i]->GetName() == c.name && device_sets[i]->GetRows() == c.size, "component name / rows");
  }

  // Ipopt: eval_g
  x[3] += 1e-3;  // a new x through Problem::SetVariables, not the one the sets were created at
  // (the adapter evaluates values for eval_g and the Jacobian alone for eval_jac_g; the kernels are instantiated per output
  // selection, so "bit for bit" is against the same selection -- the selections agree with each other to rounding, checked
  // below)
  std::vector<double> g_both(sz.n_rows), j_both(sz.nnz);
  expect(twr_batch_eval_host(B, x.data(), g_both.data(), j_both.data(), TWR_EVAL_BOTH) == TWR_OK, "twr_batch_eval_host");
  expect(twr_batch_eval_host(B, x.data(), g_ref.data(), nullptr, TWR_EVAL_VALUES) == TWR_OK, "twr_batch_eval_host values");
  expect(twr_batch_eval_host(B, x.data(), nullptr, j_ref.data(), TWR_EVAL_JACOBIAN) == TWR_OK, "twr_batch_eval_host jacobian");
  {
    double sg = 0, sj = 0, eg = 0, ej = 0;
    for (int i = 0; i < sz.n_rows; ++i) { sg = std::fmax(sg, std::fabs(g_both[i])); eg = std::fmax(eg, std::fabs(g_both[i] - g_ref[i])); }
    for (int k = 0; k < sz.nnz; ++k) { sj = std::fmax(sj, std::fabs(j_both[k])); ej = std::fmax(ej, std::fabs(j_both[k] - j_ref[k])); }
    expect(eg <= 1e-13 * sg && ej <= 1e-13 * sj, "values-only / Jacobian-only evaluations agree with the evaluation of both to rounding");
  }
  ifopt::Problem::VectorXd g = nlp.EvaluateConstraints(x.data());
  double dg = 0;
  for (int i = 0; i < sz.n_rows; ++i) dg = std::fmax(dg, std::fabs(g[i] - g_ref[i]));
  expect(dg == 0.0, "stacked GetValues equals the direct evaluation");
  // Ipopt: eval_jac_g structure + values
  ifopt::Problem::Jacobian jac = nlp.GetJacobianOfConstraints();
  jac.makeCompressed();
  expect(jac.nonZeros() == sz.nnz, "nnz of the stacked Jacobian (explicit zeros kept)");
  const int32_t* rp = twr_structure_row_ptr(S);
  const int32_t* ci = twr_structure_col_idx(S);
  bool pattern = jac.nonZeros() == sz.nnz;
  for (int r = 0; pattern && r <= sz.n_rows; ++r) pattern = jac.outerIndexPtr()[r] == rp[r];
  for (int k = 0; pattern && k < sz.nnz; ++k) pattern = jac.innerIndexPtr()[k] == ci[k];
  expect(pattern, "CSR pattern equals twr_structure_row_ptr / col_idx");
  std::vector<double> vals(sz.nnz);
  nlp.EvalNonzerosOfJacobian(x.data(), vals.data());
  double dj = 0;
  for (int k = 0; k < sz.nnz; ++k) dj = std::fmax(dj, std::fabs(vals[k] - j_ref[k]));
  expect(dj == 0.0, "EvalNonzerosOfJacobian equals the direct evaluation");
  // bounds
  ifopt::Problem::VecBound b = nlp.GetBoundsOnConstraints();
  bool bounds = static_cast<int>(b.size()) == sz.n_rows;
  for (int r = 0; bounds && r < sz.n_rows; ++r) bounds = b[r].lower_ == lo[r] && b[r].upper_ == up[r];
  expect(bounds, "GetBounds equals twr_structure_bounds");

  // evaluate what was asked (time_discretization_constraint.cc:65-96): Ipopt's eval_g, eval_g (line-search trial point),
  // eval_jac_g on that point = two values-only evaluations and ONE Jacobian-only evaluation, nothing more
  {
    const auto* first = dynamic_cast<const towr_amd::DeviceConstraintSet*>(device_sets[0].get());
    expect(first != nullptr, "device sets are DeviceConstraintSet objects");
    const towr_amd::DeviceProblem& dp = first->problem();
    const long v0 = dp.value_evaluations(), j0 = dp.jacobian_evaluations();
    std::vector<double> xa = x, xb = x;
    xa[5] += 2e-3;
    xb[5] -= 1e-3;
    (void)nlp.EvaluateConstraints(xa.data());                       // eval_g
    ifopt::Problem::VectorXd gb = nlp.EvaluateConstraints(xb.data());   // eval_g at the accepted trial point
    expect(dp.value_evaluations() == v0 + 2 && dp.jacobian_evaluations() == j0, "eval_g, eval_g: two values-only evaluations");
    nlp.EvalNonzerosOfJacobian(xb.data(), vals.data());             // eval_jac_g on the same x
    expect(dp.value_evaluations() == v0 + 2 && dp.jacobian_evaluations() == j0 + 1, "eval_jac_g on a known x: one Jacobian-only evaluation");
    nlp.EvalNonzerosOfJacobian(xb.data(), vals.data());             // again: nothing to do
    (void)nlp.EvaluateConstraints(xb.data());
    expect(dp.value_evaluations() == v0 + 2 && dp.jacobian_evaluations() == j0 + 1, "same x again: no evaluation");
    expect(twr_batch_eval_host(B, xb.data(), g_ref.data(), nullptr, TWR_EVAL_VALUES) == TWR_OK, "twr_batch_eval_host");
    expect(twr_batch_eval_host(B, xb.data(), nullptr, j_ref.data(), TWR_EVAL_JACOBIAN) == TWR_OK, "twr_batch_eval_host");
    double d2 = 0;
    for (int i = 0; i < sz.n_rows; ++i) d2 = std::fmax(d2, std::fabs(gb[i] - g_ref[i]));
    for (int k = 0; k < sz.nnz; ++k) d2 = std::fmax(d2, std::fabs(vals[k] - j_ref[k]));
    expect(d2 == 0.0, "the adapter's values-only + Jacobian-only evaluations equal the direct ones");
    // ... and x is READ once per new x, not once per request.  A sweep = what one Problem call asks:
    // n_con_sets requests for eval_g, n_con_sets * n_var_sets for eval_jac_g.  Pushed: exactly one read per variable set
    // and Problem::SetVariables (the flag of every set is raised by it, nodes_variables.cc:64-79).  Polled: one read per
    // set and sweep.  (kEveryRequest is round 4's rule: one read per set and request.)
    {
      const int nv = sz.n_var_sets, nc = sz.n_con_sets;
      std::vector<double> xc = x;
      long worst_g = 0, worst_j = 0;
      for (int k = 0; k < 3; ++k) {
        xc[7] += 1e-3;
        long r0 = dp.variable_reads();
        (void)nlp.EvaluateConstraints(xc.data());
        worst_g = std::max(worst_g, dp.variable_reads() - r0);
        xc[9] -= 1e-3;
        r0 = dp.variable_reads();
        (void)nlp.EvaluateConstraints(xc.data());
        worst_g = std::max(worst_g, dp.variable_reads() - r0);
        r0 = dp.variable_reads();
        nlp.EvalNonzerosOfJacobian(xc.data(), vals.data());
        worst_j = std::max(worst_j, dp.variable_reads() - r0);
      }
      if (mode == kPush) expect(worst_g == nv && worst_j == nv, "pushed: one read per variable set and SetVariables");
      // (Composite::GetJacobian asks its FIRST component twice -- once for the column count --, so the first set's blocks
      // repeat inside an eval_jac_g sweep and a polled problem reads x twice there)
      if (mode == kPoll) expect(worst_g == nv && worst_j <= 2 * nv, "polled per sweep: one read per variable set and sweep");
      if (mode == kStrict) expect(worst_g == nv * nc && worst_j >= nv * nc * nv, "polled on every request: one read per set and request");
      // nobody but the adapter's counted reads touched the variable sets in those sweeps
      expect(PlainVariables::reads > 0, "the stand-in counts GetValues");
      std::vector<double> g_chk(sz.n_rows);
      expect(twr_batch_eval_host(B, xc.data(), g_chk.data(), j_ref.data(), TWR_EVAL_JACOBIAN | TWR_EVAL_VALUES) == TWR_OK, "twr_batch_eval_host");
      expect(twr_batch_eval_host(B, xc.data(), nullptr, j_ref.data(), TWR_EVAL_JACOBIAN) == TWR_OK, "twr_batch_eval_host");
      double d3 = 0;
      for (int kk = 0; kk < sz.nnz; ++kk) d3 = std::fmax(d3, std::fabs(vals[kk] - j_ref[kk]));
      expect(d3 == 0.0, "the Jacobian after three iterations equals the direct evaluation");
    }
    // an x that moves BETWEEN two requests of one sweep (a caller that is not ifopt::Problem): noticed when the sets push
    // or every request polls; the per-sweep rule's contract excludes it (ifopt_adapter.h, "x changed")
    if (mode != kPoll) {
      std::vector<double> xm = xb;
      (void)nlp.EvaluateConstraints(xm.data());
      xm[2] += 5e-3;
      nlp.SetVariables(xm.data());
      const int last = sz.n_con_sets - 1;
      const ifopt::Problem::VectorXd g_last = device_sets[last]->GetValues();   // not the first request of a sweep
      expect(twr_batch_eval_host(B, xm.data(), g_ref.data(), nullptr, TWR_EVAL_VALUES) == TWR_OK, "twr_batch_eval_host");
      twr_set_info c;
      twr_structure_con_set(S, last, &c);
      double d4 = 0;
      for (int i = 0; i < c.size; ++i) d4 = std::fmax(d4, std::fabs(g_last[i] - g_ref[c.offset + i]));
      expect(d4 == 0.0, "a mid-sweep change of x is noticed");
    }
  }

  // a host NLP with a variable set of its own (ifopt convention: constraints leave the blocks of sets they do not depend on
  // empty): the device sets read their x by name and return an empty block for the foreign set
  {
    const double slack[3] = {0.1, 0.2, 0.3};
    ifopt::Problem host;
    for (int i = 0; i < sz.n_var_sets; ++i) {
      twr_set_info v;
      twr_structure_var_set(S, i, &v);
      if (i == 1) host.AddVariableSet(std::make_shared<PlainVariables>("slack-of-the-host", slack, 3));   // in the middle
      host.AddVariableSet(std::make_shared<PlainVariables>(v.name, x.data() + v.offset, v.size));
    }
    bool ok = true;
    try {
      std::shared_ptr<towr_amd::DeviceProblem> hp;
      auto host_sets = towr_amd::MakeDeviceConstraints(model, sched, prm, 0, grid, &hp);
      Configure(*hp, mode);
      for (auto& c : host_sets) host.AddConstraintSet(c);
      expect(twr_batch_eval_host(B, x.data(), nullptr, j_ref.data(), TWR_EVAL_JACOBIAN) == TWR_OK, "twr_batch_eval_host");
      ifopt::Problem::Jacobian jh = host.GetJacobianOfConstraints();
      jh.makeCompressed();
      ok = jh.nonZeros() == sz.nnz;
      twr_set_info v1;
      twr_structure_var_set(S, 1, &v1);
      for (int r = 0; ok && r < sz.n_rows; ++r) {
        int k = rp[r];
        for (const auto& kv : jh.row(r)) {
          const int col = kv.first < v1.offset ? kv.first : kv.first - 3;   // the foreign set sits in front of variable set 1
          ok = ok && !(kv.first >= v1.offset && kv.first < v1.offset + 3) && k < rp[r + 1] && ci[k] == col && kv.second == j_ref[k];
          ++k;
        }
        ok = ok && k == rp[r + 1];
      }
    } catch (const std::exception& e) {
      std::fprintf(stderr, "foreign variable set: %s\n", e.what());
      ok = false;
    }
    expect(ok, "a foreign variable set gets an empty block and does not disturb the towr sets");
  }

  // a variable set the structure needs but the NLP lacks: a clear error when the sets are linked (real ifopt's
  // Composite::GetComponent asserts and returns null there -- the adapter never calls it, it walks GetComponents())
  {
    bool threw = false;
    try {
      ifopt::Problem lacking;
      for (int i = 0; i + 1 < sz.n_var_sets; ++i) {
        twr_set_info v;
        twr_structure_var_set(S, i, &v);
        lacking.AddVariableSet(std::make_shared<PlainVariables>(v.name, x.data() + v.offset, v.size));
      }
      for (auto& c : towr_amd::MakeDeviceConstraints(model, sched, prm, 0, grid)) lacking.AddConstraintSet(c);
    } catch (const std::exception& e) {
      twr_set_info v;
      twr_structure_var_set(S, sz.n_var_sets - 1, &v);
      threw = std::strstr(e.what(), "has no variable set") != nullptr && std::strstr(e.what(), v.name) != nullptr;
    }
    expect(threw, "linking with a composite that lacks a variable set throws and names the set");
  }

  // a towr-style variable-set name the structure does not know (a typo) must throw instead of yielding a silent zero block
  {
    const double one[1] = {0.3};
    bool threw = false;
    try {
      ifopt::Problem bad;
      for (int i = 0; i < sz.n_var_sets; ++i) {
        twr_set_info v;
        twr_structure_var_set(S, i, &v);
        bad.AddVariableSet(std::make_shared<PlainVariables>(v.name, x.data() + v.offset, v.size));
      }
      bad.AddVariableSet(std::make_shared<PlainVariables>("ee-motoin_0", one, 1));
      for (auto& c : towr_amd::MakeDeviceConstraints(model, sched, prm, 0, grid)) bad.AddConstraintSet(c);
      (void)bad.GetJacobianOfConstraints();
    } catch (const std::exception& e) {
      threw = std::strstr(e.what(), "unknown variable set") != nullptr;
    }
    expect(threw, "FillJacobianBlock throws on an unknown towr-style variable set name");
  }

  std::printf("x-change detection: %s, %ld variable-set reads in all\n", ModeName(mode), shared_problem->variable_reads());
  std::printf("hopper through ifopt%s: n=%d m=%d nnz=%d sets=%d  max|dg|=%g max|dJ|=%g  %s\n", gridmap ? " on a grid_map terrain" : "",
              sz.n_vars, sz.n_rows, sz.nnz, sz.n_con_sets, dg, dj, fails ? "FAILED" : "ok");
  twr_batch_destroy(B);
  twr_structure_destroy(S);
  twr_terrain_grid_destroy(grid);
  return fails ? 1 : 0;
}
