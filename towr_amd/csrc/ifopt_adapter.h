// ifopt adapter: presents the device path as ordinary ifopt::ConstraintSet objects so that CPU
// Ipopt keeps driving the outer solve and NlpFormulation is unchanged apart from which classes
// it instantiates (INTEGRATION.md shows the three-line change in nlp_formulation.cc).
//
// COMPILE-GATED: needs <ifopt/constraint_set.h> and Eigen, which this build image does not have
// (SURVEY.md section 0); the core library does not depend on this header.  It is the only C++/Eigen-typed
// code on top of the C ABI of include/towr_amd.h.  In this image it is compiled (-Wall -Werror) and run on the
// hopper problem against tests/ifopt_stub/ (stand-ins for exactly the ifopt/Eigen surface of SURVEY App. C):
// tests/test_ifopt_adapter.py.
//
// Replaces (same component names, same row order, same Jacobian pattern incl. explicit zeros):
//   towr::TerrainConstraint        "terrain-ee-motion_<ee>"   towr/src/terrain_constraint.cc:36-108
//   towr::DynamicConstraint        "dynamic"                  towr/src/dynamic_constraint.cc:37-137
//   towr::RangeOfMotionConstraint  "rangeofmotion-<ee>"       towr/src/range_of_motion_constraint.cc:35-109
//   towr::ForceConstraint          "force-ee-force_<ee>"      towr/src/force_constraint.cc:37-171
// and, when selected in twr_params.constraint_sets,
//   towr::SplineAccConstraint      "splineacc-base-lin|ang"   towr/src/spline_acc_constraint.cc:34-88
//   towr::SwingConstraint          "swing-ee-motion_<ee>"     towr/src/swing_constraint.cc:35-121
//   towr::TotalDurationConstraint  "totalduration-<ee>"       towr/src/total_duration_constraint.cc:36-72
//   towr::BaseMotionConstraint     "baseMotion"               towr/src/base_motion_constraint.cc:38-99
// (with TWR_SET_TOTAL_TIME the variable composite also holds the reference's "ee-schedule<ee>" sets and
// the dynamic / rangeofmotion sets return their duration columns for them)
#pragma once
#if __has_include(<ifopt/constraint_set.h>)
#include <ifopt/constraint_set.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "towr_amd.h"

namespace towr_amd {

// One problem on one GPU.  All adapter sets of a problem share it; x is read from the variable sets once per new x
// (see "x changed" below), uploaded, and the kernels run once per new x and kind of result (ifopt calls GetValues /
// FillJacobianBlock once per set and per variable set).
class DeviceProblem {
 public:
  // `grid`: the gridded terrain for model.terrain_id == TWR_TERRAIN_GRID_MAP (the `Grid` height map fpowr hands
  // NlpFormulation::terrain_, footstep_plan_server.cc:155) or TWR_TERRAIN_CSV_GRID; nullptr for the analytic terrains.
  DeviceProblem(const twr_model& model, const twr_schedule& schedule, const twr_params& params, int device = 0,
                const twr_terrain_grid* grid = nullptr) {
    Check(twr_structure_create_with_grid(&model, &schedule, &params, grid, &structure_));
    const twr_structure* list[1] = {structure_};
    const int32_t map[1] = {0};
    Check(twr_batch_create(list, 1, map, 1, device, &batch_));
    Check(twr_structure_sizes(structure_, &sizes_));
    Check(twr_batch_host_buffers(batch_, &x_, &g_, &jac_));  // page-locked, owned by the batch
    lower_.resize(sizes_.n_rows);
    upper_.resize(sizes_.n_rows);
    Check(twr_structure_bounds(structure_, lower_.data(), upper_.data()));
    for (int i = 0; i < sizes_.n_var_sets; ++i) {
      twr_set_info v;
      Check(twr_structure_var_set(structure_, i, &v));
      var_sets_.push_back(v);
      VarSet s;
      s.info = v;
      vars_.push_back(s);
    }
  }
  ~DeviceProblem() {
    twr_batch_destroy(batch_);
    twr_structure_destroy(structure_);
  }
  DeviceProblem(const DeviceProblem&) = delete;
  DeviceProblem& operator=(const DeviceProblem&) = delete;

  // ---- "x changed" -------------------------------------------------------------------------------------------------
  // The reference PUSHES: NodesVariables::SetVariables -> UpdateObservers -> NodesObserver::UpdateNodes
  // (nodes_variables.cc:64-79, nodes_observer.h:52-68; PhaseDurations likewise, phase_durations.cc:52-103), so its splines
  // never ask whether x moved.  ifopt itself has no such channel (a ConstraintSet only ever sees GetVariables()), and
  // ifopt calls FillJacobianBlock once per (constraint set x variable set): 190 calls + 19 GetValues per Ipopt iteration
  // for the quadruped default list.  Re-reading all of x on each of them (rounds 3-4) is O(sets^2 * n) host work per
  // iteration -- ~10 ms with towr's allocating NodesVariables::GetValues (nodes_variables.cc:52-62) against 33 us on the
  // device.  So:
  //  * Link() (from InitVariableDependedQuantities, the pattern of force_constraint.cc:50-54) resolves every variable set
  //    ONCE, by walking the composite; a missing set is a std::runtime_error that names it.
  //  * a variable set that can push (towr's: towr_binding.h registers a NodesObserver / PhaseDurationsObserver on it
  //    through the link hook; any host: call MarkDirty) is read when its flag is set, i.e. once per SetVariables.
  //  * a set that cannot is POLLED once per SWEEP.  ifopt::Problem asks every set for its values once per eval_g
  //    (Composite::GetValues) and for every block once per eval_jac_g (ConstraintSet::GetJacobian), each right after ONE
  //    SetVariables.  So a new sweep -- the only moment x can have moved -- shows as a request that was already served
  //    since the last read, or as a change of kind (values after blocks, blocks after values); that request reads the
  //    polled sets, the others of the sweep read nothing.  Contract of this default (kPerSweep): x does not move between
  //    two requests of the same kind unless one of them repeats.  A caller that cannot promise that (it moves x and asks
  //    ANOTHER set for the same kind of result) selects kEveryRequest -- every request re-reads the polled sets, always
  //    right, O(requests * n) -- or calls MarkDirty() when it moves x.
  enum class Polling { kPerSweep, kEveryRequest };
  void set_polling(Polling p) { polling_ = p; }
  using VariablesPtr = ifopt::ConstraintSet::VariablesPtr;
  using LinkHook = std::function<void(DeviceProblem&, const std::vector<ifopt::Component::Ptr>&)>;

  // Called with the resolved variable sets (in the structure's order) at the end of every Link(); it may register
  // change notifications on them and reports each set that now pushes with EnablePush(index).
  void SetLinkHook(LinkHook hook) { link_hook_ = std::move(hook); }
  void EnablePush(int var_set) { vars_.at(static_cast<size_t>(var_set)).push = true; }
  // keeps whatever the hook registered (observer objects) alive as long as this problem
  void KeepAlive(std::shared_ptr<void> p) { keep_.push_back(std::move(p)); }
  // The push entry point: variable set `var_set` (index in the structure's order; -1 = all) has new values.
  void MarkDirty(int var_set = -1) {
    if (var_set < 0)
      for (auto& v : vars_) v.dirty = true;
    else
      vars_.at(static_cast<size_t>(var_set)).dirty = true;
  }

  void Link(const VariablesPtr& vars) {
    if (!vars) throw std::runtime_error("towr_amd: LinkWithVariables(nullptr)");
    if (linked_ == vars.get()) return;   // every set of the problem links with the same composite
    const auto comps = vars->GetComponents();
    std::vector<ifopt::Component::Ptr> found;
    for (auto& v : vars_) {
      ifopt::Component::Ptr c;
      for (const auto& k : comps)
        if (k->GetName() == v.info.name) c = k;
      if (!c) throw std::runtime_error(std::string("towr_amd: the NLP has no variable set '") + v.info.name + "' (the structure needs it)");
      if (c->GetRows() != v.info.size)
        throw std::runtime_error(std::string("towr_amd: variable set '") + v.info.name + "' has another size than the structure's");
      v.comp = c.get();   // ifopt's variable composite owns the set and outlives its constraint sets' calls
      v.push = false;
      v.dirty = true;
      found.push_back(c);
    }
    linked_ = vars.get();
    // (observers of an EARLIER link stay alive in keep_: their subjects hold raw pointers to them and may still call them --
    // all such a call does is flag a set, i.e. cost one extra read)
    have_ = 0;
    have_x_ = false;
    if (link_hook_) link_hook_(*this, found);
    n_polled_ = 0;
    for (const auto& v : vars_) n_polled_ += !v.push;
  }

  // Brings the device results for the current x up to date, evaluating ONLY what is asked for (the reference's sets do the
  // same: GetValues computes values, FillJacobianBlock derivatives -- time_discretization_constraint.cc:65-96).  Ipopt
  // calls eval_g far more often than eval_jac_g (every line-search trial point), and a values-only evaluation does not
  // move the 8 * nnz bytes of the Jacobian over PCIe.  `want` = TWR_EVAL_VALUES or TWR_EVAL_JACOBIAN: a new x evaluates
  // just that; the first FillJacobianBlock on an x whose values are already there adds the Jacobian alone.
  // `request` identifies who asks (constraint set, values or which block) for the sweep rule above.
  // x is read set by set, BY NAME (resolved in Link), so a host NLP may hold further variable sets of its own, in any
  // position: they are simply not read.
  void Update(const VariablesPtr& vars, int want, size_t request) {
    const auto t0 = std::chrono::steady_clock::now();
    Link(vars);
    bool changed = !have_x_, read = false;
    for (auto& v : vars_)
      if (v.push && v.dirty) {
        changed |= Read(v);
        read = true;
      }
    if (n_polled_ > 0) {
      if (request >= served_.size()) served_.resize(request + 1, 0);
      const bool poll = !have_x_ || polling_ == Polling::kEveryRequest || served_[request] || want != last_want_;
      if (poll) {
        for (auto& v : vars_)
          if (!v.push) changed |= Read(v);
        read = true;
      }
    }
    if (read) std::fill(served_.begin(), served_.end(), 0);
    if (request < served_.size()) served_[request] = 1;
    last_want_ = want;
    have_x_ = true;
    if (changed) have_ = 0;
    const int need = want & ~have_;
    const auto t1 = std::chrono::steady_clock::now();
    read_seconds_ += std::chrono::duration<double>(t1 - t0).count();
    if (!need) return;
    Check(twr_batch_eval_host(batch_, x_, g_, jac_, need));
    eval_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    have_ |= need;
    n_value_evals_ += (need & TWR_EVAL_VALUES) != 0;
    n_jacobian_evals_ += (need & TWR_EVAL_JACOBIAN) != 0;
  }
  // VariableSet::GetValues calls made so far (tests, scripts/latency.py: the host cost of the boundary is this count
  // times what the host's GetValues costs)
  long variable_reads() const { return n_reads_; }
  // wall time spent so far deciding whether x moved + reading it, and inside twr_batch_eval_host (upload, kernels, download)
  double read_seconds() const { return read_seconds_; }
  double eval_seconds() const { return eval_seconds_; }
  // evaluations launched so far (tests; a solver log can show the eval_g : eval_jac_g ratio with them)
  long value_evaluations() const { return n_value_evals_; }
  long jacobian_evaluations() const { return n_jacobian_evals_; }
  const std::vector<twr_set_info>& var_sets() const { return var_sets_; }
  const twr_structure* structure() const { return structure_; }
  const twr_sizes& sizes() const { return sizes_; }
  const double* g() const { return g_; }
  const double* jac() const { return jac_; }
  const std::vector<double>& lower() const { return lower_; }
  const std::vector<double>& upper() const { return upper_; }

 private:
  static void Check(int rc) {
    if (rc != TWR_OK) throw std::runtime_error(std::string("towr_amd: ") + twr_last_error());
  }
  struct VarSet {
    twr_set_info info{};
    const ifopt::Component* comp = nullptr;   // resolved by Link()
    bool push = false;                        // the set reports its changes through MarkDirty
    bool dirty = true;
  };
  // reads one variable set into the page-locked x; true if its values differ from what was there
  bool Read(VarSet& v) {
    const Eigen::VectorXd xv = v.comp->GetValues();
    ++n_reads_;
    v.dirty = false;
    if (static_cast<int>(xv.size()) != v.info.size) throw std::runtime_error(std::string("towr_amd: variable set '") + v.info.name + "' changed its size");
    if (have_x_ && std::memcmp(xv.data(), x_ + v.info.offset, sizeof(double) * v.info.size) == 0) return false;
    std::memcpy(x_ + v.info.offset, xv.data(), sizeof(double) * v.info.size);
    return true;
  }
  twr_structure* structure_ = nullptr;
  twr_batch* batch_ = nullptr;
  twr_sizes sizes_{};
  double *x_ = nullptr, *g_ = nullptr, *jac_ = nullptr;
  std::vector<double> lower_, upper_;
  std::vector<twr_set_info> var_sets_;
  std::vector<VarSet> vars_;
  const void* linked_ = nullptr;        // the variable composite vars_[].comp were resolved in
  LinkHook link_hook_;
  std::vector<std::shared_ptr<void>> keep_;
  std::vector<char> served_;            // requests answered since x was last read (polled sets only)
  int n_polled_ = 0;                    // variable sets that do not push
  Polling polling_ = Polling::kPerSweep;
  int last_want_ = 0;                   // kind of the previous request (TWR_EVAL_VALUES / TWR_EVAL_JACOBIAN)
  bool have_x_ = false;
  int have_ = 0;   // TWR_EVAL_* bits that are valid for the x in x_
  long n_value_evals_ = 0, n_jacobian_evals_ = 0, n_reads_ = 0;
  double read_seconds_ = 0.0, eval_seconds_ = 0.0;
};

class DeviceConstraintSet : public ifopt::ConstraintSet {
 public:
  DeviceConstraintSet(std::shared_ptr<DeviceProblem> problem, int set_index)
      : ifopt::ConstraintSet(kSpecifyLater, SetName(*problem, set_index)), problem_(std::move(problem)), set_index_(set_index) {
    if (twr_structure_con_set(problem_->structure(), set_index, &info_) != TWR_OK)
      throw std::runtime_error(twr_last_error());
    SetRows(info_.size);
    var_sets_ = problem_->var_sets();
    // Where the values of (this set's row r) x (variable set v) sit in the CSR value array: the columns of a
    // row ascend and a variable set is a contiguous column range, so it is one sub-range per row, found once
    // here instead of re-walking the whole pattern on every FillJacobianBlock call.
    const int32_t* row_ptr = twr_structure_row_ptr(problem_->structure());
    const int32_t* col_idx = twr_structure_col_idx(problem_->structure());
    ranges_.assign(var_sets_.size(), std::vector<int32_t>(2 * (size_t)info_.size, 0));
    for (int r = 0; r < info_.size; ++r) {
      int k = row_ptr[info_.offset + r];
      const int end = row_ptr[info_.offset + r + 1];
      for (size_t v = 0; v < var_sets_.size(); ++v) {
        ranges_[v][2 * r] = k;
        while (k < end && col_idx[k] < var_sets_[v].offset + var_sets_[v].size) ++k;
        ranges_[v][2 * r + 1] = k;
      }
    }
  }

  const DeviceProblem& problem() const { return *problem_; }
  DeviceProblem& problem() { return *problem_; }

  // ifopt::ConstraintSet::LinkWithVariables -> here (the reference's sets look their variable sets up at this point too:
  // force_constraint.cc:50-54, terrain_constraint.cc:44-47)
  void InitVariableDependedQuantities(const VariablesPtr& x) override { problem_->Link(x); }

  VectorXd GetValues() const override {
    problem_->Update(GetVariables(), TWR_EVAL_VALUES, Request(0));
    return Eigen::Map<const VectorXd>(problem_->g() + info_.offset, info_.size);
  }

  VecBound GetBounds() const override {
    VecBound b(info_.size);
    for (int i = 0; i < info_.size; ++i)
      b[i] = ifopt::Bounds(problem_->lower()[info_.offset + i], problem_->upper()[info_.offset + i]);
    return b;
  }

  void FillJacobianBlock(std::string var_set, Jacobian& jac) const override {
    // ifopt / towr convention: a constraint leaves the block of a variable set it does not depend on empty
    // (time_discretization_constraint.cc:89-96 and every FillJacobianBlock of the reference test the name and fall
    // through).  A host NLP may therefore hold variable sets of its own beside towr's: their blocks stay empty.  What
    // must NOT pass silently is a towr-style name the structure does not know (a typo, or "ee-schedule<ee>" sets without
    // TWR_SET_TOTAL_TIME): a zero block there would hide real derivatives from Ipopt, so that throws.
    size_t v = 0;
    while (v < var_sets_.size() && var_set != var_sets_[v].name) ++v;
    if (v == var_sets_.size()) {
      if (LooksLikeTowrSet(var_set))
        throw std::runtime_error("towr_amd: constraint set '" + std::string(info_.name) + "' asked for the Jacobian w.r.t. unknown variable set '" +
                                 var_set + "'");
      return;
    }
    problem_->Update(GetVariables(), TWR_EVAL_JACOBIAN, Request(1 + v));
    const int32_t* col_idx = twr_structure_col_idx(problem_->structure());
    const double* val = problem_->jac();
    const int32_t* range = ranges_[v].data();
    for (int r = 0; r < info_.size; ++r)
      for (int k = range[2 * r]; k < range[2 * r + 1]; ++k)
        jac.coeffRef(r, col_idx[k] - var_sets_[v].offset) = val[k];  // explicit zeros kept, like the reference
  }

 private:
  // towr's variable-set names (variable_names.h:47-54: "base-lin", "base-ang", "ee-motion_<ee>", "ee-force_<ee>",
  // "ee-schedule<ee>"); anything that starts like one of them is meant to be one
  static bool LooksLikeTowrSet(const std::string& name) {
    for (const char* prefix : {"base-", "ee-"})
      if (name.compare(0, std::strlen(prefix), prefix) == 0) return true;
    return false;
  }
  static std::string SetName(const DeviceProblem& p, int i) {
    twr_set_info s;
    if (twr_structure_con_set(p.structure(), i, &s) != TWR_OK) throw std::runtime_error(twr_last_error());
    return s.name;
  }
  // one id per thing ifopt asks of this set: its values, or its block w.r.t. variable set v
  size_t Request(size_t what) const { return static_cast<size_t>(set_index_) * (var_sets_.size() + 1) + what; }
  std::shared_ptr<DeviceProblem> problem_;
  int set_index_ = 0;
  twr_set_info info_{};
  std::vector<twr_set_info> var_sets_;
  std::vector<std::vector<int32_t>> ranges_;  // [variable set][2 * row + {begin, end}] into the CSR value array
};

// All device sets of one problem (twr_params.constraint_sets), in the reference's relative order.  `problem_out`: the
// shared DeviceProblem behind them (to install a link hook / call MarkDirty / read the counters).
inline std::vector<ifopt::ConstraintSet::Ptr> MakeDeviceConstraints(const twr_model& model, const twr_schedule& schedule,
                                                                    const twr_params& params, int device = 0,
                                                                    const twr_terrain_grid* grid = nullptr,
                                                                    std::shared_ptr<DeviceProblem>* problem_out = nullptr) {
  auto problem = std::make_shared<DeviceProblem>(model, schedule, params, device, grid);
  if (problem_out) *problem_out = problem;
  std::vector<ifopt::ConstraintSet::Ptr> sets;
  for (int i = 0; i < problem->sizes().n_con_sets; ++i) sets.push_back(std::make_shared<DeviceConstraintSet>(problem, i));
  return sets;
}

}  // namespace towr_amd
#endif  // __has_include(<ifopt/constraint_set.h>)
