// towr-side binding: turns the objects a towr::NlpFormulation holds into the PODs of include/towr_amd.h and returns the
// device constraint sets in place of the reference's Eigen ones -- the code a maintainer includes, not a sketch.
//
// STATUS: EXPERIMENTAL until `ref_dump --binding` has run on a box with Eigen3 + ifopt.
//
//   #include <towr_amd/csrc/towr_binding.h>
//   ...
//   for (auto c : formulation.GetVariableSets(solution)) nlp.AddVariableSet(c);
//   for (auto c : towr_amd::MakeDeviceConstraints(formulation)) nlp.AddConstraintSet(c);   // was: formulation.GetConstraints(solution)
//
// (towr/test/hopper_example.cc:72-77, fpowr/src/footstep_plan_server.cc:213-218: the only two call sites of
// NlpFormulation::GetConstraints in the reference.)  Nothing in towr changes; CPU Ipopt keeps driving the solve.
//
// COMPILE-GATED on <towr/nlp_formulation.h> and <ifopt/constraint_set.h> (this build image has neither, nor Eigen).  It is
// compiled and RUN by oracle/ref_dump (ref_dump --binding: the REAL reference's constraint sets against these device
// sets on the same NlpFormulation) on any box that has Eigen3 + ifopt.  In this image it is TYPE-CHECKED only
// (tests/test_binding_syntax.py: g++ -fsyntax-only against the real towr headers with type-level stand-ins for the
// Eigen / ifopt names they mention): every member, method and enum name of the reference used below exists with the
// shape assumed; what the code computes is unverified until ref_dump --binding runs somewhere.
//
// What is read from where (all through PUBLIC members / virtual interfaces of the reference, no accessor is added to towr):
//   robot      RobotModel::kinematic_model_ -> GetNominalStanceInBase / GetMaximumDeviationFromNominal
//              (kinematic_model.h:70-85); RobotModel::dynamic_model_ -> m(), g() (dynamic_model.h:140-150); the body
//              inertia is PRIVATE in SingleRigidBodyDynamics (single_rigid_body_dynamics.h:91-95), so it is PROBED through
//              the public interface: with R = I, omega = 0, no forces, GetDynamicViolation() returns I_b * omega_dot in its
//              angular rows (single_rigid_body_dynamics.cc:76-101) -- three evaluations give the three columns.  The same
//              probe refuses a DynamicModel subclass that is not a single rigid body.
//   schedule   Parameters::ee_phase_durations_ / ee_in_contact_at_start_ (parameters.h:168-171)
//   params     Parameters::dt_constraint_* / duration_base_polynomial_ / *_polynomials_per_*_phase_ /
//              force_limit_in_normal_direction_ / constraints_ (parameters.h:174-198)
//   terrain    HeightMap has no identifier to read back (height_map.h:70-137 only maps id -> object) and user subclasses
//              are its documented extension point (:60-64): the binding asks the object what it is (dynamic_cast ladder
//              over height_map_examples.h:45-166) and REFUSES what it cannot represent -- a silent flat-ground fallback
//              would hand Ipopt another problem.  The two gridded maps keep their cells private (grid_height_map.h:18,
//              height_map_from_csv.h:111), so they are rebuilt from what they were built FROM: the grid_map::GridMap /
//              PlanarTerrain message (GridTerrain overloads below) or the CSV file (CsvTerrain).
#pragma once
#if __has_include(<towr/nlp_formulation.h>) && __has_include(<ifopt/constraint_set.h>)
#include <towr/models/single_rigid_body_dynamics.h>
#include <towr/nlp_formulation.h>
#include <towr/terrain/examples/height_map_examples.h>
#include <towr/variables/nodes_observer.h>
#include <towr/variables/nodes_variables.h>
#include <towr/variables/phase_durations.h>
#include <towr/variables/phase_durations_observer.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ifopt_adapter.h"
#include "towr_amd.h"

#if __has_include(<towr/terrain/rapidcsv.h>)
#include <towr/terrain/rapidcsv.h>
#define TOWR_AMD_HAVE_RAPIDCSV 1
#endif
#if __has_include(<grid_map_core/grid_map_core.hpp>)
#include <grid_map_core/grid_map_core.hpp>
#define TOWR_AMD_HAVE_GRID_MAP 1
#endif

namespace towr_amd {

// A terrain as the device path sees it: the id for twr_model.terrain_id, FlatGround's height, and -- for the two gridded
// maps -- the cell data (shared: it must outlive every structure built with it; DeviceProblem keeps a reference).
struct DeviceTerrain {
  int id = TWR_TERRAIN_FLAT;
  double flat_height = 0.0;
  std::shared_ptr<twr_terrain_grid> grid;   // nullptr for the analytic terrains
};

inline std::shared_ptr<twr_terrain_grid> OwnGrid(twr_terrain_grid* g) {
  return std::shared_ptr<twr_terrain_grid>(g, [](twr_terrain_grid* p) { twr_terrain_grid_destroy(p); });
}

// The seven analytic maps carry compile-time constants only and are restated on the device (kernels.hip terrain_eval).
// Throws for every other subclass, including the gridded ones (use GridTerrain / CsvTerrain for those).
inline DeviceTerrain ToTwrTerrain(const towr::HeightMap& t) {
  DeviceTerrain d;
  // the device path carries height_map.h:136's friction coefficient as a model constant (twr_model.friction, set to 0.5
  // in ToTwrModel); a terrain that says otherwise -- whichever subclass -- would be another problem
  if (t.GetFrictionCoeff() != 0.5) throw std::runtime_error("towr_amd: friction coefficient other than height_map.h:136's 0.5");
  if (dynamic_cast<const towr::FlatGround*>(&t)) {
    d.id = TWR_TERRAIN_FLAT;
    d.flat_height = t.GetHeight(0.0, 0.0);   // FlatGround(height), height_map_examples.h:45-52
    return d;
  }
  if (dynamic_cast<const towr::Block*>(&t)) d.id = TWR_TERRAIN_BLOCK;
  else if (dynamic_cast<const towr::Stairs*>(&t)) d.id = TWR_TERRAIN_STAIRS;
  else if (dynamic_cast<const towr::Gap*>(&t)) d.id = TWR_TERRAIN_GAP;
  else if (dynamic_cast<const towr::Slope*>(&t)) d.id = TWR_TERRAIN_SLOPE;
  else if (dynamic_cast<const towr::ChimneyLR*>(&t)) d.id = TWR_TERRAIN_CHIMNEY_LR;
  else if (dynamic_cast<const towr::Chimney*>(&t)) d.id = TWR_TERRAIN_CHIMNEY;
  else
    throw std::runtime_error(
        "towr_amd: this HeightMap subclass has no device counterpart.  Gridded maps: pass GridTerrain(map) / "
        "CsvTerrain(path) to MakeDeviceConstraints; anything else: keep the CPU constraint sets for it, or add its "
        "height function to terrain_eval in kernels.hip");
  return d;
}

#ifdef TOWR_AMD_HAVE_GRID_MAP
// The `Grid` height map fpowr runs on (towr/include/towr/terrain/grid_height_map.h:15-60, built at
// fpowr/src/footstep_plan_server.cc:155 from args->terrain): pass the same grid_map::GridMap the Grid was built from
// (grid_map::GridMapRosConverter::fromMessage(args->terrain.gridmap, map)).  The float "elevation" layer, resolution and
// position go to the device as they are; the ABI assumes start index (0, 0), so a moved map is normalised first.
inline DeviceTerrain GridTerrain(grid_map::GridMap map) {
  map.convertToDefaultStartIndex();
  const grid_map::Matrix& e = map["elevation"];   // Eigen::MatrixXf, column-major [size_x][size_y]
  twr_terrain_grid* g = nullptr;
  if (twr_terrain_grid_map_create(e.data(), map.getSize()(0), map.getSize()(1), map.getResolution(), map.getPosition().x(),
                                  map.getPosition().y(), &g) != TWR_OK)
    throw std::runtime_error(std::string("towr_amd: ") + twr_last_error());
  DeviceTerrain d;
  d.id = TWR_TERRAIN_GRID_MAP;
  d.grid = OwnGrid(g);
  return d;
}
#endif

#ifdef TOWR_AMD_HAVE_RAPIDCSV
// HeightMapFromCSV (towr/include/towr/terrain/height_map_from_csv.h:13-117): the file is read exactly as its constructor
// reads it (:16-27: no header row / column, grid_(i, j) = cell(column j, row i)); the ABI takes heights[y_cell][x_cell].
inline DeviceTerrain CsvTerrain(const std::string& file_path) {
  rapidcsv::Document doc(file_path, rapidcsv::LabelParams(-1, -1));
  const int rows = static_cast<int>(doc.GetRowCount()), cols = static_cast<int>(doc.GetColumnCount());
  std::vector<double> h(static_cast<size_t>(rows) * cols);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) h[static_cast<size_t>(i) * cols + j] = doc.GetCell<double>(j, i);
  twr_terrain_grid* g = nullptr;
  if (twr_terrain_grid_create(h.data(), rows, cols, &g) != TWR_OK) throw std::runtime_error(std::string("towr_amd: ") + twr_last_error());
  DeviceTerrain d;
  d.id = TWR_TERRAIN_CSV_GRID;
  d.grid = OwnGrid(g);
  return d;
}
#endif

// Robot constants.  terrain_id / flat_height are filled by MakeDeviceConstraints from the DeviceTerrain.
inline twr_model ToTwrModel(const towr::RobotModel& robot, const towr::Parameters& params) {
  if (!robot.kinematic_model_ || !robot.dynamic_model_) throw std::runtime_error("towr_amd: RobotModel without kinematic / dynamic model");
  twr_model m;
  std::memset(&m, 0, sizeof(m));
  const auto stance = robot.kinematic_model_->GetNominalStanceInBase();
  const Eigen::Vector3d dev = robot.kinematic_model_->GetMaximumDeviationFromNominal();
  m.n_ee = static_cast<int32_t>(stance.size());
  if (m.n_ee < 1 || m.n_ee > TWR_MAX_EE) throw std::runtime_error("towr_amd: 1 .. 4 end-effectors");
  if (robot.dynamic_model_->GetEECount() != m.n_ee) throw std::runtime_error("towr_amd: kinematic and dynamic model disagree on the leg count");
  for (int e = 0; e < m.n_ee; ++e)
    for (int d = 0; d < 3; ++d) m.nominal_stance[e][d] = stance.at(e)(d);
  for (int d = 0; d < 3; ++d) m.max_dev[d] = dev(d);
  m.mass = robot.dynamic_model_->m();
  m.gravity = robot.dynamic_model_->g();
  m.friction = 0.5;   // HeightMap::friction_coeff_, height_map.h:136 (checked against the terrain in ToTwrTerrain)
  m.force_limit = params.force_limit_in_normal_direction_;

  // Body inertia by probing (see the header comment).  The probe works on a COPY of nothing: SetCurrent overwrites the
  // model's current state, which DynamicConstraint::UpdateModel rewrites before every use (dynamic_constraint.cc:119-137).
  towr::DynamicModel& dyn = *robot.dynamic_model_;
  if (!dynamic_cast<const towr::SingleRigidBodyDynamics*>(&dyn))
    throw std::runtime_error("towr_amd: the device path implements SingleRigidBodyDynamics only");
  const towr::DynamicModel::EEPos at_com(m.n_ee, Eigen::Vector3d::Zero());
  const towr::DynamicModel::EELoad no_force(m.n_ee, Eigen::Vector3d::Zero());
  double I[3][3];
  for (int c = 0; c < 3; ++c) {
    Eigen::Vector3d wd = Eigen::Vector3d::Zero();
    wd(c) = 1.0;
    dyn.SetCurrent(Eigen::Vector3d::Zero(), Eigen::Vector3d::Zero(), Eigen::Matrix3d::Identity(), Eigen::Vector3d::Zero(), wd, no_force, at_com);
    const auto v = dyn.GetDynamicViolation();   // rows AX, AY, AZ = I_b * e_c; rows LX..LZ = (0, 0, m g)
    for (int r = 0; r < 3; ++r) I[r][c] = v(r);
    if (std::fabs(v(3)) > 1e-12 || std::fabs(v(4)) > 1e-12 || std::fabs(v(5) - m.mass * m.gravity) > 1e-9 * m.mass * m.gravity)
      throw std::runtime_error("towr_amd: dynamic model does not behave like a single rigid body");
  }
  for (int r = 0; r < 3; ++r)
    for (int c = r + 1; c < 3; ++c)
      if (std::fabs(I[r][c] - I[c][r]) > 1e-12 * (std::fabs(I[0][0]) + std::fabs(I[1][1]) + std::fabs(I[2][2])))
        throw std::runtime_error("towr_amd: probed inertia tensor is not symmetric");
  // twr_model.inertia = Ixx, Iyy, Izz, Ixy, Ixz, Iyz AS GIVEN TO SingleRigidBodyDynamics(): the tensor's off-diagonal
  // entries are the NEGATED products of inertia (single_rigid_body_dynamics.cc:36-44)
  m.inertia[0] = I[0][0];
  m.inertia[1] = I[1][1];
  m.inertia[2] = I[2][2];
  m.inertia[3] = -I[0][1];
  m.inertia[4] = -I[0][2];
  m.inertia[5] = -I[1][2];
  return m;
}

inline twr_schedule ToTwrSchedule(const towr::Parameters& params) {
  twr_schedule s;
  std::memset(&s, 0, sizeof(s));
  s.n_ee = static_cast<int32_t>(params.ee_phase_durations_.size());
  if (s.n_ee < 1 || s.n_ee > TWR_MAX_EE || params.ee_in_contact_at_start_.size() != params.ee_phase_durations_.size())
    throw std::runtime_error("towr_amd: 1 .. 4 end-effectors, one contact flag each");
  for (int e = 0; e < s.n_ee; ++e) {
    const auto& d = params.ee_phase_durations_.at(e);
    if (d.empty() || d.size() > TWR_MAX_PHASES) throw std::runtime_error("towr_amd: 1 .. 32 phases per foot");
    s.n_phases[e] = static_cast<int32_t>(d.size());
    s.in_contact_at_start[e] = params.ee_in_contact_at_start_.at(e) ? 1 : 0;
    for (size_t i = 0; i < d.size(); ++i) s.phase_durations[e][i] = d[i];
  }
  return s;
}

// Discretisation + which of Parameters::constraints_ the device path takes over (all of the factory's names,
// nlp_formulation.cc:214-226).  base_z_init is what BaseMotionConstraint reads from the spline at construction
// (base_motion_constraint.cc:51-55): the initial base height.
inline twr_params ToTwrParams(const towr::Parameters& params, const towr::BaseState& initial_base) {
  twr_params p;
  if (twr_params_default(&p) != TWR_OK) throw std::runtime_error(std::string("towr_amd: ") + twr_last_error());
  p.dt_dynamic = params.dt_constraint_dynamic_;
  p.dt_rom = params.dt_constraint_range_of_motion_;
  p.duration_base_poly = params.duration_base_polynomial_;
  p.polys_per_swing = params.ee_polynomials_per_swing_phase_;
  p.polys_per_stance_force = params.force_polynomials_per_stance_phase_;
  p.constraint_sets = 0;
  for (auto name : params.constraints_) {
    switch (name) {
      case towr::Parameters::Terrain:        p.constraint_sets |= TWR_SET_TERRAIN; break;
      case towr::Parameters::Dynamic:        p.constraint_sets |= TWR_SET_DYNAMIC; break;
      case towr::Parameters::BaseAcc:        p.constraint_sets |= TWR_SET_BASE_ACC; break;
      case towr::Parameters::EndeffectorRom: p.constraint_sets |= TWR_SET_ROM; break;
      case towr::Parameters::Force:          p.constraint_sets |= TWR_SET_FORCE; break;
      case towr::Parameters::Swing:          p.constraint_sets |= TWR_SET_SWING; break;
      case towr::Parameters::TotalTime:      p.constraint_sets |= TWR_SET_TOTAL_TIME; break;
      case towr::Parameters::BaseRom:
        p.constraint_sets |= TWR_SET_BASE_ROM;
        p.dt_base_motion = params.dt_constraint_base_motion_;
        p.base_z_init = initial_base.lin.p()(2);   // z
        break;
      default: throw std::runtime_error("constraint not defined!");   // as nlp_formulation.cc:224
    }
  }
  return p;
}

// First component-name prefix of every Parameters::ConstraintName (the names the reference's constructors pass to
// ifopt: terrain_constraint.cc:38, dynamic_constraint.cc:50, spline_acc_constraint.cc:40, range_of_motion_constraint.cc:43,
// force_constraint.cc:40, swing_constraint.cc:37, total_duration_constraint.cc:38, base_motion_constraint.cc:43)
inline const char* ComponentPrefix(towr::Parameters::ConstraintName name) {
  switch (name) {
    case towr::Parameters::Terrain:        return "terrain-";
    case towr::Parameters::Dynamic:        return "dynamic";
    case towr::Parameters::BaseAcc:        return "splineacc-";
    case towr::Parameters::EndeffectorRom: return "rangeofmotion-";
    case towr::Parameters::Force:          return "force-";
    case towr::Parameters::Swing:          return "swing-";
    case towr::Parameters::TotalTime:      return "totalduration-";
    case towr::Parameters::BaseRom:        return "baseMotion";
    default: throw std::runtime_error("constraint not defined!");
  }
}

// "x changed" as a PUSH, the reference's own mechanism: NodesVariables::SetVariables -> UpdateObservers ->
// NodesObserver::UpdateNodes (nodes_variables.cc:64-79, nodes_observer.h:52-68; the reference's NodeSpline is such an
// observer, node_spline.cc:45-54), PhaseDurations::SetVariables -> UpdatePolynomialDurations (phase_durations.cc:77-103,
// phase_durations_observer.h:52).  One observer per variable set flips that set's flag in the DeviceProblem; the next
// GetValues / FillJacobianBlock reads exactly the flagged sets -- once per Problem::SetVariables, never per call.
// The base-class constructors register `this` with the subject (nodes_observer.cc:36-42, phase_durations_observer.cc:37-43);
// subjects hold RAW observer pointers and never drop them, so -- exactly like the reference's splines -- the device sets
// (which own these observers through DeviceProblem::KeepAlive) must live as long as anybody calls SetVariables on the
// variable sets they were linked with.  ifopt::Problem holds both and destroys them together.
class NodesDirtyObserver final : public towr::NodesObserver {
 public:
  NodesDirtyObserver(towr::NodesVariables* subject, DeviceProblem* problem, int var_set)
      : towr::NodesObserver(subject), problem_(problem), var_set_(var_set) {}
  void UpdateNodes() override { problem_->MarkDirty(var_set_); }

 private:
  DeviceProblem* problem_;
  int var_set_;
};

class DurationsDirtyObserver final : public towr::PhaseDurationsObserver {
 public:
  DurationsDirtyObserver(towr::PhaseDurations* subject, DeviceProblem* problem, int var_set)
      : towr::PhaseDurationsObserver(subject), problem_(problem), var_set_(var_set) {}
  void UpdatePolynomialDurations() override { problem_->MarkDirty(var_set_); }

 private:
  DeviceProblem* problem_;
  int var_set_;
};

// The link hook (DeviceProblem::SetLinkHook): runs once, when the first device set is linked with the variable composite
// -- where the reference's sets take their x->GetComponent<NodesVariablesPhaseBased>(name) (force_constraint.cc:50-54).
// A variable set of another type (a host's own) stays polled.
inline void RegisterTowrObservers(DeviceProblem& problem, const std::vector<ifopt::Component::Ptr>& sets) {
  for (size_t i = 0; i < sets.size(); ++i) {
    if (auto nodes = std::dynamic_pointer_cast<towr::NodesVariables>(sets[i])) {
      problem.KeepAlive(std::make_shared<NodesDirtyObserver>(nodes.get(), &problem, static_cast<int>(i)));
      problem.EnablePush(static_cast<int>(i));
    } else if (auto durations = std::dynamic_pointer_cast<towr::PhaseDurations>(sets[i])) {
      problem.KeepAlive(std::make_shared<DurationsDirtyObserver>(durations.get(), &problem, static_cast<int>(i)));
      problem.EnablePush(static_cast<int>(i));
    }
  }
}

// NlpFormulation::GetConstraints (nlp_formulation.cc:200-209) on the device: the sets of every name in
// params_.constraints_, in the caller's order of names, per name in the reference's creation order (per end-effector).
// `terrain`: leave empty for the analytic maps (formulation.terrain_ is asked what it is); pass GridTerrain(map) /
// CsvTerrain(path) when formulation.terrain_ is a Grid / HeightMapFromCSV.
inline towr::NlpFormulation::ContraintPtrVec MakeDeviceConstraints(const towr::NlpFormulation& f, int device = 0,
                                                                   const DeviceTerrain* terrain = nullptr) {
  if (!f.terrain_) throw std::runtime_error("towr_amd: NlpFormulation::terrain_ is not set");
  const DeviceTerrain t = terrain ? *terrain : ToTwrTerrain(*f.terrain_);
  twr_model model = ToTwrModel(f.model_, f.params_);
  model.terrain_id = t.id;
  model.flat_height = t.flat_height;
  const twr_schedule sched = ToTwrSchedule(f.params_);
  if (sched.n_ee != model.n_ee) throw std::runtime_error("towr_amd: schedule and robot disagree on the leg count");
  const twr_params prm = ToTwrParams(f.params_, f.initial_base_);
  // (the batch uploads its own copy of a gridded terrain's cells, so `t.grid` may go once the sets exist)
  std::shared_ptr<DeviceProblem> problem;
  const std::vector<ifopt::ConstraintSet::Ptr> all = MakeDeviceConstraints(model, sched, prm, device, t.grid.get(), &problem);
  problem->SetLinkHook(RegisterTowrObservers);
  towr::NlpFormulation::ContraintPtrVec out;
  for (auto name : f.params_.constraints_) {
    const std::string prefix = ComponentPrefix(name);
    for (const auto& c : all)
      if (c->GetName().compare(0, prefix.size(), prefix) == 0) out.push_back(c);
  }
  if (out.size() != all.size()) throw std::runtime_error("towr_amd: a device set matches no entry of Parameters::constraints_");
  return out;
}

}  // namespace towr_amd
#endif  // towr + ifopt headers present
