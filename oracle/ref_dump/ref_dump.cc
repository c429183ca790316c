// TEST INFRASTRUCTURE ONLY -- driver for the REAL reference (see CMakeLists.txt next to this file): builds the NLP the
// way towr/test/hopper_example.cc:45-90 and fpowr/src/footstep_plan_server.cc:147-220 do, sets the variables to a given
// x and dumps what Ipopt would be handed: g = Problem::EvaluateConstraints(x) and the Jacobian triplets of
// Problem::GetJacobianOfConstraints() (explicit zeros included).  tests/test_ref_dump.py compares the dump with the
// oracle when the executable exists.
//   ref_dump <robot id> <terrain id> <gait combo> <T> <constraint mask (TWR_SET_* bits)> <x file (one double per line, or
//            "guess")> <goal x> <out prefix>
#include <ifopt/problem.h>
#include <towr/initialization/gait_generator.h>
#include <towr/nlp_formulation.h>
#include <towr/terrain/examples/height_map_examples.h>

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 9) {
    std::fprintf(stderr, "usage: see the header of ref_dump.cc\n");
    return 2;
  }
  const int robot = std::atoi(argv[1]), terrain = std::atoi(argv[2]), combo = std::atoi(argv[3]);
  const double T = std::atof(argv[4]);
  const int mask = std::atoi(argv[5]);
  const std::string xfile = argv[6], prefix = argv[8];
  const double goal_x = std::atof(argv[7]);

  towr::NlpFormulation f;
  f.terrain_ = towr::HeightMap::MakeTerrain(static_cast<towr::HeightMap::TerrainID>(terrain));
  f.model_ = towr::RobotModel(static_cast<towr::RobotModel::Robot>(robot));
  const auto nominal = f.model_.kinematic_model_->GetNominalStanceInBase();
  const int n_ee = static_cast<int>(nominal.size());
  f.initial_ee_W_ = nominal;
  for (auto& p : f.initial_ee_W_) p.z() = 0.0;
  f.initial_base_.lin.at(towr::kPos).z() = -nominal.front().z();
  f.final_base_.lin.at(towr::kPos) << goal_x, 0.0, -nominal.front().z();
  auto gait = towr::GaitGenerator::MakeGaitGenerator(n_ee);
  gait->SetCombo(static_cast<towr::GaitGenerator::Combos>(combo));
  for (int ee = 0; ee < n_ee; ++ee) {
    f.params_.ee_phase_durations_.push_back(gait->GetPhaseDurations(T, ee));
    f.params_.ee_in_contact_at_start_.push_back(gait->IsInContactAtStart(ee));
  }
  // constraint list from the mask, in the reference's enum order (parameters.h:139-147); bit 6 = OptimizePhaseDurations
  f.params_.constraints_.clear();
  using P = towr::Parameters;
  const P::ConstraintName order[] = {P::Terrain, P::Dynamic, P::BaseAcc, P::EndeffectorRom, P::Force, P::Swing};
  for (int b = 0; b < 6; ++b)
    if (mask & (1 << b)) f.params_.constraints_.push_back(order[b]);
  if (mask & 128) f.params_.constraints_.push_back(P::BaseRom);
  if (mask & 64) f.params_.OptimizePhaseDurations();

  ifopt::Problem nlp;
  towr::SplineHolder solution;
  for (auto c : f.GetVariableSets(solution)) nlp.AddVariableSet(c);
  for (auto c : f.GetConstraints(solution)) nlp.AddConstraintSet(c);

  Eigen::VectorXd x = nlp.GetOptVariables()->GetValues();
  if (xfile != "guess") {
    std::ifstream in(xfile);
    for (int i = 0; i < x.size(); ++i) in >> x[i];
  }
  const Eigen::VectorXd g = nlp.EvaluateConstraints(x.data());
  auto jac = nlp.GetJacobianOfConstraints();
  std::ofstream og(prefix + "_g.txt"), oj(prefix + "_jac.txt"), ox(prefix + "_x.txt");
  og.precision(17);
  oj.precision(17);
  ox.precision(17);
  for (int i = 0; i < x.size(); ++i) ox << x[i] << "\n";
  for (int i = 0; i < g.size(); ++i) og << g[i] << "\n";
  for (int r = 0; r < jac.outerSize(); ++r)
    for (ifopt::Problem::Jacobian::InnerIterator it(jac, r); it; ++it) oj << it.row() << " " << it.col() << " " << it.value() << "\n";
  std::printf("ref_dump: n=%d m=%d nnz=%d\n", (int)x.size(), (int)g.size(), (int)jac.nonZeros());
  return 0;
}
