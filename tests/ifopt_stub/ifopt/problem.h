// TEST INFRASTRUCTURE ONLY -- stand-in for <ifopt/problem.h>: the two entry points Ipopt's adapter calls.
#pragma once
#include "constraint_set.h"
#include "variable_set.h"
namespace ifopt {
class Problem {
 public:
  using VectorXd = Component::VectorXd;
  using Jacobian = Component::Jacobian;
  using VecBound = Component::VecBound;
  Problem() : variables_(std::make_shared<Composite>("variable-sets", false)), constraints_("constraint-sets", false) {}
  void AddVariableSet(VariableSet::Ptr s) { variables_->AddComponent(s); }
  void AddConstraintSet(ConstraintSet::Ptr s) {
    s->LinkWithVariables(variables_);
    constraints_.AddComponent(s);
  }
  int GetNumberOfOptimizationVariables() const { return variables_->GetRows(); }
  int GetNumberOfConstraints() const { return constraints_.GetRows(); }
  VecBound GetBoundsOnConstraints() const { return constraints_.GetBounds(); }
  void SetVariables(const double* x) {
    VectorXd v(GetNumberOfOptimizationVariables());
    for (int i = 0; i < GetNumberOfOptimizationVariables(); ++i) v[i] = x[i];
    variables_->SetVariables(v);
  }
  VectorXd EvaluateConstraints(const double* x) {
    SetVariables(x);
    return constraints_.GetValues();
  }
  Jacobian GetJacobianOfConstraints() const { return constraints_.GetJacobian(); }
  void EvalNonzerosOfJacobian(const double* x, double* values) {
    SetVariables(x);
    Jacobian jac = GetJacobianOfConstraints();
    jac.makeCompressed();
    for (int i = 0; i < static_cast<int>(jac.nonZeros()); ++i) values[i] = jac.valuePtr()[i];
  }
  Composite::Ptr GetOptVariables() const { return variables_; }

 private:
  Composite::Ptr variables_;
  Composite constraints_;
};
}  // namespace ifopt
