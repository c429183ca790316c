// TEST INFRASTRUCTURE ONLY -- stand-in for <ifopt/constraint_set.h>: ConstraintSet::GetJacobian asks the derived
// class for one block per variable set and stacks the blocks column-wise, explicit zeros included (App. C).
#pragma once
#include "composite.h"
namespace ifopt {
class ConstraintSet : public Component {
 public:
  using Ptr = std::shared_ptr<ConstraintSet>;
  using VariablesPtr = Composite::Ptr;
  ConstraintSet(int n_constraints, const std::string& name) : Component(n_constraints, name) {}
  void LinkWithVariables(const VariablesPtr& x) {
    variables_ = x;
    InitVariableDependedQuantities(x);
  }
  Jacobian GetJacobian() const final {
    Jacobian jacobian(GetRows(), variables_->GetRows());
    int col = 0;
    for (const auto& vars : variables_->GetComponents()) {
      int n = vars->GetRows();
      Jacobian jac(GetRows(), n);
      FillJacobianBlock(vars->GetName(), jac);
      for (int r = 0; r < GetRows(); ++r)
        for (const auto& kv : jac.row(r)) jacobian.coeffRef(r, col + kv.first) = kv.second;
      col += n;
    }
    return jacobian;
  }
  virtual void FillJacobianBlock(std::string var_set, Jacobian& jac_block) const = 0;
  void SetVariables(const VectorXd&) final {}

 protected:
  const VariablesPtr GetVariables() const { return variables_; }

 private:
  VariablesPtr variables_;
  virtual void InitVariableDependedQuantities(const VariablesPtr&) {}
};
}  // namespace ifopt
