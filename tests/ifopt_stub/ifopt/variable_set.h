// TEST INFRASTRUCTURE ONLY -- stand-in for <ifopt/variable_set.h>.
#pragma once
#include "composite.h"
namespace ifopt {
class VariableSet : public Component {
 public:
  VariableSet(int n_var, const std::string& name) : Component(n_var, name) {}
  Jacobian GetJacobian() const final { throw std::runtime_error("not implemented for variables"); }
};
}  // namespace ifopt
