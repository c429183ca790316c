// TEST INFRASTRUCTURE ONLY -- see ../Eigen/Dense: the ifopt::CostTerm name towr's headers mention.
#pragma once
#include <ifopt/constraint_set.h>
namespace ifopt {
class CostTerm : public ConstraintSet {
 public:
  using Ptr = std::shared_ptr<CostTerm>;
  CostTerm(const std::string& name) : ConstraintSet(1, name) {}
  virtual double GetCost() const = 0;
};
}  // namespace ifopt
