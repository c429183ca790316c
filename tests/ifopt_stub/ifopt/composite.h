// TEST INFRASTRUCTURE ONLY -- stand-in for <ifopt/composite.h>: Component and Composite with the stacking
// behaviour SURVEY.md App. C records (values / bounds / Jacobians of constraint composites are row-stacked in
// insertion order; a variable composite hands every set its segment of x).
#pragma once
#include <Eigen/Dense>
#include <Eigen/Sparse>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bounds.h"

namespace ifopt {

class Component {
 public:
  using Ptr = std::shared_ptr<Component>;
  using Jacobian = Eigen::SparseMatrix<double, Eigen::RowMajor>;
  using VectorXd = Eigen::VectorXd;
  using VecBound = std::vector<Bounds>;
  static const int kSpecifyLater = -1;

  Component(int num_rows, const std::string& name) : num_rows_(num_rows), name_(name) {}
  virtual ~Component() = default;
  virtual VectorXd GetValues() const = 0;
  virtual VecBound GetBounds() const = 0;
  virtual void SetVariables(const VectorXd& x) = 0;
  virtual Jacobian GetJacobian() const = 0;
  int GetRows() const { return num_rows_; }
  std::string GetName() const { return name_; }
  void SetRows(int num_rows) { num_rows_ = num_rows; }

 private:
  int num_rows_ = kSpecifyLater;
  std::string name_;
};

class Composite : public Component {
 public:
  using Ptr = std::shared_ptr<Composite>;
  using ComponentVec = std::vector<Component::Ptr>;

  Composite(const std::string& name, bool is_cost) : Component(0, name), is_cost_(is_cost) {}
  void AddComponent(const Component::Ptr& c) {
    components_.push_back(c);
    SetRows(is_cost_ ? 1 : GetRows() + c->GetRows());
  }
  const Component::Ptr GetComponent(std::string name) const {
    for (const auto& c : components_)
      if (c->GetName() == name) return c;
    throw std::runtime_error("component " + name + " does not exist");
  }
  template <typename T>
  std::shared_ptr<T> GetComponent(const std::string& name) const {
    auto t = std::dynamic_pointer_cast<T>(GetComponent(name));
    if (!t) throw std::runtime_error("component " + name + " has another type");
    return t;
  }
  const ComponentVec GetComponents() const { return components_; }
  void ClearComponents() { components_.clear(); SetRows(0); }

  VectorXd GetValues() const override {
    VectorXd g(GetRows());
    int row = 0;
    for (const auto& c : components_) {
      g.set_segment(is_cost_ ? 0 : row, c->GetValues());
      row += c->GetRows();
    }
    return g;
  }
  VecBound GetBounds() const override {
    VecBound b;
    for (const auto& c : components_) {
      VecBound cb = c->GetBounds();
      b.insert(b.end(), cb.begin(), cb.end());
    }
    return b;
  }
  void SetVariables(const VectorXd& x) override {
    int row = 0;
    for (auto& c : components_) {
      c->SetVariables(x.segment(row, c->GetRows()));
      row += c->GetRows();
    }
  }
  Jacobian GetJacobian() const override {
    if (components_.empty()) return Jacobian(0, 0);
    int n_var = static_cast<int>(components_.front()->GetJacobian().cols());
    Jacobian jac(GetRows(), n_var);
    int row = 0;
    for (const auto& c : components_) {
      const Jacobian j = c->GetJacobian();
      for (int r = 0; r < static_cast<int>(j.rows()); ++r)
        for (const auto& kv : j.row(r)) jac.coeffRef(row + r, kv.first) = kv.second;
      row += c->GetRows();
    }
    return jac;
  }

 private:
  ComponentVec components_;
  bool is_cost_;
};

}  // namespace ifopt
