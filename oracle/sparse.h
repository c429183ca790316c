// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into the product library.
//
// Tiny row-major sparse algebra that reproduces the *structural* semantics the
// reference relies on from Eigen::SparseMatrix<double,RowMajor>:
//   * coeffRef(r,c) inserts an explicit (possibly zero) entry,
//   * scalar*A, A+B, A-B keep / union the stored pattern (zeros are kept),
//   * A*B is the conservative product: an entry exists wherever one stored
//     a(i,k) meets one stored b(k,j), whatever its value,
//   * dense.sparseView()          drops exact zeros,
//     dense.sparseView(1.0,-1.0)  keeps every entry.
// These rules decide which Jacobian entries ifopt hands to Ipopt (explicit
// zeros included), see SURVEY.md App. D quirk 10.
#pragma once
#include <algorithm>
#include <cassert>
#include <utility>
#include <vector>

namespace orc {

struct SpVec {
  int n = 0;
  std::vector<std::pair<int, double>> e;  // sorted by index, unique
  SpVec() = default;
  explicit SpVec(int n_) : n(n_) {}
  double& ref(int i) {
    assert(i >= 0 && i < n);
    auto it = std::lower_bound(e.begin(), e.end(), i,
                               [](const std::pair<int, double>& a, int b) { return a.first < b; });
    if (it == e.end() || it->first != i) it = e.insert(it, {i, 0.0});
    return it->second;
  }
  double get(int i) const {
    auto it = std::lower_bound(e.begin(), e.end(), i,
                               [](const std::pair<int, double>& a, int b) { return a.first < b; });
    return (it == e.end() || it->first != i) ? 0.0 : it->second;
  }
};

// a + sb*b with pattern union (sb is +1 or -1; values are a+b / a-b like Eigen's
// cwise binary ops, not a + (sb*b) -- identical in IEEE arithmetic).
inline SpVec add(const SpVec& a, const SpVec& b, double sb = 1.0) {
  assert(a.n == b.n);
  SpVec r(a.n);
  size_t i = 0, j = 0;
  while (i < a.e.size() || j < b.e.size()) {
    if (j >= b.e.size() || (i < a.e.size() && a.e[i].first < b.e[j].first)) {
      r.e.push_back(a.e[i++]);
    } else if (i >= a.e.size() || b.e[j].first < a.e[i].first) {
      r.e.push_back({b.e[j].first, sb * b.e[j].second});
      ++j;
    } else {
      r.e.push_back({a.e[i].first, a.e[i].second + sb * b.e[j].second});
      ++i; ++j;
    }
  }
  return r;
}
inline SpVec sub(const SpVec& a, const SpVec& b) { return add(a, b, -1.0); }
inline SpVec scale(double s, const SpVec& a) {
  SpVec r = a;
  for (auto& p : r.e) p.second *= s;
  return r;
}
inline SpVec operator+(const SpVec& a, const SpVec& b) { return add(a, b); }
inline SpVec operator-(const SpVec& a, const SpVec& b) { return sub(a, b); }
inline SpVec operator*(double s, const SpVec& a) { return scale(s, a); }
inline SpVec operator*(const SpVec& a, double s) { return scale(s, a); }

struct SpMat {
  int r = 0, c = 0;
  std::vector<SpVec> rows;
  SpMat() = default;
  SpMat(int r_, int c_) : r(r_), c(c_), rows(r_, SpVec(c_)) {}
  double& coeffRef(int i, int j) { return rows.at(i).ref(j); }
  double coeff(int i, int j) const { return rows.at(i).get(j); }
  int nonZeros() const {
    int n = 0;
    for (auto& v : rows) n += (int)v.e.size();
    return n;
  }
};

inline SpMat add(const SpMat& a, const SpMat& b, double sb = 1.0) {
  assert(a.r == b.r && a.c == b.c);
  SpMat m(a.r, a.c);
  for (int i = 0; i < a.r; ++i) m.rows[i] = add(a.rows[i], b.rows[i], sb);
  return m;
}
inline SpMat operator+(const SpMat& a, const SpMat& b) { return add(a, b); }
inline SpMat operator-(const SpMat& a, const SpMat& b) { return add(a, b, -1.0); }
inline SpMat operator*(double s, const SpMat& a) {
  SpMat m = a;
  for (auto& row : m.rows) row = scale(s, row);
  return m;
}
inline SpMat operator-(const SpMat& a) { return -1.0 * a; }

// (1 x m) * (m x n): conservative structural product.
inline SpVec mul(const SpVec& l, const SpMat& B) {
  assert(l.n == B.r);
  SpVec out(B.c);
  for (auto& lk : l.e)
    for (auto& bj : B.rows[lk.first].e) out.ref(bj.first) += lk.second * bj.second;
  return out;
}
inline SpMat mul(const SpMat& A, const SpMat& B) {
  assert(A.c == B.r);
  SpMat m(A.r, B.c);
  for (int i = 0; i < A.r; ++i) m.rows[i] = mul(A.rows[i], B);
  return m;
}
inline SpVec operator*(const SpVec& l, const SpMat& B) { return mul(l, B); }
inline SpMat operator*(const SpMat& A, const SpMat& B) { return mul(A, B); }

// dense 3-vector / 3x3 helpers -------------------------------------------------
struct V3 {
  double v[3] = {0, 0, 0};
  V3() = default;
  V3(double x, double y, double z) : v{x, y, z} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
inline V3 operator+(const V3& a, const V3& b) { return {a(0) + b(0), a(1) + b(1), a(2) + b(2)}; }
inline V3 operator-(const V3& a, const V3& b) { return {a(0) - b(0), a(1) - b(1), a(2) - b(2)}; }
inline V3 operator*(double s, const V3& a) { return {s * a(0), s * a(1), s * a(2)}; }
inline V3 operator/(const V3& a, double s) { return {a(0) / s, a(1) / s, a(2) / s}; }
inline double dot(const V3& a, const V3& b) { return a(0) * b(0) + a(1) * b(1) + a(2) * b(2); }
inline V3 cross(const V3& a, const V3& b) {
  return {a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0)};
}
struct M3 {
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double& operator()(int i, int j) { return m[i][j]; }
  double operator()(int i, int j) const { return m[i][j]; }
  M3 transpose() const {
    M3 t;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) t(i, j) = m[j][i];
    return t;
  }
};
inline V3 operator*(const M3& A, const V3& x) {
  V3 y;
  for (int i = 0; i < 3; ++i) y(i) = A(i, 0) * x(0) + A(i, 1) * x(1) + A(i, 2) * x(2);
  return y;
}
// sparse 3x3 times dense vector (sums only stored entries, in column order)
inline V3 operator*(const SpMat& A, const V3& x) {
  assert(A.r == 3 && A.c == 3);
  V3 y;
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (auto& p : A.rows[i].e) s += p.second * x(p.first);
    y(i) = s;
  }
  return y;
}
// Eigen's dense.sparseView(): drops exact zeros
inline SpMat sparse_view_pruned(const M3& A) {
  SpMat m(3, 3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (A(i, j) != 0.0) m.coeffRef(i, j) = A(i, j);
  return m;
}
// Eigen's dense.sparseView(1.0,-1.0): keeps everything
inline SpMat sparse_view_full(const M3& A) {
  SpMat m(3, 3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m.coeffRef(i, j) = A(i, j);
  return m;
}
inline SpVec sparse_view_full_row(const V3& a) {
  SpVec v(3);
  for (int i = 0; i < 3; ++i) v.ref(i) = a(i);
  return v;
}

}  // namespace orc
