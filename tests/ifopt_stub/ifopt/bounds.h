// TEST INFRASTRUCTURE ONLY -- stand-in for <ifopt/bounds.h> (surface of SURVEY.md App. C).
#pragma once
namespace ifopt {
struct Bounds {
  Bounds(double lower = 0.0, double upper = 0.0) : lower_(lower), upper_(upper) {}
  double lower_, upper_;
  void operator+=(double s) { lower_ += s; upper_ += s; }
  void operator-=(double s) { lower_ -= s; upper_ -= s; }
};
static const double inf = 1.0e20;
static const Bounds NoBound = Bounds(-inf, +inf);
static const Bounds BoundZero = Bounds(0.0, 0.0);
static const Bounds BoundGreaterZero = Bounds(0.0, +inf);
static const Bounds BoundSmallerZero = Bounds(-inf, 0.0);
}  // namespace ifopt
