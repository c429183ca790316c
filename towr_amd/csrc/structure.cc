// Host-side structure builder.  Closed-form restatement of the reference's setup-time
// logic; nothing here touches x.  Citations are file:line under /root/reference/towr/.
#include "structure.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <stdexcept>
#include <unordered_map>

namespace twr {

namespace {

// Spline::GetSegmentID + GetLocalTime (src/spline.cc:48-78): eps 1e-10, the previous segment wins
// at junctions, local time by sequential subtraction.  Must be reproduced bit for bit because the
// active polynomial decides which Jacobian columns exist.
TimeNode Locate(double t_global, const std::vector<double>& durations) {
  const double eps = 1e-10;
  double t = 0;
  int id = -1;
  for (size_t i = 0; i < durations.size(); ++i) {
    t += durations[i];
    if (t >= t_global - eps) {
      id = (int)i;
      break;
    }
  }
  if (id < 0) throw std::runtime_error("time grid exceeds spline duration");
  double t_local = t_global;
  for (int i = 0; i < id; ++i) t_local -= durations[i];
  return {id, t_local};
}

// TimeDiscretizationConstraint ctor (src/time_discretization_constraint.cc:37-50)
std::vector<double> TimeGrid(double T, double dt) {
  if (!(dt > 0)) throw std::runtime_error("dt must be positive");
  double t = 0.0;
  std::vector<double> g = {t};
  for (int i = 0; i < std::floor(T / dt); ++i) {
    t += dt;
    g.push_back(t);
  }
  g.push_back(T);
  return g;
}

struct PolyPhase {
  int phase;
  bool constant;
  int n_in_phase;
};

// BuildPolyInfos (src/nodes_variables_phase_based.cc:38-58)
std::vector<PolyPhase> PhasePolys(int n_phases, bool first_constant, int n_changing) {
  std::vector<PolyPhase> v;
  bool c = first_constant;
  for (int ph = 0; ph < n_phases; ++ph) {
    if (c)
      v.push_back({ph, true, 1});
    else
      for (int j = 0; j < n_changing; ++j) v.push_back({ph, false, n_changing});
    c = !c;
  }
  return v;
}

// NodesVariablesEEMotion / NodesVariablesEEForce::GetPhaseBasedEEParameterization
// (src/nodes_variables_phase_based.cc:210-298) as a direct index assignment.
SplineLayout PhaseBasedLayout(const double* phase_durations, int n_phases, bool first_constant, int n_changing,
                              bool is_motion, int var_offset) {
  SplineLayout s;
  auto polys = PhasePolys(n_phases, first_constant, n_changing);
  for (auto& p : polys) {
    s.durations.push_back(phase_durations[p.phase] / p.n_in_phase);  // ConvertPhaseToPolyDurations :73-84
    s.poly_phase.push_back(p.phase);
  }
  s.n_nodes = (int)polys.size() + 1;
  s.idx.assign(s.n_nodes * 6, -1);
  s.node_constant.assign(s.n_nodes, 0);
  for (int n = 0; n < s.n_nodes; ++n) {  // IsConstantNode :99-111
    bool c = false;
    if (n > 0 && polys[n - 1].constant) c = true;
    if (n < s.n_nodes - 1 && polys[n].constant) c = true;
    s.node_constant[n] = c;
  }
  int idx = var_offset;
  auto set = [&](int node, int deriv, int dim, int v) { s.idx[(node * 2 + deriv) * 3 + dim] = v; };
  for (int n = 0; n < s.n_nodes; ++n) {
    if (!s.node_constant[n]) {
      for (int d = 0; d < 3; ++d) {
        set(n, 0, d, idx++);
        if (is_motion) {
          if (d != 2) set(n, 1, d, idx++);  // swing: z velocity fixed to zero
        } else {
          set(n, 1, d, idx++);
        }
      }
    } else {
      if (n + 1 >= s.n_nodes) throw std::runtime_error("dangling constant node");
      if (is_motion)
        for (int d = 0; d < 3; ++d) {  // one position shared by both nodes of the stance polynomial
          set(n, 0, d, idx);
          set(n + 1, 0, d, idx);
          idx++;
        }
      n += 1;
    }
  }
  s.var_offset = var_offset;
  s.var_size = idx - var_offset;
  return s;
}

PolyDesc MakeEePoly(const SplineLayout& s, int q) {
  PolyDesc p;
  std::memset(&p, 0, sizeof(p));
  p.iT = 1.0 / s.durations[q];
  int gi[12];
  bool shared = true;
  for (int d = 0; d < 3; ++d)
    if (s.at(q, 0, d) < 0 || s.at(q, 0, d) != s.at(q + 1, 0, d)) shared = false;
  std::vector<int> uniq;
  for (int j = 0; j < 4; ++j)
    for (int d = 0; d < 3; ++d) {
      int node = q + (j >= 2 ? 1 : 0), deriv = j & 1;
      int v = s.at(node, deriv, d);
      if (shared && j == 2) v = -1;  // folded into p0
      gi[j * 3 + d] = v;
      if (v >= 0) uniq.push_back(v);
    }
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  if (uniq.size() > 12) throw std::runtime_error("too many slots");
  const int nslots = (int)uniq.size();
  p.xbase = uniq.empty() ? 0 : uniq.front();
  for (size_t i = 0; i < uniq.size(); ++i)
    if (uniq[i] != p.xbase + (int)i) throw std::runtime_error("polynomial variables not contiguous");
  int dim_of_slot[12], cnt[3] = {0, 0, 0};
  for (int c = 0; c < 12; ++c)
    if (gi[c] >= 0) dim_of_slot[gi[c] - p.xbase] = c % 3;
  for (int sl = 0; sl < nslots; ++sl) cnt[dim_of_slot[sl]]++;
  for (int c = 0; c < 12; ++c) {
    if (gi[c] < 0) {
      p.cand[c] = 0xFFFF;
      continue;
    }
    int sl = gi[c] - p.xbase, d = c % 3;
    int ra = 0, rb = 0, rl = 0;
    for (int s2 = 0; s2 < sl; ++s2) {
      if (dim_of_slot[s2] != (d + 1) % 3) ra++;
      if (dim_of_slot[s2] != (d + 2) % 3) rb++;
      if (dim_of_slot[s2] == d) rl++;
    }
    p.cand[c] = (uint16_t)(sl | (ra << 4) | (rb << 8) | (rl << 12));
  }
  p.meta = (uint32_t)(nslots | (cnt[0] << 4) | (cnt[1] << 8) | (cnt[2] << 12) | ((shared ? 1 : 0) << 16));
  return p;
}

}  // namespace

// ------------------------------------------------------------------ variables
void Structure::BuildVariables() {
  n_ee = schedule.n_ee;
  if (n_ee < 1 || n_ee > kMaxEE || n_ee != model.n_ee) throw std::runtime_error("n_ee mismatch");
  for (int e = 0; e < n_ee; ++e)
    if (schedule.n_phases[e] < 1 || schedule.n_phases[e] > TWR_MAX_PHASES) throw std::runtime_error("bad phase count");
  // Parameters::GetTotalTime (src/parameters.cc:112-126): first foot is the reference
  T = std::accumulate(schedule.phase_durations[0], schedule.phase_durations[0] + schedule.n_phases[0], 0.0);
  for (int e = 0; e < n_ee; ++e) {
    double Te = std::accumulate(schedule.phase_durations[e], schedule.phase_durations[e] + schedule.n_phases[e], 0.0);
    if (std::fabs(Te - T) >= 1e-6) throw std::runtime_error("phase durations of the feet do not sum to the same T");
  }
  // Parameters::GetBasePolyDurations (src/parameters.cc:82-98)
  {
    double dt = params.duration_base_poly, t_left = T;
    if (!(dt > 0)) throw std::runtime_error("duration_base_poly must be positive");
    while (t_left > 1e-10) {
      base.durations.push_back(t_left > dt ? dt : t_left);
      t_left -= dt;
    }
  }
  // NodesVariablesAll (src/nodes_variables_all.cc:34-61): node-major, px py pz vx vy vz
  base.n_nodes = (int)base.durations.size() + 1;
  base.idx.resize(base.n_nodes * 6);
  for (int n = 0; n < base.n_nodes; ++n)
    for (int dv = 0; dv < 2; ++dv)
      for (int d = 0; d < 3; ++d) base.idx[(n * 2 + dv) * 3 + d] = 6 * n + 3 * dv + d;
  base.var_size = base.n_nodes * 6;
  // variable set order: NlpFormulation::GetVariableSets (src/nlp_formulation.cc:63-93)
  int off = 0;
  off_base_lin = off;
  var_sets.push_back({"base-lin", off, base.var_size, 0, 0});
  off += base.var_size;
  off_base_ang = off;
  var_sets.push_back({"base-ang", off, base.var_size, 0, 0});
  off += base.var_size;
  for (int e = 0; e < n_ee; ++e) {  // MakeEndeffectorVariables :127-156 (stance phases constant)
    motion.push_back(PhaseBasedLayout(schedule.phase_durations[e], schedule.n_phases[e],
                                      schedule.in_contact_at_start[e] != 0, params.polys_per_swing, true, off));
    var_sets.push_back({"ee-motion_" + std::to_string(e), off, motion.back().var_size, 0, 0});
    off += motion.back().var_size;
  }
  for (int e = 0; e < n_ee; ++e) {  // MakeForceVariables :158-181 (swing phases constant)
    force.push_back(PhaseBasedLayout(schedule.phase_durations[e], schedule.n_phases[e],
                                     schedule.in_contact_at_start[e] == 0, params.polys_per_stance_force, false, off));
    var_sets.push_back({"ee-force_" + std::to_string(e), off, force.back().var_size, 0, 0});
    off += force.back().var_size;
  }
  // MakeContactScheduleVariables (src/nlp_formulation.cc:183-198), part of x only when the timings are
  // optimised (:78-82); PhaseDurations holds n_phases-1 variables (src/phase_durations.cc:45)
  timings = (params.constraint_sets & TWR_SET_TOTAL_TIME) != 0;
  if (timings)
    for (int e = 0; e < n_ee; ++e) {
      if (schedule.n_phases[e] < 2) throw std::runtime_error("optimised timings need at least two phases per foot");
      off_schedule[e] = off;
      var_sets.push_back({"ee-schedule" + std::to_string(e), off, schedule.n_phases[e] - 1, 0, 0});
      off += schedule.n_phases[e] - 1;
    }
  n_vars = off;
  mpoly.resize(n_ee);
  fpoly.resize(n_ee);
  for (int e = 0; e < n_ee; ++e) {
    for (size_t q = 0; q < motion[e].durations.size(); ++q) mpoly[e].push_back(MakeEePoly(motion[e], (int)q));
    for (size_t q = 0; q < force[e].durations.size(); ++q) fpoly[e].push_back(MakeEePoly(force[e], (int)q));
  }
}

// ------------------------------------------------------------------ time tables
void Structure::BuildTimeTables() {
  grid_dyn = TimeGrid(T, params.dt_dynamic);  // DynamicConstraint ctor, dynamic_constraint.cc:37-51
  grid_rom = TimeGrid(T, params.dt_rom);      // RangeOfMotionConstraint ctor, range_of_motion_constraint.cc:35-50
  dyn_motion.resize(n_ee);
  dyn_force.resize(n_ee);
  rom_motion.resize(n_ee);
  for (double t : grid_dyn) {
    dyn_base.push_back(Locate(t, base.durations));
    for (int e = 0; e < n_ee; ++e) {
      dyn_motion[e].push_back(Locate(t, motion[e].durations));
      dyn_force[e].push_back(Locate(t, force[e].durations));
    }
  }
  for (double t : grid_rom) {
    rom_base.push_back(Locate(t, base.durations));
    for (int e = 0; e < n_ee; ++e) rom_motion[e].push_back(Locate(t, motion[e].durations));
  }
  if (params.constraint_sets & TWR_SET_BASE_ROM) {  // BaseMotionConstraint ctor, base_motion_constraint.cc:38-41
    grid_bm = TimeGrid(T, params.dt_base_motion);
    for (double t : grid_bm) bm_base.push_back(Locate(t, base.durations));
  }
  // force / terrain node tables
  force_nodes.resize(n_ee);
  terrain_rows.resize(n_ee);
  for (int e = 0; e < n_ee; ++e) {
    const SplineLayout& f = force[e];
    const SplineLayout& m = motion[e];
    for (int n = 0; n < f.n_nodes; ++n) {
      if (f.node_constant[n]) continue;  // GetIndicesOfNonConstantNodes, nodes_variables_phase_based.cc:119-129
      int adj_poly = n == 0 ? 0 : n - 1;  // GetPhase -> GetAdjacentPolyIds(node).front(), :131-138,163-179
      int phase = f.poly_phase[adj_poly];
      int start = -1;                     // GetNodeIDAtStartOfPhase, :140-161
      for (size_t q = 0; q < m.poly_phase.size(); ++q)
        if (m.poly_phase[q] == phase) {
          start = (int)q;
          break;
        }
      if (start < 0) throw std::runtime_error("stance phase missing in ee-motion");
      ForceNode fn;
      fn.fidx = f.at(n, 0, 0);
      fn.hidx = m.at(start, 0, 0);
      if (f.at(n, 0, 1) != fn.fidx + 2 || f.at(n, 0, 2) != fn.fidx + 4 || m.at(start, 0, 1) != fn.hidx + 1)
        throw std::runtime_error("unexpected force node layout");
      force_nodes[e].push_back(fn);
    }
    for (int n = 1; n < m.n_nodes; ++n) {  // terrain_constraint.cc:49-51
      TerrainRow tr;
      tr.idx = m.at(n, 0, 0);
      tr.stride = m.at(n, 0, 1) - tr.idx;
      if (m.at(n, 0, 2) != tr.idx + 2 * tr.stride) throw std::runtime_error("unexpected motion node layout");
      terrain_rows[e].push_back(tr);
    }
  }
  // splineacc-base-*: the two NodeSpline::GetJacobianWrtNodes rows of spline_acc_constraint.cc:67-81 at
  // t = T_j of polynomial j and t = 0 of polynomial j+1 (CubicHermitePolynomial::GetDerivativeWrt{Start,End}Node,
  // polynomial.cc:140-234, kAcc); they depend on the durations only.
  if (params.constraint_sets & TWR_SET_BASE_ACC)
    for (size_t j = 0; j + 1 < base.durations.size(); ++j) {
      const double Tp = base.durations[j], Tn = base.durations[j + 1];
      const double Tp2 = std::pow(Tp, 2), Tp3 = std::pow(Tp, 3), Tn2 = std::pow(Tn, 2), Tn3 = std::pow(Tn, 3);
      const double prev[4] = {(12 * Tp) / Tp3 - 6 / Tp2, (6 * Tp) / Tp2 - 4 / Tp, 6 / Tp2 - (12 * Tp) / Tp3, (6 * Tp) / Tp2 - 2 / Tp};
      const double next[4] = {(12 * 0.0) / Tn3 - 6 / Tn2, (6 * 0.0) / Tn2 - 4 / Tn, 6 / Tn2 - (12 * 0.0) / Tn3, (6 * 0.0) / Tn2 - 2 / Tn};
      AccJunction a;
      a.c[0] = prev[0]; a.c[1] = prev[1];
      a.c[2] = prev[2] - next[0]; a.c[3] = prev[3] - next[1];
      a.c[4] = 0.0 - next[2]; a.c[5] = 0.0 - next[3];
      acc_junctions.push_back(a);
    }
  // swing-ee-motion_e: GetIndicesOfNonConstantNodes + the neighbours' position variables
  swing_nodes.resize(n_ee);
  if (params.constraint_sets & TWR_SET_SWING)
    for (int e = 0; e < n_ee; ++e) {
      const SplineLayout& m = motion[e];
      for (int n = 0; n < m.n_nodes; ++n) {
        if (m.node_constant[n]) continue;
        // "assumes ... starting and ending in stance" (swing_constraint.cc:66): the reference indexes
        // nodes.at(node_id -+ 1) and throws otherwise
        if (n == 0 || n == m.n_nodes - 1) throw std::runtime_error("swing constraint needs schedules that start and end in stance");
        SwingNode sn;
        sn.cur = m.at(n, 0, 0);
        if (m.at(n, 1, 0) != sn.cur + 1 || m.at(n, 0, 1) != sn.cur + 2 || m.at(n, 1, 1) != sn.cur + 3)
          throw std::runtime_error("unexpected swing node layout");
        sn.prev_x = m.at(n - 1, 0, 0); sn.prev_y = m.at(n - 1, 0, 1);
        sn.next_x = m.at(n + 1, 0, 0); sn.next_y = m.at(n + 1, 0, 1);
        sn.pad = 0;
        if (sn.prev_x < 0 || sn.prev_y < 0 || sn.next_x < 0 || sn.next_y < 0) throw std::runtime_error("swing neighbour is not a variable");
        swing_nodes[e].push_back(sn);
      }
    }
}

// ------------------------------------------------------------------ CSR pattern + bounds
void Structure::BuildPattern() {
  const double inf = 1e20;
  // rows are emitted straight into col_idx / row_ptr (no per-row containers: this runs once per candidate of a
  // sweep)
  row_ptr.assign(1, 0);
  col_idx.clear();
  col_idx.reserve(1 << 17);
  std::vector<int32_t>& c = col_idx;
  auto end_row = [&]() {
    if (!std::is_sorted(col_idx.begin() + row_ptr.back(), col_idx.end())) throw std::runtime_error("row not sorted");
    row_ptr.push_back((int32_t)col_idx.size());
  };
  auto emit = [&](std::initializer_list<int> cols) {
    col_idx.insert(col_idx.end(), cols.begin(), cols.end());
    end_row();
  };
  auto begin_set = [&](const std::string& name, int n) {
    SetInfo s;
    s.name = name;
    s.offset = (int)row_ptr.size() - 1;
    s.size = n;
    con_sets.push_back(s);
  };
  auto slots_cols = [&](const PolyDesc& p, int want_dim, bool equal, std::vector<int32_t>& out) {
    // columns of the slots whose dim ==/!= want_dim, ascending
    int dim_of_slot[12];
    for (int c = 0; c < 12; ++c)
      if (p.cand[c] != 0xFFFF) dim_of_slot[p.cand[c] & 0xF] = c % 3;
    const int nslots = p.meta & 0xF;
    for (int s = 0; s < nslots; ++s)
      if ((dim_of_slot[s] == want_dim) == equal) out.push_back(p.xbase + s);
  };
  auto set_cols = [&](const SplineLayout& s, int want_dim, bool equal, std::vector<int32_t>& out) {
    // all variables of a phase-based set whose dim ==/!= want_dim, ascending
    std::vector<int> dim_of(s.var_size, -1);
    for (int n = 0; n < s.n_nodes; ++n)
      for (int dv = 0; dv < 2; ++dv)
        for (int d = 0; d < 3; ++d)
          if (s.at(n, dv, d) >= 0) dim_of[s.at(n, dv, d) - s.var_offset] = d;
    for (int i = 0; i < s.var_size; ++i)
      if ((dim_of[i] == want_dim) == equal) out.push_back(s.var_offset + i);
  };
  auto sched_cols = [&](int e, std::vector<int32_t>& out) {
    for (int i = 0; i < schedule.n_phases[e] - 1; ++i) out.push_back(off_schedule[e] + i);
  };
  const int sets = params.constraint_sets;
  // --- terrain-ee-motion_e  (terrain_constraint.cc:90-108): [x, y, z] of node id = row+1
  for (int e = 0; e < n_ee && (sets & TWR_SET_TERRAIN); ++e) {
    begin_set("terrain-ee-motion_" + std::to_string(e), (int)terrain_rows[e].size());
    for (size_t r = 0; r < terrain_rows[e].size(); ++r) {
      const TerrainRow& tr = terrain_rows[e][r];
      emit({tr.idx, tr.idx + tr.stride, tr.idx + 2 * tr.stride});
      bool constant = motion[e].node_constant[r + 1];
      lower.push_back(0.0);
      upper.push_back(constant ? 0.0 : inf);  // terrain_constraint.cc:72-88
    }
  }
  // --- dynamic (dynamic_constraint.cc:73-117, single_rigid_body_dynamics.cc:103-192)
  if (sets & TWR_SET_DYNAMIC) begin_set("dynamic", (int)grid_dyn.size() * 6);
  for (size_t k = 0; k < grid_dyn.size() && (sets & TWR_SET_DYNAMIC); ++k) {
    int q = dyn_base[k].poly;
    for (int r = 0; r < 3; ++r) {  // AX, AY, AZ
      for (int node = q; node <= q + 1; ++node)  // -sum [f]x J_pos: dims != r, pos then vel per node
        for (int dv = 0; dv < 2; ++dv)
          for (int d = 0; d < 3; ++d)
            if (d != r) c.push_back(off_base_lin + 6 * node + 3 * dv + d);
      for (int i = 0; i < 12; ++i) c.push_back(off_base_ang + 6 * q + i);  // structurally full
      if (!timings) {
        for (int e = 0; e < n_ee; ++e) slots_cols(mpoly[e][dyn_motion[e][k].poly], r, false, c);  // [f]x J_p
        for (int e = 0; e < n_ee; ++e) slots_cols(fpoly[e][dyn_force[e][k].poly], r, false, c);   // [r]x J_f
      } else {  // PhaseSpline: every variable of the set (phase_spline.cc:44-51), then the duration columns
        for (int e = 0; e < n_ee; ++e) set_cols(motion[e], r, false, c);
        for (int e = 0; e < n_ee; ++e) set_cols(force[e], r, false, c);
        for (int e = 0; e < n_ee; ++e) sched_cols(e, c);  // dynamic_constraint.cc:107-113
      }
      end_row();
      lower.push_back(0.0);
      upper.push_back(0.0);
    }
    for (int d = 0; d < 3; ++d) {  // LX, LY, LZ
      for (int j = 0; j < 4; ++j) c.push_back(off_base_lin + 6 * q + 3 * j + d);  // m J_acc
      if (!timings) {
        for (int e = 0; e < n_ee; ++e) slots_cols(fpoly[e][dyn_force[e][k].poly], d, true, c);  // -J_f
      } else {
        for (int e = 0; e < n_ee; ++e) set_cols(force[e], d, true, c);
        for (int e = 0; e < n_ee; ++e) sched_cols(e, c);
      }
      end_row();
      lower.push_back(0.0);
      upper.push_back(0.0);
    }
  }
  // --- splineacc-base-lin, splineacc-base-ang (spline_acc_constraint.cc:67-88): the sparse difference
  // acc_prev - acc_next keeps the union pattern: both value kinds of nodes j, j+1, j+2 in dimension d
  // (the shared node's position entry is a structural non-zero even when the durations are equal and
  // its value cancels)
  if (sets & TWR_SET_BASE_ACC)
    for (int which = 0; which < 2; ++which) {
      const int off = which == 0 ? off_base_lin : off_base_ang;
      begin_set(which == 0 ? "splineacc-base-lin" : "splineacc-base-ang", 3 * (int)acc_junctions.size());
      for (size_t j = 0; j < acc_junctions.size(); ++j)
        for (int d = 0; d < 3; ++d) {
          for (int i = 0; i < 6; ++i) c.push_back(off + 6 * (int)j + 3 * i + d);
          end_row();
          lower.push_back(0.0);
          upper.push_back(0.0);
        }
    }
  // --- rangeofmotion-e (range_of_motion_constraint.cc:83-109)
  for (int e = 0; e < n_ee && (sets & TWR_SET_ROM); ++e) {
    begin_set("rangeofmotion-" + std::to_string(e), (int)grid_rom.size() * 3);
    for (size_t k = 0; k < grid_rom.size(); ++k) {
      int q = rom_base[k].poly;
      const PolyDesc& mp = mpoly[e][rom_motion[e][k].poly];
      for (int r = 0; r < 3; ++r) {
        for (int i = 0; i < 12; ++i) c.push_back(off_base_lin + 6 * q + i);  // -R^T J_c
        for (int i = 0; i < 12; ++i)  // DerivOfRotVecMult(inverse): row 0 does not depend on roll
          if (!(r == 0 && i % 3 == 0)) c.push_back(off_base_ang + 6 * q + i);
        if (!timings) {
          for (int s = 0; s < (int)(mp.meta & 0xF); ++s) c.push_back(mp.xbase + s);  // R^T J_p
        } else {  // b_R_w (dense) times the all-variables PhaseSpline rows, then the duration columns (:106-108)
          for (int i = 0; i < motion[e].var_size; ++i) c.push_back(motion[e].var_offset + i);
          sched_cols(e, c);
        }
        end_row();
        lower.push_back(model.nominal_stance[e][r] - model.max_dev[r]);  // :71-81
        upper.push_back(model.nominal_stance[e][r] + model.max_dev[r]);
      }
    }
  }
  // --- force-ee-force_e (force_constraint.cc:107-171)
  for (int e = 0; e < n_ee && (sets & TWR_SET_FORCE); ++e) {
    begin_set("force-ee-force_" + std::to_string(e), (int)force_nodes[e].size() * 5);
    for (const ForceNode& fn : force_nodes[e]) {
      for (int r = 0; r < 5; ++r) emit({fn.hidx, fn.hidx + 1, fn.fidx, fn.fidx + 2, fn.fidx + 4});
      lower.push_back(0.0);  upper.push_back(model.force_limit);  // :91-105
      lower.push_back(-inf); upper.push_back(0.0);
      lower.push_back(0.0);  upper.push_back(inf);
      lower.push_back(-inf); upper.push_back(0.0);
      lower.push_back(0.0);  upper.push_back(inf);
    }
  }
  // --- swing-ee-motion_e (swing_constraint.cc:86-121): rows x-pos, x-vel, y-pos, y-vel per swing node
  for (int e = 0; e < n_ee && (sets & TWR_SET_SWING); ++e) {
    begin_set("swing-ee-motion_" + std::to_string(e), 4 * (int)swing_nodes[e].size());
    for (const SwingNode& sn : swing_nodes[e]) {
      emit({sn.prev_x, sn.cur, sn.next_x});
      emit({sn.prev_x, sn.cur + 1, sn.next_x});
      emit({sn.prev_y, sn.cur + 2, sn.next_y});
      emit({sn.prev_y, sn.cur + 3, sn.next_y});
      for (int r = 0; r < 4; ++r) {
        lower.push_back(0.0);
        upper.push_back(0.0);
      }
    }
  }
  // --- baseMotion (base_motion_constraint.cc:60-90): rows AX..AZ = base-ang position, LX..LZ = base-lin position
  if (sets & TWR_SET_BASE_ROM) {
    begin_set("baseMotion", 6 * (int)grid_bm.size());
    const double dev_rad = 0.05, z0 = params.base_z_init;
    for (size_t k = 0; k < grid_bm.size(); ++k) {
      const int q = bm_base[k].poly;
      for (int r6 = 0; r6 < 6; ++r6) {
        const int off = r6 < 3 ? off_base_ang : off_base_lin, d = r6 % 3;
        emit({off + 6 * q + d, off + 6 * q + 3 + d, off + 6 * q + 6 + d, off + 6 * q + 9 + d});
      }
      lower.push_back(-dev_rad); upper.push_back(dev_rad);   // AX
      lower.push_back(-dev_rad); upper.push_back(dev_rad);   // AY
      lower.push_back(-inf);     upper.push_back(inf);       // AZ
      lower.push_back(-inf);     upper.push_back(inf);       // LX
      lower.push_back(-inf);     upper.push_back(inf);       // LY
      lower.push_back(z0 - 0.02); upper.push_back(z0 + 0.1); // LZ
    }
  }
  // --- totalduration-e (total_duration_constraint.cc:50-72): sum of the optimised durations
  for (int e = 0; e < n_ee && (sets & TWR_SET_TOTAL_TIME); ++e) {
    begin_set("totalduration-" + std::to_string(e), 1);
    sched_cols(e, c);
    end_row();
    lower.push_back(0.1);
    upper.push_back(T - 0.2);  // min_duration_last_phase
  }
  n_rows = (int)row_ptr.size() - 1;
  nnz = row_ptr[n_rows];
  for (auto& s : con_sets) {
    s.nnz_offset = row_ptr[s.offset];
    s.nnz = row_ptr[s.offset + s.size] - s.nnz_offset;
  }
}

// ------------------------------------------------------------------ device blob
void Structure::PackBlob() {
  DevStruct h;
  std::memset(&h, 0, sizeof(h));
  std::vector<char> body;
  auto put_aligned = [&](const void* src, size_t bytes, size_t align) -> uint32_t {   // (a blob starts on a 256-byte line)
    size_t off = (sizeof(DevStruct) + body.size() + align - 1) / align * align;
    body.resize(off - sizeof(DevStruct) + bytes);
    if (bytes) std::memcpy(body.data() + off - sizeof(DevStruct), src, bytes);
    return (uint32_t)off;
  };
  auto put = [&](const void* src, size_t bytes) -> uint32_t { return put_aligned(src, bytes, 16); };
  // a layout table of dyn_kernel (device_tables.h): a batch stores byte-identical ones once
  dyn_layout_tables.clear();
  auto put_layout = [&](const void* src, size_t bytes) -> uint32_t {
    const uint32_t off = put_aligned(src, bytes, 64);
    dyn_layout_tables.push_back({off, (uint32_t)bytes});
    return off;
  };
  h.n_ee = n_ee;
  h.terrain_id = model.terrain_id;
  std::vector<TerrainRow> all_rows;
  std::vector<ForceNode> all_nodes;
  std::vector<SwingNode> all_swing;
  auto family = [&](const std::string& prefix, int rows_per_item, auto& per_ee, auto& all, int32_t& row0, int32_t& nnz0) {
    // the per-ee sets of one family are adjacent in g / jac: one flat list
    for (int e = 0; e < n_ee; ++e) {
      const SetInfo* si = FindSet(prefix + std::to_string(e));
      if (!si) return;
      if (e == 0) { row0 = si->offset; nnz0 = si->nnz_offset; }
      if (si->offset != row0 + rows_per_item * (int)all.size()) throw std::runtime_error(prefix + " sets not adjacent");
      all.insert(all.end(), per_ee[e].begin(), per_ee[e].end());
    }
  };
  family("terrain-ee-motion_", 1, terrain_rows, all_rows, h.row_terrain, h.nnz_terrain);
  family("force-ee-force_", 5, force_nodes, all_nodes, h.row_force, h.nnz_force);
  family("swing-ee-motion_", 4, swing_nodes, all_swing, h.row_swing, h.nnz_swing);
  h.n_terrain_rows = (int)all_rows.size();
  h.n_force_nodes = (int)all_nodes.size();
  h.n_swing_nodes = (int)all_swing.size();
  h.off_base_ang = off_base_ang;
  h.inv_t_swing = 1.0 / 0.3;  // t_swing_avg_, swing_constraint.h:68
  if (const SetInfo* si = FindSet("splineacc-base-lin")) {
    const SetInfo* sa = FindSet("splineacc-base-ang");
    if (!sa || sa->offset != si->offset + si->size || off_base_lin != 0) throw std::runtime_error("splineacc sets not adjacent");
    h.row_acc = si->offset;
    h.nnz_acc = si->nnz_offset;
    h.n_junctions = (int)acc_junctions.size();
  }
  const SetInfo* dyn_set = FindSet("dynamic");
  const int row_dyn = dyn_set ? dyn_set->offset : 0, nnz_dyn = dyn_set ? dyn_set->nnz_offset : 0;
  int row_rom[kMaxEE] = {0, 0, 0, 0}, nnz_rom[kMaxEE] = {0, 0, 0, 0};
  const bool have_rom = FindSet("rangeofmotion-0") != nullptr;
  for (int e = 0; e < n_ee && have_rom; ++e) {
    const SetInfo* si = FindSet("rangeofmotion-" + std::to_string(e));
    row_rom[e] = si->offset;
    nnz_rom[e] = si->nnz_offset;
  }
  {  // the node head (device_tables.h): first 64 terrain rows, first 64 force nodes, at fixed offsets behind the header
    std::vector<char> head(kNodeHeadBytes, 0);
    if (!all_rows.empty()) std::memcpy(head.data(), all_rows.data(), std::min<size_t>(64, all_rows.size()) * sizeof(TerrainRow));
    if (!all_nodes.empty())
      std::memcpy(head.data() + 64 * sizeof(TerrainRow), all_nodes.data(), std::min<size_t>(64, all_nodes.size()) * sizeof(ForceNode));
    if (put(head.data(), head.size()) != kNodeHeadTerrainOff) throw std::runtime_error("node head is not the first table of the blob");
  }
  {  // candidate scoring: family and bounds of every row (device_tables.h, ScoreTables)
    if (con_sets.size() > (size_t)kMaxConSets) throw std::runtime_error("too many constraint sets");
    std::vector<double> pairs;
    std::map<std::pair<uint64_t, uint64_t>, int> pair_index;   // distinct (lower, upper), by bit pattern
    std::vector<uint16_t> meta((size_t)n_rows, 0);
    int slot_of[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, n_slots = 0;
    // (sets in row order: the slots are handed out in the order the families first appear along the rows)
    std::vector<size_t> by_row(con_sets.size());
    std::iota(by_row.begin(), by_row.end(), (size_t)0);
    std::sort(by_row.begin(), by_row.end(), [&](size_t a, size_t b) { return con_sets[a].offset < con_sets[b].offset; });
    for (size_t i : by_row) {
      const std::string& nm = con_sets[i].name;
      auto starts = [&](const char* p) { return nm.rfind(p, 0) == 0; };
      const int fam = starts("terrain-") ? 0 : starts("dynamic") ? 1 : starts("splineacc-") ? 2 : starts("rangeofmotion-") ? 3
                    : starts("force-") ? 4 : starts("swing-") ? 5 : starts("totalduration-") ? 6 : starts("baseMotion") ? 7 : -1;
      if (fam < 0) throw std::runtime_error("constraint set of an unknown family");
      if (con_sets[i].size > 0 && slot_of[fam] < 0) slot_of[fam] = n_slots++;
      if (con_sets[i].size > 0 && slot_of[fam] != n_slots - 1) throw std::runtime_error("the sets of a constraint family are not adjacent in g");
      for (int r = con_sets[i].offset; r < con_sets[i].offset + con_sets[i].size; ++r) {
        uint64_t kl, ku;
        std::memcpy(&kl, &lower[r], 8);
        std::memcpy(&ku, &upper[r], 8);
        auto it = pair_index.find({kl, ku});
        if (it == pair_index.end()) {
          it = pair_index.emplace(std::make_pair(kl, ku), (int)(pairs.size() / 2)).first;
          pairs.push_back(lower[r]);
          pairs.push_back(upper[r]);
        }
        if (it->second >= kScoreMaxPairs) throw std::runtime_error("more than 127 distinct constraint bounds in one structure");
        meta[r] = (uint16_t)(slot_of[fam] << 12 | it->second);
      }
    }
    ScoreTables sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.n_rows = n_rows;
    sc.n_pairs = (int)(pairs.size() / 2);
    for (int f = 0; f < 8; ++f) sc.slot_of_family[f] = (int8_t)slot_of[f];
    // [ ScoreTables | pairs | zero padding to kScoreHeadBytes ][ meta words ] at a FIXED offset behind the node head: score_kernel
    // asks for all of it without having seen a single field of the header
    if (sizeof(sc) + pairs.size() * sizeof(double) > (size_t)kScoreHeadBytes) throw std::runtime_error("score record head overflows");
    std::vector<char> rec(kScoreHeadBytes + (meta.size() * sizeof(uint16_t) + 15) / 16 * 16, 0);
    std::memcpy(rec.data(), &sc, sizeof(sc));
    std::memcpy(rec.data() + sizeof(sc), pairs.data(), pairs.size() * sizeof(double));
    if (!meta.empty()) std::memcpy(rec.data() + kScoreHeadBytes, meta.data(), meta.size() * sizeof(uint16_t));
    h.o_score = put(rec.data(), rec.size());
    if (h.o_score != kScoreOff) throw std::runtime_error("score record is not at its fixed offset");
  }
  h.o_force_nodes = put(all_nodes.data(), all_nodes.size() * sizeof(ForceNode));
  h.o_terrain_rows = put(all_rows.data(), all_rows.size() * sizeof(TerrainRow));
  h.o_acc = put(acc_junctions.data(), acc_junctions.size() * sizeof(AccJunction));
  if (const SetInfo* si = FindSet("baseMotion")) {
    std::vector<BaseNode> bn(grid_bm.size());
    for (size_t k = 0; k < grid_bm.size(); ++k) {
      bn[k].t = bm_base[k].t_local;
      bn[k].iT = 1.0 / base.durations[bm_base[k].poly];
      bn[k].q6 = 6 * bm_base[k].poly;
      bn[k].pad = 0;
    }
    h.row_bm = si->offset;
    h.nnz_bm = si->nnz_offset;
    h.n_bm_nodes = (int)bn.size();
    h.o_bm = put(bn.data(), bn.size() * sizeof(BaseNode));
  }
  h.o_swing_nodes = put(all_swing.data(), all_swing.size() * sizeof(SwingNode));
  // --- dynamic, optimised timings: the base-spline part of the per-node record (the rest depends on x)
  if (dyn_set && timings) {
    std::vector<DynShared> sh(grid_dyn.size());
    for (size_t k = 0; k < grid_dyn.size(); ++k) {
      std::memset(&sh[k], 0, sizeof(DynShared));
      sh[k].tb = dyn_base[k].t_local;
      sh[k].iTb = 1.0 / base.durations[dyn_base[k].poly];
      sh[k].q6 = 6 * dyn_base[k].poly;
      sh[k].voff = row_ptr[row_dyn + 6 * k] - nnz_dyn;
    }
    off_dyn_shared = put(sh.data(), sh.size() * sizeof(DynShared));
  }
  // --- dynamic, fixed timings: slices, staging maps and per-lane records with every index resolved to an LDS
  // byte offset (device_tables.h).  The put offsets are read off the CSR pattern itself, so kernel and
  // pattern cannot disagree.
  if (dyn_set && !timings) {
    const int K = (int)grid_dyn.size();
    auto poly_range = [&](const std::vector<PolyDesc>& pd, int qa, int qb, int& lo, int& hi) {
      lo = 1 << 30;
      hi = -1;
      for (int q = qa; q <= qb; ++q) {
        const int ns = (int)(pd[q].meta & 0xF);
        if (!ns) continue;
        lo = std::min(lo, pd[q].xbase);
        hi = std::max(hi, pd[q].xbase + ns);
      }
      if (hi < 0) lo = hi = 0;
    };
    auto stage_count = [&](int k0, int k1) {  // doubles of x the nodes [k0, k1) read
      int n = 2 * 6 * (dyn_base[k1 - 1].poly - dyn_base[k0].poly + 2);
      for (int e = 0; e < n_ee; ++e) {
        int lo, hi;
        poly_range(mpoly[e], dyn_motion[e][k0].poly, dyn_motion[e][k1 - 1].poly, lo, hi);
        n += hi - lo;
        poly_range(fpoly[e], dyn_force[e][k0].poly, dyn_force[e][k1 - 1].poly, lo, hi);
        n += hi - lo;
      }
      return n;
    };
    auto nvals_of = [&](int k0, int k1) { return row_ptr[row_dyn + 6 * k1] - row_ptr[row_dyn + 6 * k0]; };
    // --- DynPolyT / DynPolyL: one record pair per polynomial of every ee spline, ordered by start time, so that the records a slice
    // (a short time window) reads sit next to each other and an 8-bit index relative to the slice's first record
    // reaches all of them
    struct PolyRef {
      double t0;
      int e, kind, q;   // kind 0: ee-motion, 1: ee-force
    };
    std::vector<PolyRef> order;
    for (int e = 0; e < n_ee; ++e)
      for (int kind = 0; kind < 2; ++kind) {
        const std::vector<double>& dur = kind == 0 ? motion[e].durations : force[e].durations;
        double t0 = 0.0;   // the running sum Spline::GetSegmentID compares t against (spline.cc:52-57)
        for (size_t q = 0; q < dur.size(); ++q) {
          order.push_back({t0, e, kind, (int)q});
          t0 += dur[q];
        }
      }
    std::stable_sort(order.begin(), order.end(), [](const PolyRef& a, const PolyRef& b) { return a.t0 < b.t0; });
    std::vector<std::vector<int>> rec_of[2];   // [kind][ee][q] -> record index
    for (int kind = 0; kind < 2; ++kind) {
      rec_of[kind].resize(n_ee);
      for (int e = 0; e < n_ee; ++e) rec_of[kind][e].resize(kind == 0 ? mpoly[e].size() : fpoly[e].size());
    }
    std::vector<DynPolyT> polys_t(order.size());
    std::vector<DynPolyL> polys(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
      const PolyRef& pr = order[i];
      rec_of[pr.kind][pr.e][pr.q] = (int)i;
      const PolyDesc& pd = pr.kind == 0 ? mpoly[pr.e][pr.q] : fpoly[pr.e][pr.q];
      polys_t[i].t0 = pr.t0;
      polys_t[i].iT = pd.iT;
      DynPolyL& P = polys[i];
      std::memset(&P, 0, sizeof(P));
      int dim_of_slot[12] = {0}, slot_of[12];
      for (int c = 0; c < 12; ++c) {
        slot_of[c] = pd.cand[c] == 0xFFFF ? -1 : (pd.cand[c] & 0xF);
        if (slot_of[c] >= 0) {
          P.rel[c] = (uint8_t)slot_of[c];
          P.pres[c] = 0xFF;
          dim_of_slot[slot_of[c]] = c % 3;
        }
      }
      const int nslots = pd.meta & 0xF;
      auto rank = [&](int slot, int row_dim, bool equal) {   // position of the slot among the slots a row holds
        int r = 0;
        for (int s2 = 0; s2 < slot; ++s2) r += (dim_of_slot[s2] == row_dim) == equal;
        return r;
      };
      if (pr.kind == 0) {
        if ((pd.meta >> 16) & 1) P.flags |= 1;
        for (int d = 0; d < 3; ++d)
          if (slot_of[d] < 0) throw std::runtime_error("ee-motion node position that is not a variable");
        for (int c = 0; c < 12; ++c) {
          const int d = c % 3, src = slot_of[c] >= 0 ? c : d;   // not a variable: p0's slots (stored last by the kernel)
          P.code[2 * c + 0] = (uint8_t)(8 * rank(slot_of[src], (d + 1) % 3, false));
          P.code[2 * c + 1] = (uint8_t)(8 * rank(slot_of[src], (d + 2) % 3, false));
        }
      } else {
        bool present[4];
        for (int j = 0; j < 4; ++j) {
          present[j] = slot_of[3 * j] >= 0;
          for (int d = 1; d < 3; ++d)
            if ((slot_of[3 * j + d] >= 0) != present[j]) throw std::runtime_error("ee-force node value that is a variable in some dimensions only");
        }
        if (present[0] != present[1] || present[2] != present[3]) throw std::runtime_error("ee-force node with a constant position or velocity only");
        if (!present[0]) P.flags |= 2;
        if (!present[2]) P.flags |= 4;
        for (int c = 0; c < 12; ++c) {
          const int d = c % 3, j = c / 3;
          if (nslots == 0) continue;                      // no variables at all: codes 0, the tile starts point at trash
          const int src = present[j] ? c : (j ^ 2) * 3 + d;   // constant node: the other node's value of the same kind
          if (slot_of[src] < 0) throw std::runtime_error("ee-force polynomial layout not understood");
          P.code[3 * c + 0] = (uint8_t)(8 * rank(slot_of[src], (d + 1) % 3, false));
          P.code[3 * c + 1] = (uint8_t)(8 * rank(slot_of[src], (d + 2) % 3, false));
          P.code[3 * c + 2] = (uint8_t)(8 * rank(slot_of[src], d, true));
        }
      }
    }
    auto rec_span_ok = [&](int k0, int k1) {   // the 8-bit record indices of a slice (255 = dummy)
      int lo = 1 << 30, hi = -1;
      for (int e = 0; e < n_ee; ++e)
        for (int k : {k0, k1 - 1}) {
          lo = std::min({lo, rec_of[0][e][dyn_motion[e][k0].poly], rec_of[1][e][dyn_force[e][k0].poly]});
          hi = std::max({hi, rec_of[0][e][dyn_motion[e][k].poly], rec_of[1][e][dyn_force[e][k].poly]});
        }
      return hi - lo < kDynPolyDummy;
    };
    std::vector<DynNodeT> nodes_t(K);
    std::vector<DynNodeL> nodes(K);
    std::vector<DynSel> sel((size_t)K * 4);
    std::vector<DynTile> tiles;   // four per (slice, polynomial combination)
    std::vector<int> combo_key;   // active polynomial ids of the last combination
    dyn_slices.clear();
    dyn_staged_max = 0;
    // Slices that stage at most 128 doubles of x read the 256-byte form of their staging map and gather x with two loads per
    // lane instead of four (dyn_body XC = 2) -- if EVERY slice of a batch does.  Fine discretisations do anyway (a 12..15-node
    // slice of a K = 200 problem stages 90-130); where the general capacity would let a few slices stage a little more, they
    // are cut at 128 instead, as long as that costs at most one more slice per eight.
    auto count_slices = [&](int cap) {
      int n = 0;
      for (int k0 = 0; k0 < K; ++n) {
        int k1 = k0;
        while (k1 < K && k1 - k0 < kDynNodes && nvals_of(k0, k1 + 1) <= kDynImage && stage_count(k0, k1 + 1) <= cap && rec_span_ok(k0, k1 + 1)) ++k1;
        if (k1 == k0) return 1 << 30;
        k0 = k1;
      }
      return n;
    };
    const int n_general = count_slices(kDynXsCap), n_small = count_slices(std::min(kDynXsCap, 128));
    const int xs_cap = n_small <= n_general + (n_general + 7) / 8 ? std::min(kDynXsCap, 128) : kDynXsCap;
    for (int k0 = 0; k0 < K;) {
      int k1 = k0;
      while (k1 < K && k1 - k0 < kDynNodes && nvals_of(k0, k1 + 1) <= kDynImage && stage_count(k0, k1 + 1) <= xs_cap &&
             rec_span_ok(k0, k1 + 1))
        ++k1;
      if (k1 == k0) throw std::runtime_error("one time node of the dynamic set exceeds the LDS staging capacity");
      if (nvals_of(k0, k1) < 16) throw std::runtime_error("a time-node run with fewer than 16 Jacobian values cannot be staged");
      // staging layout of the slice: xs index 2 + e
      std::vector<uint16_t> map(256, 0);   // lane-transposed below; unused entries stage x[0] (harmless)
      std::vector<int> xidx;               // x index of staging entry e
      const int qmin = dyn_base[k0].poly, nbase = 6 * (dyn_base[k1 - 1].poly - qmin + 2);
      for (int i = 0; i < nbase; ++i) xidx.push_back(off_base_lin + 6 * qmin + i);
      for (int i = 0; i < nbase; ++i) xidx.push_back(off_base_ang + 6 * qmin + i);
      int mlo[kMaxEE], mst[kMaxEE], flo[kMaxEE], fst[kMaxEE];   // first x index / staging entry of every ee range
      for (int e = 0; e < n_ee; ++e) {
        int hi;
        poly_range(mpoly[e], dyn_motion[e][k0].poly, dyn_motion[e][k1 - 1].poly, mlo[e], hi);
        mst[e] = (int)xidx.size();
        for (int i = mlo[e]; i < hi; ++i) xidx.push_back(i);
        poly_range(fpoly[e], dyn_force[e][k0].poly, dyn_force[e][k1 - 1].poly, flo[e], hi);
        fst[e] = (int)xidx.size();
        for (int i = flo[e]; i < hi; ++i) xidx.push_back(i);
      }
      if ((int)xidx.size() > kDynXsCap) throw std::runtime_error("staging count inconsistent");
      for (size_t e = 0; e < xidx.size(); ++e) {
        if (xidx[e] < 0 || xidx[e] > 0xFFFF) throw std::runtime_error("x index does not fit the staging map");
        map[(e % 64) * 4 + e / 64] = (uint16_t)xidx[e];
      }
      DynSlice sl;
      sl.k0 = k0;
      sl.cnt = k1 - k0;
      sl.nvals = nvals_of(k0, k1);
      sl.map = put_layout(map.data(), map.size() * sizeof(uint16_t));
      sl.map2 = sl.map;
      dyn_staged_max = std::max(dyn_staged_max, (int)xidx.size());
      if (xidx.size() <= 128) {   // the 256-byte form: lane l holds entries l and 64 + l (what a batch of such slices reads)
        std::vector<uint16_t> m2(128, 0);
        for (size_t e = 0; e < xidx.size(); ++e) m2[(e % 64) * 2 + e / 64] = (uint16_t)xidx[e];
        sl.map2 = put_layout(m2.data(), m2.size() * sizeof(uint16_t));
      }
      sl.poly0 = 1 << 30;
      for (int e = 0; e < n_ee; ++e)
        sl.poly0 = std::min({sl.poly0, rec_of[0][e][dyn_motion[e][k0].poly], rec_of[1][e][dyn_force[e][k0].poly]});
      dyn_slices.push_back(sl);
      combo_key.clear();   // a new slice has its own staging layout: its first node opens a new combination
      for (int k = k0; k < k1; ++k) {
        const int row0 = row_dyn + 6 * k, v0 = row_ptr[row0];
        DynNodeL& N = nodes[k];
        std::memset(&N, 0, sizeof(N));
        nodes_t[k].t = grid_dyn[k];
        nodes_t[k].tb = dyn_base[k].t_local;
        nodes_t[k].iTb = 1.0 / base.durations[dyn_base[k].poly];
        N.sb_lin = (uint16_t)(8 * (2 + 6 * (dyn_base[k].poly - qmin)));
        N.sb_ang = (uint16_t)(8 * (2 + nbase + 6 * (dyn_base[k].poly - qmin)));
        const int node_rel = v0 - row_ptr[row_dyn + 6 * k0];
        N.nb = (uint16_t)(8 * node_rel);
        N.rs1 = (uint16_t)(8 * (row_ptr[row0 + 1] - v0));
        N.rs2 = (uint16_t)(8 * (row_ptr[row0 + 2] - v0));
        for (int d = 0; d < 3; ++d) N.rl[d] = (uint16_t)(8 * (row_ptr[row0 + 3 + d] - v0));
        // where a tile starts in a row, relative to the node's first value: the position of the polynomial's first
        // variable in the CSR row (read off the pattern itself, so kernel and pattern cannot disagree)
        auto tile_start = [&](int row, int col) -> int {
          const int32_t* b = col_idx.data() + row_ptr[row0 + row];
          const int32_t* e2 = col_idx.data() + row_ptr[row0 + row + 1];
          return (int)(std::lower_bound(b, e2, col) - col_idx.data()) - v0;
        };
        auto find = [&](int row, int col) -> int {  // position of `col` in row `row`, relative to the node's first value
          const int32_t* b = col_idx.data() + row_ptr[row0 + row];
          const int32_t* e2 = col_idx.data() + row_ptr[row0 + row + 1];
          const int32_t* it = std::lower_bound(b, e2, col);
          if (it == e2 || *it != col) throw std::runtime_error("dynamic pattern lacks an expected column");
          return (int)(it - col_idx.data()) - v0;
        };
        // the tile starts depend on the node only through the active polynomials: one record set per combination
        std::vector<int> key;
        for (int e = 0; e < n_ee; ++e) {
          key.push_back(dyn_motion[e][k].poly);
          key.push_back(dyn_force[e][k].poly);
        }
        const bool new_combo = key != combo_key;
        if (new_combo) {
          combo_key = key;
          tiles.resize(tiles.size() + 4);
        }
        const size_t tile0 = tiles.size() - 4;
        if (tile0 + 3 > 0xFFFF) throw std::runtime_error("too many polynomial combinations for 16-bit tile indices");
        for (int role = 0; role < 4; ++role) {
          DynSel& Sx = sel[(size_t)k * 4 + role];
          DynTile T;
          std::memset(&T, 0, sizeof(T));
          Sx.tile = (uint16_t)(tile0 + role);
          Sx.dm = Sx.df = (uint8_t)kDynPolyDummy;
          // trash: base-lin entry `role` of every row (the rows' first four entries are base-lin values, which the same
          // wave writes after the tiles); the dummy record's codes are 0
          const int row_start[6] = {0, N.rs1, N.rs2, N.rl[0], N.rl[1], N.rl[2]};
          for (int r = 0; r < 3; ++r) T.base_m[r] = (uint16_t)(row_start[r] + 8 * role);
          for (int r = 0; r < 6; ++r) T.base_f[r] = (uint16_t)(row_start[r] + 8 * role);
          if (role < n_ee) {
            const int e = role;
            const int qm = dyn_motion[e][k].poly, qf = dyn_force[e][k].poly;
            const PolyDesc& mp = mpoly[e][qm];
            const PolyDesc& fp = fpoly[e][qf];
            Sx.dm = (uint8_t)(rec_of[0][e][qm] - sl.poly0);
            Sx.df = (uint8_t)(rec_of[1][e][qf] - sl.poly0);
            if (rec_of[0][e][qm] - sl.poly0 >= kDynPolyDummy || rec_of[1][e][qf] - sl.poly0 >= kDynPolyDummy || rec_of[0][e][qm] < sl.poly0 ||
                rec_of[1][e][qf] < sl.poly0)
              throw std::runtime_error("polynomial record index does not fit the slice");
            T.s_m = (uint8_t)(2 + mst[e] + mp.xbase - mlo[e]);
            if ((mp.meta & 0xF) == 0) throw std::runtime_error("ee-motion polynomial without variables");
            for (int r = 0; r < 3; ++r) T.base_m[r] = (uint16_t)(8 * tile_start(r, mp.xbase));
            if ((fp.meta & 0xF) != 0) {
              T.s_f = (uint8_t)(2 + fst[e] + fp.xbase - flo[e]);
              for (int r = 0; r < 6; ++r) T.base_f[r] = (uint16_t)(8 * tile_start(r, fp.xbase));
            }
            // self-check of the decomposition  offset = tile start + 8 * rank  against the pattern, value by value (once per
            // polynomial combination: the nodes of a combination share their tile records, asserted below)
            const DynPolyL& PM = polys[rec_of[0][e][qm]];
            const DynPolyL& PF = polys[rec_of[1][e][qf]];
            for (int c = 0; c < 12 && new_combo; ++c) {
              const int d = c % 3, r1 = (d + 1) % 3, r2 = (d + 2) % 3;
              if (mp.cand[c] != 0xFFFF) {
                const int col = mp.xbase + (mp.cand[c] & 0xF);
                if (T.base_m[r1] + PM.code[2 * c] != 8 * find(r1, col) || T.base_m[r2] + PM.code[2 * c + 1] != 8 * find(r2, col))
                  throw std::runtime_error("ee-motion tile offsets disagree with the CSR pattern");
              }
              if (fp.cand[c] != 0xFFFF) {
                const int col = fp.xbase + (fp.cand[c] & 0xF);
                if (T.base_f[r1] + PF.code[3 * c] != 8 * find(r1, col) || T.base_f[r2] + PF.code[3 * c + 1] != 8 * find(r2, col) ||
                    T.base_f[3 + d] + PF.code[3 * c + 2] != 8 * find(3 + d, col))
                  throw std::runtime_error("ee-force tile offsets disagree with the CSR pattern");
              }
            }
          }
          if (!new_combo && std::memcmp(&tiles[tile0 + role], &T, sizeof(T)) != 0)
            throw std::runtime_error("tile starts differ inside one polynomial combination");
          tiles[tile0 + role] = T;
        }
      }
      k0 = k1;
    }
    // times first, then the layout tables (twr_batch_create stores byte-identical layout tables of a batch once)
    off_dyn_nodes_t = put(nodes_t.data(), nodes_t.size() * sizeof(DynNodeT));
    off_dyn_poly_t = put(polys_t.data(), polys_t.size() * sizeof(DynPolyT));
    off_dyn_nodes_l = put_layout(nodes.data(), nodes.size() * sizeof(DynNodeL));
    off_dyn_sel = put_layout(sel.data(), sel.size() * sizeof(DynSel));
    off_dyn_tile = put_layout(tiles.data(), tiles.size() * sizeof(DynTile));
    off_dyn_poly_l = put_layout(polys.data(), polys.size() * sizeof(DynPolyL));
  }
  // --- rangeofmotion-<ee>, optimised timings: per-node record templates (the pre-pass fills in the x-dependent part)
  for (int e = 0; e < n_ee && have_rom && timings; ++e) {
    std::vector<RomRec> rc(grid_rom.size());
    for (size_t k = 0; k < grid_rom.size(); ++k) {
      RomRec& R = rc[k];
      std::memset(&R, 0, sizeof(R));
      const PolyDesc& mp = mpoly[e][rom_motion[e][k].poly];
      R.tb = rom_base[k].t_local;
      R.iTb = 1.0 / base.durations[rom_base[k].poly];
      R.tm = rom_motion[e][k].t_local;
      R.iTm = mp.iT;
      R.q6 = 6 * rom_base[k].poly;
      R.xbase = mp.xbase;
      R.voff = row_ptr[row_rom[e] + 3 * k] - nnz_rom[e];
      R.meta = mp.meta;
      uint64_t slots = 0;
      for (int c = 0; c < 12; ++c) slots |= (uint64_t)(mp.cand[c] & 0xF) << (4 * c);
      R.slots[0] = (uint32_t)slots;
      R.slots[1] = (uint32_t)(slots >> 32);
    }
    off_rom_recs[e] = put(rc.data(), rc.size() * sizeof(RomRec));
  }
  // --- rangeofmotion-<ee>, fixed timings: per-node records shared by all ee, slices, per-(slice, polynomial) segments
  rom_slices.assign(n_ee, {});
  if (have_rom && !timings) {
    const int K = (int)grid_rom.size();
    std::vector<RomNode> nodes(K);
    for (int k = 0; k < K; ++k) {
      std::memset(&nodes[k], 0, sizeof(RomNode));
      nodes[k].t = grid_rom[k];
      nodes[k].tb = rom_base[k].t_local;
      nodes[k].iTb = 1.0 / base.durations[rom_base[k].poly];
      nodes[k].q6 = 6 * rom_base[k].poly;
    }
    for (int e = 0; e < n_ee; ++e) {
      const int row0 = row_rom[e];
      auto vals = [&](int k0, int k1) { return row_ptr[row0 + 3 * k1] - row_ptr[row0 + 3 * k0]; };
      auto segs = [&](int k0, int k1) {   // polynomials of the ee spline active in nodes [k0, k1)
        int n = 1;
        for (int k = k0 + 1; k < k1; ++k) n += rom_motion[e][k].poly != rom_motion[e][k - 1].poly;
        return n;
      };
      // Balanced runs: the fewest runs that respect the limits (<= 64 time nodes = lanes, the LDS image, kRomMaxSeg
      // polynomials), of (nearly) equal length -- K = 200 gives 4 x 50 rather than 64 + 64 + 64 + 8: the kernel has a
      // compile-time number of copy-out stores and pays the full count for a short tail run too.
      auto greedy = [&](int limit, std::vector<std::pair<int, int>>& out) {
        out.clear();
        for (int k0 = 0; k0 < K;) {
          int k1 = k0;
          while (k1 < K && k1 - k0 < limit && vals(k0, k1 + 1) <= kRomStage && segs(k0, k1 + 1) <= kRomMaxSeg) ++k1;
          if (k1 == k0) throw std::runtime_error("one time node exceeds the LDS staging capacity");
          out.push_back({k0, k1 - k0});
          k0 = k1;
        }
      };
      std::vector<std::pair<int, int>> runs, best;
      greedy(64, best);
      for (int limit = (K + (int)best.size() - 1) / (int)best.size(); limit < 64; ++limit) {
        greedy(limit, runs);   // the smallest run length that still needs no more runs
        if (runs.size() == best.size()) {
          best = runs;
          break;
        }
      }
      // t0 of every polynomial: the running sum Spline::GetSegmentID compares t against (spline.cc:52-57)
      std::vector<double> t0(motion[e].durations.size() + 1, 0.0);
      for (size_t q = 0; q < motion[e].durations.size(); ++q) t0[q + 1] = t0[q] + motion[e].durations[q];
      for (const auto& r : best) {
        RomSlice sl;
        sl.k0 = r.first;
        sl.cnt = r.second;
        sl.nvals = vals(r.first, r.first + r.second);
        // copy_out_fixed clamps its tail iterations to the last complete pair of the slice: a slice must hold one
        if (sl.nvals < 4) throw std::runtime_error("a time-node run with fewer than 4 Jacobian values cannot be staged");
        std::memset(sl.first, 255, sizeof(sl.first));
        std::vector<RomSeg> sg;
        for (int k = sl.k0; k < sl.k0 + sl.cnt; ++k) {
          const int q = rom_motion[e][k].poly;
          if (k > sl.k0 && q == rom_motion[e][k - 1].poly) continue;
          const PolyDesc& mp = mpoly[e][q];
          RomSeg S;
          std::memset(&S, 0, sizeof(S));
          S.t0 = t0[q];
          S.iTm = mp.iT;
          uint64_t slots = 0;
          for (int c = 0; c < 12; ++c) slots |= (uint64_t)(mp.cand[c] & 0xF) << (4 * c);
          S.slots[0] = (uint32_t)slots;
          S.slots[1] = (uint32_t)(slots >> 32);
          S.xbase = mp.xbase;
          S.meta = mp.meta;
          S.voff0 = row_ptr[row0 + 3 * k] - row_ptr[row0 + 3 * sl.k0];
          S.kfirst = k - sl.k0;
          S.node_vals = row_ptr[row0 + 3 * (k + 1)] - row_ptr[row0 + 3 * k];
          if (S.node_vals != 68 + 3 * (int)(mp.meta & 0xF)) throw std::runtime_error("rangeofmotion row lengths inconsistent");
          sl.first[sg.size()] = (uint8_t)(k - sl.k0);
          sg.push_back(S);
        }
        // every node of a segment has the same row lengths: voff = voff0 + (k - kfirst) * node_vals (checked)
        for (int k = sl.k0; k < sl.k0 + sl.cnt; ++k) {
          size_t si = 0;
          while (si + 1 < sg.size() && k - sl.k0 >= (int)sl.first[si + 1]) ++si;
          if (row_ptr[row0 + 3 * k] - row_ptr[row0 + 3 * sl.k0] != sg[si].voff0 + (k - sl.k0 - sg[si].kfirst) * sg[si].node_vals)
            throw std::runtime_error("rangeofmotion segment layout inconsistent");
        }
        for (int k = sl.k0; k < sl.k0 + sl.cnt; ++k) {   // which segment of this ee's slice node k reads: three bits per ee
          uint32_t si = 0;
          while (si + 1 < sg.size() && k - sl.k0 >= (int)sl.first[si + 1]) ++si;
          nodes[k].seg |= si << (3 * e);
        }
        sl.segs = put(sg.data(), sg.size() * sizeof(RomSeg));
        rom_slices[e].push_back(sl);
      }
    }
    off_rom_nodes = put(nodes.data(), nodes.size() * sizeof(RomNode));   // (after the loop: it fills RomNode::seg)
  }
  // --- optimised timings: polynomial tables, set-wide counts, global grid times
  if (timings) {
    PhaseTables& pt = phase_tables;
    std::memset(&pt, 0, sizeof(pt));
    auto dim_table = [&](const SplineLayout& sl) {
      std::vector<int> dim_of(sl.var_size, -1);
      for (int n = 0; n < sl.n_nodes; ++n)
        for (int dv = 0; dv < 2; ++dv)
          for (int d = 0; d < 3; ++d)
            if (sl.at(n, dv, d) >= 0) dim_of[sl.at(n, dv, d) - sl.var_offset] = d;
      return dim_of;
    };
    auto poly_table = [&](const SplineLayout& sl, const std::vector<PolyDesc>& pd, int n_changing) {
      std::vector<PhasePoly> out(pd.size());
      std::vector<int> dim_of = dim_table(sl);
      int in_phase = 0;
      for (size_t q = 0; q < pd.size(); ++q) {
        PhasePoly& pp = out[q];
        std::memset(&pp, 0, sizeof(pp));
        in_phase = (q > 0 && sl.poly_phase[q] == sl.poly_phase[q - 1]) ? in_phase + 1 : 0;
        int n_in = 0;
        for (size_t q2 = 0; q2 < pd.size(); ++q2) n_in += sl.poly_phase[q2] == sl.poly_phase[q];
        (void)n_changing;
        pp.phase = sl.poly_phase[q];
        pp.n_in_phase = n_in;
        pp.poly_in_phase = in_phase;
        pp.xbase = pd[q].xbase;
        pp.meta = pd[q].meta;
        std::memcpy(pp.cand, pd[q].cand, sizeof(pp.cand));
        const int before = (pd[q].meta & 0xF) ? pd[q].xbase - sl.var_offset : 0;
        for (int i = 0; i < before; ++i) {
          for (int r = 0; r < 3; ++r) {
            if (dim_of[i] != r) pp.base_ne[r]++;
            if (dim_of[i] == r) pp.base_eq[r]++;
          }
        }
        pp.base_all = (uint16_t)before;
      }
      return out;
    };
    for (int e = 0; e < n_ee; ++e) {
      pt.off_sched[e] = off_schedule[e];
      pt.n_phases[e] = schedule.n_phases[e];
      pt.t_total[e] = std::accumulate(schedule.phase_durations[e], schedule.phase_durations[e] + schedule.n_phases[e], 0.0);
      auto mp = poly_table(motion[e], mpoly[e], params.polys_per_swing);
      auto fp = poly_table(force[e], fpoly[e], params.polys_per_stance_force);
      pt.n_mpoly[e] = (int)mp.size();
      pt.n_fpoly[e] = (int)fp.size();
      pt.o_mpoly[e] = put(mp.data(), mp.size() * sizeof(PhasePoly));
      pt.o_fpoly[e] = put(fp.data(), fp.size() * sizeof(PhasePoly));
      std::vector<int> dm = dim_table(motion[e]), df = dim_table(force[e]);
      for (int r = 0; r < 3; ++r) {
        for (int v : dm) pt.mne[e][r] += v != r;
        for (int v : df) { pt.fne[e][r] += v != r; pt.feq[e][r] += v == r; }
      }
      pt.msize[e] = motion[e].var_size;
    }
    int sched_total = 0;
    for (int e = 0; e < n_ee; ++e) sched_total += schedule.n_phases[e] - 1;
    for (int r = 0; r < 3; ++r) {
      pt.len_ang[r] = 20 + sched_total;
      pt.len_lin[r] = 4 + sched_total;
      for (int e = 0; e < n_ee; ++e) {
        pt.len_ang[r] += pt.mne[e][r] + pt.fne[e][r];
        pt.len_lin[r] += pt.feq[e][r];
      }
      pt.node_vals += pt.len_ang[r] + pt.len_lin[r];
    }
    for (int e = 0; e < n_ee; ++e) {
      for (int r = 0; r < 3; ++r) {
        pt.rom_len[e][r] = 12 + (r == 0 ? 8 : 12) + pt.msize[e] + schedule.n_phases[e] - 1;
        pt.rom_node_vals[e] += pt.rom_len[e][r];
      }
      pt.row_rom[e] = row_rom[e];
      pt.nnz_rom[e] = nnz_rom[e];
      // rom_phase_kernel assembles whole expanded time nodes in LDS: one node must fit the 160 KB of a CU
      // (the dynamic set has the matching guard below; without it the batch is created and every evaluation fails to launch)
      if (have_rom && pt.rom_node_vals[e] > 160 * 128)
        throw std::runtime_error("optimised timings: a time node of rangeofmotion-" + std::to_string(e) + " has more than 20480 Jacobian values");
    }
    pt.o_tdyn = put(grid_dyn.data(), grid_dyn.size() * sizeof(double));
    pt.o_trom = put(grid_rom.data(), grid_rom.size() * sizeof(double));
    pt.k_dyn = dyn_set ? (int)grid_dyn.size() : 0;
    pt.k_rom = have_rom ? (int)grid_rom.size() : 0;
    pt.row_dyn = row_dyn;
    pt.nnz_dyn = nnz_dyn;
    pt.off_lin = off_base_lin;
    pt.off_ang = off_base_ang;
    pt.o_dyn_shared = off_dyn_shared;
    for (int e = 0; e < n_ee; ++e) {
      pt.o_rom_recs[e] = off_rom_recs[e];
      if (pt.n_mpoly[e] > kMaxPhasePolys || pt.n_fpoly[e] > kMaxPhasePolys) throw std::runtime_error("too many polynomials per ee spline for optimised timings");
    }
    if (const SetInfo* si = FindSet("totalduration-0")) {
      pt.row_total = si->offset;
      pt.nnz_total = si->nnz_offset;
    }
    // dynamic: byte offset of every ee value inside a time node's expanded rows, read off the CSR pattern of time
    // node 0 (the rows hold all variables of every ee set, so the ee part of the layout is the same at every node)
    if (dyn_set) {
      if (pt.node_vals > 8191) throw std::runtime_error("optimised timings: a time node of the dynamic set has more than 8191 Jacobian values");
      const int v0 = row_ptr[row_dyn];
      auto find = [&](int row, int col) -> int {
        const int32_t* b = col_idx.data() + row_ptr[row_dyn + row];
        const int32_t* e2 = col_idx.data() + row_ptr[row_dyn + row + 1];
        const int32_t* it = std::lower_bound(b, e2, col);
        if (it == e2 || *it != col) throw std::runtime_error("dynamic pattern lacks an expected column");
        return (int)(it - col_idx.data()) - v0;
      };
      std::vector<PhasePutM> pm_all;
      std::vector<PhasePutF> pf_all;
      for (int e = 0; e < n_ee; ++e) {
        const uint16_t trash = (uint16_t)(8 * (8 + e));   // base-ang entry of row AX, rewritten after the tiles
        std::vector<PhasePutM> pm(mpoly[e].size());
        std::vector<PhasePutF> pf(fpoly[e].size());
        for (size_t q = 0; q < mpoly[e].size(); ++q) {
          std::memset(&pm[q], 0, sizeof(PhasePutM));
          for (int c = 0; c < 12; ++c) {
            const int j = c / 3, d = c % 3, r1 = (d + 1) % 3, r2 = (d + 2) % 3;
            pm[q].off[j][2 * d] = pm[q].off[j][2 * d + 1] = trash;
            if (mpoly[e][q].cand[c] == 0xFFFF) continue;
            const int col = mpoly[e][q].xbase + (mpoly[e][q].cand[c] & 0xF);
            pm[q].off[j][2 * d] = (uint16_t)(8 * find(r1, col));
            pm[q].off[j][2 * d + 1] = (uint16_t)(8 * find(r2, col));
          }
          for (int j = 0; j < 4; ++j) pm[q].off[j][6] = pm[q].off[j][7] = trash;
        }
        for (size_t q = 0; q < fpoly[e].size(); ++q) {
          std::memset(&pf[q], 0, sizeof(PhasePutF));
          for (int c = 0; c < 12; ++c) {
            const int j = c / 3, d = c % 3, r1 = (d + 1) % 3, r2 = (d + 2) % 3;
            pf[q].off[j][3 * d] = pf[q].off[j][3 * d + 1] = pf[q].off[j][3 * d + 2] = trash;
            if (fpoly[e][q].cand[c] == 0xFFFF) continue;
            const int col = fpoly[e][q].xbase + (fpoly[e][q].cand[c] & 0xF);
            pf[q].off[j][3 * d] = (uint16_t)(8 * find(r1, col));
            pf[q].off[j][3 * d + 1] = (uint16_t)(8 * find(r2, col));
            pf[q].off[j][3 * d + 2] = (uint16_t)(8 * find(3 + d, col));
          }
          for (int j = 0; j < 4; ++j) pf[q].off[j][9] = pf[q].off[j][10] = pf[q].off[j][11] = trash;
        }
        pt.mput_base[e] = (int)pm_all.size();
        pt.fput_base[e] = (int)pf_all.size();
        pm_all.insert(pm_all.end(), pm.begin(), pm.end());
        pf_all.insert(pf_all.end(), pf.begin(), pf.end());
        pt.ee[e].ns = schedule.n_phases[e] - 1;
        for (int r = 0; r < 3; ++r) {
          pt.ee[e].dur_ang[r] = 8 * find(r, off_schedule[e]);
          pt.ee[e].dur_lin[r] = 8 * find(3 + r, off_schedule[e]);
        }
      }
      pt.n_mput = (int)pm_all.size();
      pt.n_fput = (int)pf_all.size();
      for (int e = 0; e < kMaxEE; ++e) {   // dummy records
        PhasePutM dm;
        PhasePutF df;
        for (int j = 0; j < 4; ++j) {
          for (int q = 0; q < 8; ++q) dm.off[j][q] = (uint16_t)(8 * (8 + e));
          for (int q = 0; q < 12; ++q) df.off[j][q] = (uint16_t)(8 * (8 + e));
        }
        pm_all.push_back(dm);
        pf_all.push_back(df);
      }
      pt.o_mput = put(pm_all.data(), pm_all.size() * sizeof(PhasePutM));
      pt.o_fput = put(pf_all.data(), pf_all.size() * sizeof(PhasePutF));
      for (int r = 1; r < 6; ++r) pt.dyn_row_off[r - 1] = 8u * (uint32_t)(row_ptr[row_dyn + r] - v0);
      if (params.polys_per_swing > 15 || params.polys_per_stance_force > 15)
        throw std::runtime_error("optimised timings: at most 15 polynomials per phase");
    }
    // the pattern builder and these closed forms must agree
    if (dyn_set && dyn_set->nnz != pt.node_vals * (int)grid_dyn.size()) throw std::runtime_error("dynamic row lengths inconsistent");
    for (int e = 0; e < n_ee && have_rom; ++e)
      if (FindSet("rangeofmotion-" + std::to_string(e))->nnz != pt.rom_node_vals[e] * (int)grid_rom.size())
        throw std::runtime_error("rangeofmotion row lengths inconsistent");
    h.timings = 1;
    h.o_phase = put(&pt, sizeof(pt));
  }
  if (model.terrain_id == TWR_TERRAIN_CSV_GRID || model.terrain_id == TWR_TERRAIN_GRID_MAP) {
    if (!grid) throw std::runtime_error("gridded terrains need twr_structure_create_with_grid");
    if (grid->grid_map != (model.terrain_id == TWR_TERRAIN_GRID_MAP))
      throw std::runtime_error("the grid handle is of the other kind (CSV heights vs grid_map elevation layer)");
    h.grid_rows = grid->rows;
    h.grid_cols = grid->cols;
    h.grid_res = grid->res;
    h.grid_eps = grid->eps;
    h.grid_px = grid->pos_x;
    h.grid_py = grid->pos_y;
  }
  {  // trajectory sampling tables
    SampleTables st;
    std::memset(&st, 0, sizeof(st));
    st.n_base = (int)base.durations.size();
    st.o_bdur = put(base.durations.data(), base.durations.size() * sizeof(double));
    st.off_lin = off_base_lin;
    st.off_ang = off_base_ang;
    st.t_total = std::accumulate(base.durations.begin(), base.durations.end(), 0.0);  // spline.cc:118-123
    for (int e = 0; e < n_ee; ++e) {
      st.n_phases[e] = schedule.n_phases[e];
      st.contact0[e] = schedule.in_contact_at_start[e] != 0;
      st.n_mpoly[e] = (int)mpoly[e].size();
      st.n_fpoly[e] = (int)fpoly[e].size();
      st.o_phdur[e] = put(schedule.phase_durations[e], schedule.n_phases[e] * sizeof(double));
      st.o_mdur[e] = put(motion[e].durations.data(), motion[e].durations.size() * sizeof(double));
      st.o_fdur[e] = put(force[e].durations.data(), force[e].durations.size() * sizeof(double));
      st.o_mdesc[e] = put(mpoly[e].data(), mpoly[e].size() * sizeof(PolyDesc));
      st.o_fdesc[e] = put(fpoly[e].data(), fpoly[e].size() * sizeof(PolyDesc));
      if (st.n_mpoly[e] > kMaxPhasePolys || st.n_fpoly[e] > kMaxPhasePolys) st.n_base = -1;  // sampling unsupported
    }
    if (st.n_base > 2 * kMaxPhasePolys) st.n_base = -1;
    h.o_sample = put(&st, sizeof(st));
    // values-only evaluation of dynamic / rangeofmotion-* with one lane per time node (device_tables.h FlatNode): fixed
    // timings only -- with optimised timings the active polynomials depend on x and the phase kernels keep that path
    const bool rom_set = FindSet("rangeofmotion-0") != nullptr;
    const SetInfo* dyn_set = FindSet("dynamic");
    flat_items_rom.clear();
    flat_items_dyn.clear();
    flat_with_rom = false;
    size_t n_flat_polys = 0;
    for (int e = 0; e < n_ee; ++e) n_flat_polys += mpoly[e].size() + fpoly[e].size();
    // x is staged in LDS (16-bit byte offsets); window starts are 16-bit indices
    if (!timings && n_vars <= kFlatXCap && n_flat_polys < 65536 && (rom_set || dyn_set)) {
      flat_row_dyn = dyn_set ? dyn_set->offset : 0;
      std::vector<FlatPoly> fp;
      int first[2 * kMaxEE] = {0};   // spline s = 2 e (ee-motion_e), 2 e + 1 (ee-force_e): its first record in fp
      auto flat_polys = [&](const std::vector<PolyDesc>& polys, const std::vector<double>& durations) {
        double t0 = 0.0;   // the running sum Spline::GetSegmentID compares t against (spline.cc:52-57)
        for (size_t q = 0; q < polys.size(); ++q) {
          FlatPoly r;
          std::memset(&r, 0, sizeof(r));
          r.t0 = t0;
          r.iT = polys[q].iT;
          t0 += durations[q];
          const bool shared = (polys[q].meta >> 16) & 1;
          for (int c = 0; c < 12; ++c) {
            const int src = shared && c >= 6 && c < 9 ? c - 6 : c;   // stance: p1 is the same variable as p0
            const int sl = polys[q].cand[src] & 0xF;
            r.off[c] = (uint16_t)(sl != 0xF ? 8 * (2 + polys[q].xbase + sl) : 0);
          }
          fp.push_back(r);
        }
      };
      for (int e = 0; e < n_ee; ++e) {
        if (rom_set) flat_row_rom[e] = FindSet("rangeofmotion-" + std::to_string(e))->offset;
        first[2 * e] = (int)fp.size();
        flat_polys(mpoly[e], motion[e].durations);
        first[2 * e + 1] = (int)fp.size();
        flat_polys(fpoly[e], force[e].durations);
      }
      off_flat_polys = h.o_flat = put(fp.data(), fp.size() * sizeof(FlatPoly));
      // items: <= 64 consecutive time nodes whose active polynomials span <= kFlatWindow per spline.  On a COARSE grid (towr's
      // defaults: 0.1 / 0.08 s against polynomials of that length) the time nodes hardly share polynomials and the windows would
      // cut items of a few time nodes: such a grid is cut at 64 time nodes alone and its lanes fetch their own records
      // (FlatWork::gather; the indices stay relative to the item's first polynomials, 8 bits)
      auto flat_items = [&](const std::vector<double>& grid, const std::vector<TimeNode>& at_base, const std::vector<std::vector<TimeNode>>& at_motion,
                            const std::vector<std::vector<TimeNode>>* at_force, std::vector<FlatItem>& items) {
        std::vector<FlatNode> fn(grid.size());
        auto poly_at = [&](int s, size_t k) { return (s & 1) ? (*at_force)[s >> 1][k].poly : at_motion[s >> 1][k].poly; };
        const int step = at_force ? 1 : 2;   // range of motion: the ee-motion splines only
        auto cut = [&](int window, bool gather) {
          items.clear();
          size_t k0 = 0;
          while (k0 < grid.size()) {
            size_t k1 = k0 + 1;
            auto fits = [&](size_t k) {
              for (int s = 0; s < 2 * n_ee; s += step)
                if (poly_at(s, k) - poly_at(s, k0) >= window) return false;
              return true;
            };
            while (k1 < grid.size() && k1 - k0 < 64 && fits(k1)) ++k1;
            FlatItem it;
            it.k0 = (int)k0;
            it.cnt = (int)(k1 - k0);
            it.gather = gather;
            for (int s = 0; s < 2 * n_ee; s += step) {
              const int lo = poly_at(s, k0), hi = poly_at(s, k1 - 1);
              it.start[s >> 2] |= (uint64_t)(first[s] + lo) << (16 * (s & 3));
              it.count |= (uint64_t)(hi - lo + 1) << (8 * s);
            }
            items.push_back(it);
            for (size_t k = k0; k < k1; ++k) {
              std::memset(&fn[k], 0, sizeof(FlatNode));
              fn[k].t = grid[k];
              fn[k].tb = at_base[k].t_local;
              fn[k].iTb = 1.0 / base.durations[at_base[k].poly];
              fn[k].q6 = 6 * at_base[k].poly;
              for (int e = 0; e < n_ee; ++e) {
                fn[k].qm[e] = (uint8_t)(at_motion[e][k].poly - at_motion[e][k0].poly);
                if (at_force) fn[k].qf[e] = (uint8_t)((*at_force)[e][k].poly - (*at_force)[e][k0].poly);
              }
            }
            k0 = k1;
          }
        };
        cut(kFlatWindow, false);
        // (an item costs about the same whatever it holds, fetching the records per lane a quarter more: the windows stay while
        // they cut at most a quarter more items than 64 time nodes each would)
        if (4 * items.size() > 5 * ((grid.size() + 63) / 64)) cut(256, true);
        return put(fn.data(), fn.size() * sizeof(FlatNode));
      };
      // coinciding grids (the BASELINE configurations choose one dt for both; towr's defaults are 0.1 / 0.08 s): the "dynamic"
      // items evaluate the range-of-motion rows of their time nodes as well -- same base point, rotation and ee positions
      flat_with_rom = rom_set && dyn_set && grid_rom == grid_dyn;
      if (rom_set && !flat_with_rom) off_flat_rom = flat_items(grid_rom, rom_base, rom_motion, nullptr, flat_items_rom);
      if (dyn_set) off_flat_dyn = flat_items(grid_dyn, dyn_base, dyn_motion, &dyn_force, flat_items_dyn);
    }
  }
  h.mass = model.mass; h.gravity = model.gravity; h.mu = model.friction; h.flat_height = model.flat_height;
  // BuildInertiaTensor (single_rigid_body_dynamics.cc:36-44): off-diagonals are the negated products of inertia
  const double* I = model.inertia;  // Ixx,Iyy,Izz,Ixy,Ixz,Iyz
  h.Ib[0] = I[0]; h.Ib[1] = -I[3]; h.Ib[2] = -I[4]; h.Ib[3] = I[1]; h.Ib[4] = -I[5]; h.Ib[5] = I[2];
  blob.resize(sizeof(DevStruct) + body.size());
  std::memcpy(blob.data(), &h, sizeof(h));
  if (!body.empty()) std::memcpy(blob.data() + sizeof(DevStruct), body.data(), body.size());
  blob.resize((blob.size() + 15) / 16 * 16);
}

const SetInfo* Structure::FindSet(const std::string& name) const {
  for (const SetInfo& s : con_sets)
    if (s.name == name) return &s;
  return nullptr;
}

void Structure::Build() {
  BuildVariables();
  BuildTimeTables();
  BuildPattern();
  PackBlob();
}
void Structure::BuildSizes() {
  BuildVariables();
  BuildTimeTables();
  BuildPattern();
}

// ------------------------------------------------------------------ terrain height (host, setup only)
// HeightMap::GetHeight of the example terrains (src/height_map_examples.cc:35-197,
// include/towr/terrain/examples/height_map_examples.h:45-166).
// HeightMapFromCSV::GetHeight (include/towr/terrain/height_map_from_csv.h:29-37).  static_cast<size_t>(x / res)
// truncates toward zero; a quotient <= -1 wraps to a huge size_t in the reference (formally undefined) and
// fails the range check -- here a signed cell index that is invalid when negative.
// Grid::GetHeight (include/towr/terrain/grid_height_map.h:29-46) over grid_map's published
// atPosition(INTER_LINEAR) (restated in the device code, kernels.hip gridmap_sample; this host copy serves the
// initial guess only): bilinear in double, rounded to float; nearest cell in the border band; FLT_MAX outside.
static float GridMapSample(const TerrainGrid& g, double x, double y) {
  const int sx = g.rows, sy = g.cols;
  const double res = g.res, lx = sx * res, ly = sy * res;
  auto in = [&](long i, long j) { return i >= 0 && j >= 0 && i < sx && j < sy; };
  auto at = [&](long i, long j) { return g.elevation[(size_t)i + (size_t)j * (size_t)sx]; };
  const long i0 = (long)(-((x - 0.5 * lx - g.pos_x) / res)), j0 = (long)(-((y - 0.5 * ly - g.pos_y) / res));
  const double tx = g.pos_x + 0.5 * lx - x, ty = g.pos_y + 0.5 * ly - y;
  const bool inside = tx >= 0.0 && ty >= 0.0 && tx < lx && ty < ly;
  const double cx0 = g.pos_x + 0.5 * lx - 0.5 * res - res * (double)i0, cy0 = g.pos_y + 0.5 * ly - 0.5 * res - res * (double)j0;
  const long ia = x >= cx0 ? i0 : i0 + 1, ja = y >= cy0 ? j0 : j0 + 1, ib = ia - 1, jb = ja - 1;
  if (in(ia, ja) && in(ib, jb)) {
    const double px = g.pos_x + 0.5 * lx - 0.5 * res - res * (double)ia, py = g.pos_y + 0.5 * ly - 0.5 * res - res * (double)ja;
    const double rx = (x - px) / res, ry = (y - py) / res, fx = 1.0 - rx, fy = 1.0 - ry;
    return (float)(at(ia, ja) * fx * fy + at(ib, ja) * rx * fy + at(ia, jb) * fx * ry + at(ib, jb) * rx * ry);
  }
  if (inside && in(i0, j0)) return at(i0, j0);
  return std::numeric_limits<float>::max();
}

double TerrainGrid::Height(double x, double y) const {
  if (grid_map) return GridMapSample(*this, x, y);
  const long xc = (long)(x / res), yc = (long)(y / res);
  if (xc < 0 || yc < 0 || xc >= cols || yc >= rows) return 0.0;
  return heights[(size_t)yc * cols + xc];
}

double TerrainHeightHost(const twr_model& m, const TerrainGrid* grid, double x, double y) {
  switch (m.terrain_id) {
    case TWR_TERRAIN_CSV_GRID:
    case TWR_TERRAIN_GRID_MAP:
      if (!grid) throw std::runtime_error("gridded terrain without a grid");
      return grid->Height(x, y);
    case TWR_TERRAIN_FLAT: return m.flat_height;
    case TWR_TERRAIN_BLOCK: {
      const double start = 0.7, len = 3.5, height = 0.5, eps = 0.03, slope = height / eps;
      double h = 0.0;
      if (start <= x && x <= start + eps) h = slope * (x - start);
      if (start + eps <= x && x <= start + len) h = height;
      return h;
    }
    case TWR_TERRAIN_STAIRS: {
      double h = 0.0;
      if (x >= 1.0) h = 0.2;
      if (x >= 1.0 + 0.4) h = 0.4;
      if (x >= 1.0 + 0.4 + 1.0) h = 0.0;
      return h;
    }
    case TWR_TERRAIN_GAP: {
      const double gs = 1.0, w = 0.5, hh = 1.5, xc = gs + w / 2.0, ge = gs + w;
      const double a = (4 * hh) / (w * w), b = -(8 * hh * xc) / (w * w), c = -(hh * (w - 2 * xc) * (w + 2 * xc)) / (w * w);
      return (gs <= x && x <= ge) ? a * x * x + b * x + c : 0.0;
    }
    case TWR_TERRAIN_SLOPE: {
      const double ss = 1.0, up = 1.0, dn = 1.0, hc = 0.7, xd = ss + up, xf = xd + dn, sl = hc / up;
      double z = 0.0;
      if (x >= ss) z = sl * (x - ss);
      if (x >= xd) z = hc - sl * (x - xd);
      if (x >= xf) z = 0.0;
      return z;
    }
    case TWR_TERRAIN_CHIMNEY: {
      const double xs = 1.0, len = 1.5, ys = 0.5, sl = 3.0;
      return (xs <= x && x <= xs + len) ? sl * (y - ys) : 0.0;
    }
    case TWR_TERRAIN_CHIMNEY_LR: {
      const double xs = 0.5, len = 1.0, ys = 0.5, sl = 2, xe1 = xs + len, xe2 = xs + 2 * len;
      double z = 0.0;
      if (xs <= x && x <= xe1) z = sl * (y - ys);
      if (xe1 <= x && x <= xe2) z = -sl * (y + ys);
      return z;
    }
  }
  throw std::runtime_error("unknown terrain id");
}

// ------------------------------------------------------------------ initial guess
// NlpFormulation::Make{Base,Endeffector,Force}Variables (src/nlp_formulation.cc:95-181) with
// NodesVariables::SetByLinearInterpolation + GetValues (src/nodes_variables.cc:52-62,126-150):
// a variable shared by two nodes ends up with the value written for the later node.
void Structure::InitialGuess(const double* lin0, const double* ang0, const double* lin1, const double* ang1,
                             const double* ee0, double* x) const {
  auto interpolate = [&](const SplineLayout& s, int var_off_delta, const double* a, const double* b) {
    double dp[3], vel[3];
    for (int d = 0; d < 3; ++d) {
      dp[d] = b[d] - a[d];
      vel[d] = dp[d] / T;
    }
    for (int n = 0; n < s.n_nodes; ++n)
      for (int d = 0; d < 3; ++d) {
        int ip = s.at(n, 0, d), iv = s.at(n, 1, d);
        if (ip >= 0) x[ip + var_off_delta] = a[d] + n / static_cast<double>(s.n_nodes - 1) * dp[d];
        if (iv >= 0) x[iv + var_off_delta] = vel[d];
      }
  };
  for (int i = 0; i < n_vars; ++i) x[i] = 0.0;
  double fl[3] = {lin1[0], lin1[1], TerrainHeightHost(model, grid.get(), lin1[0], lin1[1]) - model.nominal_stance[0][2]};
  interpolate(base, off_base_lin, lin0, fl);
  interpolate(base, off_base_ang, ang0, ang1);
  for (int e = 0; e < n_ee; ++e) {
    double yaw = ang1[2];
    // GetRotationMatrixBaseToWorld((0,0,yaw)) * nominal stance (nlp_formulation.cc:141-148)
    double cz = std::cos(yaw), sz = std::sin(yaw), c0 = std::cos(0.0), s0 = std::sin(0.0);
    const double* nb = model.nominal_stance[e];
    double R[3][3] = {{c0 * cz, cz * s0 * s0 - c0 * sz, s0 * sz + c0 * cz * s0},
                      {c0 * sz, c0 * cz + s0 * s0 * sz, c0 * s0 * sz - cz * s0},
                      {-s0, c0 * s0, c0 * c0}};
    double fe[3];
    for (int i = 0; i < 3; ++i) fe[i] = lin1[i] + (R[i][0] * nb[0] + R[i][1] * nb[1] + R[i][2] * nb[2]);
    fe[2] = TerrainHeightHost(model, grid.get(), fe[0], fe[1]);
    interpolate(motion[e], 0, ee0 + 3 * e, fe);
  }
  for (int e = 0; e < n_ee; ++e) {
    double f[3] = {0.0, 0.0, model.mass * model.gravity / n_ee};
    interpolate(force[e], 0, f, f);
  }
  if (timings)  // PhaseDurations::GetValues (src/phase_durations.cc:66-75): the given durations but the last
    for (int e = 0; e < n_ee; ++e)
      for (int i = 0; i < schedule.n_phases[e] - 1; ++i) x[off_schedule[e] + i] = schedule.phase_durations[e][i];
}

// ------------------------------------------------------------------ trajectory sampling
// fpowr::GetTrajectory (fpowr/include/fpowr/footstep_plan_extractor.h:19-53): `while (t <= T + 1e-5) { ...; t += dt; }`
int Structure::SampleCount(double dt) const {
  if (!(dt > 0)) throw std::runtime_error("dt must be positive");
  const double Tt = std::accumulate(base.durations.begin(), base.durations.end(), 0.0);
  int n = 0;
  for (double t = 0.0; t <= Tt + 1e-5; t += dt) {
    if (++n > 10000000) throw std::runtime_error("too many samples");
  }
  return n;
}

// ------------------------------------------------------------------ variable bounds
// NodesVariables::AddStartBound / AddFinalBound as called by NlpFormulation::MakeBaseVariables and
// MakeEndeffectorVariables (src/nlp_formulation.cc:109-122,151; src/nodes_variables.cc:152-181) with the
// bounded dimensions of src/parameters.cc:65-69.  A bound on a node value that is not an optimisation
// variable is silently dropped (AddBound only scans existing variables).
void Structure::VariableBounds(const double* init_base, const double* final_base, const double* ee0, double* lower,
                               double* upper) const {
  const double inf = 1e20;  // ifopt::NoBound
  for (int i = 0; i < n_vars; ++i) {
    lower[i] = -inf;
    upper[i] = inf;
  }
  auto fix = [&](const SplineLayout& s, int delta, int node, int deriv, int dim, double val) {
    int i = s.at(node, deriv, dim);
    if (i >= 0) lower[i + delta] = upper[i + delta] = val;
  };
  const int last = base.n_nodes - 1;
  for (int d = 0; d < 3; ++d) {
    fix(base, off_base_lin, 0, 0, d, init_base[d]);
    fix(base, off_base_lin, 0, 1, d, init_base[3 + d]);
    if (d != 2) fix(base, off_base_lin, last, 0, d, final_base[d]);  // bounds_final_lin_pos_ = {X,Y}
    fix(base, off_base_lin, last, 1, d, final_base[3 + d]);
    fix(base, off_base_ang, 0, 0, d, init_base[6 + d]);
    fix(base, off_base_ang, 0, 1, d, init_base[9 + d]);
    fix(base, off_base_ang, last, 0, d, final_base[6 + d]);
    fix(base, off_base_ang, last, 1, d, final_base[9 + d]);
    for (int e = 0; e < n_ee; ++e) fix(motion[e], 0, 0, 0, d, ee0[3 * e + d]);
  }
  if (timings)  // PhaseDurations::GetBounds with Parameters::bound_phase_duration_ (parameters.cc:52)
    for (int e = 0; e < n_ee; ++e)
      for (int i = 0; i < schedule.n_phases[e] - 1; ++i) {
        lower[off_schedule[e] + i] = 0.2;
        upper[off_schedule[e] + i] = 1.0;
      }
}

// ------------------------------------------------------------------ presets
// RobotModel(Robot) (src/robot_model.cc:41-68) with the constants of
// include/towr/models/examples/{monoped,biped,hyq,anymal}_model.h and models/go1/go1_model.h.
void ModelPreset(int robot, int terrain, twr_model* m) {
  std::memset(m, 0, sizeof(*m));
  auto quad = [&](double xn, double yn, double zn) {
    double s[4][3] = {{xn, yn, zn}, {xn, -yn, zn}, {-xn, yn, zn}, {-xn, -yn, zn}};  // LF RF LH RH
    std::memcpy(m->nominal_stance, s, sizeof(s));
  };
  auto set = [&](int n_ee, double mass, double ixx, double iyy, double izz, double ixy, double ixz, double iyz,
                 double dx, double dy, double dz) {
    m->n_ee = n_ee;
    m->mass = mass;
    double I[6] = {ixx, iyy, izz, ixy, ixz, iyz};
    std::memcpy(m->inertia, I, sizeof(I));
    m->max_dev[0] = dx; m->max_dev[1] = dy; m->max_dev[2] = dz;
  };
  switch (robot) {
    case TWR_ROBOT_MONOPED:
      set(1, 20, 1.2, 5.5, 6.0, 0.0, -0.2, -0.01, 0.25, 0.15, 0.2);
      m->nominal_stance[0][2] = -0.58;
      break;
    case TWR_ROBOT_BIPED:
      set(2, 20, 1.209, 5.583, 6.056, 0.005, -0.190, -0.012, 0.25, 0.15, 0.15);
      m->nominal_stance[0][1] = 0.20;  m->nominal_stance[0][2] = -0.65;
      m->nominal_stance[1][1] = -0.20; m->nominal_stance[1][2] = -0.65;
      break;
    case TWR_ROBOT_HYQ:
      set(4, 83, 4.26, 8.97, 9.88, -0.0063, 0.193, 0.0126, 0.25, 0.20, 0.10);
      quad(0.31, 0.29, -0.58);
      break;
    case TWR_ROBOT_ANYMAL:
      set(4, 29.5, 0.946438, 1.94478, 2.01835, 0.000938112, -0.00595386, -0.00146328, 0.15, 0.1, 0.10);
      quad(0.34, 0.19, -0.42);
      break;
    case TWR_ROBOT_GO1:
      set(4, 12.84, 0.0168128557, 0.063009565, 0.0716547275, -0.0002296769, -0.0002945293, -0.0000418731, 0.16, 0.12, 0.06);
      quad(0.1881, 0.04675 + 0.08, -0.3);
      break;
    default: throw std::runtime_error("unknown robot id");
  }
  if (terrain < TWR_TERRAIN_FLAT || terrain > TWR_TERRAIN_GRID_MAP) throw std::runtime_error("unknown terrain id");
  m->terrain_id = terrain;
  m->gravity = 9.80665;     // dynamic_model.cc:37
  m->friction = 0.5;        // height_map.h:136
  m->force_limit = 1000.0;  // parameters.cc:48
  m->flat_height = 0.0;
}

// ------------------------------------------------------------------ gait generator
namespace {
struct Stride {
  std::vector<double> times;
  std::vector<unsigned> contacts;  // bit e set = ee e in contact
};
enum G { Stand = 0, Flight, Walk1, Walk2, Walk2E, Run2, Run2E, Run1, Run1E, Run3, Run3E, Hop1, Hop1E, Hop2, Hop3, Hop3E, Hop5, Hop5E };

Stride DropTransition(Stride s) {  // GaitGenerator::RemoveTransition (gait_generator.cc:131-144)
  double last = s.times.back();
  s.times.pop_back();
  s.times.back() += last;
  s.contacts.pop_back();
  return s;
}
// quadruped contact states, ee bits LF=1 RF=2 LH=4 RH=8 (quadruped_gait_generator.cc:39-74)
constexpr unsigned II = 0, PI = 4, bI = 8, IP = 1, Ib = 2, Pb = 4 | 2, bP = 8 | 1, BI = 12, IB = 3, PP = 4 | 1, bb = 8 | 2,
                   Bb = 12 | 2, BP = 12 | 1, bB = 8 | 3, PB = 4 | 3, BB = 15;
Stride QuadStride(int g) {  // quadruped_gait_generator.cc:89-366
  switch (g) {
    case Stand: return {{0.3}, {BB}};
    case Flight: return {{0.3}, {Bb}};
    case Walk1: return {{0.3, 0.2, 0.3, 0.2, 0.3, 0.2, 0.3, 0.2}, {bB, BB, Bb, BB, PB, BB, BP, BB}};
    case Walk2: return {{0.25, 0.13, 0.25, 0.13, 0.25, 0.13, 0.25, 0.13}, {bB, bb, Bb, Pb, PB, PP, BP, bP}};
    case Walk2E: return DropTransition(QuadStride(Walk2));
    case Run1: return {{0.3, 0.2, 0.3, 0.2}, {bP, BB, Pb, BB}};
    case Run2: return {{0.4, 0.1, 0.4, 0.1}, {bP, II, Pb, II}};
    case Run2E: return {{0.4}, {bP}};
    case Run3: return {{0.3, 0.1, 0.3, 0.1}, {PP, II, bb, II}};
    case Run3E: return {{0.3}, {PP}};
    case Hop1: return {{0.3, 0.1, 0.3, 0.1}, {BI, II, IB, II}};
    case Hop1E: return {{0.3}, {BI}};
    case Hop2: return {{0.3, 0.4, 0.3}, {BB, II, BB}};
    case Hop3: return {{0.2, 0.3, 0.2, 0.2, 0.2, 0.3, 0.2, 0.2}, {Bb, BI, BP, bP, bB, IB, PB, Pb}};
    case Hop3E: return DropTransition(QuadStride(Hop3));
    case Hop5: return {{0.1, 0.2, 0.1, 0.1, 0.2, 0.1}, {Bb, BB, IP, Bb, BB, IP}};
  }
  throw std::runtime_error("quadruped gait not implemented");
}
Stride BipedStride(int g) {  // biped_gait_generator.cc:64-226, bits L=1 R=2
  const unsigned I = 0, P = 1, b = 2, B = 3;
  switch (g) {
    case Stand: return {{0.2}, {B}};
    case Flight: return {{0.5}, {I}};
    case Walk1: case Walk2: return {{0.3, 0.05, 0.3, 0.05}, {b, B, P, B}};
    case Run1: case Run3: return {{0.15, 0.4, 0.15 + 0.15, 0.4, 0.15}, {b, I, P, I, b}};
    case Hop1: return {{0.15, 0.5, 0.15}, {B, I, B}};
    case Hop2: return {{0.15, 0.4, 0.15}, {b, I, b}};
    case Hop3: return {{0.2, 0.2, 0.2}, {P, I, P}};
    case Hop5: return {{0.2, 0.3, 0.2, 0.2}, {P, I, b, B}};
  }
  throw std::runtime_error("biped gait not implemented");
}
Stride MonoStride(int g) {  // monoped_gait_generator.cc:50-120
  switch (g) {
    case Stand: return {{0.5}, {1}};
    case Flight: return {{0.5}, {0}};
    case Hop1: return {{0.3, 0.3}, {1, 0}};
    case Hop2: return {{0.2, 0.3}, {1, 0}};
  }
  throw std::runtime_error("monoped gait not implemented");
}
std::vector<int> ComboGaits(int n_ee, int combo) {
  static const std::vector<int> quad[5] = {{Stand, Walk2, Walk2, Walk2, Walk2E, Stand},   // quadruped_gait_generator.cc:76-87
                                           {Stand, Run2, Run2, Run2, Run2E, Stand},
                                           {Stand, Run3, Run3, Run3, Run3E, Stand},
                                           {Stand, Hop1, Hop1, Hop1, Hop1E, Stand},
                                           {Stand, Hop3, Hop3, Hop3, Hop3E, Stand}};
  static const std::vector<int> biped[5] = {{Stand, Walk1, Walk1, Walk1, Walk1, Stand},   // biped_gait_generator.cc:51-62
                                            {Stand, Run1, Run1, Run1, Run1, Stand},
                                            {Stand, Hop1, Hop1, Hop1, Stand},
                                            {Stand, Hop1, Hop2, Hop2, Stand},
                                            {Stand, Hop5, Hop5, Hop5, Stand}};
  static const std::vector<int> mono[5] = {{Stand, Hop1, Hop1, Hop1, Hop1, Stand},        // monoped_gait_generator.cc:37-48
                                           {Stand, Hop1, Hop1, Hop1, Stand},
                                           {Stand, Hop1, Hop1, Hop1, Hop1, Stand},
                                           {Stand, Hop2, Hop2, Hop2, Stand},
                                           {Stand, Hop2, Hop2, Hop2, Hop2, Hop2, Stand}};
  if (combo < 0 || combo > 4) throw std::runtime_error("combo must be 0..4");
  if (n_ee == 1) return mono[combo];
  if (n_ee == 2) return biped[combo];
  if (n_ee == 4) return quad[combo];
  throw std::runtime_error("no gait generator for this leg count");  // gait_generator.cc:43-52
}
}  // namespace

void GaitCombo(int n_ee, int combo, double t_total, double swing_scale, twr_schedule* out) {
  std::vector<double> times;
  std::vector<unsigned> contacts;
  for (int g : ComboGaits(n_ee, combo)) {  // SetGaits (gait_generator.cc:113-129)
    Stride s = n_ee == 1 ? MonoStride(g) : n_ee == 2 ? BipedStride(g) : QuadStride(g);
    times.insert(times.end(), s.times.begin(), s.times.end());
    contacts.insert(contacts.end(), s.contacts.begin(), s.contacts.end());
  }
  const unsigned all = (1u << n_ee) - 1;
  for (size_t i = 0; i < times.size(); ++i)
    if (contacts[i] != all) times[i] *= swing_scale;  // candidate enumeration knob (1.0 = reference)
  std::memset(out, 0, sizeof(*out));
  out->n_ee = n_ee;
  // GetPhaseDurations() (gait_generator.cc:76-105)
  std::vector<std::vector<double>> foot(n_ee);
  std::vector<double> acc(n_ee, 0.0);
  for (size_t ph = 0; ph + 1 < contacts.size(); ++ph)
    for (int e = 0; e < n_ee; ++e) {
      acc[e] += times[ph];
      bool cur = (contacts[ph] >> e) & 1, nxt = (contacts[ph + 1] >> e) & 1;
      if (cur != nxt) {
        foot[e].push_back(acc[e]);
        acc[e] = 0.0;
      }
    }
  for (int e = 0; e < n_ee; ++e) foot[e].push_back(acc[e] + times.back());
  for (int e = 0; e < n_ee; ++e) {
    if (foot[e].size() > TWR_MAX_PHASES) throw std::runtime_error("too many phases");
    // GetNormalizedPhaseDurations + GetPhaseDurations(T, ee) (gait_generator.cc:54-74)
    double total = std::accumulate(foot[e].begin(), foot[e].end(), 0.0);
    out->n_phases[e] = (int)foot[e].size();
    out->in_contact_at_start[e] = (contacts.front() >> e) & 1;  // IsInContactAtStart :107-111
    for (size_t i = 0; i < foot[e].size(); ++i) out->phase_durations[e][i] = (foot[e][i] / total) * t_total;
  }
}

LayoutShare ShareLayoutTables(const std::vector<const Structure*>& structs) {
  struct Seen {
    const char* bytes;
    uint32_t n;
    LayoutShare::Ref ref;
  };
  auto hash = [](const char* p, size_t n) {   // FNV-1a over 8-byte words (a bucket key only: equality is decided by memcmp)
    uint64_t h = 1469598103934665603ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      std::memcpy(&w, p + i, 8);
      h = (h ^ w) * 1099511628211ull;
      h ^= h >> 29;
    }
    for (; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
    return h;
  };
  LayoutShare out;
  out.of.resize(structs.size());
  std::unordered_map<uint64_t, std::vector<Seen>> by_hash;
  for (size_t i = 0; i < structs.size(); ++i) {
    const Structure& S = *structs[i];
    for (const Structure::TableRef& tr : S.dyn_layout_tables) {
      if ((size_t)tr.off + tr.bytes > S.blob.size()) throw std::runtime_error("layout table outside its blob");
      const char* src = S.blob.data() + tr.off;
      std::vector<Seen>& bucket = by_hash[hash(src, tr.bytes) ^ tr.bytes];
      LayoutShare::Ref ref{(int)i, tr.off};
      bool found = false;
      for (const Seen& sn : bucket)
        if (sn.n == tr.bytes && std::memcmp(sn.bytes, src, tr.bytes) == 0) {
          ref = sn.ref;
          found = true;
          break;
        }
      if (!found) {
        bucket.push_back({src, tr.bytes, ref});
        out.bytes_distinct += tr.bytes;
      }
      out.bytes_built += tr.bytes;
      out.of[i].push_back(ref);
    }
  }
  return out;
}

}  // namespace twr
