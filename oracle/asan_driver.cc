// TEST INFRASTRUCTURE ONLY -- sanitizer driver ("-fsanitize=address,undefined build of the
// CPU oracle"; extended to the product's host-side structure builder, which contains no HIP).
// Built by `make -C oracle asan` with -fsanitize=address,undefined and run by tests/test_sanitizers.py: walks
// robots x gait combos x constraint-set masks (incl. optimised timings and baseMotion) through
//   twr::Structure::Build / InitialGuess / VariableBounds   (towr_amd/csrc/structure.cc)
//   orc_create / orc_eval / orc_bounds / orc_sample_trajectory (oracle/towr_oracle.cc)
// and cross-checks sizes and the CSR pattern of the two, so that every table write and every row of the
// pattern builders executes under the sanitizers.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../towr_amd/csrc/structure.h"
#include "towr_oracle.h"

static int fails = 0;
#define CHECK(cond, ...)                  \
  do {                                    \
    if (!(cond)) {                        \
      std::fprintf(stderr, __VA_ARGS__);  \
      std::fprintf(stderr, "\n");         \
      ++fails;                            \
    }                                     \
  } while (0)

static void one_case(int robot, int terrain, int combo, double T, int sets, double dt) {
  twr_model m;
  twr::ModelPreset(robot, terrain, &m);
  twr_schedule s;
  twr::GaitCombo(m.n_ee, combo, T, 1.0, &s);
  twr_params p;
  p.dt_dynamic = dt;
  p.dt_rom = dt * 0.8;
  p.duration_base_poly = 0.1;
  p.polys_per_swing = 2;
  p.polys_per_stance_force = 3;
  p.constraint_sets = sets;
  p.reserved_ = 0;
  p.dt_base_motion = 0.025;
  p.base_z_init = -m.nominal_stance[0][2];
  twr::Structure S;
  S.model = m;
  S.schedule = s;
  S.params = p;
  S.Build();

  std::vector<int> n_ph(m.n_ee), con(m.n_ee);
  std::vector<double> pd;
  for (int e = 0; e < m.n_ee; ++e) {
    n_ph[e] = s.n_phases[e];
    con[e] = s.in_contact_at_start[e];
    for (int i = 0; i < s.n_phases[e]; ++i) pd.push_back(s.phase_durations[e][i]);
  }
  orc_problem* P = orc_create(robot, terrain, m.n_ee, n_ph.data(), pd.data(), con.data(), p.dt_dynamic, p.dt_rom,
                              p.duration_base_poly, p.polys_per_swing, p.polys_per_stance_force, m.force_limit, sets,
                              p.dt_base_motion, p.base_z_init, nullptr, 0, 0);
  CHECK(P != nullptr, "orc_create failed (robot %d combo %d sets %d)", robot, combo, sets);
  if (!P) return;
  CHECK(orc_n_vars(P) == S.n_vars, "n_vars %d vs %d", orc_n_vars(P), S.n_vars);
  CHECK(orc_n_rows(P) == S.n_rows, "n_rows %d vs %d", orc_n_rows(P), S.n_rows);

  // initial guess of both, then one full callback of the oracle on it
  std::vector<double> x(S.n_vars), xo(S.n_vars), ee(3 * m.n_ee);
  for (int e = 0; e < m.n_ee; ++e) {
    ee[3 * e] = m.nominal_stance[e][0];
    ee[3 * e + 1] = m.nominal_stance[e][1];
    ee[3 * e + 2] = 0.0;
  }
  const double z = -m.nominal_stance[0][2];
  const double lin0[3] = {0, 0, z}, ang0[3] = {0, 0, 0}, lin1[3] = {1.5, 0.1, z}, ang1[3] = {0, 0, 0.2};
  S.InitialGuess(lin0, ang0, lin1, ang1, ee.data(), x.data());
  orc_initial_guess(P, lin0, ang0, lin1, ang1, ee.data(), xo.data());
  for (int i = 0; i < S.n_vars; ++i) CHECK(std::fabs(x[i] - xo[i]) <= 1e-12 * (1 + std::fabs(xo[i])), "x0[%d] %g vs %g", i, x[i], xo[i]);

  const int nnz = orc_eval(P, x.data(), nullptr, nullptr, nullptr, nullptr);
  CHECK(nnz == S.nnz, "nnz %d vs %d (robot %d combo %d sets %d)", nnz, S.nnz, robot, combo, sets);
  std::vector<double> g(S.n_rows), v(nnz), lo(S.n_rows), up(S.n_rows);
  std::vector<int> rp(S.n_rows + 1), ci(nnz);
  orc_eval(P, x.data(), g.data(), rp.data(), ci.data(), v.data());
  if (nnz == S.nnz) {
    for (int r = 0; r <= S.n_rows; ++r) CHECK(rp[r] == S.row_ptr[r], "row_ptr[%d]", r);
    for (int i = 0; i < nnz; ++i) CHECK(ci[i] == S.col_idx[i], "col_idx[%d]", i);
  }
  for (double gv : g) CHECK(std::isfinite(gv), "non-finite g");
  orc_bounds(P, lo.data(), up.data());
  for (int r = 0; r < S.n_rows; ++r) CHECK(lo[r] == S.lower[r] && up[r] == S.upper[r], "bounds row %d", r);

  std::vector<double> vl(S.n_vars), vu(S.n_vars), ol(S.n_vars), ou(S.n_vars);
  double ib[12] = {0, 0, z, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fb[12] = {1.5, 0.1, z, 0, 0, 0, 0, 0, 0.2, 0, 0, 0};
  S.VariableBounds(ib, fb, ee.data(), vl.data(), vu.data());
  orc_variable_bounds(P, ib, fb, ee.data(), ol.data(), ou.data());
  for (int i = 0; i < S.n_vars; ++i) CHECK(vl[i] == ol[i] && vu[i] == ou[i], "variable bound %d", i);

  const int ns = orc_sample_trajectory(P, x.data(), 0.05, nullptr, 0);
  CHECK(ns == S.SampleCount(0.05), "sample count %d vs %d", ns, S.SampleCount(0.05));
  std::vector<double> traj((size_t)ns * (20 + 13 * m.n_ee));
  orc_sample_trajectory(P, x.data(), 0.05, traj.data(), ns);
  orc_destroy(P);
}

// twr::ShareLayoutTables (what twr_batch_create merges): T-siblings of a sweep at K = 200 share selector / polynomial-layout /
// tile tables, every reference points at bytes identical to the structure's own table, owners own themselves, and the
// byte counts add up.
static void sharing_case() {
  twr_model m;
  twr::ModelPreset(2, 4, &m);
  std::vector<twr::Structure> ss(11);
  for (int i = 0; i < 11; ++i) {
    const double T = i < 8 ? 1.2 + 0.2 * i : 2.0;
    const double scale = i < 8 ? 0.80 : 0.80 + 0.016 * (i - 7);
    twr::GaitCombo(m.n_ee, i == 10 ? 3 : 1, T, scale, &ss[i].schedule);
    twr_params p;
    p.dt_dynamic = p.dt_rom = T / (200 - 1.5);
    p.duration_base_poly = 0.1;
    p.polys_per_swing = 2;
    p.polys_per_stance_force = 3;
    p.constraint_sets = 27;
    p.reserved_ = 0;
    p.dt_base_motion = 0.025;
    p.base_z_init = -m.nominal_stance[0][2];
    ss[i].model = m;
    ss[i].params = p;
    ss[i].Build();
  }
  std::vector<const twr::Structure*> sp;
  for (const auto& s : ss) sp.push_back(&s);
  sp.push_back(&ss[3]);   // the same structure twice: shares everything with its first occurrence
  const twr::LayoutShare sh = twr::ShareLayoutTables(sp);
  int64_t built = 0, distinct = 0;
  for (size_t i = 0; i < sp.size(); ++i) {
    const auto& tabs = sp[i]->dyn_layout_tables;
    CHECK(sh.of[i].size() == tabs.size() && !tabs.empty(), "sharing: table count of structure %zu", i);
    for (size_t t = 0; t < tabs.size(); ++t) {
      const twr::LayoutShare::Ref r = sh.of[i][t];
      CHECK(r.owner >= 0 && r.owner <= (int)i, "sharing: owner after its user");
      const twr::Structure& O = *sp[r.owner];
      CHECK((size_t)r.off + tabs[t].bytes <= O.blob.size() && std::memcmp(O.blob.data() + r.off, sp[i]->blob.data() + tabs[t].off, tabs[t].bytes) == 0,
            "sharing: structure %zu table %zu reads other bytes", i, t);
      built += tabs[t].bytes;
      if (r.owner == (int)i && r.off == tabs[t].off) distinct += tabs[t].bytes;
    }
  }
  CHECK(built == sh.bytes_built && distinct == sh.bytes_distinct, "sharing: byte counts %lld / %lld vs %lld / %lld", (long long)built,
        (long long)distinct, (long long)sh.bytes_built, (long long)sh.bytes_distinct);
  CHECK(sh.bytes_distinct < sh.bytes_built * 85 / 100 && sh.bytes_distinct > sh.bytes_built / 3, "sharing: %lld of %lld bytes distinct",
        (long long)sh.bytes_distinct, (long long)sh.bytes_built);
  for (size_t t = 0; t < sh.of[11].size(); ++t) CHECK(sh.of[11][t].owner <= 3, "sharing: a repeated structure owns a table");
}

// the store policy of a batch (structure.h): the measured cases of DESIGN 6.R4
static void policy_case() {
  const int64_t per = 859216;   // bytes per callback of a K = 200 quadruped candidate
  CHECK(twr::StreamNonTemporal(1024, 1024, 1024 * per), "policy: the 1024-candidate sweep streams");
  CHECK(twr::StreamNonTemporal(384, 384, 384 * per), "policy: 384 candidates stream");
  CHECK(!twr::StreamNonTemporal(256, 256, 256 * per), "policy: 256 candidates (220 MB) keep plain stores");
  CHECK(!twr::StreamNonTemporal(64, 64, 64 * per), "policy: 64 candidates keep plain stores");
  CHECK(!twr::StreamNonTemporal(1, 8192, 8192 * per), "policy: one structure for 8192 problems keeps plain stores");
  CHECK(!twr::StreamNonTemporal(1024, 4096, 4096 * per), "policy: four problems per structure keep plain stores");
  CHECK(twr::StreamNonTemporal(1024, 2048, 2048 * per), "policy: two problems per structure stream");
  // the threshold is the device's memory-side cache, by architecture name: MI355X / MI300X 256 MB, a CPX
  // partition its share, an unknown device eight times its L2 -- where a 384-candidate sweep keeps plain stores
  CHECK(twr::MemorySideCacheBytes("gfx950:sramecc+:xnack-", 4 << 20) == ((int64_t)256 << 20), "cache: gfx950");
  CHECK(twr::MemorySideCacheBytes("gfx942", 4 << 20, 8) == ((int64_t)32 << 20), "cache: a CPX partition of gfx942");
  CHECK(twr::MemorySideCacheBytes("gfx90a", 8 << 20) == ((int64_t)64 << 20), "cache: unknown architecture");
  CHECK(twr::StreamNonTemporal(128, 128, 128 * per, twr::MemorySideCacheBytes("gfx90a", 8 << 20)), "policy: follows the device's cache");
  CHECK(!twr::StreamNonTemporal(256, 256, 256 * per, twr::MemorySideCacheBytes("gfx950", 4 << 20)), "policy: MI355X as measured");
}

int main() {
  sharing_case();
  policy_case();
  int cases = 0;
  const int masks[] = {27, 63, 127, 255, 2, 8 | 64, 1 | 16};
  for (int robot = 0; robot < 5; ++robot) {
    const int n_ee = robot == 0 ? 1 : (robot == 1 ? 2 : 4);
    for (int combo = 0; combo < 5; ++combo)
      for (int mi = 0; mi < 7; ++mi) {
        if ((combo + mi + robot) % 3 != 0 && !(combo == 1 && mi < 4)) continue;  // a spread, not the full product
        const int terrain = (robot + combo + mi) % 7;
        (void)n_ee;
        one_case(robot, terrain, combo, 1.4 + 0.2 * combo, masks[mi], 0.1 - 0.01 * mi);
        ++cases;
      }
  }
  std::printf("asan_driver: %d cases, %d failures\n", cases, fails);
  return fails ? 1 : 0;
}
