// Layout of the per-structure device blob shared by the host packer (structure.cc)
// and the HIP kernels (kernels.hip).  All tables are x-independent: they are what the
// reference recomputes on every call (active polynomial, local time, node->variable
// maps) hoisted to setup time.  One blob per distinct contact schedule; every problem
// of a batch that shares the schedule shares the blob (it stays L2 resident).
#pragma once
#include <stdint.h>

namespace twr {

constexpr int kMaxEE = 4;

// One cubic-Hermite polynomial of an ee-motion / ee-force spline.
// The optimisation variables that touch it are the contiguous range x[xbase, xbase+nslots)
// ("slots", ascending column order).  cand[j*3+d] describes node value (j,d), j in
// {p0,v0,p1,v1}, d in {x,y,z}:
//   bits 0-3   slot of the variable holding it, 0xF = not a variable (constant 0)
//   bits 4-7   rank of that slot among slots with dim != (d+1)%3   (dynamic ang row (d+1)%3)
//   bits 8-11  rank of that slot among slots with dim != (d+2)%3   (dynamic ang row (d+2)%3)
//   bits 12-15 rank of that slot among slots with dim == d         (dynamic lin row d)
// `shared` = 1 for a stance ee-motion polynomial: p0 and p1 are the same variable, the
// kernel folds w_p1 into w_p0 and the p1 candidates are marked absent.
// (reference: nodes_variables_phase_based.cc:210-298, node_spline.cc:84-112)
struct EePoly {
  double iT;       // 1 / duration
  int32_t xbase;
  uint8_t nslots;
  uint8_t cnt[3];  // slots per dim
  uint16_t cand[12];
  uint8_t shared;
  uint8_t pad[7];
};
static_assert(sizeof(EePoly) == 48, "EePoly layout");

struct ForceNode {   // one non-constant ee-force node (force_constraint.cc:50-60)
  int32_t fidx;      // x index of the node's force px (py = +2, pz = +4)
  int32_t hidx;      // x index of the stance foothold x (y = +1)
};
struct TerrainRow {  // one ee-motion node id >= 1 (terrain_constraint.cc:44-55)
  int32_t idx;       // x index of the node's px
  int32_t stride;    // py = idx+stride, pz = idx+2*stride (1 stance node, 2 swing node)
};

struct DevStruct {
  int32_t n_ee, n_vars, n_rows, nnz;
  int32_t k_dyn, k_rom;
  int32_t off_base_lin, off_base_ang;  // x offsets of the two base variable sets
  int32_t n_base_polys;
  int32_t terrain_id;
  // constraint-set offsets inside one problem's g / jac arrays
  int32_t row_terrain[kMaxEE], nnz_terrain[kMaxEE], n_terrain_rows[kMaxEE];
  int32_t row_dyn, nnz_dyn;
  int32_t row_rom[kMaxEE], nnz_rom[kMaxEE];
  int32_t row_force[kMaxEE], nnz_force[kMaxEE], n_force_nodes[kMaxEE];
  // byte offsets of the tables inside the blob (from the blob start)
  uint32_t o_base_iT;                                  // double[n_base_polys], 1/duration
  uint32_t o_mpoly[kMaxEE], o_fpoly[kMaxEE];           // EePoly[]
  uint32_t o_dyn_tl_base, o_dyn_base_poly, o_dyn_val_off;  // double[K], int32[K], int32[K+1]
  uint32_t o_dyn_mpoly[kMaxEE], o_dyn_tl_m[kMaxEE];    // int32[K], double[K]
  uint32_t o_dyn_fpoly[kMaxEE], o_dyn_tl_f[kMaxEE];
  uint32_t o_rom_tl_base, o_rom_base_poly;             // double[K], int32[K]
  uint32_t o_rom_mpoly[kMaxEE], o_rom_tl_m[kMaxEE], o_rom_val_off[kMaxEE];  // int32[K], double[K], int32[K+1]
  uint32_t o_force_nodes[kMaxEE];                      // ForceNode[]
  uint32_t o_terrain_rows[kMaxEE];                     // TerrainRow[]
  uint32_t pad_;
  // model constants
  double mass, gravity, mu, flat_height;
  double Ib[6];  // body inertia tensor entries (0,0),(0,1),(0,2),(1,1),(1,2),(2,2) incl. the sign of
                 // single_rigid_body_dynamics.cc:40-42
};

// One workgroup's job: a contiguous run of time nodes of one constraint set of one problem
// (dynamic: cnt <= 16, rangeofmotion-<ee>: cnt <= 64) or the node block of a problem.
struct Work {
  uint64_t blob;    // device address of the problem's DevStruct blob
  int64_t x_off, g_off, j_off;  // offsets of the problem inside the batch arrays (doubles)
  int32_t k0, cnt;
  int32_t ee;
  int32_t pad;
};
static_assert(sizeof(Work) == 48, "Work layout");

}  // namespace twr
