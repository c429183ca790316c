// Host-side problem structure: everything about one candidate contact schedule that does
// not depend on x.  Built once (twr_structure_create); the reference spreads the same
// information over NlpFormulation::GetVariableSets/GetConstraints, NodesVariables*,
// SplineHolder and the constraint constructors (see citations in structure.cc).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/towr_amd.h"
#include "device_tables.h"

namespace twr {

struct SplineLayout {
  std::vector<double> durations;                 // per polynomial
  // x index (global, stacked) of node value (node, deriv, dim); -1 = not optimised (constant 0)
  std::vector<int> idx;                          // [node][deriv(2)][dim(3)]
  std::vector<char> node_constant;               // phase based sets only
  std::vector<int> poly_phase;                   // phase of each polynomial (phase based)
  int n_nodes = 0;
  int var_offset = 0, var_size = 0;
  int at(int node, int deriv, int dim) const { return idx[(node * 2 + deriv) * 3 + dim]; }
};

struct SetInfo {
  std::string name;
  int offset = 0, size = 0, nnz_offset = 0, nnz = 0;
};

struct TimeNode {  // result of Spline::GetLocalTime for one spline at one grid time
  int poly;
  double t_local;
};

struct TerrainGrid {   // HeightMapFromCSV (include/towr/terrain/height_map_from_csv.h) or Grid (grid_height_map.h)
  std::vector<double> heights;  // CSV: [y_cell * cols + x_cell]
  int rows = 0, cols = 0;       // CSV: rows x cols; grid_map: size_x x size_y cells
  double res = 0.17, eps = 0.17 / 50;  // CSV :112-115; grid_map: resolution, resolution / 6 (grid_height_map.h:25)
  bool grid_map = false;        // Grid: float elevation layer, column-major [i + j * size_x]
  std::vector<float> elevation;
  double pos_x = 0, pos_y = 0;  // GridMap::getPosition()
  double Height(double x, double y) const;
};

struct Structure {
  twr_model model;
  std::shared_ptr<const TerrainGrid> grid;  // TWR_TERRAIN_CSV_GRID only
  twr_schedule schedule;
  twr_params params;
  int n_ee = 0;
  double T = 0;

  SplineLayout base;  // durations shared by base-lin / base-ang; idx refers to base-lin
  int off_base_lin = 0, off_base_ang = 0;
  std::vector<SplineLayout> motion, force;
  std::vector<SetInfo> var_sets, con_sets;
  int n_vars = 0, n_rows = 0, nnz = 0;
  bool timings = false;             // TWR_SET_TOTAL_TIME: phase durations are variables
  int off_schedule[kMaxEE] = {0, 0, 0, 0};
  PhaseTables phase_tables;         // filled by BuildPattern / PackBlob when timings

  std::vector<double> grid_dyn, grid_rom, grid_bm;
  std::vector<TimeNode> dyn_base, rom_base, bm_base;
  std::vector<std::vector<TimeNode>> dyn_motion, dyn_force, rom_motion;  // [ee][k]
  std::vector<std::vector<PolyDesc>> mpoly, fpoly;                       // [ee][poly]
  std::vector<std::vector<ForceNode>> force_nodes;                       // [ee]
  std::vector<std::vector<TerrainRow>> terrain_rows;                     // [ee]
  std::vector<AccJunction> acc_junctions;                                // base spline junctions
  std::vector<std::vector<SwingNode>> swing_nodes;                       // [ee]

  std::vector<int32_t> row_ptr, col_idx;
  std::vector<double> lower, upper;

  std::vector<char> blob;  // packed DevStruct + tables (device_tables.h)
  // byte offsets of the per-lane record arrays inside the blob
  uint32_t off_dyn_shared = 0, off_dyn_lanes = 0;   // optimised timings: DynShared[k] (base-spline part)
  // fixed timings: slices of the dynamic set and their tables (device_tables.h DynNodeT / DynNodeL / DynSel / DynPolyT / DynPolyL / DynTile)
  struct DynSlice {
    int k0, cnt, nvals;
    uint32_t map;        // byte offset of the slice's staging map inside the blob: uint16[64][4], lane-transposed
    uint32_t map2;       // the two-chunk form uint16[64][2] of the same map (slices that stage <= 128 doubles; else = map)
    int poly0;           // index of the first DynPolyT / DynPolyL record the slice reads (DynSel::dm / df count from it)
  };
  std::vector<DynSlice> dyn_slices;
  int dyn_staged_max = 0;   // most doubles of x one slice stages (<= 128: the batch may use the two-chunk maps)
  uint32_t off_dyn_nodes_t = 0, off_dyn_nodes_l = 0, off_dyn_sel = 0, off_dyn_tile = 0, off_dyn_poly_t = 0, off_dyn_poly_l = 0;
  struct TableRef {
    uint32_t off, bytes;
  };
  std::vector<TableRef> dyn_layout_tables;   // the layout tables of dyn_kernel inside the blob (times excluded): what a batch may
                                             // share between structures when the bytes are identical
  uint32_t off_rom_recs[kMaxEE] = {0, 0, 0, 0};   // optimised timings: RomRec[k] templates (base-spline part)
  // fixed timings: slices of rangeofmotion-<ee> (device_tables.h RomNode / RomSeg)
  struct RomSlice {
    int k0, cnt, nvals;
    uint32_t segs;               // byte offset of the slice's RomSeg records inside the blob
    uint8_t first[kRomMaxSeg];   // first lane of every segment (255: none)
  };
  std::vector<std::vector<RomSlice>> rom_slices;   // [ee]
  uint32_t off_rom_nodes = 0;
  // values-only evaluation of dynamic / rangeofmotion-*, one lane per time node (device_tables.h FlatNode / FlatPoly /
  // FlatWork): blob offsets (0 = none) and the items of a problem of this structure (twr_batch_create adds the problem's
  // addresses)
  struct FlatItem {
    int k0 = 0, cnt = 0;                 // time nodes [k0, k0 + cnt) of the grid
    uint64_t start[2] = {0, 0}, count = 0;   // FlatWork::start / count
    bool gather = false;                 // FlatWork::gather
  };
  uint32_t off_flat_polys = 0, off_flat_rom = 0, off_flat_dyn = 0;   // FlatPoly[] | FlatNode[] of the two grids
  int flat_row_dyn = 0, flat_row_rom[kMaxEE] = {0, 0, 0, 0};
  bool flat_with_rom = false;   // the two grids coincide: the "dynamic" items take the range-of-motion rows along
  std::vector<FlatItem> flat_items_rom, flat_items_dyn;

  const SetInfo* FindSet(const std::string& name) const;  // nullptr if the family is switched off
  void Build();            // throws std::runtime_error
  void BuildSizes();       // variables, time tables and CSR pattern only (n_vars / n_rows / nnz): no device tables
  void InitialGuess(const double* lin0, const double* ang0, const double* lin1, const double* ang1,
                    const double* ee0, double* x) const;
  int SampleCount(double dt) const;  // fpowr GetTrajectory: samples while t <= T + 1e-5, t accumulated
  void VariableBounds(const double* init_base, const double* final_base, const double* ee0, double* lower,
                      double* upper) const;

 private:
  void BuildVariables();
  void BuildTimeTables();
  void BuildPattern();
  void PackBlob();
};

// gait tables
// Layout tables of dyn_kernel stored once per distinct CONTENT (device_tables.h): for every structure and every entry of its
// dyn_layout_tables, which structure's copy a batch reads -- the first one of the list with the same bytes (itself if none).
// Host logic only (no device): twr_batch_create turns {owner, offset} into device addresses.
struct LayoutShare {
  struct Ref {
    int owner;          // index into the structure list
    uint32_t off;       // byte offset of the table inside the OWNER's blob
  };
  std::vector<std::vector<Ref>> of;     // [structure][i] for dyn_layout_tables[i]
  int64_t bytes_built = 0, bytes_distinct = 0;
};
LayoutShare ShareLayoutTables(const std::vector<const Structure*>& structs);
// Store policy of a batch (kernels.hip copy_out_fixed): non-temporal stores for SWEEP-LIKE batches -- fewer than
// four problems per structure on average, so that every evaluation re-reads tables and x that only one problem uses -- whose
// evaluation writes more than the Infinity Cache holds, so that plain stores would flush those tables out of it.
// The memory-side cache is a property of the DEVICE the batch lives on (HIP reports no field for it -- hipDeviceProp_t has
// l2CacheSize, the per-XCD L2 --, so it is a table by architecture name; a partitioned device, CPX / NPS4, sees its share):
//   gfx950 (MI350X / MI355X), gfx942 (MI300X / MI300A / MI325X): 256 MB;  anything else: eight times the L2 it reports
//   (a conservative stand-in: plain stores are the safe policy, they are never more than a few percent behind on a sweep,
//   while non-temporal stores cost rom_kernel 10-15 % where nothing needs protecting).
inline int64_t MemorySideCacheBytes(const char* gcn_arch_name, int64_t l2_bytes, int compute_partitions = 1) {
  const std::string arch = gcn_arch_name ? gcn_arch_name : "";
  int64_t bytes = 8 * l2_bytes;
  if (arch.rfind("gfx950", 0) == 0 || arch.rfind("gfx942", 0) == 0) bytes = (int64_t)256 << 20;
  return bytes / (compute_partitions > 0 ? compute_partitions : 1);
}
constexpr int64_t kInfinityCacheBytes = (int64_t)256 << 20;   // MI355X (what the measurements of DESIGN 6.R4 were taken on)
inline bool StreamNonTemporal(int structures_used, int n_problems, int64_t output_bytes_per_evaluation,
                              int64_t memory_side_cache_bytes = kInfinityCacheBytes) {
  return (int64_t)structures_used * 4 > n_problems && output_bytes_per_evaluation > memory_side_cache_bytes;
}
void GaitCombo(int n_ee, int combo, double t_total, double swing_scale, twr_schedule* out);
void ModelPreset(int robot, int terrain, twr_model* out);
double TerrainHeightHost(const twr_model& m, const TerrainGrid* grid, double x, double y);

}  // namespace twr
