// Device-side tables shared by the host packer (structure.cc / capi.cc) and the HIP kernels
// (kernels.hip).  Everything here is x-independent: it is what the reference recomputes on every
// call (active polynomial, local time, node->variable maps, CSR positions) hoisted to setup time and
// flattened into one fixed-size record per lane, so that a lane needs exactly two dependent memory
// round trips (its record, then its slice of x) before it can compute.
// One blob per distinct contact schedule; every problem of a batch that shares the schedule shares
// the blob (it stays L2 resident).
#pragma once
#include <stdint.h>

namespace twr {

constexpr int kMaxEE = 4;

// Candidate descriptors of one cubic-Hermite polynomial of an ee-motion / ee-force spline.
// The optimisation variables that touch the polynomial are the contiguous range
// x[xbase, xbase+nslots) ("slots", ascending column order).  Candidate c = j*3+d is node value
// (j,d), j in {p0,v0,p1,v1}, d in {x,y,z}; its 16-bit descriptor holds
//   bits 0-3   slot of the variable holding it, 0xF = not a variable (constant 0)
//   bits 4-7   rank of that slot among slots with dim != (d+1)%3   (dynamic ang row (d+1)%3)
//   bits 8-11  rank of that slot among slots with dim != (d+2)%3   (dynamic ang row (d+2)%3)
//   bits 12-15 rank of that slot among slots with dim == d         (dynamic lin row d)
// meta = nslots | cnt[0]<<4 | cnt[1]<<8 | cnt[2]<<12 | shared<<16 (cnt = slots per dim).
// `shared` marks a stance ee-motion polynomial: p0 and p1 are the same variable, the kernel folds
// w_p1 into w_p0 and the p1 candidates are marked absent.
// (reference: nodes_variables_phase_based.cc:210-298, node_spline.cc:84-112)
struct PolyDesc {
  double iT;  // 1 / duration
  int32_t xbase;
  uint32_t meta;
  uint16_t cand[12];
};

// rangeofmotion-<ee>, fixed timings (rom_kernel).  What a lane (= one time node) needs is split by what it depends on, so
// that a sweep -- every candidate its own tables, read from HBM exactly once per evaluation -- streams 10 KB per candidate
// instead of the 51 KB of one 64-byte record per (time node, ee):
//   RomNode  per time node, shared by the slices of all end-effectors: grid time, the base-spline lookup, and for every
//            end-effector which RomSeg of its slice the node reads                                            (32 B)
//   RomSeg   per (slice, polynomial of the ee-motion spline active inside the slice): everything that is constant while
//            that polynomial stays active -- its start time, 1 / duration, variable range, where the segment's rows start (48 B)
// The local time of the ee spline is t - t0 (t0 = the sum of the durations before the polynomial, accumulated like
// Spline::GetSegmentID does, spline.cc:48-60); WHICH polynomial is active is still decided on the host by the reference's
// rule (eps 1e-10, previous polynomial at a junction).  The reference subtracts the durations from t one by one
// (spline.cc:62-78): t - t0 differs from that by rounding only (a few ulp of t), far inside the 1e-9 parity bar.
constexpr int kRomStage = 4856;    // Jacobian values of one slice (LDS image, doubles); + 2 + 64 + 192 doubles = 40912 B
constexpr int kRomMaxSeg = 8;      // polynomials of the ee spline per slice (a slice is cut earlier if it would span more)
struct RomNode {
  double t;           // global time of the node (TimeDiscretizationConstraint::dts_)
  double tb, iTb;     // base spline: local time in the active polynomial, 1/duration
  int32_t q6;         // 6 * (active base polynomial): offset of its first node in base-lin / base-ang
  uint32_t seg;       // bits 3e .. 3e+2: which segment (RomSeg) of ee e's slice the node belongs to
};
static_assert(sizeof(RomNode) == 32, "RomNode layout");
struct RomSeg {
  double t0, iTm;     // ee-motion polynomial: start time, 1/duration
  uint32_t slots[2];  // 12 x 4 bit: slot of candidate c, 0xF = absent
  int32_t xbase;      // first x index of the polynomial's variables
  uint32_t meta;      // as PolyDesc::meta
  int32_t voff0;      // CSR offset of the segment's first row relative to the slice's first value
  int32_t kfirst;     // node (lane) of the slice at which the segment starts
  int32_t node_vals;  // Jacobian values per time node inside the segment (68 + 3 nslots)
  int32_t pad;
};
static_assert(sizeof(RomSeg) == 48, "RomSeg layout");

// rangeofmotion-<ee> with optimised timings: one record per time node, written by the pre-pass (64 B)
struct RomRec {
  double tb, iTb;     // base spline: local time in the active polynomial, 1/duration
  double tm, iTm;     // ee-motion spline
  int32_t q6;         // 6 * (active base polynomial): offset of its first node in base-lin / base-ang
  int32_t xbase;      // first x index of the active ee-motion polynomial's variables
  int32_t voff;       // CSR offset of row 3k relative to the set's first value
  uint32_t meta;      // as PolyDesc::meta
  uint32_t slots[2];  // 12 x 4 bit: slot of candidate c, 0xF = absent
  uint32_t pad[2];
};
static_assert(sizeof(RomRec) == 64, "RomRec layout");

// dynamic, per time node, shared by the four lanes of the quad (32 B)
struct DynShared {
  double tb, iTb;
  int32_t q6;
  int32_t voff;  // CSR offset of row 6k relative to the set's first value
  uint32_t pad[2];
};
static_assert(sizeof(DynShared) == 32, "DynShared layout");


// ---- dynamic, fixed timings (dyn_kernel): everything index-like is resolved on the host, down to LDS byte offsets.
// A slice (work item) is a run of <= 16 time nodes.  The wave first stages the slice's part of x in LDS ("xs":
// a zero pair followed by <= 254 doubles, filled cooperatively from a per-slice index map), so that a lane
// gets its node values with LDS reads instead of ~30 scattered global loads, and a value that is not an
// optimisation variable simply reads the zero slot (no selects).
#ifndef TWR_DYN_IMAGE
#define TWR_DYN_IMAGE 2238
#endif
#ifndef TWR_DYN_XS
#define TWR_DYN_XS 222
#endif
// Sized for EIGHT single-wave workgroups per CU (20 KB each; the kernel's 253 VGPRs allow no more): measured on one box
// 0.637 ms per 8192 C3 problems against 0.692 ms with 2654-value images at six per CU.
constexpr int kDynImage = TWR_DYN_IMAGE;    // Jacobian values of one slice (LDS image, doubles)
constexpr int kDynNodes = 16;      // time nodes per slice: four lanes each
constexpr int kDynG0 = kDynImage + 2;   // image + parity slack, then the constraint values of the slice (6 per node)
constexpr int kDynXsCap = TWR_DYN_XS;     // staged doubles per slice (8-bit staging indices 2..255; 0/1 = the zero pair)
// Tables of dyn_kernel, split by what they depend on (round 4: a sweep reads every candidate's tables from HBM exactly
// once per evaluation, so their size is traffic -- 33-38 KB per K = 200 quadruped candidate with these records whole,
// 88-112 KB with one 64-byte record per (time node, role) and one 128-byte offset record per (polynomial combination,
// role)) -- and, within a record, by whether the bytes hold TIMES or LAYOUT:
//   DynNodeT per time node (24 B)             grid time, local time and 1 / duration of the active base polynomial
//   DynNodeL per time node (16 B)             where the node's values sit: staging offsets of the base polynomials, row offsets
//   DynSel   per (time node, role) (4 B)      WHICH tile / polynomial records the lane reads
//   DynPolyT per polynomial of an ee spline (16 B)   start time, 1 / duration
//   DynPolyL per polynomial of an ee spline (64 B)   which of its twelve node values are variables and in which slot, and
//                                              the RANK of every tile value inside its tile
//   DynTile  per (polynomial combination of a slice, role) (20 B)   where the role's two tiles start in each row, and
//                                              where the polynomials' variables sit in the slice's staging area
// The layout tables (DynNodeL, DynSel, DynPolyL, DynTile, the staging maps) depend on the contact sequence and on WHICH
// polynomial every time node falls into, not on the durations themselves: candidates of a sweep that differ in the total
// time only (every duration and the time step scaled alike) have byte-identical layout tables, and a batch stores
// byte-identical tables ONCE (twr_batch_create shares them by content; nothing here knows about sweeps).  Such a candidate
// then owns 24 B per time node + 16 B per polynomial (~7 KB for a K = 200 quadruped).
// A tile value goes to  node base + tile start in the row (DynTile) + 8 * rank (DynPolyL);  the local time of an ee
// spline is t - t0 (as in RomSeg: the active polynomial is still chosen on the host by the reference's rule, the
// subtraction differs from the reference's one-by-one subtraction by rounding only).
struct DynNodeT {
  double t;              // global time of the node (TimeDiscretizationConstraint::dts_)
  double tb, iTb;        // base spline: local time in the active polynomial, 1/duration
};
static_assert(sizeof(DynNodeT) == 24, "DynNodeT layout");
struct DynNodeL {
  uint16_t sb_lin;       // byte offset inside xs of the active base-lin polynomial's first node value ([p0 v0 p1 v1] x 3)
  uint16_t sb_ang;
  uint16_t nb;           // byte offset of the node's first Jacobian value inside the slice image
  uint16_t rs1, rs2;     // byte offsets of rows AY, AZ relative to the node's first value (AX = 0)
  uint16_t rl[3];        // rows LX, LY, LZ
};
static_assert(sizeof(DynNodeL) == 16, "DynNodeL layout");
// per time node, shared by the four lanes of the quad: the two records as a lane holds them
struct DynNode {
  double t, tb, iTb;
  uint16_t sb_lin, sb_ang, nb, rs1, rs2, rl[3];
};
static_assert(sizeof(DynNode) == 40, "DynNode layout");
struct DynSel {
  uint16_t tile;         // index of the lane's DynTile record (DynWork::tile)
  uint8_t dm, df;        // ee-motion / ee-force polynomial: index of its DynPolyT / DynPolyL records relative to
                         // DynWork::poly_t / poly_l, 255 = the dummy records (roles >= n_ee; constants of the kernel)
};
static_assert(sizeof(DynSel) == 4, "DynSel layout");
constexpr int kDynPolyDummy = 255;
// One polynomial of an ee-motion / ee-force spline.  Candidate c = j*3+d is node value (j, d), j in {p0,v0,p1,v1}.
//   rel[c]   slot of the variable holding the candidate (its staging index is DynTile::s_m|s_f + rel), 0 when it is not a
//            variable;  pres[c] = 0xFF / 0x00: the staging indices of four candidates are (S4 + rel4) & pres4, so a value that
//            is not a variable reads the zero slot (index 0) without a select
//   code     ee-motion: [c][2] = 8 * rank of the candidate's slot among the slots of rows (d+1)%3 and (d+2)%3 of the tile;
//            ee-force:  [c][3] = the same two angular rows, then linear row d.
//            A candidate that is NOT a variable carries the code of another candidate of the same d that IS one:
//            ee-motion -> p0 (positions are always variables); the kernel stores j = 3, 2, 1 first and p0 last, so p0's
//            value overwrites the garbage.  ee-force -> the node value of the OTHER node (a force node is either all
//            variables or all constant); the kernel then stores that node's VALUE a second time (flags bits 1, 2), so no
//            order is needed.  A polynomial without any variable has codes 0 and its tile starts are pointed at base-lin
//            entries of its node, which the same wave overwrites later in program order (DynTile).
struct DynPolyT {
  double t0, iT;         // start time (sum of the durations before it), 1 / duration
};
static_assert(sizeof(DynPolyT) == 16, "DynPolyT layout");
struct DynPolyL {
  uint8_t rel[12];
  uint8_t pres[12];
  uint32_t flags;        // bit 0: stance ee-motion polynomial (p1 shares p0's variable: w_p1 folds into w_p0);
                         // bit 1: first node constant (p0, v0 store p1's, v1's value); bit 2: second node constant
  uint8_t code[36];
};
static_assert(sizeof(DynPolyL) == 64, "DynPolyL layout");
struct DynTile {
  uint16_t base_m[3];    // byte offset, relative to the node's first value, of the role's ee-motion tile in rows AX, AY, AZ
  uint16_t base_f[6];    // ee-force tile in rows AX, AY, AZ, LX, LY, LZ
  uint8_t s_m, s_f;      // staging index of slot 0 of the active ee-motion / ee-force polynomial
};
static_assert(sizeof(DynTile) == 20, "DynTile layout");

struct ForceNode {   // one non-constant ee-force node (force_constraint.cc:50-60)
  int32_t fidx;      // x index of the node's force px (py = +2, pz = +4)
  int32_t hidx;      // x index of the stance foothold x (y = +1)
};
struct TerrainRow {  // one ee-motion node id >= 1 (terrain_constraint.cc:44-55)
  int32_t idx;       // x index of the node's px
  int32_t stride;    // py = idx+stride, pz = idx+2*stride (1 stance node, 2 swing node)
};

struct AccJunction {  // one base-spline junction j (spline_acc_constraint.cc:49-81)
  double c[6];        // d(acc_end(poly j) - acc_start(poly j+1)) / d{p_j, v_j, p_j+1, v_j+1, p_j+2, v_j+2}
};
struct SwingNode {    // one non-constant ee-motion node (swing_constraint.cc:44-121)
  int32_t cur;        // x index of the node's px (vx +1, py +2, vy +3)
  int32_t prev_x, prev_y, next_x, next_y;  // x indices of the neighbours' px / py
  int32_t pad;
};

// One polynomial of a phase-based spline when the phase durations are optimised (PhaseSpline): which
// phase it belongs to (NodesVariablesPhaseBased::PolyInfo, nodes_variables_phase_based.h:70-82), its
// candidate descriptors as in PolyDesc, and where its variables sit inside rows that hold ALL variables
// of the set (phase_spline.cc:44-51).
struct PhasePoly {
  int32_t phase, n_in_phase, poly_in_phase;
  int32_t xbase;
  uint32_t meta;
  uint16_t cand[12];
  uint16_t base_ne[3];  // set variables before xbase with dim != r   (dynamic ang row r)
  uint16_t base_eq[3];  // set variables before xbase with dim == d   (dynamic lin row d)
  uint16_t base_all;    // set variables before xbase                 (rangeofmotion rows)
  uint16_t pad[3];
};
static_assert(sizeof(PhasePoly) == 64, "PhasePoly layout");

// dynamic with optimised timings: where the values of ONE polynomial's candidates go inside a time node's expanded
// rows.  The rows hold all variables of every ee set, so the row layout is a constant of the structure and the byte
// offset (relative to the node's first value) of a candidate depends on its end-effector and polynomial only: one
// record per (ee, polynomial), read off the CSR pattern like DynTile.  A candidate that is not a variable points at
// entry 8 + ee of row AX (a base-ang value the same wave writes afterwards).
struct PhasePutM {
  uint16_t off[4][8];    // [j][2 D + r]  [f]x J_p of candidate (j, D): rows (D+1)%3 and (D+2)%3 of the angular block
};
static_assert(sizeof(PhasePutM) == 64, "PhasePutM layout");
struct PhasePutF {
  uint16_t off[4][12];   // [j][3 D + r]  {[r]x J_f ; -J_f}: the same two angular rows, then linear row D
};
static_assert(sizeof(PhasePutF) == 96, "PhasePutF layout");
struct PhaseEe {         // per end-effector: its duration columns inside a time node of "dynamic" (byte offsets)
  int32_t ns;            // optimised durations (phases - 1)
  int32_t dur_ang[3], dur_lin[3];
  int32_t pad;
};
static_assert(sizeof(PhaseEe) == 32, "PhaseEe layout");
// dynamic with optimised timings, per (ee, time node): the x-dependent segment lookup, written by the pre-pass
// (phase_locate_kernel) into a scratch buffer -- one contiguous array per (problem, ee), so that a workgroup of the
// pre-pass writes whole lines -- and read by dyn_phase_kernel (64 B)
struct DynLoc {
  double tm, Tm, tf, Tf;         // local time in / duration of the active ee-motion and ee-force polynomials
  int32_t xbase_m, xbase_f;      // first x index of their variables
  uint32_t slots_m[2], slots_f[2];  // 12 x 4 bit: slot of candidate c, 0xF = absent
  uint16_t im, jf;               // the polynomials: index into the structure's PhasePutM / PhasePutF arrays (all ee)
  uint8_t cur, flags;            // current phase; bit 0: it is the last one (not a variable), bit 1: stance ee-motion polynomial
  uint8_t np_m, np_f;            // polynomials in the phase | polynomials before this one in the phase << 4
};
static_assert(sizeof(DynLoc) == 64, "DynLoc layout");

// Tables of the optimised-timings variant (TWR_SET_TOTAL_TIME), appended to DevStruct.
struct PhaseTables {
  int32_t off_sched[kMaxEE];   // x offset of ee-schedule<e>
  int32_t n_phases[kMaxEE];
  int32_t n_mpoly[kMaxEE], n_fpoly[kMaxEE];
  uint32_t o_mpoly[kMaxEE], o_fpoly[kMaxEE];  // PhasePoly arrays
  int32_t mne[kMaxEE][3], fne[kMaxEE][3], feq[kMaxEE][3];  // set-wide variable counts by dim (see PhasePoly)
  int32_t msize[kMaxEE];       // variables of ee-motion_e
  int32_t len_ang[3], len_lin[3], node_vals;  // dynamic: row lengths and values per time node
  int32_t rom_len[kMaxEE][3], rom_node_vals[kMaxEE];
  int32_t row_total, nnz_total;  // totalduration-<e> rows (adjacent)
  uint32_t o_tdyn, o_trom;     // double t_global[K] of the dynamic / rangeofmotion grids
  int32_t k_dyn, k_rom;
  int32_t row_dyn, nnz_dyn, row_rom[kMaxEE], nnz_rom[kMaxEE];
  int32_t off_lin, off_ang;
  uint32_t o_dyn_shared, o_rom_recs[kMaxEE];  // host records (their base-spline part stays x-independent)
  int32_t pad_;
  double t_total[kMaxEE];      // PhaseDurations::t_total_
  uint32_t o_mput, o_fput;     // PhasePutM / PhasePutF records of all ee, ee after ee, then 4 dummy records (ee index
                               // e without an end-effector: every offset = its trash entry)
  int32_t n_mput, n_fput;      // real records
  int32_t mput_base[kMaxEE], fput_base[kMaxEE];  // index of the ee's first record
  PhaseEe ee[kMaxEE];
  uint32_t dyn_row_off[5];     // byte offsets of rows AY, AZ, LX, LY, LZ inside a time node of "dynamic" (AX = 0)
  int32_t pad2_[3];
};
constexpr int kMaxPhasePolys = 64;  // polynomials per ee spline with optimised timings (LDS table size)

struct BaseNode {     // one time node of baseMotion (base_motion_constraint.cc:60-90)
  double t, iT;       // local time in the active base polynomial, 1/duration
  int32_t q6, pad;    // 6 * active polynomial
};

// Trajectory sampling (fpowr GetTrajectory, footstep_plan_extractor.h:19-53): polynomial tables of every
// spline.  With fixed timings the durations are constants of the structure; with optimised timings the
// kernel recomputes them from x (PhaseTables).
struct SampleTables {
  int32_t n_base;                         // polynomials of base-lin / base-ang
  int32_t n_phases[kMaxEE], contact0[kMaxEE];
  int32_t n_mpoly[kMaxEE], n_fpoly[kMaxEE];
  uint32_t o_bdur;                        // double[n_base]
  uint32_t o_phdur[kMaxEE];               // double[n_phases]
  uint32_t o_mdur[kMaxEE], o_fdur[kMaxEE];    // double[n_poly]
  uint32_t o_mdesc[kMaxEE], o_fdesc[kMaxEE];  // PolyDesc[n_poly]
  int32_t off_lin, off_ang;
  double t_total;                         // Spline::GetTotalTime of base-lin
};
struct SampleWork {       // cnt <= 64 samples of one problem from sample s0
  uint64_t blob;
  int64_t x_off, out_off;
  int32_t s0, cnt;
};
static_assert(sizeof(SampleWork) == 32, "SampleWork layout");

// Candidate scoring (twr_batch_score): per row the constraint family and the bounds.  The bounds of a structure take a
// handful of distinct (lower, upper) values (dynamic: 0 / 0; rangeofmotion-e: nominal +- deviation per dimension; force-*:
// five; terrain-*: two; ...), so a row carries 16 bits -- family << 12 | index of its pair among the DISTINCT pairs -- instead
// of two doubles: 2 bytes per row next to the 8 bytes of g the kernel reads it for (round 4: 16 bytes, 62 KB per K = 200
// quadruped candidate).  "family" in a meta word is a SLOT: the families of a structure numbered in the order of their
// first row, so that the slot never falls along the rows (a thread of score_kernel walks ascending rows and keeps ONE
// running pair of accumulators, handing it over when the slot moves on); slot_of_family maps back.  Layout in the blob,
// one record at a FIXED offset (kScoreOff, right behind the node head):
//     [ ScoreTables ][ pairs: (lower, upper) x n_pairs ][ zero padding to kScoreHeadBytes ][ meta: uint16[n_rows] ]
// The kernel knows n_rows from its work list and the addresses from the blob's alone, so it asks for the head, the meta
// words and the rows of g in ONE round trip behind the work item, without a field of the header (DevStruct::o_score ==
// kScoreOff; round 5's first form had the record behind a header field: one dependent round trip more).
constexpr int kMaxConSets = 24;   // 4 terrain + dynamic + 2 splineacc + 4 rangeofmotion + 4 force + 4 swing + baseMotion + 4 totalduration
constexpr int kScoreHeadBytes = 2048;
constexpr int kScoreMaxPairs = 127;     // distinct pairs per structure: what the 2-KB head of the record holds behind the 16-byte
                                        // ScoreTables (score_kernel keeps them all in LDS).  A real structure has one or two
                                        // dozen; twr_structure_create rejects one with more
struct ScoreTables {
  int32_t n_rows, n_pairs;
  int8_t slot_of_family[8];        // -1: the structure has no set of that family
};
static_assert(sizeof(ScoreTables) == 16, "ScoreTables layout");

// Values-only evaluation (TWR_EVAL_VALUES, fixed timings) of "dynamic" and "rangeofmotion-*": ONE LANE PER TIME NODE, all
// end-effectors (kernels.hip flat_dyn_math / flat_rom_math) -- the base splines and the rotation of a time node are evaluated
// once instead of once per end-effector (the Jacobian kernels' cuts: a quad of lanes per time node resp. one slice per
// end-effector).  Nothing a lane needs is gathered from global memory: the workgroup (four waves, the items of one problem)
// copies the problem's x into LDS (coalesced) and every wave, beside it, the WINDOW of polynomial records its time nodes use -- per spline (ee-motion_e: 2 e, ee-force_e: 2 e + 1)
// at most kFlatWindow consecutive polynomials, lane 8 s + j fetching record j of spline s --, all in ONE round trip behind
// the work item.  Tables: per time node of an item a FlatNode (which polynomials are active -- decided on the host by the
// reference's rule, spline.cc:48-78 -- as indices INTO THE ITEM'S WINDOWS), per structure one array of FlatPoly over all
// splines: start time (local time = t - t0, as in the Jacobian kernels), 1 / duration and, for each of the twelve candidates
// (node value j = p0, v0, p1, v1 x dimension), the BYTE offset of its variable in the wave's copy of x (a zero pair, then
// x) -- a candidate that is not a variable points at the zero pair, p1 of a stance ee-motion polynomial at the same variable
// as p0 (nodes_variables_phase_based.cc:210-298): no selects in the evaluation.  An item is cut where it would exceed 64 time
// nodes or kFlatWindow polynomials of one spline (structure.cc).
constexpr int kFlatXCap = 2046;    // doubles of x a problem may have for this path (16 KB of LDS per wave; C3: 640)
constexpr int kFlatWindow = 8;     // polynomials per spline and item
struct FlatPoly {
  double t0, iT;
  uint16_t off[12];        // 8 * (2 + x index), or 0 (the zero pair)
  int32_t pad[2];
};
static_assert(sizeof(FlatPoly) == 48, "FlatPoly layout");
constexpr int kFlatPolyLds = 2 * kMaxEE * kFlatWindow * (int)sizeof(FlatPoly);   // bytes: the windows of one item (3072)
struct FlatNode {
  double t;                // grid time (TimeDiscretizationConstraint::dts_)
  double tb, iTb;          // base spline: local time in the active polynomial, 1 / duration
  int32_t q6;              // 6 * (active base polynomial): offset of its first node in base-lin / base-ang
  uint8_t qm[4];           // active polynomial of ee-motion_e, relative to the item's window
  uint8_t qf[4];           // active polynomial of ee-force_e ("dynamic" items only)
  int32_t pad;
};
static_assert(sizeof(FlatNode) == 40, "FlatNode layout");
// Everything an item needs that is the same for all its lanes is in its work record -- addresses, window starts, the
// structure's offsets and rows and (for "dynamic") the model constants: one scalar round trip, nothing behind a pointer of it.
struct FlatWork {          // cnt <= 64 consecutive time nodes of the dynamic (or range-of-motion) grid of one problem; cnt = 0: an
                           // empty item (pads a problem's list to whole groups of four; x_off and n_x are the problem's all the same)
  uint64_t nodes;          // FlatNode[cnt]
  uint64_t polys;          // FlatPoly[] of the structure
  int64_t x_off, g_off;    // the problem's x / g
  int32_t k0, cnt;
  uint64_t start[2];       // 16 bits per spline: first polynomial of its window in the FlatPoly array (splines 0-3 | 4-7)
  uint64_t count;          // 8 bits per spline: polynomials in its window (0: spline not used by the item)
  int32_t n_x, n_ee;       // variables of the problem (<= kFlatXCap), end-effectors
  int32_t off_lin, off_ang;   // x offsets of base-lin / base-ang
  int32_t row_rom[kMaxEE]; // first row of "rangeofmotion-e"
  int32_t row_dyn;         // first row of "dynamic"
  int32_t with_rom;        // "dynamic" items: the two grids coincide and the lane evaluates "rangeofmotion-*" of its time node as well
  double mass, gravity;    // ("dynamic" items)
  double Ib[6];
  int32_t dynamic;         // 1: item of the "dynamic" grid, 0: of the range-of-motion grid
  int32_t gather;          // 1: a coarse grid -- no windows, every lane fetches the records of its own active polynomials
};
static_assert(sizeof(FlatWork) == 176, "FlatWork layout");
// (the kernel addresses the fields by dword)
enum FlatWorkDword {
  kFwNodes = 0, kFwPolys = 2, kFwX = 4, kFwG = 6, kFwK0 = 8, kFwCnt = 9, kFwStart = 10, kFwCount = 14, kFwNx = 16, kFwNee = 17, kFwOffLin = 18,
  kFwOffAng = 19, kFwRowRom = 20, kFwRowDyn = 24, kFwWithRom = 25, kFwMass = 26, kFwGravity = 28, kFwIb = 30, kFwDynamic = 42, kFwGather = 43
};
static_assert(offsetof(FlatWork, nodes) == 4 * kFwNodes && offsetof(FlatWork, polys) == 4 * kFwPolys && offsetof(FlatWork, x_off) == 4 * kFwX &&
                  offsetof(FlatWork, g_off) == 4 * kFwG && offsetof(FlatWork, k0) == 4 * kFwK0 && offsetof(FlatWork, cnt) == 4 * kFwCnt &&
                  offsetof(FlatWork, start) == 4 * kFwStart && offsetof(FlatWork, count) == 4 * kFwCount && offsetof(FlatWork, n_x) == 4 * kFwNx &&
                  offsetof(FlatWork, n_ee) == 4 * kFwNee && offsetof(FlatWork, off_lin) == 4 * kFwOffLin && offsetof(FlatWork, off_ang) == 4 * kFwOffAng &&
                  offsetof(FlatWork, row_rom) == 4 * kFwRowRom && offsetof(FlatWork, row_dyn) == 4 * kFwRowDyn &&
                  offsetof(FlatWork, with_rom) == 4 * kFwWithRom && offsetof(FlatWork, mass) == 4 * kFwMass &&
                  offsetof(FlatWork, gravity) == 4 * kFwGravity && offsetof(FlatWork, Ib) == 4 * kFwIb && offsetof(FlatWork, dynamic) == 4 * kFwDynamic &&
                  offsetof(FlatWork, gather) == 4 * kFwGather,
              "FlatWork dwords");

// Blob header: model constants + what the node kernel needs.  The terrain-ee-motion_e sets are
// adjacent in g / jac, and so are the force-ee-force_e sets, so each family is one flat node list.
struct DevStruct {
  int32_t n_ee, terrain_id;
  int32_t row_terrain, nnz_terrain, n_terrain_rows;  // first row / first value / rows over all ee
  int32_t row_force, nnz_force, n_force_nodes;
  uint32_t o_force_nodes;   // byte offsets inside the blob: ForceNode[n_force_nodes]
  uint32_t o_terrain_rows;  // TerrainRow[n_terrain_rows]
  // splineacc-base-lin | splineacc-base-ang (adjacent, 3 rows x 6 values per junction and set) and
  // swing-ee-motion_e (adjacent, 4 rows x 3 values per swing node); counts are 0 when a family is off
  int32_t row_acc, nnz_acc, n_junctions;
  int32_t row_swing, nnz_swing, n_swing_nodes;
  int32_t off_base_ang;     // x offset of base-ang (base-lin starts at 0)
  uint32_t o_acc;           // AccJunction[n_junctions]
  uint32_t o_swing_nodes;   // SwingNode[n_swing_nodes]
  uint32_t pad_[3];
  double inv_t_swing;       // 1 / t_swing_avg_ (swing_constraint.h:68)
  double mass, gravity, mu, flat_height;
  double Ib[6];  // body inertia tensor entries (0,0),(0,1),(0,2),(1,1),(1,2),(2,2) incl. the sign of
                 // single_rigid_body_dynamics.cc:40-42
  int32_t timings;      // 1: optimised phase durations, PhaseTables at o_phase
  uint32_t o_phase;
  int32_t row_bm, nnz_bm, n_bm_nodes;  // baseMotion: 6 rows x 4 values per time node
  uint32_t o_bm;        // BaseNode[n_bm_nodes]
  // gridded terrain: device address of the cell data (patched in by the batch) -- HeightMapFromCSV: double
  // heights[rows][cols]; Grid (grid_map): float elevation[i + j * rows], rows = size_x, cols = size_y
  uint64_t grid_ptr;
  int32_t grid_rows, grid_cols;
  double grid_res, grid_eps;
  uint32_t o_sample;    // SampleTables
  uint32_t pad2_;
  double grid_px, grid_py;  // Grid: map centre
  uint32_t o_score;         // ScoreTables
  uint32_t o_flat;          // FlatPoly[] (values-only evaluation, fixed timings); 0: none
};

// Right behind the header, at FIXED offsets: the first 64 TerrainRow and the first 64 ForceNode records (zero padded) -- what
// the first (for most structures: the only) pass of node_kernel's terrain and force waves reads.  Their address does not
// depend on any header field, so those loads go out together with the header's instead of behind them (one dependent
// round trip less in a kernel that is nothing but a chain of them).  The full tables follow as before.
constexpr uint32_t kNodeHeadTerrainOff = (uint32_t)((sizeof(DevStruct) + 15) / 16 * 16);
constexpr uint32_t kNodeHeadForceOff = kNodeHeadTerrainOff + 64 * (uint32_t)sizeof(TerrainRow);
constexpr uint32_t kNodeHeadBytes = 64 * (uint32_t)(sizeof(TerrainRow) + sizeof(ForceNode));
constexpr uint32_t kScoreOff = kNodeHeadTerrainOff + kNodeHeadBytes;   // the score record (ScoreTables), see above
static_assert(kScoreOff % 16 == 0, "score record alignment");

// Work items: one contiguous run of time nodes of one constraint set of one problem.  All
// pointers / offsets are absolute so that a workgroup needs no header lookup.
struct DynWork {          // cnt <= 16 time nodes of "dynamic"
  uint64_t nodes_t;       // DynNodeT[k0..]
  uint64_t nodes_l;       // DynNodeL[k0..]                  (layout tables: possibly another structure's identical copy)
  uint64_t sel;           // DynSel[k0 * 4 ..]
  uint64_t tile;          // DynTile records of the structure (indexed by DynSel::tile)
  uint64_t poly_t;        // the slice's first DynPolyT record (DynSel::dm / df count from it)
  uint64_t poly_l;        // the slice's first DynPolyL record
  uint64_t map;           // uint16_t[64][4]: lane l stages x[map[l][c]] at xs[2 + 64 c + l]
  uint64_t hdr;           // DevStruct (mass, gravity, inertia): of the first structure of the batch with these constants
  int64_t x_off;          // problem's x
  int64_t g_off;          // first constraint value of the run (row 6*k0 of the set)
  int64_t j_off;          // first Jacobian value of the run
  int32_t cnt, nvals;     // time nodes, Jacobian values of the run
};
static_assert(sizeof(DynWork) == 96, "DynWork layout");

struct RomWork {          // cnt <= 64 time nodes of "rangeofmotion-<ee>"
  uint64_t nodes;         // RomNode[k0..]
  uint64_t segs;          // RomSeg[] of this slice
  int64_t x_off, g_off, j_off;
  int32_t off_lin, off_ang;
  int32_t cnt, nvals;
  int32_t ee;             // the end-effector (which three bits of RomNode::seg are this slice's)
  int32_t pad;
};
static_assert(sizeof(RomWork) == 64, "RomWork layout");

struct NodeWork {         // all terrain-* and force-* sets of one problem
  uint64_t blob;
  int64_t x_off, g_off, j_off;
};
static_assert(sizeof(NodeWork) == 32, "NodeWork layout");

// Node-based sets of LARGE batches (node_chunk_kernel): one work item = up to 64 items (spline nodes / rows) of ONE family of one
// problem, everything the wave needs as absolute addresses / offsets, so that a persistent wave can prefetch the records of
// the chunk two items ahead and its x values one item ahead.  Families: 0 terrain-* rows, 1 force-* nodes, 2 splineacc-base-*
// rows, 3 swing-* nodes.  (baseMotion / totalduration rows and small batches stay with node_kernel / the fused kernel.)
struct FamWork {
  uint64_t blob;          // DevStruct (terrain constants / grid address: families 0, 1)
  uint64_t table;         // first record of the chunk: TerrainRow / ForceNode / SwingNode; family 2: AccJunction[0]
  int64_t x_off;          // problem's x
  int64_t g_off;          // first constraint value of the chunk
  int64_t j_off;          // first Jacobian value of the chunk
  int32_t cnt;            // items of the chunk (<= 64, one per lane; force: <= 32)
  int32_t i0;             // family 2: first row of the chunk inside splineacc-base-lin|ang; else 0
  int32_t aux0, aux1;     // family 2: rows per set (3 x junctions), x offset of base-ang
  double inv_t_swing;     // family 3: 1 / t_swing_avg_
};
static_assert(sizeof(FamWork) == 64, "FamWork layout");

struct PDynWork {         // optimised timings: one pass = cnt <= 4 time nodes of "dynamic".  Everything the kernel
                          // needs is an absolute address or a value here: no dependent table lookups per pass.
  uint64_t hdr;           // DevStruct (mass, gravity, inertia)
  uint64_t loc;           // DynLoc[k0 ..] of ee 0 of this problem (scratch, written by the pre-pass; ee e at + e * loc_stride)
  uint64_t shared;        // DynShared[k0 ..]
  uint64_t mput, fput;    // PhasePutM / PhasePutF arrays of the structure
  uint64_t ee;            // PhaseEe[4]
  int64_t x_off;          // problem's x
  int64_t g_off, j_off;   // first constraint value (row 6 k0 of the set) / first Jacobian value of the pass
  int32_t cnt, node_vals; // time nodes; expanded values per time node
  int32_t off_lin, off_ang, n_ee;
  uint32_t row_off[5];    // PhaseTables::dyn_row_off
  int32_t n_mput, n_fput; // records of real polynomials; four dummy records (one per ee index) follow them
  int32_t loc_stride;     // bytes between the DynLoc arrays of consecutive end-effectors (= 64 k_dyn)
  int32_t pad;
};
static_assert(sizeof(PDynWork) == 128, "PDynWork layout");

// Optimised timings, rangeofmotion-<ee>: the pre-pass (phase_locate_kernel) turns the x-dependent segment
// lookup into RomRec records in a scratch buffer, so that the persistent kernel sees the same two-step
// record -> x dependency as the fixed-timing kernel.  The records carry the phase data in RomRec::pad:
//   pad[0] = base_all | current phase << 16 | in_last_phase << 24,  pad[1] = n_in_phase | poly_in_phase << 8
struct LocWork {          // one (problem, ee)
  uint64_t blob;
  uint64_t recs;          // RomRec[k_rom] (output), 0: no rangeofmotion sets
  uint64_t dyn_loc;       // DynLoc[k_dyn] of this (problem, ee) (output; one contiguous array per ee), 0: no dynamic set
  int64_t x_off;
  int32_t ee, pad;
  int64_t pad2;
};
static_assert(sizeof(LocWork) == 48, "LocWork layout");
struct RomPhaseWork {     // one pass: cnt <= 16 time nodes of one (problem, ee), four lanes each
  uint64_t recs;          // RomRec[k0..] written by the pre-pass
  int64_t x_off, g_off, j_off;  // problem's x; first constraint value / first Jacobian value of the run
  int32_t off_lin, off_ang;
  int32_t cnt, msize;     // time nodes; variables of ee-motion_e
  int32_t ns, node_vals;  // duration variables of the ee; expanded values per time node
};
static_assert(sizeof(RomPhaseWork) == 56, "RomPhaseWork layout");

}  // namespace twr
