// fused.hpp -- the register-resident expression program shared by the host-side
// expression compiler (engine) and the fused scan kernels.
//
// The reference evaluates `filter -> select/agg` as a tree of PhysicalExpr nodes that
// each materialise a full column (polars-expr/src/expressions/{binary,aggregation,
// column,literal,cast}.rs; SURVEY.md 3.2: >= 3 full passes over memory).  Here an AExpr
// subtree is lowered to a short straight-line program over 16 virtual 64-bit slots
// that live in VGPRs; a fused kernel streams the input columns once (16-B loads per
// lane), runs the program per row in registers and hands the row to a sink
// (register aggregation, hash aggregation, join build, join probe).
//
// Control flow of the interpreter is wave-uniform: opcodes and slot numbers come from
// kernel arguments (SGPRs), so decode runs on the scalar unit and slot selection uses
// the VGPR index mode (s_set_gpr_idx_on) rather than branches.  Hot shapes (the
// BASELINE configs, TPC-H Q1/Q3) are additionally pre-instantiated with the program as
// a compile-time constant so the whole interpreter folds away.
#pragma once
#include <stdint.h>

#ifndef PLX_HD
#if defined(__HIPCC__) || defined(__HIP__)
#define PLX_HD __host__ __device__
#else
#define PLX_HD
#endif
#endif

namespace plx {
namespace fused {

constexpr int kMaxInputs = 10;
constexpr int kMaxOps = 32;
constexpr int kSlots = 16;
constexpr int kMaxAggs = 16;
constexpr int kMaxKeys = 4;          // columns of a wide (unpackable) group key
constexpr int kRows = 2;            // rows per lane per tile (one 16-B load of an 8-B column)
constexpr int kTileRows = 64 * kRows;
constexpr uint8_t kNone = 255;

enum OpCode : uint8_t {
  OP_NOP = 0,
  OP_LOAD,     // dst <- input[a] widened to 64 bit (sign/zero extension by dtype; f64 bits)
  OP_CONST,    // dst <- imm[pc], always valid
  OP_ADD_F, OP_SUB_F, OP_MUL_F, OP_DIV_F,   // f64
  OP_ADD_I, OP_SUB_I, OP_MUL_I,             // wrapping 64-bit
  OP_I2F, OP_U2F,                           // int -> f64 (AExpr::Cast inserted by type coercion)
  OP_CMP_I, OP_CMP_U, OP_CMP_F,             // c = plx_cmp_op; result 0/1
  OP_AND, OP_OR, OP_XOR, OP_NOT,            // boolean slots
  OP_IFNULL,   // dst <- valid(a) ? a : imm[pc]; result always valid (null group code)
  OP_MOV,
  OP_CANON_F,  // float key canonicalisation: -0 -> +0, any NaN -> 0x7ff8000000000000 (total_ord.rs:40-48)
  // 64-bit integer floor division / modulo, Python sign rules; divisor 0 -> null (signed.rs:35-70)
  OP_FDIV_I, OP_MOD_I, OP_FDIV_U, OP_MOD_U,
  // dst <- bit (a - imm[pc]) of lookup bitmap args.lut[c] (0 outside [0, range)); validity of a.  The membership test of a
  // semi-join whose filter side was reduced to a bitmap over its (dense) key range.
  OP_BITLOOKUP,
  // dst <- a, valid only in rows where b is valid and true.  In front of a bitmap lookup it takes the rows an earlier, cheaper conjunct of the predicate already
  // rejected out of the lookup: an invalid lane issues no load (a random lookup moves a whole line from the L2 to the CU: TPC-H Q3's orders scan with the customer
  // filter was bound by those lines, not by HBM)
  OP_MASKV
};

struct Op {
  uint8_t code, dst, a, b, c, _pad[3];
};

enum AggKind : uint8_t {
  AGG_NONE = 0,
  AGG_SUM_F,      // f64 sum of valid selected rows
  AGG_SUM_I,      // wrapping 64-bit sum
  AGG_COUNT,      // valid selected rows of src
  AGG_COUNT_ORD,  // valid, selected and not NaN (f64 src)
  AGG_LEN,        // selected rows
  AGG_MIN_F, AGG_MAX_F, AGG_MIN_I, AGG_MAX_I, AGG_MIN_U, AGG_MAX_U,
  AGG_FIRST_ROW   // smallest selected row index (group order / first())
};

struct Agg {
  uint8_t kind, src;
};

// Static part of a program: what AOT specialisations are keyed on (immediates,
// pointers and sizes are runtime).
struct Shape {
  uint8_t n_inputs, n_ops, n_aggs, pred, key;  // pred/key: slot or kNone
  // wide group key: n_keys >= 2 raw key slots (value + validity), compared word by word by the
  // multi-word hash sink; 0 when the key is a single slot (`key`).
  uint8_t n_keys, keys[kMaxKeys];
  uint8_t in_dtype[kMaxInputs];                // plx_dtype
  uint8_t in_nullable[kMaxInputs];
  Op ops[kMaxOps];
  Agg aggs[kMaxAggs];
};

// late materialisation split of a program (fused_device.hpp run_split): bit pc of `early` = op pc feeds the predicate / key
struct ProgramSplit { uint32_t early; bool any_late; };
PLX_HD constexpr uint32_t op_src_slots(const Op& op) {
  if (op.code == OP_LOAD || op.code == OP_CONST || op.code == OP_NOP) return 0;
  return (1u << op.a) | (1u << op.b);
}
PLX_HD constexpr ProgramSplit split_program(const Shape& s) {
  const uint32_t all = s.n_ops >= 32 ? ~0u : ((1u << s.n_ops) - 1u);
  uint32_t live_out = 0;
  if (s.pred != kNone) live_out |= 1u << s.pred;
  if (s.key != kNone) live_out |= 1u << s.key;
  for (int i = 0; i < s.n_keys; i++) live_out |= 1u << s.keys[i];
  uint32_t need = live_out, early = 0;
  for (int pc = s.n_ops - 1; pc >= 0; pc--) {
    const Op op = s.ops[pc];
    if (op.code == OP_NOP || !(need & (1u << op.dst))) continue;
    early |= 1u << pc;
    need &= ~(1u << op.dst);
    need |= op_src_slots(op);
  }
  for (int k = 0; k < s.n_aggs; k++) live_out |= 1u << s.aggs[k].src;
  bool ok = true;
  // a late op must still read the value its (early) producer wrote: no early op further down may reuse that slot
  for (int pl = 0; pl < s.n_ops; pl++) {
    if ((early >> pl) & 1u) continue;
    const uint32_t srcs = op_src_slots(s.ops[pl]);
    for (int sl = 0; sl < kSlots; sl++) {
      if (!((srcs >> sl) & 1u)) continue;
      int def = -1;
      for (int pc = 0; pc < pl; pc++) if (s.ops[pc].code != OP_NOP && s.ops[pc].dst == sl) def = pc;
      if (def < 0 || !((early >> def) & 1u)) continue;
      for (int pc = def + 1; pc < s.n_ops; pc++) if (((early >> pc) & 1u) && s.ops[pc].dst == sl) ok = false;
    }
  }
  // a value the sink reads whose final producer is early must not be overwritten by a late op
  for (int sl = 0; sl < kSlots; sl++) {
    if (!((live_out >> sl) & 1u)) continue;
    int def = -1;
    for (int pc = 0; pc < s.n_ops; pc++) if (s.ops[pc].code != OP_NOP && s.ops[pc].dst == sl) def = pc;
    if (def < 0 || !((early >> def) & 1u)) continue;
    for (int pc = 0; pc < s.n_ops; pc++) if (!((early >> pc) & 1u) && s.ops[pc].code != OP_NOP && s.ops[pc].dst == sl) ok = false;
  }
  if (!ok) early = all;
  return ProgramSplit{early, (early & all) != all};
}

PLX_HD constexpr bool shape_has_op(const Shape& s, uint8_t code) {
  for (int pc = 0; pc < s.n_ops; pc++) if (s.ops[pc].code == code) return true;
  return false;
}

struct Input {
  const void* values;
  const uint64_t* validity;
};
constexpr int kMaxLuts = 2;
struct Lut {
  const unsigned long long* bits;   // [range / 64 + 1] words
  uint64_t range;
};

struct Args {
  Input in[kMaxInputs];
  uint64_t imm[kMaxOps];
  Lut lut[kMaxLuts];
  int64_t n_rows;
  // generic interpreter only: its register file lives in dynamic LDS behind the sink's own LDS (set by the launcher)
  uint32_t rf_lds_offset;
  uint32_t rf_slots;
};
// slots a program touches (size of the generic interpreter's LDS register file)
inline uint32_t program_slots(const Shape& s) {
  uint32_t m = 0;
  auto up = [&](uint32_t v) { if (v != kNone && v + 1 > m) m = v + 1; };
  for (int i = 0; i < s.n_ops; i++) { up(s.ops[i].dst); }
  up(s.pred); up(s.key);
  for (int i = 0; i < s.n_keys; i++) up(s.keys[i]);
  for (int i = 0; i < s.n_aggs; i++) up(s.aggs[i].src);
  return m ? m : 1;
}

// identity element of an aggregate, as a 64-bit pattern
inline uint64_t agg_identity(uint8_t kind) {
  switch (kind) {
    case AGG_MIN_F: return 0x7ff0000000000000ull;  // +inf
    case AGG_MAX_F: return 0xfff0000000000000ull;  // -inf
    case AGG_MIN_I: return 0x7fffffffffffffffull;
    case AGG_MAX_I: return 0x8000000000000000ull;
    case AGG_MIN_U: return ~0ull;
    case AGG_MAX_U: return 0ull;
    case AGG_FIRST_ROW: return ~0ull;
    default: return 0ull;
  }
}

// ---- result finalisation (aggregate cells -> output column) -------------------------
enum FinalKind : uint8_t {
  FIN_COPY64 = 0,  // out (8 B) = cell a
  FIN_TRUNC32,     // out (4 B) = low 32 bits of cell a (i32/u32 wrapping sums, u32 counts)
  FIN_MEAN,        // out f64 (or f32) = f64(cell a) / count(cell b); null when count == 0
  FIN_MINMAX_I,    // out (width by dtype) = cell a; null when count(cell b) == 0
  FIN_MINMAX_F,    // out f64 = cell a; null when count(b) == 0; NaN when ordered count(c) == 0
  FIN_NARROW       // out (1/2 B) = low bits of cell a
};
struct FinalSpec {
  uint8_t kind, a, b, c, out_dtype;
};
// ---- key decoding (packed group key -> key column) ---------------------------------
struct KeyDecode {
  int32_t shift;        // bit position of this key's code inside the packed key
  uint64_t mask;        // (1 << bits) - 1, or ~0 for a raw 64-bit key
  int64_t min;          // value = code + min
  uint64_t null_code;   // code standing for NULL, or ~0 if none (raw keys use the valid flag)
  int32_t dtype;        // output plx_dtype
};

// ---- partitioned group-by (kernels_partition.hip) ---------------------------------------------
#if defined(__HIPCC__) || defined(__HIP__)
#define PLX_FHD __host__ __device__
#else
#define PLX_FHD
#endif
constexpr int kMaxSrc = 4;     // distinct aggregate sources a record carries (more => the HBM-table sink runs)
// Record layout: [key, source 0 .. n_src-1, (validity word), (row id)] as u64 words.  A pure function of the program
// shape, so AOT / JIT kernels have it as a compile-time constant and the host derives the identical layout.
struct RecLayout {
  uint32_t n_src;              // distinct aggregate source slots (> kMaxSrc: not representable)
  uint32_t has_valid;          // validity word (bit j = source j valid, bit 63 = key valid)
  uint32_t has_rowid;          // row index (AGG_FIRST_ROW)
  uint32_t rec_words;
  uint8_t src_slot[kMaxAggs];  // program slot of source j
  uint8_t agg_src[kMaxAggs];   // aggregate k reads source agg_src[k] (kNone: LEN / FIRST_ROW)
};
PLX_FHD constexpr bool shape_may_have_nulls(const Shape& sh) {
  for (int i = 0; i < sh.n_inputs; i++) if (sh.in_nullable[i]) return true;
  for (int i = 0; i < sh.n_ops; i++) if (sh.ops[i].code >= OP_FDIV_I && sh.ops[i].code <= OP_MOD_U) return true;   // integer div / mod: divisor 0 -> null
  return false;
}
PLX_FHD constexpr RecLayout rec_layout(const Shape& sh) {
  RecLayout L{};
  for (int k = 0; k < kMaxAggs; k++) { L.agg_src[k] = kNone; L.src_slot[k] = 0; }
  for (int k = 0; k < sh.n_aggs; k++) {
    const uint8_t kind = sh.aggs[k].kind;
    if (kind == AGG_LEN) continue;
    if (kind == AGG_FIRST_ROW) { L.has_rowid = 1; continue; }
    int j = -1;
    for (uint32_t t = 0; t < L.n_src; t++) if (L.src_slot[t] == sh.aggs[k].src) j = (int)t;
    if (j < 0) { j = (int)L.n_src; if (L.n_src < (uint32_t)kMaxAggs) L.src_slot[L.n_src] = sh.aggs[k].src; L.n_src++; }
    L.agg_src[k] = (uint8_t)j;
  }
  L.has_valid = shape_may_have_nulls(sh) ? 1 : 0;
  L.rec_words = 1 + L.n_src + L.has_valid + L.has_rowid;
  return L;
}
struct PartitionPlan {
  uint32_t log2_parts;         // P = 1 << log2_parts hash partitions
  uint32_t log2_slots;         // slots of the per-partition LDS table
  uint32_t buf_rows;           // records per LDS write-combining buffer (even)
  RecLayout rec;               // == rec_layout(shape)
};

// ---- partitioned group-by, second generation (partition2_device.hpp, kernels_partition.hip) -------------------------
// No counting pass: every scatter workgroup owns a private region of fixed-size CHUNKS and hands them to partitions on
// demand; records are packed dword streams (32-bit keys / sources stay 32-bit); partitions are aggregated either in an LDS
// open-addressing table (hash mode) or, for dense packed ids, in an LDS direct-address table (direct mode: the partition is
// the id's high bits, the record carries only the low bits).
constexpr uint32_t kP2Hash = 0, kP2Direct = 1;
constexpr uint32_t kP2ChunkRecs = 256;        // records per chunk: chunk bytes = 256 * rec_words * 4 (a multiple of the 128-B line)
constexpr uint32_t kP2MaxHot = 256;           // hot keys pre-aggregated in the scatter pass
constexpr uint32_t kNoChunk = 0xffffffffu;
// Record packing (third generation scatter, partition3_device.hpp; 0 everywhere else):
//   kPackNone    sources keep their column width
//   kPackNarrow  every 64-bit INTEGER source that is a plain column load is stored as a u32 offset from a run-time base
//                (PartPlan2::src_base: the column's cached minimum; the planner checks max - min < 2^32)
//   kPackFused   direct mode, ONE integer source, no validity / row id: the whole record is one dword, key_low | (v - base) << key_shift
//                (the planner checks key_shift + bits(max - min) <= 32)
//   kPackRowid   direct mode, no source, row id payload (the partitioned join probe): two dwords, key_low | row << key_shift (a 64-bit
//                field: key_shift + 32 row bits always fit); null keys never become records, so there is no validity dword
//   kPackPair    direct mode, ONE 64-bit source that does not narrow (an f64 sum over dictionary codes: config 5), no validity, no row id, slots < 2^16 - 1: the rows of a
//                partition travel TWO to a record of five dwords {slot0 | slot1 << 16, value0, value1} -- 10 bytes a row instead of 12.  Pairs are formed in the scatter's
//                tile sort (rank r of a partition's rows in the tile -> pair r / 2, half r % 2); a partition with an odd number of rows in a tile closes its last pair
//                with the slot 0xffff ("absent": ~1.5 % more pairs at 8192-row tiles and 256 partitions).  RecLayout2 describes ONE ROW (three dwords, as kPackNone) and
//                says rec_words = 5: the unit of the record stream, of the chunks and of chunk_fill is the PAIR.
//                Hash mode (a 64-bit key whose range is KNOWN to span < 2^48 - 1, one 64-bit value): seven dwords {off0 lo, off1 lo, off0 hi16 | off1 hi16 << 16, value0,
//                value1} with off = key - PartPlan2::key_base -- 14 bytes a row instead of 16; an absent half has the offset 2^48 - 1
//   kPackPairV   hash mode, a 64-bit key of ANY range and one Int64 value COLUMN whose range spans < 2^48 - 2^32 (config 3 on sparse keys: values spanning 2^41): the same
//                seven-dword pair with the roles swapped -- {key0, key1, voff0 lo, voff1 lo, voff0 hi16 | voff1 hi16 << 16}, voff = value - PartPlan2::src_base[0]; an absent
//                half has the high half-word 0xffff.  Bounds the planner only assumed are checked per row (pp.check_src) like every narrowed value
constexpr uint32_t kPackNone = 0, kPackNarrow = 1, kPackFused = 2, kPackRowid = 3, kPackPair = 4, kPackPairV = 5;
constexpr unsigned long long kPairVLimit = 0xffff00000000ull;      // value offsets stay below: the half-word 0xffff marks an absent half
constexpr uint32_t kPairAbsent = 0xffffu;
constexpr unsigned long long kPairAbsent48 = 0xffffffffffffull;
struct RecLayout2 {
  uint8_t n_key_cols;            // 0: one key slot (Shape::key); 2..kMaxKeys: a wide key -- Shape::keys, one 64-bit word per key column, null keys flagged in the validity dword
  uint8_t key_words;             // 1 | 2 dwords (wide key: 2 per key column)
  uint8_t key_kind;              // 0: 64-bit, 1: i32 (sign-extended on read), 2: u32 (zero-extended)
  uint8_t pack;                  // kPack*
  uint8_t n_src;
  uint8_t has_valid, valid_off;  // one dword: bit j = source j valid, bit 31 = key valid (wide key: bit 24 + i = key column i valid)
  uint8_t has_rowid, rowid_off;  // two dwords
  uint8_t rec_words;
  uint8_t row_words;             // dwords ONE ROW occupies in the scatter's registers and in its tile budget (== rec_words but for kPackPair)
  uint8_t src_kind[kMaxSrc], src_off[kMaxSrc];   // kind 0: 64-bit, 1: i32, 2: u32, 3: u32 offset from PartPlan2::src_base[j] (packing)
  uint8_t src_slot[kMaxAggs];    // program slot of source j
  uint8_t agg_src[kMaxAggs];     // aggregate k reads source agg_src[k] (kNone: LEN / FIRST_ROW)
};
// the input column a slot is loaded from when its only writer is a LOAD, else -1
PLX_FHD constexpr int slot_input(const Shape& sh, uint8_t slot) {
  int writers = 0, input = -1;
  for (int i = 0; i < sh.n_ops; i++) {
    if (sh.ops[i].dst != slot) continue;
    writers++;
    input = sh.ops[i].code == OP_LOAD ? (int)sh.ops[i].a : -1;
  }
  return writers == 1 ? input : -1;
}
// a slot that holds a plain 64-bit INTEGER column (candidates for kPackNarrow / kPackFused)
PLX_FHD constexpr bool slot_is_int64_column(const Shape& sh, uint8_t slot) {
  const int in = slot_input(sh, slot);
  return in >= 0 && (sh.in_dtype[in] == 4 || sh.in_dtype[in] == 8);     // PLX_I64, PLX_U64
}
// a slot whose only writer is the LOAD of a <= 32-bit integer column holds a sign- / zero-extended 32-bit value
PLX_FHD constexpr uint8_t narrow_kind(const Shape& sh, uint8_t slot) {
  int writers = 0, input = -1;
  for (int i = 0; i < sh.n_ops; i++) {
    if (sh.ops[i].dst != slot) continue;
    writers++;
    input = sh.ops[i].code == OP_LOAD ? (int)sh.ops[i].a : -1;
  }
  if (writers != 1 || input < 0) return 0;
  switch (sh.in_dtype[input]) {
    case 1: case 2: case 3: return 1;            // PLX_I8, PLX_I16, PLX_I32
    case 0: case 5: case 6: case 7: return 2;    // PLX_BOOL, PLX_U8, PLX_U16, PLX_U32
    default: return 0;
  }
}
PLX_FHD constexpr RecLayout2 rec_layout2(const Shape& sh, uint32_t mode, uint32_t pack = kPackNone) {
  RecLayout2 L{};
  L.pack = (uint8_t)pack;
  for (int k = 0; k < kMaxAggs; k++) { L.agg_src[k] = kNone; L.src_slot[k] = 0; }
  for (int j = 0; j < kMaxSrc; j++) { L.src_kind[j] = 0; L.src_off[j] = 0; }
  uint32_t w = 0, n_src = 0;
  L.n_key_cols = sh.n_keys;
  if (sh.n_keys) { L.key_kind = 0; L.key_words = (uint8_t)(2 * sh.n_keys); }      // hash mode only (the planner never plans direct partitions for a wide key)
  else {
    L.key_kind = mode == kP2Direct ? 2 : narrow_kind(sh, sh.key);
    L.key_words = L.key_kind ? 1 : 2;
  }
  w += L.key_words;
  for (int k = 0; k < sh.n_aggs; k++) {
    const uint8_t kind = sh.aggs[k].kind;
    if (kind == AGG_LEN) continue;
    if (kind == AGG_FIRST_ROW) { L.has_rowid = 1; continue; }
    int j = -1;
    for (uint32_t t = 0; t < n_src; t++) if (L.src_slot[t] == sh.aggs[k].src) j = (int)t;
    if (j < 0) { j = (int)n_src; if (n_src < (uint32_t)kMaxAggs) L.src_slot[n_src] = sh.aggs[k].src; n_src++; }
    L.agg_src[k] = (uint8_t)j;
  }
  L.n_src = (uint8_t)n_src;
  for (uint32_t j = 0; j < n_src && j < (uint32_t)kMaxSrc; j++) {
    L.src_kind[j] = narrow_kind(sh, L.src_slot[j]);
    if (pack != kPackNone && pack != kPackPair && pack != kPackPairV && slot_is_int64_column(sh, L.src_slot[j])) L.src_kind[j] = 3;      // (a pair's values travel whole, or as 48-bit offsets formed in the tile sort)
    L.src_off[j] = (uint8_t)w;
    w += L.src_kind[j] ? 1 : 2;
  }
  if (pack == kPackFused) { w = 1; L.src_off[0] = 0; }      // planner-checked: direct mode, one integer source, no validity, no row id
  L.has_valid = shape_may_have_nulls(sh) ? 1 : 0;
  if (pack == kPackRowid) { L.has_valid = 0; L.valid_off = 0; L.rowid_off = 0; L.rec_words = 2; L.row_words = 2; return L; }
  L.valid_off = (uint8_t)w; w += L.has_valid;
  L.rowid_off = (uint8_t)w; w += 2 * L.has_rowid;
  L.rec_words = (uint8_t)w; L.row_words = (uint8_t)w;
  if (pack == kPackPair) L.rec_words = mode == kP2Direct ? 5 : 7;      // planner-checked (pair_pack_ok): a row is {slot | 64-bit key, 64-bit value}; two rows share a record
  if (pack == kPackPairV) L.rec_words = 7;
  return L;
}
// does the shape admit kPackPair at all (the planner still checks the slot count)?
PLX_FHD constexpr bool pair_pack_ok(const Shape& sh, uint32_t mode) {
  const RecLayout2 L = rec_layout2(sh, mode, kPackNone);
  return !sh.n_keys && L.n_src == 1 && L.src_kind[0] == 0 && !L.has_valid && !L.has_rowid && L.rec_words == (mode == kP2Direct ? 3 : 4);      // (hash mode: a 64-bit key)
}
PLX_FHD constexpr bool pairv_pack_ok(const Shape& sh, uint32_t mode) {
  const RecLayout2 L = rec_layout2(sh, mode, kPackNone);
  return mode == kP2Hash && !sh.n_keys && L.key_words == 2 && L.n_src == 1 && L.src_kind[0] == 0 && slot_is_int64_column(sh, L.src_slot[0]) && !L.has_valid && !L.has_rowid && L.rec_words == 4;
}
// dwords a ROW occupies in the scatter's registers and LDS tile (kPackPair: as unpacked; two rows then share a record of 2 x row_words - 1 dwords)
PLX_FHD constexpr uint32_t scatter_row_words(const RecLayout2& L) { return (uint32_t)L.row_words; }
struct PartPlan2 {
  uint32_t mode;               // kP2Hash | kP2Direct
  uint32_t log2_parts;         // P = 1 << log2_parts partitions
  uint32_t log2_slots;         // slots of a partition's LDS table (direct mode: == key_shift; hash mode: 0 when n_slots is not a power of two)
  uint32_t n_slots;            // hash mode: slots of the partition's LDS open-addressing table (any number: slot = mulhi(hash32, n_slots); + 2 special slots behind them)
  uint32_t key_shift;          // direct mode: partition = id >> key_shift, table slot = id & ((1 << key_shift) - 1)  (interleave: see below; key_shift stays the slot bits)
  uint32_t ring_lines;         // 128-B lines of LDS staging per partition (power of two)
  uint32_t block;              // threads of a scatter workgroup
  uint32_t tiles;              // tiles each wave loads per round (template parameter of the scatter kernel)
  uint32_t chunks_per_wg;      // chunks in each scatter workgroup's private region
  uint32_t scatter_grid;
  uint32_t n_hot;              // hot keys (0 = none), log2_hot_slots = slots of their LDS lookup table, hot_copies = accumulator copies
  uint32_t log2_hot_slots, hot_copies;
  uint32_t len_idx;            // the aggregate that counts rows (occupancy of a direct-address slot)
  uint32_t rec_words;          // == rec_layout2(shape, mode).rec_words
  uint32_t ablate;             // measurement only (PLX_PART_ABLATE, results are WRONG when set): 1 = flush without the HBM stores, 2 = append without the ring writes
  uint32_t gen;                // scatter generation: 2 = rings + line flush (partition2_device.hpp), 3 = tile sort + carry lines (partition3_device.hpp)
  uint32_t pack;               // kPack* (gen 3 only): == rec_layout2(shape, mode, pack).pack
  int64_t src_base[kMaxSrc];   // packing: value a kind-3 source is stored relative to
  int64_t key_base;            // direct mode: the dense id is key - key_base (0: the program already produces dense ids)
  uint32_t oob_drop;           // direct mode: 1 = rows whose id lies outside the partitions are dropped (join probe: such keys match nothing); 0 = the query fails
  uint32_t wide_null_word;     // wide key with a nullable key column: the LDS tables keep the null mask as one more key word
  uint32_t interleave;         // direct mode, group-by: partition = the id's LOW log2_parts bits, table slot = id >> log2_parts (ids are usually handed out in order of
                               // first appearance or popularity -- dictionary codes, zipf-like keys: the high bits would put all popular ids into partition 0)
  uint32_t hash_bits;          // direct mode, join probe on keys WITHOUT a usable range: the "dense id" is the top hash_bits bits of key * kP2HashMult (0: key - key_base)
  uint32_t slice;              // direct mode, join probe on a key range: partition = id / slice, record low = id % slice (slice = range / P rounded up to a multiple of 64, so
                               // every partition is populated whatever the range; 0: partition = id >> key_shift).  slice_magic = floor(2^64 / slice); key_shift = bits of `low`
  unsigned long long slice_magic;
  uint32_t n_tags;             // hash mode, wide keys: entries of the partition's LDS tag table (a multiple of 8: buckets of eight; ~4 per group the storage holds); n_slots = groups
  uint32_t check_src;          // 1: the value bases (src_base) come from unverified bounds: a value that does not fit its narrowed field raises the scatter's flag [2]
};
// the multiplier of the join hash tables' slot hash (JoinBuildSink / ProbeAggSink; the reference's DirtyHash, polars-utils/src/hashing.rs:62-69): the hashed
// partitioned probe takes its partition from the SAME top bits, so partition p of the probe side meets exactly region p of the build table
constexpr unsigned long long kP2HashMult = 0x55fbfd6bfc5458e9ull;
// which packing a shape admits at all (the planner still has to check the value ranges): kPackFused / kPackNarrow / kPackNone
PLX_FHD constexpr uint32_t best_static_pack(const Shape& sh, uint32_t mode) {
  const RecLayout2 L = rec_layout2(sh, mode, kPackNarrow);
  bool any = false;
  for (uint32_t j = 0; j < L.n_src && j < (uint32_t)kMaxSrc; j++) any = any || L.src_kind[j] == 3;
  if (mode == kP2Direct && L.n_src == 1 && L.src_kind[0] != 0 && !L.has_valid && !L.has_rowid) return kPackFused;
  return any ? kPackNarrow : kPackNone;
}

// ---- batched result finalisation: every output column of a query in ONE launch ---------------
constexpr int kMaxFinJobs = 24;
struct FinJob {
  uint8_t is_key;                     // 1: decode a group key (kd, packed, kvalid); 0: finalise an aggregate (fs)
  FinalSpec fs;
  KeyDecode kd;
  const unsigned long long* packed;
  const unsigned char* kvalid;
  void* out;
  uint64_t* out_valid;                // may be null
};
struct FinBatch {
  int32_t n;
  FinJob jobs[kMaxFinJobs];
};

// ---- AOT specialisation table ---------------------------------------------------
// Index into kStaticShapes (fused_shapes.hpp); -1 = run the generic interpreter.
int find_static_shape(const Shape& s);

// ---- sinks ------------------------------------------------------------------------
// Hash aggregation table in HBM (open addressing, linear probing; capacity C = 2^k).
// keys[C+2]: slot C = the null-key group, slot C+1 = the group of the key whose bit
// pattern equals the EMPTY sentinel.  acc[(C+2) * n_aggs] initialised to identities.
constexpr uint64_t kEmptyKey = ~0ull;
struct HashTable {
  unsigned long long* keys;
  unsigned long long* acc;
  unsigned int* overflow;  // set when a probe sequence exceeds max_probe
  uint32_t log2_cap;
  uint32_t max_probe;
  uint32_t wave_combine;   // 1: rows of a wave that share a key are combined in registers first, ONE lane updates the table (the
                           // planner's sample pass: a key holding half of the rows must not cost half a million atomics on one address)
};

// Wide-key hash aggregation table (group keys that do not pack into one 64-bit word; the
// reference row-encodes them: group_by/mod.rs:88-94).  Open addressing on a 63-bit tag of all
// key words; a slot goes EMPTY -> tag|BUSY (claimed by CAS) -> tag (key words published):
//   tags[cap], words[n_words][cap] (word j of slot s at words[j*cap + s]; the last word is the
//   null mask when any key column is nullable), acc[cap * n_aggs].
constexpr uint64_t kBusyBit = 1ull << 63;
struct WideTable {
  unsigned long long* tags;
  unsigned long long* words;
  unsigned long long* acc;
  unsigned int* overflow;
  uint32_t log2_cap;
  uint32_t max_probe;
  uint32_t n_words;      // n_keys (+1 when has_null_word)
  uint32_t has_null_word;
};

// Fused join -> aggregate (group keys = join key + build-side columns, unique build keys):
// the build scan inserts key -> build row, the probe scan adds straight into the cells of the
// matching slot.  keys[cap+1] / head[cap+1] (slot cap = the key whose bits equal EMPTY),
// acc[(cap+1) * n_aggs].
constexpr uint32_t kNoRow32 = 0xffffffffu;
struct JoinAggTable {
  // [cap + 1] slots of 16 bytes: {key, build row (low 32 bits of the second word)}.  Key and row share a line: an insert is ONE line across the fabric (the CAS on the
  // key; the row store merges into the line the CAS just brought in), and so is a probe.  Slot `cap` = the key whose bits equal kEmptyKey (present iff its row is set).
  unsigned long long* slots;
  unsigned int* flags;    // [0] = duplicate build key seen, [1] = probe sequence overflow
  unsigned long long* acc;
  uint32_t log2_cap;
  unsigned long long* count;   // build: [0] += rows inserted (null: not wanted) -- the build side's row count after its predicate, a by-product of the build scan
  // Multi-value mode (non-null): build keys may repeat (the reference's hash tables map a key to a LIST of build rows, single_keys.rs:16-167).  links[build row] =
  // {next build row with the same key (low 32 bits; kNoRow32 ends the chain), representative row (high 32 bits)}; the slot's row word is the HEAD of its key's chain
  // (exchanged in by every insert), the word above it counts the key's duplicate inserts (from its 0xffffffff fill).  A group of the fused join -> aggregate is then
  // a build ROW: `acc` is [build rows][n_aggs] and a probe row adds to the cells of the representative of every row of its key's chain -- the representative of a
  // row is the first row of the chain that agrees with it on the build-side group columns (canonicalise_chains), so build rows that are ONE group share one cell set.
  unsigned long long* links;
  // Probe sequences wrap inside aligned WINDOWS of 2^log2_window slots (0: inside the whole table).  A table filled region by region from LDS (k::partitioned_join_build)
  // is made of such windows: each is built by one workgroup with nothing of its keys outside it.  Every site that walks a probe sequence goes through jt_next.
  // A windowed table numbers its keys densely: the high 32 bits of a slot's second word hold the key's CELL index (jt_cell), so `acc` is [keys + 1][n_aggs] instead of
  // [cap + 1][n_aggs] -- less than half the cells to initialise and to stream through at the output step.
  uint32_t log2_window;
};
PLX_HD inline uint64_t jt_next(const JoinAggTable& t, uint64_t slot) {
  const uint64_t wmask = (1ull << (t.log2_window ? t.log2_window : t.log2_cap)) - 1ull;
  return (slot & ~wmask) | ((slot + 1) & wmask);
}
// build-side columns on which two build rows of one key must agree to be ONE group (canonicalise_chains): plain fixed-width values + optional validity bitmap
constexpr int kMaxRepCols = 6;
struct RepCols { int n; const void* vals[kMaxRepCols]; const unsigned long long* valid[kMaxRepCols]; int width[kMaxRepCols]; };
PLX_HD inline unsigned long long* jt_key(const JoinAggTable& t, uint64_t s) { return t.slots + 2 * s; }
PLX_HD inline unsigned int* jt_row(const JoinAggTable& t, uint64_t s) { return reinterpret_cast<unsigned int*>(t.slots + 2 * s + 1); }
PLX_HD inline uint64_t jt_cell(const JoinAggTable& t, uint64_t s) { return t.log2_window ? (uint64_t)jt_row(t, s)[1] : s; }      // index of the slot's aggregate cells

// Direct-address ("perfect hash") variant of the fused join -> aggregate table, used when the build
// key range is small (max - min + 1 <= a few x the build rows, e.g. TPC-H orderkeys).  One BIT per key of the
// range says whether a build row with that key passed the build predicate; rank[w] (u32) = number of set bits before
// 64-bit word w, so slot(key) = rank[w] + popc(bits of the word below the key's bit) numbers the build rows in key order.
// bits + rank are range/8 + range/16 bytes (75 + 37 MB for the 6e8 TPC-H SF100 orderkeys: resident in the 256 MB Infinity
// Cache), where one u32 per key was 2.4 GB to memset and to miss in.
//   build  : the scan appends (key, row) pairs in per-wave ordinal chunks and sets the key's bit;
//   rank   : popcount per 512-bit block -> device exclusive scan over the blocks -> per-word ranks; the pairs are counted
//            too: more pairs than set bits = a duplicate build key -> the caller falls back BEFORE the probe;
//   probe  : bit test + rank -> the row's aggregates land in acc[slot];
//   output : ONE pass over the pair list: pairs whose slot was hit by a probe row are compacted together with their cells
//            (no key-ordered copy of the pairs, no pass over the slots).
constexpr unsigned int kOrdChunk = 1024;     // ordinals a wave reserves at a time (fused_sinks.hpp DirectBuildSink)
struct DirectJoinTable {
  unsigned long long* bits;      // [(range / 512 + 1) * 8] bitmap words, padded to whole blocks
  const unsigned int* rank;      // [(range / 512 + 1) * 8] set bits before each bitmap word (valid after the rank step)
  unsigned long long* ord_key;   // pair list [n_ord]
  unsigned int* ord_row;         // same, build row
  unsigned int* chunk_used;      // [n_ord / kOrdChunk + 1] ordinals handed out of each reserved chunk
  unsigned int* counter;         // [0] next ordinal chunk base
  unsigned int* flags;           // [0] duplicate build key, [1] ordinal overflow
  unsigned long long* acc;       // [n_slots * n_aggs] (probe)
  long long kmin;
  unsigned long long range;
  unsigned int n_ord;            // capacity of the pair list
  unsigned int opts;             // kDirectLateLoads
  // output step after a PARTITIONED probe: bit (key * kP2HashMult) >> (64 - log2_touch_bits) is set for every candidate key the final probe ran over; a pair whose
  // bit is clear cannot have been matched, so the compaction skips its three random lookups (bitmap word, rank, LEN cell).  null: every pair is looked up.
  const unsigned long long* touch_filter;
  unsigned int log2_touch_bits;
};
constexpr unsigned int kDirectLateLoads = 1u;   // probe: columns only the aggregates read are loaded under the hit mask (split_program)

// The selection of a filter -> frame as the predicate scan leaves it (fused_sinks.hpp BallotSink): per 128-row wave tile t the two ballots of its rows (lane l of the
// wave holds rows 2l and 2l + 1: ballots[2t] = rows of even parity, ballots[2t + 1] = odd) and the number of kept rows.
struct BallotOut {
  unsigned long long* ballots;     // [n_wave_tiles][2]
  unsigned int* counts;            // [n_wave_tiles]
};

// Probe side of a join against a direct-address table, hits only: per 128-row wave tile the ballots of the rows that pass the probe predicate AND whose key's bit is set
// (fused_sinks.hpp DirectHitsSink) -- the exact candidate rows of the materialising join when the probe keys arrive in key order (the bitmap is then walked out of the L2).
struct DirectHits {
  DirectJoinTable t;
  BallotOut out;
};

// Semi-join filter side reduced to a bitmap over its key range (an inner join whose one side has unique keys and contributes
// no column downstream only FILTERS the other side): the scan of the filter side sets bit (key - kmin) of every row that passes
// its predicate and counts those rows (fewer set bits than rows = duplicate keys: the rewrite does not apply).
struct BitmapBuild {
  unsigned long long* bits;      // [range / 64 + 1], zeroed
  unsigned long long* count;     // [0] rows inserted
  long long kmin;
  unsigned long long range;
};

// Direct-address aggregation (dense keys in [key_min, key_min + n_groups)): acc[(G+1)*n_aggs],
// slot G = null key.
struct DenseTable {
  unsigned long long* acc;
  int64_t key_min;
  int64_t n_groups;
  unsigned int* oob;     // may be null: [0] = 1 when a key fell outside the table (see LdsAggSink::Params::oob)
};

}  // namespace fused
}  // namespace plx
