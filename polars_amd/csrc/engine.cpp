// engine.cpp -- plan import, dtype inference, the materialising evaluator, the fused
// pipeline compiler and the executors.  See engine.hpp for the reference mapping.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <memory>
#include <numeric>
#include <set>
#include <sstream>

#include "fused_shapes.hpp"
#include "jit.hpp"
#include "join.hpp"
#include "kernels.hpp"
#include "kernels_fused.hpp"
#include "ops.hpp"
#include "sort.hpp"

namespace plx {
namespace engine {

using namespace fused;

struct Unsupported : std::exception {
  std::string why;
  explicit Unsupported(std::string w) : why(std::move(w)) {}
  const char* what() const noexcept override { return why.c_str(); }
};

// ------------------------------------------------------------------ import ----
Plan import_plan(const plx_ir* ir, int n_ir, const plx_aexpr* ae, int n_ae, uint32_t flags) {
  Plan p;
  p.flags = flags;
  PLX_REQUIRE(ir && n_ir > 0, PLX_ERR_INVALID, "execute_plan: empty IR arena");
  for (int i = 0; i < n_ae; i++) {
    AE e;
    e.kind = ae[i].kind; e.op = ae[i].op; e.lhs = ae[i].lhs; e.rhs = ae[i].rhs; e.dtype = ae[i].dtype; e.is_null = ae[i].is_null; e.lit = ae[i].lit;
    if (ae[i].name) e.name = ae[i].name;
    PLX_REQUIRE(e.lhs < i && e.rhs < i, PLX_ERR_INVALID, "AExpr arena must be topologically ordered (children before parents)");
    p.ae.push_back(std::move(e));
  }
  for (int i = 0; i < n_ir; i++) {
    IRN n;
    n.kind = ir[i].kind; n.input = ir[i].input; n.input_right = ir[i].input_right; n.predicate = ir[i].predicate; n.frame = ir[i].frame;
    for (int j = 0; j < ir[i].n_exprs; j++) n.exprs.push_back(ir[i].exprs[j]);
    for (int j = 0; j < ir[i].n_keys; j++) n.keys.push_back(ir[i].keys[j]);
    for (int j = 0; j < ir[i].n_keys_right; j++) n.keys_right.push_back(ir[i].keys_right[j]);
    n.how = ir[i].how; n.maintain_order = ir[i].maintain_order;
    if (ir[i].suffix) n.suffix = ir[i].suffix;
    if (n.kind == PLX_IR_SORT) {
      for (int j = 0; j < ir[i].n_keys; j++) {
        n.sort_descending.push_back(ir[i].sort_descending ? ir[i].sort_descending[j] : 0);
        n.sort_nulls_last.push_back(ir[i].sort_nulls_last ? ir[i].sort_nulls_last[j] : 0);
      }
    }
    n.slice_offset = ir[i].slice_offset; n.slice_len = ir[i].slice_len;
    auto chk = [&](int e) { PLX_REQUIRE(e >= 0 && e < n_ae, PLX_ERR_INVALID, "IR node references an expression outside the arena"); };
    for (int e : n.exprs) chk(e);
    for (int e : n.keys) chk(e);
    for (int e : n.keys_right) chk(e);
    if (n.kind == PLX_IR_FILTER) chk(n.predicate);
    PLX_REQUIRE(n.input < i && n.input_right < i, PLX_ERR_INVALID, "IR arena must be topologically ordered (inputs before consumers)");
    p.ir.push_back(std::move(n));
  }
  return p;
}

// -------------------------------------------------------------- dtype rules ----
static int sum_out_dtype(int dt) {
  switch (dt) {
    case PLX_BOOL: return PLX_U32;
    case PLX_I8: case PLX_I16: case PLX_U8: case PLX_U16: return PLX_I64;
    default: return dt;
  }
}
static bool is_cmp_op(int op) { return op >= PLX_OP_EQ && op <= PLX_OP_GE; }
static bool is_arith_op(int op) { return op >= PLX_OP_PLUS && op <= PLX_OP_MODULUS; }
static bool is_logic_op(int op) { return op >= PLX_OP_AND && op <= PLX_OP_XOR; }

int infer_dtype(const Plan& plan, int e, const Frame& schema) {
  const AE& x = plan.ae.at(e);
  switch (x.kind) {
    case PLX_AE_COLUMN: {
      int i = schema.find(x.name);
      if (i < 0) fail(PLX_ERR_NOT_FOUND, "column not found: " + x.name);
      return schema.cols[i]->dtype;
    }
    case PLX_AE_LITERAL: return x.dtype;
    case PLX_AE_ALIAS: return infer_dtype(plan, x.lhs, schema);
    case PLX_AE_CAST: return x.dtype;
    case PLX_AE_NOT: case PLX_AE_IS_NULL: case PLX_AE_IS_NOT_NULL: return PLX_BOOL;
    case PLX_AE_FILL_NULL: return infer_dtype(plan, x.lhs, schema);
    case PLX_AE_LEN: return PLX_U32;
    case PLX_AE_AGG: {
      int in = infer_dtype(plan, x.lhs, schema);
      switch (x.op) {
        case PLX_AGG_SUM: return sum_out_dtype(in);
        case PLX_AGG_MEAN: return in == PLX_F32 ? PLX_F32 : PLX_F64;
        case PLX_AGG_COUNT: case PLX_AGG_LEN: return PLX_U32;
        default: return in;
      }
    }
    case PLX_AE_BINARY: {
      int l = infer_dtype(plan, x.lhs, schema), r = infer_dtype(plan, x.rhs, schema);
      if (is_cmp_op(x.op) || is_logic_op(x.op)) return PLX_BOOL;
      PLX_REQUIRE(l == r, PLX_ERR_INVALID, std::string("binary expression operands have different dtypes (") + dtype_name(l) + ", " + dtype_name(r) + "); the optimizer's type coercion must insert casts");
      if (x.op == PLX_OP_TRUE_DIVIDE && !dtype_is_float(l)) return PLX_F64;
      return l;
    }
    default: fail(PLX_ERR_UNSUPPORTED, "unsupported AExpr kind " + std::to_string(x.kind));
  }
}

std::string output_name(const Plan& plan, int e) {
  const AE& x = plan.ae.at(e);
  switch (x.kind) {
    case PLX_AE_ALIAS: return x.name;
    case PLX_AE_COLUMN: return x.name;
    case PLX_AE_LITERAL: return "literal";
    case PLX_AE_LEN: return "len";
    default: return x.lhs >= 0 ? output_name(plan, x.lhs) : "literal";  // leftmost leaf, like polars
  }
}

static bool contains_agg(const Plan& plan, int e) {
  if (e < 0) return false;
  const AE& x = plan.ae[e];
  if (x.kind == PLX_AE_AGG || x.kind == PLX_AE_LEN) return true;
  return contains_agg(plan, x.lhs) || contains_agg(plan, x.rhs);
}
static bool contains_column_outside_agg(const Plan& plan, int e) {
  if (e < 0) return false;
  const AE& x = plan.ae[e];
  if (x.kind == PLX_AE_AGG || x.kind == PLX_AE_LEN) return false;
  if (x.kind == PLX_AE_COLUMN) return true;
  return contains_column_outside_agg(plan, x.lhs) || contains_column_outside_agg(plan, x.rhs);
}
static void collect_aggs(const Plan& plan, int e, std::vector<int>& out) {
  if (e < 0) return;
  const AE& x = plan.ae[e];
  if (x.kind == PLX_AE_AGG || x.kind == PLX_AE_LEN) { if (std::find(out.begin(), out.end(), e) == out.end()) out.push_back(e); return; }
  collect_aggs(plan, x.lhs, out);
  collect_aggs(plan, x.rhs, out);
}

// -------------------------------------------------- materialising evaluator ----
// One kernel per node; every node materialises its column (reference-shaped).
// `overrides` maps expression ids to precomputed columns (aggregation results).
struct Evaluated { ColumnPtr col; bool scalar; };  // scalar: length-1, broadcastable

static plx_scalar scalar_of(const ColumnPtr& c, bool* valid) {
  plx_scalar s; s.u = 0;
  uint8_t v = 0xff; int32_t hv = 0;
  uint64_t buf[2] = {0, 0};
  column_to_host(c, buf, &v, &hv);
  *valid = v & 1;
  if (c->dtype == PLX_BOOL) s.u = buf[0] & 1; else memcpy(&s, buf, (size_t)dtype_width(c->dtype));
  return s;
}

static Evaluated eval(const Plan& plan, int e, const Frame& df, const std::map<int, ColumnPtr>* overrides) {
  check_cancel();
  if (overrides) { auto it = overrides->find(e); if (it != overrides->end()) return {it->second, false}; }
  const AE& x = plan.ae.at(e);
  switch (x.kind) {
    case PLX_AE_COLUMN: {
      int i = df.find(x.name);
      if (i < 0) fail(PLX_ERR_NOT_FOUND, "column not found: " + x.name);
      return {df.cols[i], false};
    }
    case PLX_AE_LITERAL: return {ops::full_column(x.dtype, x.lit, !x.is_null, 1), true};
    case PLX_AE_ALIAS: return eval(plan, x.lhs, df, overrides);
    case PLX_AE_CAST: { Evaluated c = eval(plan, x.lhs, df, overrides); return {ops::cast(c.col, x.dtype), c.scalar}; }
    case PLX_AE_NOT: { Evaluated c = eval(plan, x.lhs, df, overrides); return {ops::bool_not(c.col), c.scalar}; }
    case PLX_AE_IS_NULL: case PLX_AE_IS_NOT_NULL: {
      // the validity bitmap IS the answer: a Boolean column that shares it as its values (no kernel), never null itself
      Evaluated c = eval(plan, x.lhs, df, overrides);
      ColumnPtr b;
      if (c.col->validity && column_null_count(c.col) > 0) {
        b = std::make_shared<Column>();
        b->dtype = PLX_BOOL; b->len = c.col->len; b->values = c.col->validity; b->null_count = 0;
      } else {
        plx_scalar one; one.u = 1;
        b = ops::full_column(PLX_BOOL, one, true, c.col->len);
      }
      return {x.kind == PLX_AE_IS_NULL ? ops::bool_not(b) : b, c.scalar};
    }
    case PLX_AE_FILL_NULL: {
      Evaluated c = eval(plan, x.lhs, df, overrides);
      const AE& l = plan.ae.at(x.rhs);
      PLX_REQUIRE(l.kind == PLX_AE_LITERAL && !l.is_null, PLX_ERR_UNSUPPORTED, "fill_null with a non-literal value");
      PLX_REQUIRE(l.dtype == c.col->dtype, PLX_ERR_INVALID, std::string("fill_null literal dtype ") + dtype_name(l.dtype) + " differs from the column's " + dtype_name(c.col->dtype));
      return {ops::fill_null(c.col, l.lit), c.scalar};
    }
    case PLX_AE_LEN: { ops::ScalarValue s; s.dtype = PLX_U32; s.valid = true; s.v.u = (uint32_t)df.height; return {ops::scalar_column(s), true}; }
    case PLX_AE_AGG: {
      Evaluated c = eval(plan, x.lhs, df, overrides);
      return {ops::scalar_column(ops::reduce(x.op, c.col)), true};
    }
    case PLX_AE_BINARY: {
      Evaluated l = eval(plan, x.lhs, df, overrides), r = eval(plan, x.rhs, df, overrides);
      const bool both_scalar = l.scalar && r.scalar;
      if (is_logic_op(x.op)) {
        ColumnPtr a = l.col, b = r.col;
        if (l.scalar && !r.scalar) { bool v; plx_scalar s = scalar_of(a, &v); a = ops::full_column(PLX_BOOL, s, v, b->len); }
        if (r.scalar && !l.scalar) { bool v; plx_scalar s = scalar_of(b, &v); b = ops::full_column(PLX_BOOL, s, v, a->len); }
        return {ops::bool_binop(x.op - PLX_OP_AND, a, b), both_scalar};
      }
      if (is_cmp_op(x.op)) {
        const int op = x.op - PLX_OP_EQ;
        if (r.scalar && !l.scalar) { bool v; plx_scalar s = scalar_of(r.col, &v); PLX_REQUIRE(l.col->dtype == r.col->dtype, PLX_ERR_INVALID, "cmp: dtype mismatch"); return {ops::cmp_scalar(op, l.col, s, !v), false}; }
        if (l.scalar && !r.scalar) {
          // lit OP col == col OP' lit with the operator mirrored
          static const int mirror[6] = {PLX_EQ, PLX_NE, PLX_GT, PLX_GE, PLX_LT, PLX_LE};
          bool v; plx_scalar s = scalar_of(l.col, &v); PLX_REQUIRE(l.col->dtype == r.col->dtype, PLX_ERR_INVALID, "cmp: dtype mismatch");
          return {ops::cmp_scalar(mirror[op], r.col, s, !v), false};
        }
        return {ops::cmp(op, l.col, r.col), both_scalar};
      }
      PLX_REQUIRE(is_arith_op(x.op), PLX_ERR_UNSUPPORTED, "unsupported operator");
      const int op = x.op - PLX_OP_PLUS;
      if (r.scalar && !l.scalar) {
        bool v; plx_scalar s = scalar_of(r.col, &v);
        PLX_REQUIRE(l.col->dtype == r.col->dtype, PLX_ERR_INVALID, "arith: dtype mismatch");
        if (!v) return {ops::full_column(op == PLX_TRUE_DIV && !dtype_is_float(l.col->dtype) ? PLX_F64 : l.col->dtype, plx_scalar{0}, false, l.col->len), false};
        return {ops::arith_scalar(op, l.col, s, false), false};
      }
      if (l.scalar && !r.scalar) {
        bool v; plx_scalar s = scalar_of(l.col, &v);
        PLX_REQUIRE(l.col->dtype == r.col->dtype, PLX_ERR_INVALID, "arith: dtype mismatch");
        if (!v) return {ops::full_column(op == PLX_TRUE_DIV && !dtype_is_float(r.col->dtype) ? PLX_F64 : r.col->dtype, plx_scalar{0}, false, r.col->len), false};
        return {ops::arith_scalar(op, r.col, s, true), false};
      }
      return {ops::arith(op, l.col, r.col), both_scalar};
    }
    default: fail(PLX_ERR_UNSUPPORTED, "unsupported AExpr kind");
  }
}

static ColumnPtr broadcast(const Evaluated& ev, int64_t height) {
  if (!ev.scalar || ev.col->len == height) return ev.col;
  bool v; plx_scalar s = scalar_of(ev.col, &v);
  return ops::full_column(ev.col->dtype, s, v, height);
}

// ----------------------------------------------------- fused program compiler ----
struct DNode {
  uint8_t code = OP_NOP, c = 0;
  int a = -1, b = -1;
  uint64_t imm = 0;
  int col = -1;     // frame column index for OP_LOAD
  char ty = 'i';    // 'i' signed, 'u' unsigned 64, 'f' f64, 'b' bool
  bool nullable = false;
  int uses = 0, slot = -1;
  bool emitted = false;
};

class Compiler {
 public:
  Compiler(const Plan& p, const Frame& f) : plan(p), df(&f), n_rows(f.height) {}
  const Plan& plan;
  const Frame* df;               // name resolution / dtype inference scope (may be switched to a view of the same rows)
  int64_t n_rows;
  std::vector<ColumnPtr> cols;   // distinct input columns referenced so far
  std::vector<DNode> nodes;
  std::map<std::string, int> memo;
  std::vector<std::pair<uint8_t, int>> aggs;  // (kind, src node or -1)
  int pred = -1, key = -1;
  std::vector<int> wide_keys;  // raw key nodes of a wide (multi-word) group key

  int add(DNode n) {
    std::ostringstream k;
    k << (int)n.code << ':' << n.a << ':' << n.b << ':' << (int)n.c << ':' << n.imm << ':' << n.col << ':' << n.ty;
    auto it = memo.find(k.str());
    if (it != memo.end()) return it->second;
    nodes.push_back(n);
    memo[k.str()] = (int)nodes.size() - 1;
    return (int)nodes.size() - 1;
  }
  int mk(uint8_t code, int a, int b, char ty, uint8_t c = 0) {
    DNode n; n.code = code; n.a = a; n.b = b; n.ty = ty; n.c = c;
    n.nullable = (a >= 0 && nodes[a].nullable) || (b >= 0 && nodes[b].nullable);
    return add(n);
  }
  int konst(uint64_t bits, char ty) { DNode n; n.code = OP_CONST; n.imm = bits; n.ty = ty; return add(n); }
  int konst_f(double d) { uint64_t b; memcpy(&b, &d, 8); return konst(b, 'f'); }
  int ifnull(int a, uint64_t code) { DNode n; n.code = OP_IFNULL; n.a = a; n.b = a; n.imm = code; n.ty = nodes[a].ty; n.nullable = false; return add(n); }
  // membership of integer node `a` in lookup bitmap `lut` (args.lut[lut] is filled in by the caller before the launch)
  // node `a` with its validity cut down to the rows where boolean node `m` is valid and true (OP_MASKV)
  int mask_valid(int a, int m) { DNode n; n.code = OP_MASKV; n.a = a; n.b = m; n.ty = nodes[a].ty; n.nullable = true; return add(n); }
  int bit_lookup(int a, int lut, int64_t kmin) { DNode n; n.code = OP_BITLOOKUP; n.a = a; n.b = a; n.c = (uint8_t)lut; n.imm = (uint64_t)kmin; n.ty = 'b'; n.nullable = nodes[a].nullable; return add(n); }
  int col_id(const ColumnPtr& c) {
    for (size_t i = 0; i < cols.size(); i++) if (cols[i].get() == c.get()) return (int)i;
    cols.push_back(c);
    return (int)cols.size() - 1;
  }
  int load(int frame_col) {
    const ColumnPtr& c = df->cols[frame_col];
    const int col = col_id(c);
    DNode n; n.code = OP_LOAD; n.col = col; n.nullable = (bool)c->validity || c->null_count > 0;
    switch (c->dtype) {
      case PLX_F64: n.ty = 'f'; break;
      case PLX_U64: n.ty = 'u'; break;
      case PLX_BOOL: n.ty = 'b'; break;
      case PLX_F32: throw Unsupported("f32 columns are evaluated in f32 by the per-node kernels, not in the f64 fused program");
      default: n.ty = 'i'; break;
    }
    return add(n);
  }
  static uint64_t widen_literal(int dt, plx_scalar s) {
    switch (dt) {
      case PLX_I8: return (uint64_t)(int64_t)(int8_t)s.u;
      case PLX_I16: return (uint64_t)(int64_t)(int16_t)s.u;
      case PLX_I32: return (uint64_t)(int64_t)(int32_t)s.u;
      case PLX_U8: return s.u & 0xff;
      case PLX_U16: return s.u & 0xffff;
      case PLX_U32: return s.u & 0xffffffffull;
      case PLX_BOOL: return s.u & 1;
      default: return s.u;
    }
  }
  bool is_literal(int e, double* as_f64, int dt_hint) const {
    const AE* x = &plan.ae[e];
    while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs];
    if (x->kind != PLX_AE_LITERAL || x->is_null) return false;
    switch (x->dtype) {
      case PLX_F64: *as_f64 = x->lit.f64; return true;
      case PLX_F32: *as_f64 = x->lit.f32; return true;
      case PLX_U64: *as_f64 = (double)x->lit.u; return true;
      case PLX_BOOL: return false;
      default: *as_f64 = (double)(int64_t)widen_literal(x->dtype, x->lit); return true;
    }
    (void)dt_hint;
  }

  int lower(int e) {
    const AE& x = plan.ae.at(e);
    switch (x.kind) {
      case PLX_AE_COLUMN: {
        int i = df->find(x.name);
        if (i < 0) fail(PLX_ERR_NOT_FOUND, "column not found: " + x.name);
        return load(i);
      }
      case PLX_AE_LITERAL: {
        if (x.is_null) throw Unsupported("null literal");
        if (x.dtype == PLX_F32) throw Unsupported("f32 literal");
        char ty = x.dtype == PLX_F64 ? 'f' : x.dtype == PLX_U64 ? 'u' : x.dtype == PLX_BOOL ? 'b' : 'i';
        return konst(widen_literal(x.dtype, x.lit), ty);
      }
      case PLX_AE_ALIAS: return lower(x.lhs);
      case PLX_AE_NOT: { int a = lower(x.lhs); if (nodes[a].ty != 'b') throw Unsupported("not on non-boolean"); return mk(OP_NOT, a, a, 'b'); }
      case PLX_AE_IS_NULL: case PLX_AE_IS_NOT_NULL: {
        // valid(a) as a value, with the opcodes the kernels already have: (a ==bits a) is 1 with a's validity; IFNULL(.., 0) turns
        // the null into 0.  A source that cannot be null folds to a constant.
        int a = lower(x.lhs);
        int nn = nodes[a].nullable ? ifnull(mk(OP_CMP_U, a, a, 'b', (uint8_t)PLX_EQ), 0) : konst(1, 'b');
        return x.kind == PLX_AE_IS_NOT_NULL ? nn : mk(OP_NOT, nn, nn, 'b');
      }
      case PLX_AE_FILL_NULL: {
        const int dt = infer_dtype(plan, x.lhs, *df);
        const AE& l = plan.ae.at(x.rhs);
        if (l.kind != PLX_AE_LITERAL || l.is_null) throw Unsupported("fill_null with a non-literal value");
        if (l.dtype != dt) fail(PLX_ERR_INVALID, std::string("fill_null literal dtype ") + dtype_name(l.dtype) + " differs from the column's " + dtype_name(dt));
        if (dt == PLX_F32) throw Unsupported("f32 fill_null");
        int a = lower(x.lhs);
        return nodes[a].nullable ? ifnull(a, widen_literal(l.dtype, l.lit)) : a;
      }
      case PLX_AE_CAST: {
        int from = infer_dtype(plan, x.lhs, *df), to = x.dtype;
        int a = lower(x.lhs);
        if (from == to) return a;
        if (to == PLX_F64 && dtype_is_int(from)) return mk(nodes[a].ty == 'u' ? OP_U2F : OP_I2F, a, a, 'f');
        if (dtype_is_int(from) && (to == PLX_I64) && from != PLX_U64) return a;  // value-preserving widening
        if (dtype_is_unsigned(from) && to == PLX_U64) return a;
        throw Unsupported(std::string("cast ") + dtype_name(from) + " -> " + dtype_name(to) + " inside a fused program");
      }
      case PLX_AE_BINARY: {
        int ldt = infer_dtype(plan, x.lhs, *df), rdt = infer_dtype(plan, x.rhs, *df);
        if (is_logic_op(x.op)) {
          int a = lower(x.lhs), b = lower(x.rhs);
          if (nodes[a].ty != 'b' || nodes[b].ty != 'b') throw Unsupported("bitwise and/or on integers");
          int n = mk(x.op == PLX_OP_AND ? OP_AND : x.op == PLX_OP_OR ? OP_OR : OP_XOR, a, b, 'b');
          return n;
        }
        if (ldt != rdt) fail(PLX_ERR_INVALID, std::string("binary expression operands have different dtypes (") + dtype_name(ldt) + ", " + dtype_name(rdt) + ")");
        if (is_cmp_op(x.op)) {
          if (ldt == PLX_BOOL) throw Unsupported("comparison of boolean columns");
          int a = lower(x.lhs), b = lower(x.rhs);
          uint8_t code = nodes[a].ty == 'f' ? OP_CMP_F : nodes[a].ty == 'u' ? OP_CMP_U : OP_CMP_I;
          return mk(code, a, b, 'b', (uint8_t)(x.op - PLX_OP_EQ));
        }
        if (!(ldt == PLX_I64 || ldt == PLX_U64 || ldt == PLX_F64)) throw Unsupported(std::string("arithmetic on ") + dtype_name(ldt) + " wraps at the column width; done by the per-node kernels");
        const bool f = ldt == PLX_F64;
        switch (x.op) {
          case PLX_OP_PLUS: { int a = lower(x.lhs), b = lower(x.rhs); return mk(f ? OP_ADD_F : OP_ADD_I, a, b, nodes[a].ty); }
          case PLX_OP_MINUS: { int a = lower(x.lhs), b = lower(x.rhs); return mk(f ? OP_SUB_F : OP_SUB_I, a, b, nodes[a].ty); }
          case PLX_OP_MULTIPLY: { int a = lower(x.lhs), b = lower(x.rhs); return mk(f ? OP_MUL_F : OP_MUL_I, a, b, nodes[a].ty); }
          case PLX_OP_TRUE_DIVIDE: {
            int a = lower(x.lhs);
            if (!f) a = mk(nodes[a].ty == 'u' ? OP_U2F : OP_I2F, a, a, 'f');
            double lit;
            if (is_literal(x.rhs, &lit, rdt)) {
              // col / lit == col * (1 / lit)  (float.rs:113-115, signed.rs:218-221)
              int inv = konst_f(1.0 / lit);
              return mk(OP_MUL_F, a, inv, 'f');
            }
            int b = lower(x.rhs);
            if (!f) b = mk(nodes[b].ty == 'u' ? OP_U2F : OP_I2F, b, b, 'f');
            return mk(OP_DIV_F, a, b, 'f');
          }
          case PLX_OP_FLOOR_DIVIDE: case PLX_OP_MODULUS: {
            if (f) throw Unsupported("float floor-div / mod inside a fused program");
            int a = lower(x.lhs), b = lower(x.rhs);
            const bool u = nodes[a].ty == 'u';
            const uint8_t code = x.op == PLX_OP_FLOOR_DIVIDE ? (u ? OP_FDIV_U : OP_FDIV_I) : (u ? OP_MOD_U : OP_MOD_I);
            int n = mk(code, a, b, nodes[a].ty);
            nodes[n].nullable = true;   // divisor 0 -> null (signed.rs:35-70)
            return n;
          }
          default: throw Unsupported("operator inside a fused program");
        }
      }
      default: throw Unsupported("aggregation nested inside a row expression");
    }
  }

  int add_agg(uint8_t kind, int src) {
    for (size_t i = 0; i < aggs.size(); i++) if (aggs[i].first == kind && aggs[i].second == src) return (int)i;
    if ((int)aggs.size() >= kMaxAggs) throw Unsupported("more than " + std::to_string(kMaxAggs) + " distinct aggregates in one pass");
    aggs.push_back({kind, src});
    return (int)aggs.size() - 1;
  }
  int count_agg(int src) { return nodes[src].nullable ? add_agg(AGG_COUNT, src) : add_agg(AGG_LEN, -1); }

  // lowers one AGG / LEN node; returns how to finalise it
  FinalSpec lower_agg(int e) {
    const AE& x = plan.ae.at(e);
    FinalSpec fs{}; fs.a = fs.b = fs.c = kNone;
    if (x.kind == PLX_AE_LEN) { fs.kind = FIN_TRUNC32; fs.a = (uint8_t)add_agg(AGG_LEN, -1); fs.out_dtype = PLX_U32; return fs; }
    const int in_dt = infer_dtype(plan, x.lhs, *df);
    if (x.op == PLX_AGG_LEN) { fs.kind = FIN_TRUNC32; fs.a = (uint8_t)add_agg(AGG_LEN, -1); fs.out_dtype = PLX_U32; return fs; }
    // Boolean inputs: sum = number of true values (UInt32, sum_output_dtype) and mean = their fraction; a boolean slot holds 0 / 1,
    // so both reuse the integer cells.  min / max of booleans would need a bit-packed finalisation: per-node path.
    if (in_dt == PLX_BOOL && x.op != PLX_AGG_COUNT && x.op != PLX_AGG_SUM && x.op != PLX_AGG_MEAN) throw Unsupported("min / max of a boolean expression");
    int src = lower(x.lhs);
    switch (x.op) {
      case PLX_AGG_SUM:
        fs.out_dtype = (uint8_t)sum_out_dtype(in_dt);
        if (in_dt == PLX_F64) { fs.kind = FIN_COPY64; fs.a = (uint8_t)add_agg(AGG_SUM_F, src); }
        else { fs.kind = dtype_width(fs.out_dtype) == 4 ? FIN_TRUNC32 : FIN_COPY64; fs.a = (uint8_t)add_agg(AGG_SUM_I, src); }
        return fs;
      case PLX_AGG_MEAN: {
        int sf = src;
        if (in_dt != PLX_F64) sf = mk(nodes[src].ty == 'u' ? OP_U2F : OP_I2F, src, src, 'f');
        fs.kind = FIN_MEAN; fs.a = (uint8_t)add_agg(AGG_SUM_F, sf); fs.b = (uint8_t)count_agg(src); fs.out_dtype = PLX_F64;
        return fs;
      }
      case PLX_AGG_MIN: case PLX_AGG_MAX: {
        const bool mn = x.op == PLX_AGG_MIN;
        fs.out_dtype = (uint8_t)in_dt;
        if (in_dt == PLX_F64) { fs.kind = FIN_MINMAX_F; fs.a = (uint8_t)add_agg(mn ? AGG_MIN_F : AGG_MAX_F, src); fs.b = (uint8_t)count_agg(src); fs.c = (uint8_t)add_agg(AGG_COUNT_ORD, src); }
        else { fs.kind = FIN_MINMAX_I; fs.a = (uint8_t)add_agg(nodes[src].ty == 'u' ? (mn ? AGG_MIN_U : AGG_MAX_U) : (mn ? AGG_MIN_I : AGG_MAX_I), src); fs.b = (uint8_t)count_agg(src); }
        return fs;
      }
      case PLX_AGG_COUNT: fs.kind = FIN_TRUNC32; fs.a = (uint8_t)count_agg(src); fs.out_dtype = PLX_U32; return fs;
      default: throw Unsupported("aggregation kind");
    }
  }

  // ---- scheduling: DFS post-order, lowest-free-slot allocation, slots freed at last use
  Shape shape{};
  Args args{};
  std::vector<int> input_cols;
  std::vector<bool> slot_busy = std::vector<bool>(kSlots, false);

  void count_uses(int n, std::vector<bool>& seen) {
    if (n < 0) return;
    nodes[n].uses++;
    if (seen[n]) return;
    seen[n] = true;
    if (nodes[n].code != OP_LOAD && nodes[n].code != OP_CONST) {
      count_uses(nodes[n].a, seen);
      if (nodes[n].b != nodes[n].a) count_uses(nodes[n].b, seen);
    }
  }
  void release(int n) {
    if (n < 0) return;
    if (--nodes[n].uses == 0) slot_busy[nodes[n].slot] = false;
  }
  int emit(int n) {
    DNode& d = nodes[n];
    if (d.emitted) return d.slot;
    const bool leaf = d.code == OP_LOAD || d.code == OP_CONST;
    int sa = -1, sb = -1;
    if (!leaf) {
      sa = emit(d.a);
      sb = (d.b == d.a) ? sa : emit(d.b);
      release(d.a);
      if (d.b != d.a) release(d.b);
    }
    int slot = -1;
    for (int s = 0; s < kSlots; s++) if (!slot_busy[s]) { slot = s; break; }
    if (slot < 0) throw Unsupported("expression needs more than 16 live values");
    slot_busy[slot] = true;
    if (shape.n_ops >= kMaxOps) throw Unsupported("expression program longer than 32 ops");
    const int pc = shape.n_ops++;
    Op op{}; op.code = d.code; op.dst = (uint8_t)slot; op.c = d.c;
    if (d.code == OP_LOAD) {
      int idx = -1;
      for (size_t i = 0; i < input_cols.size(); i++) if (input_cols[i] == d.col) idx = (int)i;
      if (idx < 0) {
        if ((int)input_cols.size() >= kMaxInputs) throw Unsupported("more than 10 input columns");
        input_cols.push_back(d.col); idx = (int)input_cols.size() - 1;
        const ColumnPtr& c = cols[d.col];
        shape.in_dtype[idx] = (uint8_t)c->dtype; shape.in_nullable[idx] = d.nullable ? 1 : 0;
        args.in[idx].values = c->data(); args.in[idx].validity = c->valid_words();
      }
      op.a = (uint8_t)idx;
    } else if (d.code == OP_CONST) {
      args.imm[pc] = d.imm;
    } else {
      op.a = (uint8_t)sa; op.b = (uint8_t)sb;
      if (d.code == OP_IFNULL || d.code == OP_BITLOOKUP) args.imm[pc] = d.imm;
    }
    shape.ops[pc] = op;
    d.slot = slot; d.emitted = true;
    return slot;
  }
  void finish() {
    std::vector<bool> seen(nodes.size(), false);
    std::vector<int> roots;
    if (pred >= 0) roots.push_back(pred);
    if (key >= 0) roots.push_back(key);
    for (int kn : wide_keys) roots.push_back(kn);
    for (auto& a : aggs) if (a.second >= 0) roots.push_back(a.second);
    for (int r : roots) count_uses(r, seen);   // each root reference holds its slot to the end
    shape.pred = kNone; shape.key = kNone;
    // Loads first: every column load of a row tile is issued before the first dependent op, so a lane has all
    // its input columns in flight at once (memory-level parallelism) instead of load -> wait -> use per column.
    {
      std::vector<int> order;
      std::vector<bool> vis(nodes.size(), false);
      std::function<void(int)> dfs = [&](int n) {
        if (n < 0 || vis[n]) return;
        vis[n] = true;
        if (nodes[n].code == OP_LOAD) { order.push_back(n); return; }
        if (nodes[n].code == OP_CONST) return;
        dfs(nodes[n].a);
        if (nodes[n].b != nodes[n].a) dfs(nodes[n].b);
      };
      for (int r : roots) dfs(r);
      for (int n : order) emit(n);
    }
    if (pred >= 0) shape.pred = (uint8_t)emit(pred);
    if (key >= 0) shape.key = (uint8_t)emit(key);
    shape.n_keys = (uint8_t)wide_keys.size();
    for (size_t i = 0; i < wide_keys.size(); i++) shape.keys[i] = (uint8_t)emit(wide_keys[i]);
    shape.n_aggs = (uint8_t)aggs.size();
    for (size_t i = 0; i < aggs.size(); i++) {
      shape.aggs[i].kind = aggs[i].first;
      shape.aggs[i].src = aggs[i].second >= 0 ? (uint8_t)emit(aggs[i].second) : 0;
    }
    shape.n_inputs = (uint8_t)input_cols.size();
    args.n_rows = n_rows;
  }
};

// ----------------------------------------------------------- group keys -----------
struct KeyPart {
  int expr = -1;
  int dtype = 0;
  bool nullable = false;   // the key expression can be null (=> the output key column carries a validity bitmap)
  KeyDecode dec{};
};
struct KeyPlan {
  std::vector<KeyPart> parts;
  bool packed = false;   // keys bit-packed into a dense id (always valid)
  bool wide = false;     // 2..4 raw key columns compared word by word (WideAggSink)
  bool wide_nullable = false;
  std::vector<int> wide_nodes;
  int total_bits = 64;
  std::string note;      // what the planner did to get the key's range, for the plan description
};

static int ceil_log2_u64(uint64_t x) { int b = 0; while (b < 64 && (1ull << b) < x) b++; return b; }

// Lowers the group keys into one 64-bit key node. Narrow / multi-column keys are packed
// using column min/max statistics; a single wide key is used raw.
// Bounds of an integer column nobody has statistics for, GUESSED from a strided sample of 2^20 rows (1024 runs of 1024 rows: ~20 us) and widened by 1/1024 of the sampled
// span on either side (for uniformly spread values the expected gap between the sample's extremes and the column's is a millionth of the span).  They are installed as
// UNTRUSTED bounds (like plx_column_set_bounds): every kernel that addresses a table or narrows a value with them checks each row, so a wrong guess costs a second run of
// the query from an exact range pass (fused_groupby), never a wrong answer.  What this buys: one collect() over a column seen for the first time plans like every later
// one -- direct-address tables, narrowed values -- without the exact min / max passes over key and value columns (8 B / row each: 3 ms of a 9 ms first run at 1e9 rows).
static bool assume_ranges() { static const bool v = [] { const char* e = getenv("PLX_ASSUME_RANGES"); return !(e && e[0] == '0'); }(); return v; }
static bool assume_range(const ColumnPtr& col) {
  if (!assume_ranges() || !col || col->range_state != 0 || col->no_assume || !col->values || col->len < ((int64_t)1 << 24) || !dtype_is_int(col->dtype) || col->dtype == PLX_U64) return false;
  int64_t smn = 0, smx = 0;
  if (!k::sample_minmax(col, &smn, &smx, 1024)) return false;
  const unsigned __int128 span = (unsigned __int128)((__int128)smx - (__int128)smn);
  if (span >= ((unsigned __int128)1 << 62)) return false;
  const int64_t slack = (int64_t)(span >> 10) + 64;
  int64_t lo_lim = INT64_MIN, hi_lim = INT64_MAX;
  switch (col->dtype) {
    case PLX_I8: lo_lim = -128; hi_lim = 127; break; case PLX_U8: lo_lim = 0; hi_lim = 255; break;
    case PLX_I16: lo_lim = -32768; hi_lim = 32767; break; case PLX_U16: lo_lim = 0; hi_lim = 65535; break;
    case PLX_I32: lo_lim = INT32_MIN; hi_lim = INT32_MAX; break; case PLX_U32: lo_lim = 0; hi_lim = 0xffffffffll; break;
    default: break;
  }
  __int128 lo = std::max<__int128>((__int128)smn - slack, lo_lim);
  __int128 hi = std::min<__int128>((__int128)smx + slack, hi_lim);
  // non-negative values whose span costs the same number of bits from 0 as from their minimum: from 0 (ids and counts usually start there, and a key program without
  // the subtraction is the shape the ahead-of-time kernels were built for)
  if (smn >= 0) {
    auto bits_of = [](__int128 span) { int b = 1; while (b < 62 && ((__int128)1 << b) < span + 2) b++; return b; };
    if (bits_of(hi) == bits_of(hi - lo)) lo = 0;
  }
  // everything up to the next power of two above the span costs the same number of key bits (and value bits): take it -- a heavy-tailed id column (ids handed out by
  // popularity) shows its largest ids to no sample
  { int b = 1; while (b < 62 && ((__int128)1 << b) < hi - lo + 2) b++; hi = std::min<__int128>(lo + ((__int128)1 << b) - 2, hi_lim); }
  col->range_state = 1; col->range_min = (int64_t)lo; col->range_max = (int64_t)hi; col->range_trusted = false; col->range_assumed = true;
  return true;
}
static bool learn_dense_ranges() { const char* e = getenv("PLX_LEARN_DENSE_RANGE"); return !(e && e[0] == '0'); }   // (measurement / tests: 0 = the first run plans without a range pass)
static KeyPlan lower_keys(Compiler& c, const std::vector<int>& key_exprs) {
  KeyPlan kp;
  const Plan& plan = c.plan;
  const int nk = (int)key_exprs.size();
  std::vector<int> knodes(nk);
  struct Info { bool have_range = false; int64_t mn = 0, mx = 0; bool nullable = false; ColumnPtr col; };
  std::vector<Info> info(nk);
  bool all_packable = true;
  for (int i = 0; i < nk; i++) {
    const int e = key_exprs[i];
    KeyPart part; part.expr = e; part.dtype = infer_dtype(plan, e, *c.df);
    knodes[i] = c.lower(e);
    info[i].nullable = c.nodes[knodes[i]].nullable;
    part.nullable = info[i].nullable;
    const AE* x = &plan.ae[e];
    while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs];
    if (part.dtype == PLX_BOOL) { info[i].have_range = true; info[i].mn = 0; info[i].mx = 1; }
    else if (dtype_is_int(part.dtype) && x->kind == PLX_AE_COLUMN) {
      ColumnPtr col = c.df->cols[c.df->find(x->name)];
      bool cheap = dtype_width(part.dtype) <= 2 || nk > 1 || col->range_state != 0;
      // A single wide key whose range nobody has looked at yet: the range pass over it costs 8 B / row (1.5 ms per 1e9 rows) -- worth it exactly when the keys are
      // dense ids, which then take the direct-address tables at once instead of hash partitions on this run and direct ones on the next (1e9 rows over 1e6 dense ids:
      // first run 10.2 -> 7 ms; zipf-distributed ones, whose sample undercounts the groups and overflowed the hash tables once: 22.8 -> 9 ms).  A sample of 65536
      // rows says whether they LOOK dense (sparse 64-bit keys span far more than 2^26 in any sample: no pass for them).
      if (!cheap && part.dtype != PLX_U64 && col->values && col->len >= ((int64_t)1 << 24) && learn_dense_ranges()) {
        int64_t smn = 0, smx = 0;
        if (k::sample_minmax(col, &smn, &smx) && (unsigned __int128)((__int128)smx - (__int128)smn) < ((unsigned __int128)1 << 26)) {
          cheap = true;
          // (a guess is safe for ONE non-nullable key: every id below 2^bits decodes to the right key whatever the bounds were, anything beyond is reported by the
          //  tables; a nullable key's null code sits right above the assumed maximum, and the parts of a multi-column key overflow into each other: exact passes there)
          if (nk == 1 && !info[i].nullable && assume_range(col)) kp.note += "KeyRange{" + std::string(x->name) + ": sample looks dense -> bounds assumed from the sample, checked per row}; ";
          else kp.note += "KeyRange{" + std::string(x->name) + ": sample looks dense -> range pass}; ";
        }
      }
      // several key columns: whether they bit-pack is decided from their ranges -- guessed from the sample like a single key's (a sampled span beyond 2^62 settles it
      // without any pass: such a column packs with nothing)
      bool hopeless = false;
      if (cheap && nk > 1 && col->range_state == 0 && part.dtype != PLX_U64 && dtype_width(part.dtype) > 2 && col->values && col->len >= ((int64_t)1 << 24) && assume_ranges() && !col->no_assume) {
        int64_t smn = 0, smx = 0;
        if (k::sample_minmax(col, &smn, &smx, 1024) && (unsigned __int128)((__int128)smx - (__int128)smn) >= ((unsigned __int128)1 << 61)) hopeless = true;
        // (otherwise: the exact pass -- bounds that are only guessed must not pack several columns into one id: a value beyond its part's bits would spill into the next part)
      }
      if (part.dtype == PLX_U64 || hopeless) all_packable = false;
      else if (cheap) {
        // bounds that are only GUESSED (assume_range, possibly by an earlier query: as a single key under a filter -- only the passing rows were checked -- or as a narrowed
        // value column) may address ONE non-nullable key's table, where a wrong guess is reported; under a null code or packed next to other columns a value beyond them
        // would merge groups silently.  There: exact statistics, unless the guess has been verified over every row since.
        const bool may_assume = (nk == 1 && !info[i].nullable) || col->range_verified;
        if (col->values && ops::int_range(col, &info[i].mn, &info[i].mx, may_assume)) info[i].have_range = true;
        else if (col->range_state == 1) { info[i].have_range = true; info[i].mn = col->range_min; info[i].mx = col->range_max; }
        else if (col->range_state == 2) { info[i].have_range = true; info[i].mn = 0; info[i].mx = 0; }
        else all_packable = false;
      } else all_packable = false;
    } else all_packable = false;
    kp.parts.push_back(part);
  }
  int bits_total = 0;
  std::vector<int> bits(nk, 0);
  if (all_packable) {
    for (int i = 0; i < nk; i++) {
      unsigned __int128 range = (unsigned __int128)((__int128)info[i].mx - (__int128)info[i].mn) + 1 + (info[i].nullable ? 1 : 0);
      if (range > ((unsigned __int128)1 << 62)) { all_packable = false; break; }
      bits[i] = std::max(1, ceil_log2_u64((uint64_t)range));
      bits_total += bits[i];
    }
    if (bits_total > 62) all_packable = false;
  }
  if (all_packable) {
    kp.packed = true; kp.total_bits = bits_total;
    int acc = -1, shift = 0;
    for (int i = 0; i < nk; i++) {
      int n = knodes[i];
      if (info[i].mn != 0) n = c.mk(OP_SUB_I, n, c.konst((uint64_t)info[i].mn, 'i'), 'i');
      const uint64_t null_code = info[i].nullable ? (uint64_t)((__int128)info[i].mx - (__int128)info[i].mn + 1) : ~0ull;
      if (info[i].nullable) n = c.ifnull(n, null_code);
      if (shift) n = c.mk(OP_MUL_I, n, c.konst(1ull << shift, 'i'), 'i');
      acc = acc < 0 ? n : c.mk(OP_ADD_I, acc, n, 'i');
      KeyDecode& d = kp.parts[i].dec;
      d.shift = shift; d.mask = (1ull << bits[i]) - 1; d.min = info[i].mn; d.null_code = null_code; d.dtype = kp.parts[i].dtype;
      shift += bits[i];
    }
    c.key = acc;
    return kp;
  }
  if (nk != 1) {
    // wide key: the reference row-encodes (group_by/mod.rs:88-94); here each key column stays one
    // 64-bit word (floats canonicalised) and the hash sink compares the words.
    if (nk > kMaxKeys) throw Unsupported("more than " + std::to_string(kMaxKeys) + " group keys that do not bit-pack");
    kp.wide = true; kp.packed = false; kp.total_bits = 64 * nk;
    for (int i = 0; i < nk; i++) {
      int n = knodes[i];
      if (kp.parts[i].dtype == PLX_F64) n = c.mk(OP_CANON_F, n, n, 'f');
      kp.wide_nodes.push_back(n);
      kp.wide_nullable = kp.wide_nullable || c.nodes[n].nullable;
      KeyDecode& d = kp.parts[i].dec;
      d.shift = 0; d.mask = ~0ull; d.min = 0; d.null_code = ~0ull; d.dtype = kp.parts[i].dtype;
    }
    c.key = -1;
    c.wide_keys = kp.wide_nodes;
    return kp;
  }
  int n = knodes[0];
  if (kp.parts[0].dtype == PLX_F64) n = c.mk(OP_CANON_F, n, n, 'f');
  c.key = n;
  KeyDecode& d = kp.parts[0].dec;
  d.shift = 0; d.mask = ~0ull; d.min = 0; d.null_code = ~0ull; d.dtype = kp.parts[0].dtype;  // raw key: nullness comes from the valid flag
  kp.packed = false; kp.total_bits = 64;
  return kp;
}

// ------------------------------------------------------ fused pipeline driver -----
struct FusedAggResult {
  int64_t n_groups = 0;
  Buf acc;            // [n_groups][n_aggs] cells
  Buf packed_keys;    // [n_groups] u64 (group-by only)
  Buf key_valid;      // [n_groups] u8 flags (raw keys) or null
  Buf wide_words;     // wide keys: [n_keys][stride] u64
  Buf wide_valid;     // wide keys: [n_keys][stride] u8
  int64_t wide_stride = 0;
  int n_aggs = 0;
};

static uint64_t next_pow2(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }

// distinct-count estimate from a sample: solve d = G (1 - exp(-S / G)) for G
static double estimate_groups(double d, double S) {
  if (d >= S * 0.999) return 1e18;
  double lo = d, hi = 1e15;
  for (int it = 0; it < 200; it++) { double mid = std::sqrt(lo * hi); double f = mid * (1.0 - std::exp(-S / mid)); if (f < d) lo = mid; else hi = mid; }
  return hi;
}

// occupied slots of an aggregation table -> dense (packed key, valid flag, cells) arrays in `res`.
// Small tables are compacted in ONE pass into slot-count-sized outputs (no count pass, one sync).
static int64_t compact_into(const uint64_t* keys, const uint64_t* acc, int64_t n_slots, int64_t cap, int n_aggs, int occ_agg, FusedAggResult& res) {
  int64_t upper = n_slots;
  if (n_slots > (int64_t(1) << 16)) upper = k::table_compact(keys, acc, n_slots, cap, n_aggs, occ_agg, nullptr, nullptr, nullptr);
  const int64_t g1 = std::max<int64_t>(upper, 1);
  res.packed_keys = dev_alloc(sizeof(uint64_t) * (size_t)g1);
  res.key_valid = dev_alloc((size_t)g1);
  res.acc = dev_alloc(sizeof(uint64_t) * (size_t)g1 * n_aggs);
  const int64_t g = k::table_compact(keys, acc, n_slots, cap, n_aggs, occ_agg, res.packed_keys->as<uint64_t>(), res.key_valid->as<uint8_t>(), res.acc->as<uint64_t>());
  res.n_groups = g; res.n_aggs = n_aggs;
  return g;
}

static int64_t run_hash_agg(const Shape& sh, const Args& args, int static_id, int log2_cap, int len_idx, FusedAggResult& out, bool count_only) {
  const uint64_t cap = 1ull << log2_cap;
  const int64_t slots = (int64_t)cap + 2;
  Buf keys = dev_alloc(sizeof(uint64_t) * (size_t)slots);
  Buf acc = dev_alloc(sizeof(uint64_t) * (size_t)slots * sh.n_aggs);
  Buf ovf = dev_alloc_zero(8);
  k::fill_u64(keys->as<uint64_t>(), slots, kEmptyKey);
  k::init_agg_cells(acc->as<uint64_t>(), slots, sh);
  HashTable t; t.keys = keys->as<unsigned long long>(); t.acc = acc->as<unsigned long long>(); t.overflow = ovf->as<unsigned int>(); t.wave_combine = 0;
  t.log2_cap = (uint32_t)log2_cap; t.max_probe = (uint32_t)std::min<uint64_t>(cap, 1u << 10);
  k::fused_hash_agg(sh, args, t, static_id);
  uint32_t o = 0; d2h_sync(&o, ovf->ptr, 4);
  if (o) return -1;
  if (count_only) return k::table_compact(keys->as<uint64_t>(), acc->as<uint64_t>(), slots, (int64_t)cap, sh.n_aggs, -1, nullptr, nullptr, nullptr);
  (void)len_idx;
  return compact_into(keys->as<uint64_t>(), acc->as<uint64_t>(), slots, (int64_t)cap, sh.n_aggs, -1, out);
}

static Args offset_args(const Shape& sh, const Args& a, int64_t row0, int64_t rows);
// `sample_blocks` > 0 (the planner's distinct-count sample; count_only): only that many evenly spaced blocks of args.n_rows / ... rows are aggregated -- `sample_rows` in
// all -- instead of the whole input: a prefix says nothing about clustered or sorted data (it undercounts, and the partitioned path then runs into full LDS tables).
// Blocks stay below the JIT threshold, so the sample runs the interpreter: no run-time compilation for a pass over a million rows.
static int64_t run_wide_agg(const Shape& sh, const Args& args, int log2_cap, bool nullable, FusedAggResult& out, bool count_only, int sample_blocks = 0, int64_t sample_rows = 0) {
  const uint64_t cap = 1ull << log2_cap;
  const int nw = sh.n_keys + (nullable ? 1 : 0);
  Buf tags = dev_alloc(sizeof(uint64_t) * cap);
  Buf words = dev_alloc(sizeof(uint64_t) * cap * (size_t)nw);
  Buf acc = dev_alloc(sizeof(uint64_t) * cap * sh.n_aggs);
  Buf ovf = dev_alloc_zero(8);
  k::fill_u64(tags->as<uint64_t>(), (int64_t)cap, kEmptyKey);
  k::init_agg_cells(acc->as<uint64_t>(), (int64_t)cap, sh);
  WideTable t; t.tags = tags->as<unsigned long long>(); t.words = words->as<unsigned long long>(); t.acc = acc->as<unsigned long long>();
  t.overflow = ovf->as<unsigned int>(); t.log2_cap = (uint32_t)log2_cap; t.max_probe = (uint32_t)std::min<uint64_t>(cap, 1u << 10);
  t.n_words = (uint32_t)nw; t.has_null_word = nullable ? 1u : 0u;
  if (sample_blocks > 0) {
    const int64_t n = args.n_rows, per = (sample_rows / sample_blocks) & ~(int64_t)127, stride = (n / sample_blocks) & ~(int64_t)127;
    for (int b = 0; b < sample_blocks; b++) {
      const int64_t row0 = (int64_t)b * stride, rows = std::min<int64_t>(per, n - row0);
      if (rows > 0) k::fused_wide_agg(sh, offset_args(sh, args, row0, rows), t);
    }
  } else k::fused_wide_agg(sh, args, t);
  uint32_t o = 0; d2h_sync(&o, ovf->ptr, 4);
  if (o) return -1;
  int64_t g = k::wide_compact(t, sh.n_keys, sh.n_aggs, 0, nullptr, nullptr, nullptr);
  if (count_only) return g;
  out.n_groups = g; out.n_aggs = sh.n_aggs;
  const int64_t stride = std::max<int64_t>(g, 1);
  out.wide_stride = stride;
  out.wide_words = dev_alloc(sizeof(uint64_t) * (size_t)stride * sh.n_keys);
  out.wide_valid = dev_alloc((size_t)stride * sh.n_keys);
  out.acc = dev_alloc(sizeof(uint64_t) * (size_t)stride * sh.n_aggs);
  k::wide_compact(t, sh.n_keys, sh.n_aggs, stride, out.wide_words->as<uint64_t>(), out.wide_valid->as<uint8_t>(), out.acc->as<uint64_t>());
  return g;
}

// Sample of the input for the group-by planner: kSampleBlocks evenly spaced blocks of rows (a prefix alone says nothing about
// clustered data) aggregated into one HBM hash table.  -> number of distinct keys in the sample (-1: table overflow) and the
// heavy hitters (keys holding >= 1/1024 of the sampled rows), which the partitioned path pre-aggregates instead of scattering.
constexpr int kSampleBlocks = 8;
// rows the partitioned group-by samples per query (distinct-count estimate + heavy hitters): 2^20 strided rows cost ~0.15 ms at
// 1e9 rows; a key is "hot" from 1/1024 of the sample up, i.e. 1024 occurrences
constexpr int64_t kPartSampleRows = (int64_t)1 << 20;
constexpr int64_t kHotFraction = 4096;
static Args offset_args(const Shape& sh, const Args& a, int64_t row0, int64_t rows) {
  Args o = a;
  for (int i = 0; i < sh.n_inputs; i++) {
    if (o.in[i].values) o.in[i].values = (const uint8_t*)o.in[i].values + (sh.in_dtype[i] == PLX_BOOL ? (size_t)(row0 / 8) : (size_t)row0 * dtype_width(sh.in_dtype[i]));
    if (o.in[i].validity) o.in[i].validity = o.in[i].validity + row0 / 64;
  }
  o.n_rows = rows;
  return o;
}
// Returns the number of distinct keys in the sample (-1: table overflow).  *groups_est (if given): estimated number of groups of the
// whole input -- the heavy hitters are taken out of the sample first (estimate_groups assumes equally likely keys: one key holding
// half of the rows would halve the estimate and the LDS tables planned from it would run at twice their load).
static int64_t sample_keys(const Shape& full, const Args& args, int static_id, int full_len_idx, int64_t S, std::vector<uint64_t>* hot, double* groups_est) {
  // the sample only counts rows per key: the query's other aggregates are dropped (half the table atomics for a two-aggregate
  // query), which makes it a different program shape -- 2^20 rows in blocks below the JIT threshold run the interpreter, fast enough
  Shape sh = full;
  int len_idx = full_len_idx;
  if (full_len_idx >= 0 && full.n_aggs > 1) { sh.n_aggs = 1; sh.aggs[0] = full.aggs[full_len_idx]; len_idx = 0; static_id = -1; }
  const int log2_cap = ceil_log2_u64((uint64_t)S) + 1;
  const uint64_t cap = 1ull << log2_cap;
  const int64_t slots = (int64_t)cap + 2;
  Buf keys = dev_alloc(sizeof(uint64_t) * (size_t)slots), acc = dev_alloc(sizeof(uint64_t) * (size_t)slots * sh.n_aggs), ovf = dev_alloc_zero(8);
  k::fill_u64(keys->as<uint64_t>(), slots, kEmptyKey);
  k::init_agg_cells(acc->as<uint64_t>(), slots, sh);
  HashTable t; t.keys = keys->as<unsigned long long>(); t.acc = acc->as<unsigned long long>(); t.overflow = ovf->as<unsigned int>(); t.wave_combine = 1;
  t.log2_cap = (uint32_t)log2_cap; t.max_probe = 1u << 14;
  const int64_t n = args.n_rows, per = (S / kSampleBlocks) & ~(int64_t)127, stride = (n / kSampleBlocks) & ~(int64_t)127;
  for (int b = 0; b < kSampleBlocks; b++) {
    const int64_t row0 = (int64_t)b * stride;
    const int64_t rows = std::min<int64_t>(per, n - row0);
    if (rows > 0) k::fused_hash_agg(sh, offset_args(sh, args, row0, rows), t, static_id);
  }
  uint32_t o = 0; d2h_sync(&o, ovf->ptr, 4);
  if (o) return -1;
  uint64_t hot_rows = 0;
  // a key is "hot" from 1/4096 of the sampled rows (256 of 2^20: a stable count) -- at 1e9 rows ~2.4e5 rows that would otherwise serialise on one LDS address
  // in one aggregation workgroup (~0.5 ms); the kP2MaxHot heaviest are kept
  if (hot) k::select_hot_keys(t, sh.n_aggs, len_idx, (uint64_t)std::max<int64_t>(64, (per * kSampleBlocks) / kHotFraction), hot, &hot_rows);
  const int64_t d = k::table_compact(keys->as<uint64_t>(), acc->as<uint64_t>(), slots, (int64_t)cap, sh.n_aggs, -1, nullptr, nullptr, nullptr);
  if (groups_est) {
    const double n_hot = hot ? (double)hot->size() : 0.0, s_rest = std::max(1.0, (double)(per * kSampleBlocks) - (double)hot_rows), d_rest = std::max(0.0, (double)d - n_hot);
    *groups_est = d < 0 ? 1e18 : (d_rest > 0 ? estimate_groups(d_rest, s_rest) : 0.0) + n_hot;
  }
  return d;
}
// PLX_PROBE_PARTITIONED: 0 = never, 2 = whenever the kernels are available (tests: small inputs, any key order), default = by size / density / key order
// The partitioned (LDS-filled) build of a join hash table: PLX_JOIN_PART_BUILD = 0 never | 1 (default) build sides of >= 2^24 rows | 2 whenever the geometry allows.
// Windows of 2^13 slots (96 KB of LDS: 8-byte keys + 4-byte rows), no larger than a region of the partitioned probe (256 regions: kernels_partition.hip probe_pass_kernel).
static const uint32_t kJoinWindowLog2 = [] { const char* e = getenv("PLX_JOIN_WINDOW_LOG2"); const int v = e ? atoi(e) : 13; return (uint32_t)(v >= 10 && v <= 13 ? v : 13); }();      // (PLX_JOIN_WINDOW_LOG2: measurement)
static bool partitioned_build_wanted(int64_t build_rows, int log2_cap) {
  const char* e = getenv("PLX_JOIN_PART_BUILD");      // (read at every build: the tests switch it)
  const int mode = e ? atoi(e) : 1;
  if (mode <= 0 || log2_cap < (int)kJoinWindowLog2 + 8) return false;
  return mode >= 2 || build_rows >= ((int64_t)1 << 24);
}
static int partitioned_probe_mode() { const char* e = getenv("PLX_PROBE_PARTITIONED"); return e && e[0] == '0' ? 0 : e && e[0] == '2' ? 2 : 1; }
static bool probe_late_loads() { static const bool v = [] { const char* e = getenv("PLX_PROBE_LATE"); return !(e && e[0] == '0'); }(); return v; }
static int part_version() { static const int v = [] { const char* e = getenv("PLX_PART_V"); return (e && e[0] == '1') ? 1 : 2; }(); return v; }
static bool hot_keys_enabled() { static const bool v = [] { const char* e = getenv("PLX_PART_HOT"); return !(e && e[0] == '0'); }(); return v; }

// Value ranges of the record sources that are plain integer columns (cached column statistics, one reduction pass the first time a
// column is asked): the partitioned group-by packs its records with them (fused::kPackNarrow / kPackFused).
static void source_ranges(const Compiler& c, k::SrcRange out[kMaxSrc]) {
  const Shape& sh = c.shape;
  const RecLayout2 L = rec_layout2(sh, kP2Hash, kPackNarrow);
  for (uint32_t j = 0; j < L.n_src && j < (uint32_t)kMaxSrc; j++) {
    out[j] = k::SrcRange{};
    const int in = slot_input(sh, L.src_slot[j]);
    if (in < 0 || in >= (int)c.input_cols.size()) continue;
    const ColumnPtr& col = c.cols[c.input_cols[in]];
    if (!dtype_is_int(col->dtype) || col->dtype == PLX_U64) continue;
    // make_record2 stores (v - base) in 32 bits WITHOUT a per-row range check: only ranges the library computed itself may narrow a value
    // (bounds declared by the caller -- plx_column_set_bounds, IPC dictionary sizes -- are checked per row where they address tables, never trusted here)
    // -- except the planner's own guesses (assume_range): those narrow WITH a per-row check (PartPlan2::check_src) and a second run if it fails
    if (col->range_state == 0) assume_range(col);
    if (col->range_state != 0 && !col->range_trusted && !col->range_assumed) continue;
    int64_t mn = 0, mx = 0;
    if (ops::int_range(col, &mn, &mx, true)) { out[j].known = true; out[j].mn = mn; out[j].mx = mx; out[j].check = col->range_assumed && !col->range_verified; }
  }
}
// does any input column of the compiled query carry bounds nobody measured (declared by the caller, or assumed by the planner)?  Then the table sinks report ids outside
// their tables (LdsAggSink / DenseAggSink `oob`) and the host looks at the flag.
static bool untrusted_key_bounds(const Compiler& c) {
  for (auto& col : c.cols) if (col->range_state == 1 && !col->range_trusted) return true;
  return false;
}
static void check_oob_flag(const Buf& oob) {
  if (!oob) return;
  uint32_t f = 0;
  d2h_sync(&f, oob->ptr, 4);
  if (f) fail(PLX_ERR_INVALID, "group key outside the bounds declared or assumed for its column");
}
// A partitioned run WITHOUT a predicate put every row of its narrowed value columns through the per-row check (PartPlan2::check_src) and raised no flag: their assumed
// bounds are now verified -- valid for every row, if not tight -- and later runs take the kernels without the check.
static void mark_sources_verified(const Compiler& c) {
  const Shape& sh = c.shape;
  if (sh.pred != kNone) return;
  const RecLayout2 L = rec_layout2(sh, kP2Hash, kPackNarrow);
  for (uint32_t j = 0; j < L.n_src && j < (uint32_t)kMaxSrc; j++) {
    const int in = slot_input(sh, L.src_slot[j]);
    if (in < 0 || in >= (int)c.input_cols.size()) continue;
    const ColumnPtr& col = c.cols[c.input_cols[in]];
    if (col->range_assumed) col->range_verified = true;
  }
}

// The planner's sample of a key column (see Column::key_sample).
// `sig`: the program that produced the sampled key VALUES (ops + immediates): the same column enters as raw values on its first group-by and -- once its range has
// been learned -- as packed ids (key - min) on the next; hot keys of one encoding never match rows of the other (a stale list is harmless for the result
// -- the rows simply take the scatter -- but one key holding half of the rows then costs 28 ms of same-address LDS atomics instead of 0.1 ms)
struct KeySample { int64_t n_rows = 0, distinct = -1; double groups_est = 1e18; std::vector<uint64_t> hot; bool with_hot = false; uint64_t sig = 0; };
static uint64_t key_program_signature(const Shape& sh, const Args& args) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } };
  mix(&sh, sizeof(Shape));
  mix(args.imm, sizeof(args.imm));
  return h;
}
// the frame column a single-column group key reads directly, when the query has no predicate (then the sample depends on nothing else)
static ColumnPtr plain_key_column(const Compiler& c, const KeyPlan& kp) {
  if (c.shape.pred != kNone || kp.parts.size() != 1) return nullptr;
  const AE* x = &c.plan.ae[kp.parts[0].expr];
  while (x->kind == PLX_AE_ALIAS) x = &c.plan.ae[x->lhs];
  if (x->kind != PLX_AE_COLUMN) return nullptr;
  const int ci = c.df->find(x->name);
  return ci >= 0 && c.df->cols[ci]->len == c.args.n_rows ? c.df->cols[ci] : nullptr;
}
static bool sample_cache_enabled() { static const bool v = [] { const char* e = getenv("PLX_SAMPLE_CACHE"); return !(e && e[0] == '0'); }(); return v; }
static int64_t sample_keys(const Shape& full, const Args& args, int static_id, int full_len_idx, int64_t S, std::vector<uint64_t>* hot, double* groups_est);
static int64_t sample_keys_cached(const ColumnPtr& key_col, const Shape& sh, const Args& args, int static_id, int len_idx, int64_t S, std::vector<uint64_t>* hot, double* groups_est,
                                  std::string& desc) {
  // (atomic_load / atomic_store: two host threads may plan group-bys over the same column at once -- each on its own stream, core.cpp)
  const std::shared_ptr<void> held = key_col && sample_cache_enabled() ? std::atomic_load(&key_col->key_sample) : nullptr;
  if (held) {
    const KeySample& ks = *std::static_pointer_cast<KeySample>(held);
    if (ks.n_rows == args.n_rows && (ks.with_hot || !hot) && ks.sig == key_program_signature(sh, args)) {
      if (hot) *hot = ks.hot;
      if (groups_est) *groups_est = ks.groups_est;
      desc += "cached_";
      return ks.distinct;
    }
  }
  double g = 1e18;
  const int64_t d = sample_keys(sh, args, static_id, len_idx, S, hot, &g);
  if (groups_est) *groups_est = g;
  if (key_col && sample_cache_enabled() && d >= 0) {
    auto ks = std::make_shared<KeySample>();
    ks->n_rows = args.n_rows; ks->distinct = d; ks->groups_est = g; ks->with_hot = hot != nullptr; ks->sig = key_program_signature(sh, args);
    if (hot) ks->hot = *hot;
    std::atomic_store(&key_col->key_sample, std::shared_ptr<void>(ks));
  }
  return d;
}

static void run_fused_groupby(Compiler& c, const KeyPlan& kp, int len_idx, FusedAggResult& res, std::string& desc) {
  const Shape& sh = c.shape;
  const Args& args = c.args;
  const int static_id = find_static_shape(sh);
  const int64_t n = args.n_rows;
  res.n_aggs = sh.n_aggs;
  if (n == 0) { res.n_groups = 0; res.acc = dev_alloc(8); res.packed_keys = dev_alloc(8); return; }
  if (kp.packed && kp.total_bits <= 12 && k::lds_agg_copies(1 << kp.total_bits, sh.n_aggs) > 0) {
    const int G = 1 << kp.total_bits;
    Buf cells = dev_alloc(sizeof(uint64_t) * (size_t)G * sh.n_aggs);
    Buf oob = untrusted_key_bounds(c) ? dev_alloc_zero(8) : nullptr;      // key bounds nobody measured: the sink reports a group id outside the table
    k::fused_lds_agg(sh, args, G, static_id, cells->as<uint64_t>(), oob ? oob->as<unsigned int>() : nullptr);
    desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+lds_table(G=" + std::to_string(G) + ",copies=" + std::to_string(k::lds_agg_copies(G, sh.n_aggs)) + ")";
    compact_into(nullptr, cells->as<uint64_t>(), G, -1, sh.n_aggs, len_idx, res);
    check_oob_flag(oob);
    res.key_valid = nullptr;  // packed keys carry their own null codes
    return;
  }
  // large inputs over many (packed) group ids: per-row atomics on the dense HBM table are bound by the device atomic
  // rate just like the hash table -> partition + LDS aggregation on the packed id (kernels_partition.hip)
  if (kp.packed && kp.total_bits > 12 && !(c.plan.flags & PLX_PLAN_NO_PARTITION) && n >= ((int64_t)1 << 24)) {
    const double est = std::min((double)((uint64_t)1 << std::min(kp.total_bits, 40)), (double)n);
    if (part_version() == 2) {
      // second generation: direct-address LDS tables over the dense id when they fit (hash tables otherwise); a strided sample
      // finds the heavy hitters, which are aggregated in the scatter pass instead of being scattered
      std::vector<uint64_t> hot;
      double est2 = est;
      if (hot_keys_enabled() || kp.total_bits > 25) {
        const int64_t S = kPartSampleRows;
        double g_est = 1e18;
        const int64_t d = sample_keys_cached(plain_key_column(c, kp), sh, args, static_id, len_idx, S, hot_keys_enabled() ? &hot : nullptr, &g_est, desc);
        if (d >= 0) est2 = std::min(est, std::min(g_est, (double)n) * 1.3);
        desc += "sample(distinct=" + std::to_string(d) + ",hot=" + std::to_string(hot.size()) + ")+";
      }
      if (c.plan.group_hint > 0) { est2 = std::min(est, c.plan.group_hint * 1.02 + 64.0); desc += "groups<=" + std::to_string((int64_t)c.plan.group_hint) + "(plan)+"; }
      PartPlan2 p2;
      k::SrcRange ranges[kMaxSrc];
      source_ranges(c, ranges);
      // (too many id bits for direct-address tables -> hash partitions of the packed id: its range is [0, 2^total_bits) by construction -- when the bounds behind the
      // packing were measured, not guessed -- so ids of < 48 bits travel as 48-bit offsets, two rows a record: fused::kPackPair)
      k::SrcRange id_rng;
      if (kp.total_bits < 48 && !untrusted_key_bounds(c)) { id_rng.known = true; id_rng.mn = 0; id_rng.mx = (int64_t)(((uint64_t)1 << kp.total_bits) - 1); }
      if (k::partition_plan2(sh, est2, kp.total_bits, len_idx, n, (int)hot.size(), &p2, ranges, &id_rng)) {
        std::string pd;
        Buf ok, okv, oacc;
        const int64_t g = k::partitioned_agg2(sh, args, p2, static_id, hot, &ok, &okv, &oacc, &pd);
        if (g >= 0) {
          if (p2.check_src) mark_sources_verified(c);
          res.n_groups = g; res.n_aggs = sh.n_aggs; res.packed_keys = ok; res.key_valid = nullptr; res.acc = oacc;   // packed ids carry their own null codes
          desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+" + pd;
          return;
        }
        desc += g == -2 ? "v2-unavailable+" : "lds-overflow+";
      }
    }
    PartitionPlan pp;
    if (part_version() == 1 && k::partition_plan(sh, est, false, &pp)) {
      std::string pd;
      Buf ok, okv, oacc;
      const int64_t g = k::partitioned_agg(sh, args, pp, static_id, &ok, &okv, &oacc, &pd);
      if (g >= 0) {
        res.n_groups = g; res.n_aggs = sh.n_aggs; res.packed_keys = ok; res.key_valid = nullptr; res.acc = oacc;   // packed ids carry their own null codes
        desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+" + pd;
        return;
      }
      desc += "lds-overflow+";
    }
  }
  if (kp.packed && kp.total_bits <= 28 && ((size_t)sh.n_aggs << (kp.total_bits + 3)) <= (size_t(8) << 30)) {
    const int64_t G = (int64_t)1 << kp.total_bits;
    Buf cells = dev_alloc(sizeof(uint64_t) * (size_t)(G + 1) * sh.n_aggs);
    k::init_agg_cells(cells->as<uint64_t>(), G + 1, sh);
    Buf oob = untrusted_key_bounds(c) ? dev_alloc_zero(8) : nullptr;
    DenseTable t; t.acc = cells->as<unsigned long long>(); t.key_min = 0; t.n_groups = G; t.oob = oob ? oob->as<unsigned int>() : nullptr;
    k::fused_dense_agg(sh, args, t, static_id);
    desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+dense_hbm_table(G=" + std::to_string(G) + ")";
    compact_into(nullptr, cells->as<uint64_t>(), G, -1, sh.n_aggs, len_idx, res);
    check_oob_flag(oob);
    res.key_valid = nullptr;
    return;
  }
  // hash table: size from a sampled distinct-count estimate, grow x4 on overflow
  int log2_cap;
  const int64_t S = (int64_t)1 << 22;
  if (n <= 2 * S) log2_cap = std::max(10, ceil_log2_u64((uint64_t)n * 2));
  else {
    Args sa = args; sa.n_rows = S;
    FusedAggResult tmp;
    std::vector<uint64_t> hot;
    const bool may_partition = !(c.plan.flags & PLX_PLAN_NO_PARTITION) && n >= ((int64_t)1 << 24);
    const bool strided = may_partition && !kp.wide && part_version() == 2;
    const int64_t Sd = strided || kp.wide ? kPartSampleRows : S;
    double g_est = -1.0;
    int64_t d = kp.wide ? run_wide_agg(sh, args, 21, kp.wide_nullable, tmp, true, kSampleBlocks, Sd)
                        : (strided ? sample_keys_cached(plain_key_column(c, kp), sh, args, static_id, len_idx, Sd, hot_keys_enabled() ? &hot : nullptr, &g_est, desc)
                                   : run_hash_agg(sh, sa, static_id, 23, len_idx, tmp, true));
    double G = d < 0 ? 1e18 : (g_est >= 0.0 ? g_est : estimate_groups((double)d, (double)Sd));
    G = std::min(G, (double)n);
    const bool hinted = c.plan.group_hint > 0;
    if (hinted) { G = std::min(c.plan.group_hint, (double)n); desc += "groups<=" + std::to_string((int64_t)c.plan.group_hint) + "(plan)+"; }
    log2_cap = std::max(12, ceil_log2_u64((uint64_t)(G * 2.0) + 1));
    desc += "sample(distinct=" + std::to_string(d) + "/" + std::to_string(Sd) + ")+";
    // many rows, many groups: per-row global atomics are bound by the ~24 G/s device atomic rate; partition the
    // rows and aggregate each partition in LDS instead (kernels_partition.hip)
    // a wide (multi-column, unpackable) key takes the same partitioned path: records carry one word per key column, partitions come from the hash of the
    // words + null mask, the LDS tables compare word by word (the reference row-encodes such keys: crates/polars-row, hash_keys.rs:334 RowEncodedKeys)
    if (kp.wide && may_partition && part_version() == 2 && G >= 4096.0) {
      k::SrcRange ranges[kMaxSrc];
      source_ranges(c, ranges);
      double plan_for = hinted ? G * 1.02 + 64.0 : G * 1.3;
      for (int attempt = 0; attempt < 3; attempt++) {
        PartPlan2 p2;
        if (!k::partition_plan2(sh, plan_for, -1, len_idx, n, 0, &p2, ranges)) break;
        std::string pd;
        Buf ok, okv, oacc;
        int64_t stride = 0;
        const int64_t g = k::partitioned_agg2(sh, args, p2, static_id, {}, &ok, &okv, &oacc, &pd, nullptr, &stride);
        if (g >= 0) {
          if (p2.check_src) mark_sources_verified(c);
          res.n_groups = g; res.n_aggs = sh.n_aggs; res.wide_words = ok; res.wide_valid = okv; res.wide_stride = stride; res.acc = oacc;
          desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+" + pd;
          return;
        }
        if (g == -2) { desc += "v2-unavailable+"; break; }
        desc += "lds-overflow(P=" + std::to_string(1u << p2.log2_parts) + ")+";
        if (p2.log2_parts >= 9) break;
        // a table filled up: the sample undercounted (clustered keys).  One more attempt at the LARGEST plan (512 partitions); if that overflows too the HBM table takes over --
        // never a ladder of full scatter + aggregate passes over all rows
        plan_for = std::max(plan_for * 2.0, (double)((uint64_t)p2.n_slots << 9) * 0.8);
      }
    }
    if (!kp.wide && !(c.plan.flags & PLX_PLAN_NO_PARTITION) && G >= 4096.0 && n >= ((int64_t)1 << 24)) {
      bool any_null = c.key >= 0 && c.nodes[c.key].nullable;
      for (auto& a : c.aggs) if (a.second >= 0 && c.nodes[a.second].nullable) any_null = true;
      if (part_version() == 2) {
        k::SrcRange ranges[kMaxSrc];
        source_ranges(c, ranges);
        // A raw signed-integer key column scanned without a predicate: the scatter pass also records the exact key range, which is
        // cached on the column like any other statistic -- the NEXT group-by / join on it can plan dense (direct-address) tables.
        ColumnPtr stat_col;
        if (sh.pred == kNone && kp.parts.size() == 1 && dtype_is_int(kp.parts[0].dtype) && kp.parts[0].dtype != PLX_U64) {
          const AE* x = &c.plan.ae[kp.parts[0].expr];
          while (x->kind == PLX_AE_ALIAS) x = &c.plan.ae[x->lhs];
          if (x->kind == PLX_AE_COLUMN) { const int ci = c.df->find(x->name); if (ci >= 0 && c.df->cols[ci]->range_state == 0 && c.df->cols[ci]->len == n) stat_col = c.df->cols[ci]; }
        }
        // ... and once the exact range of such a key is known (learned that way, or computed): 64-bit keys spanning < 2^48 travel as 48-bit offsets, two rows a record
        // (fused::kPackPair in hash mode; only ranges the library computed itself: a declared or assumed range is never used unchecked)
        k::SrcRange key_rng;
        if (kp.parts.size() == 1 && dtype_is_int(kp.parts[0].dtype) && kp.parts[0].dtype != PLX_U64) {
          const AE* x = &c.plan.ae[kp.parts[0].expr];
          while (x->kind == PLX_AE_ALIAS) x = &c.plan.ae[x->lhs];
          if (x->kind == PLX_AE_COLUMN) {
            const int ci = c.df->find(x->name);
            if (ci >= 0 && c.df->cols[ci]->range_state == 1 && c.df->cols[ci]->range_trusted) { key_rng.known = true; key_rng.mn = c.df->cols[ci]->range_min; key_rng.mx = c.df->cols[ci]->range_max; }
          }
        }
        // How many groups to plan for.  The estimate assumes equally likely keys; heavy hitters in the sample mean a heavy TAIL too, and a tail the sample
        // undercounts badly (zipf 1.1 over 1e6 keys: 1.5e5 distinct keys in 2^20 sampled rows, 1e6 in 1e9 rows): with skew the tables are planned for 4 x the
        // estimate.  A table that fills up anyway is reported by the aggregation pass; the plan is then doubled (more partitions) and the pass repeated -- never
        // the per-row HBM-table path, which a skewed input turns into seconds of same-address atomics.
        double plan_for = hinted ? G * 1.02 + 64.0 : (!hot.empty() && G < 1e17) ? G * 4.0 : G * 1.3;      // (a bound from the plan is not an estimate: no safety factor)
        for (int attempt = 0; attempt < 3; attempt++) {
          PartPlan2 p2;
          if (!k::partition_plan2(sh, plan_for, -1, len_idx, n, (int)hot.size(), &p2, ranges, &key_rng)) {
            if (attempt == 0 && !hot.empty() && k::partition_plan2(sh, G * 1.3, -1, len_idx, n, (int)hot.size(), &p2, ranges, &key_rng)) { /* 4 x does not fit 512 partitions: the plain estimate does */ }
            else break;
          }
          std::string pd;
          Buf ok, okv, oacc;
          int64_t key_range[2] = {1, 0};
          const int64_t g = k::partitioned_agg2(sh, args, p2, static_id, hot, &ok, &okv, &oacc, &pd, stat_col ? key_range : nullptr);
          if (g >= 0) {
            if (stat_col && key_range[0] <= key_range[1]) { stat_col->range_state = 1; stat_col->range_min = key_range[0]; stat_col->range_max = key_range[1]; stat_col->range_trusted = true; pd += "+key_range_learned"; }
            if (p2.check_src) mark_sources_verified(c);
            res.n_groups = g; res.n_aggs = sh.n_aggs; res.packed_keys = ok; res.key_valid = okv; res.acc = oacc;
            desc += "hot=" + std::to_string(hot.size()) + "+" + std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+" + pd;
            return;
          }
          if (g == -2) { desc += "v2-unavailable+"; break; }
          desc += "lds-overflow(P=" + std::to_string(1u << p2.log2_parts) + ")+";
          if (p2.log2_parts >= 9) break;
          plan_for = std::max(plan_for * 2.0, (double)((uint64_t)p2.n_slots << p2.log2_parts) * 1.01);      // beyond what this plan's tables hold at all
        }
      }
      if (part_version() == 1) {
      PartitionPlan pp;
      if (k::partition_plan(sh, G * 1.3, any_null, &pp)) {
        std::string pd;
        Buf ok, okv, oacc;
        const int64_t g = k::partitioned_agg(sh, args, pp, static_id, &ok, &okv, &oacc, &pd);
        if (g >= 0) {
          res.n_groups = g; res.n_aggs = sh.n_aggs; res.packed_keys = ok; res.key_valid = okv; res.acc = oacc;
          desc += std::string("fused_scan[") + jit::program_mode(static_id, args.n_rows) + "]+" + pd;
          return;
        }
        desc += "lds-overflow+";
      }
      }   // first-generation kernels (PLX_PART_V=1 only)
    }
  }
  for (int attempt = 0; attempt < 8; attempt++) {
    int64_t g = kp.wide ? run_wide_agg(sh, args, log2_cap, kp.wide_nullable, res, false) : run_hash_agg(sh, args, static_id, log2_cap, len_idx, res, false);
    if (g >= 0) {
      desc += std::string("fused_scan[") + jit::program_mode(kp.wide ? -1 : static_id, args.n_rows) + "]+" + (kp.wide ? "wide_hash_hbm_table(words=" + std::to_string(sh.n_keys + (kp.wide_nullable ? 1 : 0)) + ",cap=2^" : "hash_hbm_table(cap=2^") + std::to_string(log2_cap) + ")";
      return;
    }
    log2_cap += 2;
    desc += "grow+";
    PLX_REQUIRE(log2_cap <= 34, PLX_ERR_OOM, "group-by hash table would exceed 2^34 slots");
  }
  fail(PLX_ERR_OOM, "group-by hash table kept overflowing");
}

// Output columns are allocated here and filled by ONE finalize_batch launch per query
// (was one launch per key and per aggregate: ~10 launches + 2 popcount syncs for TPC-H Q1).
static ColumnPtr finalize_column(const FusedAggResult& r, const FinalSpec& fs, FinBatch& batch) {
  const int64_t G = r.n_groups;
  auto out = std::make_shared<Column>();
  out->dtype = fs.out_dtype; out->len = G;
  out->values = dev_alloc(values_bytes(fs.out_dtype, G));
  const bool nullable = fs.kind == FIN_MEAN || fs.kind == FIN_MINMAX_I || fs.kind == FIN_MINMAX_F;
  if (nullable) out->validity = dev_alloc_zero(bitmap_bytes(G)); else out->null_count = 0;
  FinalSpec f = fs;
  if (fs.kind == FIN_COPY64 && dtype_width(fs.out_dtype) < 4) f.kind = FIN_NARROW;
  PLX_REQUIRE(batch.n < kMaxFinJobs, PLX_ERR_UNSUPPORTED, "too many output columns in one fused aggregation");
  FinJob& j = batch.jobs[batch.n++];
  j = FinJob{};
  j.is_key = 0; j.fs = f; j.out = out->values->ptr; j.out_valid = nullable ? out->validity->as<uint64_t>() : nullptr;
  return out;
}

static ColumnPtr decode_key_column(const FusedAggResult& r, const KeyPart& part, int key_index, FinBatch& batch) {
  const int64_t G = r.n_groups;
  auto out = std::make_shared<Column>();
  out->dtype = part.dtype; out->len = G;
  out->values = part.dtype == PLX_BOOL ? dev_alloc_zero(bitmap_bytes(G)) : dev_alloc(values_bytes(part.dtype, G));
  if (part.nullable) out->validity = dev_alloc_zero(bitmap_bytes(G)); else out->null_count = 0;   // a null key forms its own group
  KeyDecode kd = part.dec;
  if (part.dtype == PLX_F32) kd.dtype = PLX_F32;
  PLX_REQUIRE(batch.n < kMaxFinJobs, PLX_ERR_UNSUPPORTED, "too many output columns in one fused aggregation");
  FinJob& j = batch.jobs[batch.n++];
  j = FinJob{};
  j.is_key = 1; j.kd = kd; j.out = out->values->ptr; j.out_valid = part.nullable ? out->validity->as<uint64_t>() : nullptr;
  if (r.wide_words) {
    j.packed = r.wide_words->as<unsigned long long>() + (size_t)key_index * r.wide_stride;
    j.kvalid = r.wide_valid->as<unsigned char>() + (size_t)key_index * r.wide_stride;
  } else {
    j.packed = r.packed_keys->as<unsigned long long>();
    j.kvalid = r.key_valid ? r.key_valid->as<unsigned char>() : nullptr;
  }
  return out;
}

// order groups by first occurrence (maintain_order): host argsort of the FIRST_ROW cells
static ColumnPtr first_row_permutation(const FusedAggResult& r, int first_idx) {
  const int64_t G = r.n_groups;
  std::vector<uint64_t> cells((size_t)G * r.n_aggs);
  d2h_sync(cells.data(), r.acc->ptr, cells.size() * 8);
  std::vector<uint32_t> perm((size_t)G);
  std::iota(perm.begin(), perm.end(), 0u);
  std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return cells[(size_t)a * r.n_aggs + first_idx] < cells[(size_t)b * r.n_aggs + first_idx]; });
  return column_from_host(PLX_U32, perm.data(), nullptr, 0, G);
}

// ---- machine-readable dump of a compiled pipeline (dump_program_json) ----
struct ProgramDump {
  std::string json;
};
static thread_local ProgramDump* t_program_dump = nullptr;   // set only for the duration of a dump_program_json call
static std::string jstr(const std::string& x) { std::string o = "\""; for (char ch : x) { if (ch == '"' || ch == '\\') o += '\\'; o += ch; } return o + "\""; }
// the register program of one compiled scan as JSON fields (no braces); input names are looked up in `frame` by buffer identity
static std::string program_fields(const Compiler& c, const Frame& frame) {
  std::ostringstream o;
  const Shape& sh = c.shape;
  o << "\"n_rows\":" << c.args.n_rows << ",\"inputs\":[";
  auto name_of = [&](int col_id) -> std::string {   // input_cols holds the compiler's own column ids: map the buffer back to its frame column
    const ColumnPtr& buf = c.cols[col_id];
    for (size_t j = 0; j < frame.cols.size(); j++) if (frame.cols[j] == buf) return frame.names[j];
    return "?";
  };
  for (int i = 0; i < sh.n_inputs; i++)
    o << (i ? "," : "") << "{\"name\":" << jstr(name_of(c.input_cols[i])) << ",\"dtype\":" << (int)sh.in_dtype[i] << ",\"nullable\":" << (int)sh.in_nullable[i] << "}";
  o << "],\"ops\":[";
  for (int i = 0; i < sh.n_ops; i++)
    o << (i ? "," : "") << "[" << (int)sh.ops[i].code << "," << (int)sh.ops[i].dst << "," << (int)sh.ops[i].a << "," << (int)sh.ops[i].b << "," << (int)sh.ops[i].c << ",\"" << c.args.imm[i] << "\"]";
  o << "],\"pred\":" << (int)sh.pred << ",\"key\":" << (int)sh.key << ",\"keys\":[";
  for (int i = 0; i < sh.n_keys; i++) o << (i ? "," : "") << (int)sh.keys[i];
  o << "],\"aggs\":[";
  for (int i = 0; i < sh.n_aggs; i++) o << (i ? "," : "") << "[" << (int)sh.aggs[i].kind << "," << (int)sh.aggs[i].src << "]";
  o << "]";
  // ops the predicate / key depend on (a probe scan runs the others only for the lanes that find a build row)
  const ProgramSplit split = split_program(sh);
  o << ",\"early_mask\":" << split.early << ",\"any_late\":" << (split.any_late ? 1 : 0);
  return o.str();
}
static std::string finals_outputs_fields(const Plan& plan, const std::vector<int>& agg_nodes, const std::vector<FinalSpec>& specs, const std::vector<int>& out_exprs) {
  std::ostringstream o;
  o << "\"finals\":[";
  for (size_t i = 0; i < specs.size(); i++)
    o << (i ? "," : "") << "{\"kind\":" << (int)specs[i].kind << ",\"a\":" << (int)specs[i].a << ",\"b\":" << (int)specs[i].b << ",\"c\":" << (int)specs[i].c << ",\"out_dtype\":" << (int)specs[i].out_dtype << "}";
  o << "],\"outputs\":[";
  for (size_t i = 0; i < out_exprs.size(); i++) {
    int e = out_exprs[i];
    while (plan.ae[e].kind == PLX_AE_ALIAS) e = plan.ae[e].lhs;
    int idx = -1;                                   // index into finals when the output IS one aggregate (else: a row expression over aggregates)
    for (size_t j = 0; j < agg_nodes.size(); j++) if (agg_nodes[j] == e) idx = (int)j;
    o << (i ? "," : "") << "{\"name\":" << jstr(output_name(plan, out_exprs[i])) << ",\"final\":" << idx << "}";
  }
  o << "]";
  return o.str();
}
static std::string compiled_json(const char* kind, const Plan& plan, const Compiler& c, const KeyPlan* kp, const std::vector<int>& agg_nodes,
                                 const std::vector<FinalSpec>& specs, const std::vector<int>& out_exprs, int len_idx, int first_idx, bool maintain_order) {
  std::ostringstream o;
  o << "{\"kind\":\"" << kind << "\"," << program_fields(c, *c.df);
  o << ",\"len_idx\":" << len_idx << ",\"first_idx\":" << first_idx << ",\"maintain_order\":" << (maintain_order ? 1 : 0);
  if (kp) {
    o << ",\"key_plan\":{\"packed\":" << (kp->packed ? 1 : 0) << ",\"wide\":" << (kp->wide ? 1 : 0) << ",\"parts\":[";
    for (size_t i = 0; i < kp->parts.size(); i++) {
      const KeyPart& kpart = kp->parts[i];
      o << (i ? "," : "") << "{\"name\":" << jstr(output_name(plan, kpart.expr)) << ",\"dtype\":" << kpart.dtype << ",\"nullable\":" << (kpart.nullable ? 1 : 0) << ",\"shift\":" << kpart.dec.shift
        << ",\"mask\":\"" << kpart.dec.mask << "\",\"min\":\"" << kpart.dec.min << "\",\"null_code\":\"" << kpart.dec.null_code << "\"}";
    }
    o << "]}";
  }
  o << "," << finals_outputs_fields(plan, agg_nodes, specs, out_exprs) << "}";
  return o.str();
}
static void dump_compiled(ProgramDump* dump, const char* kind, const Plan& plan, const Compiler& c, const KeyPlan* kp, const std::vector<int>& agg_nodes,
                          const std::vector<FinalSpec>& specs, const std::vector<int>& out_exprs, int len_idx, int first_idx, bool maintain_order) {
  if (dump) dump->json = compiled_json(kind, plan, c, kp, agg_nodes, specs, out_exprs, len_idx, first_idx, maintain_order);
}

// Select(aggregations) over [Filter]* over `src`
static bool fused_select(Plan& plan, const IRN& node, const std::vector<int>& preds, const FramePtr& src, FramePtr& out, Shape* shape_out, int* sid_out,
                         std::string* why, bool compile_only) {
  for (int e : node.exprs) if (!contains_agg(plan, e) || contains_column_outside_agg(plan, e)) { if (why) *why = "select mixes row expressions and aggregations"; return false; }
  Compiler c(plan, *src);
  std::vector<int> agg_nodes;
  std::vector<FinalSpec> specs;
  try {
    int p = -1;
    for (int pe : preds) { int n = c.lower(pe); if (c.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : c.mk(OP_AND, p, n, 'b'); }
    c.pred = p;
    for (int e : node.exprs) collect_aggs(plan, e, agg_nodes);
    for (int a : agg_nodes) specs.push_back(c.lower_agg(a));
    c.finish();
  } catch (const Unsupported& u) { if (why) *why = u.why; return false; }
  const int static_id = find_static_shape(c.shape);
  if (shape_out) *shape_out = c.shape;
  if (sid_out) *sid_out = static_id;
  if (compile_only) { dump_compiled(t_program_dump, "select", plan, c, nullptr, agg_nodes, specs, node.exprs, -1, -1, false); return true; }
  FusedAggResult r; r.n_groups = 1; r.n_aggs = c.shape.n_aggs;
  std::vector<uint64_t> host(kMaxAggs, 0);
  if (src->height == 0) { for (int k2 = 0; k2 < c.shape.n_aggs; k2++) host[k2] = agg_identity(c.shape.aggs[k2].kind); }
  else k::fused_regagg(c.shape, c.args, static_id, host.data());
  r.acc = dev_alloc(sizeof(uint64_t) * kMaxAggs);
  h2d_async(r.acc->ptr, host.data(), sizeof(uint64_t) * (size_t)c.shape.n_aggs);
  PLX_HIP(hipStreamSynchronize(stream()));
  plan.desc += std::string("FusedFilterAgg{fused_scan[") + jit::program_mode(static_id, c.args.n_rows) + "]+register_sink, inputs=" + std::to_string(c.shape.n_inputs) + ", ops=" + std::to_string(c.shape.n_ops) + ", aggs=" + std::to_string(c.shape.n_aggs) + "}; ";
  std::map<int, ColumnPtr> overrides;
  FinBatch batch{};
  for (size_t i = 0; i < agg_nodes.size(); i++) overrides[agg_nodes[i]] = finalize_column(r, specs[i], batch);
  k::finalize_batch(r.acc->as<uint64_t>(), r.n_aggs, r.n_groups, batch);
  out = std::make_shared<Frame>();
  out->height = 1;
  Frame one; one.height = 1;
  for (int e : node.exprs) {
    Evaluated ev = eval(plan, e, one, &overrides);
    out->names.push_back(output_name(plan, e));
    out->cols.push_back(ev.col);
  }
  return true;
}

// GroupBy(keys, aggregations) over [Filter]* over `src`
static bool fused_groupby(Plan& plan, const IRN& node, const std::vector<int>& preds, const FramePtr& src, FramePtr& out, Shape* shape_out, int* sid_out,
                          std::string* why, bool compile_only) {
  for (int e : node.exprs) if (!contains_agg(plan, e) || contains_column_outside_agg(plan, e)) { if (why) *why = "group_by aggregation list contains a non-aggregated column"; return false; }
  Compiler c(plan, *src);
  std::vector<int> agg_nodes;
  std::vector<FinalSpec> specs;
  KeyPlan kp;
  int len_idx = -1, first_idx = -1;
  try {
    int p = -1;
    for (int pe : preds) { int n = c.lower(pe); if (c.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : c.mk(OP_AND, p, n, 'b'); }
    c.pred = p;
    kp = lower_keys(c, node.keys);
    len_idx = c.add_agg(AGG_LEN, -1);
    for (int e : node.exprs) collect_aggs(plan, e, agg_nodes);
    for (int a : agg_nodes) specs.push_back(c.lower_agg(a));
    if (node.maintain_order) first_idx = c.add_agg(AGG_FIRST_ROW, -1);
    c.finish();
  } catch (const Unsupported& u) { if (why) *why = u.why; return false; }
  if (shape_out) *shape_out = c.shape;
  if (sid_out) *sid_out = find_static_shape(c.shape);
  if (compile_only) { dump_compiled(t_program_dump, "group_by", plan, c, &kp, agg_nodes, specs, node.exprs, len_idx, first_idx, node.maintain_order != 0); return true; }
  FusedAggResult r;
  std::string d;
  try {
    run_fused_groupby(c, kp, len_idx, r, d);
  } catch (const Error& e) {
    // a row outside bounds the planner had only ASSUMED (assume_range): forget the guesses, never guess about these columns again, and run the query once more --
    // its statistics now come from exact passes.  (Bounds the caller declared are the caller's promise: that error stands.)
    bool guessed = false;
    for (auto& col : c.cols) if (col->range_assumed) { col->range_state = 0; col->range_trusted = true; col->range_assumed = false; col->range_verified = false; col->no_assume = true; std::atomic_store(&col->key_sample, std::shared_ptr<void>()); guessed = true; }
    if (!guessed || e.code != PLX_ERR_INVALID) throw;
    plan.desc += "AssumedBoundsViolated{exact statistics, second run}; ";
    return fused_groupby(plan, node, preds, src, out, shape_out, sid_out, why, compile_only);
  }
  plan.desc += kp.note + "FusedFilterGroupBy{" + d + ", inputs=" + std::to_string(c.shape.n_inputs) + ", ops=" + std::to_string(c.shape.n_ops) + ", aggs=" + std::to_string(c.shape.n_aggs) + ", groups=" + std::to_string(r.n_groups) + "}; ";
  out = std::make_shared<Frame>();
  out->height = r.n_groups;
  FinBatch batch{};
  for (size_t pi = 0; pi < kp.parts.size(); pi++) { out->names.push_back(output_name(plan, kp.parts[pi].expr)); out->cols.push_back(decode_key_column(r, kp.parts[pi], (int)pi, batch)); }
  std::map<int, ColumnPtr> overrides;
  for (size_t i = 0; i < agg_nodes.size(); i++) overrides[agg_nodes[i]] = finalize_column(r, specs[i], batch);
  k::finalize_batch(r.acc->as<uint64_t>(), r.n_aggs, r.n_groups, batch);
  Frame gframe; gframe.height = r.n_groups;
  for (int e : node.exprs) {
    Evaluated ev = eval(plan, e, gframe, &overrides);
    out->names.push_back(output_name(plan, e));
    out->cols.push_back(broadcast(ev, r.n_groups));
  }
  if (node.maintain_order && r.n_groups > 1) {
    ColumnPtr perm = first_row_permutation(r, first_idx);
    for (auto& col : out->cols) col = ops::gather(col, perm);
  }
  return true;
}


static int peel_filters_node(const Plan& plan, int input) { while (plan.ir[input].kind == PLX_IR_FILTER) input = plan.ir[input].input; return input; }
// peel [Filter]* below an aggregation node
static int peel_filters(const Plan& plan, int input, std::vector<int>& preds) {
  while (plan.ir[input].kind == PLX_IR_FILTER) { preds.push_back(plan.ir[input].predicate); input = plan.ir[input].input; }
  std::reverse(preds.begin(), preds.end());
  return input;
}

// ------------------------------------------------ fused Join -> GroupBy pipeline ----
// GroupBy(keys, aggs) directly over an inner Join of two ([Filter]* Scan) inputs, when
//   * the join has one plain-column integer key pair,
//   * the group keys are the join key plus plain columns of the BUILD side (the shorter input,
//     hash_join/mod.rs:41-50), so a group is one build row,
//   * every aggregate reads PROBE-side columns only,
//   * the build keys that survive the build-side predicate are unique (checked at run time).
// TPC-H Q3 has this shape.  Pipeline: count build rows -> build scan (predicate fused) -> probe scan
// (predicate + expressions fused, aggregates land in the matching slot) -> compact -> gather the
// build-side key columns.  Anything else returns false and the caller runs the per-node path.
// ---- inner joins that only FILTER (semi-join rewrite) ---------------------------------------------------------------------
// `X JOIN Y ON x = y` where one side (the filter side) has unique join keys and contributes no column to anything above the
// join is a semi join of the other (payload) side: TPC-H Q3's `customer[c_mktsegment == ..] JOIN orders` only restricts orders.
// The filter side becomes a membership bitmap over its key range (one fused scan of the filter side), the payload side's
// scan tests `bit(key)` as one more conjunct of its predicate (OP_BITLOOKUP) -- no join output is materialised.
struct SemiFilter {
  FramePtr F;                 // filter side
  std::vector<int> fpreds;    // its predicates
  int fkey = -1;              // join key column of F
  int pkey = -1;              // join key column of the payload frame
};
static void collect_columns(const Plan& plan, int e, std::set<std::string>& out) {
  if (e < 0) return;
  const AE& x = plan.ae.at(e);
  if (x.kind == PLX_AE_COLUMN) { out.insert(x.name); return; }
  collect_columns(plan, x.lhs, out);
  collect_columns(plan, x.rhs, out);
}
// Resolves one input of the outer join to a scan node: [Filter]* Scan, or [Filter]* Join(inner; A, B) with A, B = [Filter]* Scan
// where one of A / B is a pure filter with respect to `used` (the column names referenced above).  Appends the payload side's
// predicates to `preds` and the filter to `semis`; returns the payload scan node or -1 (why set).
static int resolve_join_side(const Plan& plan, int node, std::set<std::string> used, std::vector<int>& preds, std::vector<SemiFilter>& semis, std::string* why) {
  auto no = [&](const char* m) { if (why) *why = m; return -1; };
  const int n = peel_filters(plan, node, preds);
  if (plan.ir[n].kind == PLX_IR_SCAN) return n;
  if (plan.ir[n].kind != PLX_IR_JOIN) return no("join inputs are not filtered scans");
  const IRN& j = plan.ir[n];
  if (j.how != PLX_JOIN_INNER || j.keys.size() != 1 || j.keys_right.size() != 1) return no("nested join is not a single-key inner join");
  auto plain = [&](int e) -> const AE* { const AE* x = &plan.ae[e]; while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs]; return x->kind == PLX_AE_COLUMN ? x : nullptr; };
  const AE* ka = plain(j.keys[0]);
  const AE* kb = plain(j.keys_right[0]);
  if (!ka || !kb) return no("nested join keys are expressions");
  std::vector<int> ap, bp;
  const int a = peel_filters(plan, j.input, ap), b = peel_filters(plan, j.input_right, bp);
  if (plan.ir[a].kind != PLX_IR_SCAN || plan.ir[b].kind != PLX_IR_SCAN) return no("nested join inputs are not filtered scans");
  FramePtr A = get_frame(plan.ir[a].frame), B = get_frame(plan.ir[b].frame);
  const int kai = A->find(ka->name), kbi = B->find(kb->name);
  if (kai < 0 || kbi < 0) return no("nested join key column not found");
  if (A->cols[kai]->dtype != B->cols[kbi]->dtype || !dtype_is_int(A->cols[kai]->dtype) || A->cols[kai]->dtype == PLX_U64) return no("nested join key is not a signed / narrow integer column pair of one dtype");
  for (size_t i = 0; i < B->names.size(); i++) if ((int)i != kbi && A->find(B->names[i]) >= 0) return no("nested join sides share a column name (suffix renaming is not modelled)");
  for (int pe : preds) collect_columns(plan, pe, used);        // predicates above the nested join see the joined frame
  bool used_a = false, used_b = false;
  for (auto& nm : A->names) used_a = used_a || used.count(nm);
  for (size_t i = 0; i < B->names.size(); i++) if ((int)i != kbi) used_b = used_b || used.count(B->names[i]);
  if (used.count(kb->name) && kb->name != ka->name) return no("the right key of the nested join is referenced above it (coalesced away)");
  SemiFilter sf;
  int payload = -1;
  if (!used_a) { sf.F = A; sf.fpreds = ap; sf.fkey = kai; sf.pkey = kbi; payload = b; preds.insert(preds.begin(), bp.begin(), bp.end()); }
  else if (!used_b) { sf.F = B; sf.fpreds = bp; sf.fkey = kbi; sf.pkey = kai; payload = a; preds.insert(preds.begin(), ap.begin(), ap.end()); }
  else return no("both sides of the nested join are referenced above it");
  semis.push_back(sf);
  return payload;
}
// key range of a semi filter's key column: exact statistics when the column has data, the declared range of a placeholder
static bool semi_key_range(const SemiFilter& sf, int64_t* mn, int64_t* mx) {
  const ColumnPtr& c = sf.F->cols[sf.fkey];
  if (c->values) return ops::int_range(c, mn, mx);
  if (c->range_state == 1) { *mn = c->range_min; *mx = c->range_max; return true; }
  *mn = 0; *mx = 0;
  return true;
}

static bool fused_join_groupby(Plan& plan, const IRN& gb, FramePtr& out, std::string* why, std::vector<Shape>* shapes_out = nullptr, bool compile_only = false) {
  auto no = [&](const char* m) { if (why) *why = m; return false; };
  if (gb.input < 0 || plan.ir[gb.input].kind != PLX_IR_JOIN) return no("input is not a join");
  const IRN& jn = plan.ir[gb.input];
  if ((jn.how != PLX_JOIN_INNER && jn.how != PLX_JOIN_LEFT) || jn.keys.size() != 1 || jn.keys_right.size() != 1) return no("not a single-key inner or left join");
  // LEFT join (single_keys_left.rs:106-195: every left row survives; rows without a match carry nulls in the right table's columns): the matched rows are the inner join's
  // -- the same pipeline with the RIGHT table as the build side -- and the unmatched ones are a group-by of their own over the left table, keyed by the join key, behind
  // the predicate "key not among the build keys" (a membership bitmap over the build key range, tested inside the scan: OP_BITLOOKUP); their groups carry nulls in the
  // build-side group columns.  Needs a build key range a bitmap can cover; otherwise the per-node path runs the join.
  const bool left_join = jn.how == PLX_JOIN_LEFT;
  if (gb.maintain_order) return no("maintain_order");
  auto plain = [&](int e) -> const AE* { const AE* x = &plan.ae[e]; while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs]; return x->kind == PLX_AE_COLUMN ? x : nullptr; };
  const AE* lkx = plain(jn.keys[0]);
  const AE* rkx = plain(jn.keys_right[0]);
  if (!lkx || !rkx) return no("join keys are expressions");
  std::vector<int> lpreds, rpreds;
  std::vector<SemiFilter> lsemis, rsemis;
  std::set<std::string> used;                       // column names referenced above the join inputs
  for (int e : gb.keys) collect_columns(plan, e, used);
  for (int e : gb.exprs) collect_columns(plan, e, used);
  used.insert(lkx->name); used.insert(rkx->name);
  const int lsrc = resolve_join_side(plan, jn.input, used, lpreds, lsemis, why);
  if (lsrc < 0) return false;
  const int rsrc = resolve_join_side(plan, jn.input_right, used, rpreds, rsemis, why);
  if (rsrc < 0) return false;
  if (lsemis.size() + rsemis.size() > (size_t)kMaxLuts) return no("more nested filter joins than lookup bitmaps");
  FramePtr L = get_frame(plan.ir[lsrc].frame), R = get_frame(plan.ir[rsrc].frame);
  const int lki = L->find(lkx->name), rki = R->find(rkx->name);
  if (lki < 0 || rki < 0) return no("join key column not found");
  const int kdt = L->cols[lki]->dtype;
  if (kdt != R->cols[rki]->dtype || !dtype_is_int(kdt)) return no("join key is not an integer column pair of one dtype");
  if (L->height >= 0xffffffffll || R->height >= 0xffffffffll) return no("side exceeds u32 row indices");
  const bool build_right = left_join || L->height > R->height;   // det_hash_prone_order: probe = the longer relation; a left join probes with its left table (single_keys_left.rs)
  const FramePtr& B = build_right ? R : L;
  const FramePtr& P = build_right ? L : R;
  const int bki = build_right ? rki : lki, pki = build_right ? lki : rki;
  const std::vector<int>& bpreds = build_right ? rpreds : lpreds;
  const std::vector<int>& ppreds = build_right ? lpreds : rpreds;
  const std::vector<SemiFilter>& bsemis = build_right ? rsemis : lsemis;
  const std::vector<SemiFilter>& psemis = build_right ? lsemis : rsemis;
  // joined-frame naming (_finish_join, general.rs:17-49): left columns, then right columns except the coalesced right key
  struct Src { int side; int idx; };   // side 0 = left, 1 = right
  std::map<std::string, Src> joined;
  for (size_t i = 0; i < L->names.size(); i++) joined[L->names[i]] = {0, (int)i};
  for (size_t i = 0; i < R->names.size(); i++) {
    if ((int)i == rki) continue;
    std::string name = R->names[i];
    if (joined.count(name)) name += jn.suffix;
    if (joined.count(name)) return no("duplicate output column name");
    joined[name] = {1, (int)i};
  }
  const int build_side = build_right ? 1 : 0;
  // group keys
  struct GKey { bool is_join_key; int build_col; int expr; int dtype; };
  std::vector<GKey> gkeys;
  bool has_join_key = false;
  for (int e : gb.keys) {
    const AE* x = plain(e);
    if (!x) return no("group key is an expression");
    auto it = joined.find(x->name);
    if (it == joined.end()) fail(PLX_ERR_NOT_FOUND, "column not found: " + x->name);
    const Src sc = it->second;
    const bool is_key = (sc.side == 0 && sc.idx == lki);   // the coalesced key column carries the left name
    if (is_key) { has_join_key = true; gkeys.push_back({true, -1, e, kdt}); }
    else if (sc.side == build_side) gkeys.push_back({false, sc.idx, e, B->cols[sc.idx]->dtype});
    else return no("group key from the probe side that is not the join key");
  }
  if (!has_join_key) return no("group keys do not include the join key");
  // aggregates: probe-side columns only, resolved through the joined names
  Frame pview; pview.height = P->height;
  for (auto& kv : joined) if (kv.second.side != build_side) { pview.names.push_back(kv.first); pview.cols.push_back(P->cols[kv.second.idx]); }
  if (!build_right) { pview.names.push_back(lkx->name); pview.cols.push_back(P->cols[pki]); }  // coalesced key: probe values == build values on matches
  std::function<bool(int)> probe_only = [&](int e) -> bool {
    if (e < 0) return true;
    const AE& x = plan.ae[e];
    if (x.kind == PLX_AE_COLUMN) return pview.find(x.name) >= 0;
    return probe_only(x.lhs) && probe_only(x.rhs);
  };
  for (int e : gb.exprs) {
    if (!contains_agg(plan, e) || contains_column_outside_agg(plan, e)) return no("aggregation list contains a non-aggregated column");
    if (!probe_only(e)) return no("aggregate reads a build-side column");
  }
  // ---- compile the three programs
  Compiler cnt(plan, *B), cb(plan, *B), cp(plan, *P), cs(plan, *P), ca(plan, *P);      // ca: the left join's unmatched rows (group-by over the probe side)
  KeyPlan akp;
  std::vector<FinalSpec> aspecs;
  int a_len_idx = -1;
  int64_t a_kmn = 0, a_kmx = 0;
  uint64_t a_range = 0;
  if (left_join) {
    if (psemis.size() + 1 > (size_t)kMaxLuts) return no("left join: no lookup bitmap left for the membership test");
    if (kdt == PLX_U64) return no("left join on UInt64 keys");
    bool have = false;                                          // (a placeholder column -- compile-only callers -- has its declared range or none)
    if (B->height > 0 && B->cols[bki]->values) have = ops::int_range(B->cols[bki], &a_kmn, &a_kmx);
    else if (B->height > 0 && B->cols[bki]->range_state == 1) { a_kmn = B->cols[bki]->range_min; a_kmx = B->cols[bki]->range_max; have = true; }
    const unsigned __int128 range128 = have ? (unsigned __int128)((__int128)a_kmx - (__int128)a_kmn) + 1 : 1;
    if (range128 > ((unsigned __int128)1 << 34) || (have && range128 > (unsigned __int128)B->height * 256 + 4096)) return no("left join: build key range too wide for the membership bitmap");
    a_range = (uint64_t)range128;
  }
  std::vector<std::unique_ptr<Compiler>> csemi;      // one program per semi filter: build-side filters first, then probe-side
  std::vector<int> agg_nodes; std::vector<FinalSpec> specs;
  int len_idx = -1;
  try {
    // conjunction of the side's predicates and of the membership tests of its semi filters (lookup bitmap i = args.lut[lut0 + i])
    auto and_preds = [&](Compiler& c, const std::vector<int>& preds, const std::vector<SemiFilter>& semis, int lut0) {
      int p = -1;
      for (int pe : preds) { int n = c.lower(pe); if (c.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : c.mk(OP_AND, p, n, 'b'); }
      for (size_t i = 0; i < semis.size(); i++) {
        int64_t mn = 0, mx = 0;
        if (!semi_key_range(semis[i], &mn, &mx)) mn = mx = 0;      // empty filter side: the bitmap is empty, nothing matches
        // (rows the cheaper conjuncts already rejected do not look the bitmap up: OP_MASKV; the conjunction is the same either way)
        const int kn = c.load(semis[i].pkey);
        const int n = c.bit_lookup(p < 0 ? kn : c.mask_valid(kn, p), lut0 + (int)i, mn);
        p = p < 0 ? n : c.mk(OP_AND, p, n, 'b');
      }
      return p;
    };
    cnt.pred = and_preds(cnt, bpreds, bsemis, 0);
    const int bk_cnt = cnt.load(bki);
    cnt.add_agg(cnt.nodes[bk_cnt].nullable ? AGG_COUNT : AGG_LEN, cnt.nodes[bk_cnt].nullable ? bk_cnt : -1);
    cnt.finish();
    cb.pred = and_preds(cb, bpreds, bsemis, 0);
    cb.key = cb.load(bki);
    cb.finish();
    cp.pred = and_preds(cp, ppreds, psemis, (int)bsemis.size());
    cp.key = cp.load(pki);
    cp.df = &pview;
    len_idx = cp.add_agg(AGG_LEN, -1);
    for (int e : gb.exprs) collect_aggs(plan, e, agg_nodes);
    for (int a : agg_nodes) specs.push_back(cp.lower_agg(a));
    cp.finish();
    // the probe side's predicate + key alone, row id as payload: the program of the partitioned probe's scatter (k::partitioned_probe_hits)
    cs.pred = and_preds(cs, ppreds, psemis, (int)bsemis.size());
    cs.key = cs.load(pki);
    cs.df = &pview;
    cs.add_agg(AGG_FIRST_ROW, -1);
    cs.finish();
    if (left_join) {
      ca.df = &pview;
      int p = and_preds(ca, ppreds, psemis, 0);                                  // (its own lookup numbering: the probe side's semi filters, then the membership bitmap)
      const int kn = ca.load(pki);
      const int member = ca.ifnull(ca.bit_lookup(p < 0 ? kn : ca.mask_valid(kn, p), (int)psemis.size(), a_kmn), 0);      // a null key is among nobody's keys
      const int unmatched = ca.mk(OP_NOT, member, member, 'b');
      ca.pred = p < 0 ? unmatched : ca.mk(OP_AND, p, unmatched, 'b');
      int jk_expr = -1;
      for (auto& gk : gkeys) if (gk.is_join_key) jk_expr = gk.expr;
      akp = lower_keys(ca, std::vector<int>{jk_expr});
      a_len_idx = ca.add_agg(AGG_LEN, -1);
      for (int a : agg_nodes) aspecs.push_back(ca.lower_agg(a));
      ca.finish();
    }
    for (const std::vector<SemiFilter>* sv : {&bsemis, &psemis}) {
      for (const SemiFilter& sf : *sv) {
        csemi.emplace_back(new Compiler(plan, *sf.F));
        Compiler& cf = *csemi.back();
        cf.pred = and_preds(cf, sf.fpreds, {}, 0);
        cf.key = cf.load(sf.fkey);
        cf.finish();
      }
    }
  } catch (const Unsupported& u) { if (why) *why = u.why; return false; }
  // count, build, probe, the semi filters' scans, and LAST the probe side's predicate + key program of the partitioned probe's scatter
  if (shapes_out) { shapes_out->push_back(cnt.shape); shapes_out->push_back(cb.shape); shapes_out->push_back(cp.shape); for (auto& cf : csemi) shapes_out->push_back(cf->shape); shapes_out->push_back(cs.shape); }
  if (compile_only) {
    if (t_program_dump) {   // the three scans + how groups, group keys and outputs are derived from them (tests/program_eval.py evaluate_join)
      std::ostringstream o;
      o << "{\"kind\":\"join_group_by\",\"how\":\"" << (left_join ? "left" : "inner") << "\",\"build_side\":\"" << (build_right ? "right" : "left") << "\",\"build_key\":" << jstr(B->names[bki]) << ",\"probe_key\":" << jstr(P->names[pki])
        << ",\"count\":{" << program_fields(cnt, *B) << "},\"build\":{" << program_fields(cb, *B) << "},\"probe\":{" << program_fields(cp, *P) << "},\"len_idx\":" << len_idx << ",\"group_keys\":[";
      for (size_t i = 0; i < gkeys.size(); i++)
        o << (i ? "," : "") << "{\"name\":" << jstr(output_name(plan, gkeys[i].expr)) << ",\"is_join_key\":" << (gkeys[i].is_join_key ? 1 : 0) << ",\"build_col\":"
          << (gkeys[i].is_join_key ? std::string("null") : jstr(B->names[gkeys[i].build_col])) << ",\"dtype\":" << gkeys[i].dtype << "}";
      o << "]," << finals_outputs_fields(plan, agg_nodes, specs, gb.exprs) << ",\"semis\":[";
      {
        size_t ci = 0;
        for (int side = 0; side < 2; side++) {
          const std::vector<SemiFilter>& sv = side == 0 ? bsemis : psemis;
          for (size_t i = 0; i < sv.size(); i++, ci++) {
            int64_t mn = 0, mx = 0; semi_key_range(sv[i], &mn, &mx);
            o << (ci ? "," : "") << "{\"side\":\"" << (side == 0 ? "build" : "probe") << "\",\"lut\":" << (side == 0 ? i : bsemis.size() + i) << ",\"kmin\":\"" << mn << "\",\"kmax\":\"" << mx
              << "\",\"filter_key\":" << jstr(sv[i].F->names[sv[i].fkey]) << ",\"payload_key\":" << jstr((side == 0 ? B : P)->names[sv[i].pkey]) << ",\"filter\":{" << program_fields(*csemi[ci], *sv[i].F) << "}}";
          }
        }
      }
      o << "]";
      // a left join's unmatched rows: a group-by program over the probe side whose predicate ends in NOT member(key) -- lookup bitmap `lut` = the build keys that pass the build program
      if (left_join) o << ",\"unmatched\":{\"lut\":" << psemis.size() << ",\"kmin\":\"" << a_kmn << "\",\"range\":\"" << a_range << "\",\"program\":"
                       << compiled_json("group_by", plan, ca, &akp, agg_nodes, aspecs, gb.exprs, a_len_idx, -1, false) << "}";
      o << "}";
      t_program_dump->json = o.str();
    }
    return true;
  }
  // ---- run
  // semi filters first: one fused scan of each filter side -> membership bitmap; its set bits must equal the rows that passed
  // (unique filter keys), otherwise the nested join multiplies rows and the per-node path has to run it
  std::vector<Buf> lut_bits;
  {
    size_t ci = 0;
    for (int side = 0; side < 2; side++) {
      const std::vector<SemiFilter>& sv = side == 0 ? bsemis : psemis;
      for (size_t i = 0; i < sv.size(); i++, ci++) {
        const SemiFilter& sf = sv[i];
        int64_t mn = 0, mx = 0;
        const bool have = sf.F->height > 0 && semi_key_range(sf, &mn, &mx);
        const unsigned __int128 range128 = have ? (unsigned __int128)((__int128)mx - (__int128)mn) + 1 : 1;
        if (range128 > ((unsigned __int128)1 << 34) || (have && range128 > (unsigned __int128)sf.F->height * 256 + 4096)) return no("nested filter join: key range too wide for a bitmap");
        const uint64_t range = (uint64_t)range128;
        Buf bits = dev_alloc_zero(sizeof(uint64_t) * (size_t)(range / 64 + 2)), rows_dev = dev_alloc_zero(8);
        if (have) {
          BitmapBuild bb; bb.bits = bits->as<unsigned long long>(); bb.count = rows_dev->as<unsigned long long>(); bb.kmin = mn; bb.range = range;
          Compiler& cf = *csemi[ci];
          k::fused_bitmap_build(cf.shape, cf.args, bb, find_static_shape(cf.shape));
          uint64_t rows_in = 0;
          d2h_sync(&rows_in, rows_dev->ptr, 8);
          if ((uint64_t)k::bitmap_popcount(bits->as<uint64_t>(), (int64_t)range) != rows_in) return no("nested filter join: the filter side's keys are not unique");
          plan.desc += "SemiFilter{" + sf.F->names[sf.fkey] + " -> bitmap range=" + std::to_string(range) + " rows=" + std::to_string(rows_in) + "/" + std::to_string(sf.F->height) + "}; ";
        }
        const Lut lut{bits->as<unsigned long long>(), range};
        const int li = side == 0 ? (int)i : (int)(bsemis.size() + i);
        if (side == 0) { cnt.args.lut[li] = lut; cb.args.lut[li] = lut; } else { cp.args.lut[li] = lut; cs.args.lut[li] = lut; if (left_join) ca.args.lut[i] = lut; }
        lut_bits.push_back(bits);
      }
    }
  }
  uint64_t nb = 0;
  // Duplicate build keys (the reference's tables map a key to a LIST of build rows: single_keys.rs:16-167, probe_inner emits one pair per entry, single_keys_inner.rs:11-38):
  // the hash-table pipeline then runs in multi-value mode -- chains of build rows per key, a group = a build ROW (or the rows of a key that agree on the build-side group
  // columns: canonicalise_chains), every probe row adds to the cells of each row of its key's chain.  Possible when those group columns are integer-typed (compared bitwise).
  bool known_dups = B->height > 0 && B->cols[bki]->repeats_as_build_key, multi = false;
  uint32_t merged_rows = 0;      // multi-value mode: build rows that share their group with an earlier row of their key
  RepCols rep_cols{};
  bool multi_ok = !(getenv("PLX_JOIN_MULTI") && getenv("PLX_JOIN_MULTI")[0] == '0');
  for (auto& gk : gkeys) {
    if (gk.is_join_key) continue;
    const ColumnPtr& col = B->cols[gk.build_col];
    const int w = col->dtype == PLX_BOOL || col->dtype == PLX_F32 || col->dtype == PLX_F64 ? 0 : dtype_width(col->dtype);
    if (!w || rep_cols.n >= kMaxRepCols) { multi_ok = false; break; }
    rep_cols.vals[rep_cols.n] = col->data(); rep_cols.valid[rep_cols.n] = (const unsigned long long*)col->valid_words(); rep_cols.width[rep_cols.n] = w; rep_cols.n++;
  }
  const int probe_static_id = find_static_shape(cp.shape);
  FusedAggResult r; r.n_aggs = cp.shape.n_aggs;
  auto rows = std::make_shared<Column>();
  rows->dtype = PLX_U32; rows->null_count = 0;
  int64_t G = 0;
  bool done = false;
  // -- direct-address table when the build key range is small (cached column statistics); needs no count pass
  int64_t kmn = 0, kmx = 0;
  if (!(plan.flags & PLX_PLAN_NO_DIRECT_JOIN) && !known_dups && B->height > 0 && kdt != PLX_U64 && ops::int_range(B->cols[bki], &kmn, &kmx)) {
    const unsigned __int128 range128 = (unsigned __int128)((__int128)kmx - (__int128)kmn) + 1;
    // pair list capacity: every build row may pass + one ordinal chunk (1024, kOrdChunk) per wave
    const uint64_t ord_cap = (uint64_t)B->height + (uint64_t)k::scan_waves(B->height) * 1024 + 1024;
    if (range128 <= ((unsigned __int128)1 << 34) && range128 <= (unsigned __int128)B->height * 256 && ord_cap < 0xfffffff0ull) {
      const uint64_t range = (uint64_t)range128;
      const size_t n_blocks = (size_t)(range / 512 + 1), n_words = n_blocks * 8;
      Buf bits = dev_alloc_zero(sizeof(uint64_t) * n_words), rank = dev_alloc(sizeof(uint32_t) * n_words);
      Buf okey = dev_alloc(sizeof(uint64_t) * ord_cap), orow = dev_alloc(sizeof(uint32_t) * ord_cap), used = dev_alloc_zero(sizeof(uint32_t) * (ord_cap / 1024 + 2));
      Buf meta = dev_alloc_zero(32);   // [0] ordinal counter, [2..3] flags, [4..5] pairs appended (u64)
      DirectJoinTable dt{}; dt.bits = bits->as<unsigned long long>(); dt.rank = rank->as<unsigned int>(); dt.ord_key = okey->as<unsigned long long>(); dt.ord_row = orow->as<unsigned int>();
      dt.chunk_used = used->as<unsigned int>(); dt.counter = meta->as<unsigned int>(); dt.flags = meta->as<unsigned int>() + 2; dt.acc = nullptr; dt.kmin = kmn; dt.range = range; dt.n_ord = (unsigned int)ord_cap;
      dt.opts = probe_late_loads() ? kDirectLateLoads : 0u;
      k::fused_direct_build(cb.shape, cb.args, dt, find_static_shape(cb.shape));
      // every wave closes its last chunk when it finishes, so chunk_used is final once the build scan is: the rank launch counts
      // the pairs over the whole reserved capacity (the ordinal counter itself is only read back with the rank's sync)
      uint64_t pairs = 0;
      nb = k::direct_rank(dt, rank->as<uint32_t>(), (int64_t)ord_cap, meta->as<uint64_t>() + 2, &pairs);   // synchronises: flags and the ordinal counter are final too
      uint32_t m4[4] = {0, 0, 0, 0};
      d2h_sync(m4, meta->ptr, 16);
      PLX_REQUIRE(!m4[3], PLX_ERR_INVALID, "direct join build: ordinal overflow");
      if (pairs != nb) known_dups = true;                          // two pairs shared a bit: duplicate build keys -- the hash-table pipeline below runs them in multi-value mode
      else {
      const uint32_t n_used = m4[0];
      const int64_t n_slots = (int64_t)nb, s1 = std::max<int64_t>(n_slots, 1);
      Buf acc2 = dev_alloc(sizeof(uint64_t) * (size_t)s1 * cp.shape.n_aggs);
      k::init_agg_cells(acc2->as<uint64_t>(), s1, cp.shape);   // LEN = 0: build rows no probe row matched never show up
      dt.acc = acc2->as<unsigned long long>();
      // Probe.  Keys in (rough) key order walk the bitmap out of the L2; keys in no order fetch one line per row across the fabric: those
      // are partitioned by key range first and probed against LDS-resident bitmap slices, and the ordinary probe kernel then runs over
      // the matching rows only (gathered).  Worth it when the bitmap is far larger than an L2 (4 MB) and few rows can match.
      std::string probe_how = "probe_agg";
      bool probed = false;
      Buf touch;
      const int pmode = partitioned_probe_mode();
      if (pmode == 2 || (pmode == 1 && P->height >= ((int64_t)1 << 24) && range >= ((uint64_t)1 << 28) && nb * 8 <= range)) {
        const ColumnPtr& pk = P->cols[pki];
        if (pk->order_state == 0) pk->order_state = k::sample_sortedness(pk) >= 0.9 ? 1 : 2;
        ColumnPtr hits;
        std::string pd;
        if ((pmode == 2 || pk->order_state == 2) && k::partitioned_probe_hits(cs.shape, cs.args, dt, nb, find_static_shape(cs.shape), &hits, &pd)) {
          if (hits->len > 0) {
            Args a2 = cp.args;
            a2.n_rows = hits->len;
            std::vector<ColumnPtr> srcs;
            for (int i = 0; i < cp.shape.n_inputs; i++) srcs.push_back(cp.cols[cp.input_cols[i]]);
            std::vector<ColumnPtr> keep = ops::gather_columns(srcs, hits);
            for (int i = 0; i < cp.shape.n_inputs; i++) { a2.in[i].values = keep[i]->data(); a2.in[i].validity = cp.shape.in_nullable[i] ? keep[i]->valid_words() : nullptr; }
            k::fused_direct_probe_agg(cp.shape, a2, dt, probe_static_id);
            // the pair-list compaction below looks every build pair up in bitmap, rank and LEN cells -- random lines for a shuffled build side.  Only keys
            // among the candidates can have been matched: a 2^26-bit filter of their hashes (8 MB: cache-resident) lets the compaction skip the rest
            const int key_in = slot_input(cp.shape, cp.shape.key);
            if (key_in >= 0 && (size_t)key_in < keep.size()) {
              constexpr unsigned int kLog2TouchBits = 26;
              touch = dev_alloc_zero(sizeof(uint64_t) * ((size_t)1 << (kLog2TouchBits - 6)));
              ColumnPtr k64 = keep[key_in]->dtype == PLX_I64 ? keep[key_in] : ops::cast(keep[key_in], PLX_I64);
              k::touch_filter_set(k64->values->as<int64_t>(), k64->valid_words(), hits->len, kLog2TouchBits, touch->as<uint64_t>());
              dt.touch_filter = touch->as<unsigned long long>(); dt.log2_touch_bits = kLog2TouchBits;
            }
            PLX_HIP(hipStreamSynchronize(stream()));       // the gathered columns live until the kernels have read them
          }
          probe_how = pd + "+gather+probe_agg";
          probed = true;
        }
      }
      if (!probed) k::fused_direct_probe_agg(cp.shape, cp.args, dt, probe_static_id);
      // one pass over the pair list into buffers sized for every slot (G <= n_slots)
      r.packed_keys = dev_alloc(sizeof(uint64_t) * (size_t)s1);
      r.acc = dev_alloc(sizeof(uint64_t) * (size_t)s1 * r.n_aggs);
      rows->values = dev_alloc(values_bytes(PLX_U32, s1));
      G = n_slots ? k::direct_agg_compact(dt, (int64_t)std::min<uint64_t>(n_used, ord_cap), r.n_aggs, len_idx, r.packed_keys->as<uint64_t>(), rows->values->as<uint32_t>(), r.acc->as<uint64_t>()) : 0;
      r.n_groups = G;
      rows->len = G;
      plan.desc += std::string("FusedJoinGroupBy{build=") + (build_right ? "right" : "left") + " rows=" + std::to_string(nb) + "/" + std::to_string(B->height) + " direct-address table range=" +
                   std::to_string(range) + " (bitmap + rank) unique-keys, probe rows=" + std::to_string(P->height) + ", fused_scan[" + jit::program_mode(probe_static_id, cp.args.n_rows) + "]+" + probe_how + ", aggs=" +
                   std::to_string(r.n_aggs) + ", groups=" + std::to_string(G) + "}; ";
      done = true;
      }  // unique build keys
    }
  }
  // Size of the build table.  The number of build rows that pass the build side's predicate is a by-product of the build scan itself (JoinBuildSink counts what
  // it inserts), so a large build side is sized from a strided sample of the count program (4 blocks of 2^18 rows, + 25 %) instead of a counting pass over the
  // whole side; if the sample misjudged (table more than 0.7 full, or a probe sequence overflowed) the table is rebuilt once from the exact count.
  auto exact_count = [&]() -> uint64_t { std::vector<uint64_t> host(kMaxAggs, 0); k::fused_regagg(cnt.shape, cnt.args, find_static_shape(cnt.shape), host.data()); return host[0]; };
  bool sized_by_sample = false;
  if (!done && B->height > 0) {
    static const bool no_sample = getenv("PLX_JOIN_COUNT_SAMPLE") && getenv("PLX_JOIN_COUNT_SAMPLE")[0] == '0';
    if (B->height >= ((int64_t)1 << 24) && !no_sample) {
      constexpr int kCountBlocks = 4;                  // every block is a launch + a host round trip (~40 us)
      const int64_t per = (int64_t)1 << 18, stride = (B->height / kCountBlocks) & ~(int64_t)127;
      uint64_t hits = 0, seen = 0;
      for (int b = 0; b < kCountBlocks; b++) {
        const int64_t row0 = (int64_t)b * stride, rows_b = std::min<int64_t>(per, B->height - row0);
        if (rows_b <= 0) continue;
        std::vector<uint64_t> host(kMaxAggs, 0);
        k::fused_regagg(cnt.shape, offset_args(cnt.shape, cnt.args, row0, rows_b), -1, host.data());
        hits += host[0]; seen += (uint64_t)rows_b;
      }
      nb = (uint64_t)((double)hits / (double)std::max<uint64_t>(seen, 1) * (double)B->height * 1.25) + 4096;
      sized_by_sample = true;
    } else nb = exact_count();
  }
  // PLX_JOIN_TRACE=1 (debugging aid): every stage of the hash-table pipeline announced on stderr behind a stream synchronisation -- a stage that hangs is the last one named
  static const bool jtrace = getenv("PLX_JOIN_TRACE") && getenv("PLX_JOIN_TRACE")[0] == '1';
#define JTRACE(...) do { if (jtrace) { PLX_HIP(hipStreamSynchronize(stream())); fprintf(stderr, "[plx join] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
  if (!done) {
  Buf keys, flags, acc, links;
  JoinAggTable t{};
  int log2_cap = 4;
  uint64_t cap = 0;
  std::string build_how;
  bool pbuild_off = false;
  k::JoinCells jcells;
  if (known_dups) { B->cols[bki]->repeats_as_build_key = true; if (!multi_ok) return no("build keys are not unique (and a build-side group column is not integer-typed)"); multi = true; }
  bool resized = false;
  for (int attempt = 0; attempt < 4; attempt++) {
    if (multi && !links) links = dev_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(B->height, 1));
    // (the sampled count already carries 25 %: x1.6 keeps the load at or below ~0.6 without doubling a table that x2 would push over the next power of two)
    log2_cap = std::max(4, ceil_log2_u64((uint64_t)((double)std::max<uint64_t>(nb, 1) * (sized_by_sample ? 1.6 : 2.0))));
    cap = 1ull << log2_cap;
    keys = dev_alloc(sizeof(uint64_t) * 2 * (cap + 1)); flags = dev_alloc_zero(32);       // {key, row} slots: kEmptyKey and kNoRow32 are both all-ones
    t.slots = keys->as<unsigned long long>(); t.flags = flags->as<unsigned int>(); t.acc = nullptr;
    t.count = flags->as<unsigned long long>() + 1; t.log2_cap = (uint32_t)log2_cap;
    t.links = multi ? links->as<unsigned long long>() : nullptr;
    JTRACE("build attempt %d multi=%d cap=2^%d nb=%llu", attempt, (int)multi, log2_cap, (unsigned long long)nb);
    // a large build side with (so far) unique keys: partitioned, the table filled window by window from LDS -- no device atomic per row (k::partitioned_join_build)
    t.log2_window = 0;
    build_how = "join_build";
    bool pbuilt = false;
    if (!multi && !pbuild_off && partitioned_build_wanted(B->height, log2_cap)) {
      t.log2_window = kJoinWindowLog2;
      std::string bd;
      pbuilt = k::partitioned_join_build(cb.shape, cb.args, find_static_shape(cb.shape), t, &jcells, &bd);
      if (pbuilt) build_how = bd; else t.log2_window = 0;
    }
    if (!pbuilt) {
      PLX_HIP(hipMemsetAsync(keys->ptr, 0xff, sizeof(uint64_t) * 2 * (cap + 1), stream()));
      k::fused_join_build(cb.shape, cb.args, t, find_static_shape(cb.shape));
    }
    JTRACE("build done");
    uint64_t fl64[2] = {0, 0};
    d2h_sync(fl64, flags->ptr, 16);
    const uint32_t dup = (uint32_t)fl64[0], ovf = (uint32_t)(fl64[0] >> 32);
    if (dup) {
      if (!multi_ok) return no("build keys are not unique (and a build-side group column is not integer-typed)");
      B->cols[bki]->repeats_as_build_key = true;
      multi = true;                                  // build once more, chaining the rows of a key (the table's size stays: it was planned for the rows, not the keys)
      continue;
    }
    if (!resized && (ovf || (sized_by_sample && fl64[1] * 10 > cap * 7))) {      // the sample misjudged (on whichever attempt: the multi-value rebuild keeps the sampled size): once more, from the exact count
      nb = ovf ? exact_count() : fl64[1];
      sized_by_sample = false; resized = true;
      continue;
    }
    if (ovf && pbuilt) { pbuild_off = true; continue; }                             // a window filled up although the table is sized right (keys that crowd one window): the plain build probes the whole table
    PLX_REQUIRE(!ovf, PLX_ERR_OOM, "join build: probe sequence overflow");
    nb = fl64[1];                                                                   // exact from here on
    break;
  }
  if (multi) {
    Buf cflags = dev_alloc_zero(8);
    constexpr unsigned int kMaxChain = 1024;
    JTRACE("canonicalise: %d columns", rep_cols.n);
    k::canonicalise_chains(t, rep_cols, kMaxChain, cflags->as<unsigned int>());
    JTRACE("canonicalise done");
    uint32_t cfl[2] = {0, 0};
    d2h_sync(cfl, cflags->ptr, 8);
    if (cfl[0]) return no("a build key repeats more than 1024 times");
    merged_rows = cfl[1];
  }
  // one cell set per KEY, also in multi-value mode (the groups are expanded from the key's cells by chains_agg_compact): a slot's cells, or -- a windowed table numbers
  // its keys -- the key's
  const int64_t n_cells = t.log2_window ? (int64_t)nb + 1 : (int64_t)cap + 1;
  acc = dev_alloc(sizeof(uint64_t) * (size_t)n_cells * cp.shape.n_aggs);
  t.acc = acc->as<unsigned long long>();
  k::init_agg_cells(acc->as<uint64_t>(), n_cells, cp.shape);
  // Probe.  A table beyond the caches (64 MB) probed by a much longer relation: every probe row would fetch a line across the fabric (SF100 Q3 on keys without
  // a dense range: 3.2e8 random probes of a 400 MB table).  The probe side is partitioned by the key's HASH instead and each partition is tested against an
  // LDS Bloom filter of the table's keys in it (k::partitioned_hash_probe_hits: the radix-partitioned probe of single_keys_inner.rs:11-149 with the partition's
  // filter in LDS); the hash probe below then runs over the surviving candidates only and compares whole keys.
  std::string hprobe_how = "probe_agg";
  bool hprobed = false;
  {
    const int pmode = partitioned_probe_mode();
    if (pmode == 2 || (pmode == 1 && P->height >= ((int64_t)1 << 24) && (cap + 1) * 16 > ((uint64_t)64 << 20) && nb * 16 <= (uint64_t)P->height)) {
      ColumnPtr hits;
      std::string pd;
      if (k::partitioned_hash_probe_hits(cs.shape, cs.args, t, nb, find_static_shape(cs.shape), &hits, &pd)) {
        if (hits->len > 0) {
          Args a2 = cp.args;
          a2.n_rows = hits->len;
          std::vector<ColumnPtr> srcs;
          for (int i = 0; i < cp.shape.n_inputs; i++) srcs.push_back(cp.cols[cp.input_cols[i]]);
          std::vector<ColumnPtr> keep = ops::gather_columns(srcs, hits);
          for (int i = 0; i < cp.shape.n_inputs; i++) { a2.in[i].values = keep[i]->data(); a2.in[i].validity = cp.shape.in_nullable[i] ? keep[i]->valid_words() : nullptr; }
          k::fused_probe_agg(cp.shape, a2, t, probe_static_id);
          PLX_HIP(hipStreamSynchronize(stream()));       // the gathered columns live until the kernels have read them
        }
        hprobe_how = pd + "+gather+probe_agg";
        hprobed = true;
      }
    }
  }
  JTRACE("probe (partitioned: %d)", (int)hprobed);
  if (!hprobed) k::fused_probe_agg(cp.shape, cp.args, t, probe_static_id);
  JTRACE("probe done");
  // ONE compaction pass into buffers sized for every build row (G <= nb): no counting pass over the table first
  const int64_t g1 = std::max<int64_t>((int64_t)nb, 1);
  if (!multi) {
    r.packed_keys = dev_alloc(sizeof(uint64_t) * (size_t)g1);
    r.acc = dev_alloc(sizeof(uint64_t) * (size_t)g1 * r.n_aggs);
    rows->values = dev_alloc(values_bytes(PLX_U32, g1));
  }
  // (measured on SF100 Q3 with hashed keys: listing the touched slots from the candidates' probe -- one returning atomic per row -- costs that probe 0.5 ms, a
  // candidates' filter in front of the LEN cells 0.2 ms; streaming the LEN cells is 0.37 ms)
  if (multi) G = k::chains_agg_compact(t, cp.shape, acc->as<uint64_t>(), len_idx, &rows->values, &r.acc);
  else if (t.log2_window) G = k::cells_agg_compact(jcells, acc->as<uint64_t>(), n_cells, r.n_aggs, len_idx, r.packed_keys->as<uint64_t>(), rows->values->as<uint32_t>(), r.acc->as<uint64_t>());
  else G = k::join_agg_compact(t, r.n_aggs, len_idx, r.packed_keys->as<uint64_t>(), rows->values->as<uint32_t>(), r.acc->as<uint64_t>());
  JTRACE("compacted: %lld groups", (long long)G);
  PLX_REQUIRE(multi || G <= g1, PLX_ERR_INVALID, "join: more groups than build rows");
  r.n_groups = G;
  rows->len = G;
  plan.desc += std::string("FusedJoinGroupBy{build=") + (build_right ? "right" : "left") + " rows=" + std::to_string(nb) + "/" + std::to_string(B->height) + " hash table cap=2^" + std::to_string(log2_cap) + " [" + build_how + "]" +
               (multi ? " multi-value (row chains, a group = a build row; " + std::to_string(merged_rows) + " rows share another row's group), probe rows=" : " unique-keys, probe rows=") + std::to_string(P->height) + ", fused_scan[" + jit::program_mode(probe_static_id, cp.args.n_rows) + "]+" + hprobe_how + ", aggs=" + std::to_string(r.n_aggs) + ", groups=" + std::to_string(G) + "}; ";
  }  // hash-table path
  // ---- output frame: keys, then aggregates
  out = std::make_shared<Frame>();
  out->height = G;
  FinBatch batch{};
  for (auto& gk : gkeys) {
    out->names.push_back(output_name(plan, gk.expr));
    if (gk.is_join_key && multi) out->cols.push_back(ops::gather(B->cols[bki], rows));      // (inner join: the build row's key IS the group's key)
    else if (gk.is_join_key) {
      KeyPart part; part.expr = gk.expr; part.dtype = gk.dtype;
      part.dec.shift = 0; part.dec.mask = ~0ull; part.dec.min = 0; part.dec.null_code = ~0ull; part.dec.dtype = gk.dtype;
      out->cols.push_back(decode_key_column(r, part, 0, batch));
    } else out->cols.push_back(ops::gather(B->cols[gk.build_col], rows));
  }
  std::map<int, ColumnPtr> overrides;
  for (size_t i = 0; i < agg_nodes.size(); i++) overrides[agg_nodes[i]] = finalize_column(r, specs[i], batch);
  k::finalize_batch(r.acc->as<uint64_t>(), r.n_aggs, G, batch);
  Frame gframe; gframe.height = G;
  for (int e : gb.exprs) {
    Evaluated ev = eval(plan, e, gframe, &overrides);
    out->names.push_back(output_name(plan, e));
    out->cols.push_back(broadcast(ev, G));
  }
  if (left_join) {
    // the unmatched left rows: membership bitmap of the build keys that pass the build side's predicate, then a fused group-by over the left table
    Buf bits = dev_alloc_zero(sizeof(uint64_t) * (size_t)(a_range / 64 + 2)), rows_dev = dev_alloc_zero(8);
    if (B->height > 0) {
      BitmapBuild bb; bb.bits = bits->as<unsigned long long>(); bb.count = rows_dev->as<unsigned long long>(); bb.kmin = a_kmn; bb.range = a_range;
      k::fused_bitmap_build(cb.shape, cb.args, bb, find_static_shape(cb.shape));
    }
    ca.args.lut[psemis.size()] = Lut{bits->as<unsigned long long>(), a_range};
    FusedAggResult ar;
    std::string ad;
    try {
      run_fused_groupby(ca, akp, a_len_idx, ar, ad);
    } catch (const Error& e) {
      // a row outside bounds the planner had only ASSUMED for the probe key / value columns (lower_keys(ca) and source_ranges run outside fused_groupby's own handler): forget
      // the guesses, never guess about these columns again, and run the whole join once more from exact statistics -- like fused_groupby does
      bool guessed = false;
      for (auto& col : ca.cols) if (col->range_assumed) { col->range_state = 0; col->range_trusted = true; col->range_assumed = false; col->range_verified = false; col->no_assume = true; std::atomic_store(&col->key_sample, std::shared_ptr<void>()); guessed = true; }
      if (!guessed || e.code != PLX_ERR_INVALID) throw;
      plan.desc += "AssumedBoundsViolated{exact statistics, second run}; ";
      return fused_join_groupby(plan, gb, out, why, shapes_out, compile_only);
    }
    plan.desc += "LeftJoinUnmatched{membership bitmap range=" + std::to_string(a_range) + ", " + akp.note + "FusedFilterGroupBy{" + ad + ", groups=" + std::to_string(ar.n_groups) + "}}; ";
    const int64_t U = ar.n_groups;
    auto tail = std::make_shared<Frame>();
    tail->height = U;
    FinBatch abatch{};
    for (auto& gk : gkeys) {
      tail->names.push_back(output_name(plan, gk.expr));
      if (gk.is_join_key) tail->cols.push_back(decode_key_column(ar, akp.parts[0], 0, abatch));
      else tail->cols.push_back(ops::full_column(B->cols[gk.build_col]->dtype, plx_scalar{0}, false, U));       // no build row: null
    }
    std::map<int, ColumnPtr> aover;
    for (size_t i = 0; i < agg_nodes.size(); i++) aover[agg_nodes[i]] = finalize_column(ar, aspecs[i], abatch);
    if (U > 0) k::finalize_batch(ar.acc->as<uint64_t>(), ar.n_aggs, U, abatch);
    Frame aframe; aframe.height = U;
    for (int e : gb.exprs) {
      Evaluated ev = eval(plan, e, aframe, &aover);
      tail->names.push_back(output_name(plan, e));
      tail->cols.push_back(broadcast(ev, U));
    }
    if (U > 0) {
      for (size_t i = 0; i < out->cols.size(); i++) {
        ColumnPtr b = tail->cols[i];
        if (b->dtype != out->cols[i]->dtype) b = ops::cast(b, out->cols[i]->dtype);
        out->cols[i] = G > 0 ? ops::concat({out->cols[i], b}) : b;
      }
      out->height = G + U;
    }
  }
  return true;
}

#undef JTRACE

// ----------------------------------------------------------------- executors ----
static FramePtr exec_node(Plan& plan, int node_id);

// [Filter]* over a materialised frame without FilterExec's intermediates (filter.rs:94-145: predicate column -> Boolean mask -> one filter call per column): the
// conjunction of the predicates is compiled into a register program whose scan leaves only 16 bytes of ballots + a count per 128 rows (k::fused_ballots), a device
// scan turns the counts into offsets, and ONE kernel then writes the kept rows of every fixed-width column densely and in order (k::compact_by_ballots); validity
// bitmaps and Boolean columns go through the bitmap compaction of kernels_filter.hip with the same selection.  Traffic: predicate inputs once + payload once + output
// once.  `want_rows`: also the kept row indices; rows_only: nothing else.  false: the predicate does not compile (f32 arithmetic, null literals...) -- per-node path.
// member (may be null): one more conjunct -- the row's key column `key_col` is (anti: is NOT) a member of the bitmap `bits` over [kmin, kmin + range): the probe side of a semi /
// anti join whose other side has become a membership bitmap (fused_semi_anti_frame).  A null key is nobody's member: semi drops it, anti keeps it.
struct MemberTest { int key_col; const unsigned long long* bits; int64_t kmin; uint64_t range; bool anti; };
static bool fused_filter_frame(Plan& plan, const std::vector<int>& preds, const FramePtr& src, FramePtr& out, std::string* why, ColumnPtr* want_rows = nullptr, bool rows_only = false,
                               const MemberTest* member = nullptr) {
  Compiler c(plan, *src);
  try {
    int p = -1;
    for (int pe : preds) { int n = c.lower(pe); if (c.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : c.mk(OP_AND, p, n, 'b'); }
    if (member) {
      const int kn = c.load(member->key_col);
      int m = c.ifnull(c.bit_lookup(p < 0 ? kn : c.mask_valid(kn, p), 0, member->kmin), 0);      // (rows the predicates already rejected do not look the bitmap up: OP_MASKV)
      if (member->anti) m = c.mk(OP_NOT, m, m, 'b');
      p = p < 0 ? m : c.mk(OP_AND, p, m, 'b');
    }
    if (p < 0) throw Unsupported("no predicate");
    c.pred = p;
    c.finish();
    if (member) c.args.lut[0] = Lut{member->bits, member->range};
  } catch (const Unsupported& u) { if (why) *why = u.why; return false; }
  const int64_t n = src->height;
  out = std::make_shared<Frame>();
  out->names = src->names;
  auto empty_rows = [] { auto e = std::make_shared<Column>(); e->dtype = PLX_U32; e->len = 0; e->null_count = 0; e->values = dev_alloc(8); return e; };
  if (n == 0) { out->cols = src->cols; out->height = 0; if (want_rows) *want_rows = empty_rows(); return true; }
  PLX_REQUIRE(n < 0xffffffffll || !want_rows, PLX_ERR_UNSUPPORTED, "filter: row indices beyond u32 IdxSize");
  const int64_t n_wt = (n + 127) / 128;
  Buf ballots = dev_alloc(sizeof(uint64_t) * 2 * (size_t)n_wt), counts = dev_alloc(sizeof(uint32_t) * (size_t)n_wt);
  BallotOut bo{ballots->as<unsigned long long>(), counts->as<unsigned int>()};
  const int static_id = find_static_shape(c.shape);
  k::fused_ballots(c.shape, c.args, bo, static_id);
  const k::Selection sel = k::selection_finish(ballots, counts, n);
  const int64_t m = sel.n_out;
  out->height = m;
  ColumnPtr rows;
  if (want_rows) { rows = empty_rows(); rows->len = m; rows->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(m, 1))); *want_rows = rows; }
  int n_moved = 0, n_bitmaps = 0;
  std::vector<ColumnPtr> outs(rows_only ? 0 : src->cols.size());
  // A SPARSE selection (under a sixteenth of the rows: a selective predicate, a semi join against a small side) does not stream every column through the compaction
  // kernel -- that reads all of them whatever is kept -- but takes the kept ROW IDS (one thread per wave tile walks the ballots) and gathers: the lines of the kept rows only.
  static const bool no_sparse = getenv("PLX_FILTER_SPARSE") && getenv("PLX_FILTER_SPARSE")[0] == '0';
  if (!rows_only && !no_sparse && m < n && m * 16 <= n && n < 0xffffffffll) {
    ColumnPtr ids = rows;
    if (!ids) { ids = empty_rows(); ids->len = m; ids->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(m, 1))); }
    k::compact_by_ballots(sel, k::CompactCols{}, ids->values->as<uint32_t>());
    out->cols.assign(src->cols.size(), nullptr);
    for (size_t b0 = 0; b0 < src->cols.size(); b0 += (size_t)k::kGatherMultiMax) {
      std::vector<ColumnPtr> part(src->cols.begin() + b0, src->cols.begin() + std::min(src->cols.size(), b0 + (size_t)k::kGatherMultiMax));
      std::vector<ColumnPtr> got = ops::gather_columns(part, ids);
      for (size_t j = 0; j < got.size(); j++) out->cols[b0 + j] = got[j];
    }
    PLX_HIP(hipStreamSynchronize(stream()));
    plan.desc += "FusedFilter{fused_scan[" + std::string(jit::program_mode(static_id, n)) + "]+ballots -> scan -> row ids -> gather x" + std::to_string(src->cols.size()) + (want_rows ? ", row ids" : "") +
                 ", kept=" + std::to_string(m) + "/" + std::to_string(n) + "}; ";
    return true;
  }
  if (m == n && !rows_only) { out->cols = src->cols; }                      // every row kept (filter/mod.rs:47-49)
  else if (!rows_only) {
    k::CompactCols cc{};
    bool need_plan = false;
    for (size_t i = 0; i < src->cols.size(); i++) {
      const ColumnPtr& col = src->cols[i];
      auto o = std::make_shared<Column>();
      o->dtype = col->dtype; o->len = m;
      const int w = dtype_width(col->dtype);
      o->values = col->dtype == PLX_BOOL ? dev_alloc_zero(bitmap_bytes(m)) : dev_alloc(values_bytes(col->dtype, m));
      if (col->validity) { o->validity = dev_alloc_zero(bitmap_bytes(m)); need_plan = true; } else o->null_count = 0;
      if (col->dtype == PLX_BOOL || !(w == 1 || w == 2 || w == 4 || w == 8)) need_plan = true;
      outs[i] = o;
    }
    // fixed-width values: kCompactMaxCols columns per launch of the all-columns kernel
    for (size_t i = 0; i < src->cols.size();) {
      cc = k::CompactCols{};
      for (; i < src->cols.size() && cc.n_cols < k::kCompactMaxCols; i++) {
        const ColumnPtr& col = src->cols[i];
        const int w = dtype_width(col->dtype);
        if (col->dtype == PLX_BOOL || !(w == 1 || w == 2 || w == 4 || w == 8)) continue;
        cc.in[cc.n_cols] = col->data(); cc.out[cc.n_cols] = outs[i]->values->ptr; cc.width[cc.n_cols] = (uint8_t)w; cc.n_cols++;
      }
      if (cc.n_cols || (rows && !n_moved)) k::compact_by_ballots(sel, cc, rows && !n_moved ? rows->values->as<uint32_t>() : nullptr);
      if (rows && !n_moved) rows = nullptr;
      n_moved += cc.n_cols;
    }
    if (need_plan) {
      Buf mask;
      const k::FilterPlan fp = k::selection_to_plan(sel, &mask);
      for (size_t i = 0; i < src->cols.size(); i++) {
        const ColumnPtr& col = src->cols[i];
        const int w = dtype_width(col->dtype);
        if (col->dtype == PLX_BOOL || !(w == 1 || w == 2 || w == 4 || w == 8)) {
          k::filter_apply(fp, w, col->data(), col->valid_words(), outs[i]->values->ptr, outs[i]->validity ? outs[i]->validity->as<uint64_t>() : nullptr);
          n_bitmaps++;
        } else if (col->validity) { k::filter_apply(fp, 0, col->valid_words(), nullptr, outs[i]->validity->ptr, nullptr); n_bitmaps++; }      // the validity bitmap compacted as a bitmap
      }
      PLX_HIP(hipStreamSynchronize(stream()));       // `mask` and the plan's offsets live until the compactions that read them are done
    }
    out->cols = outs;
  }
  if (rows) k::compact_by_ballots(sel, k::CompactCols{}, rows->values->as<uint32_t>());      // (row ids only)
  PLX_HIP(hipStreamSynchronize(stream()));           // the selection's buffers live until the kernels that read them are done
  plan.desc += "FusedFilter{fused_scan[" + std::string(jit::program_mode(static_id, n)) + "]+ballots -> scan -> " + (rows_only ? std::string("row ids") : "compaction of " + std::to_string(n_moved) +
               " columns in one pass" + (n_bitmaps ? " (+" + std::to_string(n_bitmaps) + " bitmaps through the selection bitmap)" : "") + (want_rows ? ", row ids" : "")) +
               ", kept=" + std::to_string(m) + "/" + std::to_string(n) + "}; ";
  return true;
}

static FramePtr exec_filter(Plan& plan, const IRN& n) {
  if (!(plan.flags & PLX_PLAN_NO_FUSION)) {
    std::vector<int> preds;
    const int src_node = peel_filters(plan, n.input, preds);
    preds.push_back(n.predicate);
    FramePtr src = exec_node(plan, src_node);
    FramePtr out; std::string why;
    if (fused_filter_frame(plan, preds, src, out, &why)) return out;
    plan.desc += "(filter not fused: " + why + ") ";
    plan.memo[src_node] = src;
  }
  FramePtr in = exec_node(plan, n.input);
  Evaluated m = eval(plan, n.predicate, *in, nullptr);
  ColumnPtr mask = broadcast(m, in->height);
  PLX_REQUIRE(mask->dtype == PLX_BOOL, PLX_ERR_INVALID, "filter predicate must be boolean");
  auto pm = ops::prepare_mask(mask);
  auto out = std::make_shared<Frame>();
  out->names = in->names;
  out->height = ops::prepared_rows(*pm);
  for (auto& c : in->cols) out->cols.push_back(ops::filter_prepared(c, *pm));
  plan.desc += "Filter{cmp->bitmap, filter_compact x" + std::to_string(in->cols.size()) + "}; ";
  return out;
}

static FramePtr exec_select(Plan& plan, const IRN& n, bool hstack, FramePtr in_given = nullptr) {
  FramePtr in = in_given ? in_given : exec_node(plan, n.input);
  auto out = std::make_shared<Frame>();
  std::vector<Evaluated> evs;
  bool all_scalar = true;
  for (int e : n.exprs) { evs.push_back(eval(plan, e, *in, nullptr)); all_scalar = all_scalar && evs.back().scalar; }
  if (hstack) {
    *out = *in;
    for (size_t i = 0; i < n.exprs.size(); i++) {
      ColumnPtr c = broadcast(evs[i], in->height);
      std::string name = output_name(plan, n.exprs[i]);
      int at = out->find(name);
      if (at >= 0) out->cols[at] = c; else { out->names.push_back(name); out->cols.push_back(c); }
    }
    plan.desc += "HStack{per-node kernels}; ";
    return out;
  }
  out->height = all_scalar ? 1 : in->height;
  for (size_t i = 0; i < n.exprs.size(); i++) {
    out->names.push_back(output_name(plan, n.exprs[i]));
    out->cols.push_back(all_scalar ? evs[i].col : broadcast(evs[i], in->height));
  }
  plan.desc += "Select{per-node kernels}; ";
  return out;
}

static FramePtr exec_groupby_materialised(Plan& plan, const IRN& n, const FramePtr& in) {
  // evaluate key and aggregation-input expressions as columns, then the generic grouped aggregation
  std::vector<ColumnPtr> keys, values;
  std::vector<int> aggs, agg_nodes;
  for (int e : n.keys) keys.push_back(broadcast(eval(plan, e, *in, nullptr), in->height));
  for (int e : n.exprs) collect_aggs(plan, e, agg_nodes);
  for (int a : agg_nodes) {
    const AE& x = plan.ae[a];
    if (x.kind == PLX_AE_LEN) { aggs.push_back(PLX_AGG_LEN); values.push_back(nullptr); }
    else { aggs.push_back(x.op); values.push_back(broadcast(eval(plan, x.lhs, *in, nullptr), in->height)); }
  }
  std::vector<ColumnPtr> okeys, oaggs;
  std::string d;
  groupby_columns(keys, values, aggs, n.maintain_order != 0, okeys, oaggs, &d);
  plan.desc += "GroupBy{materialised inputs; " + d + "}; ";
  auto out = std::make_shared<Frame>();
  out->height = okeys.empty() ? 0 : okeys[0]->len;
  for (size_t i = 0; i < n.keys.size(); i++) { out->names.push_back(output_name(plan, n.keys[i])); out->cols.push_back(okeys[i]); }
  std::map<int, ColumnPtr> overrides;
  for (size_t i = 0; i < agg_nodes.size(); i++) overrides[agg_nodes[i]] = oaggs[i];
  Frame gframe; gframe.height = out->height;
  for (int e : n.exprs) {
    Evaluated ev = eval(plan, e, gframe, &overrides);
    out->names.push_back(output_name(plan, e));
    out->cols.push_back(broadcast(ev, out->height));
  }
  return out;
}


// ------------------------------------------------ fused Join -> frame pipeline ----
// Join(inner | left; one plain integer key pair) of two [Filter]* inputs that RETURNS A FRAME (the reference: JoinExec polars-mem-engine/src/executors/join.rs:41-121 ->
// _inner_join_from_series / _left_join_from_series polars-ops/src/frame/join/mod.rs:564-652 -> hash_join_tuples_inner single_keys_inner.rs:40-149 -> gathers).
// The filters never materialise their frames: the build side's predicate is fused into the build scan (k::fused_join_build: 16-byte {key, row} slots, chains of
// rows for duplicate keys), the probe side's into the scatter of the partitioned probe (k::partitioned_hash_probe_hits: rows radix-partitioned by the key's hash,
// each partition tested against an LDS filter of its region of the table) or -- joins that keep most probe rows, left joins, small tables -- into the one-pass
// row-id compaction (k::fused_filter); join::join_pairs then looks every CANDIDATE up once and lays the (probe row, build row) pairs out, and the payload columns
// `want` names (null: all) are gathered at them, several columns per launch.  PLX_JOIN_MATERIALISE: 0 = never, 2 = any size (tests), default from 2^24 probe rows.
static int join_materialise_mode() { const char* e = getenv("PLX_JOIN_MATERIALISE"); return e && e[0] == '0' ? 0 : e && e[0] == '2' ? 2 : 1; }
static bool fused_join_frame(Plan& plan, const IRN& jn, const std::set<std::string>* want, FramePtr& out, std::string* why, uint64_t* build_rows_out = nullptr, int* build_side_out = nullptr) {
  auto no = [&](const char* m) { if (why) *why = m; return false; };
  const int mode = join_materialise_mode();
  if (mode == 0) return no("disabled (PLX_JOIN_MATERIALISE=0)");
  if ((jn.how != PLX_JOIN_INNER && jn.how != PLX_JOIN_LEFT) || jn.keys.size() != 1 || jn.keys_right.size() != 1) return no("not a single-key inner or left join");
  const bool left_join = jn.how == PLX_JOIN_LEFT;
  auto plain = [&](int e) -> const AE* { const AE* x = &plan.ae[e]; while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs]; return x->kind == PLX_AE_COLUMN ? x : nullptr; };
  const AE* lkx = plain(jn.keys[0]);
  const AE* rkx = plain(jn.keys_right[0]);
  if (!lkx || !rkx) return no("join keys are expressions");
  std::vector<int> lpreds, rpreds;
  const int lsrc = peel_filters(plan, jn.input, lpreds), rsrc = peel_filters(plan, jn.input_right, rpreds);
  // (cheap checks on scans first: a join this path does not take must not have executed its inputs twice)
  auto height_of = [&](int node) -> int64_t { return plan.ir[node].kind == PLX_IR_SCAN ? get_frame(plan.ir[node].frame)->height : -1; };
  if (mode == 1 && height_of(lsrc) >= 0 && height_of(rsrc) >= 0 && std::max(height_of(lsrc), height_of(rsrc)) < ((int64_t)1 << 24)) return no("small inputs");
  FramePtr L = exec_node(plan, lsrc), R = exec_node(plan, rsrc);
  plan.memo[lsrc] = L; plan.memo[rsrc] = R;
  const int lki = L->find(lkx->name), rki = R->find(rkx->name);
  if (lki < 0 || rki < 0) return no("join key column not found");
  const int kdt = L->cols[lki]->dtype;
  if (kdt != R->cols[rki]->dtype || !dtype_is_int(kdt)) return no("join key is not an integer column pair of one dtype");
  if (L->height >= 0xffffffffll || R->height >= 0xffffffffll) return no("side exceeds u32 row indices");
  const bool build_right = left_join || L->height > R->height;       // det_hash_prone_order (hash_join/mod.rs:41-50); a left join probes with its left table
  const FramePtr& B = build_right ? R : L;
  const FramePtr& P = build_right ? L : R;
  if (mode == 1 && P->height < ((int64_t)1 << 24)) return no("small inputs");
  const int bki = build_right ? rki : lki, pki = build_right ? lki : rki;
  const std::vector<int>& bpreds = build_right ? rpreds : lpreds;
  const std::vector<int>& ppreds = build_right ? lpreds : rpreds;
  // output columns (_finish_join, general.rs:17-49): left columns, then right columns except the coalesced right key; clashes get the suffix
  struct OutCol { std::string name; int side; int idx; };     // side 0 = left, 1 = right
  std::vector<OutCol> outs;
  {
    std::set<std::string> seen;
    for (size_t i = 0; i < L->names.size(); i++) { outs.push_back({L->names[i], 0, (int)i}); seen.insert(L->names[i]); }
    for (size_t i = 0; i < R->names.size(); i++) {
      if ((int)i == rki) continue;
      std::string name = R->names[i];
      if (seen.count(name)) name += jn.suffix;
      if (seen.count(name)) return no("duplicate output column name");
      seen.insert(name);
      outs.push_back({name, 1, (int)i});
    }
  }
  Compiler cnt(plan, *B), cb(plan, *B), cs(plan, *P);
  try {
    auto and_preds = [&](Compiler& c, const std::vector<int>& preds) {
      int p = -1;
      for (int pe : preds) { int n = c.lower(pe); if (c.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : c.mk(OP_AND, p, n, 'b'); }
      return p;
    };
    cnt.pred = and_preds(cnt, bpreds);
    const int bk_cnt = cnt.load(bki);
    cnt.add_agg(cnt.nodes[bk_cnt].nullable ? AGG_COUNT : AGG_LEN, cnt.nodes[bk_cnt].nullable ? bk_cnt : -1);
    cnt.finish();
    cb.pred = and_preds(cb, bpreds);
    cb.key = cb.load(bki);
    cb.finish();
    cs.pred = and_preds(cs, ppreds);
    cs.key = cs.load(pki);
    cs.add_agg(AGG_FIRST_ROW, -1);
    cs.finish();
  } catch (const Unsupported& u) { if (why) *why = u.why; return false; }
  // ---- direct-address table when the build keys have a dense range (cached column statistics): a bitmap over the key range + a rank per word + slot -> build row.  Probe
  // keys in key order walk the bitmap out of the L2 (the hits come out of the probe scan itself, in ballot form: DirectHitsSink); keys in no order are partitioned first
  // (k::partitioned_probe_hits).  Two pairs on one bit = duplicate build keys -> the hash-table pipeline below (chains).
  uint64_t nb = 0;
  bool done_direct = false, multi = B->height > 0 && B->cols[bki]->repeats_as_build_key;
  ColumnPtr pidx, bidx;
  std::string cand_how, pd, table_how;
  const int pmode = partitioned_probe_mode();
  int64_t kmn = 0, kmx = 0;
  if (!(plan.flags & PLX_PLAN_NO_DIRECT_JOIN) && !multi && B->height > 0 && kdt != PLX_U64 && ops::int_range(B->cols[bki], &kmn, &kmx)) {
    const unsigned __int128 range128 = (unsigned __int128)((__int128)kmx - (__int128)kmn) + 1;
    const uint64_t ord_cap = (uint64_t)B->height + (uint64_t)k::scan_waves(B->height) * 1024 + 1024;
    if (range128 <= ((unsigned __int128)1 << 34) && range128 <= (unsigned __int128)B->height * 256 && ord_cap < 0xfffffff0ull) {
      const uint64_t range = (uint64_t)range128;
      const size_t n_words = (size_t)(range / 512 + 1) * 8;
      Buf bits = dev_alloc_zero(sizeof(uint64_t) * n_words), rank = dev_alloc(sizeof(uint32_t) * n_words);
      Buf okey = dev_alloc(sizeof(uint64_t) * ord_cap), orow = dev_alloc(sizeof(uint32_t) * ord_cap), used = dev_alloc_zero(sizeof(uint32_t) * (ord_cap / 1024 + 2));
      Buf meta = dev_alloc_zero(32);
      DirectJoinTable dt{}; dt.bits = bits->as<unsigned long long>(); dt.rank = rank->as<unsigned int>(); dt.ord_key = okey->as<unsigned long long>(); dt.ord_row = orow->as<unsigned int>();
      dt.chunk_used = used->as<unsigned int>(); dt.counter = meta->as<unsigned int>(); dt.flags = meta->as<unsigned int>() + 2; dt.acc = nullptr; dt.kmin = kmn; dt.range = range; dt.n_ord = (unsigned int)ord_cap;
      k::fused_direct_build(cb.shape, cb.args, dt, find_static_shape(cb.shape));
      uint64_t pairs = 0;
      nb = k::direct_rank(dt, rank->as<uint32_t>(), (int64_t)ord_cap, meta->as<uint64_t>() + 2, &pairs);
      uint32_t m4[4] = {0, 0, 0, 0};
      d2h_sync(m4, meta->ptr, 16);
      PLX_REQUIRE(!m4[3], PLX_ERR_INVALID, "direct join build: ordinal overflow");
      if (pairs != nb) { multi = true; B->cols[bki]->repeats_as_build_key = true; }
      else {
        Buf slot_row = dev_alloc(sizeof(uint32_t) * (size_t)std::max<uint64_t>(nb, 1));
        k::direct_slot_rows(dt, (int64_t)std::min<uint64_t>(m4[0], ord_cap), slot_row->as<uint32_t>());
        ColumnPtr cand;
        if (!left_join) {
          const ColumnPtr& pk = P->cols[pki];
          if (pmode == 2 || (pmode == 1 && P->height >= ((int64_t)1 << 24) && range >= ((uint64_t)1 << 28) && nb * 8 <= range)) {
            if (pk->order_state == 0) pk->order_state = k::sample_sortedness(pk) >= 0.9 ? 1 : 2;
            std::string ppd;
            if ((pmode == 2 || pk->order_state == 2) && k::partitioned_probe_hits(cs.shape, cs.args, dt, nb, find_static_shape(cs.shape), &cand, &ppd)) cand_how = ppd;
            else cand = nullptr;
          }
          if (!cand) {
            // the probe scan itself: predicate + bitmap test per row, ballots out; the hit rows' indices from the ballots (every one of them matches)
            const int64_t np = P->height, n_wt = (np + 127) / 128;
            cand = std::make_shared<Column>();
            cand->dtype = PLX_U32; cand->null_count = 0; cand->len = 0; cand->values = dev_alloc(8);
            if (np > 0) {
              Buf ballots = dev_alloc(sizeof(uint64_t) * 2 * (size_t)n_wt), counts = dev_alloc(sizeof(uint32_t) * (size_t)n_wt);
              const int sid = find_static_shape(cs.shape);
              k::fused_direct_hits(cs.shape, cs.args, dt, BallotOut{ballots->as<unsigned long long>(), counts->as<unsigned int>()}, sid);
              const k::Selection sel = k::selection_finish(ballots, counts, np);
              cand->len = sel.n_out;
              cand->values = dev_alloc(values_bytes(PLX_U32, std::max<int64_t>(sel.n_out, 1)));
              k::compact_by_ballots(sel, k::CompactCols{}, cand->values->as<uint32_t>());
              PLX_HIP(hipStreamSynchronize(stream()));
              cand_how = std::string("fused_scan[") + jit::program_mode(sid, np) + "]+direct hits (ballots -> row ids)";
            }
          }
        } else if (!ppreds.empty()) {
          FramePtr none; std::string fwhy;
          const size_t mark = plan.desc.size();
          if (!fused_filter_frame(plan, ppreds, P, none, &fwhy, &cand, true)) { if (why) *why = "probe-side predicate: " + fwhy; return false; }
          cand_how = "probe rows by " + plan.desc.substr(mark);
          plan.desc.resize(mark);
          while (!cand_how.empty() && (cand_how.back() == ' ' || cand_how.back() == ';')) cand_how.pop_back();
        } else cand_how = "every probe row";
        join::join_pairs_direct(jn.how, P->cols[pki], cand, dt, slot_row->as<uint32_t>(), pidx, bidx, &pd);
        PLX_HIP(hipStreamSynchronize(stream()));
        table_how = "direct-address table range=" + std::to_string(range) + " (bitmap + rank + slot rows) unique-keys";
        done_direct = true;
      }
    }
  }
  if (build_rows_out && done_direct) *build_rows_out = nb;
  if (build_side_out) *build_side_out = build_right ? 1 : 0;
  if (!done_direct) {
  // ---- build (the hash-table pipeline of fused_join_groupby: sized from a strided sample of the count program, rebuilt once if the sample misjudged; duplicate keys -> chains)
  auto exact_count = [&]() -> uint64_t { std::vector<uint64_t> host(kMaxAggs, 0); k::fused_regagg(cnt.shape, cnt.args, find_static_shape(cnt.shape), host.data()); return host[0]; };
  bool sized_by_sample = false;
  if (B->height > 0) {
    if (B->height >= ((int64_t)1 << 24)) {
      constexpr int kCountBlocks = 4;
      const int64_t per = (int64_t)1 << 18, stride = (B->height / kCountBlocks) & ~(int64_t)127;
      uint64_t hits = 0, seen = 0;
      for (int b = 0; b < kCountBlocks; b++) {
        const int64_t row0 = (int64_t)b * stride, rows_b = std::min<int64_t>(per, B->height - row0);
        if (rows_b <= 0) continue;
        std::vector<uint64_t> host(kMaxAggs, 0);
        k::fused_regagg(cnt.shape, offset_args(cnt.shape, cnt.args, row0, rows_b), -1, host.data());
        hits += host[0]; seen += (uint64_t)rows_b;
      }
      nb = (uint64_t)((double)hits / (double)std::max<uint64_t>(seen, 1) * (double)B->height * 1.25) + 4096;
      sized_by_sample = true;
    } else nb = exact_count();
  }
  Buf keys, flags, links;
  JoinAggTable t{};
  int log2_cap = 4;
  uint64_t cap = 0;
  bool resized = false, pbuild_off = false;
  std::string build_how;
  for (int attempt = 0; attempt < 4; attempt++) {
    if (multi && !links) links = dev_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(B->height, 1));
    log2_cap = std::max(8, ceil_log2_u64((uint64_t)((double)std::max<uint64_t>(nb, 1) * (sized_by_sample ? 1.6 : 2.0))));
    cap = 1ull << log2_cap;
    keys = dev_alloc(sizeof(uint64_t) * 2 * (cap + 1)); flags = dev_alloc_zero(32);
    t.slots = keys->as<unsigned long long>(); t.flags = flags->as<unsigned int>(); t.acc = nullptr;
    t.count = flags->as<unsigned long long>() + 1; t.log2_cap = (uint32_t)log2_cap;
    t.links = multi ? links->as<unsigned long long>() : nullptr;
    // (as in fused_join_groupby: a large build side with unique keys is binned into the table's windows and filled from LDS)
    t.log2_window = 0;
    build_how.clear();
    bool pbuilt = false;
    if (!multi && !pbuild_off && partitioned_build_wanted(B->height, log2_cap)) {
      t.log2_window = kJoinWindowLog2;
      pbuilt = k::partitioned_join_build(cb.shape, cb.args, find_static_shape(cb.shape), t, nullptr, &build_how);
      if (!pbuilt) t.log2_window = 0;
    }
    if (!pbuilt) {
      PLX_HIP(hipMemsetAsync(keys->ptr, 0xff, sizeof(uint64_t) * 2 * (cap + 1), stream()));
      k::fused_join_build(cb.shape, cb.args, t, find_static_shape(cb.shape));
    }
    uint64_t fl64[2] = {0, 0};
    d2h_sync(fl64, flags->ptr, 16);
    const uint32_t dup = (uint32_t)fl64[0], ovf = (uint32_t)(fl64[0] >> 32);
    if (dup && !multi) { B->cols[bki]->repeats_as_build_key = true; multi = true; continue; }      // build once more, chaining the rows of a key
    if (!resized && (ovf || fl64[1] * 10 > cap * 7)) { nb = ovf ? exact_count() : fl64[1]; sized_by_sample = false; resized = true; continue; }      // the sample misjudged: once more, from the exact count
    if (ovf && pbuilt) { pbuild_off = true; continue; }
    PLX_REQUIRE(!ovf, PLX_ERR_OOM, "join build: probe sequence overflow");
    nb = fl64[1];
    break;
  }
  if (build_rows_out) *build_rows_out = nb;
  // ---- candidates
  ColumnPtr cand;
  if (!left_join && (pmode == 2 || (pmode == 1 && P->height >= ((int64_t)1 << 24) && (cap + 1) * 16 > ((uint64_t)64 << 20) && nb * 16 <= (uint64_t)P->height))) {
    std::string pd;
    if (k::partitioned_hash_probe_hits(cs.shape, cs.args, t, nb, find_static_shape(cs.shape), &cand, &pd)) cand_how = pd;
    else cand = nullptr;
  }
  if (!cand && !ppreds.empty()) {
    FramePtr none; std::string fwhy;
    const size_t mark = plan.desc.size();
    if (!fused_filter_frame(plan, ppreds, P, none, &fwhy, &cand, true)) { if (why) *why = "probe-side predicate: " + fwhy; return false; }
    cand_how = "probe rows by " + plan.desc.substr(mark);
    plan.desc.resize(mark);
    while (!cand_how.empty() && (cand_how.back() == ' ' || cand_how.back() == ';')) cand_how.pop_back();
  }
  if (!cand && cand_how.empty()) cand_how = "every probe row";
  // ---- pairs
  join::join_pairs(jn.how, P->cols[pki], cand, t, pidx, bidx, &pd);
  PLX_HIP(hipStreamSynchronize(stream()));
  table_how = "hash table cap=2^" + std::to_string(log2_cap) + (build_how.empty() ? "" : " [" + build_how + "]") + (multi ? " multi-value (row chains)" : " unique-keys");
  }  // hash-table pipeline
  // ---- payload: one multi-column gather per side
  const ColumnPtr& lidx = build_right ? pidx : bidx;
  const ColumnPtr& ridx = build_right ? bidx : pidx;
  out = std::make_shared<Frame>();
  out->height = pidx->len;
  std::vector<int> pick[2];
  std::vector<size_t> slot_of[2];
  for (size_t i = 0; i < outs.size(); i++) {
    if (want && !want->count(outs[i].name)) continue;
    out->names.push_back(outs[i].name); out->cols.push_back(nullptr);
    pick[outs[i].side].push_back(outs[i].idx); slot_of[outs[i].side].push_back(out->cols.size() - 1);
  }
  int n_gathered = 0;
  for (int side = 0; side < 2; side++) {
    const FramePtr& F = side == 0 ? L : R;
    const ColumnPtr& idx = side == 0 ? lidx : ridx;
    for (size_t b0 = 0; b0 < pick[side].size(); b0 += (size_t)k::kGatherMultiMax) {
      std::vector<ColumnPtr> srcs;
      for (size_t j = b0; j < std::min(pick[side].size(), b0 + (size_t)k::kGatherMultiMax); j++) srcs.push_back(F->cols[pick[side][j]]);
      std::vector<ColumnPtr> got = ops::gather_columns(srcs, idx);
      for (size_t j = 0; j < got.size(); j++) out->cols[slot_of[side][b0 + j]] = got[j];
      n_gathered += (int)got.size();
    }
  }
  PLX_HIP(hipStreamSynchronize(stream()));
  plan.desc += std::string("FusedJoinFrame{") + (left_join ? "left" : "inner") + ", build=" + (build_right ? "right" : "left") + " rows=" + std::to_string(nb) + "/" + std::to_string(B->height) + " " + table_how +
               ", probe rows=" + std::to_string(P->height) + ", candidates: " + cand_how + ", " + pd + ", gather x" + std::to_string(n_gathered) + "}; ";
  return true;
}

// Join(semi | anti, one integer key pair) over two `[Filter]*` inputs -> the left rows whose key is (not) among the right side's, left columns, left order
// (polars-ops/src/frame/join/hash_join/single_keys_semi_anti.rs: a hash SET of the right keys, one lookup per left row; dispatch_left_right.rs).  Here the right side
// -- its predicate fused into the scan -- becomes a membership BITMAP over its key range (BitmapBuildSink: the sink of the filter joins of §4.1), and the join is a
// FILTER of the left side: its own predicates AND the bitmap test in one predicate program, ballots -> one compaction pass over all left columns (fused_filter_frame).
// No pairs, no gathers.  Needs a right key range a bitmap can cover (<= 2^34 keys and <= 256 x the right rows); otherwise the per-node join runs.
static bool fused_semi_anti_frame(Plan& plan, const IRN& jn, FramePtr& out, std::string* why) {
  auto no = [&](const char* m) { if (why) *why = m; return false; };
  if (join_materialise_mode() == 0) return no("disabled (PLX_JOIN_MATERIALISE=0)");
  if ((jn.how != PLX_JOIN_SEMI && jn.how != PLX_JOIN_ANTI) || jn.keys.size() != 1 || jn.keys_right.size() != 1) return no("not a single-key semi or anti join");
  auto plain = [&](int e) -> const AE* { const AE* x = &plan.ae[e]; while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs]; return x->kind == PLX_AE_COLUMN ? x : nullptr; };
  const AE* lkx = plain(jn.keys[0]);
  const AE* rkx = plain(jn.keys_right[0]);
  if (!lkx || !rkx) return no("join keys are expressions");
  std::vector<int> lpreds, rpreds;
  const int lsrc = peel_filters(plan, jn.input, lpreds), rsrc = peel_filters(plan, jn.input_right, rpreds);
  FramePtr L = exec_node(plan, lsrc), R = exec_node(plan, rsrc);
  plan.memo[lsrc] = L; plan.memo[rsrc] = R;
  const int lki = L->find(lkx->name), rki = R->find(rkx->name);
  if (lki < 0 || rki < 0) return no("join key column not found");
  const int kdt = L->cols[lki]->dtype;
  if (kdt != R->cols[rki]->dtype || !dtype_is_int(kdt) || kdt == PLX_U64) return no("join key is not a signed / narrow integer column pair of one dtype");
  int64_t mn = 0, mx = 0;
  const bool have = R->height > 0 && R->cols[rki]->values && ops::int_range(R->cols[rki], &mn, &mx);
  const unsigned __int128 range128 = have ? (unsigned __int128)((__int128)mx - (__int128)mn) + 1 : 1;
  if (range128 > ((unsigned __int128)1 << 34) || (have && range128 > (unsigned __int128)R->height * 256 + 4096)) return no("right key range too wide for a membership bitmap");
  const uint64_t range = (uint64_t)range128;
  Buf bits = dev_alloc_zero(sizeof(uint64_t) * (size_t)(range / 64 + 2)), rows_dev = dev_alloc_zero(8);
  uint64_t rows_in = 0;
  if (have) {
    Compiler cb(plan, *R);
    try {
      int p = -1;
      for (int pe : rpreds) { int n = cb.lower(pe); if (cb.nodes[n].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? n : cb.mk(OP_AND, p, n, 'b'); }
      cb.pred = p;
      cb.key = cb.load(rki);
      cb.finish();
    } catch (const Unsupported& u) { if (why) *why = "right side: " + u.why; return false; }
    BitmapBuild bb; bb.bits = bits->as<unsigned long long>(); bb.count = rows_dev->as<unsigned long long>(); bb.kmin = mn; bb.range = range;
    k::fused_bitmap_build(cb.shape, cb.args, bb, find_static_shape(cb.shape));
    d2h_sync(&rows_in, rows_dev->ptr, 8);
  }
  const MemberTest mt{lki, bits->as<unsigned long long>(), mn, range, jn.how == PLX_JOIN_ANTI};
  const size_t mark = plan.desc.size();
  std::string fwhy;
  if (!fused_filter_frame(plan, lpreds, L, out, &fwhy, nullptr, false, &mt)) { if (why) *why = "left side: " + fwhy; return false; }
  std::string fd = plan.desc.substr(mark);
  plan.desc.resize(mark);
  while (!fd.empty() && (fd.back() == ' ' || fd.back() == ';')) fd.pop_back();
  plan.desc += std::string("FusedSemiAntiJoin{") + (jn.how == PLX_JOIN_ANTI ? "anti" : "semi") + ", right rows=" + std::to_string(rows_in) + "/" + std::to_string(R->height) + " -> membership bitmap range=" +
               std::to_string(range) + ", left rows=" + std::to_string(L->height) + " filtered by " + fd + "}; ";
  return true;
}

static FramePtr exec_join(Plan& plan, const IRN& n) {
  if (!(plan.flags & PLX_PLAN_NO_FUSION) && (n.how == PLX_JOIN_SEMI || n.how == PLX_JOIN_ANTI)) {
    FramePtr out; std::string why;
    if (fused_semi_anti_frame(plan, n, out, &why)) return out;
    plan.desc += "(semi / anti join not fused: " + why + ") ";
  } else if (!(plan.flags & PLX_PLAN_NO_FUSION)) {
    FramePtr out; std::string why;
    if (fused_join_frame(plan, n, nullptr, out, &why)) return out;
    if (why != "small inputs") plan.desc += "(join not fused: " + why + ") ";
  }
  FramePtr left = exec_node(plan, n.input);
  FramePtr right = exec_node(plan, n.input_right);
  PLX_REQUIRE(!n.keys.empty() && n.keys.size() == n.keys_right.size(), PLX_ERR_INVALID, "join: left_on / right_on length mismatch");
  ColumnPtr lk, rk;
  std::string packed_desc;
  if (n.keys.size() == 1) {
    lk = broadcast(eval(plan, n.keys[0], *left, nullptr), left->height);
    rk = broadcast(eval(plan, n.keys_right[0], *right, nullptr), right->height);
  } else {
    // Multi-column keys: the reference row-encodes them (polars-row via join/mod.rs:367-370).  Integer / boolean
    // keys are instead packed into ONE Int64 using the joint value range of both sides:
    //   packed = sum_j (key_j - min_j) * stride_j,  stride_j = prod_{i>j} (max_i - min_i + 1)
    // A null in any key column makes the packed key null, i.e. the row never matches (nulls_equal = false).
    std::vector<ColumnPtr> lks, rks;
    std::vector<int64_t> mins, spans;
    for (size_t j = 0; j < n.keys.size(); j++) {
      ColumnPtr a = broadcast(eval(plan, n.keys[j], *left, nullptr), left->height);
      ColumnPtr b = broadcast(eval(plan, n.keys_right[j], *right, nullptr), right->height);
      PLX_REQUIRE(a->dtype == b->dtype, PLX_ERR_INVALID, "join keys have different dtypes");
      PLX_REQUIRE(dtype_is_int(a->dtype) || a->dtype == PLX_BOOL, PLX_ERR_UNSUPPORTED, "multi-column join keys must be integer / boolean / dictionary codes on this path");
      if (a->dtype != PLX_I64) { PLX_REQUIRE(a->dtype != PLX_U64, PLX_ERR_UNSUPPORTED, "multi-column join on UInt64 keys"); a = ops::cast(a, PLX_I64); b = ops::cast(b, PLX_I64); }
      int64_t amn = 0, amx = 0, bmn = 0, bmx = 0;
      const bool ha = ops::int_range(a, &amn, &amx), hb = ops::int_range(b, &bmn, &bmx);
      int64_t mn = ha ? amn : bmn, mx = ha ? amx : bmx;
      if (ha && hb) { mn = std::min(amn, bmn); mx = std::max(amx, bmx); }
      if (!ha && !hb) { mn = 0; mx = 0; }
      PLX_REQUIRE((double)mx - (double)mn < 9e18, PLX_ERR_UNSUPPORTED, "multi-column join keys span more than 63 bits");
      lks.push_back(a); rks.push_back(b); mins.push_back(mn); spans.push_back(mx - mn + 1);
    }
    double total = 1;
    for (int64_t sp : spans) total *= (double)sp;
    PLX_REQUIRE(total < 9.0e18, PLX_ERR_UNSUPPORTED, "multi-column join keys do not pack into 63 bits (needs row encoding)");
    auto pack = [&](std::vector<ColumnPtr>& ks) {
      ColumnPtr acc;
      int64_t stride = 1;
      for (size_t j = ks.size(); j-- > 0;) {
        plx_scalar s; s.i = mins[j];
        ColumnPtr t = ops::arith_scalar(PLX_SUB, ks[j], s, false);
        if (stride != 1) { plx_scalar m; m.i = stride; t = ops::arith_scalar(PLX_MUL, t, m, false); }
        acc = acc ? ops::arith(PLX_ADD, acc, t) : t;
        stride *= spans[j];
      }
      return acc;
    };
    lk = pack(lks); rk = pack(rks);
    packed_desc = "packed " + std::to_string(n.keys.size()) + " key columns into Int64; ";
  }
  ColumnPtr li, ri;
  std::string d;
  join::join_indices(n.how, lk, rk, li, ri, &d);
  d = packed_desc + d;
  if (n.how == PLX_JOIN_SEMI || n.how == PLX_JOIN_ANTI) {
    // left columns only, left order (single_keys_semi_anti.rs; _finish_join is not involved)
    plan.desc += "Join{" + d + ", gather x" + std::to_string(left->cols.size()) + "}; ";
    auto out = std::make_shared<Frame>();
    out->height = li->len; out->names = left->names;
    for (auto& c : left->cols) out->cols.push_back(ops::gather(c, li));
    return out;
  }
  plan.desc += "Join{" + d + ", gather x" + std::to_string(left->cols.size() + right->cols.size()) + "}; ";
  // _finish_join (polars-ops/src/frame/join/general.rs:17-49): left columns, then right columns
  // except the right key when it is a plain column coalesced into the left key; name clashes get the suffix.
  auto out = std::make_shared<Frame>();
  out->height = li->len;
  for (size_t i = 0; i < left->cols.size(); i++) { out->names.push_back(left->names[i]); out->cols.push_back(ops::gather(left->cols[i], li)); }
  std::vector<std::string> coalesced;   // right key columns merged into the left key (both sides plain columns)
  for (size_t j = 0; j < n.keys.size(); j++) {
    const AE* rkx = &plan.ae[n.keys_right[j]];
    while (rkx->kind == PLX_AE_ALIAS) rkx = &plan.ae[rkx->lhs];
    const AE* lkx = &plan.ae[n.keys[j]];
    while (lkx->kind == PLX_AE_ALIAS) lkx = &plan.ae[lkx->lhs];
    if (rkx->kind == PLX_AE_COLUMN && lkx->kind == PLX_AE_COLUMN) coalesced.push_back(rkx->name);
  }
  for (size_t i = 0; i < right->cols.size(); i++) {
    if (std::find(coalesced.begin(), coalesced.end(), right->names[i]) != coalesced.end()) continue;  // coalesced key
    std::string name = right->names[i];
    if (out->find(name) >= 0) name += n.suffix;
    out->names.push_back(name);
    out->cols.push_back(ops::gather(right->cols[i], ri));
  }
  return out;
}

// IR::Sort (+ a Slice directly above it: only the first offset+len rows of the order are produced, top-k)
static FramePtr exec_sort(Plan& plan, const IRN& n, int64_t limit) {
  FramePtr in = exec_node(plan, n.input);
  PLX_REQUIRE(!n.keys.empty(), PLX_ERR_INVALID, "sort needs at least one key");
  std::vector<sort::SortKey> keys;
  for (size_t j = 0; j < n.keys.size(); j++) {
    sort::SortKey sk;
    sk.col = broadcast(eval(plan, n.keys[j], *in, nullptr), in->height);
    sk.descending = n.sort_descending.size() > j && n.sort_descending[j];
    sk.nulls_last = n.sort_nulls_last.size() > j && n.sort_nulls_last[j];
    keys.push_back(sk);
  }
  std::string d;
  ColumnPtr idx = sort::sort_indices(keys, limit, &d);
  auto out = std::make_shared<Frame>();
  out->names = in->names; out->height = idx->len;
  for (auto& c : in->cols) out->cols.push_back(ops::gather(c, idx));
  plan.desc += "Sort{" + d + ", gather x" + std::to_string(in->cols.size()) + "}; ";
  return out;
}

static FramePtr exec_slice(Plan& plan, const IRN& n) {
  PLX_REQUIRE(n.slice_len >= 0, PLX_ERR_INVALID, "slice length must be non-negative");
  FramePtr in;
  const IRN* src = n.input >= 0 ? &plan.ir[n.input] : nullptr;
  PLX_REQUIRE(src, PLX_ERR_INVALID, "slice without an input");
  if (src->kind == PLX_IR_SORT && n.slice_offset >= 0 && n.slice_offset <= (int64_t)1 << 40 && n.slice_len <= (int64_t)1 << 40) {
    check_cancel();
    in = exec_sort(plan, *src, n.slice_offset + n.slice_len);
  } else in = exec_node(plan, n.input);
  // slice_offsets (polars-core/src/utils/mod.rs:340-358): negative offsets count from the end, both ends clamped
  const int64_t h = in->height;
  int64_t start = n.slice_offset < 0 ? n.slice_offset + h : n.slice_offset;
  int64_t stop = start + n.slice_len;   // offsets are far below the i64 range here (h < 2^32, len checked by the caller)
  start = std::min(std::max<int64_t>(start, 0), h); stop = std::min(std::max<int64_t>(stop, 0), h);
  if (start == 0 && stop == h) { plan.desc += "Slice{no-op}; "; return in; }
  auto out = std::make_shared<Frame>();
  out->names = in->names; out->height = stop - start;
  for (auto& c : in->cols) out->cols.push_back(ops::slice_copy(c, start, stop - start));
  plan.desc += "Slice{" + std::to_string(start) + ", " + std::to_string(stop - start) + "}; ";
  return out;
}

static FramePtr exec_node(Plan& plan, int node_id) {
  check_cancel();
  PLX_REQUIRE(node_id >= 0 && node_id < (int)plan.ir.size(), PLX_ERR_INVALID, "bad IR node index");
  const IRN& n = plan.ir[node_id];
  const bool fuse = !(plan.flags & PLX_PLAN_NO_FUSION);
  { auto it = plan.memo.find(node_id); if (it != plan.memo.end()) return it->second; }
  switch (n.kind) {
    case PLX_IR_SCAN: return get_frame(n.frame);
    case PLX_IR_FILTER: return exec_filter(plan, n);
    case PLX_IR_HSTACK: return exec_select(plan, n, true);
    case PLX_IR_SELECT: {
      if (fuse && n.input >= 0 && plan.ir[n.input].kind == PLX_IR_JOIN) {
        // projection pushed into the join: only the columns the select reads are gathered at the pairs
        std::set<std::string> want;
        for (int e : n.exprs) collect_columns(plan, e, want);
        FramePtr j; std::string why;
        if (fused_join_frame(plan, plan.ir[n.input], &want, j, &why)) return exec_select(plan, n, false, j);
        if (why != "small inputs") plan.desc += "(join not fused: " + why + ") ";
      }
      if (fuse) {
        std::vector<int> preds;
        int src_node = peel_filters(plan, n.input, preds);
        bool aggs_only = !n.exprs.empty();
        for (int e : n.exprs) aggs_only = aggs_only && contains_agg(plan, e) && !contains_column_outside_agg(plan, e);
        if (aggs_only) {
          FramePtr src = exec_node(plan, src_node);
          FramePtr out; std::string why;
          if (fused_select(plan, n, preds, src, out, nullptr, nullptr, &why, false)) return out;
          plan.desc += "(not fused: " + why + ") ";
          plan.memo[src_node] = src;      // the per-node path below runs the same subtree: once is enough
        }
      }
      return exec_select(plan, n, false);
    }
    case PLX_IR_GROUPBY: {
      if (fuse && n.input >= 0 && plan.ir[n.input].kind == PLX_IR_JOIN) {
        FramePtr out; std::string why;
        if (fused_join_groupby(plan, n, out, &why)) return out;
        // The in-place form needs a group to be a build row and the aggregates to read the probe side.  Anything else -- aggregates over build-side columns
        // (sum(l_quantity * ps_supplycost)), group keys without the join key or from the probe side -- takes the PAIR form: the fused join -> frame pipeline
        // restricted to the columns the group-by reads (filters fused into build scan and candidate selection, pairs, one gather per side), then the ordinary fused
        // group-by over those joined columns.  The reference has no such restriction either (JoinExec -> GroupByExec, executors/join.rs:41-121, group_by.rs:60-98).
        std::set<std::string> want;
        for (int e : n.keys) collect_columns(plan, e, want);
        for (int e : n.exprs) collect_columns(plan, e, want);
        const size_t mark = plan.desc.size();
        FramePtr joined; std::string why2;
        uint64_t build_rows = 0;
        int build_side = 0;
        if (fused_join_frame(plan, plan.ir[n.input], &want, joined, &why2, &build_rows, &build_side)) {
          std::vector<int> no_preds;
          std::string why3;
          const std::string jd = plan.desc.substr(mark);
          plan.desc.resize(mark);
          const size_t mark2 = plan.desc.size();
          // group keys that are all functions of the build row (the join key or plain build-side columns): at most one group per surviving build row
          {
            const IRN& jn = plan.ir[n.input];
            FramePtr bf = exec_node(plan, peel_filters_node(plan, build_side ? jn.input_right : jn.input));
            bool of_build = !n.keys.empty();
            for (int e : n.keys) {
              const AE* x = &plan.ae[e];
              while (x->kind == PLX_AE_ALIAS) x = &plan.ae[x->lhs];
              const AE* lk = &plan.ae[jn.keys[0]];
              while (lk->kind == PLX_AE_ALIAS) lk = &plan.ae[lk->lhs];
              of_build = of_build && x->kind == PLX_AE_COLUMN && (x->name == lk->name || bf->find(x->name) >= 0 || (x->name.size() > jn.suffix.size() && bf->find(x->name.substr(0, x->name.size() - jn.suffix.size())) >= 0));
            }
            plan.group_hint = of_build ? (double)std::max<uint64_t>(build_rows, 1) : 0.0;
          }
          const bool gb_fused = fused_groupby(plan, n, no_preds, joined, out, nullptr, nullptr, &why3, false);
          plan.group_hint = 0;
          if (!gb_fused) out = exec_groupby_materialised(plan, n, joined);
          const std::string gd = plan.desc.substr(mark2);
          plan.desc.resize(mark2);
          plan.desc += "FusedJoinGroupBy{pair form (in-place form: " + why + "): " + jd + "then " + gd + "}; ";
          return out;
        }
        plan.desc += "(join+group_by not fused: " + why + (why2 == "small inputs" ? "" : "; pair form: " + why2) + ") ";
      }
      if (fuse) {
        std::vector<int> preds;
        int src_node = peel_filters(plan, n.input, preds);
        FramePtr src = exec_node(plan, src_node);
        FramePtr out; std::string why;
        if (fused_groupby(plan, n, preds, src, out, nullptr, nullptr, &why, false)) return out;
        plan.desc += "(not fused: " + why + ") ";
        plan.memo[src_node] = src;
      }
      FramePtr in = exec_node(plan, n.input);
      return exec_groupby_materialised(plan, n, in);
    }
    case PLX_IR_JOIN: return exec_join(plan, n);
    case PLX_IR_SORT: return exec_sort(plan, n, -1);
    case PLX_IR_SLICE: return exec_slice(plan, n);
    default: fail(PLX_ERR_UNSUPPORTED, "IR node kind " + std::to_string(n.kind) + " is outside the hot path (run it on the CPU engine)");
  }
}

FramePtr execute(Plan& plan, int root) {
  plan.desc.clear();
  return exec_node(plan, root);
}

bool describe_join_fusion(Plan& plan, int root, std::vector<Shape>* shapes, std::string* why_not) {
  const IRN& n = plan.ir.at(root);
  FramePtr out;
  return fused_join_groupby(plan, n, out, why_not, shapes, true);
}

bool describe_filter_fusion(Plan& plan, int root, Shape* shape, std::string* why_not) {
  const IRN& n = plan.ir.at(root);
  PLX_REQUIRE(n.kind == PLX_IR_FILTER, PLX_ERR_INVALID, "describe_filter_fusion: root is not a Filter");
  std::vector<int> preds;
  const int src_node = peel_filters(plan, n.input, preds);
  preds.push_back(n.predicate);
  PLX_REQUIRE(plan.ir[src_node].kind == PLX_IR_SCAN, PLX_ERR_UNSUPPORTED, "describe_filter_fusion: source must be a scan");
  FramePtr src = get_frame(plan.ir[src_node].frame);
  Compiler c(plan, *src);
  try {
    int p = -1;
    for (int pe : preds) { int nn = c.lower(pe); if (c.nodes[nn].ty != 'b') throw Unsupported("predicate is not boolean"); p = p < 0 ? nn : c.mk(OP_AND, p, nn, 'b'); }
    c.pred = p;
    c.finish();
  } catch (const Unsupported& u) { if (why_not) *why_not = u.why; return false; }
  if (shape) *shape = c.shape;
  return true;
}

bool describe_fusion(Plan& plan, int root, Shape* shape, int* static_id, std::string* why_not) {
  const IRN& n = plan.ir.at(root);
  std::vector<int> preds;
  int src_node = peel_filters(plan, n.input, preds);
  PLX_REQUIRE(plan.ir[src_node].kind == PLX_IR_SCAN, PLX_ERR_UNSUPPORTED, "describe_fusion: source must be a scan");
  FramePtr src = get_frame(plan.ir[src_node].frame);
  FramePtr out;
  if (n.kind == PLX_IR_SELECT) return fused_select(plan, n, preds, src, out, shape, static_id, why_not, true);
  if (n.kind == PLX_IR_GROUPBY) return fused_groupby(plan, n, preds, src, out, shape, static_id, why_not, true);
  if (why_not) *why_not = "root is neither Select nor GroupBy";
  return false;
}

bool dump_program_json(Plan& plan, int root, std::string* json, std::string* why_not) {
  ProgramDump d;
  t_program_dump = &d;
  bool ok = false;
  try {
    const IRN& n = plan.ir.at(root);
    if (n.kind == PLX_IR_GROUPBY && n.input >= 0 && plan.ir[n.input].kind == PLX_IR_JOIN) ok = describe_join_fusion(plan, root, nullptr, why_not);
    else ok = describe_fusion(plan, root, nullptr, nullptr, why_not);
  } catch (...) { t_program_dump = nullptr; throw; }
  t_program_dump = nullptr;
  if (ok && json) *json = d.json;
  return ok && !d.json.empty();
}

// generic grouped aggregation over already materialised columns: builds a LOAD-only
// program and reuses the fused table kernels.
void groupby_columns(const std::vector<ColumnPtr>& keys, const std::vector<ColumnPtr>& values, const std::vector<int>& aggs, bool maintain_order,
                     std::vector<ColumnPtr>& out_keys, std::vector<ColumnPtr>& out_aggs, std::string* desc) {
  PLX_REQUIRE(!keys.empty(), PLX_ERR_INVALID, "group_by needs at least one key");
  PLX_REQUIRE(values.size() == aggs.size(), PLX_ERR_INVALID, "group_by: values/aggs length mismatch");
  // synthesise a plan: frame of k0..kn, v0..vm; GroupBy over a scan
  auto f = std::make_shared<Frame>();
  f->height = keys[0]->len;
  Plan p;
  IRN scan; scan.kind = PLX_IR_SCAN; p.ir.push_back(scan);
  IRN gb; gb.kind = PLX_IR_GROUPBY; gb.input = 0; gb.maintain_order = maintain_order ? 1 : 0;
  for (size_t i = 0; i < keys.size(); i++) {
    PLX_REQUIRE(keys[i]->len == f->height, PLX_ERR_SHAPE, "group_by: key length mismatch");
    f->names.push_back("__k" + std::to_string(i)); f->cols.push_back(keys[i]);
    AE c; c.kind = PLX_AE_COLUMN; c.name = f->names.back(); p.ae.push_back(c);
    gb.keys.push_back((int)p.ae.size() - 1);
  }
  for (size_t i = 0; i < values.size(); i++) {
    AE a;
    if (aggs[i] == PLX_AGG_LEN || !values[i]) { a.kind = PLX_AE_LEN; }
    else {
      PLX_REQUIRE(values[i]->len == f->height, PLX_ERR_SHAPE, "group_by: value length mismatch");
      f->names.push_back("__v" + std::to_string(i)); f->cols.push_back(values[i]);
      AE c; c.kind = PLX_AE_COLUMN; c.name = f->names.back(); p.ae.push_back(c);
      a.kind = PLX_AE_AGG; a.op = aggs[i]; a.lhs = (int)p.ae.size() - 1;
    }
    p.ae.push_back(a);
    AE al; al.kind = PLX_AE_ALIAS; al.lhs = (int)p.ae.size() - 1; al.name = "__a" + std::to_string(i); p.ae.push_back(al);
    gb.exprs.push_back((int)p.ae.size() - 1);
  }
  p.ir.push_back(gb);
  FramePtr out; std::string why;
  std::vector<int> preds;
  if (!fused_groupby(p, p.ir[1], preds, f, out, nullptr, nullptr, &why, false)) {
    // f32 / bool value columns: aggregate them through an f64 / integer view
    std::vector<ColumnPtr> v2 = values; bool changed = false;
    for (size_t i = 0; i < v2.size(); i++) if (v2[i] && v2[i]->dtype == PLX_F32) { v2[i] = ops::cast(v2[i], PLX_F64); changed = true; }
    std::vector<ColumnPtr> k2 = keys;
    for (auto& kc : k2) if (kc->dtype == PLX_F32) { kc = ops::cast(kc, PLX_F64); changed = true; }
    if (!changed) fail(PLX_ERR_UNSUPPORTED, "group_by: " + why);
    std::vector<ColumnPtr> ok, oa;
    groupby_columns(k2, v2, aggs, maintain_order, ok, oa, desc);
    for (size_t i = 0; i < keys.size(); i++) out_keys.push_back(keys[i]->dtype == PLX_F32 ? ops::cast(ok[i], PLX_F32) : ok[i]);
    for (size_t i = 0; i < values.size(); i++) {
      bool was_f32 = values[i] && values[i]->dtype == PLX_F32;
      // f32 sums / means / min / max report f32 (reduce/mean.rs:29-80)
      out_aggs.push_back(was_f32 && oa[i]->dtype == PLX_F64 && aggs[i] != PLX_AGG_COUNT ? ops::cast(oa[i], PLX_F32) : oa[i]);
    }
    return;
  }
  if (desc) *desc = p.desc;
  for (size_t i = 0; i < keys.size(); i++) out_keys.push_back(out->cols[i]);
  for (size_t i = 0; i < values.size(); i++) out_aggs.push_back(out->cols[keys.size() + i]);
}

}  // namespace engine
}  // namespace plx
