// fused_device.hpp -- device side of the fused scan kernels, shared by kernels_fused.hip and
// kernels_partition.hip: the VGPR register file, column loads, the expression-program interpreter
// (AOT-foldable), aggregate combine / identity / per-row value rules and the atomic cell updates.
// See kernels_fused.hip for the kernel structure and the reference mapping.
#pragma once
#include "dev.hpp"
#include "fused.hpp"
#include "fused_shapes.hpp"
#include "kconfig.hpp"

namespace plx {
namespace k {

using namespace dev;
using namespace fused;

typedef unsigned long long u64x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));

// Register file of the expression program: kSlots 64-bit values per row + a validity bit per (slot, row).
//  * RegFile (VGPRs): used by the AOT programs, where every slot index is a compile-time constant.
//  * LdsRegFile: used by the generic interpreter, where slot numbers are run-time (wave-uniform) values.  A
//    dynamically indexed VGPR vector is lowered to a tree of scalar compare/branch per access (the generic scan
//    kernel was 28.8 k instructions and ran at 0.7 TB/s); in LDS a slot access is one ds_read/ds_write_b64 at
//    base + ((slot * kRows + row) * 64 + lane) * 8 -- wave-private, consecutive lanes on consecutive banks.
struct RegFile {
  // plain arrays, not ext_vector types: every index is a compile-time constant (AOT / JIT programs), so each (row, slot) is its
  // own scalar value and the slots a program never touches cost no register (a 16-wide vector is kept alive as a whole
  // across loop back-edges: 80 VGPRs per register file whatever the program uses)
  uint64_t v[kRows][kSlots];
  uint32_t valid[kSlots];  // bit r of element s: row r of slot s is valid
  __device__ __forceinline__ uint64_t get(int r, int s) const { return v[r][s]; }
  __device__ __forceinline__ uint32_t getv(int s) const { return valid[s]; }
  __device__ __forceinline__ void set(int r, int s, uint64_t x) { v[r][s] = x; }
  __device__ __forceinline__ void setv(int s, uint32_t m) { valid[s] = m; }
};
struct LdsRegFile {
  unsigned long long* vals;   // [slots][kRows][64]
  unsigned int* vbits;        // [slots][64]
  int lane;
  __device__ __forceinline__ uint64_t get(int r, int s) const { return vals[(s * kRows + r) * 64 + lane]; }
  __device__ __forceinline__ uint32_t getv(int s) const { return vbits[s * 64 + lane]; }
  __device__ __forceinline__ void set(int r, int s, uint64_t x) { vals[(s * kRows + r) * 64 + lane] = x; }
  __device__ __forceinline__ void setv(int s, uint32_t m) { vbits[s * 64 + lane] = m; }
};
// bytes of LDS one wave's LdsRegFile needs
__host__ __device__ constexpr unsigned lds_regfile_bytes_per_wave(unsigned slots) { return slots * (kRows * 64 * 8 + 64 * 4); }

__device__ __forceinline__ int dtype_width_dev(int dt) {
  switch (dt) {
    case PLX_I8: case PLX_U8: return 1;
    case PLX_I16: case PLX_U16: return 2;
    case PLX_I32: case PLX_U32: case PLX_F32: return 4;
    default: return 8;
  }
}
__device__ __forceinline__ double as_f(uint64_t x) { return __longlong_as_double((long long)x); }
__device__ __forceinline__ uint64_t as_u(double x) { return (uint64_t)__double_as_longlong(x); }

// ---- column loads -----------------------------------------------------------------
template <class T, bool FULL>
__device__ __forceinline__ void load2(const void* base_ptr, int64_t row0, int64_t n, uint64_t out[kRows]) {
  const T* p = reinterpret_cast<const T*>(base_ptr);
  if constexpr (FULL) {
    Pack<T, kRows> x = load_pack<T, kRows>(p + row0);
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      if constexpr (is_fp<T>::value) out[r] = as_u((double)x.v[r]);
      else out[r] = (uint64_t)(long long)x.v[r];  // sign- or zero-extends by T
    }
  } else {
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      int64_t i = row0 + r; if (i > n - 1) i = n - 1;
      T x = p[i];
      if constexpr (is_fp<T>::value) out[r] = as_u((double)x);
      else out[r] = (uint64_t)(long long)x;
    }
  }
}

template <bool FULL>
__device__ __forceinline__ void load_input(const Input& in, int dtype, int64_t row0, int64_t n, uint64_t out[kRows], uint32_t& vbits) {
  switch (dtype) {
    case PLX_I64: case PLX_U64: case PLX_F64: load2<uint64_t, FULL>(in.values, row0, n, out); break;
    case PLX_I32: load2<int32_t, FULL>(in.values, row0, n, out); break;
    case PLX_U32: load2<uint32_t, FULL>(in.values, row0, n, out); break;
    case PLX_I16: load2<int16_t, FULL>(in.values, row0, n, out); break;
    case PLX_U16: load2<uint16_t, FULL>(in.values, row0, n, out); break;
    case PLX_I8: load2<int8_t, FULL>(in.values, row0, n, out); break;
    case PLX_U8: load2<uint8_t, FULL>(in.values, row0, n, out); break;
    case PLX_BOOL: {
      int64_t i = row0; if (!FULL && i > n - 1) i = n - 1;
      uint64_t w = reinterpret_cast<const uint64_t*>(in.values)[i >> 6] >> (i & 63);
      out[0] = w & 1; out[1] = (w >> 1) & 1;
    } break;
    default: out[0] = out[1] = 0; break;
  }
  vbits = (1u << kRows) - 1;
  if (in.validity) {
    int64_t i = row0; if (!FULL && i > n - 1) i = n - 1;
    vbits = (uint32_t)(in.validity[i >> 6] >> (i & 63)) & ((1u << kRows) - 1);
    if (!FULL && row0 + 1 > n - 1) vbits &= 1u;  // second row clamped: validity irrelevant (row masked out)
  }
}

// ---- generic interpreter: column prefetch ------------------------------------------------------------
// The interpreter's OP_LOAD writes a dynamically indexed register, so the loaded value is needed at once and
// the loads of a tile would run one after the other (load -> wait -> next load).  Instead the first kPrefetch
// input columns are fetched up front into statically indexed registers as RAW bits (no dependent ALU until
// OP_LOAD converts them): all of a tile's columns are in flight together.
constexpr int kPrefetch = 8;
typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));
struct Prefetched {
  u64x8 bits0, bits1;  // zero-extended raw element bits of rows 0 / 1 (BOOL: the bitmap word of row 0)
  u64x8 vword;         // raw validity word of row 0 (all ones when the column has no bitmap)
};

template <bool FULL>
__device__ __forceinline__ void prefetch_inputs(const Shape& sh, const Args& args, int64_t row0, Prefetched& pf) {
#pragma unroll
  for (int i = 0; i < kPrefetch; i++) {
    if (i < sh.n_inputs) {   // wave-uniform; keeps the loop fully unrolled so register indices stay static
    const Input& in = args.in[i];
    int64_t i0 = row0, i1 = row0 + 1;
    if (!FULL) { if (i0 > args.n_rows - 1) i0 = args.n_rows - 1; if (i1 > args.n_rows - 1) i1 = args.n_rows - 1; }
    uint64_t b0 = 0, b1 = 0;
    switch (dtype_width_dev(sh.in_dtype[i]) * (sh.in_dtype[i] == PLX_BOOL ? 0 : 1)) {
      case 8: if (FULL) { const Pack<uint64_t, 2> x = load_pack<uint64_t, 2>(reinterpret_cast<const uint64_t*>(in.values) + row0); b0 = x.v[0]; b1 = x.v[1]; }
              else { b0 = reinterpret_cast<const uint64_t*>(in.values)[i0]; b1 = reinterpret_cast<const uint64_t*>(in.values)[i1]; } break;
      case 4: if (FULL) { const Pack<uint32_t, 2> x = load_pack<uint32_t, 2>(reinterpret_cast<const uint32_t*>(in.values) + row0); b0 = x.v[0]; b1 = x.v[1]; }
              else { b0 = reinterpret_cast<const uint32_t*>(in.values)[i0]; b1 = reinterpret_cast<const uint32_t*>(in.values)[i1]; } break;
      case 2: if (FULL) { const Pack<uint16_t, 2> x = load_pack<uint16_t, 2>(reinterpret_cast<const uint16_t*>(in.values) + row0); b0 = x.v[0]; b1 = x.v[1]; }
              else { b0 = reinterpret_cast<const uint16_t*>(in.values)[i0]; b1 = reinterpret_cast<const uint16_t*>(in.values)[i1]; } break;
      case 1: if (FULL) { const Pack<uint8_t, 2> x = load_pack<uint8_t, 2>(reinterpret_cast<const uint8_t*>(in.values) + row0); b0 = x.v[0]; b1 = x.v[1]; }
              else { b0 = reinterpret_cast<const uint8_t*>(in.values)[i0]; b1 = reinterpret_cast<const uint8_t*>(in.values)[i1]; } break;
      default: b0 = reinterpret_cast<const uint64_t*>(in.values)[i0 >> 6]; break;   // BOOL: bitmap word of row 0 (rows 2l, 2l+1 share it)
    }
    pf.bits0[i] = b0; pf.bits1[i] = b1;
    pf.vword[i] = in.validity ? in.validity[i0 >> 6] : ~0ull;
    }
  }
}

// Phase 2 of the generic prologue: the LOAD ops are the first n_inputs ops of every program and op i reads input i
// (the host compiler emits loads first, one per distinct column -- engine.cpp Compiler::finish), so input i's raw
// bits (static register index) are widened by dtype and written to the LDS slot op i names.
template <bool FULL, class RF>
__device__ __forceinline__ void store_prefetched(const Shape& sh, const Args& args, int64_t row0, const Prefetched& pf, RF& rf) {
  int64_t i0 = row0; if (!FULL && i0 > args.n_rows - 1) i0 = args.n_rows - 1;
#pragma unroll
  for (int i = 0; i < kPrefetch; i++) {
    if (i < sh.n_inputs) {
      const uint64_t b0 = pf.bits0[i], b1 = pf.bits1[i];
      uint64_t o0, o1;
      switch (sh.in_dtype[i]) {
        case PLX_I32: o0 = (uint64_t)(long long)(int32_t)b0; o1 = (uint64_t)(long long)(int32_t)b1; break;
        case PLX_I16: o0 = (uint64_t)(long long)(int16_t)b0; o1 = (uint64_t)(long long)(int16_t)b1; break;
        case PLX_I8: o0 = (uint64_t)(long long)(int8_t)b0; o1 = (uint64_t)(long long)(int8_t)b1; break;
        case PLX_BOOL: { const uint64_t w = b0 >> (i0 & 63); o0 = w & 1; o1 = (w >> 1) & 1; } break;
        default: o0 = b0; o1 = b1; break;   // 64-bit types and zero-extended unsigned types
      }
      uint32_t vb = (1u << kRows) - 1;
      if (args.in[i].validity) {
        vb = (uint32_t)(pf.vword[i] >> (i0 & 63)) & ((1u << kRows) - 1);
        if (!FULL && row0 + 1 > args.n_rows - 1) vb &= 1u;
      }
      const int dst = sh.ops[i].dst;
      rf.set(0, dst, o0); rf.set(1, dst, o1); rf.setv(dst, vb);
    }
  }
}

// ---- one program step ---------------------------------------------------------------
template <bool FULL, class RF>
__device__ __forceinline__ void exec_op(const Op op, const Shape& sh, const Args& args, int pc, int64_t row0, RF& rf) {
  uint64_t a[kRows], b[kRows], d[kRows];
  uint32_t va = (1u << kRows) - 1, vb = (1u << kRows) - 1, vd;
  if (op.code == OP_LOAD) {
    load_input<FULL>(args.in[op.a], sh.in_dtype[op.a], row0, args.n_rows, d, vd);
  } else if (op.code == OP_CONST) {
#pragma unroll
    for (int r = 0; r < kRows; r++) d[r] = args.imm[pc];
    vd = (1u << kRows) - 1;
  } else {
#pragma unroll
    for (int r = 0; r < kRows; r++) { a[r] = rf.get(r, op.a); b[r] = rf.get(r, op.b); }
    va = rf.getv(op.a); vb = rf.getv(op.b);
    vd = va & vb;
    switch (op.code) {
      case OP_ADD_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u(as_f(a[r]) + as_f(b[r]));
        break;
      case OP_SUB_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u(as_f(a[r]) - as_f(b[r]));
        break;
      case OP_MUL_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u(as_f(a[r]) * as_f(b[r]));
        break;
      case OP_DIV_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u(as_f(a[r]) / as_f(b[r]));
        break;
      case OP_ADD_I:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = a[r] + b[r];
        break;
      case OP_SUB_I:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = a[r] - b[r];
        break;
      case OP_MUL_I:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = a[r] * b[r];
        break;
      case OP_I2F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u((double)(long long)a[r]);
        vd = va;
        break;
      case OP_U2F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = as_u((double)a[r]);
        vd = va;
        break;
      case OP_CMP_I:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = cmp_apply<long long>(op.c, (long long)a[r], (long long)b[r]) ? 1 : 0;
        break;
      case OP_CMP_U:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = cmp_apply<unsigned long long>(op.c, a[r], b[r]) ? 1 : 0;
        break;
      case OP_CMP_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = cmp_apply<double>(op.c, as_f(a[r]), as_f(b[r])) ? 1 : 0;
        break;
      case OP_AND: {  // Kleene (polars-compute/src/boolean.rs: and)
        uint32_t ta = 0, tb = 0;
#pragma unroll
        for (int r = 0; r < kRows; r++) { d[r] = a[r] & b[r] & 1; ta |= (uint32_t)(a[r] & 1) << r; tb |= (uint32_t)(b[r] & 1) << r; }
        vd = (~tb & vb) | (~ta & va) | (ta & va & tb & vb);
      } break;
      case OP_OR: {   // Kleene (boolean.rs: or)
        uint32_t ta = 0, tb = 0;
#pragma unroll
        for (int r = 0; r < kRows; r++) { d[r] = (a[r] | b[r]) & 1; ta |= (uint32_t)(a[r] & 1) << r; tb |= (uint32_t)(b[r] & 1) << r; }
        vd = (ta & va) | (tb & vb) | (~ta & va & ~tb & vb);
      } break;
      case OP_XOR:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = (a[r] ^ b[r]) & 1;
        break;
      case OP_NOT:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = (~a[r]) & 1;
        vd = va;
        break;
      case OP_CANON_F:
#pragma unroll
        for (int r = 0; r < kRows; r++) { double f = as_f(a[r]); d[r] = (f != f) ? 0x7ff8000000000000ull : as_u(f + 0.0); }
        vd = va;
        break;
      case OP_FDIV_I: case OP_MOD_I: {
        uint32_t nz = 0;
        // 64-bit integer division is a ~100-instruction software routine on gfx950; when every lane's
        // operands are non-negative and fit 31 bits (wave-uniform test) the 32-bit divide gives the same result.
        bool narrow = true;
#pragma unroll
        for (int r = 0; r < kRows; r++) narrow = narrow && (((a[r] | b[r]) >> 31) == 0);
        if (__all(narrow)) {
#pragma unroll
          for (int r = 0; r < kRows; r++) {
            const uint32_t x = (uint32_t)a[r], y = (uint32_t)b[r];
            d[r] = y ? (uint64_t)(op.code == OP_FDIV_I ? x / y : x % y) : 0ull;
            nz |= (uint32_t)(y != 0) << r;
          }
        } else {
#pragma unroll
          for (int r = 0; r < kRows; r++) {
            const long long x = (long long)a[r], y = (long long)b[r];
            long long q = 0, m = 0;
            if (y == -1) { q = (long long)(0ull - (unsigned long long)x); m = 0; }   // wrapping_div(MIN, -1) = MIN
            else if (y != 0) { q = x / y; m = x % y; if (m != 0 && ((x < 0) != (y < 0))) { q -= 1; m += y; } }
            d[r] = (uint64_t)(op.code == OP_FDIV_I ? q : m);
            nz |= (uint32_t)(y != 0) << r;
          }
        }
        vd &= nz;
      } break;
      case OP_FDIV_U: case OP_MOD_U: {
        uint32_t nz = 0;
#pragma unroll
        for (int r = 0; r < kRows; r++) {
          const uint64_t x = a[r], y = b[r];
          d[r] = y ? (op.code == OP_FDIV_U ? x / y : x % y) : 0ull;
          nz |= (uint32_t)(y != 0) << r;
        }
        vd &= nz;
      } break;
      case OP_IFNULL:
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = ((va >> r) & 1) ? a[r] : args.imm[pc];
        vd = (1u << kRows) - 1;
        break;
      case OP_BITLOOKUP: {
        const Lut& lut = args.lut[op.c < kMaxLuts ? op.c : 0];
#pragma unroll
        for (int r = 0; r < kRows; r++) {
          const uint64_t idx = a[r] - args.imm[pc];
          d[r] = (((va >> r) & 1) && idx < lut.range) ? ((lut.bits[idx >> 6] >> (idx & 63)) & 1ull) : 0ull;
        }
        vd = va;
      } break;
      case OP_MASKV: {
        uint32_t tb = 0;
#pragma unroll
        for (int r = 0; r < kRows; r++) { d[r] = a[r]; tb |= (uint32_t)(b[r] & 1) << r; }
        vd = va & vb & tb;
      } break;
      default:  // OP_MOV / OP_NOP
#pragma unroll
        for (int r = 0; r < kRows; r++) d[r] = a[r];
        vd = va;
        break;
    }
    vd &= (1u << kRows) - 1;
  }
#pragma unroll
  for (int r = 0; r < kRows; r++) rf.set(r, op.dst, d[r]);
  rf.setv(op.dst, vd);
}

// ---- program providers ----------------------------------------------------------
struct DynProg { static constexpr bool kStatic = false; static constexpr int kId = -1; };
template <int ID> struct StatProg {
  static constexpr bool kStatic = true; static constexpr int kId = ID;
  static constexpr Shape shape() { return static_shape(ID); }
};

template <class P, bool FULL, class RF>
__device__ __forceinline__ void run_program(const Shape& dsh, const Args& args, int64_t row0, RF& rf) {
  if constexpr (P::kStatic) {
    constexpr Shape sh = P::shape();
#pragma unroll
    for (int pc = 0; pc < sh.n_ops; pc++) exec_op<FULL>(sh.ops[pc], sh, args, pc, row0, rf);
  } else {
    // generic interpreter: all column loads in flight first, then widen + store into the LDS slots, then the ops
    Prefetched pf;
    pf.bits0 = 0; pf.bits1 = 0; pf.vword = 0;
    prefetch_inputs<FULL>(dsh, args, row0, pf);
    store_prefetched<FULL>(dsh, args, row0, pf, rf);
    const int first = dsh.n_inputs < kPrefetch ? dsh.n_inputs : kPrefetch;
    for (int pc = first; pc < dsh.n_ops; pc++) exec_op<FULL>(dsh.ops[pc], dsh, args, pc, row0, rf);
  }
}

// Split execution for software pipelining (AOT programs only): the host compiler emits every column load first, so
// ops [0, n_leading_loads) can be issued for the NEXT tile while the current tile is still being processed; the
// loaded registers are simply not touched until run_rest.
__host__ __device__ constexpr int leading_loads(const Shape& s) {
  int n = 0;
  while (n < s.n_ops && s.ops[n].code == OP_LOAD) n++;
  return n;
}
template <class P, class RF>
__device__ __forceinline__ void run_loads_full(const Args& args, int64_t row0, RF& rf) {
  constexpr Shape sh = P::shape();
  constexpr int nl = leading_loads(sh);
#pragma unroll
  for (int pc = 0; pc < nl; pc++) exec_op<true>(sh.ops[pc], sh, args, pc, row0, rf);
}
template <class P, class RF>
__device__ __forceinline__ void run_rest_full(const Args& args, int64_t row0, RF& rf) {
  constexpr Shape sh = P::shape();
  constexpr int nl = leading_loads(sh);
#pragma unroll
  for (int pc = nl; pc < sh.n_ops; pc++) exec_op<true>(sh.ops[pc], sh, args, pc, row0, rf);
}

// ---- late materialisation (AOT / JIT programs) -------------------------------------------------------------------------
// A probe scan needs the predicate and the join key of EVERY row but the aggregate inputs only of the rows that find a build
// row, and in a selective join those are few and far between.  The ops are split at compile time into the ones the predicate /
// key depend on (run for the whole tile) and the rest (run under the lanes' hit mask: a lane without a hit issues no load, so
// 64-byte segments without a hit are never fetched).  Slots are reused by the host compiler, so the split is only taken when
// running all "early" ops before all "late" ones provably reads and leaves the same values as program order.
template <class P, bool FULL, bool EARLY, class RF>
__device__ __forceinline__ void run_split(const Args& args, int64_t row0, RF& rf) {
  constexpr Shape sh = P::shape();
  constexpr ProgramSplit sp = split_program(sh);
#pragma unroll
  for (int pc = 0; pc < sh.n_ops; pc++)
    if (((sp.early >> pc) & 1u) == (EARLY ? 1u : 0u)) exec_op<FULL>(sh.ops[pc], sh, args, pc, row0, rf);
}

// Register file of a kernel running program provider P: VGPRs for AOT programs, the wave's slice of dynamic LDS
// (at args.rf_lds_offset, after the sink's own LDS) for the generic interpreter.
template <class P> struct RegFileOf { using type = RegFile; };
template <> struct RegFileOf<DynProg> { using type = LdsRegFile; };
template <class P>
__device__ __forceinline__ typename RegFileOf<P>::type make_regfile(const Args& args) {
  if constexpr (P::kStatic) { return RegFile{}; }
  else {
    extern __shared__ unsigned long long plx_dyn_lds[];
    const unsigned per_wave = lds_regfile_bytes_per_wave(args.rf_slots);
    unsigned char* base = reinterpret_cast<unsigned char*>(plx_dyn_lds) + args.rf_lds_offset + (threadIdx.x >> 6) * per_wave;
    LdsRegFile rf;
    rf.vals = reinterpret_cast<unsigned long long*>(base);
    rf.vbits = reinterpret_cast<unsigned int*>(base + (size_t)args.rf_slots * kRows * 64 * 8);
    rf.lane = lane_id();
    return rf;
  }
}

// ---- aggregate combine (bit patterns) -----------------------------------------------
__device__ __forceinline__ uint64_t agg_combine(uint8_t kind, uint64_t x, uint64_t y) {
  switch (kind) {
    case AGG_SUM_F: return as_u(as_f(x) + as_f(y));
    case AGG_MIN_F: return as_u(min_ign<double>(as_f(x), as_f(y)));
    case AGG_MAX_F: return as_u(max_ign<double>(as_f(x), as_f(y)));
    case AGG_MIN_I: return (uint64_t)((long long)x < (long long)y ? (long long)x : (long long)y);
    case AGG_MAX_I: return (uint64_t)((long long)x > (long long)y ? (long long)x : (long long)y);
    case AGG_MIN_U: case AGG_FIRST_ROW: return x < y ? x : y;
    case AGG_MAX_U: return x > y ? x : y;
    default: return x + y;  // SUM_I, COUNT, COUNT_ORD, LEN
  }
}
__device__ __forceinline__ uint64_t agg_identity_dev(uint8_t kind) {
  switch (kind) {
    case AGG_MIN_F: return 0x7ff0000000000000ull;
    case AGG_MAX_F: return 0xfff0000000000000ull;
    case AGG_MIN_I: return 0x7fffffffffffffffull;
    case AGG_MAX_I: return 0x8000000000000000ull;
    case AGG_MIN_U: case AGG_FIRST_ROW: return ~0ull;
    default: return 0ull;
  }
}
// value an aggregate contributes for one row: `sel` = row selected and (where it matters) src valid
__device__ __forceinline__ uint64_t agg_row_value(uint8_t kind, uint64_t v, bool pass, bool valid, uint64_t row) {
  switch (kind) {
    case AGG_LEN: return pass ? 1ull : 0ull;
    case AGG_COUNT: return (pass && valid) ? 1ull : 0ull;
    case AGG_COUNT_ORD: { double f = as_f(v); return (pass && valid && f == f) ? 1ull : 0ull; }
    case AGG_FIRST_ROW: return pass ? row : ~0ull;
    case AGG_MIN_F: { double f = as_f(v); return (pass && valid && f == f) ? v : 0x7ff0000000000000ull; }
    case AGG_MAX_F: { double f = as_f(v); return (pass && valid && f == f) ? v : 0xfff0000000000000ull; }
    case AGG_SUM_F: case AGG_SUM_I: return (pass && valid) ? v : 0ull;   // +0.0 / 0 identities share the zero pattern
    default: return (pass && valid) ? v : agg_identity_dev(kind);
  }
}

// ---- device-scope atomic update of one aggregate cell -------------------------------
__device__ __forceinline__ void atomic_agg(uint8_t kind, unsigned long long* cell, uint64_t v) {
  switch (kind) {
    case AGG_SUM_F: unsafeAtomicAdd(reinterpret_cast<double*>(cell), as_f(v)); break;  // global_atomic_add_f64
    case AGG_MIN_I: atomicMin(reinterpret_cast<long long*>(cell), (long long)v); break;
    case AGG_MAX_I: atomicMax(reinterpret_cast<long long*>(cell), (long long)v); break;
    case AGG_MIN_U: case AGG_FIRST_ROW: atomicMin(cell, (unsigned long long)v); break;
    case AGG_MAX_U: atomicMax(cell, (unsigned long long)v); break;
    case AGG_MIN_F: case AGG_MAX_F: {
      unsigned long long old = *cell;
      for (;;) {
        uint64_t nv = agg_combine(kind, old, v);
        if (nv == old) break;
        unsigned long long prev = atomicCAS(cell, old, (unsigned long long)nv);
        if (prev == old) break;
        old = prev;
      }
    } break;
    default: atomicAdd(cell, (unsigned long long)v); break;
  }
}

// ---- LDS atomic update of one aggregate cell ----------------------------------------------
__device__ __forceinline__ void lds_atomic_agg(uint8_t kind, unsigned long long* cell, uint64_t v) {
  switch (kind) {
    case AGG_SUM_F: unsafeAtomicAdd(reinterpret_cast<double*>(cell), as_f(v)); break;  // ds_add_f64
    case AGG_MIN_I: atomicMin(reinterpret_cast<long long*>(cell), (long long)v); break;
    case AGG_MAX_I: atomicMax(reinterpret_cast<long long*>(cell), (long long)v); break;
    case AGG_MIN_U: case AGG_FIRST_ROW: atomicMin(cell, (unsigned long long)v); break;
    case AGG_MAX_U: atomicMax(cell, (unsigned long long)v); break;
    case AGG_MIN_F: case AGG_MAX_F: {
      unsigned long long old = *cell;
      for (;;) {
        uint64_t nv = agg_combine(kind, old, v);
        if (nv == old) break;
        unsigned long long prev = atomicCAS(cell, old, (unsigned long long)nv);
        if (prev == old) break;
        old = prev;
      }
    } break;
    default: atomicAdd(cell, (unsigned long long)v); break;
  }
}

template <class S, class RF>
__device__ __forceinline__ void atomic_row(const S& sh, const RF& rf, int r, int64_t row, unsigned long long* cells) {
#pragma unroll
  for (int k = 0; k < kMaxAggs; k++) {
    if (k < sh.n_aggs) {
      const Agg ag = sh.aggs[k];
      uint64_t v = rf.get(r, ag.src);
      bool valid = (rf.getv(ag.src) >> r) & 1;
      uint64_t x = agg_row_value(ag.kind, v, true, valid, (uint64_t)row);
      if (x != agg_identity_dev(ag.kind) || ag.kind == AGG_SUM_F) {
        if (ag.kind == AGG_SUM_F && !valid) continue;
        atomic_agg(ag.kind, cells + k, x);
      }
    }
  }
}

// ---- the scan kernels ------------------------------------------------------------------
template <class P, class RF>
__device__ __forceinline__ bool tile_rows(const Shape& dsh, const Args& args, int64_t tile, RF& rf, bool pass[kRows], int64_t& row0) {
  const int lane = lane_id();
  const int64_t base = tile * kTileRows;
  row0 = base + (int64_t)lane * kRows;
  const bool full = base + kTileRows <= args.n_rows;  // wave-uniform
  if (full) run_program<P, true>(dsh, args, row0, rf);
  else run_program<P, false>(dsh, args, row0, rf);
  uint8_t pred;
  if constexpr (P::kStatic) { constexpr Shape sh = P::shape(); pred = sh.pred; } else pred = dsh.pred;
#pragma unroll
  for (int r = 0; r < kRows; r++) {
    bool ok = full || (row0 + r < args.n_rows);
    if (pred != kNone) ok = ok && (rf.get(r, pred) & 1) && ((rf.getv(pred) >> r) & 1);
    pass[r] = ok;
  }
  return full;
}

// ---- host: launching the generic interpreter -----------------------------------------------------------
// Its register file lives in dynamic LDS behind the sink's own LDS: returns the Args / LDS size to launch with.
struct DynLaunch { Args args; size_t lds; };
inline DynLaunch dyn_launch(const Shape& sh, const Args& a, size_t sink_lds) {
  DynLaunch d{a, 0};
  d.args.rf_slots = program_slots(sh);
  d.args.rf_lds_offset = (uint32_t)((sink_lds + 15) & ~(size_t)15);
  d.lds = d.args.rf_lds_offset + (size_t)(kBlock / 64) * lds_regfile_bytes_per_wave(d.args.rf_slots);
  return d;
}

}  // namespace k
}  // namespace plx
