"""Pieces shared by every generated HIP kernel: the include-free prelude, type
and math-call rewriting of expression text, the embedded metadata symbol.

The generated translation unit has no `#include`: it must compile both offline
(`hipcc -x hip --cuda-device-only`, at `sodac` time) and at run time through
hiprtc (libsoda_hip.so), so it spells HIP's keywords as the clang attributes
they expand to and uses the amdgcn builtins for work-item ids.
"""
import hashlib
import json
import re

from . import spec as specmod

ABI_VERSION = 7

# A launch whose box (inputs + outputs) is larger than this streams: nothing it writes
# is still cached when the next launch reads it, and its stores go out non-temporal
# (the MI355X's Infinity Cache holds 256 MiB; measured cross-over of the 3-D block
# form between boxes of 262 MB, where bypassing costs 5 %, and 325 MB, where it gains
# 5 %: profiles/r03_blk_variants.txt)
NT_STREAMING_BYTES = 288 * 1024 * 1024

# element types as builtin spellings (no <stdint.h> in the translation unit)
BUILTIN_TYPE = {
    'uint8_t': 'unsigned char', 'int8_t': 'signed char',
    'uint16_t': 'unsigned short', 'int16_t': 'short',
    'uint32_t': 'unsigned int', 'int32_t': 'int',
    'uint64_t': 'unsigned long long', 'int64_t': 'long long',
    'float': 'float', 'double': 'double', '_Float16': '_Float16',
}
_TYPE_RE = re.compile(r'\b(u?int(?:8|16|32|64)_t)\b')

# Math calls.  The reference's CPU path compiles the expression text as C++ with
# <cmath> but without `using namespace std` (reference host.py:12-36), so an
# unqualified `sqrt(x)` on a float is the C function `double sqrt(double)`:
# the argument is promoted and the rest of the expression continues in double.
# The wrappers below reproduce exactly that overload set on the device.
_DOUBLE_UNARY = ('cos sin tan acos asin atan cosh sinh tanh acosh asinh atanh exp '
                 'log log10 exp2 expm1 log1p log2 logb cbrt erf erfc tgamma lgamma '
                 'ceil floor trunc round rint nearbyint fabs sqrt').split()
_DOUBLE_BINARY = ('atan2 pow hypot fmod remainder copysign nextafter fdim fmax '
                  'fmin').split()
# (double, int) -> double; double -> integer: as <math.h> declares them globally
_DOUBLE_INT = ('ldexp', 'scalbn', 'scalbln')
_TO_INTEGER = {'ilogb': 'int', 'lround': 'long', 'llround': 'long long',
               'lrint': 'long', 'llrint': 'long long'}
_CALL_RE = re.compile(r'\b(%s)\s*\(' % '|'.join(
    _DOUBLE_UNARY + _DOUBLE_BINARY + list(_DOUBLE_INT) + sorted(_TO_INTEGER) +
    ['fma', 'abs', 'min', 'max', 'select']))


def builtin_type(c_type):
  return BUILTIN_TYPE[c_type]


def device_expr(text):
  """Expression text of the spec -> device C++ (loads still as placeholders)."""
  text = _TYPE_RE.sub(lambda m: BUILTIN_TYPE[m.group(1)], text)
  return _CALL_RE.sub(lambda m: 'soda_fn_%s(' % m.group(1), text)


def used_functions(spec):
  found = set()
  for stage in spec['stages']:
    for text in [stage['expr']] + [l['expr'] for l in stage['lets']]:
      found.update(m.group(1) for m in _CALL_RE.finditer(text))
  return found


def math_wrappers(names):
  out = []
  for name in sorted(names):
    if name == 'sqrt':
      out.append('DEV double soda_fn_sqrt(double x) { return __builtin_sqrt(x); }')
    elif name == 'fabs':
      out.append('DEV double soda_fn_fabs(double x) { return __builtin_fabs(x); }')
    elif name in ('floor', 'ceil', 'trunc', 'rint', 'nearbyint', 'round'):
      out.append('DEV double soda_fn_%s(double x) { return __builtin_%s(x); }'
                 % (name, name))
    elif name in _DOUBLE_UNARY:
      out.append('extern "C" __attribute__((device)) double __ocml_%s_f64(double);'
                 % name)
      out.append('DEV double soda_fn_%s(double x) { return __ocml_%s_f64(x); }'
                 % (name, name))
    elif name in ('fmax', 'fmin', 'copysign'):
      out.append('DEV double soda_fn_%s(double x, double y) '
                 '{ return __builtin_%s(x, y); }' % (name, name))
    elif name in _DOUBLE_BINARY:
      out.append('extern "C" __attribute__((device)) double '
                 '__ocml_%s_f64(double, double);' % name)
      out.append('DEV double soda_fn_%s(double x, double y) '
                 '{ return __ocml_%s_f64(x, y); }' % (name, name))
    elif name in _DOUBLE_INT:
      out.append('extern "C" __attribute__((device)) double __ocml_ldexp_f64(double, int);')
      out.append('DEV double soda_fn_%s(double x, %s n) { return __ocml_ldexp_f64(x, '
                 '(int)(n < -100000 ? -100000 : (n > 100000 ? 100000 : n))); }'
                 % (name, 'long' if name == 'scalbln' else 'int'))
    elif name == 'ilogb':
      out.append('extern "C" __attribute__((device)) int __ocml_ilogb_f64(double);')
      out.append('DEV int soda_fn_ilogb(double x) { return __ocml_ilogb_f64(x); }')
    elif name in _TO_INTEGER:
      out.append('DEV %s soda_fn_%s(double x) { return (%s)__builtin_%s(x); }' % (
          _TO_INTEGER[name], name, _TO_INTEGER[name],
          'round' if 'round' in name else 'rint'))
    elif name == 'fma':
      out.append('DEV double soda_fn_fma(double x, double y, double z) '
                 '{ return __builtin_fma(x, y, z); }')
    elif name == 'abs':   # <cstdlib>'s int abs(int) is the visible one
      out.append('DEV int soda_fn_abs(int x) { return x < 0 ? -x : x; }')
    elif name in ('min', 'max'):
      cmp = '<' if name == 'min' else '>'
      out.append('template <typename A, typename B> DEV auto soda_fn_%s(A a, B b) '
                 '-> decltype(a + b) { return (b %s a) ? b : a; }' % (name, cmp))
    elif name == 'select':
      out.append('template <typename C, typename A, typename B> DEV auto '
                 'soda_fn_select(C c, A a, B b) -> decltype(a + b) '
                 '{ return c ? a : b; }')
  return out


def program_hash(spec):
  """Identity of what a blob computes: stages, types, windows.  Tuning knobs
  (burst width, unroll factor, tile size, iterate) are deliberately excluded;
  the reference ties kernel and host together with -D guards instead
  (reference hls_kernel.py:154-160)."""
  essential = dict(
      app_name=spec['app_name'], dim=spec['dim'],
      inputs=[(t['name'], t['c_type']) for t in spec['inputs']],
      outputs=spec['outputs'],
      stages=[(s['name'], s['c_type'], s['expr'],
               [(l['name'], l['c_type'], l['expr']) for l in s['lets']])
              for s in spec['stages']])
  return hashlib.sha1(json.dumps(essential, sort_keys=True).encode()).hexdigest()


PRELUDE = '''\
// Generated by sodac (SODA HIP back end, soda_hip %(version)s) for gfx950.
// kernel: %(app)s    program hash: %(hash)s
// Compile: hipcc -x hip --offload-arch=gfx950 --cuda-device-only \\
//          --no-gpu-bundle-output -O3 -ffp-contract=off -fno-slp-vectorize \\
//          -fwrapv -std=c++17
// or hand this text to libsoda_hip.so (hiprtc).  -ffp-contract=off and -fwrapv
// are part of the contract: no fused multiply-add, and signed overflow wraps
// as it does in the reference's g++-built CPU path.
#define GLOBAL extern "C" __attribute__((global))
#define DEV static __attribute__((device)) inline __attribute__((always_inline))
#define WG_SIZE(n) __attribute__((amdgpu_flat_work_group_size(n, n)))
typedef long long i64;

// by-value kernel argument; layout = soda_hip_args of include/soda_hip.h
struct soda_hip_args {
  void* tensor[16];
  i64 dims[4];
  i64 box_lo[4];
  i64 box_hi[4];
  i64 param[4];
};

// workgroup barrier that also orders LDS traffic (what __syncthreads() is)
DEV void soda_block_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// The same for kernels that keep LDS-direct loads (global_load_lds) in flight
// across the barrier: a release fence would have to wait for them (they are LDS
// writes: vmcnt(0)) and undo the prefetch.  Waits for this wavefront's own DS
// traffic only; the memory clobber keeps the compiler from moving LDS accesses
// across.  Rows that arrive by LDS-direct load are consumed by the wavefront that
// issued them, after its own explicit vmcnt wait.
DEV void soda_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");
}
// 16 bytes from LDS through an instruction the compiler does not see as an LDS
// read: before a visible read of memory that LDS-direct loads write, it waits
// for ALL of them (vmcnt(0)), whichever slot they target; the kernels wait for
// exactly the row they need (explicit vmcnt) and then read it with this.
typedef float soda_f4 __attribute__((ext_vector_type(4)));
DEV soda_f4 soda_lds_read_f4(const void* p) {
  soda_f4 v;
  asm volatile("ds_read_b128 %%0, %%1\\n\\ts_waitcnt lgkmcnt(0)"
               : "=v"(v) : "v"((unsigned)(unsigned long long)p) : "memory");
  return v;
}
// Wide strips: a lane's 8 consecutive floats from LDS as four register pairs
// {f[c], f[c+4]} (ds_read2_b32 puts its two dwords into one 64-bit pair): the
// pairs of the packed form without register moves.  Invisible to the compiler
// like soda_lds_read_f4, for the same reason.
typedef float soda_pk2 __attribute__((ext_vector_type(2)));
DEV void soda_lds_read_pairs4(const void* p, soda_pk2& a, soda_pk2& b, soda_pk2& c,
                              soda_pk2& d) {
  asm volatile("ds_read2_b32 %%0, %%4 offset1:4\\n\\t"
               "ds_read2_b32 %%1, %%4 offset0:1 offset1:5\\n\\t"
               "ds_read2_b32 %%2, %%4 offset0:2 offset1:6\\n\\t"
               "ds_read2_b32 %%3, %%4 offset0:3 offset1:7\\n\\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
               : "v"((unsigned)(unsigned long long)p) : "memory");
}
DEV int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// value of `v` in lane-1 / lane+1 of the 64-lane wavefront (DPP wave shifts;
// the lane with no such neighbour receives 0)
DEV int dpp_from_below(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
DEV int dpp_from_above(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
template <typename T, bool BELOW> DEV T lane_neighbour(T v) {
  if constexpr (sizeof(T) == 4) {
    const int w = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(T, BELOW ? dpp_from_below(w) : dpp_from_above(w));
  } else if constexpr (sizeof(T) == 8) {
    struct halves { int lo, hi; };
    halves h = __builtin_bit_cast(halves, v);
    h.lo = BELOW ? dpp_from_below(h.lo) : dpp_from_above(h.lo);
    h.hi = BELOW ? dpp_from_below(h.hi) : dpp_from_above(h.hi);
    return __builtin_bit_cast(T, h);
  } else if constexpr (sizeof(T) == 2) {
    const int w = __builtin_bit_cast(unsigned short, v);
    return __builtin_bit_cast(T, (unsigned short)(BELOW ? dpp_from_below(w)
                                                        : dpp_from_above(w)));
  } else {
    const int w = __builtin_bit_cast(unsigned char, v);
    return __builtin_bit_cast(T, (unsigned char)(BELOW ? dpp_from_below(w)
                                                       : dpp_from_above(w)));
  }
}
// Rows across the two 32-lane halves of a wavefront (v_permlane32_swap): `first`
// and `last` are a lane's first and last tile rows.  The upper half receives in
// `above` the lower half's `last` (the row above its first one), the lower half
// receives in `below` the upper half's `first`; the other halves get their own
// values back (tile-edge halo, never used for a stored cell).
template <typename T> DEV void rows_across_halves(T first, T last, T& above, T& below) {
  static_assert(sizeof(T) == 4, "rows_across_halves: 4-byte elements");
  const auto r = __builtin_amdgcn_permlane32_swap(
      __builtin_bit_cast(unsigned, first), __builtin_bit_cast(unsigned, last),
      false, false);
  above = __builtin_bit_cast(T, (unsigned)r[0]);
  below = __builtin_bit_cast(T, (unsigned)r[1]);
}
// ... where the lane without such a neighbour keeps `edge` instead of receiving 0 (DPP
// without bound_ctrl leaves `old` in lanes that have no source lane): the first / last
// lane of a seam-free strip (kernel_stream2d, align='exact') hold there what the strip's
// edge loads fetched
template <typename T, bool BELOW> DEV T lane_neighbour_or(T v, T edge) {
  if constexpr (sizeof(T) == 8) {
    struct halves { int lo, hi; };
    halves h = __builtin_bit_cast(halves, v), e = __builtin_bit_cast(halves, edge);
    h.lo = BELOW ? __builtin_amdgcn_update_dpp(e.lo, h.lo, 0x138, 0xf, 0xf, false)
                 : __builtin_amdgcn_update_dpp(e.lo, h.lo, 0x130, 0xf, 0xf, false);
    h.hi = BELOW ? __builtin_amdgcn_update_dpp(e.hi, h.hi, 0x138, 0xf, 0xf, false)
                 : __builtin_amdgcn_update_dpp(e.hi, h.hi, 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(T, h);
  } else {
    int w, o;
    if constexpr (sizeof(T) == 4) {
      w = __builtin_bit_cast(int, v);
      o = __builtin_bit_cast(int, edge);
    } else if constexpr (sizeof(T) == 2) {
      w = __builtin_bit_cast(unsigned short, v);
      o = __builtin_bit_cast(unsigned short, edge);
    } else {
      w = __builtin_bit_cast(unsigned char, v);
      o = __builtin_bit_cast(unsigned char, edge);
    }
    const int r = BELOW ? __builtin_amdgcn_update_dpp(o, w, 0x138, 0xf, 0xf, false)
                        : __builtin_amdgcn_update_dpp(o, w, 0x130, 0xf, 0xf, false);
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, r);
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (unsigned short)r);
    else return __builtin_bit_cast(T, (unsigned char)r);
  }
}
template <typename T> DEV T from_lane_below_or(T v, T edge) { return lane_neighbour_or<T, true>(v, edge); }
template <typename T> DEV T from_lane_above_or(T v, T edge) { return lane_neighbour_or<T, false>(v, edge); }
template <typename T> DEV T from_lane_below(T v) { return lane_neighbour<T, true>(v); }
template <typename T> DEV T from_lane_above(T v) { return lane_neighbour<T, false>(v); }
// Packed pairs (kernel_stream2d_wp, pairs=1): the neighbour lane's pair as two
// SCALARS.  v_pk_add_f32 cannot take a DPP operand, so `pair + shifted pair`
// costs one packed add and two v_mov_b32_dpp; as two scalar adds the compiler
// folds each shift into its add (v_add_f32_dpp): two instructions instead of
// three, same IEEE operations.
typedef float pk2 __attribute__((ext_vector_type(2)));
struct pk2_shifted {
  float lo, hi;
  __attribute__((device)) inline __attribute__((always_inline)) operator pk2() const { return pk2{lo, hi}; }
};
DEV pk2_shifted pk_from_lane_below(pk2 v) {
  return pk2_shifted{from_lane_below(v[0]), from_lane_below(v[1])};
}
DEV pk2_shifted pk_from_lane_above(pk2 v) {
  return pk2_shifted{from_lane_above(v[0]), from_lane_above(v[1])};
}
// Wide strips (pairs=2): a lane holds 2C consecutive columns, the pair's low
// half the first C and its high half the last C.  The column before the lane's
// first one is the lower lane's last (high half); the column before the high
// half's first one is the lane's own low half - and mirrored upwards.
DEV pk2_shifted pk_wide_below(pk2 v) {
  return pk2_shifted{from_lane_below(v[1]), v[0]};
}
DEV pk2_shifted pk_wide_above(pk2 v) {
  return pk2_shifted{v[1], from_lane_above(v[0])};
}
// The two scalar results go through an empty asm so that instruction selection
// cannot put them back together as one packed operation on assembled operands.
DEV pk2 pk_of_scalars(float lo, float hi) {
  asm("" : "+v"(lo));
  asm("" : "+v"(hi));
  return pk2{lo, hi};
}
// Shader clock under load (soda_hip_clock_probe_start, include/soda_hip.h): ONE wavefront
// that sleeps for `spins` x ~8 k cycles beside whatever the chip is running and reports
// how many shader cycles (s_memtime) and how many ticks of the constant 100 MHz clock
// (s_memrealtime) went by; no VGPRs to speak of, no LDS - it fits next to any kernel.
GLOBAL WG_SIZE(64) void soda_hip_clock_probe(unsigned long long* out, int spins) {
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane_id() == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
#define SODA_PK_SHIFTED_OP(OP)                                                  \
  DEV pk2 operator OP(pk2 a, pk2_shifted b) { return pk_of_scalars(a[0] OP b.lo, a[1] OP b.hi); } \
  DEV pk2 operator OP(pk2_shifted a, pk2 b) { return pk_of_scalars(a.lo OP b[0], a.hi OP b[1]); } \
  DEV pk2 operator OP(pk2_shifted a, pk2_shifted b) { return pk_of_scalars(a.lo OP b.lo, a.hi OP b.hi); } \
  DEV pk2 operator OP(pk2_shifted a, float b) { return pk_of_scalars(a.lo OP b, a.hi OP b); } \
  DEV pk2 operator OP(float a, pk2_shifted b) { return pk_of_scalars(a OP b.lo, a OP b.hi); }
SODA_PK_SHIFTED_OP(+)
SODA_PK_SHIFTED_OP(-)
SODA_PK_SHIFTED_OP(*)
SODA_PK_SHIFTED_OP(/)
#undef SODA_PK_SHIFTED_OP
DEV pk2 operator-(pk2_shifted a) { return pk2{-a.lo, -a.hi}; }
DEV pk2 operator+(pk2_shifted a) { return pk2{a.lo, a.hi}; }
'''


def prelude(spec, version):
  return PRELUDE % dict(version=version, app=spec['app_name'],
                        hash=program_hash(spec))


def meta_symbol(spec, kernels, extra=None):
  """`soda_hip_meta`: JSON the run time reads back from the loaded blob."""
  meta = dict(abi=ABI_VERSION, program_hash=program_hash(spec),
              app_name=spec['app_name'], kernels=kernels, spec=spec)
  if extra:
    meta.update(extra)
  text = json.dumps(meta, sort_keys=True, separators=(',', ':'))
  escaped = text.replace('\\', '\\\\').replace('"', '\\"')
  # split into adjacent literals to keep lines readable
  chunks = [escaped[i:i + 100] for i in range(0, len(escaped), 100)]
  # never split inside an escape sequence
  fixed = []
  carry = ''
  for ch in chunks:
    ch = carry + ch
    carry = ''
    n = len(ch) - len(ch.rstrip('\\'))
    if n % 2 == 1:
      carry = '\\'
      ch = ch[:-1]
    fixed.append(ch)
  if carry:
    fixed.append(carry)
  body = '\n'.join('    "%s"' % ch for ch in fixed)
  return ('extern "C" __attribute__((device)) __attribute__((used)) '
          'const char soda_hip_meta[] =\n%s;\n' % body)


def read_meta_from_source(text):
  """Recovers the metadata JSON from generated kernel TEXT (no device needed)."""
  m = re.search(r'soda_hip_meta\[\] =\n((?:    ".*"\n?)+);', text)
  if not m:
    raise ValueError('no soda_hip_meta in kernel text')
  raw = ''.join(line.strip()[1:-1] for line in m.group(1).splitlines()
                if line.strip())
  return json.loads(raw.replace('\\"', '"').replace('\\\\', '\\'))


def tensor_index(spec):
  names = [t['name'] for t in spec['inputs']] + [s['name'] for s in spec['stages']]
  return {n: i for i, n in enumerate(names)}


def elem_sizes(spec):
  types = specmod.tensor_c_types(spec)
  idx = tensor_index(spec)
  out = [0] * len(idx)
  for n, i in idx.items():
    out[i] = specmod.ELEM_SIZE[types[n]]
  return out


def cell_assignment(stage, target, load, emit, indent):
  """Emits the C++ of ONE cell of a fused kernel: the stage's `let`s, then
  `target = expression`, operands substituted by `load(tensor, rel)`.

  An operand that comes from a neighbouring LANE (from_lane_below / from_lane_above:
  DPP) must be read with every lane active: inside the right-hand side of `&&` or
  `||` it would be read only by the lanes that get there, and a lane whose
  neighbour took the other way would receive 0 instead of the neighbour's value.
  Expressions that short-circuit therefore get their lane-crossing operands
  evaluated first, into temporaries (found by tests/random_programs.py:
  operator_program; the per-stage kernels read memory and are not concerned)."""
  texts = [let['expr'] for let in stage['lets']] + [stage['expr']]
  hoisted = []
  if any('&&' in t or '||' in t for t in texts):
    seen = {}
    plain = load

    def load(tensor, rel):      # noqa: F811 - the hoisting wrapper
      text = plain(tensor, rel)
      if 'from_lane' not in text:
        return text
      if text not in seen:
        seen[text] = 'soda_lane%d' % len(seen)
        hoisted.append('const auto %s = %s;' % (seen[text], text))
      return seen[text]
  lets = [(builtin_type(let['c_type']), let['name'],
           specmod.substitute_loads(device_expr(let['expr']), load))
          for let in stage['lets']]
  value = specmod.substitute_loads(device_expr(stage['expr']), load)
  if not lets and not hoisted:
    emit('%s%s = %s;' % (indent, target, value))
    return
  emit(indent + '{')
  for text in hoisted:
    emit('%s  %s' % (indent, text))
  for ctype, name, text in lets:
    emit('%s  const %s %s = %s;' % (indent, ctype, name, text))
  emit('%s  %s = %s;' % (indent, target, value))
  emit(indent + '}')

