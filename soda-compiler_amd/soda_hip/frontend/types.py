"""DSL element types (reference src/haoda/util.py:9-13, :145-180).

The DSL spells types `uint16`, `int32`, `float`, `double`, `half`, `float32`,
`float64`, and arbitrary widths such as `uint5` or `float18_6`.  The HIP back end
supports the widths a GPU register holds natively; the arbitrary-width forms map
to Xilinx `ap_[u]int<N>` in the reference and are rejected by `hip_type`.
"""
import re

from .errors import SemanticError

TYPE_RE = re.compile(r'(?:u?int[1-9]\d*(?:_[1-9]\d*)?|float[1-9]\d*(?:_[1-9]\d*)?'
                     r'|float|double|half)\Z')
_NATIVE_INT = ('uint8', 'uint16', 'uint32', 'uint64',
               'int8', 'int16', 'int32', 'int64')
_FLOAT_WIDTH = {'float': 32, 'double': 64, 'half': 16}


def is_type_name(text):
  return TYPE_RE.match(text) is not None


def is_float(haoda_type):
  return haoda_type in ('half', 'double') or haoda_type.startswith('float')


def width_in_bits(haoda_type):
  if haoda_type in _FLOAT_WIDTH:
    return _FLOAT_WIDTH[haoda_type]
  m = re.match(r'(?:uint|int|float)(\d+)', haoda_type)
  if not m:
    raise SemanticError('unknown type: %s' % haoda_type)
  return int(m.group(1))


def width_in_bytes(haoda_type):
  return (width_in_bits(haoda_type) - 1) // 8 + 1


def c_type(haoda_type):
  """The C spelling the reference uses (util.get_c_type)."""
  if haoda_type in _NATIVE_INT:
    return haoda_type + '_t'
  if haoda_type == 'float32':
    return 'float'
  if haoda_type == 'float64':
    return 'double'
  for prefix in ('int', 'uint'):
    if haoda_type.startswith(prefix):
      return 'ap_%s<%s>' % (prefix, haoda_type[len(prefix):])
  return haoda_type


def hip_type(haoda_type):
  """C type usable in HIP device code and in the CPU oracle; raises for the
  FPGA-only arbitrary-precision types."""
  ct = c_type(haoda_type)
  if ct in ('float', 'double') or ct.endswith('_t'):
    return ct
  if ct == 'half':
    return '_Float16'
  raise SemanticError(
      'type %s has no native GPU representation (HIP back end supports '
      '[u]int8/16/32/64, half, float, double)' % haoda_type)


def numpy_name(haoda_type):
  ct = hip_type(haoda_type)
  return {'float': 'float32', 'double': 'float64', '_Float16': 'float16'}.get(
      ct, ct[:-2])
