"""`.soda` text -> `Program` (parser) -> `Stencil` (analysis)."""
from .errors import SemanticError, SodaError, SodaSyntaxError
from .parser import parse, parse_expression
from .stencil import Stencil, stencil_from_program


def load(path, **overrides):
  """Parses a `.soda` file and analyses it; keyword overrides as on the
  `sodac` command line (burst_width, unroll_factor, tile_size, iterate)."""
  with open(path) as f:
    return stencil_from_program(parse(f.read()), **overrides)


def loads(text, **overrides):
  return stencil_from_program(parse(text), **overrides)
