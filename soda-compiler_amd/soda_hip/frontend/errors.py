"""Error types of the SODA front end.

`sodac` turns both into exit status 1, the way the reference driver does for
`TextXSyntaxError` / `SemanticError` (reference src/sodac:129-136).
"""


class SodaError(Exception):
  """Base class; anything derived from it is a user-facing diagnostic."""


class SodaSyntaxError(SodaError):
  def __init__(self, message, line=None, col=None):
    self.line, self.col = line, col
    where = '' if line is None else ' at line %d, column %d' % (line, col)
    super().__init__('syntax error%s: %s' % (where, message))


class SemanticError(SodaError):
  pass
