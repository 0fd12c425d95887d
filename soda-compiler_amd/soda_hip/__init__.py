"""soda_hip: an MI355X (gfx950) back end for the SODA stencil DSL.

Sub-packages:
  frontend  `.soda` text -> Program -> Stencil analysis
  codegen   Stencil -> HIP kernel text, Python host shim, C header
            (`codegen.backend` is the sodac plug-in: add_arguments/print_code)
  runtime   ctypes binding of libsoda_hip.so, `<app>_test` protocol, multi-GPU
"""
__version__ = '0.1.0'
