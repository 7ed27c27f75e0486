"""tla_rust_amd — MI355X-native explicit-state model checker for the spacejam/tla-rust specs.

The package is a thin ctypes binding of the C ABI in include/tlamc.h (libtlamc.so: hand-written
HIP kernels for gfx950).  There is no CPU fallback: importing works anywhere (so the cfg
front-end and the symbol table can be tested without a GPU), but every compute entry point
raises if the shared library is missing or no HIP device is present.
"""
from .binding import (Engine, McError, Program, ResolvedSpec, Result, SPEC_IDS, VERDICTS, cfg_parse, check_files, device_count, lib,  # noqa: F401
                      pcal_translate, spec_desc, spec_resolve, state_bytes, state_format)
