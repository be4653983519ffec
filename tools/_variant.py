"""Development tools only: `PTE_LIB=<path> python tools/<tool>.py` runs the tool against a tuning build of libpte (tools/build_variant.sh).
The PRODUCT never reads the variable (since round 5 pigeons_amd loads pigeons.jl_amd/lib/libpte.so and nothing else): a tool opts in by calling
apply() after it has put the package on sys.path and before it makes an engine."""
import os


def apply():
    p = os.environ.get("PTE_LIB")
    if p:
        from pigeons_amd import _lib
        _lib.use_library(p)
    return p
