"""Tool / test plumbing (NOT the product binding): honour the C12381_LIB convention of the A/B scripts, the clock probe and the variant
tests — `import tools.libsel` before the first Context selects that build of the C ABI through crypto12381_amd.capi.use_library().
The binding itself (crypto12381_amd/capi.py) reads no environment variable."""
import os

from crypto12381_amd import capi

_path = os.environ.get("C12381_LIB")
if _path:
    capi.use_library(_path)
