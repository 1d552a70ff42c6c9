"""Import alias: ``dn-splatter_amd/`` (the package directory the build contract names) is not a
valid Python identifier, so ``import dn_splatter_amd`` resolves here and runs that directory's
``__init__.py`` with ``__path__`` pointing at it.  No code lives in this shim."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "dn-splatter_amd"
__path__ = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
