"""Importable alias of the product package, whose directory name
(``optimized-sparse-retrieval-for-high-performance-rag-pipelines_amd``) is not a Python identifier.
``import sparse_rx`` executes that package's ``__init__`` with its directory as the package path."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "optimized-sparse-retrieval-for-high-performance-rag-pipelines_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
del _f
