"""Import alias for the product package.

The product lives in ``quadruped-trajectory-optimization-stack_amd/`` (the directory name the
project layout prescribes; hyphens make it un-importable by name), so this stub maps it onto the
importable name ``qtos_amd``: ``import qtos_amd.planner`` loads
``quadruped-trajectory-optimization-stack_amd/planner.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "quadruped-trajectory-optimization-stack_amd")
__path__.insert(0, _PKG_DIR)
PKG_DIR = _PKG_DIR
