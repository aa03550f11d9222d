"""Import alias: the package directory is named `visual-context-attentional-gan_amd` (not a valid Python
identifier), so `import vcagan_amd` loads that directory as the package `vcagan_amd`.
`vcagan_amd.models.generator` etc. mirror the reference's `src.models.*` modules."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'visual-context-attentional-gan_amd')
_spec = _u.spec_from_file_location('vcagan_amd', _os.path.join(_d, '__init__.py'), submodule_search_locations=[_d])
_mod = _u.module_from_spec(_spec)
_sys.modules['vcagan_amd'] = _mod
_spec.loader.exec_module(_mod)
