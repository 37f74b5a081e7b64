"""Importable alias of the package directory
``applying-slowfast-networks-to-video-object-segmentation_amd/`` (whose name, fixed by the
project layout, is not a valid Python identifier).  ``import sfvos_amd`` executes that
directory's ``__init__.py`` with this module as the package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'applying-slowfast-networks-to-video-object-segmentation_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _f
