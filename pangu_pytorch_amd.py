"""Import alias: the package directory is named `pangu-pytorch_amd/` (not a valid Python identifier), so this
module turns itself into that package: `import pangu_pytorch_amd; pangu_pytorch_amd.PanguModel`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pangu-pytorch_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
