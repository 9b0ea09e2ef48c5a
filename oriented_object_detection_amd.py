"""Import shim: the product package lives in the directory `oriented-object-detection_amd/` (not a valid Python
identifier), so this module gives it an importable name: `import oriented_object_detection_amd as ood`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "oriented-object-detection_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
