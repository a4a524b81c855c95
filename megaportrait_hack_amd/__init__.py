"""Import alias: the product package lives in `megaportrait-hack_amd/` (a directory name Python
cannot import directly because of the hyphen); `import megaportrait_hack_amd` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "megaportrait-hack_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
