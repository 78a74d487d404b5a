"""Import alias: the product package lives in `echo-tts_amd/` (hyphenated, per the repo layout);
this shim makes it importable as `echo_tts_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "echo-tts_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
