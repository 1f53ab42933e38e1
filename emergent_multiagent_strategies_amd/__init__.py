"""Import alias.  The product package lives in ``emergent-multiagent-strategies_amd/``
(a directory name Python cannot import directly); this shim makes it importable as
``emergent_multiagent_strategies_amd`` by pointing the package path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "emergent-multiagent-strategies_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
