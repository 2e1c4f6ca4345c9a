"""Import-time backend resolver (reference: src/MuyGPyS/_src/util.py:9-32).

``_collect_implementation("muygpys_amd._src.gp.tensors", "_F2", ...)`` imports
``<package>.<backend>`` and returns the named attributes; each family ``__init__`` binds
them to module-level names exactly like the reference does.
"""

from muygpys_amd.config import config


def _collect_implementation(package, *funcs):
    if config.state.backend == "hip":
        return _collect_functions(package + ".hip", *funcs)
    raise ValueError(f'MuyGPyS backend is in bad state "{config.state.backend}"')


def _collect_functions(package, *funcs):
    return tuple(getattr(__import__(package, fromlist=[f]), f) for f in funcs)


def export_backend(module_name: str, namespace: dict, names: str):
    """Bind the space-separated backend function ``names`` of family ``module_name`` into
    ``namespace`` (a family package's globals) from its ``<backend>`` submodule, and return them
    as the package's ``__all__``.  One call per family ``__init__`` replaces the reference's
    hand-written tuple assignment around ``_collect_implementation``."""
    wanted = names.split()
    for name, fn in zip(wanted, _collect_implementation(module_name, *wanted)):
        namespace[name] = fn
    return wanted


def auto_str(klass):
    """Same contract as util.py:35-45: print public members."""

    def __str__(self):
        members = ", ".join(f"{k}={v}" for k, v in vars(self).items() if not k.startswith("_"))
        return f"{type(self).__name__}({members})"

    klass.__str__ = __str__
    return klass
