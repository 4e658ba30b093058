"""Run the reference's own drivers on the HIP path without editing them.

``install()`` registers this package's modules under the module names the
reference imports (``core.varFreeEnergy``, ``core.gradL_Ab``, ``core.fwd_ode``,
``core.bwd_ode``, ``core.likelihood``, ``core.kl0``, ``core.gradEsde_mS``,
``dynamics.sys_*``, ``auxiliary.numerical``, ``auxiliary.momGauss``,
``auxiliary.optimize``, ``auxiliary.collect_obs``, ``auxiliary.initialize_Ab0``).  The
module the reference has and this package does not replace (``core.smoothing``) keeps
coming from the reference tree when it is on ``sys.path``; its ``core/smoothing.py:5-10``
then binds to the functions registered here.
"""
import importlib
import sys
import types

_ALIASES = {
    "core.varFreeEnergy": "vgpa_amd.core.varFreeEnergy",
    "core.gradL_Ab": "vgpa_amd.core.gradL_Ab",
    "core.fwd_ode": "vgpa_amd.core.fwd_ode",
    "core.bwd_ode": "vgpa_amd.core.bwd_ode",
    "core.likelihood": "vgpa_amd.core.likelihood",
    "core.kl0": "vgpa_amd.core.kl0",
    "core.gradEsde_mS": "vgpa_amd.core.gradEsde_mS",
    "dynamics.sys_ornstein_uhlenbeck": "vgpa_amd.dynamics.sys_ornstein_uhlenbeck",
    "dynamics.sys_double_well": "vgpa_amd.dynamics.sys_double_well",
    "dynamics.sys_lorenz_63": "vgpa_amd.dynamics.sys_lorenz_63",
    "dynamics.sys_lorenz_63_ut": "vgpa_amd.dynamics.sys_lorenz_63_ut",
    "dynamics.sys_lorenz_96_ut": "vgpa_amd.dynamics.sys_lorenz_96_ut",
    "auxiliary.numerical": "vgpa_amd.auxiliary.numerical",
    "auxiliary.momGauss": "vgpa_amd.auxiliary.momGauss",
    "auxiliary.optimize": "vgpa_amd.auxiliary.optimize",
    "auxiliary.collect_obs": "vgpa_amd.auxiliary.collect_obs",
    "auxiliary.initialize_Ab0": "vgpa_amd.auxiliary.initialize_Ab0",
}


def _package(name):
    """The package ``name`` as importable now (the reference's, if its tree is on
    sys.path), else an empty stand-in package."""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except ImportError:
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
        return pkg


def install(reference_root=None, only=None):
    """Alias the modules.  ``reference_root``: path of a reference checkout to put on
    sys.path (for its drivers and the modules not replaced).  ``only``: iterable of
    reference module names to alias (default: all).  Returns the aliased names."""
    if reference_root and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    done = []
    for ref_name, ours in _ALIASES.items():
        if only is not None and ref_name not in only:
            continue
        mod = importlib.import_module(ours)
        pkg_name, leaf = ref_name.split(".")
        pkg = _package(pkg_name)
        sys.modules[ref_name] = mod
        setattr(pkg, leaf, mod)
        done.append(ref_name)
    return done


def uninstall():
    """Drop the aliases (modules imported from them keep their bindings)."""
    for ref_name, ours in _ALIASES.items():
        mod = sys.modules.get(ref_name)
        if mod is not None and mod.__name__ == ours:
            del sys.modules[ref_name]
            pkg = sys.modules.get(ref_name.split(".")[0])
            if pkg is not None and getattr(pkg, ref_name.split(".")[1], None) is mod:
                delattr(pkg, ref_name.split(".")[1])
