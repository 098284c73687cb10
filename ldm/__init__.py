"""Alias package: lets the reference's dotted config targets (``ldm.modules...``,
configs/stable-diffusion/v1-finetune-ada.yaml:5,87,108,125) resolve to the MI355X implementation
in ``adaprompt_amd.ldm`` without editing the yaml.  Contains no logic of its own.

Resolution order for ``ldm.<name>`` (this repo AHEAD of the reference checkout on ``sys.path``):
  1. ``adaprompt_amd.ldm.<name>`` when this repo mirrors the module -- the SAME module object is
     returned (no second copy of the classes), its own ``__spec__`` / ``__name__`` left untouched;
  2. otherwise the reference's own ``ldm/<name>`` (``ldm.modules.embedding_manager``,
     ``ldm.modules.encoders.modules``, ``ldm.data.*``, ``ldm.modules.x_transformer`` ...):
     every later ``sys.path`` entry (and ``$ADAPROMPT_REFERENCE_ROOT``) holding an ``ldm/`` directory is a
     fall-through root, so the un-mirrored boundary callees keep importing as they always did.
The finder answers for every ``ldm.*`` name whatever ``sys.modules['ldm']`` currently is -- the reference's
``adaface/subj_basis_generator.py:23`` rebinds it to the ``adaface`` package at import time."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_roots():
    """directories named ``ldm`` of a reference checkout: ``$ADAPROMPT_REFERENCE_ROOT/ldm`` first, then every
    ``sys.path`` entry (other than this repo) that holds one."""
    roots = []
    env = os.environ.get("ADAPROMPT_REFERENCE_ROOT")
    cands = ([env] if env else []) + [p or os.getcwd() for p in sys.path]
    for base in cands:
        d = os.path.join(base, "ldm")
        try:
            if os.path.isdir(d) and not os.path.samefile(d, _HERE) and d not in roots:
                roots.append(d)
        except OSError:
            continue
    return roots


_REF_MODULES = {}


def reference_module(fullname):
    """the reference's OWN copy of ``ldm.<...>`` (a module this repo mirrors only in part, e.g. ``ldm.util``), loaded
    from the first fall-through root under a private name; None when no reference checkout is on the path.  The
    mirrored module forwards the names it does not define to it (module ``__getattr__``), so
    ``from ldm.util import get_clip_tokens_for_string`` in the reference's embedding manager keeps working."""
    if fullname in _REF_MODULES:
        return _REF_MODULES[fullname]
    mod = None
    rel = fullname.split(".")[1:]
    for root in reference_roots():
        path = os.path.join(root, *rel) + ".py"
        if os.path.isfile(path):
            spec = importlib.util.spec_from_file_location("_adaprompt_reference." + fullname, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                del sys.modules[spec.name]
                raise
            break
    _REF_MODULES[fullname] = mod
    return mod


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("ldm."):
            return None
        real = "adaprompt_amd." + fullname
        try:
            have = importlib.util.find_spec(real) is not None
        except (ModuleNotFoundError, ValueError):
            have = False
        if have:
            mod = importlib.import_module(real)
            return importlib.util.spec_from_loader(fullname, _ExistingLoader(mod), is_package=hasattr(mod, "__path__"))
        rel = fullname.split(".")[1:]
        for root in reference_roots():
            base = os.path.join(root, *rel)
            if os.path.isfile(os.path.join(base, "__init__.py")):
                return importlib.util.spec_from_file_location(fullname, os.path.join(base, "__init__.py"),
                                                              submodule_search_locations=[base])
            if os.path.isfile(base + ".py"):
                return importlib.util.spec_from_file_location(fullname, base + ".py")
            if os.path.isdir(base):          # namespace directory of the reference (it ships no __init__.py files)
                spec = importlib.machinery.ModuleSpec(fullname, None, is_package=True)
                spec.submodule_search_locations = [base]
                return spec
        return None


class _ExistingLoader(importlib.abc.Loader):
    """loader for a module that already exists: ``create_module`` returns it, ``exec_module`` does nothing.  The
    module's own ``__spec__`` / ``__loader__`` / ``__name__`` are restored after importlib's attribute pass."""

    def __init__(self, mod):
        self.mod = mod
        self.saved = {k: getattr(mod, k, None) for k in ("__spec__", "__loader__", "__package__", "__name__",
                                                         "__file__", "__path__") if hasattr(mod, k)}

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        for k, v in self.saved.items():
            try:
                setattr(module, k, v)
            except (AttributeError, TypeError):
                pass


sys.meta_path.insert(0, _AliasFinder())
# a name that survives the reference's rebinding of sys.modules['ldm'] (adaface/subj_basis_generator.py:23)
sys.modules["_adaprompt_ldm_alias"] = sys.modules[__name__]
