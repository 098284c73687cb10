"""Alias package: lets the reference's dotted config targets (``ldm.modules...``,
configs/stable-diffusion/v1-finetune-ada.yaml:5,87,108,125) resolve to the MI355X implementation
in ``adaprompt_amd.ldm`` without editing the yaml.  Contains no logic of its own."""
import importlib
import importlib.abc
import importlib.util
import sys


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("ldm."):
            return None
        real = "adaprompt_amd." + fullname
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        mod = importlib.import_module("adaprompt_amd." + spec.name)
        return mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
