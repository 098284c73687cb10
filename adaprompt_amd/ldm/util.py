"""Config plumbing of the drop-in boundary: ``instantiate_from_config`` with the reference's
``{target: dotted.path, params: {...}}`` convention (reference ldm/util.py:104-111, 142-147).
Dotted targets of the reference yaml (``ldm.modules.diffusionmodules.openaimodel.UNetModel``,
``ldm.models.autoencoder.AutoencoderKL`` ...) resolve to this package's MI355X implementations,
either through the top-level ``ldm`` alias package of this repo or by the rewrite below."""
import importlib

_PREFIX = "adaprompt_amd."


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if module.startswith("ldm."):
        module = _PREFIX + module
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config, **kwargs):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()), **kwargs)


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d
