"""Name -> predictor registry with the same protocol as the reference's
pysgg/utils/registry.py:9-45 (`registry.ROI_RELATION_PREDICTOR`, modeling/registry.py:16):
a dict whose `register(name)` works as a decorator or as a direct call.

The reference's `register` asserts the name is not taken (utils/registry.py:4-6), so a drop-in
replacement cannot simply register itself next to the original; `install()` overwrites the two
VETO entries of an existing pysgg registry instead ("Registration").
"""


class Registry(dict):
    def register(self, name, module=None):
        if module is not None:
            self._add(name, module)
            return module

        def deco(fn):
            self._add(name, fn)
            return fn

        return deco

    def _add(self, name, module):
        if name in self:
            raise AssertionError("%r is already registered" % (name,))
        self[name] = module


ROI_RELATION_PREDICTOR = Registry()


def make_roi_relation_predictor(cfg, in_channels):
    """Same lookup as roi_relation_predictors.py:4152-4154."""
    func = ROI_RELATION_PREDICTOR[cfg.MODEL.ROI_RELATION_HEAD.PREDICTOR]
    return func(cfg, in_channels)


def install(target_registry=None):
    """Point the reference's registry at the MI355X predictors.

    With no argument, imports `pysgg.modeling.registry` (the reference must be importable).
    Returns the registry that was patched."""
    from . import predictor  # noqa: F401  (registers into ROI_RELATION_PREDICTOR)
    if target_registry is None:
        from pysgg.modeling import registry as ref_registry  # type: ignore
        target_registry = ref_registry.ROI_RELATION_PREDICTOR
    for name in ("VETOPredictor", "VETOPredictor_MEET"):
        dict.__setitem__(target_registry, name, ROI_RELATION_PREDICTOR[name])
    return target_registry
