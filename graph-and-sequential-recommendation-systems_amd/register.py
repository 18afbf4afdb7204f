"""Dataset and model registry -- the role of the reference's register.py (register.py:16-55):
importing this module builds `dataset` from world.config and exposes MODELS[name] -> class."""
from . import model as _model
from . import world
from .dataloader import Loader
from .world import cprint  # noqa: F401  (re-exported like the reference does)

for _label, _value in (("comment", world.comment), ("tensorboard", world.tensorboard),
                       ("LOAD", world.LOAD), ("Weight path", world.PATH)):
    print(f"{_label}: {_value}")

# Loader(config) is the constructor of this package; forks that read world.config themselves take no argument
try:
    dataset = Loader(world.config)
except TypeError:
    dataset = Loader()

MODELS = {key: getattr(_model, cls) for key, cls in (("mf", "PureMF"), ("lgn", "LightGCN")) if hasattr(_model, cls)}

if world.model_name not in MODELS:
    raise ValueError(f"model '{world.model_name}' is not registered; choose one of {sorted(MODELS)} with --model")
