"""Dataset & model registry, mirror of the reference's register.py (register.py:16-55).
Importing it instantiates `dataset = Loader(world.config)` like the reference."""
from . import world
from . import model
from .world import cprint  # noqa: F401
from .dataloader import Loader

print("comment:", world.comment)
print("tensorboard:", world.tensorboard)
print("LOAD:", world.LOAD)
print("Weight path:", world.PATH)

try:
    dataset = Loader(world.config)
except TypeError:
    dataset = Loader()

MODELS = {}
if hasattr(model, 'PureMF'):
    MODELS['mf'] = model.PureMF
if hasattr(model, 'LightGCN'):
    MODELS['lgn'] = model.LightGCN

if world.model_name not in MODELS:
    raise ValueError(
        f"Requested model '{world.model_name}' is not available. "
        f"Available models: {list(MODELS.keys())}. "
        f"Ensure model.py defines the class, or run with --model one of {list(MODELS.keys())}.")
