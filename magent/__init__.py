"""`import magent` -- the import name the reference's scripts use, aliased onto the MI355X engine package.

Nothing lives here: every name forwards to magent_amd (engine wrapper, model hosting, helpers, the PyTorch DQN that
stands where the reference keeps its TensorFlow one)."""
import sys

import magent_amd
from magent_amd import gridworld, model, utility

sys.modules[__name__ + ".gridworld"] = gridworld
sys.modules[__name__ + ".model"] = model
sys.modules[__name__ + ".utility"] = utility

GridWorld = gridworld.GridWorld
ProcessingModel = model.ProcessingModel
round = utility.rec_round
