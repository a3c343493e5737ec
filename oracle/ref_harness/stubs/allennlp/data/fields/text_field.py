from typing import Dict

import torch

from . import TextField  # noqa: F401

TextFieldTensors = Dict[str, Dict[str, torch.Tensor]]
