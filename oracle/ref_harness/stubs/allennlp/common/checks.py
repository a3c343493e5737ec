class ConfigurationError(Exception):
    def __init__(self, message: str = ""):
        super().__init__(message)
        self.message = message


def check_for_gpu(device) -> None:  # allennlp/common/checks.py: raises when a requested GPU is absent
    import torch

    devs = device if isinstance(device, (list, tuple)) else [device]
    for d in devs:
        if isinstance(d, torch.device):
            d = -1 if d.type == "cpu" else (d.index or 0)
        if d is not None and int(d) >= 0 and not torch.cuda.is_available():
            raise ConfigurationError("a GPU was requested but torch.cuda is not available")
