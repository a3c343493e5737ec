from .metric import Metric


class Auc(Metric):  # import surface only (model_memory.py:20)
    pass
