class PytorchSeq2SeqWrapper:  # import surface only (model_memory.py:15)
    pass


class LstmSeq2SeqEncoder:
    pass
