from .data_loaders import DataLoader, TensorDict  # noqa: F401
from .dataset_readers.dataset_reader import DatasetReader  # noqa: F401
from .fields import Field  # noqa: F401
from .fields.text_field import TextFieldTensors  # noqa: F401
from .instance import Batch, Instance, allennlp_collate  # noqa: F401
from .token_indexers import TokenIndexer  # noqa: F401
from .tokenizers import Token, Tokenizer  # noqa: F401
from .vocabulary import Vocabulary  # noqa: F401
from . import instance  # noqa: F401
