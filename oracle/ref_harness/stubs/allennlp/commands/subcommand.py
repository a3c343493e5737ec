from allennlp.common import Registrable


class Subcommand(Registrable):
    pass
