class Tqdm:
    @staticmethod
    def tqdm(iterable=None, *a, **kw):
        class _Bar:
            def __init__(self, it):
                self._it = it

            def __iter__(self):
                return iter(self._it)

            def set_description(self, *a, **kw):
                pass

        return _Bar(iterable)
