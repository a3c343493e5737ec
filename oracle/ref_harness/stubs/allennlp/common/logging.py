FILE_FRIENDLY_LOGGING = False


def prepare_global_logging(*a, **kw):
    pass
