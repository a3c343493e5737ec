def cached_path(url_or_filename, cache_dir=None, extract_archive=False, force_extract=False):
    return str(url_or_filename)  # local paths only (no network here)
