def download_file_from_google_drive(*a, **k): raise RuntimeError('no network')
