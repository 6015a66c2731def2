# Real package body; imported as ``troute_amd`` (see ../troute_amd/__init__.py).
