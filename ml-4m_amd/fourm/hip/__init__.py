"""gfx950 kernel bindings (``_lib``), tensor front end (``ops``), step executor (``engine``) and
inference helpers (``functional``)."""
