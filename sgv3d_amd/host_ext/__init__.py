"""Host-only compiled helpers (no device code, no arithmetic of the path).  ``stamp()`` returns the ``tensor_stamp`` function of
the compiled module ``stamp_ext`` (built by this directory's Makefile / ``__graft_entry__.build()``), or None when it has not been
built: callers then run their Python loop, which computes the same value."""
import glob
import importlib.util
import os

_FN = False


def stamp():
    global _FN
    if _FN is False:
        _FN = None
        if not os.environ.get("SGV3D_NO_HOST_EXT"):
            found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp_ext*.so")))
            if found:
                try:
                    import torch  # noqa: F401   (the module links libtorch)
                    spec = importlib.util.spec_from_file_location("stamp_ext", found[-1])
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    _FN = mod.tensor_stamp
                except Exception:          # an unloadable helper is not an error: the Python loop is the definition
                    _FN = None
    return _FN
