"""Command-line helper shared by the example scripts."""
import argparse

_TRUE, _FALSE = {"1", "t", "true", "y", "yes", "on"}, {"0", "f", "false", "n", "no", "off"}


def str2bool(val):
    """argparse `type=` for flags given as words: `--log false`, `--gui yes`.  Booleans pass through."""
    if isinstance(val, bool):
        return val
    word = str(val).strip().lower()
    if word in _TRUE or word in _FALSE:
        return word in _TRUE
    raise argparse.ArgumentTypeError(f"expected one of {sorted(_TRUE | _FALSE)}, got {val!r}")
