"""Tiny host helpers that `ba_core` imports from the reference's loader (ref:bundle_adjust/loader.py:23-24, 27-40)."""
import sys


def flush_print(*args, **kwargs):
    print(*args, **kwargs)
    sys.stdout.flush()


def display_dict(d):
    """Pretty-print a configuration dict, one `key: value` per line."""
    width = max((len(str(k)) for k in d), default=0)
    for k, v in d.items():
        print("    - {}: {}".format(str(k).ljust(width), v))
    print("")
