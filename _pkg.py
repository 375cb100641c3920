"""Loads the product package, whose directory name (stellar-random-walk_amd) is not a Python identifier."""
import importlib.util
import os
import sys

NAME = "stellar_random_walk_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stellar-random-walk_amd")
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
