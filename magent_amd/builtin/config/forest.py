"""built-in game forest (see _games.py for the source of the values)"""
from ._games import make


def get_config(map_size):
    return make("forest", map_size)
