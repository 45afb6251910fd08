"""built-in game double_attack (see _games.py for the source of the values)"""
from ._games import make


def get_config(map_size):
    return make("double_attack", map_size)
