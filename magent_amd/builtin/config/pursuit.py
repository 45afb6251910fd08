"""built-in game pursuit (see _games.py for the source of the values)"""
from ._games import make


def get_config(map_size):
    return make("pursuit", map_size)
