from magent_amd.builtin.rule_model import RandomActor

__all__ = ["RandomActor"]
