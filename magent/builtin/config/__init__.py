"""built-in games live in magent_amd.builtin.config"""
