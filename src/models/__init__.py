from .mgfn import *  # noqa: F401,F403
