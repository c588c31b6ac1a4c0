"""Import-path alias: make the reference's own import lines resolve to the MI355X hot path.

The reference's entry point does `from agents import *` and looks the agent class up by name (main.py:4, :29-33); its
agent imports `from graphs.models.LLICTI_nets import LLICTI`, `from graphs.losses.rate_dist import ...`,
`from loggers.rate import RateLogger` (agents/llicti_agent.py:1-12).  `install()` registers this package's modules under
those top-level names, so a maintainer switches the encode/decode path by adding two lines in front of the reference's
imports:

    import llicti_amd.dropin
    llicti_amd.dropin.install()
    from agents import *                                     # -> llicti_amd.agents (LLICTIAgent)
    from graphs.models.LLICTI_nets import LLICTI             # -> llicti_amd.graphs.models.LLICTI_nets

Nothing is copied or monkey-patched inside the reference tree; a name that is already imported from somewhere else is
refused (mixing the two implementations silently would be worse than failing).
"""
from __future__ import annotations

import importlib
import sys

ALIASES = {
    "graphs": "llicti_amd.graphs",
    "graphs.models": "llicti_amd.graphs.models",
    "graphs.models.LLICTI_nets": "llicti_amd.graphs.models.LLICTI_nets",
    "graphs.losses": "llicti_amd.graphs.losses",
    "graphs.losses.rate_dist": "llicti_amd.graphs.losses.rate_dist",
    "agents": "llicti_amd.agents",
    "agents.llicti_agent": "llicti_amd.agents.llicti_agent",
    "loggers": "llicti_amd.loggers",
    "loggers.rate": "llicti_amd.loggers.rate",
}


def install():
    """Register the aliases in sys.modules (idempotent).  Returns the alias -> module dict."""
    out = {}
    for alias, real in ALIASES.items():
        mod = importlib.import_module(real)
        have = sys.modules.get(alias)
        if have is not None and have is not mod:
            raise ImportError(f"'{alias}' is already imported from {getattr(have, '__file__', have)!r}; "
                              "call llicti_amd.dropin.install() before the reference's own imports")
        sys.modules[alias] = mod
        out[alias] = mod
    return out


def uninstall():
    for alias, real in ALIASES.items():
        if sys.modules.get(alias) is sys.modules.get(real):
            sys.modules.pop(alias, None)
