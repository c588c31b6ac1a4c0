"""`from agents import *` gives the agent classes by name, as the reference's package does (agents/__init__.py:6-10;
main.py:29-33 looks `config.agent` up in globals())."""
from .llicti_agent import LLICTIAgent  # noqa: F401

__all__ = ["LLICTIAgent"]
