import sys, logging
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.basicConfig(level=logging.INFO, format="%(message)s")
from llicti_amd.agents.llicti_agent import LLICTIAgent
from llicti_amd.config import default_config
for cont in ("ac", "rans16"):
    print("== container", cont)
    a = LLICTIAgent(default_config(test_data="synthetic:512x768x3", container=cont))
    a.run()
