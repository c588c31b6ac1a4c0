"""llicti_amd -- MI355X-native encode/decode hot path of the LLICTI learned lossless image codec.

Only what the path needs lives here: `csrc/` (hand-written gfx950 HIP kernels behind a C-ABI,
`include/llicti_hip.h`), a ctypes binding (`_lib`), the canonical weight packing (`weights`), and the
host-side mirror of the reference's interface for this path (`graphs.models.LLICTI_nets.LLICTI`,
`agents.llicti_agent.LLICTIAgent`).  There is no CPU fallback: without the HIP library every
compute entry point raises.
"""
__version__ = "0.1.0"
