"""Diagnostic switches for the tools: `PSX_SWITCHES="no_dif=1 stamp_round=2" python tools/x.py` (read HERE, by the tool, and
handed to psx_debug_switch -- the library itself reads nothing from the environment) or apply("no_dual=1")."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def apply(spec=None):
    from paresis_amd import ops
    spec = os.environ.get("PSX_SWITCHES", "") if spec is None else spec
    for item in spec.replace(",", " ").split():
        name, _, val = item.partition("=")
        ops.debug_switch(name, int(val) if val else 1)
    act = ops.debug_switches_active()
    if act:
        sys.stderr.write("diagnostic switches: %s\n" % act)
    return act
