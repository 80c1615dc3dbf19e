"""Child-process entry of the exchange probe (see stats_exchange._probe_main): python tools/xchg_probe.py <file-store path>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value.stats_exchange import _probe_main  # noqa: E402

sys.exit(_probe_main(sys.argv[1]))
