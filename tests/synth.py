"""Re-export of the package's seeded generators so fixtures/tests share one definition."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "arrowspace-rs_amd"))
from arrowspace_amd.synth import *  # noqa: F401,F403,E402
from arrowspace_amd.synth import mixture, mixture_centres, gaussian_rows, gaussian, queries_from  # noqa: F401,E402
