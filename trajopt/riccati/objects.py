"""trajopt.riccati.objects (/root/reference/trajopt/riccati/objects.py)."""
from trajopt_amd.riccati.objects import *  # noqa: F401,F403
